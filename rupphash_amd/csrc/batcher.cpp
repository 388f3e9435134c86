// batcher.cpp -- thread-safe single-image entry point with internal batching.
//
// The reference calls generate_pdq_features once per file from many rayon workers
// (/root/reference/src/scanner.rs:1202-1205, :1410): blocking calls, one decoded image each, of whatever size the file had.
// One image per GPU call would spend its time in launch and PCIe latency, so concurrent callers are coalesced here, and the
// transfers of consecutive batches are pipelined:
//
//   * five pipeline slots, each with its own HIP stream, pinned host staging and device buffers; at most three batches are in
//     flight (the H2D of one overlaps the kernels / D2H of the others) while another slot fills and one is read out;
//   * a caller joins the open batch and copies its pixels into the slot's pinned staging itself (outside the lock: the copies of
//     different callers run in parallel on their own cores).  The batch's pixels then cross PCIe in ONE transfer: a 0.8 MB copy
//     costs 23-46 us (17-35 GB/s), 6 MB and more run at 52-57 GB/s (tools/h2d_rate.cpp), so per-caller transfers lose;
//   * a batch goes as soon as nobody is still copying into it AND the pipeline has room -- no timer: under load the batch size
//     adapts to the arrival rate (callers that arrive while the pipeline is full accumulate), an isolated caller is served at
//     once.  `max_wait_us` (default 0) optionally holds a non-full batch back for more callers;
//   * a batch takes at most a third (half, below six callers) of the callers that were recently inside at once: blocking callers
//     that all sit in ONE batch march in lockstep (copy, transfer, kernel, wake, copy, ...) and nothing overlaps; smaller groups
//     fall out of step after the first round and then one group copies while another's pixels cross PCIe and a third one's
//     kernel runs (measured: three groups beat two by 10-17 % from 16 callers up and tie below; four are no better);
//   * a batch may hold images of different geometry (a scan of mixed-size photos): they share the transfer and the
//     synchronisation, and every run of consecutive images with one geometry is one rph_pdq_hash_batch_dev call on the slot's stream.
// Whichever caller finds the batch ready becomes its leader, runs it and wakes the others.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <vector>

#include "rph_internal.h"

namespace {

constexpr int kSlots = 5;                          // pipeline depth: up to three batches in flight + one filling + one being read out
constexpr uint32_t kMaxInflight = 3;
constexpr size_t kSlotPixelBytes = (size_t)64 << 20;  // pinned staging per slot (one larger image still gets a slot of its own size)
static const bool kTrace = getenv("RPH_BATCHER_TRACE") != nullptr;  // per-batch phase times on stderr when the context goes
constexpr uint32_t kSlotImages = 4096;             // result records per slot
constexpr size_t kRecordBytes = 32 + 4 + 1 + 1024 + 16;  // hash, quality, valid, coefficients (+ alignment slack)

double g_ev[4];  // trace: summed stage times on the GPU timeline (ms; updated by batch leaders, read at teardown)

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *h_px = nullptr, *h_out = nullptr;  // pinned: pixels in, result records out
    void *d_px = nullptr, *d_out = nullptr;
    size_t px_cap = 0;
    bool busy = false;  // owned by a batch (filling, in flight, or results still being read)

    bool ensure(size_t px_bytes)
    {
        if (!stream && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        if (!h_out) {
            if (hipHostMalloc((void **)&h_out, (size_t)kSlotImages * kRecordBytes) != hipSuccess || hipMalloc(&d_out, (size_t)kSlotImages * kRecordBytes) != hipSuccess)
                return false;
        }
        if (px_cap < px_bytes) {  // only ever called on an idle slot
            if (h_px) (void)hipHostFree(h_px);
            if (d_px) (void)hipFree(d_px);
            h_px = nullptr;
            d_px = nullptr;
            px_cap = 0;
            if (hipHostMalloc((void **)&h_px, px_bytes) != hipSuccess || hipMalloc(&d_px, px_bytes) != hipSuccess) return false;
            px_cap = px_bytes;
        }
        return true;
    }
    void release()
    {
        for (void *p : {(void *)h_px, (void *)h_out})
            if (p) (void)hipHostFree(p);
        for (void *p : {d_px, d_out})
            if (p) (void)hipFree(p);
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        *this = Slot();
    }
};

struct Item {
    uint32_t w, h, channels;
    size_t off;  // byte offset of the packed image in the slot's staging
};

struct Batch {
    Slot *slot = nullptr;
    std::vector<Item> items;
    size_t bytes = 0;
    uint32_t copying = 0, readers = 0;
    uint32_t cap = 1;      // images it accepts
    bool want_coeffs = false;  // some caller asked for the coefficients (1 KB per image more on the way back)
    // the results of n images come back in ONE transfer: [n x 32 hash][n x f32 quality][n valid][pad to 16][n x 1024 coefficients]
    size_t o_q = 0, o_v = 0, o_c = 0;
    bool closed = false;   // no further joins
    bool running = false;  // a leader has taken it
    bool done = false;
    int status = RPH_OK;
    std::chrono::steady_clock::time_point last_join, t_open, t_copied, t_done;
    std::condition_variable cv;  // its callers wait here: done (all), or a chance to lead it (one)
};

struct Batcher {
    std::mutex mu;
    std::condition_variable cv_slot;  // a slot became free
    std::shared_ptr<Batch> open;
    std::vector<std::shared_ptr<Batch>> live;  // batches that exist (<= kSlots): filling, waiting for the pipeline, in flight
    Slot slots[kSlots];
    uint32_t inflight = 0;
    uint32_t inside = 0;    // callers currently in rph_pdq_hash_one
    double crowd = 1.0;     // slowly decaying maximum of `inside`
    uint32_t max_batch = 256;
    uint32_t max_wait_us = 0;
    uint64_t n_batches = 0, n_images = 0;
    double t_fill = 0, t_wait = 0, t_run = 0, t_read = 0;  // trace: open -> last copy done, -> leader starts, run_batch, done -> slot free
    ~Batcher()
    {
        if (kTrace && n_batches)
            fprintf(stderr, "batcher: %llu batches, %.1f images each; per batch: fill %.0f us, wait for pipeline %.0f us, run %.0f us, read out %.0f us\n",
                    (unsigned long long)n_batches, (double)n_images / n_batches, 1e6 * t_fill / n_batches, 1e6 * t_wait / n_batches, 1e6 * t_run / n_batches,
                    1e6 * t_read / n_batches);
        if (kTrace && n_batches)
            fprintf(stderr, "         on the GPU timeline per batch: H2D %.0f us, kernels %.0f us, D2H %.0f us\n", 1e3 * g_ev[0] / n_batches, 1e3 * g_ev[1] / n_batches,
                    1e3 * g_ev[2] / n_batches);
        for (Slot &s : slots) s.release();
    }
};

std::mutex g_registry_mu;
std::map<rph_ctx *, std::unique_ptr<Batcher>> g_registry;

Batcher &batcher_of(rph_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_registry_mu);
    auto &slot = g_registry[ctx];
    if (!slot) slot.reset(new Batcher());
    return *slot;
}

inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

int run_batch(rph_ctx *ctx, Batch &b)
{
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    Slot &s = *b.slot;
    hipEvent_t ev[4] = {};
    if (kTrace) {
        for (auto &e : ev) (void)hipEventCreate(&e);
        (void)hipEventRecord(ev[0], s.stream);
    }
    const uint32_t n = (uint32_t)b.items.size();
    b.o_q = (size_t)n * 32;
    b.o_v = b.o_q + (size_t)n * 4;
    b.o_c = align16(b.o_v + n);
    uint8_t *d_out = (uint8_t *)s.d_out;
    RPH_HIP_CHECK(hipMemcpyAsync(s.d_px, s.h_px, b.bytes, hipMemcpyHostToDevice, s.stream));
    if (kTrace) (void)hipEventRecord(ev[1], s.stream);
    for (uint32_t first = 0; first < n;) {  // one launch sequence per run of equal geometry (packed back to back at a uniform stride)
        const Item &a = b.items[first];
        const size_t image_bytes = (size_t)a.w * a.h * a.channels, stride = align16(image_bytes);
        uint32_t m = 1;
        while (first + m < n && b.items[first + m].w == a.w && b.items[first + m].h == a.h && b.items[first + m].channels == a.channels &&
               b.items[first + m].off == a.off + (size_t)m * stride)
            m++;
        const int rc = rph_pdq_hash_batch_dev(ctx, (const uint8_t *)s.d_px + a.off, m, a.w, a.h, a.channels, (size_t)a.w * a.channels, stride,
                                              d_out + (size_t)first * 32, d_out + b.o_q + (size_t)first * 4,
                                              b.want_coeffs ? d_out + b.o_c + (size_t)first * 1024 : nullptr, nullptr, d_out + b.o_v + first, s.stream);
        if (rc != RPH_OK) {
            (void)hipStreamSynchronize(s.stream);
            return rc;
        }
        first += m;
    }
    if (kTrace) (void)hipEventRecord(ev[2], s.stream);
    RPH_HIP_CHECK(hipMemcpyAsync(s.h_out, s.d_out, b.want_coeffs ? b.o_c + (size_t)n * 1024 : b.o_v + n, hipMemcpyDeviceToHost, s.stream));
    if (kTrace) (void)hipEventRecord(ev[3], s.stream);
    RPH_HIP_CHECK(hipStreamSynchronize(s.stream));
    if (kTrace) {
        float ms = 0;
        for (int i = 0; i < 3; i++) {
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            g_ev[i] += ms;
        }
        for (auto &e : ev) (void)hipEventDestroy(e);
    }
    return RPH_OK;
}

}  // namespace

void rph_batcher_forget(rph_ctx *ctx)
{
    std::unique_ptr<Batcher> victim;
    {
        std::lock_guard<std::mutex> lock(g_registry_mu);
        auto it = g_registry.find(ctx);
        if (it != g_registry.end()) {
            victim = std::move(it->second);
            g_registry.erase(it);
        }
    }
    if (victim) (void)hipSetDevice(ctx->device);  // the slots are released on this device by ~Batcher
}

extern "C" int rph_pdq_batcher_config(rph_ctx *ctx, uint32_t max_batch, uint32_t max_wait_us)
{
    return rph_guarded("rph_pdq_batcher_config", [&]() -> int {
        if (!ctx || max_batch == 0 || max_batch > 65536) return RPH_ERR_INVALID_ARG;
        Batcher &B = batcher_of(ctx);
        std::lock_guard<std::mutex> lock(B.mu);
        B.max_batch = std::min(max_batch, kSlotImages);
        B.max_wait_us = max_wait_us;
        return RPH_OK;
    });
}

extern "C" int rph_pdq_batcher_stats(rph_ctx *ctx, uint64_t *n_batches, uint64_t *n_images)
{
    return rph_guarded("rph_pdq_batcher_stats", [&]() -> int {
        if (!ctx) return RPH_ERR_INVALID_ARG;
        Batcher &B = batcher_of(ctx);
        std::lock_guard<std::mutex> lock(B.mu);
        if (n_batches) *n_batches = B.n_batches;
        if (n_images) *n_images = B.n_images;
        return RPH_OK;
    });
}

extern "C" int rph_pdq_hash_one(rph_ctx *ctx, const uint8_t *px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride,
                                uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *valid_out)
{
    return rph_guarded("rph_pdq_hash_one", [&]() -> int {
        if (!ctx || !px || !hash32_out || (channels != 1 && channels != 3 && channels != 4) || row_stride < (size_t)w * channels || w == 0 ||
            h == 0) {
            rph_set_error("rph_pdq_hash_one: invalid argument");
            return RPH_ERR_INVALID_ARG;
        }
        Batcher &B = batcher_of(ctx);
        const size_t line = (size_t)w * channels, image_bytes = line * h, need = align16(image_bytes);
        std::shared_ptr<Batch> b;
        uint32_t slot_index = 0;
        size_t off = 0;
        std::unique_lock<std::mutex> lock(B.mu);
        struct Inside {
            Batcher &B;
            explicit Inside(Batcher &b) : B(b)
            {
                B.inside++;
                B.crowd = std::max((double)B.inside, B.crowd * 0.98);
            }
            ~Inside() { B.inside--; }  // (runs with the lock held: every exit path below re-locks first)
        } inside_guard(B);
        // ---- join the open batch, or open one on a free slot
        for (;;) {
            if (B.open && !B.open->closed && B.open->items.size() < B.open->cap && B.open->bytes + need <= B.open->slot->px_cap) {
                b = B.open;
                break;
            }
            if (B.open && !B.open->closed) B.open->closed = true;  // full: it goes as soon as its copies are done and a slot is free
            Slot *free_slot = nullptr;
            for (Slot &s : B.slots)
                if (!s.busy) {
                    free_slot = &s;
                    break;
                }
            if (free_slot) {
                if (hipSetDevice(ctx->device) != hipSuccess || !free_slot->ensure(std::max(kSlotPixelBytes, need))) {
                    rph_set_error("rph_pdq_hash_one: staging allocation failed (%zu bytes)", std::max(kSlotPixelBytes, need));
                    return RPH_ERR_OOM;
                }
                free_slot->busy = true;
                b = std::make_shared<Batch>();
                b->slot = free_slot;
                const double groups = B.crowd >= 6.0 ? 3.0 : 2.0;
                b->cap = std::min<uint32_t>(B.max_batch, std::max<uint32_t>(1, (uint32_t)((B.crowd + groups - 1.0) / groups)));
                b->t_open = std::chrono::steady_clock::now();
                b->items.reserve(b->cap);
                B.open = b;
                B.live.push_back(b);
                break;
            }
            B.cv_slot.wait(lock);  // all slots taken: wait for one to come back
        }
        slot_index = (uint32_t)b->items.size();
        off = b->bytes;
        b->items.push_back(Item{w, h, channels, off});
        b->bytes += need;
        b->copying++;
        b->readers++;
        b->last_join = std::chrono::steady_clock::now();
        if (coeffs_out) b->want_coeffs = true;
        if (b->items.size() >= b->cap) b->closed = true;
        lock.unlock();
        // ---- copy this caller's pixels into its place (outside the lock: copies of different callers run in parallel)
        {
            uint8_t *dst = b->slot->h_px + off;
            if (row_stride == line)
                std::memcpy(dst, px, image_bytes);
            else
                for (uint32_t y = 0; y < h; y++) std::memcpy(dst + (size_t)y * line, px + (size_t)y * row_stride, line);
        }
        lock.lock();
        b->copying--;
        if (kTrace) b->t_copied = std::chrono::steady_clock::now();
        // ---- wait for the batch; whoever finds it ready to go runs it
        while (!b->done) {
            if (!b->running && b->copying == 0 && B.inflight < kMaxInflight) {
                bool go = b->closed;  // a full batch goes at once
                if (!go) {
                    const auto linger_until = b->last_join + std::chrono::microseconds(B.max_wait_us);
                    if (B.max_wait_us == 0 || std::chrono::steady_clock::now() >= linger_until)
                        go = true;
                    else {
                        b->cv.wait_until(lock, linger_until);
                        continue;
                    }
                }
                if (go) {
                    b->closed = true;
                    b->running = true;
                    if (B.open == b) B.open.reset();  // later callers start a new batch
                    B.inflight++;
                    B.n_batches++;
                    B.n_images += b->items.size();
                    const auto t_go = std::chrono::steady_clock::now();
                    lock.unlock();
                    const int rc = run_batch(ctx, *b);
                    lock.lock();
                    if (kTrace) {
                        b->t_done = std::chrono::steady_clock::now();
                        B.t_fill += std::chrono::duration<double>(b->t_copied - b->t_open).count();
                        B.t_wait += std::chrono::duration<double>(t_go - b->t_copied).count();
                        B.t_run += std::chrono::duration<double>(b->t_done - t_go).count();
                    }
                    b->status = rc;
                    b->done = true;
                    B.inflight--;
                    b->cv.notify_all();
                    for (auto &other : B.live)  // the pipeline has room again: one caller of every waiting batch gets to look
                        if (other != b && !other->running) other->cv.notify_one();
                    break;
                }
            }
            b->cv.wait(lock);
        }
        const int rc = b->status;
        lock.unlock();
        if (rc == RPH_OK) {
            const uint8_t *out = b->slot->h_out;
            std::memcpy(hash32_out, out + (size_t)slot_index * 32, 32);
            if (quality_out) std::memcpy(quality_out, out + b->o_q + (size_t)slot_index * 4, 4);
            if (coeffs_out) std::memcpy(coeffs_out, out + b->o_c + (size_t)slot_index * 1024, 1024);
            if (valid_out) *valid_out = out[b->o_v + slot_index];
        }
        lock.lock();
        if (--b->readers == 0) {  // last caller out: the slot is free again
            if (kTrace) B.t_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - b->t_done).count();
            b->slot->busy = false;
            B.live.erase(std::remove(B.live.begin(), B.live.end(), b), B.live.end());
            B.cv_slot.notify_all();
        }
        return rc;
    });
}
