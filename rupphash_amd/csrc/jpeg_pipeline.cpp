// jpeg_pipeline.cpp -- host side of the JPEG path (row N3 of SURVEY 8f) and its C ABI: chunking, the two lanes of staging / device
// buffers, host entropy decoding or stream preparation for the device walk, reconstruction + hashing sub-batches, the one-file queue.
// The kernels are in jpeg_kernels.hip, the entropy decoder and the stream preparation in jpeg_host.cpp.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "jpeg_device.h"
#include "rph_internal.h"

namespace {

#define RPH_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != RPH_OK) return rc_; \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// A pinned host buffer with its device twin, grown on demand (the caller has synchronised the stream that used it)
struct Twin {
    uint8_t *h = nullptr, *d = nullptr;
    size_t cap = 0;
    void release()
    {
        if (h) (void)hipHostFree(h);
        if (d) (void)hipFree(d);
        h = d = nullptr;
        cap = 0;
    }
    int reserve(size_t bytes, bool host = true, bool dev = true)
    {
        if (cap >= bytes) return RPH_OK;
        release();
        bytes = align_up(bytes + bytes / 4, 4096);
        if (host) RPH_HIP_CHECK(hipHostMalloc((void **)&h, bytes));
        if (dev) RPH_HIP_CHECK(hipMalloc((void **)&d, bytes));
        cap = bytes;
        return RPH_OK;
    }
};

constexpr size_t RES_BYTES = 32 + 4 + 1024 + 256 + 4;  // per image: hash, quality, coefficients, dihedral, valid + entropy status (padded)
struct ResView {  // the per-image result arrays inside one buffer laid out for `images` images
    uint8_t *hash, *quality, *coeffs, *dihedral, *valid, *status;
    ResView(uint8_t *p, size_t images)
    {
        hash = p;
        quality = hash + images * 32;
        coeffs = quality + images * 4;
        dihedral = coeffs + images * 1024;
        valid = dihedral + images * 256;
        status = valid + images;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------
// The pipeline.  One JPEG batch call per context at a time (ctx->jpeg_mu); buffers are kept in the context across calls.
//   host entropy:   chunks of <= 192 MB of coefficients, two slots: the host threads decode chunk k + 1 while the device works on k
//   device entropy: chunks as large as the coefficient buffer allows (tens of thousands of images: one per lane), the host only
//                   prepares streams; reconstruction + hashing then runs over the chunk in sub-batches through small buffers
// ---------------------------------------------------------------------------------------------------------------------------
struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // device entropy: the slot's chunk (work on another slot's stream) has delivered its results
    Twin coef;              // host entropy: pinned staging + device; device entropy: unused
    Twin stream_bytes;      // device entropy: de-stuffed entropy bytes
    Twin meta;              // descriptors (planes | images | tables | HImage | order | DeviceLut)
    Twin res;               // results
    size_t res_images = 0;
    uint8_t *d_segwork = nullptr;  // device entropy, segmented streams: states | records of round 0 | out positions | segment -> file map
    size_t segwork_cap = 0;
    unsigned long long *d_pmask = nullptr;  // device entropy, progressive files: which coefficients are nonzero, one word per block
    size_t pmask_cap = 0;
    PCorr *d_pcorr = nullptr;               // and the records of their AC refinement scans (jpeg_device.h)
    size_t pcorr_cap = 0;
    uint8_t *d_pdc = nullptr;               // and the bits of their DC refinement scans, one byte per block
    size_t pdc_cap = 0;
    void release()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        if (res.d == res.h) res.d = nullptr;  // (one allocation: see reserve_res)
        coef.release();
        stream_bytes.release();
        meta.release();
        res.release();
        if (d_segwork) (void)hipFree(d_segwork);
        if (d_pmask) (void)hipFree(d_pmask);
        if (d_pcorr) (void)hipFree(d_pcorr);
        if (d_pdc) (void)hipFree(d_pdc);
        if (stream) (void)hipStreamDestroy(stream);
        if (done) (void)hipEventDestroy(done);
        *this = Slot();
    }
    int ready()
    {
        if (!stream) RPH_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (!done) RPH_HIP_CHECK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        return RPH_OK;
    }
    int reserve_res(size_t images)  // the caller has made sure nothing in flight still uses the slot's buffers
    {
        if (res_images >= images) return RPH_OK;
        RPH_HIP_CHECK(hipStreamSynchronize(stream));
        images += images / 4;
        // pinned host memory the kernels write into directly (32 B .. 1.3 KB per image cross PCIe as they are produced): a copy back at the
        // end of a chunk queued behind the other lanes' kernels and held the lane up for tens of milliseconds
        if (res.d == res.h) res.d = nullptr;
        RPH_TRY(res.reserve(images * RES_BYTES, true, false));
        res.d = res.h;
        res_images = images;
        return RPH_OK;
    }
};
constexpr int JPEG_LANES = 4;  // chunks in flight in the device-entropy pipeline (the host-entropy pipeline uses the first two)
struct JpegPipe {
    Slot slot[JPEG_LANES];
    // reconstruction buffers (sample planes, packed pixels) shared by the slots' sub-batches: used in stream order, one sub-batch at a
    // time per stream; each slot owns one pair
    uint8_t *d_planes[JPEG_LANES] = {}, *d_out[JPEG_LANES] = {};
    size_t recon_coef_bytes[JPEG_LANES] = {};
    // device entropy: the chunk's coefficient buffer (one: chunks run back to back on one stream)
    int16_t *d_coef = nullptr;
    size_t d_coef_bytes = 0;
    // the per-file records of the last call (std::vector<Job>, defined below), kept: a call of 100 000 files spent 15 ms constructing and
    // first-touching 120 MB of them before the first byte moved
    void *jobs_cache = nullptr;
    void (*jobs_cache_free)(void *) = nullptr;
    void release()
    {
        if (jobs_cache) jobs_cache_free(jobs_cache);
        jobs_cache = nullptr;
        for (int b = 0; b < JPEG_LANES; b++) {
            slot[b].release();
            if (d_planes[b]) (void)hipFree(d_planes[b]);
            if (d_out[b]) (void)hipFree(d_out[b]);
            d_planes[b] = d_out[b] = nullptr;
            recon_coef_bytes[b] = 0;
        }
        if (d_coef) (void)hipFree(d_coef);
        d_coef = nullptr;
        d_coef_bytes = 0;
    }
    // sample planes and packed pixels for sub-batches of up to `coef_need` bytes of coefficients
    int reserve_recon(int b, size_t coef_need, hipStream_t s)
    {
        if (recon_coef_bytes[b] >= coef_need) return RPH_OK;
        RPH_HIP_CHECK(hipStreamSynchronize(s));
        if (d_planes[b]) (void)hipFree(d_planes[b]);
        if (d_out[b]) (void)hipFree(d_out[b]);
        d_planes[b] = d_out[b] = nullptr;
        recon_coef_bytes[b] = 0;
        RPH_HIP_CHECK(hipMalloc((void **)&d_planes[b], coef_need / 2 + 256));  // 64 bytes of samples per 128 bytes of coefficients
        // packed pixels never exceed the coefficient bytes (4:2:0: both 3 w h; Luma8: w h against 2 w h), plus row / image padding
        RPH_HIP_CHECK(hipMalloc((void **)&d_out[b], coef_need + coef_need / 8 + 65536));
        recon_coef_bytes[b] = coef_need;
        return RPH_OK;
    }
};

constexpr size_t CHUNK_COEF_BYTES = (size_t)192 << 20;   // host entropy, per slot: ~250 images of 512x512 4:2:0
constexpr size_t SUB_COEF_BYTES = (size_t)4 << 30;        // device entropy: reconstruction sub-batch (~5400 such images)
constexpr size_t MAX_IMAGE_COEF_BYTES = (size_t)3 << 30;  // one image beyond this is refused (RPH_ERR_UNSUPPORTED)
// A frame header may announce any geometry: a file of a few hundred bytes that declares 20000 x 20000 would have gigabytes of pinned
// memory zeroed on its behalf.  Every coded block takes at least one bit of the file (sequential: a DC category code and an end-of-block
// code, two bits; a progressive file's first DC scan: one), so a header that announces more blocks than eight per file byte cannot be honest:
// refused before anything is allocated.
static inline bool frame_is_plausible(const rphj::Frame &f, size_t file_len) { return (size_t)f.total_blocks <= 8 * file_len + 64; }
constexpr uint32_t CHUNK_MAX_IMAGES = 4096;               // host entropy
constexpr size_t SUB_MAX_IMAGES = 16384;                  // images per reconstruction sub-batch (grid.y of the kernels: 3 planes each)
constexpr uint32_t DEVICE_ENTROPY_MIN_FILES = 512;        // automatic mode: below this many lanes (files, or restart intervals) the host decodes (latency)

// Huffman tables of a chunk, one per distinct content (most sequential files of a collection share the four Annex K tables; a
// progressive file brings a dozen of its own).  Sixteen threads ask at once: the index is cut into 64 shards by the tables' hash, each
// with a lock of its own, and the tables themselves lie in blocks that never move (an id is a slot, taken with an atomic add).
struct TableStore {
    static constexpr uint32_t SHARDS = 64, BLOCK = 1024, MAX_BLOCKS = 8192;
    struct Shard {
        std::mutex mu;
        std::unordered_multimap<uint64_t, uint32_t> by_hash;
    } shard[SHARDS];
    std::mutex grow_mu;
    std::atomic<rphj::DeviceLut *> lut_block[MAX_BLOCKS];
    std::atomic<rphj::TableSpec *> spec_block[MAX_BLOCKS];
    std::atomic<uint32_t> count{0};
    uint64_t serial = next_serial();  // tells a thread's cache of ids that it belongs to another store
    TableStore()
    {
        for (uint32_t b = 0; b < MAX_BLOCKS; b++) lut_block[b].store(nullptr, std::memory_order_relaxed), spec_block[b].store(nullptr, std::memory_order_relaxed);
    }
    ~TableStore()
    {
        for (uint32_t b = 0; b < MAX_BLOCKS; b++) {
            delete[] lut_block[b].load(std::memory_order_relaxed);
            delete[] spec_block[b].load(std::memory_order_relaxed);
        }
    }
    TableStore(const TableStore &) = delete;
    TableStore &operator=(const TableStore &) = delete;
    static uint64_t next_serial()
    {
        static std::atomic<uint64_t> n{1};
        return n.fetch_add(1);
    }
    static bool same(const rphj::TableSpec &a, const rphj::TableSpec &b)
    {
        return a.total == b.total && memcmp(a.counts + 1, b.counts + 1, 16) == 0 && memcmp(a.symbols, b.symbols, a.total) == 0;
    }
    const rphj::TableSpec &spec(uint32_t id) const { return spec_block[id / BLOCK].load(std::memory_order_acquire)[id % BLOCK]; }
    uint32_t size() const { return count.load(std::memory_order_acquire); }
    void copy_luts(rphj::DeviceLut *dst) const  // (when the threads are done)
    {
        const uint32_t n = size();
        for (uint32_t first = 0; first < n; first += BLOCK)
            memcpy(dst + first, lut_block[first / BLOCK].load(std::memory_order_acquire), (size_t)std::min(BLOCK, n - first) * sizeof(rphj::DeviceLut));
    }
    uint32_t find_locked(const Shard &sh, uint64_t h, const rphj::TableSpec &t) const
    {
        auto range = sh.by_hash.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (same(spec(it->second), t)) return it->second;
        return UINT32_MAX;
    }
    uint32_t new_slot()  // UINT32_MAX: full (the file then goes to the host decoder)
    {
        const uint32_t id = count.fetch_add(1, std::memory_order_acq_rel);
        if (id >= BLOCK * MAX_BLOCKS) return UINT32_MAX;
        const uint32_t b = id / BLOCK;
        if (!lut_block[b].load(std::memory_order_acquire) || !spec_block[b].load(std::memory_order_acquire)) {
            std::lock_guard<std::mutex> lock(grow_mu);
            if (!spec_block[b].load(std::memory_order_acquire)) spec_block[b].store(new rphj::TableSpec[BLOCK], std::memory_order_release);
            if (!lut_block[b].load(std::memory_order_acquire)) lut_block[b].store(new rphj::DeviceLut[BLOCK], std::memory_order_release);
        }
        return id;
    }
    // Every file of a call asks for its tables, from sixteen threads: each thread remembers the last few it was given (no lock at all
    // for them: the four tables a collection of sequential files shares); a table nobody has seen is built OUTSIDE the lock.
    static uint32_t intern(void *self, const rphj::TableSpec &t)
    {
        TableStore &T = *static_cast<TableStore *>(self);
        uint64_t h = 1469598103934665603ULL;
        for (int q = 1; q <= 16; q++) h = (h ^ t.counts[q]) * 1099511628211ULL;
        for (int q = 0; q < t.total; q++) h = (h ^ t.symbols[q]) * 1099511628211ULL;
        struct Recent {
            uint64_t serial = 0, hash[8];
            uint32_t id[8];
            rphj::TableSpec spec[8];
            int n = 0, next = 0;
        };
        static thread_local Recent recent;
        if (recent.serial != T.serial) {
            recent.serial = T.serial;
            recent.n = recent.next = 0;
        }
        for (int i = 0; i < recent.n; i++)
            if (recent.hash[i] == h && same(recent.spec[i], t)) return recent.id[i];
        Shard &sh = T.shard[(h >> 7) % SHARDS];
        uint32_t id;
        {
            std::lock_guard<std::mutex> lock(sh.mu);
            id = T.find_locked(sh, h, t);
        }
        if (id == UINT32_MAX) {
            rphj::DeviceLut L;
            if (rphj::build_device_lut(t, L) != RPH_OK) return UINT32_MAX;
            std::lock_guard<std::mutex> lock(sh.mu);
            id = T.find_locked(sh, h, t);  // (another thread may have been quicker)
            if (id == UINT32_MAX) {
                id = T.new_slot();
                if (id == UINT32_MAX) return UINT32_MAX;
                T.lut_block[id / BLOCK].load(std::memory_order_acquire)[id % BLOCK] = L;
                T.spec_block[id / BLOCK].load(std::memory_order_acquire)[id % BLOCK] = t;
                sh.by_hash.emplace(h, id);
            }
        }
        const int slot = recent.n < 8 ? recent.n++ : (recent.next = (recent.next + 1) & 7);
        recent.hash[slot] = h;
        recent.id[slot] = id;
        recent.spec[slot] = t;
        return id;
    }
};

struct Job {
    Job() {}  // user-provided on purpose: std::vector<Job>(n) then runs the member initialisers only instead of zeroing ~1 KB per job first
    const uint8_t *data = nullptr;
    size_t len = 0;
    rphj::Frame frame;
    int status = RPH_OK;
    uint64_t first_block = 0;  // within the chunk's coefficient buffer
    rphj::StreamPlan plan;     // device entropy
    const int16_t *pre = nullptr;  // coefficients already decoded by the caller into pinned memory (rph_jpeg_pdq_hash_one)
    std::vector<uint32_t> marks;   // device entropy: where the restart intervals of a one-scan file begin (empty: walk the file with one lane)
    size_t stream_off = 0, stream_used = 0;
};

using Jobs = std::vector<Job>;

// bytes of the chunk's stream buffer a file may need (prepare_stream: 32 zero bytes behind every scan)
inline size_t stream_cap(const Job &j) { return align_up(j.len + 160 + (j.frame.progressive ? 32 * (size_t)rphj::MAX_PROG_SCANS : 0), 16); }

// Channels of the pixels the device writes for a file: Rgb8 only where the caller reads RGB (rph_jpeg_decode); a colour file that
// only the hasher reads is written as its Rec.601 luma (a third of the bytes, and the PDQ paths start from luma anyway: Luma8 input
// is borrowed as it is, pdqhash.rs:176; 512x512 Luma8 has its own form of the fused kernel)
inline uint32_t out_channels(const rphj::Frame &f, bool rgb_wanted) { return (f.ncomp == 1 || !rgb_wanted) ? 1u : 3u; }

size_t out_bytes_of(const rphj::Frame &f, uint32_t channels)
{
    const size_t stride = (size_t)channels * align_up(f.w, 8);
    return align_up(stride * f.h, 64);
}

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline int trace_level()  // RPH_JPEG_TRACE=1: synchronise after every device phase and print its time; 2: host-side timestamps only (no extra synchronisation)
{
    static const int level = getenv("RPH_JPEG_TRACE") ? atoi(getenv("RPH_JPEG_TRACE")) : 0;
    return level;
}
inline bool trace_on() { return trace_level() == 1; }
#define RPH_JPEG_STAMP(...)                                   \
    do {                                                      \
        if (trace_level() == 2) {                             \
            fprintf(stderr, "[rph_jpeg %9.1f ms] ", now_ms() - g_trace_t0); \
            fprintf(stderr, __VA_ARGS__);                     \
            fprintf(stderr, "\n");                            \
        }                                                     \
    } while (0)
static double g_trace_t0 = 0;

// Host threads when the caller does not say: what this process may actually use (its affinity mask, and the cgroup CPU quota a
// container runs under -- hardware_concurrency() reports the machine's 256 threads inside a 16-CPU container)
unsigned default_threads()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<unsigned>(n, (unsigned)std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "max 100000" or "1600000 100000"
        char q[32] = "";
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(h, "%lld", &period) != 1) period = 100000;
            fclose(h);
        }
    }
    if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    return n;
}

template <class F>
void parallel_for(size_t first, size_t last, unsigned threads, F &&body)
{
    std::atomic<size_t> next{first};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= last) return;
            body(i);
        }
    };
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, threads), last > first ? last - first : 1);
    if (nt <= 1) {
        work();
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t + 1 < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

struct Outputs {
    uint8_t *hash = nullptr;
    float *quality = nullptr;
    float *coeffs = nullptr;
    uint8_t *dihedral = nullptr;
    uint8_t *valid = nullptr;
    int32_t *status = nullptr;
    uint8_t *pixels = nullptr;  // single-image decode: packed w * h * channels
    bool want_hash = true;
};

// Descriptors of the chunk idx[first..last) and where they live in the slot's meta buffer (host and device at the same offsets)
struct ChunkDesc {
    size_t m = 0;
    size_t off_planes = 0, off_images = 0, off_tables = 0, off_end = 0;
    std::vector<uint32_t> image_of, plane_of;  // per chunk position: index of its JImage / first JPlane (UINT32_MAX: not decodable)
    uint32_t n_planes = 0, n_images = 0;
    const PRef *d_refs = nullptr;   // device entropy, progressive files: the AC refinement scans of their planes and the records of
    const PCorr *d_corr = nullptr;  // their corrections (the IDCT kernel applies them)
    const uint8_t *d_dcbits = nullptr;
};

// Sub-batch boundaries are where the offsets of planes and pixels restart from 0: `sub_of[r]` = first chunk position of r's sub-batch.
int build_descriptors(Jobs &jobs, const std::vector<uint32_t> &idx, size_t first, size_t last, int flavour, bool rgb_wanted, size_t sub_coef_bytes, uint8_t *h_meta,
                      size_t meta_base, ChunkDesc &D, std::vector<size_t> &sub_starts)
{
    const size_t m = last - first;
    D.m = m;
    D.off_planes = meta_base;
    D.off_images = D.off_planes + m * 3 * sizeof(JPlane);
    D.off_tables = D.off_images + m * sizeof(JImage);
    D.off_end = D.off_tables + m * 3 * 128;
    JPlane *hp = reinterpret_cast<JPlane *>(h_meta + D.off_planes);
    JImage *hi = reinterpret_cast<JImage *>(h_meta + D.off_images);
    uint16_t *hq = reinterpret_cast<uint16_t *>(h_meta + D.off_tables);
    D.image_of.assign(m, UINT32_MAX);
    D.plane_of.assign(m, UINT32_MAX);
    D.n_planes = D.n_images = 0;
    sub_starts.clear();
    sub_starts.push_back(0);
    size_t plane_bytes = 0, out_bytes = 0, sub_blocks = 0, sub_images = 0;
    uint32_t sub_first_plane = 0;
    static const bool fuse = !getenv("RPH_JPEG_NO_FUSED");  // (A/B: the two-kernel reconstruction for every image)
    for (size_t r = 0; r < m; r++) {
        Job &j = jobs[idx[first + r]];
        if (j.status != RPH_OK) continue;
        const rphj::Frame &f = j.frame;
        // a sub-batch is full when its coefficients would not fit the reconstruction buffers, or at 16 384 images (the kernels take
        // the plane / image index from blockIdx.y, which ends at 65 535): offsets restart
        if (sub_blocks && ((sub_blocks + f.total_blocks) * 128 > sub_coef_bytes || sub_images == SUB_MAX_IMAGES)) {
            sub_starts.push_back(r);
            plane_bytes = out_bytes = sub_blocks = 0;
            sub_images = 0;
            sub_first_plane = D.n_planes;
        }
        sub_blocks += f.total_blocks;
        sub_images++;
        JImage im;
        memset(&im, 0, sizeof im);
        D.plane_of[r] = D.n_planes;
        // three components (luma sampled once or twice the chroma) and only the hasher reads the pixels: IDCT + upsampling + colour in one kernel
        const bool fused = fuse && f.ncomp == 3 && out_channels(f, rgb_wanted) == 1 && f.comp[0].H <= 2 && f.comp[0].V <= 2 && f.comp[1].H == 1 && f.comp[1].V == 1 &&
                           f.comp[2].H == 1 && f.comp[2].V == 1;
        im.fused = fused;
        im.first_plane = D.n_planes - sub_first_plane;
        im.tiles_x = (f.comp[0].blocks_w + 15) / 16, im.tiles_y = (f.comp[0].blocks_h + 7) / 8;
        for (int c = 0; c < f.ncomp; c++) {
            const rphj::Comp &kc = f.comp[c];
            JPlane pl;
            pl.first_block = j.first_block + kc.first_block;
            pl.out_off = plane_bytes;
            pl.blocks_w = kc.blocks_w;
            pl.blocks_h = kc.blocks_h;
            pl.qt = D.n_planes;
            pl.pitch = kc.blocks_w * 8;
            pl.real_bw = kc.real_bw;
            pl.real_bh = kc.real_bh;
            pl.ref_first = pl.ref_count = 0;
            pl.fused = fused, pl.pad_ = 0;
            memcpy(hq + (size_t)D.n_planes * 64, f.qt[kc.tq], 128);
            im.plane_off[c] = plane_bytes;
            im.pitch[c] = pl.pitch;
            plane_bytes += (size_t)pl.pitch * kc.blocks_h * 8;
            hp[D.n_planes++] = pl;
        }
        im.w = f.w;
        im.h = f.h;
        im.ncomp = (uint32_t)f.ncomp;
        im.hs = im.vs = 1;
        if (f.ncomp == 3) {
            im.hs = f.comp[0].H / f.comp[1].H;
            im.vs = f.comp[0].V / f.comp[1].V;
            im.cw = flavour == RPH_JPEG_LIBJPEG ? f.comp[1].samp_w : f.comp[1].blocks_w * 8;
            im.ch = flavour == RPH_JPEG_LIBJPEG ? f.comp[1].samp_h : f.comp[1].blocks_h * 8;
        }
        const uint32_t och = out_channels(f, rgb_wanted);
        im.luma_out = f.ncomp == 3 && och == 1;
        im.out_stride = (uint32_t)((size_t)och * align_up(f.w, 8));
        im.out_off = out_bytes;
        out_bytes += out_bytes_of(f, och);
        D.image_of[r] = D.n_images;
        hi[D.n_images++] = im;
    }
    return RPH_OK;
}

// IDCT + upsampling / colour + hashing of chunk positions [r0, r1) (one sub-batch: its planes and pixels fit the slot's reconstruction buffers)
int reconstruct_and_hash(rph_ctx *ctx, JpegPipe &P, int b, Slot &S, Jobs &jobs, const std::vector<uint32_t> &idx, size_t first, const ChunkDesc &D, size_t r0,
                         size_t r1, const int16_t *d_coef, int flavour, const Outputs &out, hipStream_t s)
{
    // the planes and images of the sub-batch are contiguous in the descriptor arrays
    uint32_t p0 = UINT32_MAX, p1 = 0, i0 = UINT32_MAX, i1 = 0, max_blocks = 0, max_groups = 0;
    for (size_t r = r0; r < r1; r++) {
        if (D.image_of[r] == UINT32_MAX) continue;
        const rphj::Frame &f = jobs[idx[first + r]].frame;
        p0 = std::min(p0, D.plane_of[r]);
        p1 = std::max(p1, D.plane_of[r] + (uint32_t)f.ncomp);
        i0 = std::min(i0, D.image_of[r]);
        i1 = std::max(i1, D.image_of[r] + 1);
        for (int c = 0; c < f.ncomp; c++) max_blocks = std::max(max_blocks, f.comp[c].blocks_w * f.comp[c].blocks_h);
        max_groups = std::max<uint32_t>(max_groups, (uint32_t)(((f.w + 7) / 8) * (size_t)f.h));
    }
    if (i0 == UINT32_MAX) return RPH_OK;
    const JPlane *dp = reinterpret_cast<const JPlane *>(S.meta.d + D.off_planes) + p0;
    const JImage *di = reinterpret_cast<const JImage *>(S.meta.d + D.off_images) + i0;
    const uint16_t *dq = reinterpret_cast<const uint16_t *>(S.meta.d + D.off_tables);
    const JImage *hi_all = reinterpret_cast<const JImage *>(S.meta.h + D.off_images);
    uint32_t n_fused = 0, max_tiles = 0;
    for (uint32_t q = i0; q < i1; q++)
        if (hi_all[q].fused) {
            n_fused++;
            max_tiles = std::max(max_tiles, hi_all[q].tiles_x * hi_all[q].tiles_y);
        }
    if (n_fused < i1 - i0) {  // (the plane kernels skip the images the fused kernel takes)
        RPH_TRY(rph_jpeg_launch_idct(flavour, max_blocks, p1 - p0, s, d_coef, dq, dp, P.d_planes[b], D.d_refs, D.d_corr, D.d_dcbits));
        RPH_TRY(rph_jpeg_launch_color(flavour, max_groups, i1 - i0, s, P.d_planes[b], di, P.d_out[b]));
    }
    if (n_fused) RPH_TRY(rph_jpeg_launch_fused(flavour, max_tiles, i1 - i0, s, d_coef, dq, dp, di, P.d_out[b], D.d_refs, D.d_corr, D.d_dcbits));
    if (!out.want_hash) return RPH_OK;
    // hash runs of equal geometry where the pixels lie (generate_pdq_features, scanner.rs:1410)
    ResView R(S.res.d, S.res_images);
    const JImage *hi = reinterpret_cast<const JImage *>(S.meta.h + D.off_images);
    for (size_t r = r0; r < r1;) {
        if (D.image_of[r] == UINT32_MAX) {
            r++;
            continue;
        }
        const rphj::Frame &f = jobs[idx[first + r]].frame;
        const uint32_t och = out_channels(f, out.pixels != nullptr);
        size_t e = r + 1;
        while (e < r1 && D.image_of[e] != UINT32_MAX && jobs[idx[first + e]].frame.w == f.w && jobs[idx[first + e]].frame.h == f.h && jobs[idx[first + e]].frame.ncomp == f.ncomp) e++;
        RPH_TRY(rph_pdq_hash_batch_dev(ctx, P.d_out[b] + hi[D.image_of[r]].out_off, (uint32_t)(e - r), f.w, f.h, och, (size_t)och * align_up(f.w, 8), out_bytes_of(f, och),
                                       R.hash + r * 32, out.quality ? R.quality + r * 4 : nullptr, out.coeffs ? R.coeffs + r * 1024 : nullptr,
                                       out.dihedral ? R.dihedral + r * 256 : nullptr, R.valid + r, s));
        r = e;
    }
    return RPH_OK;
}

// The results of a chunk are fetched only when the chunk is known to be finished.  Enqueued behind the chunk's kernels instead, the
// transfer would sit in the copy engine's queue waiting for them, and the next chunk's upload -- same engine, another stream --
// would wait behind it (measured: uploads did not overlap the other lane's kernels at all).
// the first m entries of the slot's result arrays cleared by the host (the lane is idle: its previous chunk has delivered)
void zero_results(Slot &S, size_t m, const Outputs &out)
{
    ResView H(S.res.h, S.res_images);
    memset(H.hash, 0, m * 32);
    memset(H.quality, 0, m * 4);
    if (out.coeffs) memset(H.coeffs, 0, m * 1024);
    if (out.dihedral) memset(H.dihedral, 0, m * 256);
    memset(H.valid, 0, m);
    memset(H.status, 0, m);
}

int fetch_results(Slot &S, size_t m, const Outputs &out, bool entropy_status)
{
    (void)S, (void)m, (void)out, (void)entropy_status;  // the kernels wrote into the slot's pinned results buffer; the caller has waited for them
    return RPH_OK;
}

// results of a finished chunk -> the caller's arrays (scattered through idx)
void scatter_results(const Slot &S, Jobs &jobs, const std::vector<uint32_t> &idx, size_t first, size_t last, const Outputs &out, bool entropy_status)
{
    ResView R(S.res.h, S.res_images);
    for (size_t r = 0; r < last - first; r++) {
        const uint32_t g = idx[first + r];
        Job &j = jobs[g];
        if (entropy_status && j.status == RPH_OK && R.status[r]) j.status = RPH_ERR_INVALID_ARG;  // the device walk met a corrupt stream
        const bool ok = j.status == RPH_OK;
        if (out.hash) ok ? (void)memcpy(out.hash + (size_t)g * 32, R.hash + r * 32, 32) : (void)memset(out.hash + (size_t)g * 32, 0, 32);
        if (out.quality) ok ? (void)memcpy(out.quality + g, R.quality + r * 4, 4) : (void)memset(out.quality + g, 0, 4);
        if (out.coeffs) ok ? (void)memcpy(out.coeffs + (size_t)g * 256, R.coeffs + r * 1024, 1024) : (void)memset(out.coeffs + (size_t)g * 256, 0, 1024);
        if (out.dihedral) ok ? (void)memcpy(out.dihedral + (size_t)g * 256, R.dihedral + r * 256, 256) : (void)memset(out.dihedral + (size_t)g * 256, 0, 256);
        if (out.valid) out.valid[g] = ok ? R.valid[r] : 0;
    }
}

// ---- host entropy decoding: the files idx[...] in chunks over the two slots
int run_host_entropy(rph_ctx *ctx, JpegPipe &P, Jobs &jobs, const std::vector<uint32_t> &idx, int flavour, unsigned threads, const Outputs &out)
{
    struct Pending {
        bool active = false;
        size_t first = 0, last = 0;
    } pend[2];
    auto finish = [&](int b) -> int {
        if (!pend[b].active) return RPH_OK;
        RPH_HIP_CHECK(hipStreamSynchronize(P.slot[b].stream));
        RPH_TRY(fetch_results(P.slot[b], pend[b].last - pend[b].first, out, false));
        scatter_results(P.slot[b], jobs, idx, pend[b].first, pend[b].last, out, false);
        pend[b].active = false;
        return RPH_OK;
    };
    const size_t n = idx.size();
    int k = 0;
    RPH_JPEG_STAMP("buffers ready");
    for (size_t first = 0; first < n; k++) {
        size_t last = first, blocks = 0;
        while (last < n && last - first < CHUNK_MAX_IMAGES) {
            const Job &j = jobs[idx[last]];
            const size_t nb = j.status == RPH_OK ? (size_t)j.frame.total_blocks : 0;
            if (last > first && (blocks + nb) * 128 > CHUNK_COEF_BYTES) break;
            blocks += nb;
            last++;
        }
        const int b = k & 1;
        RPH_TRY(finish(b));
        Slot &S = P.slot[b];
        RPH_TRY(S.ready());
        const size_t m = last - first;
        const size_t coef_need = std::max(CHUNK_COEF_BYTES, blocks * 128);
        if (S.coef.cap < coef_need) {
            RPH_HIP_CHECK(hipStreamSynchronize(S.stream));
            RPH_TRY(S.coef.reserve(coef_need));
        }
        RPH_TRY(P.reserve_recon(b, coef_need, S.stream));
        RPH_TRY(S.reserve_res(std::max<size_t>(m, std::min<size_t>(n, CHUNK_MAX_IMAGES))));
        const size_t meta_need = std::max<size_t>(m, std::min<size_t>(n, CHUNK_MAX_IMAGES)) * (3 * sizeof(JPlane) + sizeof(JImage) + 3 * 128);
        if (S.meta.cap < meta_need) {
            RPH_HIP_CHECK(hipStreamSynchronize(S.stream));
            RPH_TRY(S.meta.reserve(meta_need));
        }
        {
            uint64_t fb = 0;
            for (size_t i = first; i < last; i++) {
                Job &j = jobs[idx[i]];
                j.first_block = fb;
                if (j.status == RPH_OK) fb += j.frame.total_blocks;
            }
        }
        int16_t *h_coef = reinterpret_cast<int16_t *>(S.coef.h);
        bool predecoded = false;
        for (size_t i = first; i < last; i++) predecoded |= jobs[idx[i]].pre != nullptr;
        if (!predecoded)
            parallel_for(first, last, threads, [&](size_t i) {
                Job &j = jobs[idx[i]];
                if (j.status == RPH_OK) j.status = rphj::decode_coefficients(j.data, j.len, j.frame, h_coef + j.first_block * 64);
            });
        RPH_JPEG_STAMP("lane %d: chunk %d buffers sized", b, k);
        ChunkDesc D;
        std::vector<size_t> subs;
        RPH_TRY(build_descriptors(jobs, idx, first, last, flavour, out.pixels != nullptr, SIZE_MAX / 256, S.meta.h, 0, D, subs));
        hipStream_t s = S.stream;
        zero_results(S, m, out);
        if (D.n_images) {
            if (predecoded) {  // every caller decoded into its own pinned buffer: the copy engine takes the coefficients from there
                for (size_t i = first; i < last; i++) {
                    const Job &j = jobs[idx[i]];
                    if (j.status == RPH_OK)
                        RPH_HIP_CHECK(hipMemcpyAsync(S.coef.d + j.first_block * 128, j.pre, (size_t)j.frame.total_blocks * 128, hipMemcpyHostToDevice, s));
                }
            } else {
                RPH_HIP_CHECK(hipMemcpyAsync(S.coef.d, S.coef.h, blocks * 128, hipMemcpyHostToDevice, s));
            }
            RPH_HIP_CHECK(hipMemcpyAsync(S.meta.d, S.meta.h, D.off_end, hipMemcpyHostToDevice, s));
            RPH_TRY(reconstruct_and_hash(ctx, P, b, S, jobs, idx, first, D, 0, m, reinterpret_cast<const int16_t *>(S.coef.d), flavour, out, s));
        }
        if (out.pixels && m == 1 && jobs[idx[first]].status == RPH_OK) {  // single-image decode: rows without their padding
            const rphj::Frame &f = jobs[idx[first]].frame;
            const size_t row = (size_t)f.ncomp * f.w, stride = (size_t)f.ncomp * align_up(f.w, 8);
            RPH_HIP_CHECK(hipMemcpy2DAsync(out.pixels, row, P.d_out[b], stride, row, f.h, hipMemcpyDeviceToHost, s));
        }
        pend[b].active = true;
        pend[b].first = first;
        pend[b].last = last;
        first = last;
    }
    RPH_TRY(finish(k & 1));
    RPH_TRY(finish((k + 1) & 1));
    return RPH_OK;
}

// ---- device entropy decoding: the sequential files idx[...]; files the device walk does not take come back in `leftover` for the host
// entropy bytes per second the segment passes (synchronisation + walk) move when the device is theirs: 2 GB in 14 + 18 ms
inline double seg_rate()
{
    static const double r = getenv("RPH_JPEG_SEG_RATE") ? atof(getenv("RPH_JPEG_SEG_RATE")) * 1e9 : 60e9;
    return r;
}

int run_device_entropy(rph_ctx *ctx, JpegPipe &P, Jobs &jobs, std::vector<uint32_t> idx, int flavour, unsigned threads, const Outputs &out,
                       std::vector<uint32_t> &leftover)
{
    // images of similar stream length share a wave: sort the whole list by file length first (chunks then are slices of it)
    {  // (keys side by side: sorting through the job records themselves took 22 ms per 100 000 files)
        std::vector<std::pair<uint64_t, uint32_t>> key(idx.size());
        for (size_t i = 0; i < idx.size(); i++) key[i] = {~(uint64_t)jobs[idx[i]].len, (uint32_t)i};  // longest first, equal lengths as they came
        std::sort(key.begin(), key.end());
        std::vector<uint32_t> sorted(idx.size());
        for (size_t i = 0; i < idx.size(); i++) sorted[i] = idx[key[i].second];
        idx.swap(sorted);
    }
    RPH_JPEG_STAMP("files sorted by length");
    // the chunk's coefficient buffer: as much of the free device memory as is reasonable, but no more than this call can use
    size_t need = 0;
    for (uint32_t g : idx) need += (size_t)jobs[g].frame.total_blocks * 128;
    size_t free_b = 0, total_b = 0;
    RPH_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::max<size_t>((size_t)1 << 30, std::min<size_t>((free_b + P.d_coef_bytes) / 2, (size_t)96 << 30));
    // Up to JPEG_LANES lanes of resources (stream, staging, a share of the coefficient buffer, reconstruction buffers), chunks take them in
    // turn: the host prepares chunk k + 1 and its bytes cross PCIe while chunk k is on the device, and the latency-bound walk of one
    // chunk runs beside the bandwidth-bound reconstruction of the other.  A call is cut into about four chunks when it is large
    // enough for each to still fill the device's lanes (16 GB of coefficients = 20 000 images of 512x512); a smaller call is one chunk.
    // The walk of a chunk takes as long as its longest file (~0.65 us per entropy byte: 21 ms for 29 KB files, 236 ms for 366 KB photos)
    // however few files it has, and walks of different chunks only overlap pairwise (two lanes): small files are cut into four chunks
    // for the pipelining, files whose lanes have long streams (photos without markers, segments switched off) into two so that the walks are
    // not paid four times.
    size_t max_len = 0;  // longest stream one lane will walk: a file, or one restart interval of it (as the frame header announces them)
    size_t longest_plain = 0, plain_bytes = 0, plain_files = 0;  // files without restart intervals that are long enough for segments
    for (uint32_t g : idx) {
        const rphj::Frame &f = jobs[g].frame;
        const uint64_t mcus = (uint64_t)f.mcus_x * f.mcus_y, intervals = f.restart_interval ? (mcus + f.restart_interval - 1) / f.restart_interval : 1;
        const size_t lane_len = jobs[g].len / (size_t)std::max<uint64_t>(1, intervals);
        if (!f.restart_interval && !f.progressive && ctx->jpeg_seg_bytes && jobs[g].len >= ctx->jpeg_seg_min_bytes) {
            longest_plain = std::max(longest_plain, jobs[g].len), plain_bytes += jobs[g].len, plain_files++;
            continue;
        }
        max_len = std::max(max_len, lane_len);
    }
    bool by_segments = false;
    if (plain_files) {  // will a quarter of them be walked as segments (the chunk loop decides the same way, per chunk)?
        const double t_whole = 0.65e-6 * (double)longest_plain * std::max(1.0, (double)plain_files / 4 / 65536.0), t_seg = (double)plain_bytes / 4 / seg_rate();
        by_segments = t_seg < 0.7 * t_whole;
        max_len = std::max(max_len, by_segments ? (size_t)ctx->jpeg_seg_bytes : longest_plain);
    }
    size_t min_chunk = (size_t)16 << 30, parts = 0.65e-6 * (double)max_len > 0.08 ? 2 : 4;
    // A call whose long streams are cut into segments has no long lane: its walk scales with the bytes, and the call is bound by PCIe
    // (the entropy bytes of 20 000 photos cross it in 143 ms of the call's 240).  Many small chunks then keep the copy engine busy from
    // the first prepared chunk to the last and leave little device work behind the last upload (20 000 photos: 4 chunks 75 k files/s,
    // 8 chunks 81 k, 16 chunks 86 k, 32 chunks 71 k).
    if (by_segments && 0.65e-6 * (double)max_len <= 0.08) min_chunk = (size_t)4 << 30, parts = 16;
    if (const char *e = getenv("RPH_JPEG_CHUNK_GB")) min_chunk = (size_t)atoi(e) << 30;  // experiments
    if (const char *e = getenv("RPH_JPEG_PARTS")) parts = (size_t)std::max(1, atoi(e));
    const size_t chunk_target = std::min(need, std::max(need / parts + 128, min_chunk));
    const bool single = need <= chunk_target && need <= budget;
    const int lanes = single ? 1 : (int)std::min<size_t>(JPEG_LANES, (need + chunk_target - 1) / chunk_target);  // chunks in flight
    const size_t want = single ? need : std::min(budget, (size_t)lanes * chunk_target);
    if (P.d_coef_bytes < want) {
        RPH_HIP_CHECK(hipDeviceSynchronize());
        if (P.d_coef) (void)hipFree(P.d_coef);
        P.d_coef = nullptr;
        P.d_coef_bytes = 0;
        RPH_HIP_CHECK(hipMalloc((void **)&P.d_coef, want));
        P.d_coef_bytes = want;
    }
    const size_t region = (P.d_coef_bytes / (size_t)lanes) / 128 * 128;
    const size_t chunk_bytes = std::min(region, chunk_target);
    struct Pending {
        bool active = false;
        size_t first = 0, last = 0;
    } pend[JPEG_LANES];
    auto finish = [&](int b) -> int {
        if (!pend[b].active) return RPH_OK;
        RPH_JPEG_STAMP("lane %d: waiting for its chunk", b);
        RPH_HIP_CHECK(hipEventSynchronize(P.slot[b].done));
        RPH_JPEG_STAMP("lane %d: chunk done", b);
        RPH_TRY(fetch_results(P.slot[b], pend[b].last - pend[b].first, out, true));
        scatter_results(P.slot[b], jobs, idx, pend[b].first, pend[b].last, out, true);
        RPH_JPEG_STAMP("lane %d: results scattered", b);
        pend[b].active = false;
        return RPH_OK;
    };
    for (int b = 0; b < lanes; b++) RPH_TRY(P.slot[b].ready());
    {
        size_t max_img = 0;  // a sub-batch holds at least one image
        for (uint32_t g : idx) max_img = std::max(max_img, (size_t)jobs[g].frame.total_blocks * 128);
        const size_t recon = std::max(std::min(SUB_COEF_BYTES, std::max(std::min(want, chunk_bytes), (size_t)64 << 20)), max_img);
        for (int b = 0; b < lanes; b++) RPH_TRY(P.reserve_recon(b, recon, P.slot[b].stream));
    }
    const size_t n = idx.size();
    int k = 0;
    RPH_JPEG_STAMP("buffers ready");
    for (size_t first = 0; first < n; k++) {
        size_t last = first, blocks = 0, file_bytes = 0;
        while (last < n) {
            const Job &j = jobs[idx[last]];
            const size_t nb = (size_t)j.frame.total_blocks;
            if (last > first && (blocks + nb) * 128 > chunk_bytes) break;
            blocks += nb;
            file_bytes += stream_cap(j);
            last++;
        }
        if (blocks * 128 > region) {  // one image larger than a whole region: the host path takes it
            leftover.push_back(idx[first]);
            first = last;
            k--;
            continue;
        }
        const int b = k % lanes;
        RPH_TRY(finish(b));  // the lane is free again once its previous chunk (`lanes` chunks back) has delivered its results
        Slot &S = P.slot[b];
        hipStream_t s = S.stream;
        int16_t *d_coef = P.d_coef + (size_t)b * (region / 2);  // (int16 elements: region bytes per lane)
        const size_t m = last - first;
        RPH_TRY(S.reserve_res(m));
        RPH_TRY(S.stream_bytes.reserve(file_bytes + 64));
        // ---- streams and scan plans (host threads: memchr + memcpy)
        const double t0 = now_ms();
        {
            size_t off = 0;
            uint64_t fb = 0;
            for (size_t i = first; i < last; i++) {
                Job &j = jobs[idx[i]];
                j.stream_off = off;
                off += stream_cap(j);
                j.first_block = fb;
                fb += j.frame.total_blocks;
            }
        }
        TableStore store;
        std::vector<HImage> himgs(m);
        RPH_JPEG_STAMP("lane %d: chunk %d laid out", b, k);
        parallel_for(first, last, threads, [&](size_t i) {
            Job &j = jobs[idx[i]];
            HImage &hi = himgs[i - first];
            memset(&hi, 0, sizeof hi);
            j.status = rphj::prepare_stream(j.data, j.len, j.frame, j.plan, S.stream_bytes.h + j.stream_off, stream_cap(j), &j.stream_used, &TableStore::intern, &store, &j.marks);
            if (j.status != RPH_OK) return;
            const rphj::Frame &f = j.frame;
            hi.first_block = j.first_block;
            hi.stream_base = j.stream_off;
            hi.mcus_x = f.mcus_x;
            hi.mcus_y = f.mcus_y;
            hi.n_scans = f.progressive ? 0u : (uint32_t)j.plan.n_scans;
            hi.ncomp = (uint32_t)f.ncomp;
            for (int c = 0; c < f.ncomp; c++) {
                const rphj::Comp &kc = f.comp[c];
                hi.comp[c] = HComp{kc.blocks_w, kc.real_bw, kc.real_bh, (uint32_t)kc.first_block, kc.H, kc.V};
            }
            for (int q = 0; q < (f.progressive ? 0 : j.plan.n_scans); q++) {
                const rphj::ScanPlan &sp = j.plan.scan[q];
                HScan &hs = hi.scan[q];
                hs.off = sp.stream_off;
                hs.len = sp.stream_len;
                hs.restart_interval = sp.restart_interval;
                hs.ns = sp.ns;
                for (int c = 0; c < sp.ns; c++) {
                    hs.ci[c] = sp.ci[c];
                    hs.dc[c] = sp.dc[c];
                    hs.ac[c] = sp.ac[c];
                }
            }
        });
        const double t_prep = now_ms();
        RPH_JPEG_STAMP("lane %d: chunk %d prepared (%zu files)", b, k, m);
        // files the walk does not take (more than four scans) go back to the host decoder; they keep their place in the chunk as holes.
        // Work items: one per restart interval where a file has them, else one per file; lanes take them longest first.
        // Segments (three decoding passes over every byte, but a lane per KB) or one lane per file (one pass, as long as the longest file)?
        // The walk of whole files takes ~0.65 us per byte of the longest one while the chunk has fewer lanes than the device (65 536); the
        // segment passes move ~31 GB/s of entropy bytes (5 275 photos of 366 KB: 67 ms against 236; 25 000 files of 158 KB: 129 ms against
        // 103).  Whole-file walks also leave most of the device to the other lane's chunk, so segments must win clearly.
        bool use_segments = ctx->jpeg_seg_bytes != 0;
        if (use_segments && ctx->jpeg_seg_min_bytes > 0) {
            const double t_whole = 0.65e-6 * (double)jobs[idx[first]].len * std::max(1.0, (double)m / 65536.0), t_seg = (double)file_bytes / seg_rate();
            use_segments = t_seg < 0.7 * t_whole;
        }
        std::vector<HItem> items;
        std::vector<uint32_t> item_len;
        std::vector<SegFile> seg_files;
        std::vector<PScan> pscans;            // the scans of the chunk's progressive files
        std::vector<uint32_t> prog_order;     // the progressive files (indices into the chunk)
        std::vector<uint32_t> pwaits;         // scans (indices into pscans) that must have ended before a scan begins (PScan::wait_first / wait_count)
        std::vector<PRef> prefs;              // their AC refinement scans, grouped by plane (file order within a plane)
        struct PlaneRefs {
            uint32_t r, first[3], count[3];
        };
        std::vector<PlaneRefs> plane_refs;
        uint64_t prog_blocks = 0;             // their blocks: one mask word each
        uint64_t prog_corr = 0;               // records of their AC refinement scans
        uint64_t prog_dcb = 0;                // bytes of their DC refinement scans
        uint32_t n_segs = 0;
        bool all_one_scan = true;  // every sequential file of the chunk has one scan: its MCUs cover all blocks of its components
        items.reserve(m);
        for (size_t i = first; i < last; i++) {
            Job &j = jobs[idx[i]];
            if (j.status == RPH_ERR_UNSUPPORTED || j.status == RPH_ERR_CAPACITY) leftover.push_back(idx[i]);
            if (j.status != RPH_OK) continue;
            const uint32_t r = (uint32_t)(i - first);
            if (j.frame.progressive) {  // one lane per scan (jpeg_prog_kernel)
                himgs[r].mask_first = (uint32_t)prog_blocks;
                prog_blocks += j.frame.total_blocks;
                himgs[r].pscan_first = (uint32_t)pscans.size();
                himgs[r].pscan_count = (uint32_t)j.plan.prog.size();
                const size_t p0 = pscans.size();
                for (const rphj::ScanPlan &sp : j.plan.prog) {
                    PScan ps;
                    ps.off = sp.stream_off, ps.len = sp.stream_len, ps.ns = sp.ns, ps.ss = sp.ss, ps.se = sp.se, ps.ah = sp.ah, ps.al = sp.al;
                    for (int c = 0; c < 3; c++) ps.ci[c] = sp.ci[c], ps.dc[c] = sp.dc[c];
                    ps.ac = sp.ac[0];
                    ps.image = r;
                    ps.corr_first = 0;
                    ps.dcb[0] = ps.dcb[1] = ps.dcb[2] = 0;
                    if (sp.ss == 0 && sp.ah > 0) {
                        for (int c = 0; c < sp.ns && c < 3; c++) {
                            const rphj::Comp &kc = j.frame.comp[sp.ci[c]];
                            ps.dcb[c] = (uint32_t)prog_dcb;
                            prog_dcb += (uint64_t)kc.blocks_w * kc.blocks_h;
                        }
                    }
                    if (sp.ss > 0 && sp.ah > 0) {
                        const rphj::Comp &kc = j.frame.comp[sp.ci[0]];
                        ps.corr_first = (uint32_t)prog_corr;
                        prog_corr += (uint64_t)kc.real_bw * kc.real_bh;
                    }
                    // The scans this one must come after: the earlier scans of the file that share a component and a coefficient with it.
                    // An AC scan follows up to two of them (the longest) block by block (PScan::chase); the others must have ended.
                    ps.chase[0] = ps.chase[1] = PSCAN_NONE;
                    ps.wait_first = (uint32_t)pwaits.size();
                    uint32_t my_comps = 0;
                    for (uint32_t c = 0; c < ps.ns && c < 3; c++) my_comps |= 1u << ps.ci[c];
                    uint32_t deps[rphj::MAX_PROG_SCANS], n_deps = 0;
                    for (size_t q = p0; q < pscans.size(); q++) {
                        const PScan &e = pscans[q];
                        uint32_t its = 0;
                        for (uint32_t c = 0; c < e.ns && c < 3; c++) its |= 1u << e.ci[c];
                        if ((its & my_comps) && e.ss <= ps.se && ps.ss <= e.se) deps[n_deps++] = (uint32_t)q;
                    }
                    if (ps.ss > 0) {  // (AC scans of one component walk the same raster of blocks): the two longest are followed
                        for (int k = 0; k < 2; k++) {
                            uint32_t best = PSCAN_NONE;
                            for (uint32_t d = 0; d < n_deps; d++)
                                if (deps[d] != PSCAN_NONE && (best == PSCAN_NONE || pscans[deps[d]].len > pscans[deps[best]].len)) best = d;
                            if (best == PSCAN_NONE) break;
                            ps.chase[k] = deps[best];
                            deps[best] = PSCAN_NONE;
                        }
                        for (uint32_t d = 0; d < n_deps; d++)
                            if (deps[d] != PSCAN_NONE) pwaits.push_back(deps[d]);
                    } else if (ps.ah == 0) {
                        pwaits.insert(pwaits.end(), deps, deps + n_deps);
                    }  // (a DC refinement scan writes its bits beside the coefficients and reads nothing: it waits for nobody)
                    ps.wait_count = (uint32_t)pwaits.size() - ps.wait_first;
                    pscans.push_back(ps);
                }
                PlaneRefs pr;
                pr.r = r;
                for (uint32_t c = 0; c < 3; c++) {
                    pr.first[c] = (uint32_t)prefs.size();
                    for (size_t q = p0; q < pscans.size(); q++) {
                        const PScan &x = pscans[q];
                        if (x.ss > 0 && x.ah > 0 && x.ci[0] == c) prefs.push_back(PRef{x.corr_first, x.al});
                        if (x.ss == 0 && x.ah > 0)
                            for (uint32_t k = 0; k < x.ns && k < 3; k++)
                                if (x.ci[k] == c) prefs.push_back(PRef{x.dcb[k], x.al | PREF_DC});
                    }
                    pr.count[c] = (uint32_t)prefs.size() - pr.first[c];
                }
                plane_refs.push_back(pr);
                prog_order.push_back(r);
                continue;
            }
            if (j.plan.n_scans != 1 || j.plan.scan[0].ns != j.frame.ncomp) all_one_scan = false;
            if (j.frame.ncomp == 1 && (j.frame.comp[0].blocks_w != j.frame.comp[0].real_bw || j.frame.comp[0].blocks_h != j.frame.comp[0].real_bh))
                all_one_scan = false;  // (a one-component scan walks the real blocks only: a padded grid keeps its zeroing)
            if (j.marks.empty()) {
                const rphj::ScanPlan &sp0 = j.plan.scan[0];
                if (use_segments && j.plan.n_scans == 1 && sp0.restart_interval == 0 && sp0.stream_len >= ctx->jpeg_seg_min_bytes && sp0.stream_len < ((uint32_t)1 << 28)) {
                    // a long stream without restart markers: cut into segments that synchronise on the device (jpeg_device.h)
                    const rphj::Frame &f = j.frame;
                    SegFile sf;
                    sf.image = r;
                    sf.first_seg = n_segs;
                    sf.n_segs = (sp0.stream_len + ctx->jpeg_seg_bytes - 1) / ctx->jpeg_seg_bytes;
                    sf.first_item = 0;  // set below, behind the host's items
                    sf.total_mcus = sp0.ns == 1 ? f.comp[sp0.ci[0]].real_bw * f.comp[sp0.ci[0]].real_bh : f.mcus_x * f.mcus_y;
                    sf.scan_bits = sp0.stream_len * 8;
                    n_segs += sf.n_segs;
                    seg_files.push_back(sf);
                    continue;
                }
                items.push_back(HItem{r, HITEM_ALL_SCANS, 0, 0, 0});
                item_len.push_back((uint32_t)std::min<size_t>(j.len, 0xFFFFFFFFu));
                continue;
            }
            const rphj::ScanPlan &sp = j.plan.scan[0];
            const rphj::Frame &f = j.frame;
            const uint64_t mcus = sp.ns == 1 ? (uint64_t)f.comp[sp.ci[0]].real_bw * f.comp[sp.ci[0]].real_bh : (uint64_t)f.mcus_x * f.mcus_y;
            const uint32_t n_int = (uint32_t)j.marks.size() + 1;
            for (uint32_t k = 0; k < n_int; k++) {
                const uint32_t off = k ? j.marks[k - 1] : 0, end = k + 1 < n_int ? j.marks[k] : sp.stream_len;
                const uint64_t m_first = (uint64_t)k * sp.restart_interval;
                items.push_back(HItem{r, 0, (uint32_t)m_first, (uint32_t)std::min<uint64_t>(sp.restart_interval, mcus - m_first), off});
                item_len.push_back(end > off ? end - off : 0);
            }
        }
        RPH_JPEG_STAMP("lane %d: chunk %d items listed", b, k);
        // The lanes of the launch: 64 files of one scan script make a batch, and wave k of a batch walks the k-th scan of each of them -- one
        // kind of scan per wave, and a scan's producers in earlier workgroups of the same launch, which start first (jpeg_kernels.hip).
        std::vector<uint32_t> pitems;
        {
            auto signature = [&](uint32_t r) {
                uint64_t h = 1469598103934665603ull;
                for (uint32_t q = 0; q < himgs[r].pscan_count; q++) {
                    const PScan &x = pscans[himgs[r].pscan_first + q];
                    const uint32_t f[6] = {x.ns, x.ss, x.se, x.ah, x.al, x.ci[0] | (x.ci[1] << 8) | (x.ci[2] << 16)};
                    for (uint32_t v : f) h = (h ^ v) * 1099511628211ull;
                }
                return h;
            };
            std::vector<std::pair<uint64_t, uint32_t>> sig(prog_order.size());  // (script, file), files of a script in the chunk's order: longest first
            for (size_t t = 0; t < prog_order.size(); t++) sig[t] = {signature(prog_order[t]), prog_order[t]};
            std::stable_sort(sig.begin(), sig.end(), [](const std::pair<uint64_t, uint32_t> &a, const std::pair<uint64_t, uint32_t> &b) { return a.first < b.first; });
            // A wave = (batch, k-th scan).  Waves go into the grid by how much work still hangs on them -- their own (bytes x the rate of their
            // kind of scan) plus the longest chain of scans behind them -- so the scans on a file's critical path (libjpeg's script: luma 6-63,
            // then its two refinements) start at once for every batch and the short scans fill the slots that are left; a producer always
            // ranks above its consumers, i.e. comes first in the grid.
            struct Wave {
                double rank;
                uint32_t batch, first, count, k;  // files sig[first .. first + count), their k-th scans
            };
            std::vector<Wave> waves;
            uint32_t n_batches = 0;
            for (size_t t0 = 0; t0 < sig.size();) {
                size_t t1 = t0;
                while (t1 < sig.size() && t1 - t0 < 64 && sig[t1].first == sig[t0].first) t1++;
                const HImage &h0 = himgs[sig[t0].second];  // (the batch's longest file stands for all of them)
                const uint32_t n_scans = h0.pscan_count;
                std::vector<double> rank(n_scans, 0.0);
                for (uint32_t q = n_scans; q-- > 0;) {
                    const PScan &x = pscans[h0.pscan_first + q];
                    double behind = 0.0;
                    for (uint32_t c = q + 1; c < n_scans; c++) {  // scans that wait for q or follow it
                        const PScan &y = pscans[h0.pscan_first + c];
                        bool dep = y.chase[0] == h0.pscan_first + q || y.chase[1] == h0.pscan_first + q;
                        for (uint32_t w = 0; w < y.wait_count && !dep; w++) dep = pwaits[y.wait_first + w] == h0.pscan_first + q;
                        if (dep) behind = std::max(behind, rank[c]);
                    }
                    rank[q] = (double)x.len * (x.ss && x.ah ? 1.05 : 0.55) + 1.0 + behind;
                }
                for (uint32_t q = 0; q < n_scans; q++) waves.push_back(Wave{rank[q], n_batches, (uint32_t)t0, (uint32_t)(t1 - t0), q});
                n_batches++;
                t0 = t1;
            }
            std::stable_sort(waves.begin(), waves.end(), [](const Wave &a, const Wave &b) { return a.rank > b.rank; });
            pitems.reserve(waves.size() * 64);
            for (const Wave &w : waves)
                for (uint32_t l = 0; l < 64; l++) pitems.push_back(l < w.count ? himgs[sig[w.first + l].second].pscan_first + w.k : PSCAN_NONE);
        }
        RPH_JPEG_STAMP("lane %d: chunk %d waves ordered", b, k);
        std::vector<uint32_t> order(items.size());
        for (uint32_t t = 0; t < order.size(); t++) order[t] = t;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return item_len[a] > item_len[b]; });
        // the segments' items follow the host's: the device writes them, and the lanes take them as they lie (they are short, so they go last)
        const uint32_t n_ordered = (uint32_t)items.size();
        uint32_t n_items = n_ordered;
        for (SegFile &sf : seg_files) {
            sf.first_item = n_items;
            n_items += sf.n_segs;
        }
        struct {
            uint32_t n;
            size_t size() const { return n; }
            bool empty() const { return n == 0; }
        } luts{store.size()};  // (the tables' count: they lie in the store's blocks)
        // ---- meta buffer: reconstruction descriptors | HImage | order | tables
        const size_t recon_bytes = m * (3 * sizeof(JPlane) + sizeof(JImage) + 3 * 128);
        // (the items of the segments exist on the device only: d_meta has room for them, the upload stops before them)
        const size_t off_himg = align_up(recon_bytes, 16), off_order = off_himg + m * sizeof(HImage), off_luts = align_up(off_order + order.size() * 4, 16),
                     off_segf = align_up(off_luts + luts.size() * sizeof(rphj::DeviceLut), 16), off_pscan = align_up(off_segf + seg_files.size() * sizeof(SegFile), 16),
                     off_porder = off_pscan + pscans.size() * sizeof(PScan), off_pitems = off_porder + prog_order.size() * 4,
                     off_prefs = align_up(off_pitems + pitems.size() * 4, 16), off_pwaits = align_up(off_prefs + prefs.size() * sizeof(PRef), 16),
                     off_items = align_up(off_pwaits + pwaits.size() * 4, 16),
                     upload_bytes = off_items + items.size() * sizeof(HItem), meta_bytes = off_items + (size_t)n_items * sizeof(HItem);
        RPH_TRY(S.meta.reserve(meta_bytes));
        const size_t segwork = align_up((size_t)n_segs * sizeof(SegState), 16) + rph_jpeg_segment_work_bytes(n_segs);
        if (n_segs && S.segwork_cap < segwork) {
            if (S.d_segwork) (void)hipFree(S.d_segwork);
            S.d_segwork = nullptr, S.segwork_cap = 0;
            RPH_HIP_CHECK(hipMalloc((void **)&S.d_segwork, segwork + segwork / 4));
            S.segwork_cap = segwork + segwork / 4;
        }
        if (prog_blocks >= ((uint64_t)1 << 32)) return RPH_ERR_CAPACITY;  // (a chunk's coefficients are capped far below: 2^32 blocks are 512 GB)
        if (prog_corr >= ((uint64_t)1 << 32) || prog_dcb >= ((uint64_t)1 << 32)) return RPH_ERR_CAPACITY;
        if (prog_dcb && S.pdc_cap < prog_dcb) {
            if (S.d_pdc) (void)hipFree(S.d_pdc);
            S.d_pdc = nullptr, S.pdc_cap = 0;
            RPH_HIP_CHECK(hipMalloc((void **)&S.d_pdc, prog_dcb + prog_dcb / 4 + 64));
            S.pdc_cap = prog_dcb + prog_dcb / 4 + 64;
        }
        if (prog_corr && S.pcorr_cap < prog_corr * sizeof(PCorr)) {
            if (S.d_pcorr) (void)hipFree(S.d_pcorr);
            S.d_pcorr = nullptr, S.pcorr_cap = 0;
            RPH_HIP_CHECK(hipMalloc((void **)&S.d_pcorr, prog_corr * sizeof(PCorr) + prog_corr * 4));
            S.pcorr_cap = prog_corr * sizeof(PCorr) + prog_corr * 4;
        }
        // (one mask word per block, and behind them one progress word per scan)
        const size_t pmask_need = prog_blocks * 8 + align_up(pscans.size() * 4, 16);
        if (prog_blocks && S.pmask_cap < pmask_need) {
            if (S.d_pmask) (void)hipFree(S.d_pmask);
            S.d_pmask = nullptr, S.pmask_cap = 0;
            RPH_HIP_CHECK(hipMalloc((void **)&S.d_pmask, pmask_need + pmask_need / 4));
            S.pmask_cap = pmask_need + pmask_need / 4;
        }
        ChunkDesc D;
        std::vector<size_t> subs;
        RPH_TRY(build_descriptors(jobs, idx, first, last, flavour, false, P.recon_coef_bytes[b], S.meta.h, 0, D, subs));
        memcpy(S.meta.h + off_himg, himgs.data(), m * sizeof(HImage));
        memcpy(S.meta.h + off_items, items.data(), items.size() * sizeof(HItem));
        memcpy(S.meta.h + off_order, order.data(), order.size() * 4);
        if (!luts.empty()) store.copy_luts(reinterpret_cast<rphj::DeviceLut *>(S.meta.h + off_luts));
        if (n_segs) memcpy(S.meta.h + off_segf, seg_files.data(), seg_files.size() * sizeof(SegFile));
        if (!prog_order.empty()) {
            memcpy(S.meta.h + off_pscan, pscans.data(), pscans.size() * sizeof(PScan));
            memcpy(S.meta.h + off_porder, prog_order.data(), prog_order.size() * 4);
            memcpy(S.meta.h + off_pitems, pitems.data(), pitems.size() * 4);
            if (!pwaits.empty()) memcpy(S.meta.h + off_pwaits, pwaits.data(), pwaits.size() * 4);
            if (!prefs.empty()) {
                memcpy(S.meta.h + off_prefs, prefs.data(), prefs.size() * sizeof(PRef));
                JPlane *hp = reinterpret_cast<JPlane *>(S.meta.h + D.off_planes);
                for (const PlaneRefs &pr : plane_refs) {
                    if (D.plane_of[pr.r] == UINT32_MAX) continue;
                    const int nc = jobs[idx[first + pr.r]].frame.ncomp;
                    for (int c = 0; c < nc && c < 3; c++) hp[D.plane_of[pr.r] + c].ref_first = pr.first[c], hp[D.plane_of[pr.r] + c].ref_count = pr.count[c];
                }
                D.d_refs = reinterpret_cast<const PRef *>(S.meta.d + off_prefs);
                D.d_corr = S.d_pcorr;
                D.d_dcbits = S.d_pdc;
            }
        }
        RPH_JPEG_STAMP("lane %d: chunk %d descriptors written", b, k);
        // ---- device: streams up, zeroed coefficients, the walk, then reconstruction + hashing sub-batch by sub-batch
        const double t_desc = now_ms();
        const bool tr = trace_on();  // RPH_JPEG_TRACE: synchronise after every phase and print where the time goes (stderr)
        double t_up = 0, t_seg = 0, t_zero = 0, t_walk = 0, t_rec = 0;
        auto lap = [&](double &t) {
            if (tr) {
                (void)hipStreamSynchronize(s);
                t = now_ms();
            }
        };
        ResView R(S.res.d, S.res_images);
        zero_results(S, m, out);
        if (n_items || !prog_order.empty()) {
            RPH_HIP_CHECK(hipMemcpyAsync(S.stream_bytes.d, S.stream_bytes.h, file_bytes + 64, hipMemcpyHostToDevice, s));
            RPH_HIP_CHECK(hipMemcpyAsync(S.meta.d, S.meta.h, upload_bytes, hipMemcpyHostToDevice, s));
            lap(t_up);
            if (n_segs) {  // streams without markers: their segments find their entries and become walk items
                SegState *d_segs = reinterpret_cast<SegState *>(S.d_segwork);
                void *d_segtab = S.d_segwork + align_up((size_t)n_segs * sizeof(SegState), 16);
                RPH_TRY(rph_jpeg_launch_segments(s, S.stream_bytes.d, reinterpret_cast<const HImage *>(S.meta.d + off_himg), reinterpret_cast<const SegFile *>(S.meta.d + off_segf),
                                                 (uint32_t)seg_files.size(), d_segs, n_segs, ctx->jpeg_seg_bytes, d_segtab, 8,
                                                 reinterpret_cast<const rphj::DeviceLut *>(S.meta.d + off_luts), (uint32_t)luts.size(), reinterpret_cast<HItem *>(S.meta.d + off_items)));
            }
            lap(t_seg);
            // The coefficients start from zero -- unless the walk writes whole blocks and covers every block of the chunk: sequential files of
            // one scan (interleaved, or one component), no progressive file.  (A file whose walk breaks off is reported as damaged; what its
            // remaining blocks hold is not looked at.)
            if (!(rph_jpeg_walk_writes_whole_blocks(n_items) && all_one_scan && prog_order.empty())) RPH_HIP_CHECK(hipMemsetAsync(d_coef, 0, blocks * 128, s));
            lap(t_zero);
            if (n_items)
                RPH_TRY(rph_jpeg_launch_walk(s, S.stream_bytes.d, reinterpret_cast<const HImage *>(S.meta.d + off_himg), reinterpret_cast<const HItem *>(S.meta.d + off_items),
                                             reinterpret_cast<const uint32_t *>(S.meta.d + off_order), n_ordered, n_items,
                                             reinterpret_cast<const rphj::DeviceLut *>(S.meta.d + off_luts), (uint32_t)luts.size(), d_coef, R.status));
            if (prog_blocks) RPH_HIP_CHECK(hipMemsetAsync(S.d_pmask, 0, prog_blocks * 8 + align_up(pscans.size() * 4, 16), s));
            RPH_TRY(rph_jpeg_launch_prog(s, S.stream_bytes.d, reinterpret_cast<const HImage *>(S.meta.d + off_himg), reinterpret_cast<const PScan *>(S.meta.d + off_pscan),
                                         (uint32_t)pscans.size(), reinterpret_cast<const uint32_t *>(S.meta.d + off_pitems), (uint32_t)pitems.size(),
                                         reinterpret_cast<const uint32_t *>(S.meta.d + off_pwaits), reinterpret_cast<const rphj::DeviceLut *>(S.meta.d + off_luts),
                                         (uint32_t)luts.size(), d_coef, S.d_pmask, reinterpret_cast<uint32_t *>(S.d_pmask + prog_blocks), S.d_pcorr, (size_t)prog_corr, S.d_pdc, (size_t)prog_dcb, R.status));
            lap(t_walk);
            for (size_t q = 0; q < subs.size(); q++) {
                const size_t r0 = subs[q], r1 = q + 1 < subs.size() ? subs[q + 1] : m;
                RPH_TRY(reconstruct_and_hash(ctx, P, b, S, jobs, idx, first, D, r0, r1, d_coef, flavour, out, s));
            }
            lap(t_rec);
            if (tr && n_segs) rph_jpeg_debug_segment_stats(s, S.d_segwork + align_up((size_t)n_segs * sizeof(SegState), 16), n_segs);  // (behind the timed phases)
            if (tr)
                fprintf(stderr, "[rph_jpeg] chunk of %zu files in %zu lanes (%.1f MB of entropy bytes, %.2f GB of coefficients, %zu tables, %zu sub-batches): prepare %.1f ms, "
                                "descriptors %.1f ms, upload %.1f ms, %u segments %.1f ms, zero %.1f ms, walk %.1f ms, reconstruct + hash %.1f ms\n",
                        m, (size_t)n_items, file_bytes / 1e6, blocks * 128 / 1e9, luts.size(), subs.size(), t_prep - t0, t_desc - t_prep, t_up - t_desc, n_segs, t_seg - t_up, t_zero - t_seg, t_walk - t_zero, t_rec - t_walk);
        }
        RPH_HIP_CHECK(hipEventRecord(S.done, s));
        RPH_JPEG_STAMP("lane %d: chunk %d enqueued", b, k);
        pend[b].active = true;
        pend[b].first = first;
        pend[b].last = last;
        first = last;
    }
    for (int t = 0; t < lanes; t++) RPH_TRY(finish((k + t) % lanes));  // oldest first
    // files handed to the host decoder start over there
    for (uint32_t g : leftover) jobs[g].status = RPH_OK;
    return RPH_OK;
}

int run_batch(rph_ctx *ctx, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads, Outputs out, Jobs *prepared = nullptr)
{
    if (flavour != RPH_JPEG_ZUNE && flavour != RPH_JPEG_LIBJPEG) {
        rph_set_error("rph_jpeg: unknown flavour %d", flavour);
        return RPH_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(ctx->jpeg_mu);
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->jpeg) ctx->jpeg = new JpegPipe();
    JpegPipe &P = *static_cast<JpegPipe *>(ctx->jpeg);
    unsigned threads = n_threads ? n_threads : default_threads();
    threads = std::min(threads, 256u);

    g_trace_t0 = now_ms();
    RPH_JPEG_STAMP("call: %u files", n);
    if (!prepared && !P.jobs_cache) {
        P.jobs_cache = new Jobs();
        P.jobs_cache_free = [](void *p) { delete static_cast<Jobs *>(p); };
    }
    Jobs &jobs = prepared ? *prepared : *static_cast<Jobs *>(P.jobs_cache);
    if (!prepared && jobs.size() < n) jobs.resize(n);  // (grown by this thread: first-touching the storage from the parsing threads is 5x slower, page faults under contention)
    if (!prepared)
        parallel_for(0, n, n >= 1024 ? threads : 1, [&](size_t i) {
            Job &j = jobs[i];
            j.data = data[i];
            j.len = len[i];
            j.pre = nullptr;
            j.status = (j.data && j.len) ? rphj::parse_frame(j.data, j.len, j.frame) : RPH_ERR_INVALID_ARG;
            if (j.status == RPH_OK && (j.frame.total_blocks * 128 > MAX_IMAGE_COEF_BYTES || !frame_is_plausible(j.frame, j.len))) j.status = RPH_ERR_UNSUPPORTED;
        });
    RPH_JPEG_STAMP("frames parsed");
    // Which files walk their Huffman streams on the device?  In automatic mode, those for which it is estimated to pay:
    //  * sequential files when the call has lanes to fill -- one per file, or one per restart interval where the frame header announces
    //    them (a few dozen photos with restart markers are thousands of short streams); a long stream without markers is cut into
    //    segments, whose passes cost ~10 ms of launches before they scale: counted as a lane per 8 KB, so that ~50 photos qualify (16 host
    //    threads decode 130 MB/s each);
    //  * progressive files, one lane each whatever their size, when the host threads (~41 MB/s each) would need longer for all of them
    //    than the device needs for the longest (~0.6 us per byte of the file: its scans follow one another block by block).
    std::vector<uint32_t> host_idx, dev_idx;
    const bool may_device = ctx->jpeg_entropy != 0 && out.want_hash && !prepared;
    uint64_t lanes = 0, prog_bytes = 0, prog_longest = 0;
    for (uint32_t i = 0; i < n; i++) {
        const Job &j = jobs[i];
        if (j.status != RPH_OK || !may_device) continue;
        const rphj::Frame &f = j.frame;
        if (f.progressive) {
            prog_bytes += j.len;
            prog_longest = std::max<uint64_t>(prog_longest, j.len);
        } else if (f.restart_interval) {
            lanes += ((uint64_t)f.mcus_x * f.mcus_y + f.restart_interval - 1) / f.restart_interval;
        } else if (ctx->jpeg_seg_bytes && j.len >= ctx->jpeg_seg_min_bytes) {
            lanes += std::max<uint64_t>(1, j.len / 8192);
        } else {
            lanes += 1;
        }
    }
    const bool seq_on_device = may_device && (ctx->jpeg_entropy == 1 || lanes >= DEVICE_ENTROPY_MIN_FILES);
    const bool prog_on_device = may_device && ctx->jpeg_progressive_on_device &&
                                (ctx->jpeg_entropy == 1 || (double)prog_bytes / ((double)threads * 41e6) > 0.6e-6 * (double)prog_longest);
    for (uint32_t i = 0; i < n; i++) {
        const Job &j = jobs[i];
        if (j.status == RPH_OK && (j.frame.progressive ? prog_on_device : seq_on_device))
            dev_idx.push_back(i);
        else
            host_idx.push_back(i);
    }
    if (!dev_idx.empty()) {
        std::vector<uint32_t> leftover;
        RPH_TRY(run_device_entropy(ctx, P, jobs, dev_idx, flavour, threads, out, leftover));
        host_idx.insert(host_idx.end(), leftover.begin(), leftover.end());
        std::sort(host_idx.begin(), host_idx.end());
    }
    if (!host_idx.empty()) RPH_TRY(run_host_entropy(ctx, P, jobs, host_idx, flavour, threads, out));
    RPH_JPEG_STAMP("all chunks done");
    int worst = RPH_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (out.status) out.status[i] = jobs[i].status;
        if (jobs[i].status != RPH_OK) worst = jobs[i].status;
    }
    if (n == 1 && worst != RPH_OK) rph_set_error("rph_jpeg: not decodable (status %d)", worst);
    return n == 1 ? worst : RPH_OK;  // a batch reports per image (status / valid); one image reports itself
}

}  // namespace

void rph_jpeg_forget_threads(rph_ctx *ctx)  // rph_shutdown: no caller is inside the library any more
{
    for (void *p : ctx->jpeg_thread_buffers) (void)hipHostFree(p);
    ctx->jpeg_thread_buffers.clear();
}

void rph_jpeg_forget(rph_ctx *ctx)
{
    if (ctx->jpeg) {
        JpegPipe *P = static_cast<JpegPipe *>(ctx->jpeg);
        P->release();
        delete P;
        ctx->jpeg = nullptr;
    }
}

extern "C" {

int rph_jpeg_info(const uint8_t *data, size_t len, uint32_t *w, uint32_t *h, uint32_t *channels)
{
    return rph_guarded("rph_jpeg_info", [&]() -> int {
        if (!data || !w || !h || !channels) {
            rph_set_error("rph_jpeg_info: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        rphj::Frame f;
        const int rc = rphj::parse_frame(data, len, f);
        if (rc != RPH_OK) {
            rph_set_error("rph_jpeg_info: %s", rc == RPH_ERR_UNSUPPORTED ? "unsupported kind of JPEG" : "not a JPEG stream");
            return rc;
        }
        *w = f.w;
        *h = f.h;
        *channels = (uint32_t)f.ncomp;
        return RPH_OK;
    });
}

int rph_jpeg_coefficients(const uint8_t *data, size_t len, uint32_t *geometry, uint16_t *qt, int16_t *coef, size_t cap_blocks, uint64_t *total_blocks)
{
    return rph_guarded("rph_jpeg_coefficients", [&]() -> int {
        if (!data || !geometry || !qt || !total_blocks) {
            rph_set_error("rph_jpeg_coefficients: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        rphj::Frame f;
        int rc = rphj::parse_frame(data, len, f);
        if (rc != RPH_OK) return rc;
        *total_blocks = f.total_blocks;
        std::vector<int16_t> tmp;
        int16_t *dst = coef;
        if (!coef) {
            tmp.resize((size_t)f.total_blocks * 64);
            dst = tmp.data();
        } else if (cap_blocks < f.total_blocks) {
            rph_set_error("rph_jpeg_coefficients: %llu blocks, capacity %zu", (unsigned long long)f.total_blocks, cap_blocks);
            return RPH_ERR_CAPACITY;
        }
        rc = rphj::decode_coefficients(data, len, f, dst);
        if (rc != RPH_OK) {
            rph_set_error("rph_jpeg_coefficients: entropy decoding failed (status %d)", rc);
            return rc;
        }
        for (int c = 0; c < f.ncomp; c++) {
            const rphj::Comp &k = f.comp[c];
            uint32_t *g = geometry + 8 * c;
            g[0] = k.blocks_w, g[1] = k.blocks_h, g[2] = k.H, g[3] = k.V, g[4] = k.tq, g[5] = k.samp_w, g[6] = k.samp_h, g[7] = (uint32_t)k.first_block;
        }
        for (int t = 0; t < 4; t++) {
            if (f.qt_present[t])
                memcpy(qt + 64 * t, f.qt[t], 128);
            else
                memset(qt + 64 * t, 0, 128);
        }
        return RPH_OK;
    });
}

int rph_jpeg_set_segments(rph_ctx *ctx, uint32_t min_stream_bytes, uint32_t segment_bytes)
{
    if (!ctx || (segment_bytes && (segment_bytes < 64 || segment_bytes > 65536 || (segment_bytes & 3)))) {
        rph_set_error("rph_jpeg_set_segments: invalid argument");
        return RPH_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(ctx->jpeg_mu);
    ctx->jpeg_seg_min_bytes = min_stream_bytes;
    ctx->jpeg_seg_bytes = segment_bytes;
    return RPH_OK;
}

int rph_jpeg_release(rph_ctx *ctx)
{
    if (!ctx) {
        rph_set_error("rph_jpeg_release: null argument");
        return RPH_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(ctx->jpeg_mu);
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    rph_jpeg_forget(ctx);
    return RPH_OK;
}

int rph_jpeg_set_entropy(rph_ctx *ctx, int where)
{
    if (!ctx || where < 0 || where > 3) {
        rph_set_error("rph_jpeg_set_entropy: invalid argument");
        return RPH_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(ctx->jpeg_mu);
    ctx->jpeg_entropy = where == 3 ? 1 : where;
    ctx->jpeg_progressive_on_device = where == 3 ? 0 : 1;
    return RPH_OK;
}

int rph_jpeg_decode(rph_ctx *ctx, const uint8_t *data, size_t len, int flavour, uint8_t *pixels_out)
{
    return rph_guarded("rph_jpeg_decode", [&]() -> int {
        if (!ctx || !data || !pixels_out) {
            rph_set_error("rph_jpeg_decode: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        Outputs o;
        o.pixels = pixels_out;
        o.want_hash = false;
        return run_batch(ctx, &data, &len, 1, flavour, 1, o);
    });
}

// One file per call from many threads (the reference's scan loop: load_image_fast + generate_pdq_features on every rayon worker,
// scanner.rs:1202, :1410).  Callers that arrive while a batch is on its way wait and leave together as the next batch, which one of
// them (the leader) runs through the reconstruction + hashing stages of rph_jpeg_pdq_hash_batch.  Every caller undoes the entropy
// coding of its own file first, on its own core, into a pinned buffer of its own that the copy engine reads directly.
namespace {
struct OneRequest {
    const uint8_t *data;
    size_t len;
    int flavour;
    rphj::Frame frame;
    const int16_t *coef;  // the caller's pinned buffer, decoded by the caller
    uint8_t hash[32];
    float quality, coeffs[256];
    uint8_t valid;
    int32_t status;
    bool want_coeffs, done;
};
// One pinned coefficient buffer per calling thread (a scan worker decodes thousands of files into it).  It belongs to the context:
// rph_shutdown frees it (a thread-local destructor would call into the HIP runtime at thread or process exit, possibly after the
// runtime is gone); `serial` tells a later context at the same address from the one that owned the buffer.
struct ThreadPinned {
    int16_t *p = nullptr;
    size_t cap = 0;
    rph_ctx *owner = nullptr;
    uint64_t serial = 0;
};
thread_local ThreadPinned tls_coef;
}  // namespace

int rph_jpeg_pdq_hash_one(rph_ctx *ctx, const uint8_t *data, size_t len, int flavour, uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *valid_out)
{
    return rph_guarded("rph_jpeg_pdq_hash_one", [&]() -> int {
        if (!ctx || !data || !hash32_out || (flavour != RPH_JPEG_ZUNE && flavour != RPH_JPEG_LIBJPEG)) {
            rph_set_error("rph_jpeg_pdq_hash_one: invalid argument");
            return RPH_ERR_INVALID_ARG;
        }
        // ---- this thread: frame header and entropy decoding into its own pinned buffer (all callers do this side by side)
        OneRequest me;
        me.data = data, me.len = len, me.flavour = flavour, me.want_coeffs = coeffs_out != nullptr, me.done = false, me.valid = 0, me.quality = 0.f, me.coef = nullptr;
        me.status = rphj::parse_frame(data, len, me.frame);
        if (me.status == RPH_OK && (me.frame.total_blocks * 128 > MAX_IMAGE_COEF_BYTES || !frame_is_plausible(me.frame, len))) me.status = RPH_ERR_UNSUPPORTED;
        if (me.status == RPH_OK) {
            const size_t need = (size_t)me.frame.total_blocks * 128;
            if (tls_coef.owner != ctx || tls_coef.serial != ctx->serial) tls_coef = ThreadPinned();  // another (or an earlier) context's buffer is not ours to use
            if (tls_coef.cap < need) {
                RPH_HIP_CHECK(hipSetDevice(ctx->device));
                std::lock_guard<std::mutex> reg(ctx->jpeg_qmu);
                if (tls_coef.p) {
                    auto &v = ctx->jpeg_thread_buffers;
                    v.erase(std::remove(v.begin(), v.end(), (void *)tls_coef.p), v.end());
                    (void)hipHostFree(tls_coef.p);
                }
                tls_coef = ThreadPinned();
                const size_t cap = align_up(need + need / 2, 1 << 20);
                RPH_HIP_CHECK(hipHostMalloc((void **)&tls_coef.p, cap));
                tls_coef.cap = cap;
                tls_coef.owner = ctx;
                tls_coef.serial = ctx->serial;
                ctx->jpeg_thread_buffers.push_back(tls_coef.p);
            }
            me.status = rphj::decode_coefficients(data, len, me.frame, tls_coef.p);
            me.coef = tls_coef.p;
        }
        if (me.status != RPH_OK) {
            memset(hash32_out, 0, 32);
            if (quality_out) *quality_out = 0.f;
            if (coeffs_out) memset(coeffs_out, 0, 1024);
            if (valid_out) *valid_out = 0;
            rph_set_error("rph_jpeg_pdq_hash_one: not decodable here (status %d)", me.status);
            return me.status;
        }
        // ---- the device part: with everyone else who is waiting, as one batch, run by one of them
        std::unique_lock<std::mutex> lk(ctx->jpeg_qmu);
        ctx->jpeg_waiting.push_back(&me);
        while (!me.done) {
            if (ctx->jpeg_leader) {
                ctx->jpeg_qcv.wait(lk);
                continue;
            }
            ctx->jpeg_leader = true;
            std::vector<void *> batch;
            batch.swap(ctx->jpeg_waiting);
            lk.unlock();
            for (int fl = 0; fl < 2; fl++) {  // (callers may ask for different arithmetic flavours: one pass each)
                std::vector<OneRequest *> reqs;
                for (void *p : batch)
                    if (static_cast<OneRequest *>(p)->flavour == fl) reqs.push_back(static_cast<OneRequest *>(p));
                if (reqs.empty()) continue;
                const uint32_t n = (uint32_t)reqs.size();
                int rc = RPH_OK;
                std::vector<uint8_t> hashes((size_t)n * 32), valid(n);
                std::vector<float> quality(n), coeffs;
                std::vector<int32_t> status(n);
                try {
                    bool any_coeffs = false;
                    Jobs jobs(n);
                    for (uint32_t i = 0; i < n; i++) {
                        jobs[i].data = reqs[i]->data, jobs[i].len = reqs[i]->len, jobs[i].frame = reqs[i]->frame, jobs[i].pre = reqs[i]->coef;
                        any_coeffs |= reqs[i]->want_coeffs;
                    }
                    coeffs.resize(any_coeffs ? (size_t)n * 256 : 0);
                    Outputs o;
                    o.hash = hashes.data(), o.quality = quality.data(), o.coeffs = any_coeffs ? coeffs.data() : nullptr, o.valid = valid.data(), o.status = status.data();
                    rc = run_batch(ctx, nullptr, nullptr, n, fl, 1, o, &jobs);
                } catch (...) {
                    rc = RPH_ERR_OOM;
                }
                for (uint32_t i = 0; i < n; i++) {
                    OneRequest &r = *reqs[i];
                    r.status = (rc != RPH_OK && status[i] == RPH_OK) ? rc : status[i];  // a failure of the call itself fails all of its files
                    memcpy(r.hash, &hashes[(size_t)i * 32], 32);
                    r.quality = quality[i];
                    r.valid = r.status == RPH_OK ? valid[i] : 0;
                    if (r.want_coeffs && !coeffs.empty()) memcpy(r.coeffs, &coeffs[(size_t)i * 256], 1024);
                }
            }
            lk.lock();
            for (void *p : batch) static_cast<OneRequest *>(p)->done = true;
            ctx->jpeg_leader = false;
            ctx->jpeg_qcv.notify_all();
        }
        lk.unlock();
        memcpy(hash32_out, me.hash, 32);
        if (quality_out) *quality_out = me.quality;
        if (coeffs_out) memcpy(coeffs_out, me.coeffs, 1024);
        if (valid_out) *valid_out = me.valid;
        if (me.status != RPH_OK) rph_set_error("rph_jpeg_pdq_hash_one: failed (status %d)", me.status);
        return me.status;
    });
}

int rph_jpeg_pdq_hash_batch(rph_ctx *ctx, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads, uint8_t *hash32_out,
                            float *quality_out, float *coeffs_out, uint8_t *dihedral_out, uint8_t *valid_out, int32_t *status_out)
{
    return rph_guarded("rph_jpeg_pdq_hash_batch", [&]() -> int {
        if (!ctx || (n && (!data || !len)) || !hash32_out) {
            rph_set_error("rph_jpeg_pdq_hash_batch: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (n == 0) return RPH_OK;
        Outputs o;
        o.hash = hash32_out;
        o.quality = quality_out;
        o.coeffs = coeffs_out;
        o.dihedral = dihedral_out;
        o.valid = valid_out;
        o.status = status_out;
        return run_batch(ctx, data, len, n, flavour, n_threads, o);
    });
}

}  // extern "C"
