// rph_api.cpp -- extern "C" surface of librupphash_hip.so (see include/rupphash.h).
// Host-pointer entry points stage through device memory and call the *_dev twins; there is no
// CPU implementation of any kernel behind this API.
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "rph_internal.h"

static thread_local char g_err[512] = "";

void rph_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

namespace {
// RAII device buffer for the host-pointer wrappers
struct DevBuf {
    void *p = nullptr;
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes)
    {
        RPH_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
        return RPH_OK;
    }
    template <class T>
    T *as() const
    {
        return reinterpret_cast<T *>(p);
    }
};
#define RPH_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != RPH_OK) return rc_; \
    } while (0)

hipStream_t pick(rph_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }
}  // namespace

static int sweep_host(rph_ctx *ctx, const uint8_t *variants, uint32_t n_variants, const uint8_t *hashes32,
                      const uint8_t *low_conf, const uint8_t *has_features, uint64_t n, uint32_t thr, uint32_t part,
                      uint32_t nparts, rph_edge *edges, uint64_t cap, uint64_t *n_edges_out)
{
    if (!ctx || (!hashes32 && n) || !n_edges_out || (!edges && cap)) {
        rph_set_error("hamming sweep: null argument");
        return RPH_ERR_INVALID_ARG;
    }
    *n_edges_out = 0;
    if (n < 2) return RPH_OK;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    DevBuf d_h, d_v, d_lc, d_hf, d_e, d_cnt;
    RPH_TRY(d_h.alloc(n * 32));
    RPH_HIP_CHECK(hipMemcpyAsync(d_h.p, hashes32, n * 32, hipMemcpyHostToDevice, ctx->stream));
    const uint8_t *rows = (const uint8_t *)d_h.p;
    if (variants) {
        RPH_TRY(d_v.alloc(n * 32 * n_variants));
        RPH_HIP_CHECK(hipMemcpyAsync(d_v.p, variants, n * 32 * n_variants, hipMemcpyHostToDevice, ctx->stream));
        rows = (const uint8_t *)d_v.p;
    } else {
        n_variants = 1;
    }
    if (low_conf) {
        RPH_TRY(d_lc.alloc(n));
        RPH_HIP_CHECK(hipMemcpyAsync(d_lc.p, low_conf, n, hipMemcpyHostToDevice, ctx->stream));
    }
    if (has_features) {
        RPH_TRY(d_hf.alloc(n));
        RPH_HIP_CHECK(hipMemcpyAsync(d_hf.p, has_features, n, hipMemcpyHostToDevice, ctx->stream));
    }
    RPH_TRY(d_e.alloc(cap * sizeof(rph_edge)));
    RPH_TRY(d_cnt.alloc(8));
    RPH_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, 8, ctx->stream));
    RPH_TRY(rph_launch_hamming_sweep(ctx, rows, n_variants, (const uint8_t *)d_h.p, (const uint8_t *)d_lc.p,
                                     (const uint8_t *)d_hf.p, n, thr, part, nparts, (rph_edge *)d_e.p, cap,
                                     (unsigned long long *)d_cnt.p, ctx->stream, ctx->hamming_kernel));
    unsigned long long cnt = 0;
    RPH_HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n_edges_out = cnt;
    const uint64_t take = std::min<uint64_t>(cnt, cap);
    if (take) RPH_HIP_CHECK(hipMemcpy(edges, d_e.p, take * sizeof(rph_edge), hipMemcpyDeviceToHost));
    if (cnt > cap) {
        rph_set_error("hamming sweep: %llu edges found, capacity %llu", cnt, (unsigned long long)cap);
        return RPH_ERR_CAPACITY;
    }
    return RPH_OK;
}

// Run a sweep whose edge count is not known in advance: grow the edge buffer until it fits.
template <class F>
static int sweep_growing(uint64_t n, std::vector<rph_edge> &edges, F &&run)
{
    uint64_t cap = std::max<uint64_t>(1u << 20, 4 * n);
    for (int attempt = 0; attempt < 8; attempt++) {
        edges.resize(cap);
        uint64_t found = 0;
        int rc = run(edges.data(), cap, &found);
        if (rc == RPH_OK) {
            edges.resize(found);
            return RPH_OK;
        }
        if (rc != RPH_ERR_CAPACITY) return rc;
        cap = found + found / 8 + 1024;
    }
    return RPH_ERR_CAPACITY;
}

// 1/3/4 channels, rows do not overlap, images do not overlap (a single image needs no image_stride)
static bool pdq_geometry_ok(uint32_t n, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride)
{
    if (channels != 1 && channels != 3 && channels != 4) return false;
    if (row_stride < (size_t)w * channels) return false;
    return n <= 1 || image_stride >= row_stride * (h ? h - 1 : 0) + (size_t)w * channels;
}

// ---- staging pipe of rph_pdq_hash_batch ----
namespace {
constexpr size_t kPipeChunkBytes = (size_t)64 << 20;
struct HostPipe {
    hipStream_t stream[2] = {nullptr, nullptr};
    uint8_t *h_px[2] = {nullptr, nullptr}, *h_hash[2] = {nullptr, nullptr}, *h_d[2] = {nullptr, nullptr}, *h_v[2] = {nullptr, nullptr};
    float *h_q[2] = {nullptr, nullptr}, *h_c[2] = {nullptr, nullptr};
    void *d_px[2] = {nullptr, nullptr}, *d_hash[2] = {nullptr, nullptr}, *d_q[2] = {nullptr, nullptr}, *d_c[2] = {nullptr, nullptr}, *d_d[2] = {nullptr, nullptr},
         *d_v[2] = {nullptr, nullptr};
    size_t px_bytes = 0;
    uint32_t images = 0;
    void release()
    {
        for (int b = 0; b < 2; b++) {
            if (stream[b]) (void)hipStreamSynchronize(stream[b]);
            for (void *p : {(void *)h_px[b], (void *)h_hash[b], (void *)h_d[b], (void *)h_v[b], (void *)h_q[b], (void *)h_c[b]})
                if (p) (void)hipHostFree(p);
            for (void *p : {d_px[b], d_hash[b], d_q[b], d_c[b], d_d[b], d_v[b]})
                if (p) (void)hipFree(p);
            if (stream[b]) (void)hipStreamDestroy(stream[b]);
        }
        *this = HostPipe();
    }
};

int pipe_of(rph_ctx *ctx, size_t px_bytes, uint32_t images, HostPipe **out)
{
    if (!ctx->pipe) ctx->pipe = new HostPipe();
    HostPipe &P = *static_cast<HostPipe *>(ctx->pipe);
    if (P.px_bytes < px_bytes || P.images < images) {
        const size_t nb = std::max(px_bytes, P.px_bytes);
        const uint32_t ni = std::max(images, P.images);
        P.release();
        for (int b = 0; b < 2; b++) {
            RPH_HIP_CHECK(hipStreamCreateWithFlags(&P.stream[b], hipStreamNonBlocking));
            RPH_HIP_CHECK(hipHostMalloc((void **)&P.h_px[b], nb));
            RPH_HIP_CHECK(hipMalloc(&P.d_px[b], nb));
            RPH_HIP_CHECK(hipHostMalloc((void **)&P.h_hash[b], (size_t)ni * 32));
            RPH_HIP_CHECK(hipHostMalloc((void **)&P.h_q[b], (size_t)ni * 4));
            RPH_HIP_CHECK(hipHostMalloc((void **)&P.h_c[b], (size_t)ni * 1024));
            RPH_HIP_CHECK(hipHostMalloc((void **)&P.h_d[b], (size_t)ni * 256));
            RPH_HIP_CHECK(hipHostMalloc((void **)&P.h_v[b], ni));
            RPH_HIP_CHECK(hipMalloc(&P.d_hash[b], (size_t)ni * 32));
            RPH_HIP_CHECK(hipMalloc(&P.d_q[b], (size_t)ni * 4));
            RPH_HIP_CHECK(hipMalloc(&P.d_c[b], (size_t)ni * 1024));
            RPH_HIP_CHECK(hipMalloc(&P.d_d[b], (size_t)ni * 256));
            RPH_HIP_CHECK(hipMalloc(&P.d_v[b], ni));
        }
        P.px_bytes = nb;
        P.images = ni;
    }
    *out = &P;
    return RPH_OK;
}

// pageable -> pinned with a few threads (one core moves ~10 GB/s, PCIe takes ~55)
void parallel_copy(uint8_t *dst, const uint8_t *src, size_t bytes)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>(std::min(8u, hw), bytes / ((size_t)4 << 20) + 1);
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t part = ((bytes / nt) + 4095) & ~(size_t)4095;
    for (unsigned t = 1; t < nt; t++) {
        const size_t lo = std::min(bytes, part * t), hi = std::min(bytes, part * (t + 1));
        if (hi > lo) th.emplace_back([=] { std::memcpy(dst + lo, src + lo, hi - lo); });
    }
    std::memcpy(dst, src, std::min(bytes, part));
    for (auto &x : th) x.join();
}
}  // namespace

void rph_pipe_forget(rph_ctx *ctx)
{
    if (ctx->pipe) {
        HostPipe *P = static_cast<HostPipe *>(ctx->pipe);
        P->release();
        delete P;
        ctx->pipe = nullptr;
    }
}

extern "C" {

int rph_abi_version(void) { return RPH_ABI_VERSION; }
const char *rph_last_error(void) { return g_err; }
const char *rph_status_string(int s)
{
    switch (s) {
        case RPH_OK: return "ok";
        case RPH_ERR_INVALID_ARG: return "invalid argument";
        case RPH_ERR_NO_DEVICE: return "no usable gfx950 device";
        case RPH_ERR_HIP: return "HIP runtime error";
        case RPH_ERR_OOM: return "out of device memory";
        case RPH_ERR_UNSUPPORTED: return "unsupported input";
        case RPH_ERR_CAPACITY: return "output capacity too small";
    }
    return "unknown status";
}

int rph_init(int device, rph_ctx **out)
{
    return rph_guarded("rph_init", [&]() -> int {
        if (!out) return RPH_ERR_INVALID_ARG;
        *out = nullptr;
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
            rph_set_error("rph_init: no HIP device visible (this library has no CPU fallback)");
            return RPH_ERR_NO_DEVICE;
        }
        if (device < 0 || device >= count) {
            rph_set_error("rph_init: device %d out of range (%d visible)", device, count);
            return RPH_ERR_NO_DEVICE;
        }
        hipDeviceProp_t prop;
        RPH_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            rph_set_error("rph_init: device %d is %s; this build targets gfx950 (MI355X) only", device, prop.gcnArchName);
            return RPH_ERR_NO_DEVICE;
        }
        RPH_HIP_CHECK(hipSetDevice(device));
        rph_ctx *ctx = new rph_ctx();
        static std::atomic<uint64_t> next_serial{1};
        ctx->serial = next_serial.fetch_add(1);
        ctx->device = device;
        ctx->compute_units = prop.multiProcessorCount;
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            rph_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            delete ctx;
            return RPH_ERR_HIP;
        }
        *out = ctx;
        return RPH_OK;
    });
}

int rph_shutdown(rph_ctx *ctx)
{
    if (!ctx) return RPH_ERR_INVALID_ARG;
    rph_batcher_forget(ctx);
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    rph_pipe_forget(ctx);
    rph_resize_forget(ctx);
    rph_jpeg_forget(ctx);
    rph_jpeg_forget_threads(ctx);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (auto &kv : ctx->ll_scratch) (void)hipFree(kv.second.p);
    if (ctx->sink) (void)hipFree(ctx->sink);
    if (ctx->scratch_done) (void)hipEventDestroy(ctx->scratch_done);
    if (ctx->sweep_scratch) (void)hipFree(ctx->sweep_scratch);
    if (ctx->sweep_done) (void)hipEventDestroy(ctx->sweep_done);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RPH_OK;
}

int rph_device_info(rph_ctx *ctx, char *name64, int *cus, uint64_t *total_mem)
{
    if (!ctx) return RPH_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    RPH_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->device));
    if (name64) snprintf(name64, 64, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return RPH_OK;
}

int rph_synchronize(rph_ctx *ctx)
{
    if (!ctx) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPH_OK;
}

void *rph_stream(rph_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int rph_hamming_set_kernel(rph_ctx *ctx, int which)
{
    if (!ctx || which < 0 || which > 4) return RPH_ERR_INVALID_ARG;
    ctx->hamming_kernel = which;
    return RPH_OK;
}

int rph_pdq_set_kernel(rph_ctx *ctx, int which)
{
    if (!ctx || which < 0 || which > 6) return RPH_ERR_INVALID_ARG;
    ctx->pdq_kernel = which;
    return RPH_OK;
}

// ------------------------------------------------------------------------------------------
// PDQ
// ------------------------------------------------------------------------------------------
int rph_pdq_hash_batch_dev(rph_ctx *ctx, const void *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                           size_t row_stride, size_t image_stride, void *d_hash32, void *d_quality, void *d_coeffs,
                           void *d_dihedral, void *d_valid, void *stream)
{
    if (!ctx || (!d_px && n) || !d_hash32 || !pdq_geometry_ok(n, w, h, channels, row_stride, image_stride)) {
        rph_set_error("rph_pdq_hash_batch: invalid argument (n=%u %ux%ux%u row_stride=%zu image_stride=%zu)", n, w, h,
                      channels, row_stride, image_stride);
        return RPH_ERR_INVALID_ARG;
    }
    if (n == 0) return RPH_OK;
    hipStream_t s = pick(ctx, stream);
    std::lock_guard<std::mutex> lock(ctx->mu);
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    if (w < RPH_PDQ_MIN_DIM || h < RPH_PDQ_MIN_DIM) {
        // generate_pdq_features returns None (pdqhash.rs:167-169): valid = 0, outputs zeroed
        RPH_HIP_CHECK(hipMemsetAsync(d_hash32, 0, (size_t)n * 32, s));
        if (d_quality) RPH_HIP_CHECK(hipMemsetAsync(d_quality, 0, (size_t)n * 4, s));
        if (d_coeffs) RPH_HIP_CHECK(hipMemsetAsync(d_coeffs, 0, (size_t)n * 1024, s));
        if (d_dihedral) RPH_HIP_CHECK(hipMemsetAsync(d_dihedral, 0, (size_t)n * 256, s));
        if (d_valid) RPH_HIP_CHECK(hipMemsetAsync(d_valid, 0, n, s));
        return RPH_OK;
    }
    if (w > RPH_PDQ_MAX_DIM || h > RPH_PDQ_MAX_DIM)  // pre-downsample to the <= 512 px thumbnail first (pdqhash.rs:181-191)
        return rph_launch_pdq_resized(ctx, (const uint8_t *)d_px, n, w, h, channels, row_stride, image_stride, (uint8_t *)d_hash32,
                                      (float *)d_quality, (float *)d_coeffs, (uint8_t *)d_dihedral, (uint8_t *)d_valid, s);
    if (ctx->pdq_kernel >= 1 && ctx->pdq_kernel != 5 && w == 512 && h == 512 && (channels == 3 || channels == 1) && (row_stride % 4) == 0 && (image_stride % 4) == 0 &&
        ((uintptr_t)d_px % 4) == 0 && row_stride <= ((size_t)1 << 22)) {  // the fused kernel uses 32-bit in-image offsets
        int rc = rph_launch_pdq_fused512(ctx, (const uint8_t *)d_px, n, row_stride, image_stride, (uint8_t *)d_hash32,
                                         (float *)d_quality, (float *)d_coeffs, (uint8_t *)d_dihedral, (uint8_t *)d_valid, s, channels);
        if (rc != RPH_ERR_UNSUPPORTED) return rc;
    }
    // every other Luma8 geometry from 128 x 128: the streaming single-pass kernel (pdq_stream.hip) -- one wave per image, ~0.4 ms from an
    // image's first byte to its hash however few there are, so calls of a few images (the one-image-per-call queue) keep the multi-pass
    // kernels, which spread an image over the chip (~0.15 ms)
    const bool stream_ok = ctx->pdq_kernel >= 1 && ctx->pdq_kernel != 5 && (n >= RPH_STREAM_MIN_IMAGES || ctx->pdq_kernel == 6);
    if (stream_ok && rph_pdq_stream_supported((const uint8_t *)d_px, w, h, channels, row_stride, image_stride)) {
        return rph_launch_pdq_stream(ctx, (const uint8_t *)d_px, n, w, h, row_stride, image_stride, (uint8_t *)d_hash32, (float *)d_quality, (float *)d_coeffs,
                                     (uint8_t *)d_dihedral, (uint8_t *)d_valid, s);
    }
    if (stream_ok && rph_pdq_stream_color_supported((const uint8_t *)d_px, w, h, channels, row_stride, image_stride))
        return rph_launch_pdq_stream_color(ctx, (const uint8_t *)d_px, n, w, h, channels, row_stride, image_stride, (uint8_t *)d_hash32, (float *)d_quality,
                                           (float *)d_coeffs, (uint8_t *)d_dihedral, (uint8_t *)d_valid, s);
    return rph_launch_pdq_generic(ctx, (const uint8_t *)d_px, n, w, h, channels, row_stride, image_stride, (uint8_t *)d_hash32,
                                  (float *)d_quality, (float *)d_coeffs, (uint8_t *)d_dihedral, (uint8_t *)d_valid, s);
}

int rph_pdq_hash_batch(rph_ctx *ctx, const uint8_t *px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                       size_t row_stride, size_t image_stride, uint8_t *hash32_out, float *quality_out, float *coeffs_out,
                       uint8_t *dihedral_out, uint8_t *valid_out)
{
    return rph_pdq_hash_batch_keep(ctx, px, n, w, h, channels, row_stride, image_stride, hash32_out, quality_out, coeffs_out, dihedral_out, valid_out,
                                   nullptr, nullptr, nullptr);
}

}  // extern "C"

// rph_pdq_hash_batch, optionally leaving device-resident copies of the per-image results behind (rph_multi: the hash blocks
// go straight into the all-gather, multi.cpp).  d_*_keep: device arrays of n records on ctx's device, or nullptr.
int rph_pdq_hash_batch_keep(rph_ctx *ctx, const uint8_t *px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride,
                            size_t image_stride, uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *dihedral_out,
                            uint8_t *valid_out, void *d_hash_keep, void *d_quality_keep, void *d_dihedral_keep)
{
    return rph_guarded("rph_pdq_hash_batch", [&]() -> int {
        if (!ctx || (!px && n) || !hash32_out) {
            rph_set_error("rph_pdq_hash_batch: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (n == 0) return RPH_OK;
        if (!pdq_geometry_ok(n, w, h, channels, row_stride, image_stride)) {
            rph_set_error("rph_pdq_hash_batch: invalid argument (n=%u %ux%ux%u row_stride=%zu image_stride=%zu)", n, w, h, channels,
                          row_stride, image_stride);
            return RPH_ERR_INVALID_ARG;
        }
        RPH_HIP_CHECK(hipSetDevice(ctx->device));
        // Chunks of <= 64 MiB of pixels alternate between two staging sets (pinned host + device) and two streams: while chunk k
        // crosses PCIe and is hashed, the host threads copy chunk k + 1 from the caller's (pageable) memory into the other pinned set.
        const size_t one_image = (size_t)(h ? h - 1 : 0) * row_stride + (size_t)w * channels;
        const size_t per = n > 1 ? image_stride : std::max<size_t>(one_image, 1);
        const uint32_t chunk = (uint32_t)std::min<size_t>(n, std::max<size_t>(1, kPipeChunkBytes / per));
        std::lock_guard<std::mutex> pipe_lock(ctx->pipe_mu);
        HostPipe *P = nullptr;
        RPH_TRY(pipe_of(ctx, per * (chunk - 1) + one_image, chunk, &P));
        struct Pending {
            uint32_t first = 0, m = 0;
            bool active = false;
        } pend[2];
        auto finish = [&](int b) -> int {  // results of the chunk that used set b -> the caller's arrays
            if (!pend[b].active) return RPH_OK;
            RPH_HIP_CHECK(hipStreamSynchronize(P->stream[b]));
            const uint32_t first = pend[b].first, m = pend[b].m;
            std::memcpy(hash32_out + (size_t)first * 32, P->h_hash[b], (size_t)m * 32);
            if (quality_out) std::memcpy(quality_out + first, P->h_q[b], (size_t)m * 4);
            if (coeffs_out) std::memcpy(coeffs_out + (size_t)first * 256, P->h_c[b], (size_t)m * 1024);
            if (dihedral_out) std::memcpy(dihedral_out + (size_t)first * 256, P->h_d[b], (size_t)m * 256);
            if (valid_out) std::memcpy(valid_out + first, P->h_v[b], m);
            pend[b].active = false;
            return RPH_OK;
        };
        int k = 0;
        for (uint32_t first = 0; first < n; first += chunk, k++) {
            const int b = k & 1;
            RPH_TRY(finish(b));
            const uint32_t m = std::min(chunk, n - first);
            const size_t bytes = (size_t)(m - 1) * per + one_image;  // the last image may be shorter than image_stride in the caller's buffer
            parallel_copy(P->h_px[b], px + (size_t)first * per, bytes);
            hipStream_t s = P->stream[b];
            RPH_HIP_CHECK(hipMemcpyAsync(P->d_px[b], P->h_px[b], bytes, hipMemcpyHostToDevice, s));
            const bool want_q = quality_out || d_quality_keep, want_d = dihedral_out || d_dihedral_keep;
            RPH_TRY(rph_pdq_hash_batch_dev(ctx, P->d_px[b], m, w, h, channels, row_stride, per, P->d_hash[b], want_q ? P->d_q[b] : nullptr,
                                           coeffs_out ? P->d_c[b] : nullptr, want_d ? P->d_d[b] : nullptr, valid_out ? P->d_v[b] : nullptr, s));
            if (d_hash_keep) RPH_HIP_CHECK(hipMemcpyAsync((uint8_t *)d_hash_keep + (size_t)first * 32, P->d_hash[b], (size_t)m * 32, hipMemcpyDeviceToDevice, s));
            if (d_quality_keep) RPH_HIP_CHECK(hipMemcpyAsync((float *)d_quality_keep + first, P->d_q[b], (size_t)m * 4, hipMemcpyDeviceToDevice, s));
            if (d_dihedral_keep)
                RPH_HIP_CHECK(hipMemcpyAsync((uint8_t *)d_dihedral_keep + (size_t)first * 256, P->d_d[b], (size_t)m * 256, hipMemcpyDeviceToDevice, s));
            RPH_HIP_CHECK(hipMemcpyAsync(P->h_hash[b], P->d_hash[b], (size_t)m * 32, hipMemcpyDeviceToHost, s));
            if (quality_out) RPH_HIP_CHECK(hipMemcpyAsync(P->h_q[b], P->d_q[b], (size_t)m * 4, hipMemcpyDeviceToHost, s));
            if (coeffs_out) RPH_HIP_CHECK(hipMemcpyAsync(P->h_c[b], P->d_c[b], (size_t)m * 1024, hipMemcpyDeviceToHost, s));
            if (dihedral_out) RPH_HIP_CHECK(hipMemcpyAsync(P->h_d[b], P->d_d[b], (size_t)m * 256, hipMemcpyDeviceToHost, s));
            if (valid_out) RPH_HIP_CHECK(hipMemcpyAsync(P->h_v[b], P->d_v[b], m, hipMemcpyDeviceToHost, s));
            pend[b].first = first;
            pend[b].m = m;
            pend[b].active = true;
        }
        RPH_TRY(finish(k & 1));
        RPH_TRY(finish((k + 1) & 1));
        return RPH_OK;
    });
}

extern "C" {

int rph_pdq_hashes_from_coeffs_dev(rph_ctx *ctx, const void *d_coeffs, uint32_t n, void *d_hash32, void *d_dihedral,
                                   void *stream)
{
    if (!ctx || (!d_coeffs && n) || (!d_hash32 && !d_dihedral)) {
        rph_set_error("rph_pdq_hashes_from_coeffs: null argument");
        return RPH_ERR_INVALID_ARG;
    }
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    return rph_launch_pdq_from_coeffs((const float *)d_coeffs, n, (uint8_t *)d_hash32, (uint8_t *)d_dihedral, pick(ctx, stream));
}

int rph_pdq_hashes_from_coeffs(rph_ctx *ctx, const float *coeffs, uint32_t n, uint8_t *hash32_out, uint8_t *dihedral_out)
{
    return rph_guarded("rph_pdq_hashes_from_coeffs", [&]() -> int {
        if (!ctx || (!coeffs && n) || (!hash32_out && !dihedral_out)) {
            rph_set_error("rph_pdq_hashes_from_coeffs: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (n == 0) return RPH_OK;
        RPH_HIP_CHECK(hipSetDevice(ctx->device));
        DevBuf d_c, d_h, d_d;
        RPH_TRY(d_c.alloc((size_t)n * 1024));
        if (hash32_out) RPH_TRY(d_h.alloc((size_t)n * 32));
        if (dihedral_out) RPH_TRY(d_d.alloc((size_t)n * 256));
        RPH_HIP_CHECK(hipMemcpyAsync(d_c.p, coeffs, (size_t)n * 1024, hipMemcpyHostToDevice, ctx->stream));
        RPH_TRY(rph_pdq_hashes_from_coeffs_dev(ctx, d_c.p, n, d_h.p, d_d.p, ctx->stream));
        if (hash32_out) RPH_HIP_CHECK(hipMemcpyAsync(hash32_out, d_h.p, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
        if (dihedral_out) RPH_HIP_CHECK(hipMemcpyAsync(dihedral_out, d_d.p, (size_t)n * 256, hipMemcpyDeviceToHost, ctx->stream));
        RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return RPH_OK;
    });
}

// ------------------------------------------------------------------------------------------
// Hamming
// ------------------------------------------------------------------------------------------
int rph_hamming_all_pairs_dev(rph_ctx *ctx, const void *d_hashes32, uint64_t n, uint32_t threshold, uint32_t part,
                              uint32_t nparts, void *d_edges, uint64_t cap, void *d_count, void *stream)
{
    if (!ctx || (!d_hashes32 && n) || !d_count || (!d_edges && cap)) {
        rph_set_error("rph_hamming_all_pairs: null argument");
        return RPH_ERR_INVALID_ARG;
    }
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    return rph_launch_hamming_sweep(ctx, (const uint8_t *)d_hashes32, 1, (const uint8_t *)d_hashes32, nullptr, nullptr, n, threshold,
                                    part, nparts, (rph_edge *)d_edges, cap, (unsigned long long *)d_count, pick(ctx, stream),
                                    ctx->hamming_kernel);
}

int rph_hamming_variant_pairs_dev(rph_ctx *ctx, const void *d_variants, uint32_t n_variants, const void *d_hashes32,
                                  const void *d_low_conf, uint64_t n, uint32_t similarity, uint32_t part, uint32_t nparts,
                                  void *d_edges, uint64_t cap, void *d_count, void *stream)
{
    if (!ctx || ((!d_variants || !d_hashes32) && n) || !d_count || (!d_edges && cap)) {
        rph_set_error("rph_hamming_variant_pairs: null argument");
        return RPH_ERR_INVALID_ARG;
    }
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    return rph_launch_hamming_sweep(ctx, (const uint8_t *)d_variants, n_variants, (const uint8_t *)d_hashes32,
                                    (const uint8_t *)d_low_conf, nullptr, n, similarity, part, nparts, (rph_edge *)d_edges, cap,
                                    (unsigned long long *)d_count, pick(ctx, stream), ctx->hamming_kernel);
}

int rph_hamming_all_pairs(rph_ctx *ctx, const uint8_t *hashes32, uint64_t n, uint32_t threshold, uint32_t part,
                          uint32_t nparts, rph_edge *edges, uint64_t cap, uint64_t *n_edges_out)
{
    return rph_guarded("rph_hamming_all_pairs", [&]() -> int {
        return sweep_host(ctx, nullptr, 1, hashes32, nullptr, nullptr, n, threshold, part, nparts, edges, cap, n_edges_out);
    });
}

int rph_hamming_variant_pairs(rph_ctx *ctx, const uint8_t *variants, uint32_t n_variants, const uint8_t *hashes32,
                              const uint8_t *low_conf, uint64_t n, uint32_t similarity, uint32_t part, uint32_t nparts,
                              rph_edge *edges, uint64_t cap, uint64_t *n_edges_out)
{
    return rph_guarded("rph_hamming_variant_pairs", [&]() -> int {
        if (!variants && n) {
            rph_set_error("rph_hamming_variant_pairs: variants is null");
            return RPH_ERR_INVALID_ARG;
        }
        return sweep_host(ctx, variants, n_variants, hashes32, low_conf, nullptr, n, similarity, part, nparts, edges, cap,
                          n_edges_out);
    });
}

int rph_hamming_all_pairs64_dev(rph_ctx *ctx, const void *d_hashes64, uint64_t n, uint32_t threshold, uint32_t part,
                                uint32_t nparts, void *d_edges, uint64_t cap, void *d_count, void *stream)
{
    if (!ctx || (!d_hashes64 && n) || !d_count || (!d_edges && cap)) {
        rph_set_error("rph_hamming_all_pairs64: null argument");
        return RPH_ERR_INVALID_ARG;
    }
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    return rph_launch_hamming64_sweep((const uint64_t *)d_hashes64, n, threshold, part, nparts, (rph_edge *)d_edges, cap,
                                      (unsigned long long *)d_count, pick(ctx, stream), ctx->hamming_kernel);
}

int rph_hamming_all_pairs64(rph_ctx *ctx, const uint64_t *hashes64, uint64_t n, uint32_t threshold, uint32_t part,
                            uint32_t nparts, rph_edge *edges, uint64_t cap, uint64_t *n_edges_out)
{
    return rph_guarded("rph_hamming_all_pairs64", [&]() -> int {
        if (!ctx || (!hashes64 && n) || !n_edges_out || (!edges && cap)) {
            rph_set_error("rph_hamming_all_pairs64: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        *n_edges_out = 0;
        if (n < 2) return RPH_OK;
        RPH_HIP_CHECK(hipSetDevice(ctx->device));
        DevBuf d_h, d_e, d_cnt;
        RPH_TRY(d_h.alloc(n * 8));
        RPH_TRY(d_e.alloc(cap * sizeof(rph_edge)));
        RPH_TRY(d_cnt.alloc(8));
        RPH_HIP_CHECK(hipMemcpyAsync(d_h.p, hashes64, n * 8, hipMemcpyHostToDevice, ctx->stream));
        RPH_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, 8, ctx->stream));
        RPH_TRY(rph_launch_hamming64_sweep((const uint64_t *)d_h.p, n, threshold, part, nparts, (rph_edge *)d_e.p, cap,
                                           (unsigned long long *)d_cnt.p, ctx->stream, ctx->hamming_kernel));
        unsigned long long cnt = 0;
        RPH_HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *n_edges_out = cnt;
        const uint64_t take = std::min<uint64_t>(cnt, cap);
        if (take) RPH_HIP_CHECK(hipMemcpy(edges, d_e.p, take * sizeof(rph_edge), hipMemcpyDeviceToHost));
        if (cnt > cap) {
            rph_set_error("hamming64 sweep: %llu edges found, capacity %llu", cnt, (unsigned long long)cap);
            return RPH_ERR_CAPACITY;
        }
        return RPH_OK;
    });
}

int rph_find_groups64(rph_ctx *ctx, const uint64_t *hashes64, uint64_t n, uint32_t max_dist, uint32_t *members,
                      uint32_t *offsets, uint32_t *n_groups_out)
{
    return rph_guarded("rph_find_groups64", [&]() -> int {
        if (!ctx || (!hashes64 && n) || !members || !offsets || !n_groups_out) {
            rph_set_error("rph_find_groups64: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        *n_groups_out = 0;
        offsets[0] = 0;
        if (n < 2) return RPH_OK;
        std::vector<rph_edge> edges;
        RPH_TRY(sweep_growing(n, edges, [&](rph_edge *e, uint64_t cap, uint64_t *found) {
            return rph_hamming_all_pairs64(ctx, hashes64, n, max_dist, 0, 1, e, cap, found);
        }));
        return rph_host_find_groups(edges.data(), edges.size(), n, members, offsets, n_groups_out);
    });
}

int rph_find_groups_from_edges(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members, uint32_t *offsets,
                               uint32_t *n_groups_out)
{
    return rph_guarded("rph_find_groups_from_edges", [&]() -> int {
        if ((!edges && n_edges) || !members || !offsets || !n_groups_out) return RPH_ERR_INVALID_ARG;
        return rph_host_find_groups(edges, n_edges, n, members, offsets, n_groups_out);
    });
}

int rph_union_find_groups(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members, uint32_t *offsets,
                          uint32_t *n_groups_out)
{
    return rph_guarded("rph_union_find_groups", [&]() -> int {
        if ((!edges && n_edges) || !members || !offsets || !n_groups_out) return RPH_ERR_INVALID_ARG;
        return rph_host_union_find(edges, n_edges, n, members, offsets, n_groups_out);
    });
}

int rph_find_groups256(rph_ctx *ctx, const uint8_t *hashes32, uint64_t n, uint32_t max_dist, uint32_t *members,
                       uint32_t *offsets, uint32_t *n_groups_out)
{
    return rph_guarded("rph_find_groups256", [&]() -> int {
        if (!ctx || (!hashes32 && n) || !members || !offsets || !n_groups_out) {
            rph_set_error("rph_find_groups256: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        *n_groups_out = 0;
        offsets[0] = 0;
        if (n < 2) return RPH_OK;
        std::vector<rph_edge> edges;
        RPH_TRY(sweep_growing(n, edges, [&](rph_edge *e, uint64_t cap, uint64_t *found) {
            return sweep_host(ctx, nullptr, 1, hashes32, nullptr, nullptr, n, max_dist, 0, 1, e, cap, found);
        }));
        return rph_host_find_groups(edges.data(), edges.size(), n, members, offsets, n_groups_out);
    });
}

int rph_group_files_pdq(rph_ctx *ctx, const uint8_t *hashes32, const float *coeffs, const uint8_t *has_features,
                        const int32_t *quality, uint64_t n, uint32_t similarity, uint32_t *members, uint32_t *offsets,
                        uint32_t *n_groups_out, uint64_t *comparison_count_out)
{
    return rph_guarded("rph_group_files_pdq", [&]() -> int {
        if (!ctx || (!hashes32 && n) || !members || !offsets || !n_groups_out) {
            rph_set_error("rph_group_files_pdq: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (similarity > RPH_MAX_SIMILARITY_256) {
            // scanner.rs:1650-1655 asserts this
            rph_set_error("Similarity distances above %u require R=4 bit-flip checks, which are not implemented.",
                          RPH_MAX_SIMILARITY_256);
            return RPH_ERR_INVALID_ARG;
        }
        *n_groups_out = 0;
        offsets[0] = 0;
        if (comparison_count_out) *comparison_count_out = 0;
        if (n < 2) return RPH_OK;
        RPH_HIP_CHECK(hipSetDevice(ctx->device));

        // Everything between the caller's arrays and the edge list stays on the device: the hashes go up once, the 8 dihedral
        // variants of every file are produced there from its coefficients (scanner.rs:1621-1623; the coefficients cross PCIe in
        // 256 MiB pieces through a fixed staging buffer) and feed the variant sweep directly.
        std::vector<uint8_t> low_conf;
        if (quality) {
            low_conf.resize(n);
            for (uint64_t i = 0; i < n; i++) low_conf[i] = (uint8_t)rph_is_low_pdq_quality(quality[i]);
        }
        hipStream_t s = ctx->stream;
        DevBuf d_h, d_var, d_lc, d_hf, d_stage, d_e, d_cnt;
        RPH_TRY(d_h.alloc(n * 32));
        RPH_HIP_CHECK(hipMemcpyAsync(d_h.p, hashes32, n * 32, hipMemcpyHostToDevice, s));
        if (quality) {
            RPH_TRY(d_lc.alloc(n));
            RPH_HIP_CHECK(hipMemcpyAsync(d_lc.p, low_conf.data(), n, hipMemcpyHostToDevice, s));
        }
        const bool use_hf = coeffs && has_features;
        if (use_hf) {
            RPH_TRY(d_hf.alloc(n));
            RPH_HIP_CHECK(hipMemcpyAsync(d_hf.p, has_features, n, hipMemcpyHostToDevice, s));
        }
        if (coeffs) {
            RPH_TRY(d_var.alloc(n * 256));
            const uint64_t step = 1u << 18;  // 256 MiB of coefficients per piece
            RPH_TRY(d_stage.alloc(std::min<uint64_t>(step, n) * 1024));
            for (uint64_t first = 0; first < n; first += step) {
                const uint32_t m = (uint32_t)std::min<uint64_t>(step, n - first);
                RPH_HIP_CHECK(hipMemcpyAsync(d_stage.p, coeffs + first * 256, (size_t)m * 1024, hipMemcpyHostToDevice, s));
                RPH_TRY(rph_launch_pdq_from_coeffs((const float *)d_stage.p, m, nullptr, d_var.as<uint8_t>() + first * 256, s));
            }
            if (use_hf) RPH_TRY(rph_launch_featureless_variants(d_h.as<uint8_t>(), d_hf.as<uint8_t>(), n, d_var.as<uint8_t>(), s));
        }
        // Edge capacity: 32 per file to begin with (device memory is plentiful, 12 B each); a sweep that finds more is repeated
        // once into a buffer of the size it reported.
        std::vector<rph_edge> edges;
        uint64_t cap = std::max<uint64_t>(1u << 20, 32 * n);
        RPH_TRY(d_cnt.alloc(8));
        for (int attempt = 0;; attempt++) {
            RPH_TRY(d_e.alloc(cap * sizeof(rph_edge)));
            RPH_HIP_CHECK(hipMemsetAsync(d_cnt.p, 0, 8, s));
            RPH_TRY(rph_launch_hamming_sweep(ctx, coeffs ? d_var.as<uint8_t>() : d_h.as<uint8_t>(), coeffs ? 8 : 1, d_h.as<uint8_t>(), d_lc.as<uint8_t>(),
                                             use_hf ? d_hf.as<uint8_t>() : nullptr, n, similarity, 0, 1, d_e.as<rph_edge>(), cap,
                                             d_cnt.as<unsigned long long>(), s, ctx->hamming_kernel));
            unsigned long long found = 0;
            RPH_HIP_CHECK(hipMemcpyAsync(&found, d_cnt.p, 8, hipMemcpyDeviceToHost, s));
            RPH_HIP_CHECK(hipStreamSynchronize(s));
            if (found <= cap) {
                edges.resize(found);
                if (found) RPH_HIP_CHECK(hipMemcpy(edges.data(), d_e.p, found * sizeof(rph_edge), hipMemcpyDeviceToHost));
                break;
            }
            if (attempt >= 2) {
                rph_set_error("rph_group_files_pdq: edge list kept growing (%llu)", found);
                return RPH_ERR_CAPACITY;
            }
            (void)hipFree(d_e.p);
            d_e.p = nullptr;
            cap = found + found / 16 + 1024;
        }
        if (comparison_count_out) *comparison_count_out = edges.size();
        return rph_host_union_find(edges.data(), edges.size(), n, members, offsets, n_groups_out);
    });
}

int rph_mih_build256(rph_ctx *ctx, const uint8_t *hashes32, uint64_t n, uint32_t *offsets, uint32_t *values)
{
    return rph_guarded("rph_mih_build256", [&]() -> int {
        if (!ctx || (!hashes32 && n) || !offsets || (!values && n)) {
            rph_set_error("rph_mih_build256: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        RPH_HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n_off = (size_t)16 * 65536 + 1;
        DevBuf d_h, d_o, d_v;
        RPH_TRY(d_h.alloc(n * 32));
        RPH_TRY(d_o.alloc(n_off * 4));
        RPH_TRY(d_v.alloc(n * 16 * 4));
        RPH_HIP_CHECK(hipMemcpyAsync(d_h.p, hashes32, n * 32, hipMemcpyHostToDevice, ctx->stream));
        {
            std::lock_guard<std::mutex> lock(ctx->mu);
            RPH_TRY(rph_launch_mih_build256(ctx, (const uint8_t *)d_h.p, n, (uint32_t *)d_o.p, (uint32_t *)d_v.p, ctx->stream));
        }
        RPH_HIP_CHECK(hipMemcpyAsync(offsets, d_o.p, n_off * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (n) RPH_HIP_CHECK(hipMemcpyAsync(values, d_v.p, n * 16 * 4, hipMemcpyDeviceToHost, ctx->stream));
        RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return RPH_OK;
    });
}

int rph_mih_build64(rph_ctx *ctx, const uint64_t *hashes64, uint64_t n, uint32_t *offsets, uint32_t *values)
{
    return rph_guarded("rph_mih_build64", [&]() -> int {
        if (!ctx || (!hashes64 && n) || !offsets || (!values && n)) {
            rph_set_error("rph_mih_build64: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        RPH_HIP_CHECK(hipSetDevice(ctx->device));
        const size_t n_off = (size_t)8 * 256 + 1;
        DevBuf d_h, d_o, d_v;
        RPH_TRY(d_h.alloc(n * 8));
        RPH_TRY(d_o.alloc(n_off * 4));
        RPH_TRY(d_v.alloc(n * 8 * 4));
        std::lock_guard<std::mutex> lock(ctx->mu);
        RPH_HIP_CHECK(hipMemcpyAsync(d_h.p, hashes64, n * 8, hipMemcpyHostToDevice, ctx->stream));
        RPH_TRY(rph_launch_mih_build64(ctx, (const uint64_t *)d_h.p, n, (uint32_t *)d_o.p, (uint32_t *)d_v.p, ctx->stream));
        RPH_HIP_CHECK(hipMemcpyAsync(offsets, d_o.p, n_off * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (n) RPH_HIP_CHECK(hipMemcpyAsync(values, d_v.p, n * 8 * 4, hipMemcpyDeviceToHost, ctx->stream));
        RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return RPH_OK;
    });
}

// ------------------------------------------------------------------------------------------
// synthetic workloads, device memory helpers, events
// ------------------------------------------------------------------------------------------
int rph_synth_images_dev(rph_ctx *ctx, void *d_out, uint64_t first_k, uint32_t n, uint32_t w, uint32_t h, uint32_t seed,
                         void *stream)
{
    if (!ctx || (!d_out && n)) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    return rph_launch_synth_images((uint8_t *)d_out, first_k, n, w, h, seed, pick(ctx, stream));
}

int rph_synth_hashes_dev(rph_ctx *ctx, void *d_out, uint64_t first, uint64_t count, uint64_t n_total, uint64_t seed,
                         uint64_t n_clusters, void *stream)
{
    if (!ctx || (!d_out && count)) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    return rph_launch_synth_hashes((uint8_t *)d_out, first, count, n_total, seed, n_clusters, pick(ctx, stream));
}

int rph_read_stream_dev(rph_ctx *ctx, const void *d_buf, size_t bytes, void *stream)
{
    if (!ctx || !d_buf || ((uintptr_t)d_buf % 16) != 0) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->sink) RPH_HIP_CHECK(hipMalloc((void **)&ctx->sink, 16));
    return rph_launch_read_stream(d_buf, bytes, ctx->sink, pick(ctx, stream));
}

int rph_dev_alloc(rph_ctx *ctx, size_t bytes, void **out)
{
    if (!ctx || !out) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    RPH_HIP_CHECK(hipMalloc(out, bytes ? bytes : 1));
    return RPH_OK;
}
int rph_dev_free(rph_ctx *ctx, void *p)
{
    if (!ctx) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    if (p) RPH_HIP_CHECK(hipFree(p));
    return RPH_OK;
}
int rph_dev_upload(rph_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || ((!d_dst || !src) && bytes)) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    RPH_HIP_CHECK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPH_OK;
}
int rph_dev_download(rph_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || ((!dst || !d_src) && bytes)) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    RPH_HIP_CHECK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RPH_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return RPH_OK;
}
int rph_dev_memset(rph_ctx *ctx, void *d_dst, int value, size_t bytes, void *stream)
{
    if (!ctx || (!d_dst && bytes)) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    RPH_HIP_CHECK(hipMemsetAsync(d_dst, value, bytes, pick(ctx, stream)));
    return RPH_OK;
}

int rph_stream_create(rph_ctx *ctx, void **stream_out)
{
    if (!ctx || !stream_out) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    RPH_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = s;
    return RPH_OK;
}

int rph_stream_synchronize(rph_ctx *ctx, void *stream)
{
    if (!ctx) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    RPH_HIP_CHECK(hipStreamSynchronize(pick(ctx, stream)));
    return RPH_OK;
}

int rph_stream_destroy(rph_ctx *ctx, void *stream)
{
    if (!ctx || !stream) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(ctx->mu);
    RPH_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    if (ctx->scratch_stream == (hipStream_t)stream) ctx->scratch_used = false;  // its work is complete: nothing left to order behind
    if (ctx->sweep_stream == (hipStream_t)stream) ctx->sweep_used = false;
    auto ll = ctx->ll_scratch.find((hipStream_t)stream);
    if (ll != ctx->ll_scratch.end()) {
        (void)hipFree(ll->second.p);
        ctx->ll_scratch.erase(ll);
    }
    RPH_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return RPH_OK;
}

int rph_event_create(rph_ctx *ctx, void **event_out)
{
    if (!ctx || !event_out) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    hipEvent_t e;
    RPH_HIP_CHECK(hipEventCreate(&e));
    *event_out = e;
    return RPH_OK;
}
int rph_event_record(rph_ctx *ctx, void *event, void *stream)
{
    if (!ctx || !event) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipEventRecord((hipEvent_t)event, pick(ctx, stream)));
    return RPH_OK;
}
int rph_event_elapsed_ms(rph_ctx *ctx, void *start, void *stop, float *ms)
{
    if (!ctx || !start || !stop || !ms) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
    RPH_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return RPH_OK;
}
int rph_event_destroy(rph_ctx *ctx, void *event)
{
    if (!ctx || !event) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
    return RPH_OK;
}

}  // extern "C"
