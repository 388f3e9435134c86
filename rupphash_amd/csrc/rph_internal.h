// rph_internal.h -- shared by the translation units of librupphash_hip.so (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <condition_variable>
#include <exception>
#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/rupphash.h"

struct rph_ctx {
    int device = 0;
    int compute_units = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // scratch for the generic (multi-pass) PDQ kernel: two f32 planes per in-flight image
    float *scratch = nullptr;
    size_t scratch_bytes = 0;
    uint32_t *sink = nullptr;  // 4-byte result slot of the read-stream probe
    // the scratch planes are shared by every caller stream: a launch on another stream first waits for the last user's event
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_used = false;
    // sample scratch of the low-latency PDQ kernel, one per caller stream (133 KB per image of a launch chunk): launches on different
    // streams -- the one-image queue's pipeline slots -- share nothing and need no ordering between them
    struct LLScratch {
        float *p = nullptr;
        size_t bytes = 0;
    };
    std::map<hipStream_t, LLScratch> ll_scratch;
    // rph_pdq_hash_batch (host pointers): two pinned staging sets + device twins, alternated over two streams so that the host
    // copy / H2D of one chunk overlaps the transfer and kernels of the other (rph_api.cpp); one host-batch call at a time
    std::mutex pipe_mu;
    void *pipe = nullptr;
    // pre-downsample (> 512 px inputs): u8 planes (full-resolution luma, horizontal pass, thumbnail) and the per-geometry
    // coefficient tables, kept across calls; ordered between streams together with `scratch` (the generic kernel follows on the
    // same stream and records scratch_done)
    uint8_t *rz_scratch = nullptr;
    size_t rz_bytes = 0;
    void *axis_cache = nullptr;  // resize_kernels.hip
    // scratch of the popcount-sorted sweep (sorted hashes, permutation, popcounts, radix-sort workspace); same stream ordering rule
    void *sweep_scratch = nullptr;
    size_t sweep_scratch_bytes = 0;
    hipEvent_t sweep_done = nullptr;
    hipStream_t sweep_stream = nullptr;
    bool sweep_used = false;
    // 2 = fp4 MFMA formulation of the sweep's fast path, popcount-sorted {0,1} operands for plain all-pairs sweeps (default),
    // 3 = fp4 MFMA with +-1 operands everywhere, 4 = sorted {0,1} at every size, 1 = int8 MFMA, 0 = VALU xor + popcount
    int hamming_kernel = 2;
    // 512x512 RGB8: 0 = always generic; 1 / 2 = fused one-wave-per-image kernel (64- / 128-px strips); 3 = fused low-latency kernel (eight
    // waves per image); 4 = automatic: low-latency below 768 images per call, one-wave-per-image (64-px strips) from there
    int pdq_kernel = 4;
    // JPEG path (jpeg_pipeline.cpp): two chunk slots (pinned staging, device buffers, stream), kept across calls; one batch call at a time
    std::mutex jpeg_mu;
    void *jpeg = nullptr;
    // where the Huffman streams of sequential files are decoded: 0 = host threads, 1 = device (one image per lane), 2 = automatic
    // (device from 2048 sequential files per call: the walk of one image is serial, so it needs tens of thousands of images in flight)
    int jpeg_entropy = 2;
    int jpeg_progressive_on_device = 1;  // 0: progressive files stay with the host threads (rph_jpeg_set_entropy(ctx, 3))
    // device walk of streams without restart markers: from jpeg_seg_min_bytes of entropy data a stream is cut into segments of
    // jpeg_seg_bytes that synchronise on the device and are walked side by side (0 = never)
    uint32_t jpeg_seg_min_bytes = 8192, jpeg_seg_bytes = 1024;
    // rph_jpeg_pdq_hash_one: callers that arrive while a batch is on its way wait here and leave together as the next batch
    std::mutex jpeg_qmu;
    std::condition_variable jpeg_qcv;
    std::vector<void *> jpeg_waiting;
    bool jpeg_leader = false;
    std::vector<void *> jpeg_thread_buffers;  // the callers' pinned coefficient buffers (one per calling thread), freed with the context
    uint64_t serial = 0;                      // distinguishes this context from an earlier one at the same address
};

void rph_set_error(const char *fmt, ...);

#define RPH_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            rph_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? RPH_ERR_OOM : RPH_ERR_HIP;                    \
        }                                                                                    \
    } while (0)

// ---- launchers implemented in the .hip files (all asynchronous on `stream`) ----
// pdq_kernels.hip
int rph_launch_pdq_generic(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                           size_t row_stride, size_t image_stride, uint8_t *d_hash, float *d_quality, float *d_coeffs,
                           uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream);
int rph_launch_pdq_fused512(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, size_t row_stride, size_t image_stride,
                            uint8_t *d_hash, float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid,
                            hipStream_t stream, uint32_t channels = 3);  // channels: 3 (Rgb8) or 1 (Luma8)
// pdq_stream.hip: one streaming kernel for Luma8 images of 128..512 x 128..512 with dword-aligned rows; automatic mode takes it from
// RPH_STREAM_MIN_IMAGES images per call (below that a wave per image leaves the chip empty and the image waits ~0.4 ms for its one wave)
constexpr uint32_t RPH_STREAM_MIN_IMAGES = 384;
bool rph_pdq_stream_supported(const uint8_t *d_px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride);
int rph_launch_pdq_stream(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride, uint8_t *d_hash,
                          float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream);
bool rph_pdq_stream_color_supported(const uint8_t *d_px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride);
int rph_launch_pdq_stream_color(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride,
                                uint8_t *d_hash, float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream);
int rph_launch_pdq_resized(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                           size_t row_stride, size_t image_stride, uint8_t *d_hash, float *d_quality, float *d_coeffs,
                           uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream);
int rph_launch_pdq_from_coeffs(const float *d_coeffs, uint32_t n, uint8_t *d_hash, uint8_t *d_dihedral, hipStream_t stream);
int rph_launch_lowconf_from_quality(const float *d_quality, const uint8_t *d_valid, uint64_t n, uint8_t *d_low, hipStream_t stream);
int rph_launch_featureless_variants(const uint8_t *d_hashes, const uint8_t *d_has_features, uint64_t n, uint8_t *d_variants, hipStream_t stream);
// hamming_kernels.hip
int rph_launch_hamming_sweep(rph_ctx *ctx, const uint8_t *d_rows, uint32_t n_variants, const uint8_t *d_cols, const uint8_t *d_low_conf,
                             const uint8_t *d_has_features, uint64_t n, uint32_t threshold, uint32_t part, uint32_t nparts, rph_edge *d_edges,
                             uint64_t cap, unsigned long long *d_count, hipStream_t stream, int use_mfma);
int rph_launch_hamming64_sweep(const uint64_t *d_hashes, uint64_t n, uint32_t threshold, uint32_t part, uint32_t nparts,
                               rph_edge *d_edges, uint64_t cap, unsigned long long *d_count, hipStream_t stream, int use_mfma);
int rph_launch_mih_build256(rph_ctx *ctx, const uint8_t *d_hashes, uint64_t n, uint32_t *d_offsets, uint32_t *d_values,
                            hipStream_t stream);
int rph_launch_mih_build64(rph_ctx *ctx, const uint64_t *d_hashes, uint64_t n, uint32_t *d_offsets, uint32_t *d_values,
                           hipStream_t stream);
// synth_kernels.hip
int rph_launch_synth_images(uint8_t *d_out, uint64_t first_k, uint32_t n, uint32_t w, uint32_t h, uint32_t seed,
                            hipStream_t stream);
int rph_launch_synth_hashes(uint8_t *d_out, uint64_t first, uint64_t count, uint64_t n_total, uint64_t seed,
                            uint64_t n_clusters, hipStream_t stream);

int rph_launch_read_stream(const void *d_buf, size_t bytes, uint32_t *d_sink, hipStream_t stream);

// Runs an entry point's body; no C++ exception crosses the C ABI ("nothing aborts", include/rupphash.h)
template <class F>
static inline int rph_guarded(const char *where, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        rph_set_error("%s: out of host memory", where);
        return RPH_ERR_OOM;
    } catch (const std::exception &e) {
        rph_set_error("%s: %s", where, e.what());
        return RPH_ERR_HIP;
    } catch (...) {
        rph_set_error("%s: unknown exception", where);
        return RPH_ERR_HIP;
    }
}

// rph_api.cpp
void rph_pipe_forget(rph_ctx *ctx);
int rph_pdq_hash_batch_keep(rph_ctx *ctx, const uint8_t *px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride,
                            size_t image_stride, uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *dihedral_out,
                            uint8_t *valid_out, void *d_hash_keep, void *d_quality_keep, void *d_dihedral_keep);
// batcher.cpp
void rph_batcher_forget(rph_ctx *ctx);
// resize_kernels.hip
void rph_resize_forget(rph_ctx *ctx);
// jpeg_pipeline.cpp
void rph_jpeg_forget(rph_ctx *ctx);
void rph_jpeg_forget_threads(rph_ctx *ctx);

// host_grouping.cpp
int rph_host_union_find(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members, uint32_t *offsets,
                        uint32_t *n_groups_out);
int rph_host_find_groups(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members, uint32_t *offsets,
                         uint32_t *n_groups_out);
