// pdq_kernels.hip -- PDQ hashing kernels for gfx950 (generic multi-pass path + hashes from coefficients).
//
// Replaces /root/reference/src/pdqhash.rs:238-262 (generate_pdq_from_luma) and :59-87
// (to_hash, generate_dihedral_hashes) for batches of equally sized images.
//
// Generic path (any w, h in 5..512; 1/3/4 channels): f32 planes in HBM scratch,
//   luma -> [box rows -> box cols] x 2 -> decimate + tail (one wave per image).
// Every 1-D box line is walked by ONE thread in the reference's exact order
// (sum += in[ri]; sum -= in[li]; out = sum / cur_win), so it is bit-exact by construction.
// It is the correctness baseline and the path for odd geometries; 512x512 RGB8 goes
// through the fused single-pass kernel in pdq_fused512.hip.
#include "pdq_tail.hpp"
#include "rph_internal.h"

namespace {

// ---- to_luma601 (pdqhash.rs:268-284) + u8 -> f32 (:244) ----
__global__ void __launch_bounds__(256) luma_kernel(const uint8_t *__restrict__ px, uint32_t n, uint32_t w, uint32_t h,
                                                   uint32_t channels, size_t row_stride, size_t image_stride,
                                                   float *__restrict__ plane)
{
    const uint64_t total = (uint64_t)n * w * h;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t img = (uint32_t)(t / ((uint64_t)w * h));
        const uint32_t rem = (uint32_t)(t - (uint64_t)img * w * h);
        const uint32_t y = rem / w, x = rem - y * w;
        const uint8_t *p = px + (size_t)img * image_stride + (size_t)y * row_stride + (size_t)x * channels;
        uint32_t v;
        if (channels == 1)
            v = p[0];
        else
            v = (299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u;
        plane[t] = (float)v;
    }
}

// ---- box_one_d_float (pdqhash.rs:341-396), one line per thread ----
__device__ __forceinline__ void box_one_d(const float *__restrict__ in, float *__restrict__ out, uint32_t len,
                                          size_t stride, uint32_t win)
{
    const uint32_t maxlen = len > 1 ? len : 1;
    win = win < 1 ? 1 : (win > maxlen ? maxlen : win);
    const uint32_t half = (win + 2) / 2;
    const uint32_t phase_1 = half - 1, phase_2 = win - half + 1, phase_3 = len > win ? len - win : 0, phase_4 = half - 1;
    size_t li = 0, ri = 0, oi = 0;
    float sum = 0.0f, cur = 0.0f;
    for (uint32_t t = 0; t < phase_1; t++) {
        sum = sum + in[ri];
        cur = cur + 1.0f;
        ri += stride;
    }
    for (uint32_t t = 0; t < phase_2; t++) {
        sum = sum + in[ri];
        cur = cur + 1.0f;
        out[oi] = sum / cur;
        ri += stride;
        oi += stride;
    }
    // The bulk of the line.  The loads do not depend on the running sum, so they are issued eight steps ahead of the
    // (strictly sequential) arithmetic; the order of the additions and of the division is the reference's.
    uint32_t t = 0;
    for (; t + 8 <= phase_3; t += 8) {
        float r[8], l[8], o[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            r[k] = in[ri + k * stride];
            l[k] = in[li + k * stride];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            sum = sum + r[k];
            sum = sum - l[k];
            o[k] = sum / cur;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) out[oi + k * stride] = o[k];
        li += 8 * stride;
        ri += 8 * stride;
        oi += 8 * stride;
    }
    for (; t < phase_3; t++) {
        sum = sum + in[ri];
        sum = sum - in[li];
        out[oi] = sum / cur;
        li += stride;
        ri += stride;
        oi += stride;
    }
    for (uint32_t t = 0; t < phase_4; t++) {
        sum = sum - in[li];
        cur = cur - 1.0f;
        out[oi] = sum / cur;
        li += stride;
        oi += stride;
    }
}

// box_along_rows_float (pdqhash.rs:398-402): thread = (image, row)
__global__ void __launch_bounds__(64) box_rows_kernel(const float *__restrict__ in, float *__restrict__ out, uint32_t n,
                                                      uint32_t h, uint32_t w, uint32_t win)
{
    const uint64_t line = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= (uint64_t)n * h) return;
    box_one_d(in + line * w, out + line * w, w, 1, win);
}

// box_along_cols_float (pdqhash.rs:404-408): thread = (image, column)
__global__ void __launch_bounds__(64) box_cols_kernel(const float *__restrict__ in, float *__restrict__ out, uint32_t n,
                                                      uint32_t h, uint32_t w, uint32_t win)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * w) return;
    const uint32_t img = (uint32_t)(t / w), col = (uint32_t)(t - (uint64_t)img * w);
    const size_t base = (size_t)img * h * w + col;
    box_one_d(in + base, out + base, h, w, win);
}

// decimate_float::<64,64> (pdqhash.rs:428-443) + tail; one wave per image
__global__ void __launch_bounds__(64) tail_generic_kernel(const float *__restrict__ plane, uint32_t n, uint32_t h, uint32_t w,
                                                          uint8_t *hash, float *quality, float *coeffs, uint8_t *dihedral)
{
    __shared__ float lds[rph::TAIL_LDS_FLOATS];
    const uint32_t img = blockIdx.x;
    const int lane = threadIdx.x;
    const float *p = plane + (size_t)img * h * w;
    const uint32_t cj = ((uint32_t)(lane * 2 + 1) * w) / 128u;
    float b[64];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const uint32_t ri = ((uint32_t)(i * 2 + 1) * h) / 128u;
        b[i] = p[(size_t)ri * w + cj];
    }
    rph::pdq_tail(b, lds, lane, hash ? hash + (size_t)img * 32 : nullptr, quality ? quality + img : nullptr,
                  coeffs ? coeffs + (size_t)img * 256 : nullptr, dihedral ? dihedral + (size_t)img * 256 : nullptr);
}


// ---- the same passes with the row lines staged through LDS (the default; the kernels above stay as the plain baseline,
// rph_pdq_set_kernel(ctx, 0)).  A lane that walks its own row reads one cache line per lane and step; here a wave (64 rows) loads
// 64-column tiles with lane = column (whole lines), walks them out of LDS with lane = row, and stores its outputs with lane = column again.
// The arithmetic per line is the reference's, step for step (pdqhash.rs:341-396), written as one loop over t = index of the incoming sample:
//   t < len: sum += in[t] (and cur += 1 while the window fills: t < win);  t >= win: sum -= in[t - win] (and cur -= 1 once t >= len);
//   t >= half - 1: out[t - (half - 1)] = sum / cur.
// The second row pass only feeds decimate_float (pdqhash.rs:428-443), which keeps 64 columns: it emits those and nothing else, so the second
// column pass and the tail work on [h][64] values per image.
constexpr int BT_HIST = 8;                  // columns of the previous tile kept in front of the current one (window <= 8)
constexpr int BT_PITCH = BT_HIST + 65;      // floats per tile row (odd: lane = row accesses are conflict-free)

struct BoxGeo {
    uint32_t win, half, lead;  // window, (win + 2) / 2, half - 1
};
__device__ __forceinline__ BoxGeo box_geo(uint32_t len, uint32_t win)
{
    const uint32_t maxlen = len > 1 ? len : 1;
    win = win < 1 ? 1 : (win > maxlen ? maxlen : win);
    BoxGeo g;
    g.win = win;
    g.half = (win + 2) / 2;
    g.lead = g.half - 1;
    return g;
}

// SRC: 0 = an f32 plane [n][h][w]; 1 = the caller's Luma8 pixels; 3 = the caller's Rgb8 / Rgba8 pixels (to_luma601 on the way in).  (A template
// parameter and not a branch on `channels`: with the branch every load of the prefetch waited for the one before.)
// SAMPLED: only the 64 columns decimate_float keeps are written, to out[line][64]; else the whole line, out[line][w].
// One tile in LDS, used in place: the sample that enters at step s sits in slot s; the output of step s (column c0 + s - lead) goes to slot
// s - win, whose sample has just been subtracted for the last time.  The next tile is fetched into registers while this one is walked.
template <int SRC, bool SAMPLED>
__global__ void __launch_bounds__(64) box_rows_tiled_kernel(const void *__restrict__ src_, uint32_t n, uint32_t h, uint32_t w, uint32_t channels, size_t row_stride,
                                                            size_t image_stride, uint32_t win_, float *__restrict__ out)
{
    __shared__ float t_in[64 * BT_PITCH];
    const uint32_t lane = threadIdx.x;
    const uint64_t lines = (uint64_t)n * h, line0 = (uint64_t)blockIdx.x * 64;
    const uint32_t n_rows = (uint32_t)min((uint64_t)64, lines - line0);
    const BoxGeo g = box_geo(w, win_);
    // source offset of this lane's own line (rows past the end repeat the last line: loaded, never stored)
    const uint64_t my_line = min(line0 + lane, lines - 1);
    constexpr bool SRC_U8 = SRC != 0;
    size_t my_off;
    if (SRC_U8) {
        const uint32_t img = (uint32_t)(my_line / h), y = (uint32_t)(my_line - (uint64_t)img * h);
        my_off = (size_t)img * image_stride + (size_t)y * row_stride;
    } else {
        my_off = (size_t)my_line * w;
    }
    const uint32_t off_lo = (uint32_t)my_off, off_hi = (uint32_t)(my_off >> 32);
    float pre[64];  // the tile being fetched: row rr at column c0 + lane
    auto fetch = [&](uint32_t c0) {
        const uint32_t c = min(c0 + lane, w - 1);  // (columns past the end repeat the last one: never used by a step with t < w)
#pragma unroll
        for (int rr = 0; rr < 64; rr++) {
            const size_t off = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)off_lo, rr) | (size_t)(uint32_t)__builtin_amdgcn_readlane((int)off_hi, rr) << 32;
            if (SRC_U8) {
                const uint8_t *p = static_cast<const uint8_t *>(src_) + off + (size_t)c * channels;
                pre[rr] = SRC == 1 ? (float)p[0] : (float)((299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u);
            } else {
                pre[rr] = static_cast<const float *>(src_)[off + c];
            }
        }
    };
    float sum = 0.0f, cur = 0.0f;
    uint32_t next_sample = 0;  // SAMPLED: column of sample j = ((2 j + 1) w) / 128
    const uint32_t total = w + g.lead;
    float *mine = t_in + lane * BT_PITCH + BT_HIST;
    // Outputs leave one tile late: they are read out of LDS into registers after the walk and stored while the next tile is walked, so that
    // nothing waits for a store (loads and stores share one counter: a wait for the prefetched tile is a wait for every store before it).
    float ost[64];
    float *ost_to = out;
    bool ost_ok = false;
    auto flush = [&]() {
        if (ost_ok) {
#pragma unroll
            for (int rr = 0; rr < 64; rr++)
                if ((uint32_t)rr < n_rows) ost_to[(size_t)rr * (SAMPLED ? 64 : w)] = ost[rr];
        }
    };
    fetch(0);
    for (uint32_t c0 = 0; c0 < total; c0 += 64) {
        if (c0 < w) {
#pragma unroll
            for (int rr = 0; rr < 64; rr++) t_in[rr * BT_PITCH + BT_HIST + lane] = pre[rr];
        }
        __syncthreads();
        flush();
        if (c0 + 64 < w) fetch(c0 + 64);
        // ---- walk: lane = row
        const uint32_t steps = min(64u, total - c0);
        if (c0 >= g.win && c0 + 64 <= w) {  // every step adds, subtracts and emits, over a full window
            const int d = (int)g.win;
            for (int s = 0; s < 64; s += 8) {
                float r[8], l[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    r[k] = mine[s + k];
                    l[k] = mine[s + k - d];
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    sum = sum + r[k];
                    sum = sum - l[k];
                    l[k] = sum / cur;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) mine[s + k - d] = l[k];
            }
        } else {
            for (uint32_t s = 0; s < steps; s++) {
                const uint32_t t = c0 + s;
                if (t < w) {
                    sum = sum + mine[s];
                    if (t < g.win) cur = cur + 1.0f;
                }
                if (t >= g.win) {
                    sum = sum - mine[(int)s - (int)g.win];
                    if (t >= w) cur = cur - 1.0f;
                }
                if (t >= g.lead) mine[(int)s - (int)g.win] = sum / cur;
            }
        }
        __syncthreads();
        // ---- outputs into registers, lane = column again
        if (!SAMPLED) {
            const int64_t oc = (int64_t)c0 + lane - g.lead;  // the output of step s = lane
            ost_ok = oc >= 0 && oc < (int64_t)w && lane < steps;
            ost_to = out + (size_t)line0 * w + (ost_ok ? (size_t)oc : 0);
            const float *from = t_in + BT_HIST + (int)lane - (int)g.win;
#pragma unroll
            for (int rr = 0; rr < 64; rr++) ost[rr] = from[rr * BT_PITCH];
        } else {
            // samples whose step lies in this tile: j0 .. j0 + nj - 1, lane = sample
            const uint32_t j0 = next_sample;
            while (next_sample < 64 && ((2u * next_sample + 1u) * w) / 128u + g.lead < c0 + steps) next_sample++;
            const uint32_t nj = next_sample - j0;
            ost_ok = lane < nj;
            if (nj) {
                const uint32_t j = j0 + min(lane, nj - 1);
                const int slot = (int)(((2u * j + 1u) * w) / 128u + g.lead - c0) - (int)g.win;
                const float *from = t_in + BT_HIST + slot;
                ost_to = out + (size_t)line0 * 64 + j;
#pragma unroll
                for (int rr = 0; rr < 64; rr++) ost[rr] = from[rr * BT_PITCH];
            }
        }
        __syncthreads();
        // the last BT_HIST slots (the samples still to be subtracted among them) move in front of the next tile
        float keep[BT_HIST];
#pragma unroll
        for (int k = 0; k < BT_HIST; k++) keep[k] = mine[64 - BT_HIST + k];
#pragma unroll
        for (int k = 0; k < BT_HIST; k++) mine[k - BT_HIST] = keep[k];
        __syncthreads();
    }
    flush();
}

// second column pass on the 64 kept columns + decimate_float's rows + tail; one wave per image, lane = kept column.
// The column is walked 64 rows at a time out of LDS (the loads of a chunk are independent of the running sum and go out together).
__global__ void __launch_bounds__(64) tail_sampled_kernel(const float *__restrict__ kept, uint32_t n, uint32_t h, uint32_t win_, uint8_t *hash, float *quality,
                                                          float *coeffs, uint8_t *dihedral)
{
    __shared__ float lds[rph::TAIL_LDS_FLOATS];
    __shared__ float t_k[(BT_HIST + 64) * 64];  // rows r0 - 8 .. r0 + 63 of this image's kept columns
    __shared__ float t_b[64 * 64];              // the 64 kept rows
    const uint32_t img = blockIdx.x;
    const int lane = threadIdx.x;
    const float *p = kept + (size_t)img * h * 64 + lane;
    const BoxGeo g = box_geo(h, win_);
    float sum = 0.0f, cur = 0.0f;
    uint32_t next_sample = 0, next_row = h / 128u;  // row of sample i = ((2 i + 1) h) / 128
    const uint32_t total = h + g.lead;
    for (uint32_t r0 = 0; r0 < total; r0 += 64) {
        if (r0 < h) {
            const uint32_t rows = min(64u, h - r0);
            for (uint32_t rr = 0; rr < rows; rr++) t_k[(BT_HIST + rr) * 64 + lane] = p[(size_t)(r0 + rr) * 64];
        }
        const uint32_t steps = min(64u, total - r0);
        const float *mine = t_k + BT_HIST * 64 + lane;
#pragma unroll 4
        for (uint32_t s = 0; s < steps; s++) {
            const uint32_t t = r0 + s;
            if (t < h) {
                sum = sum + mine[s * 64];
                if (t < g.win) cur = cur + 1.0f;
            }
            if (t >= g.win) {
                sum = sum - mine[((int)s - (int)g.win) * 64];
                if (t >= h) cur = cur - 1.0f;
            }
            if (t >= g.lead) {
                const float o = sum / cur;
                while (next_sample < 64 && next_row == t - g.lead) {  // (heights below 64 keep a row more than once)
                    t_b[next_sample * 64 + lane] = o;
                    next_sample++;
                    next_row = ((2u * next_sample + 1u) * h) / 128u;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < BT_HIST; k++) t_k[k * 64 + lane] = t_k[(64 + k) * 64 + lane];
    }
    float b[64];
#pragma unroll
    for (int i = 0; i < 64; i++) b[i] = t_b[i * 64 + lane];
    rph::pdq_tail(b, lds, lane, hash ? hash + (size_t)img * 32 : nullptr, quality ? quality + img : nullptr,
                  coeffs ? coeffs + (size_t)img * 256 : nullptr, dihedral ? dihedral + (size_t)img * 256 : nullptr);
}

// PdqFeatures::to_hash / generate_dihedral_hashes for stored coefficients; one wave per image
__global__ void __launch_bounds__(64) from_coeffs_kernel(const float *__restrict__ coeffs, uint32_t n, uint8_t *hash,
                                                         uint8_t *dihedral)
{
    __shared__ float lds_c[256];
    const uint32_t img = blockIdx.x;
    const int lane = threadIdx.x;
    float c[4];
#pragma unroll
    for (int m = 0; m < 4; m++) c[m] = coeffs[(size_t)img * 256 + lane + 64 * m];
    rph::hashes_from_coeffs(c, lds_c, lane, hash ? hash + (size_t)img * 32 : nullptr,
                            dihedral ? dihedral + (size_t)img * 256 : nullptr);
}

// files without features own their hash as the only variant (scanner.rs:1624-1627): every slot of such a file becomes its hash
__global__ void __launch_bounds__(256) featureless_variants_kernel(const uint4 *__restrict__ hashes, const uint8_t *__restrict__ has_features, uint64_t n,
                                                                   uint4 *variants)
{
    // thread = (file, slot, half of the 32-byte hash)
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n * 16; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = t >> 4;
        if (!has_features[i]) variants[t] = hashes[2 * i + (t & 1)];
    }
}

// stored quality of scanner.rs:1416-1417, (q * 100).round().clamp(0, 100) as u16 (round half away from zero = roundf), against
// PDQ_MIN_QUALITY (scanner.rs:1588-1594): 1 = low confidence.  Images that were not hashable (valid == 0) have no quality: not low.
__global__ void __launch_bounds__(256) lowconf_kernel(const float *__restrict__ quality, const uint8_t *__restrict__ valid, uint64_t n, uint8_t *low)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float stored = roundf(quality[i] * 100.0f);
        stored = stored < 0.0f ? 0.0f : (stored > 100.0f ? 100.0f : stored);
        low[i] = (uint8_t)((!valid || valid[i]) && stored < (float)RPH_PDQ_MIN_QUALITY);
    }
}

}  // namespace

int rph_launch_lowconf_from_quality(const float *d_quality, const uint8_t *d_valid, uint64_t n, uint8_t *d_low, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    const uint64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(lowconf_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, stream, d_quality, d_valid, n, d_low);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

int rph_launch_featureless_variants(const uint8_t *d_hashes, const uint8_t *d_has_features, uint64_t n, uint8_t *d_variants, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    const uint64_t want = (n * 16 + 255) / 256;
    hipLaunchKernelGGL(featureless_variants_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, stream, (const uint4 *)d_hashes, d_has_features, n,
                       (uint4 *)d_variants);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

int rph_launch_pdq_from_coeffs(const float *d_coeffs, uint32_t n, uint8_t *d_hash, uint8_t *d_dihedral, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    constexpr uint32_t CHUNK = 1u << 24;  // gridDim.x * 64 threads must stay below 2^32
    for (uint32_t first = 0; first < n; first += CHUNK) {
        const uint32_t m = (n - first) < CHUNK ? (n - first) : CHUNK;
        hipLaunchKernelGGL(from_coeffs_kernel, dim3(m), dim3(64), 0, stream, d_coeffs + (size_t)first * 256, m,
                           d_hash ? d_hash + (size_t)first * 32 : nullptr, d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr);
        RPH_HIP_CHECK(hipGetLastError());
        if (n - first <= CHUNK) break;
    }
    return RPH_OK;
}

int rph_launch_pdq_generic(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                           size_t row_stride, size_t image_stride, uint8_t *d_hash, float *d_quality, float *d_coeffs,
                           uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    if (d_valid) RPH_HIP_CHECK(hipMemsetAsync(d_valid, 1, n, stream));
    const size_t plane_bytes = (size_t)(w > 64 ? w : 64) * h * sizeof(float);  // (a plane also holds the 64 kept columns of every row)
    // two planes per image in flight; the scratch is capped at 2 GiB
    uint32_t chunk = (uint32_t)(((size_t)1 << 30) / plane_bytes);  // up to 1 GiB per plane: short launches leave the chip half empty (one wave per image in the tail)
    if (chunk < 1) chunk = 1;
    if (chunk > n) chunk = n;
    const size_t need = 2 * plane_bytes * chunk;
    if (ctx->scratch_bytes < need) {
        // kernels of any stream may still be using the old scratch
        RPH_HIP_CHECK(hipDeviceSynchronize());
        if (ctx->scratch) RPH_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        RPH_HIP_CHECK(hipMalloc((void **)&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    // one scratch for all caller streams (the caller holds ctx->mu while enqueueing): order this launch behind the previous
    // user's kernels if they went to another stream
    if (!ctx->scratch_done) RPH_HIP_CHECK(hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming));
    if (ctx->scratch_used && ctx->scratch_stream != stream) RPH_HIP_CHECK(hipStreamWaitEvent(stream, ctx->scratch_done, 0));
    float *a = ctx->scratch, *b = ctx->scratch + (size_t)chunk * (plane_bytes / sizeof(float));
    const uint32_t win_rows = (w + 63) / 64;  // window along rows = ceil(cols / 64)   pdqhash.rs:246
    const uint32_t win_cols = (h + 63) / 64;  // window along cols = ceil(rows / 64)   pdqhash.rs:247
    for (uint32_t first = 0; first < n; first += chunk) {
        const uint32_t m = (n - first) < chunk ? (n - first) : chunk;
        if (ctx->pdq_kernel != 0 && win_rows <= (uint32_t)BT_HIST && win_cols <= (uint32_t)BT_HIST) {  // rows through LDS tiles, the second half of the filter on the 64 kept columns only
            const unsigned row_blocks = (unsigned)(((uint64_t)m * h + 63) / 64);
            if (channels == 1)
                hipLaunchKernelGGL((box_rows_tiled_kernel<1, false>), dim3(row_blocks), dim3(64), 0, stream, (const void *)(d_px + (size_t)first * image_stride), m, h, w,
                                   channels, row_stride, image_stride, win_rows, a);
            else
                hipLaunchKernelGGL((box_rows_tiled_kernel<3, false>), dim3(row_blocks), dim3(64), 0, stream, (const void *)(d_px + (size_t)first * image_stride), m, h, w,
                                   channels, row_stride, image_stride, win_rows, a);
            hipLaunchKernelGGL(box_cols_kernel, dim3((unsigned)(((uint64_t)m * w + 63) / 64)), dim3(64), 0, stream, a, b, m, h, w, win_cols);
            float *kept = a;  // [m][h][64]; plane a has been consumed by the column pass
            hipLaunchKernelGGL((box_rows_tiled_kernel<0, true>), dim3(row_blocks), dim3(64), 0, stream, (const void *)b, m, h, w, 1u, (size_t)0, (size_t)0, win_rows, kept);
            hipLaunchKernelGGL(tail_sampled_kernel, dim3(m), dim3(64), 0, stream, kept, m, h, win_cols, d_hash ? d_hash + (size_t)first * 32 : nullptr,
                               d_quality ? d_quality + first : nullptr, d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr,
                               d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr);
            RPH_HIP_CHECK(hipGetLastError());
            continue;
        }
        const uint64_t total = (uint64_t)m * w * h;
        const uint64_t want = (total + 255) / 256;
        hipLaunchKernelGGL(luma_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, stream,
                           d_px + (size_t)first * image_stride, m, w, h, channels, row_stride, image_stride, a);
        for (int rep = 0; rep < 2; rep++) {  // PDQ_NUM_JAROSZ_XY_PASSES = 2 (pdqhash.rs:18, :422-425)
            hipLaunchKernelGGL(box_rows_kernel, dim3((unsigned)(((uint64_t)m * h + 63) / 64)), dim3(64), 0, stream, a, b, m, h, w,
                               win_rows);
            hipLaunchKernelGGL(box_cols_kernel, dim3((unsigned)(((uint64_t)m * w + 63) / 64)), dim3(64), 0, stream, b, a, m, h, w,
                               win_cols);
        }
        hipLaunchKernelGGL(tail_generic_kernel, dim3(m), dim3(64), 0, stream, a, m, h, w,
                           d_hash ? d_hash + (size_t)first * 32 : nullptr, d_quality ? d_quality + first : nullptr,
                           d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr,
                           d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr);
        RPH_HIP_CHECK(hipGetLastError());
    }
    RPH_HIP_CHECK(hipEventRecord(ctx->scratch_done, stream));
    ctx->scratch_stream = stream;
    ctx->scratch_used = true;
    return RPH_OK;
}
