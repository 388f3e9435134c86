// resize_kernels.hip -- the reference's pre-downsample for inputs with a side > 512 px
// (/root/reference/src/pdqhash.rs:181-220: to_luma601 at full resolution, then resize_luma_fast to the
// aspect-preserving thumbnail of calculate_target_dimensions, then generate_pdq_from_luma).
//
// resize_luma_fast is the third-party crate fast_image_resize 6.1.0 (Cargo.lock:1853), Convolution(Box) on U8; its
// source is not part of the reference tree, so this is a statement of its published algorithm (Pillow-style
// two-pass fixed-point convolution: f64 window weights -> i16 coefficients at an adaptive precision ->
// clip8((round + sum) >> precision), horizontal pass into a u8 intermediate, then vertical).  PARITY UNPINNED.
#include <algorithm>
#include <cmath>
#include <map>
#include <utility>
#include <vector>

#include "rph_internal.h"

namespace {

struct Axis {
    std::vector<uint32_t> start, size;
    std::vector<int16_t> coef;  // [out][window]
    std::vector<int32_t> c1;    // the one coefficient of every tap of an output (a box: the taps of a window weigh the same)
    bool uniform = true;        // ... which holds for every output
    int window = 0, precision = 0;
};

double box_filter(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }

// precompute_coefficients + Normalizer16 of the crate (one axis)
Axis build_axis(uint32_t in_size, uint32_t out_size)
{
    Axis a;
    const double scale = (double)in_size / (double)out_size;
    const double filter_scale = scale > 1.0 ? scale : 1.0;
    const double radius = 0.5 * filter_scale;  // Box support 0.5
    a.window = (int)std::ceil(radius) * 2 + 1;
    const double recip = 1.0 / filter_scale;
    std::vector<double> w((size_t)out_size * a.window, 0.0);
    a.start.assign(out_size, 0);
    a.size.assign(out_size, 0);
    double max_w = 0.0;
    for (uint32_t o = 0; o < out_size; o++) {
        const double in_center = ((double)o + 0.5) * scale;
        const uint32_t x_min = (uint32_t)std::max(0.0, std::floor(in_center - radius));
        const uint32_t x_max = (uint32_t)std::min((double)in_size, std::ceil(in_center + radius));
        const double center = in_center - 0.5;
        uint32_t bound_start = x_min, bound_end = x_max;
        double *cw = &w[(size_t)o * a.window];
        int cnt = 0;
        double ww = 0.0;
        for (uint32_t x = x_min; x < x_max; x++) {
            const double v = box_filter(((double)x - center) * recip);
            if (x == bound_start && v == 0.0 && cnt == 0) {
                bound_start++;  // zero leading coefficients are not used
            } else {
                cw[cnt++] = v;
                ww += v;
            }
        }
        for (int i = cnt - 1; i >= 0; i--) {  // nor zero trailing ones
            if (bound_end <= bound_start || cw[i] != 0.0) break;
            bound_end--;
        }
        if (ww != 0.0)
            for (int i = 0; i < cnt; i++) cw[i] /= ww;
        for (int i = 0; i < cnt; i++) max_w = std::max(max_w, cw[i]);
        a.start[o] = bound_start;
        a.size[o] = bound_end - bound_start;
    }
    for (int cur = 0; cur < 16; cur++) {
        a.precision = cur;
        if ((int32_t)std::llround(max_w * (double)(1 << (cur + 1))) >= (1 << 15)) break;
    }
    a.coef.resize(w.size());
    for (size_t i = 0; i < w.size(); i++) a.coef[i] = (int16_t)std::llround(w[i] * (double)(1 << a.precision));
    a.c1.assign(out_size, 0);
    for (uint32_t o = 0; o < out_size; o++) {
        const int16_t *k = &a.coef[(size_t)o * a.window];
        a.c1[o] = a.size[o] ? k[0] : 0;
        for (uint32_t i = 1; i < a.size[o]; i++) a.uniform = a.uniform && k[i] == k[0];
    }
    return a;
}

struct DevAxis {
    const uint32_t *start, *size;
    const int16_t *coef;
    const int32_t *c1;
    int window, precision;
};

// to_luma601 (pdqhash.rs:268-284) at full resolution, u8 out
__global__ void __launch_bounds__(256) luma_u8_kernel(const uint8_t *__restrict__ px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                                                      size_t row_stride, size_t image_stride, uint8_t *__restrict__ out)
{
    const uint64_t total = (uint64_t)n * w * h;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t img = (uint32_t)(t / ((uint64_t)w * h));
        const uint32_t rem = (uint32_t)(t - (uint64_t)img * w * h);
        const uint32_t y = rem / w, x = rem - y * w;
        const uint8_t *p = px + (size_t)img * image_stride + (size_t)y * row_stride + (size_t)x * channels;
        out[t] = channels == 1 ? p[0] : (uint8_t)((299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u);
    }
}

__device__ __forceinline__ uint8_t clip8(int32_t v, int precision)
{
    v >>= precision;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [n][h][w] -> dst [n][h][nw]
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t n, uint32_t w,
                                                       uint32_t h, uint32_t nw, DevAxis ax)
{
    const uint64_t total = (uint64_t)n * h * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = (uint32_t)(t % nw);
        const uint64_t line = t / nw;  // img * h + y
        const uint8_t *row = src + line * w + ax.start[o];
        const int16_t *k = ax.coef + (size_t)o * ax.window;
        int32_t ss = 1 << (ax.precision - 1);
        for (uint32_t i = 0; i < ax.size[o]; i++) ss += (int32_t)row[i] * (int32_t)k[i];
        dst[t] = clip8(ss, ax.precision);
    }
}

// vertical pass: src [n][h][nw] -> dst [n][nh][nw]
__global__ void __launch_bounds__(256) resize_v_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t n, uint32_t h,
                                                       uint32_t nw, uint32_t nh, DevAxis ay)
{
    const uint64_t total = (uint64_t)n * nh * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(t % nw);
        const uint32_t o = (uint32_t)((t / nw) % nh);
        const uint32_t img = (uint32_t)(t / ((uint64_t)nw * nh));
        const uint8_t *col = src + ((size_t)img * h + ay.start[o]) * nw + x;
        const int16_t *k = ay.coef + (size_t)o * ay.window;
        int32_t ss = 1 << (ay.precision - 1);
        for (uint32_t i = 0; i < ay.size[o]; i++) ss += (int32_t)col[(size_t)i * nw] * (int32_t)k[i];
        dst[t] = clip8(ss, ay.precision);
    }
}

unsigned grid_for(uint64_t total) { return (unsigned)std::min<uint64_t>((total + 255) / 256, 65536); }

// ---- both passes in one kernel.  A workgroup owns one image x 64 output columns x `th` output rows: every input row those output rows
// need is convolved horizontally into a u8 tile in LDS (the u8 rounding of the two-pass form is kept: it is part of the algorithm), then the
// tile is convolved vertically.  Luma is formed on the way in (to_luma601, pdqhash.rs:268-284), so an Rgb8 source is read once and neither the
// full-size luma plane nor the half-resized plane exists in HBM.  No 64-bit division per pixel: the grid is (column tile, row tile, image).
template <int CH>
__device__ __forceinline__ int32_t luma_at(const uint8_t *p)
{
    if (CH == 1) return p[0];
    return (int32_t)((299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u);
}

constexpr int RZ_TW = 64;      // output columns per workgroup

// luma of pixel i of a run of pixels held as dwords (static byte positions)
template <int CH>
__device__ __forceinline__ int32_t luma_of(const uint32_t *d, int i)
{
    const int B = i * CH;
    const uint32_t r = (d[B >> 2] >> (8 * (B & 3))) & 0xFFu;
    if (CH == 1) return (int32_t)r;
    const uint32_t g = (d[(B + 1) >> 2] >> (8 * ((B + 1) & 3))) & 0xFFu, b = (d[(B + 2) >> 2] >> (8 * ((B + 2) & 3))) & 0xFFu;
    return (int32_t)((299u * r + 587u * g + 114u * b + 500u) / 1000u);
}

// luma of pixel i of a row staged in LDS (byte reads)
template <int CH>
__device__ __forceinline__ int32_t luma_lds(const uint8_t *p, int i)
{
    if (CH == 1) return p[i];
    return (int32_t)((299u * p[i * CH] + 587u * p[i * CH + 1] + 114u * p[i * CH + 2] + 500u) / 1000u);
}

// WMAX: the horizontal window if it is 3, 5 or 7 (sources up to 6 x the thumbnail: every tap and its weight unrolled), 0: any window.
// Horizontal pass: a wave takes RB input rows at a time; the part of each row this column tile needs is fetched as whole aligned dwords, lane =
// dword (adjacent lanes, adjacent addresses: an output's own taps start every ~2.5 bytes, which the memory pipe takes almost lane by lane),
// into the wave's own LDS row buffers, and the taps are read from there.  The wave index is made scalar, so row numbers, row pointers and the
// vertical tables are scalar too.  NIT: dwords per lane and staged row (the host picks the instantiation that covers raw_pitch).
template <int CH, int WMAX, int NIT>
__global__ void __launch_bounds__(256) resize_fused_kernel(const uint8_t *__restrict__ px, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride,
                                                           uint32_t nw, uint32_t nh, DevAxis ax, DevAxis ay, uint32_t th, uint32_t tile_rows, uint32_t raw_pitch,
                                                           uint8_t *__restrict__ dst, uint32_t dst_pitch)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t rz_tile[];  // [tile_rows][RZ_TW], then [4 waves][RB][raw_pitch] row buffers
    constexpr int RB = 4;  // rows in flight per wave
    const uint32_t lane = threadIdx.x & 63, sub = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t img = blockIdx.z, r0 = blockIdx.y * th, r1 = min(r0 + th, nh), c0 = blockIdx.x * RZ_TW, o = c0 + lane;
    const uint32_t in0 = ay.start[r0];
    uint32_t in1 = ay.start[r1 - 1] + ay.size[r1 - 1];
    in1 = min(min(in1, h), in0 + tile_rows);  // (the host sized tile_rows as the maximum over the row tiles: the clamp never bites)
    uint8_t *rowbuf = rz_tile + ((tile_rows * RZ_TW + 15u) & ~15u) + sub * RB * raw_pitch;
    {
        const uint32_t c_last = min(c0 + RZ_TW, nw) - 1;
        const uint32_t x_lo = ax.start[c0], x_hi = min(ax.start[c_last] + ax.size[c_last], w);  // source columns of this tile
        const uint32_t oc = min(o, nw - 1);
        const uint32_t xs = ax.start[oc], xn = ax.size[oc];
        const int16_t *k = ax.coef + (size_t)oc * ax.window;
        int32_t kk[WMAX ? WMAX : 1];
#pragma unroll
        for (int i = 0; i < WMAX; i++) kk[i] = (uint32_t)i < xn ? (int32_t)k[i] : 0;  // (absent taps carry weight 0; their bytes are whatever follows in the buffer)
        const int32_t half = 1 << (ax.precision - 1);
        const uint8_t *img_base = px + (size_t)img * image_stride + (size_t)x_lo * CH;
        const uint32_t span_bytes = (x_hi - x_lo) * CH;
        for (uint32_t y = in0 + sub * RB; y < in1; y += 4 * RB) {
            uint32_t shift[RB], v[RB][NIT];
#pragma unroll
            for (int q = 0; q < RB; q++) {  // RB x NIT independent loads, then the LDS writes
                const uint8_t *row = img_base + (size_t)min(y + q, in1 - 1) * row_stride;
                shift[q] = (uint32_t)(reinterpret_cast<uintptr_t>(row) & 3u);
                const uint32_t *row4 = reinterpret_cast<const uint32_t *>(row - shift[q]);
                const uint32_t ndw = (shift[q] + span_bytes + 3) / 4;
#pragma unroll
                for (int it = 0; it < NIT; it++) v[q][it] = row4[min(lane + 64u * it, ndw - 1)];
            }
#pragma unroll
            for (int q = 0; q < RB; q++) {
                uint32_t *to = reinterpret_cast<uint32_t *>(rowbuf + q * raw_pitch);
#pragma unroll
                for (int it = 0; it < NIT; it++)
                    if ((lane + 64u * it) * 4 < raw_pitch) to[lane + 64u * it] = v[q][it];
            }
            __builtin_amdgcn_wave_barrier();  // the wave reads back what the wave wrote: program order in the LDS queue is enough, the compiler must keep it
#pragma unroll
            for (int q = 0; q < RB; q++) {
                const uint8_t *taps = rowbuf + q * raw_pitch + shift[q] + (xs - x_lo) * CH;
                int32_t ss = half;
                if (WMAX) {
                    // the taps' bytes as aligned dwords + a byte shift: one byte read per tap and channel costs an LDS pass each (lanes that
                    // want different bytes of one dword are served one after the other), dword reads of neighbouring lanes are one pass
                    constexpr int ND = WMAX ? (WMAX * CH + 3) / 4 : 1;
                    const uint32_t at = (uint32_t)(taps - rz_tile), sh = at & 3u;
                    const uint32_t *t4 = reinterpret_cast<const uint32_t *>(rz_tile + (at & ~3u));
                    uint32_t raw[ND + 1], d[ND];
#pragma unroll
                    for (int j = 0; j <= ND; j++) raw[j] = t4[j];
#pragma unroll
                    for (int j = 0; j < ND; j++) d[j] = __builtin_amdgcn_alignbyte(raw[j + 1], raw[j], sh);
#pragma unroll
                    for (int i = 0; i < WMAX; i++) ss += luma_of<CH>(d, i) * kk[i];
                } else {
                    for (uint32_t i = 0; i < xn; i++) ss += luma_lds<CH>(taps, (int)i) * (int32_t)k[i];
                }
                if (y + q < in1) rz_tile[(y + q - in0) * RZ_TW + lane] = clip8(ss, ax.precision);
            }
        }
    }
    __syncthreads();
    if (o < nw) {
        const int32_t half = 1 << (ay.precision - 1);
        for (uint32_t r = r0 + sub; r < r1; r += 4) {  // r is the same for the whole wave: start, size and weights come through the scalar cache
            const uint32_t ys = ay.start[r] - in0, yn = min(ay.size[r], tile_rows - min(ys, tile_rows));
            const int16_t *k = ay.coef + (size_t)r * ay.window;
            const uint8_t *col = rz_tile + ys * RZ_TW + lane;
            int32_t ss = half;
            for (uint32_t i = 0; i < yn; i++) ss += (int32_t)col[i * RZ_TW] * (int32_t)k[i];
            dst[((size_t)img * nh + r) * dst_pitch + o] = clip8(ss, ay.precision);
        }
    }
}

// ---- both passes on the matrix pipe (Luma8 sources; the default).  A box convolution is a window SUM times one coefficient:
//   clip8((half + sum_i in[start + i] * k) >> p)  =  clip8((half + k * S) >> p),  S = the window sum (integer arithmetic: the same number),
// and window sums of many outputs are one product with a 0/1 band matrix.  One wave owns 32 output rows x 64 output columns:
//   pass 1  v_mfma_i32_32x32x32_i8, A = 32 source rows x 32 source bytes straight from memory (lane = row), B = the band matrix of 32 output
//           columns: the sums arrive with lane = output column and the 32 source rows spread over 16 registers x 2 lane halves -- rounded,
//           clipped and packed to bytes that is exactly the B operand layout of
//   pass 2  the same instruction with A = the band matrix of the 32 output rows over those 32 source rows: lane = output column, registers =
//           output rows; v_permlane32_swap gives every lane all 32 rows of one of the 64 columns, and a row is stored as 64 adjacent bytes.
// Nothing goes through LDS.  (Bytes are unsigned, the instruction is signed: operands ^ 0x80, and 128 x the window size comes back in the
// rounding constant.)  Rows of any alignment: aligned dwords + v_alignbyte, so the range check of the buffer resource never cuts a pixel.
typedef int rz_v4i __attribute__((ext_vector_type(4)));
typedef int rz_v16i __attribute__((ext_vector_type(16)));
typedef unsigned int rz_v4u __attribute__((ext_vector_type(4)));
constexpr int RZ_KG = 6;  // K steps per group of loads

// bits [lo, hi) of a 16-slot group as 16 bytes of 0 / 1
__device__ __forceinline__ rz_v4i band16(int lo, int hi)
{
    lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo);
    hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
    const uint32_t m = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    rz_v4i r;
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = (int)((((m >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u);
    return r;
}

__global__ void __launch_bounds__(64) resize_mfma_kernel(const uint8_t *__restrict__ px, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride, uint32_t nw,
                                                         uint32_t nh, DevAxis ax, DevAxis ay, uint8_t *__restrict__ dst, uint32_t dst_pitch, size_t dst_stride)
{
    const int lane = threadIdx.x, n = lane & 31, kh = lane >> 5;
    const uint32_t img = blockIdx.z, r0 = blockIdx.y * 32, o0 = blockIdx.x * 64;
    // the image as a buffer of whole dwords around its bytes
    const uint8_t *base = px + (size_t)img * image_stride;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(base) & 3u);
    const uint32_t bytes = (uint32_t)((size_t)(h - 1) * row_stride + w);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(base - mis), 0, (int)((mis + bytes + 3u) & ~3u), 0x00027000);

    // source window of the task (uniform)
    const uint32_t o_last = min(o0 + 63u, nw - 1u), r_last = min(r0 + 31u, nh - 1u);
    const int x_lo = (int)ax.start[o0] & ~15, x_hi = (int)(ax.start[o_last] + ax.size[o_last]);
    const int y_lo = (int)ay.start[r0], y_hi = (int)(ay.start[r_last] + ay.size[r_last]);
    const int n_ks = (x_hi - x_lo + 31) / 32, n_yb = (y_hi - y_lo + 31) / 32;

    // per-lane tables of pass 1: the two column blocks' windows in slots of the lane's own half, coefficient and rounding constant
    int xs[2], xe[2], cx[2], c0[2], ks_lo[2], ks_hi[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        const uint32_t o = o0 + 32 * cb + n, oc = min(o, nw - 1u);
        const int st = (int)ax.start[oc], sz = o < nw ? (int)ax.size[oc] : 0;
        xs[cb] = st - x_lo - 16 * kh;
        xe[cb] = xs[cb] + sz;
        cx[cb] = ax.c1[oc];
        c0[cb] = (1 << (ax.precision - 1)) + 128 * sz * cx[cb];
        // K steps that meet this block's windows (uniform)
        const uint32_t ob0 = min(o0 + 32u * cb, nw - 1u), ob1 = min(o0 + 32u * cb + 31u, nw - 1u);
        ks_lo[cb] = ((int)ax.start[ob0] - x_lo) / 32;
        ks_hi[cb] = ((int)(ax.start[ob1] + ax.size[ob1]) - x_lo + 31) / 32;
        if (o0 + 32u * cb >= nw) ks_hi[cb] = ks_lo[cb] = 0;
    }
    // pass 2: this lane's output row as an A-operand row, its window relative to the first source row
    const uint32_t rr = min(r0 + (uint32_t)n, nh - 1u);
    const int ys = (int)ay.start[rr] - y_lo, ye = ys + (r0 + (uint32_t)n < nh ? (int)ay.size[rr] : 0);

    rz_v4i bx0[2][RZ_KG];  // band matrices of K steps 0 .. RZ_KG - 1
#pragma unroll
    for (int cb = 0; cb < 2; cb++)
#pragma unroll
        for (int u = 0; u < RZ_KG; u++) bx0[cb][u] = band16(xs[cb] - 32 * u, xe[cb] - 32 * u);
    const bool unaligned = (mis | (uint32_t)(row_stride & 3)) != 0;  // rows on dword boundaries: the 16 bytes of a step are four whole dwords
    // pass 2's per-row constants: lane n holds those of row r0 + n (read back with v_readlane: the row of a register is static)
    const int my_cy = ay.c1[rr], my_sz = (int)ay.size[rr];

    // One group of K steps: the group's loads go out together; every step multiplies into BOTH column blocks (a block whose windows the
    // step does not meet has an all-zero band matrix there: no branch, no copies of the accumulators).
    auto k_group = [&](int ks0, uint32_t row_off, bool row_ok, bool first, rz_v16i(&acc1)[2]) {
        rz_v4u d4[RZ_KG];
        uint32_t d5[RZ_KG], sh[RZ_KG];
#pragma unroll
        for (int u = 0; u < RZ_KG; u++) {
            const uint32_t a = row_off + 32u * (uint32_t)(ks0 + u), al = (row_ok && ks0 + u < n_ks) ? a & ~3u : 0x80000000u;
            sh[u] = a & 3u;
            d4[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, al, 0, 0);
            d5[u] = 0;
            if (unaligned) d5[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, al + 16u, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < RZ_KG; u++) {
            rz_v4i av;
            av[0] = (int)(__builtin_amdgcn_alignbyte(d4[u][1], d4[u][0], sh[u]) ^ 0x80808080u);
            av[1] = (int)(__builtin_amdgcn_alignbyte(d4[u][2], d4[u][1], sh[u]) ^ 0x80808080u);
            av[2] = (int)(__builtin_amdgcn_alignbyte(d4[u][3], d4[u][2], sh[u]) ^ 0x80808080u);
            av[3] = (int)(__builtin_amdgcn_alignbyte(d5[u], d4[u][3], sh[u]) ^ 0x80808080u);
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
                acc1[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, first ? bx0[cb][u] : band16(xs[cb] - 32 * (ks0 + u), xe[cb] - 32 * (ks0 + u)), acc1[cb], 0, 0, 0);
        }
    };

    rz_v16i acc2[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) acc2[cb] = rz_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#pragma unroll 1
    for (int b = 0; b < n_yb; b++) {
        const int yb = y_lo + 32 * b, y = yb + n;
        const bool row_ok = y < y_hi && y < (int)h;
        const uint32_t row_off = mis + (uint32_t)y * (uint32_t)row_stride + (uint32_t)x_lo + 16u * (uint32_t)kh;
        rz_v16i acc1[2];
#pragma unroll
        for (int cb = 0; cb < 2; cb++) acc1[cb] = rz_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        k_group(0, row_off, row_ok, true, acc1);
#pragma unroll 1
        for (int ks0 = RZ_KG; ks0 < n_ks; ks0 += RZ_KG) k_group(ks0, row_off, row_ok, false, acc1);  // (sources beyond ~2.9 x the thumbnail)
        // band matrix of the output rows over source rows yb + 8 q + 4 kh + i (slot (kh, 4 q + i))
        rz_v4i by;
        {
            const int lo = ys - 32 * b, hi = ye - 32 * b;
            const int l2 = lo < 0 ? 0 : (lo > 32 ? 32 : lo), h2 = hi < 0 ? 0 : (hi > 32 ? 32 : hi);
            const uint32_t m = h2 > l2 ? (uint32_t)((1ull << h2) - 1ull) & ~(uint32_t)((1ull << l2) - 1ull) : 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) by[q] = (int)((((m >> (8 * q + 4 * kh)) & 15u) * 0x00204081u) & 0x01010101u);
        }
#pragma unroll
        for (int cb = 0; cb < 2; cb++) {
            rz_v4i hb;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t d = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    int v = (__mul24(acc1[cb][4 * q + i], cx[cb]) + c0[cb]) >> ax.precision;  // |sum| < 2^15, coefficient <= 2^15
                    asm volatile("" : "+v"(v));  // (opaque between shift and clamp: ROCm 7.2 fuses them into v_ashr_pk_u8_i32 and then ORs bytes into bits 16..31 of its result as if the instruction cleared them)
                    v = v < 0 ? 0 : (v > 255 ? 255 : v);
                    d |= (uint32_t)v << (8 * i);
                }
                hb[q] = (int)(d ^ 0x80808080u);
            }
            acc2[cb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(by, hb, acc2[cb], 0, 0, 0);
        }
    }
    // lane l: column o0 + l; rows r0 + 8 q + i from P[4 q + i], r0 + 8 q + 4 + i from R[4 q + i]
    const uint32_t o = o0 + (uint32_t)lane;
    uint8_t *out = dst + (size_t)img * dst_stride + (size_t)r0 * dst_pitch + o;
    const int half_y = 1 << (ay.precision - 1);
    int res[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const auto sw = __builtin_amdgcn_permlane32_swap(acc2[0][j], acc2[1][j], false, false);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int row = 8 * (j >> 2) + 4 * t + (j & 3);
            const int cy = __builtin_amdgcn_readlane(my_cy, row), sz = __builtin_amdgcn_readlane(my_sz, row);
            int v = (__mul24((int)sw[t] + 128 * sz, cy) + half_y) >> ay.precision;
            asm volatile("" : "+v"(v));
            res[row] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
    }
    if (o < nw) {
#pragma unroll
        for (int row = 0; row < 32; row++)
            if (r0 + (uint32_t)row < nh) out[(uint32_t)row * dst_pitch] = (uint8_t)res[row];  // (uniform)
    }
}

template <int CH, int NIT>
void launch_resize_fused_n(dim3 grid, size_t lds, hipStream_t stream, const uint8_t *src, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride, uint32_t nw,
                           uint32_t nh, const DevAxis &dx, const DevAxis &dy, uint32_t th, uint32_t tile_rows, uint32_t raw_pitch, uint8_t *dst, uint32_t dst_pitch)
{
    switch (dx.window) {
    case 3: hipLaunchKernelGGL((resize_fused_kernel<CH, 3, NIT>), grid, dim3(256), lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, dst, dst_pitch); break;
    case 5: hipLaunchKernelGGL((resize_fused_kernel<CH, 5, NIT>), grid, dim3(256), lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, dst, dst_pitch); break;
    case 7: hipLaunchKernelGGL((resize_fused_kernel<CH, 7, NIT>), grid, dim3(256), lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, dst, dst_pitch); break;
    default: hipLaunchKernelGGL((resize_fused_kernel<CH, 0, NIT>), grid, dim3(256), lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, dst, dst_pitch); break;
    }
}

constexpr int RZ_NIT_MAX = 8;  // staged rows of up to 8 x 256 bytes

template <int CH>
void launch_resize_fused(dim3 grid, size_t lds, hipStream_t stream, const uint8_t *src, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride, uint32_t nw,
                         uint32_t nh, const DevAxis &dx, const DevAxis &dy, uint32_t th, uint32_t tile_rows, uint32_t raw_pitch, uint8_t *dst, uint32_t dst_pitch)
{
    const uint32_t nit = (raw_pitch / 4 + 63) / 64;
#define RZ_GO(N) launch_resize_fused_n<CH, N>(grid, lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, dst, dst_pitch)
    if (nit <= 1) RZ_GO(1);
    else if (nit <= 2) RZ_GO(2);
    else if (nit <= 3) RZ_GO(3);
    else if (nit <= 4) RZ_GO(4);
    else RZ_GO(RZ_NIT_MAX);
#undef RZ_GO
}

}  // namespace

// Device copies of the two coefficient tables of a source geometry, built once and kept in the context (a scan meets few
// distinct geometries: the sizes its cameras produce).
namespace {
struct DevAxisOwner {
    DevAxis ax{};
    void *start = nullptr, *size = nullptr, *coef = nullptr, *c1 = nullptr;
    std::vector<uint32_t> h_start, h_size;  // host copies: the fused kernel's tile height is chosen from them
    bool uniform = false;
};
struct AxisCache {
    std::map<std::pair<uint32_t, uint32_t>, DevAxisOwner> axes;  // (in_size, out_size) -> tables
};
constexpr size_t kAxisCacheMax = 256;

int device_axis(rph_ctx *ctx, uint32_t in_size, uint32_t out_size, DevAxis &out, const DevAxisOwner **owner = nullptr)
{
    if (!ctx->axis_cache) ctx->axis_cache = new AxisCache();
    AxisCache &c = *static_cast<AxisCache *>(ctx->axis_cache);
    auto it = c.axes.find({in_size, out_size});
    if (it == c.axes.end()) {
        if (c.axes.size() >= kAxisCacheMax) {  // rare: drop everything (kernels of any stream may still read the tables)
            RPH_HIP_CHECK(hipDeviceSynchronize());
            for (auto &kv : c.axes) {
                (void)hipFree(kv.second.start);
                (void)hipFree(kv.second.size);
                (void)hipFree(kv.second.coef);
                (void)hipFree(kv.second.c1);
            }
            c.axes.clear();
        }
        const Axis a = build_axis(in_size, out_size);
        DevAxisOwner o;
        RPH_HIP_CHECK(hipMalloc(&o.start, a.start.size() * 4));
        RPH_HIP_CHECK(hipMalloc(&o.size, a.size.size() * 4));
        RPH_HIP_CHECK(hipMalloc(&o.coef, std::max<size_t>(a.coef.size(), 1) * 2));
        RPH_HIP_CHECK(hipMemcpy(o.start, a.start.data(), a.start.size() * 4, hipMemcpyHostToDevice));  // synchronous: once per geometry
        RPH_HIP_CHECK(hipMemcpy(o.size, a.size.data(), a.size.size() * 4, hipMemcpyHostToDevice));
        RPH_HIP_CHECK(hipMemcpy(o.coef, a.coef.data(), a.coef.size() * 2, hipMemcpyHostToDevice));
        RPH_HIP_CHECK(hipMalloc(&o.c1, a.c1.size() * 4));
        RPH_HIP_CHECK(hipMemcpy(o.c1, a.c1.data(), a.c1.size() * 4, hipMemcpyHostToDevice));
        o.uniform = a.uniform;
        o.ax = DevAxis{(const uint32_t *)o.start, (const uint32_t *)o.size, (const int16_t *)o.coef, (const int32_t *)o.c1, a.window, a.precision};
        o.h_start = a.start;
        o.h_size = a.size;
        it = c.axes.emplace(std::make_pair(in_size, out_size), std::move(o)).first;
    }
    out = it->second.ax;
    if (owner) *owner = &it->second;
    return RPH_OK;
}
}  // namespace

void rph_resize_forget(rph_ctx *ctx)
{
    if (ctx->axis_cache) {
        AxisCache *c = static_cast<AxisCache *>(ctx->axis_cache);
        for (auto &kv : c->axes) {
            (void)hipFree(kv.second.start);
            (void)hipFree(kv.second.size);
            (void)hipFree(kv.second.coef);
            (void)hipFree(kv.second.c1);
        }
        delete c;
        ctx->axis_cache = nullptr;
    }
    if (ctx->rz_scratch) (void)hipFree(ctx->rz_scratch);
    ctx->rz_scratch = nullptr;
    ctx->rz_bytes = 0;
}

// Called with ctx->mu held (rph_pdq_hash_batch_dev).  Asynchronous on `stream`: nothing is allocated per call once the
// geometry has been seen and the scratch has grown to the batch size, and nothing waits for the device.
int rph_launch_pdq_resized(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                           size_t row_stride, size_t image_stride, uint8_t *d_hash, float *d_quality, float *d_coeffs,
                           uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    uint32_t nw, nh;
    rph_pdq_target_dimensions(w, h, RPH_PDQ_MAX_DIM, &nw, &nh);  // pdqhash.rs:183
    DevAxis dx, dy;
    const DevAxisOwner *ox = nullptr, *oy = nullptr;
    int rc;
    if ((rc = device_axis(ctx, w, nw, dx, &ox)) != RPH_OK || (rc = device_axis(ctx, h, nh, dy, &oy)) != RPH_OK) return rc;
    // (the second lookup may have emptied a full cache and with it the first one's entry: look both up again, now both are there or fit)
    if ((rc = device_axis(ctx, w, nw, dx, &ox)) != RPH_OK || (rc = device_axis(ctx, h, nh, dy, &oy)) != RPH_OK) return rc;
    // widest stretch of source columns a tile of RZ_TW output columns needs -> bytes per staged row (+ 3 alignment bytes, whole dwords, 16-byte pitch)
    uint32_t span = 0;
    for (uint32_t c0 = 0; c0 < nw; c0 += RZ_TW) {
        const uint32_t c1 = std::min(c0 + (uint32_t)RZ_TW, nw) - 1;
        span = std::max(span, std::min(ox->h_start[c1] + ox->h_size[c1], w) - ox->h_start[c0]);
    }
    // (+ the widest window once more: taps of weight 0 behind an output's last real one are still read from the row buffer)
    const uint32_t raw_pitch = (uint32_t)((((size_t)span + (size_t)dx.window) * channels + 3 + 3 + 8 + 15) & ~(size_t)15);

    // the fused kernel: rows of output per workgroup such that the input rows they need fit a 32 KB tile
    uint32_t th = 16, tile_rows = 0;
    const bool may_fuse = (channels == 1 || channels == 3 || channels == 4) && ctx->pdq_kernel != 0;  // (rph_pdq_set_kernel(ctx, 0): the plain two-pass kernels, the baseline the tests compare with)
    for (; may_fuse; th /= 2) {
        tile_rows = 0;
        for (uint32_t r0 = 0; r0 < nh; r0 += th) {
            const uint32_t r1 = std::min(r0 + th, nh);
            tile_rows = std::max(tile_rows, oy->h_start[r1 - 1] + oy->h_size[r1 - 1] - oy->h_start[r0]);
        }
        if ((size_t)tile_rows * RZ_TW <= 32768 || th == 1) break;
    }
    const bool fused = may_fuse && (size_t)tile_rows * RZ_TW <= 32768 && (nh + th - 1) / th <= 65535 && raw_pitch <= (uint32_t)RZ_NIT_MAX * 256;

    // the thumbnail's rows start on 16-byte boundaries where the streaming hasher (pdq_stream.hip) reads them as whole dwords
    const uint32_t np = fused ? (nw + 15u) & ~15u : nw;
    const size_t full = fused ? 0 : (size_t)w * h, tmp = fused ? 0 : (size_t)nw * h, small = (size_t)np * nh;
    // images per launch: the streaming hasher needs nothing but the thumbnails (1 GiB of them); the multi-pass hasher takes as many images
    // as its f32 planes allow; the two-pass resize is bounded by its full-size luma planes
    const bool stream_hasher = fused && ctx->pdq_kernel != 5 && ctx->pdq_kernel != 0 && (n >= RPH_STREAM_MIN_IMAGES || ctx->pdq_kernel == 6) &&
                               rph_pdq_stream_supported(nullptr, nw, nh, 1, np, small);
    uint32_t chunk = (uint32_t)std::max<size_t>(1, (fused ? (size_t)1 << 30 : (size_t)256 << 20) / (stream_hasher ? small : (fused ? small * 4 : full)));
    chunk = std::min(std::min(chunk, n), 65535u);
    const size_t need = (full + tmp + small) * chunk;
    if (ctx->rz_bytes < need) {
        RPH_HIP_CHECK(hipDeviceSynchronize());  // kernels of any stream may still be using the old planes
        if (ctx->rz_scratch) RPH_HIP_CHECK(hipFree(ctx->rz_scratch));
        ctx->rz_scratch = nullptr;
        ctx->rz_bytes = 0;
        RPH_HIP_CHECK(hipMalloc((void **)&ctx->rz_scratch, need));
        ctx->rz_bytes = need;
    }
    // the planes are shared by every caller stream, like the generic kernel's f32 planes, and are ordered by the same event
    if (!ctx->scratch_done) RPH_HIP_CHECK(hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming));
    if (ctx->scratch_used && ctx->scratch_stream != stream) RPH_HIP_CHECK(hipStreamWaitEvent(stream, ctx->scratch_done, 0));
    uint8_t *p_luma = ctx->rz_scratch, *p_tmp = p_luma + full * chunk, *p_small = p_tmp + tmp * chunk;
    for (uint32_t first = 0; first < n; first += chunk) {
        const uint32_t m = std::min(chunk, n - first);
        if (fused) {
            const dim3 grid((nw + RZ_TW - 1) / RZ_TW, (nh + th - 1) / th, m);
            const uint8_t *src = d_px + (size_t)first * image_stride;
            const size_t lds = (((size_t)tile_rows * RZ_TW + 15) & ~(size_t)15) + (size_t)4 * 4 * raw_pitch;
            if (channels == 1 && ox->uniform && oy->uniform && ctx->pdq_kernel != 5 && (size_t)h * row_stride < ((size_t)1 << 31))
                hipLaunchKernelGGL(resize_mfma_kernel, dim3((nw + 63) / 64, (nh + 31) / 32, m), dim3(64), 0, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy,
                                   p_small, np, small);
            else if (channels == 1)
                launch_resize_fused<1>(grid, lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, p_small, np);
            else if (channels == 3)
                launch_resize_fused<3>(grid, lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, p_small, np);
            else
                launch_resize_fused<4>(grid, lds, stream, src, w, h, row_stride, image_stride, nw, nh, dx, dy, th, tile_rows, raw_pitch, p_small, np);
        } else {
        hipLaunchKernelGGL(luma_u8_kernel, dim3(grid_for((uint64_t)m * full)), dim3(256), 0, stream, d_px + (size_t)first * image_stride, m, w,
                           h, channels, row_stride, image_stride, p_luma);
        hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((uint64_t)m * h * nw)), dim3(256), 0, stream, (const uint8_t *)p_luma, p_tmp, m, w, h, nw, dx);
        hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((uint64_t)m * nh * nw)), dim3(256), 0, stream, (const uint8_t *)p_tmp, p_small, m, h, nw, nh, dy);
        }
        RPH_HIP_CHECK(hipGetLastError());
        // generate_pdq_from_luma on the thumbnail (no second size check in the reference: a 4000x5 input is hashed from 512x1).
        // It records scratch_done on `stream` when it is through, which also covers the planes above.
        if (stream_hasher) {  // one streaming kernel per thumbnail
            rc = rph_launch_pdq_stream(ctx, (const uint8_t *)p_small, m, nw, nh, np, small, d_hash + (size_t)first * 32, d_quality ? d_quality + first : nullptr,
                                       d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr, d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr,
                                       d_valid ? d_valid + first : nullptr, stream);
            if (rc != RPH_OK) return rc;
            RPH_HIP_CHECK(hipEventRecord(ctx->scratch_done, stream));  // (the multi-pass hasher records it itself)
            ctx->scratch_stream = stream;
            ctx->scratch_used = true;
            continue;
        }
        rc = rph_launch_pdq_generic(ctx, (const uint8_t *)p_small, m, nw, nh, 1, np, small, d_hash + (size_t)first * 32,
                                    d_quality ? d_quality + first : nullptr, d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr,
                                    d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr, d_valid ? d_valid + first : nullptr, stream);
        if (rc != RPH_OK) return rc;
    }
    return RPH_OK;
}

// debug / tests: the thumbnails the last pre-downsample call of this context left in its scratch (first `bytes` bytes, rows of align16(new_w))
extern "C" int rph_debug_copy_thumbnails(rph_ctx *ctx, void *host_dst, size_t bytes)
{
    if (!ctx || !host_dst || bytes > ctx->rz_bytes) return RPH_ERR_INVALID_ARG;
    RPH_HIP_CHECK(hipDeviceSynchronize());
    RPH_HIP_CHECK(hipMemcpy(host_dst, ctx->rz_scratch, bytes, hipMemcpyDeviceToHost));
    return RPH_OK;
}
