// pdq_fused512.hip -- fused single-pass PDQ hash of 512x512 RGB8 images for gfx950.
//
// Replaces, for this geometry, the whole of generate_pdq_from_luma + to_luma601
// (/root/reference/src/pdqhash.rs:238-284, 341-460) and feeds the shared tail (pdq_tail.hpp).
// The image is read from HBM exactly once (786 432 B); nothing but the outputs is written back.
//
// One wave (64 lanes) owns one image and never synchronises with another wave.  It walks the image
// in 8 bands of 64 rows x 4 strips of 128 columns (128 px = 384 B = exactly three 128-B cache lines, so
// no line is ever requested by two tiles).  Per (band, strip) tile:
//   LOAD   two half tiles of 32 rows; lane (c, g) = 8 columns x 8 rows: two global_load_dwordx3 per row,
//          Rec.601 luma (exact integers), packed to f16; the 7 luma rows above come from the lane with g-1
//          through LDS (or from the rows saved by the previous half tile / band), and the vertical 8-row
//          window sums V (<= 2040, exact in f16) go to a 64x128 f16 tile in LDS.
//   SCAN   lane r = image row: walks the tile left to right keeping the horizontal 8-window sum Hs
//          of V in f32 (= the pass-1 box value x 64, an exact integer) and the reference's pass-2 row
//          running sum (sum += in[ri]; sum -= in[li]) in the reference's order; emits the pass-2 row
//          value at the 8 sampled columns x = 8j+4.
// Twice per band the sampled values go through the pass-2 column chain (lane = sampled column), which
// leaves B[i][j] in registers, lane j holding column j -- exactly what pdq_tail() consumes.
//
// Exactness (tests/test_fused_scheme.py proves each point against the sequential oracle):
//   * pass 1 is an exact 8x8 integer box sum / 64 except on the frame: columns {0,1,2,508,509,510}
//     (windows of 5,6,7 in the row pass) are inexact and their column pass is run as the reference's
//     sequential running sum ("edge chains", 6 lanes); rows {0,1,2,508,509,510} divide an exact sum once.
//   * all pass-2 quantities are carried scaled by 64 (power of two: commutes with f32 rounding).
//   * no FMA contraction anywhere except where the operands are exact integers (luma).
#include "pdq_tail.hpp"
#include "rph_internal.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// Strip geometry.  Two builds of the same kernel:
//   SW = 64 : 8 strips, tile 64x64, one build step per tile, 17.9 KB LDS -> 8 waves per CU (fastest: the kernel is
//             VALU-issue bound, so the extra occupancy wins), but a strip row is 192 B = 1.5 cache lines, so every third
//             line is requested by two tiles (L2->fabric traffic 1.30 x algorithmic, profiles/r01_pmc_summary.txt)
//   SW = 128: 4 strips, tile 64x128 built as two 32-row halves, 26.1 KB LDS -> 6 waves per CU; a strip row is 384 B =
//             exactly 3 lines, traffic 1.11 x algorithmic (the rest is the edge pre-pass)
template <int SW_, int CH_ = 3, bool PARK_ = true>
struct Geo {
    static constexpr int SW = SW_;                       // strip width in pixels
    static constexpr int CH = CH_;                       // 3 = Rgb8 input, 1 = Luma8 input (the reference borrows a Luma8 image as it is, pdqhash.rs:176)
    static constexpr int NSTRIP = 512 / SW;
    static constexpr int CL = SW / 8;                    // lanes across a strip row, 8 px each
    static constexpr int NG = 64 / CL;                   // lane groups of 8 rows -> NG * 8 rows per build step
    static constexpr int NH = 64 / (NG * 8);             // build steps ("halves") per tile: 1 or 2
    static constexpr int NOCT = SW / 8;                  // octets per strip in the scan
    static constexpr int TILE_PITCH = SW * 2 + 16;       // bytes per V-tile row: SW f16 + 16 B pad (conflict-free b128 reads, lane = row)
    static constexpr int TILE_BYTES = 64 * TILE_PITCH;   // the sample transpose buffer and the tail scratch alias the dead tile
    static constexpr int OFF_TILE = 0;
    // exchange area: NH == 2 -> rows 32..63 of the tile (dead, or not yet written, whenever it is used); NH == 1 -> the whole dead tile
    static constexpr int OFF_XCHG = NH == 2 ? 32 * TILE_PITCH : 0;
    static constexpr int OFF_STATE = TILE_BYTES;         // 7 luma rows x 512 columns, f16 = 7168 B
    static constexpr int OFF_EDGE = OFF_STATE + 7168;    // 6 chains x 64 rows x f32 = 1536 B (rv in, edge values out, in place)
    // The states of the two column recurrences that are touched only between strips -- the pass-2 column chain (sum + 8 ring values per
    // lane, twice per band) and the six edge chains (once per band) -- wait in LDS while the strips are walked: 18 VGPRs less across the
    // strip loop, which is what the 64-px build was short of (10 spilled VGPRs = 5.2 KB of scratch traffic per image).  The 2520 bytes
    // still leave 8 waves per CU (8 x 20 440 B); the 128-px build has the registers and would lose a wave to them.
#ifndef RPH_PARK_CHAINS
#define RPH_PARK_CHAINS 1
#endif
    static constexpr bool PARK = RPH_PARK_CHAINS && PARK_ && SW == 64;  // (PARK_ = false: the low-latency kernel, whose waves run one band each and keep no such state)
    static constexpr int OFF_PARK_C = OFF_EDGE + 1536;                 // [9][64] floats
    static constexpr int OFF_PARK_E = OFF_PARK_C + (PARK ? 9 * 64 * 4 : 0);  // [9][6] floats
    static constexpr int LDS_BYTES = OFF_PARK_E + (PARK ? 9 * 6 * 4 : 0);    // SW 64: 20440 B (17920 without the parked states), SW 128: 26112 B
    static constexpr int WAVES_PER_SIMD = 2;
    static_assert(64 * 33 * 4 <= TILE_BYTES, "sample transpose buffer must fit in the dead V tile");
    static_assert((NG - 1) * 7 * CL * 16 <= (NH == 2 ? 32 : 64) * TILE_PITCH, "exchange area must fit");
    static_assert(rph::TAIL_LDS_FLOATS * 4 <= TILE_BYTES, "tail scratch must fit in the dead V tile");
};
constexpr int SAMP_PITCH = 33;                  // sample transpose buffer: 64 rows x 33 floats = 8448 B

struct Px8 {  // 8 RGB pixels = 24 bytes = 6 dwords, loaded as 2 x dwordx3
    uint32_t d[6];
};
struct __attribute__((packed, aligned(4))) U3 {
    uint32_t x, y, z;
};

struct __attribute__((packed, aligned(4))) U2 {
    uint32_t x, y;
};
template <int CH = 3>
__device__ __forceinline__ Px8 load_px8(const uint8_t *p)
{
    Px8 r;
    if (CH == 1) {  // 8 luma bytes: d[0], d[1]
        const U2 a = *reinterpret_cast<const U2 *>(p);
        r.d[0] = a.x; r.d[1] = a.y; r.d[2] = r.d[3] = r.d[4] = r.d[5] = 0;
        return r;
    }
    const U3 a = *reinterpret_cast<const U3 *>(p);
    const U3 b = *reinterpret_cast<const U3 *>(p + 12);
    r.d[0] = a.x; r.d[1] = a.y; r.d[2] = a.z; r.d[3] = b.x; r.d[4] = b.y; r.d[5] = b.z;
    return r;
}

// to_luma601 (pdqhash.rs:270-273): (299 r + 587 g + 114 b + 500) / 1000, truncating.  In f32 every
// intermediate is an integer < 2^24 (fma is exact) and trunc(num * 0.001f) equals the integer quotient
// for every reachable numerator (tests/test_fused_scheme.py::test_luma_float_formula...).
template <int P>
__device__ __forceinline__ float luma_px(const Px8 &v)
{
    constexpr int B = 3 * P;
    const float r = (float)((v.d[B >> 2] >> (8 * (B & 3))) & 0xFFu);
    const float g = (float)((v.d[(B + 1) >> 2] >> (8 * ((B + 1) & 3))) & 0xFFu);
    const float b = (float)((v.d[(B + 2) >> 2] >> (8 * ((B + 2) & 3))) & 0xFFu);
    const float num = __builtin_fmaf(299.0f, r, __builtin_fmaf(587.0f, g, __builtin_fmaf(114.0f, b, 500.0f)));
    return __builtin_truncf(num * 0.001f);
}

template <int CH = 3>
__device__ __forceinline__ void luma8(const Px8 &v, float (&l)[8])
{
    if (CH == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = (float)((v.d[i >> 2] >> (8 * (i & 3))) & 0xFFu);
        return;
    }
    l[0] = luma_px<0>(v); l[1] = luma_px<1>(v); l[2] = luma_px<2>(v); l[3] = luma_px<3>(v);
    l[4] = luma_px<4>(v); l[5] = luma_px<5>(v); l[6] = luma_px<6>(v); l[7] = luma_px<7>(v);
}

// 8 pixels -> 4 packed f16 luma pairs.
// Bytes become f16 without a conversion: v_perm_b32 pairs each byte with the constant 0x64, and 0x6400 | v is
// the f16 number 1024 + v.  v_dot2_f32_f16 then forms 299 r + 587 g + 114 b (+ the bias 1024 * 1000, removed by
// the accumulator constant) with f32 accumulation: all products and sums are integers < 2^24, hence exact.
struct Row8 {
    h2 q[4];
};
__device__ __forceinline__ h2 h2_bits(uint32_t u) { return __builtin_bit_cast(h2, u); }
__device__ __forceinline__ h2 byte_pair_lo(uint32_t d) { return h2_bits(__builtin_amdgcn_perm(0x64646464u, d, 0x04010400u)); }
__device__ __forceinline__ h2 byte_pair_hi(uint32_t d) { return h2_bits(__builtin_amdgcn_perm(0x64646464u, d, 0x04030402u)); }

// a.x * k.x + a.y * k.y with f32 accumulation from the inline constant 0: the three-operand VOP3P form (the compiler would
// otherwise emit v_mov 0 + the two-address v_dot2c)
// `after` is not read by the instruction: it only orders this row behind the previous one, so that the scheduler cannot
// hoist the dot products of all eight prefetched rows to the top of the loop (64 live values, spills).
__device__ __forceinline__ float dot2_from_zero(h2 a, h2 k, uint32_t after)
{
    float r;
    asm volatile("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(r) : "v"(__builtin_bit_cast(uint32_t, a)), "s"(__builtin_bit_cast(uint32_t, k)), "v"(after));
    return r;
}

struct Pairs12 {
    h2 P[12];  // byte stream r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 | ... : pair p holds bytes 2p, 2p+1 as f16 1024 + byte
};
__device__ __forceinline__ Pairs12 byte_pairs(const Px8 &v)
{
    Pairs12 r;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        r.P[2 * i] = byte_pair_lo(v.d[i]);
        r.P[2 * i + 1] = byte_pair_hi(v.d[i]);
    }
    return r;
}
__device__ __forceinline__ Row8 luma_row(const Pairs12 &pr, bool zero, uint32_t &order)
{
    const h2 *P = pr.P;
    const h2 k_rg = h2{(_Float16)299.0f, (_Float16)587.0f};   // (r, g)
    const h2 k_b0 = h2{(_Float16)114.0f, (_Float16)0.0f};      // (b, next r)
    const h2 k_0r = h2{(_Float16)0.0f, (_Float16)299.0f};      // (prev b, r)
    const h2 k_gb = h2{(_Float16)587.0f, (_Float16)114.0f};    // (g, b)
    // The 1024 bias of every byte adds 1024 * (299 + 587 + 114) = 1024 * 1000 to the numerator: n = 1 024 000 + (299 r + 587 g + 114 b)
    // (< 2^24, exact), so y = fma(n, 0.001f, 0.5f) = 1024 + (299 r + 587 g + 114 b + 500) / 1000 lies in [1024.5, 1280), where
    // consecutive f16 numbers are 1 apart: the round-toward-zero f16 conversion IS the floor (fl(0.001f) > 0.001 and the fused
    // rounding error is < 1.3e-4, far below the 0.001 spacing of the fractional parts), and the accumulators start from the
    // inline constant 0 (no v_mov of a bias).
    // 0.5 kept in a VGPR the compiler cannot see through: with an inline-constant addend it would pick the VOP3 v_fma_f32, which
    // issues at half the rate of the VOP2 v_fmamk_f32 (tools/valu_rate.hip)
    float vhalf;
    asm("v_mov_b32 %0, 0.5" : "=v"(vhalf));
    h2 packed[4];
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {  // two pixels = 6 bytes = pairs 3pp, 3pp+1, 3pp+2
        const float n0 = __builtin_amdgcn_fdot2(P[3 * pp], k_rg, dot2_from_zero(P[3 * pp + 1], k_b0, order), false);
        const float n1 = __builtin_amdgcn_fdot2(P[3 * pp + 1], k_0r, dot2_from_zero(P[3 * pp + 2], k_gb, order), false);
        const float y0 = __builtin_fmaf(n0, 0.001f, vhalf), y1 = __builtin_fmaf(n1, 0.001f, vhalf);  // v_fmamk_f32 (VOP2, full rate)
        packed[pp] = __builtin_bit_cast(h2, __builtin_amdgcn_cvt_pkrtz(y0, y1)) - h2{(_Float16)1024.0f, (_Float16)1024.0f};
    }
    order = __builtin_bit_cast(uint32_t, packed[3]);
    Row8 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r.q[i] = packed[i];
        if (zero) r.q[i] = h2{(_Float16)0, (_Float16)0};
    }
    return r;
}
// Luma8 input: the bytes are the luma; pairs of them become f16 (1024 + v) by the same v_perm, and the bias leaves exactly
__device__ __forceinline__ Row8 luma_row_gray(const Px8 &v, bool zero, uint32_t &order)
{
    const h2 bias = h2{(_Float16)1024.0f, (_Float16)1024.0f};
    Row8 r;
    r.q[0] = byte_pair_lo(v.d[0]) - bias;
    r.q[1] = byte_pair_hi(v.d[0]) - bias;
    r.q[2] = byte_pair_lo(v.d[1]) - bias;
    r.q[3] = byte_pair_hi(v.d[1]) - bias;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (zero) r.q[i] = h2{(_Float16)0, (_Float16)0};
    order = __builtin_bit_cast(uint32_t, r.q[3]);
    return r;
}
template <int CH = 3>
__device__ __forceinline__ Row8 pack_row(const Px8 &v, bool zero, uint32_t &order)
{
    if (CH == 1) return luma_row_gray(v, zero, order);
    return luma_row(byte_pairs(v), zero, order);
}
__device__ __forceinline__ uint4 row_bits(const Row8 &r)
{
    uint4 u;
    u.x = __builtin_bit_cast(uint32_t, r.q[0]);
    u.y = __builtin_bit_cast(uint32_t, r.q[1]);
    u.z = __builtin_bit_cast(uint32_t, r.q[2]);
    u.w = __builtin_bit_cast(uint32_t, r.q[3]);
    return u;
}
__device__ __forceinline__ Row8 bits_row(const uint4 &u)
{
    Row8 r;
    r.q[0] = __builtin_bit_cast(h2, u.x);
    r.q[1] = __builtin_bit_cast(h2, u.y);
    r.q[2] = __builtin_bit_cast(h2, u.z);
    r.q[3] = __builtin_bit_cast(h2, u.w);
    return r;
}

// IEEE quotient N / d for the integer numerators that occur here (multiples of 8 below 2^22) and
// d in {4,5,6,7,8}: Markstein's sequence, 1 mul + 2 fma (test_markstein_division_is_ieee...).
__device__ __forceinline__ float div_small(float N, float d, float dinv)
{
    const float q0 = N * dinv;
    const float r = __builtin_fmaf(-q0, d, N);
    return __builtin_fmaf(r, dinv, q0);
}

// Phase separator inside the wave.  The workgroup IS one wave and the LDS executes a wave's DS instructions in
// issue order, so a later ds_read sees an earlier ds_write of any lane without waiting; all that is needed is
// that the compiler keeps the program order.  (__syncthreads() would also drain vmcnt and so serialise the
// prefetched global loads of the next tile behind every phase.)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

// ---------------------------------------------------------------------------------------------
// The kernel
// ---------------------------------------------------------------------------------------------
struct Wave {
    uint8_t *lds;
    const uint8_t *img;
    size_t row_stride;
    uint32_t rs32;  // row_stride (the launcher guarantees 512 * row_stride < 2^32)
    int lane;
    // scan state (lane = row of the band), carried across the 8 strips of a band
    float hs, sum, ring[8];
    uint4 pv;  // previous octet of V (8 x f16)
    float row_d, row_dinv;  // pass-1 column-window divisor of this lane's row (8 except on the frame)
    // pass-2 row outputs at the sampled columns of the current half band (lane = row): 8 enter per strip at the
    // top and the file shifts down by 8, so after 4 strips smp[t] is sample slot t of the half band
    float smp[32];
    float smp_last;  // out[508] (sample 63), produced by the last strip's epilogue
    // pass-2 column chain (lane = sampled column j), carried across the whole image
    float csum, cring[8];
    // edge chains (lanes 0..5), carried across the whole image
    float ecs, ering[8];
    // rows of the decimated buffer produced by the current band (lane = column), fed to the streaming tail
    float bnew[8];
    rph::TailAcc tail;
    bool want_quality;
};

// luma row of lane group g, local row k, in half h of band b
template <class G>
__device__ __forceinline__ int half_row(int b, int h, int g, int k) { return 64 * b + 4 + (G::NG * 8) * h + 8 * g + k; }

// Byte offset (from the image base, < 2^32: the launcher checks 512 * row_stride) of the first of the 8 rows this lane loads in
// half tile (b, s, h), and the largest offset it may use (row 511 of its columns: rows beyond the image are re-reads of row 511
// and are zeroed in luma_row)
template <class G>
__device__ __forceinline__ void half_offsets(const Wave &w, int b, int s, int h, uint32_t &off0, uint32_t &off_max)
{
    const int c = w.lane & (G::CL - 1), g = w.lane / G::CL;
    const uint32_t col = (uint32_t)(G::SW * s + 8 * c) * (uint32_t)G::CH;
    off0 = (uint32_t)half_row<G>(b, h, g, 0) * w.rs32 + col;
    off_max = 511u * w.rs32 + col;
}
template <int CH>
__device__ __forceinline__ Px8 load_px8_at(const Wave &w, uint32_t off) { return load_px8<CH>(w.img + off); }  // uniform base + 32-bit lane offset

// LOAD phase 1 (only for the very first tile of the image): issue the 16 loads of half tile (b, s, h)
template <class G>
__device__ __forceinline__ void half_issue(const Wave &w, int b, int s, int h, Px8 (&pre)[8])
{
    uint32_t off, off_max;
    half_offsets<G>(w, b, s, h, off, off_max);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        pre[k] = load_px8_at<G::CH>(w, off);
        off = off + w.rs32;
        off = off < off_max ? off : off_max;
    }
}

// LOAD phase 2a: luma of the 8 prefetched rows of half tile (b, *, h); as soon as a row's bytes have been widened, its
// registers are refilled with the same row of the next half tile (nb, ns, nh) -- the loads are spread over the luma phase
template <class G, bool LAST_BAND>
__device__ __forceinline__ void half_luma(const Wave &w, int b, int h, Px8 (&pre)[8], Row8 (&L)[8], bool has_next, int nb, int ns, int nh)
{
    const int g = w.lane / G::CL;
    uint32_t off, off_max;
    half_offsets<G>(w, nb, ns, nh, off, off_max);
    uint32_t order = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (G::CH == 1) {
            const Px8 cur = pre[k];
            if (has_next) pre[k] = load_px8_at<1>(w, off);
            off = off + w.rs32;
            off = off < off_max ? off : off_max;
            L[k] = luma_row_gray(cur, LAST_BAND && (half_row<G>(b, h, g, k) >= 512), order);
            continue;
        }
        const Pairs12 pr = byte_pairs(pre[k]);
        if (has_next) pre[k] = load_px8_at<3>(w, off);
        off = off + w.rs32;
        off = off < off_max ? off : off_max;
        L[k] = luma_row(pr, LAST_BAND && (half_row<G>(b, h, g, k) >= 512), order);
    }
}

// LOAD phase 2b: exchange of the 7 rows above, vertical window sums, 32 rows of the V tile
template <class G>
__device__ __forceinline__ void half_build(Wave &w, int s, int h, const Row8 (&L)[8])
{
    const int c = w.lane & (G::CL - 1), g = w.lane / G::CL;
    uint8_t *tile = w.lds + G::OFF_TILE;
    uint8_t *xchg = w.lds + G::OFF_XCHG;
    uint8_t *state = w.lds + G::OFF_STATE + (G::SW * s + 8 * c) * 2;

    // publish rows 1..7 for the lane group below (g + 1)
    if (g < G::NG - 1) {
        uint8_t *x = xchg + ((g * 7) * G::CL + c) * 16;
#pragma unroll
        for (int k = 1; k < 8; k++) *reinterpret_cast<uint4 *>(x + (k - 1) * (G::CL * 16)) = row_bits(L[k]);
    }
    wave_lds_fence();
    // the 7 luma rows above this lane's first row: from lane group g-1, or for g == 0 the rows saved by the
    // previous half tile of these columns (previous band's second half, or this tile's first half)
    Row8 hist[7];
    {
        const uint8_t *hp = (g == 0) ? state : xchg + (((g - 1) * 7) * G::CL + c) * 16;
        const int stride = (g == 0) ? 1024 : G::CL * 16;
#pragma unroll
        for (int j = 0; j < 7; j++) hist[j] = bits_row(*reinterpret_cast<const uint4 *>(hp + j * stride));
    }
    wave_lds_fence();
    if (g == G::NG - 1) {  // rows 1..7 of the last lane group are the history of the next half tile of these columns
#pragma unroll
        for (int k = 1; k < 8; k++) *reinterpret_cast<uint4 *>(state + (k - 1) * 1024) = row_bits(L[k]);
    }
    // V[m] = sum of the 8 luma rows ending at own row m  (V row = luma rows -3..+4 around it)
    const int row0 = (G::NG * 8) * h + 8 * g;
    Row8 v;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h2 a = hist[0].q[i] + hist[1].q[i];
        h2 bb = hist[2].q[i] + hist[3].q[i];
        h2 cc = hist[4].q[i] + hist[5].q[i];
        h2 dd = hist[6].q[i] + L[0].q[i];
        v.q[i] = (a + bb) + (cc + dd);
    }
    *reinterpret_cast<uint4 *>(tile + (row0 + 0) * G::TILE_PITCH + 16 * c) = row_bits(v);
#pragma unroll
    for (int m = 1; m < 8; m++) {
#pragma unroll
        for (int i = 0; i < 4; i++) v.q[i] = (v.q[i] - hist[m - 1].q[i]) + L[m].q[i];  // subtract first: stays <= 2040
        *reinterpret_cast<uint4 *>(tile + (row0 + m) * G::TILE_PITCH + 16 * c) = row_bits(v);
    }
    wave_lds_fence();
}

// The horizontal 8-window sum Hs advances by one column as Hs += V[x + 4] - V[x - 4].  The difference of the two f16 values is
// exact (|d| <= 2040) and is formed for two columns at once (v_pk_add_f16 with a negated operand); v_fma_mix_f32 then adds one
// half of it to the f32 sum, reading the f16 operand directly (no v_cvt).  All sums are exact integers, so this equals the direct
// accumulation.  1.5 VALU instructions per column; the SIMD's VALU issue (85 % busy with two waves, profiles/) is what the kernel
// is short of, so instruction COUNT is what matters here: the s_nop the compiler puts behind a high-half read (gfx950 op_sel
// forwarding hazard) only delays the issuing wave, and its partner takes the slot.
__device__ __forceinline__ uint32_t h2_diff(uint32_t cur, uint32_t prev)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, cur) - __builtin_bit_cast(h2, prev));
}
template <int HI>
__device__ __forceinline__ float hs_step(float hs, uint32_t packed)
{
    float r;
    if (HI)
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(hs));
    else
        asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(hs));
    return r;
}
template <int HI>
__device__ __forceinline__ float hs_sub(float hs, uint32_t packed)
{
    float r;
    if (HI)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(hs));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(packed), "v"(hs));
    return r;
}

// one step of the pass-2 row chain: input in2 (scaled by 64) enters, the input 8 back leaves
template <int E>
__device__ __forceinline__ void row_step(Wave &w, float in2)
{
    constexpr int slot = (E + 4) & 7;  // ri mod 8 with ri = xv - 4
    w.sum = w.sum + in2;
    w.sum = w.sum - w.ring[slot];
    w.ring[slot] = in2;
}

template <bool EDGE_ROWS>
__device__ __forceinline__ float pass1_value(const Wave &w, float hs)
{
    // 64 x pass-1 value at an interior column: Hs (8x8 box sum) for full windows, one IEEE division on the frame rows
    if (EDGE_ROWS) return div_small(hs * 8.0f, w.row_d, w.row_dinv);
    return hs;
}

// SCAN of strip s of the current band.  FIRST/LAST: strip 0 / strip 7 (frame columns).
template <class G, bool EDGE_ROWS, bool FIRST, bool LAST>
__device__ __forceinline__ void scan_strip(Wave &w)
{
    const int r = w.lane;
    const uint8_t *tp = w.lds + G::OFF_TILE + r * G::TILE_PITCH;
    const float *edge = reinterpret_cast<const float *>(w.lds + G::OFF_EDGE);
    float fresh[G::NOCT];
#pragma unroll
    for (int q = 0; q < G::NOCT; q++) {
        const uint4 cur = *reinterpret_cast<const uint4 *>(tp + 16 * q);
        const uint32_t dq[4] = {h2_diff(cur.x, w.pv.x), h2_diff(cur.y, w.pv.y), h2_diff(cur.z, w.pv.z), h2_diff(cur.w, w.pv.w)};
        const bool first_octet = FIRST && q == 0;
        float e0 = 0.f, e1 = 0.f, e2 = 0.f;
        if (first_octet) {  // pass-1 values of columns 0,1,2 come from the edge chains
            e0 = edge[0 * 64 + r];
            e1 = edge[1 * 64 + r];
            e2 = edge[2 * 64 + r];
        }
#define RPH_STEP(E)                                                              \
    {                                                                            \
        w.hs = hs_step<(E) & 1>(w.hs, dq[(E) >> 1]);                             \
        float in2 = pass1_value<EDGE_ROWS>(w, w.hs);                             \
        if (first_octet && (E) == 4) in2 = e0;                                   \
        if (first_octet && (E) == 5) in2 = e1;                                   \
        if (first_octet && (E) == 6) in2 = e2;                                   \
        if (!(first_octet && (E) < 4)) row_step<E>(w, in2); /* ri = xv - 4 >= 0 */ \
    }
        RPH_STEP(0) RPH_STEP(1) RPH_STEP(2) RPH_STEP(3) RPH_STEP(4)
        // output o = xv - 8 = 8j + 4 right after element 4 of octet j + 1
        fresh[q] = w.sum * 0.125f;  // sample slot G::NOCT * s + q (slot 0 of the image = octet 0 is a dummy)
        RPH_STEP(5) RPH_STEP(6) RPH_STEP(7)
#undef RPH_STEP
        w.pv = cur;
    }
    if (LAST) {
        // columns 508..511: V beyond the image is 0; 508,509,510 come from the edge chains, 511 has a 4-wide window
        const float e3 = edge[3 * 64 + r], e4 = edge[4 * 64 + r], e5 = edge[5 * 64 + r];
        w.hs = hs_sub<0>(w.hs, w.pv.x); row_step<0>(w, e3);
        w.hs = hs_sub<1>(w.hs, w.pv.x); row_step<1>(w, e4);
        w.hs = hs_sub<0>(w.hs, w.pv.y); row_step<2>(w, e5);
        w.hs = hs_sub<1>(w.hs, w.pv.y);
        {
            const float in511 = EDGE_ROWS ? div_small(w.hs * 16.0f, w.row_d, w.row_dinv) : w.hs * 2.0f;
            row_step<3>(w, in511);
        }
        // phase 4, first step: out[508] = (sum - in[504]) / 7   (ring slot of ri = 504 is 0)
        w.sum = w.sum - w.ring[0];
        w.smp_last = w.sum / 7.0f;
    }
#pragma unroll
    for (int i = 0; i < 32 - G::NOCT; i++) w.smp[i] = w.smp[i + G::NOCT];
#pragma unroll
    for (int q = 0; q < G::NOCT; q++) w.smp[32 - G::NOCT + q] = fresh[q];
}

__device__ __forceinline__ void scan_reset(Wave &w, int b)
{
    w.hs = 0.f;
    w.sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) w.ring[i] = 0.f;
    w.pv = make_uint4(0, 0, 0, 0);
    // column-window size of row y = 64 b + lane: 5,6,7,8 | 8 ... | 7,6,5,4
    const int y = 64 * b + w.lane;
    const int lo = y - 3 < 0 ? 0 : y - 3, hi = y + 4 > 511 ? 511 : y + 4;
    const float d = (float)(hi - lo + 1);
    w.row_d = d;
    w.row_dinv = 1.0f / d;
}

// pass-2 column chain over the 64 rows of band b for the sampled columns of one half band.
// First half: sample slots 0..31 = [dummy, j = 0..30] -> lanes j = 0..30 read slot j + 1.
// Second half: slots 0..31 = j = 31..62, slot 32 = j = 63      -> lanes j = 31..63 read slot j - 31.
template <class G>
__device__ __forceinline__ void col_pass(Wave &w, int b, bool second_half)
{
    // transpose through the V tile, which is dead between the last scan of the half band and the next tile build
    float *tb = reinterpret_cast<float *>(w.lds + G::OFF_TILE);
#pragma unroll
    for (int t = 0; t < 32; t++) tb[w.lane * SAMP_PITCH + t] = w.smp[t];
    tb[w.lane * SAMP_PITCH + 32] = w.smp_last;
    wave_lds_fence();
    const bool active = second_half ? (w.lane >= 31) : (w.lane < 31);
    if (active) {
        float *park = reinterpret_cast<float *>(w.lds + G::OFF_PARK_C) + w.lane;
        if (G::PARK) {
            w.csum = park[0];
#pragma unroll
            for (int e = 0; e < 8; e++) w.cring[e] = park[64 * (e + 1)];
        }
        const float *samp = tb + (second_half ? w.lane - 31 : w.lane + 1);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            float in[8];
#pragma unroll
            for (int e = 0; e < 8; e++) in[e] = samp[(8 * u + e) * SAMP_PITCH];
#pragma unroll
            for (int e = 0; e < 8; e++) {  // input row y = 64 b + 8 u + e; (y & 7) == e
                w.csum = w.csum + in[e];
                w.csum = w.csum - w.cring[e];
                w.cring[e] = in[e];
                // y = 8 i + 8 entered: output row o = y - 4 = 8 i + 4, i = 8 b + u - 1; /8 for the window, /64 for the scale
                if (e == 0) w.bnew[u] = w.csum * (0.125f * 0.015625f);
            }
        }
        if (G::PARK) {
            park[0] = w.csum;
#pragma unroll
            for (int e = 0; e < 8; e++) park[64 * (e + 1)] = w.cring[e];
        }
    }
    wave_lds_fence();
}

// edge pre-pass: pass-1 values (x64) of columns 0,1,2,508,509,510 for V rows 64b .. 64b+63
//   (FIRST: also consumes the chain's phase 1 = luma rows 0..3)
template <int CH>
__device__ __forceinline__ void edge_rowvals(const Wave &w, int y, bool live, float (&rv)[6])
{
    // lane = luma row y: R = horizontal clipped window sum of luma (exact), rowval x 64 = (64 R) / window; the numerators are
    // multiples of 64 below 2^17 and the divisors 5, 6, 7, so div_small gives the IEEE quotient
    const int yc = y > 511 ? 511 : y;
    const Px8 pl = load_px8<CH>(w.img + (size_t)yc * w.row_stride);
    const Px8 pr = load_px8<CH>(w.img + (size_t)yc * w.row_stride + 504 * CH);
    float l[8], r[8];
    luma8<CH>(pl, l);
    luma8<CH>(pr, r);
    const float r0 = (((l[0] + l[1]) + l[2]) + l[3]) + l[4];
    const float r1 = r0 + l[5];
    const float r2 = r1 + l[6];
    const float q510 = (((r[3] + r[4]) + r[5]) + r[6]) + r[7];
    const float q509 = q510 + r[2];
    const float q508 = q509 + r[1];
    rv[0] = div_small(r0 * 64.0f, 5.0f, 1.0f / 5.0f);
    rv[1] = div_small(r1 * 64.0f, 6.0f, 1.0f / 6.0f);
    rv[2] = div_small(r2 * 64.0f, 7.0f, 1.0f / 7.0f);
    rv[3] = div_small(q508 * 64.0f, 7.0f, 1.0f / 7.0f);
    rv[4] = div_small(q509 * 64.0f, 6.0f, 1.0f / 6.0f);
    rv[5] = div_small(q510 * 64.0f, 5.0f, 1.0f / 5.0f);
    if (!live) {
#pragma unroll
        for (int k = 0; k < 6; k++) rv[k] = 0.f;
    }
}

template <class G>
__device__ __forceinline__ void edge_prologue(Wave &w)
{
    float *edge = reinterpret_cast<float *>(w.lds + G::OFF_EDGE);
    float rv[6];
    edge_rowvals<G::CH>(w, w.lane & 3, true, rv);
    if (w.lane < 4) {
#pragma unroll
        for (int k = 0; k < 6; k++) edge[k * 64 + w.lane] = rv[k];
    }
    wave_lds_fence();
    w.ecs = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) w.ering[i] = 0.f;
    if (w.lane < 6) {  // phase 1 of the column box: accumulate rows 0..3, no output
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float v = edge[w.lane * 64 + t];
            w.ecs = w.ecs + v;
            w.ering[t] = v;
        }
        if (G::PARK) {
            float *park = reinterpret_cast<float *>(w.lds + G::OFF_PARK_E) + w.lane;
            park[0] = w.ecs;
#pragma unroll
            for (int i = 0; i < 8; i++) park[6 * (i + 1)] = w.ering[i];
        }
    }
    wave_lds_fence();
}

template <class G, int KIND>  // 0 = first band, 1 = middle, 2 = last band
__device__ __forceinline__ void edge_band(Wave &w, int b)
{
    float *edge = reinterpret_cast<float *>(w.lds + G::OFF_EDGE);
    {
        float rv[6];
        const int y = 64 * b + 4 + w.lane;
        edge_rowvals<G::CH>(w, y, y < 512, rv);
#pragma unroll
        for (int k = 0; k < 6; k++) edge[k * 64 + w.lane] = rv[k];
    }
    wave_lds_fence();
    if (w.lane < 6) {
        float *mine = edge + w.lane * 64;
        float *park = reinterpret_cast<float *>(w.lds + G::OFF_PARK_E) + w.lane;
        if (G::PARK) {
            w.ecs = park[0];
#pragma unroll
            for (int i = 0; i < 8; i++) w.ering[i] = park[6 * (i + 1)];
        }
#pragma unroll 1
        for (int u = 0; u < 8; u++) {
            float in[8], out[8];
#pragma unroll
            for (int e = 0; e < 8; e++) in[e] = mine[8 * u + e];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                // chain input row y = 64b + 4 + 8u + e, ring slot (y & 7) = (e + 4) & 7, output row o = y - 4 = 64b + 8u + e
                constexpr int dummy = 0;
                (void)dummy;
                const int slot = (e + 4) & 7;
                const bool add = !(KIND == 2 && u == 7 && e >= 4);  // y >= 512: phase 4, nothing enters
                if (add) w.ecs = w.ecs + in[e];
                w.ecs = w.ecs - w.ering[slot];
                w.ering[slot] = in[e];
                float o;
                if (KIND == 0 && u == 0 && e < 3)
                    o = w.ecs / (float)(5 + e);                       // o = 0,1,2: windows 5,6,7
                else if (KIND == 2 && u == 7 && e >= 4)
                    o = w.ecs / (float)(11 - e);                      // o = 508..511: windows 7,6,5,4
                else
                    o = w.ecs * 0.125f;
                out[e] = o;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) mine[8 * u + e] = out[e];
        }
        if (G::PARK) {
            park[0] = w.ecs;
#pragma unroll
            for (int i = 0; i < 8; i++) park[6 * (i + 1)] = w.ering[i];
        }
    }
    wave_lds_fence();
}

// Low-latency form: where the samples of a half band go instead of into this wave's column chain ([slot 0..64][row 0..63] floats
// of band b in the image's sample scratch; slot k holds sampled column k - 1, slot 0 is the dummy)
template <class G>
__device__ __forceinline__ void store_samples(const Wave &w, float *band_samples, bool second_half)
{
    float *p = band_samples + (second_half ? 32 * 64 : 0) + w.lane;
    asm volatile("" : "+v"(p));  // the 33 store addresses are formed here, from this one pointer (hoisted out of the strip loop they are 66 live VGPRs)
#pragma unroll
    for (int t = 0; t < 32; t++) p[t * 64] = w.smp[t];
    if (second_half) p[32 * 64] = w.smp_last;
}

// LL = false: the band of a wave that walks the whole image (edge chains, column chain and tail carried from band to band).
// LL = true: the band of one of the eight waves that share an image (pdq_fused512_ll_kernel): the edge values of the band are already
// in the wave's edge area, the prefetch stops at the band's end, and the pass-2 row samples go to `samples` for the final chain.
template <class G, int KIND, bool LL = false>
__device__ __forceinline__ void do_band(Wave &w, int b, Px8 (&pre)[8], float *samples = nullptr)
{
    constexpr bool EDGE_ROWS = KIND != 1;
    if (!LL) edge_band<G, KIND>(w, b);
    scan_reset(w, b);
#pragma unroll 1
    for (int s = 0; s < G::NSTRIP; s++) {
#pragma unroll 1
        for (int h = 0; h < G::NH; h++) {
            Row8 L[8];
            // next build step in processing order: (b,s,1) -> (b,s+1,0) -> ... -> (b+1,0,0)
            int nb = b, ns = s, nh = h + 1;
            if (nh == G::NH) {
                nh = 0;
                ns = s + 1;
                if (ns == G::NSTRIP) {
                    ns = 0;
                    nb = b + 1;
                }
            }
            half_luma<G, KIND == 2>(w, b, h, pre, L, LL ? nb == b : nb < 8, nb, ns, nh);
            half_build<G>(w, s, h, L);
        }
        if (s == 0)
            scan_strip<G, EDGE_ROWS, true, false>(w);
        else if (s == G::NSTRIP - 1)
            scan_strip<G, EDGE_ROWS, false, true>(w);
        else
            scan_strip<G, EDGE_ROWS, false, false>(w);
        wave_lds_fence();
        if (LL) {
            if (s == G::NSTRIP / 2 - 1) store_samples<G>(w, samples, false);
            if (s == G::NSTRIP - 1) store_samples<G>(w, samples, true);
        } else {
            if (s == G::NSTRIP / 2 - 1) col_pass<G>(w, b, false);
            if (s == G::NSTRIP - 1) col_pass<G>(w, b, true);
        }
        wave_lds_fence();
    }
    if (LL) return;
    // band b completed decimated rows i = 8 b + u - 1 (u = 0 of band 0 is a dummy): feed them to the tail in order
#pragma unroll
    for (int u = 0; u < 8; u++)
        if (KIND != 0 || u > 0) rph::tail_row(w.tail, w.bnew[u], 8 * b + u - 1, w.lane, w.want_quality);
}

template <class G>
__global__ void __launch_bounds__(64, G::WAVES_PER_SIMD) pdq_fused512_kernel(const uint8_t *__restrict__ px, uint32_t n, size_t row_stride,
                                                          size_t image_stride, uint8_t *hash, float *quality, float *coeffs,
                                                          uint8_t *dihedral, uint8_t *valid)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[G::LDS_BYTES];
    const uint32_t img = blockIdx.x;
    Wave w;
    w.lds = lds;
    w.img = px + (size_t)img * image_stride;
    w.row_stride = row_stride;
    w.rs32 = (uint32_t)row_stride;
    w.lane = threadIdx.x;
    w.csum = 0.f;
    w.smp_last = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) w.smp[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) w.cring[i] = 0.f;
    if (G::PARK) {
#pragma unroll
        for (int i = 0; i < 9; i++) reinterpret_cast<float *>(lds + G::OFF_PARK_C)[64 * i + w.lane] = 0.f;
    }
    rph::tail_init(w.tail);
    w.want_quality = quality != nullptr;
#pragma unroll
    for (int u = 0; u < 8; u++) w.bnew[u] = 0.f;

    // ---- prologue: luma rows 0..3 become the first band's history (rows -3..-1 are outside: zero)
    {
        for (int j = 0; j < 3; j++)
            for (int t = w.lane; t < 64; t += 64) *reinterpret_cast<uint4 *>(lds + G::OFF_STATE + j * 1024 + t * 16) = make_uint4(0, 0, 0, 0);
        // rows 0..3 x 64 column chunks of 8 px = 256 (row, chunk) slots, 64 per iteration
        uint32_t order = 0;
#pragma unroll 1
        for (int it = 0; it < 4; it++) {
            const int slot = it * 64 + w.lane;
            const int chunk = slot & 63, row = slot >> 6;
            const Px8 p = load_px8<G::CH>(w.img + (size_t)row * row_stride + (size_t)(8 * chunk) * G::CH);
            *reinterpret_cast<uint4 *>(lds + G::OFF_STATE + (3 + row) * 1024 + (8 * chunk) * 2) = row_bits(pack_row<G::CH>(p, false, order));
        }
    }
    edge_prologue<G>(w);

    Px8 pre[8];
    half_issue<G>(w, 0, 0, 0, pre);
    do_band<G, 0>(w, 0, pre);
#pragma unroll 1
    for (int b = 1; b < 7; b++) do_band<G, 1>(w, b, pre);
    do_band<G, 2>(w, 7, pre);

    // pass-2 column chain, phase 4 first step: out[508] = (csum - in[504]) / 7  (ring slot 0), unscale by 64
    if (G::PARK) {
        w.csum = reinterpret_cast<const float *>(lds + G::OFF_PARK_C)[w.lane];
        w.cring[0] = reinterpret_cast<const float *>(lds + G::OFF_PARK_C)[64 + w.lane];
    }
    w.csum = w.csum - w.cring[0];
    rph::tail_row(w.tail, (w.csum / 7.0f) * 0.015625f, 63, w.lane, w.want_quality);

    __syncthreads();
    rph::tail_finish(w.tail, reinterpret_cast<float *>(lds), w.lane, hash + (size_t)img * 32, quality ? quality + img : nullptr,
                  coeffs ? coeffs + (size_t)img * 256 : nullptr, dihedral ? dihedral + (size_t)img * 256 : nullptr);
    if (valid && w.lane == 0) valid[img] = 1;
}

// ---------------------------------------------------------------------------------------------
// Low-latency form: eight waves share an image, one band each.
//
// The kernel above gives an image to one wave: ~0.3 ms from its first byte to its hash, which is fine when thousands of images
// are in flight and is the whole cost when a handful are (the one-image-per-call queue, batcher.cpp).  Here a workgroup of eight
// waves takes an image; wave b runs band b with the same code.  What is sequential across bands is cut out and done cheaply:
//   * the 7 luma rows above a band: every wave recomputes them from the image (7 rows = 1.4 % of the pixels);
//   * the six edge chains (pass-1 column recurrence of the frame columns): all waves compute their band's chain INPUTS into a
//     shared table, then wave b runs the recurrence from row 0 to its own band (2 instructions per row for the part above);
//   * the pass-2 column chain and the tail: the waves leave their row samples in a scratch buffer (133 KB per image, L2
//     resident), and wave 0 runs the chain over all 512 rows at the end.
// Same bits as the other kernels (tests/test_gpu_parity.py forces each).  ~60 us per image, one image per CU at a time.
// ---------------------------------------------------------------------------------------------
constexpr int LL_SAMPLE_FLOATS = 8 * 65 * 64;  // per image: [band][slot][row]
constexpr int LL_EDGE_PITCH = 520;             // chain inputs of rows 0..515 (512..515: nothing enters)

template <class G, int KIND>
__device__ __forceinline__ void ll_band(Wave &w, int b, float *samples)
{
    Px8 pre[8];
    half_issue<G>(w, b, 0, 0, pre);
    do_band<G, KIND, true>(w, b, pre, samples);
}

template <class G>
__global__ void __launch_bounds__(512, 2) pdq_fused512_ll_kernel(const uint8_t *__restrict__ px, uint32_t n, size_t row_stride, size_t image_stride,
                                                                 float *__restrict__ sample_scratch, uint8_t *hash, float *quality, float *coeffs,
                                                                 uint8_t *dihedral, uint8_t *valid)
{
    static_assert(G::NH == 1, "the low-latency kernel is built on the 64-px strip geometry");
    __shared__ __attribute__((aligned(16))) uint8_t lds[8 * G::LDS_BYTES + 6 * LL_EDGE_PITCH * 4];
    const uint32_t img = blockIdx.x;
    const int band = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and known to the compiler as such (scalar branches, scalar addresses)
    Wave w;
    w.lds = lds + band * G::LDS_BYTES;
    w.img = px + (size_t)img * image_stride;
    w.row_stride = row_stride;
    w.rs32 = (uint32_t)row_stride;
    w.lane = threadIdx.x & 63;
    w.smp_last = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) w.smp[i] = 0.f;
    w.want_quality = quality != nullptr;
    float *edge_all = reinterpret_cast<float *>(lds + 8 * G::LDS_BYTES);
    float *samples = sample_scratch + (size_t)img * LL_SAMPLE_FLOATS;

    // ---- chain inputs of this band's rows (the chains' phase-1 rows 0..3 are ordinary inputs here)
    {
        float rv[6];
        edge_rowvals<G::CH>(w, 64 * band + w.lane, true, rv);
#pragma unroll
        for (int k = 0; k < 6; k++) edge_all[k * LL_EDGE_PITCH + 64 * band + w.lane] = rv[k];
        if (band == 7 && w.lane < 8) {
#pragma unroll
            for (int k = 0; k < 6; k++) edge_all[k * LL_EDGE_PITCH + 512 + w.lane] = 0.f;
        }
    }
    // ---- the 7 luma rows above the band's first build step: rows 64 b - 3 .. 64 b + 3 (band 0: rows -3..-1 are outside: zero)
    {
        uint32_t order = 0;
#pragma unroll 1
        for (int it = 0; it < 7; it++) {
            const int row = 64 * band - 3 + it;  // state row `it`; every lane takes one 8-px chunk
            uint4 bits = make_uint4(0, 0, 0, 0);
            if (row >= 0) bits = row_bits(pack_row<G::CH>(load_px8<G::CH>(w.img + (size_t)row * row_stride + (size_t)(8 * w.lane) * G::CH), false, order));
            *reinterpret_cast<uint4 *>(w.lds + G::OFF_STATE + it * 1024 + (8 * w.lane) * 2) = bits;
        }
    }
    __syncthreads();
    // ---- edge chains: lanes 0..5 of every wave run the recurrence from row 0 to the end of their own band
    if (w.lane < 6) {
        const float *in = edge_all + w.lane * LL_EDGE_PITCH;
        float *mine = reinterpret_cast<float *>(w.lds + G::OFF_EDGE) + w.lane * 64;
        float ecs = 0.f, ering[8];
#pragma unroll
        for (int t = 0; t < 8; t++) ering[t] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++) {  // phase 1 of the column box: rows 0..3 accumulate, no output
            ecs = ecs + in[t];
            ering[t] = in[t];
        }
#pragma unroll 1
        for (int bb = 0; bb < band; bb++)  // bands above: only the state advances (row y = 64 bb + 4 + 8 u + e, ring slot y & 7)
#pragma unroll 1
            for (int u = 0; u < 8; u++) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = in[64 * bb + 4 + 8 * u + e];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    ecs = ecs + v[e];
                    ecs = ecs - ering[(e + 4) & 7];
                    ering[(e + 4) & 7] = v[e];
                }
            }
#pragma unroll 1
        for (int u = 0; u < 8; u++) {  // own band: outputs for rows 64 band + 8 u + e (edge_band with the band index a run-time value)
            float v[8], out[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = in[64 * band + 4 + 8 * u + e];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const bool past_end = band == 7 && u == 7 && e >= 4;  // y >= 512: phase 4, nothing enters
                if (!past_end) ecs = ecs + v[e];
                ecs = ecs - ering[(e + 4) & 7];
                ering[(e + 4) & 7] = v[e];
                float o = ecs * 0.125f;
                if (band == 0 && u == 0 && e < 3) o = ecs / (float)(5 + e);   // o = 0,1,2: windows 5,6,7
                if (past_end) o = ecs / (float)(11 - e);                       // o = 508..511: windows 7,6,5,4
                out[e] = o;
            }
#pragma unroll
            for (int e = 0; e < 8; e++) mine[8 * u + e] = out[e];
        }
    }
    wave_lds_fence();

    // ---- the band itself
    float *band_samples = samples + (size_t)band * (65 * 64);
    if (band == 0)
        ll_band<G, 0>(w, 0, band_samples);
    else if (band == 7)
        ll_band<G, 2>(w, 7, band_samples);
    else
        ll_band<G, 1>(w, band, band_samples);
    __syncthreads();  // every wave's samples are written (the barrier waits for their stores) before wave 0 reads them
    if (band != 0) return;

    // ---- pass-2 column chain over all 512 rows (lane j = sampled column j reads slot j + 1) and the streaming tail
    w.csum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) w.cring[i] = 0.f;
    rph::tail_init(w.tail);
    const float *col = samples + (size_t)(w.lane + 1) * 64;
#pragma unroll 1
    for (int bb = 0; bb < 8; bb++) {
        const float *cb = col + (size_t)bb * (65 * 64);
#pragma unroll 1
        for (int u = 0; u < 8; u++) {
            const float4 lo = *reinterpret_cast<const float4 *>(cb + 8 * u), hi = *reinterpret_cast<const float4 *>(cb + 8 * u + 4);
            const float in[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float bnew = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) {  // input row y = 64 bb + 8 u + e; (y & 7) == e
                w.csum = w.csum + in[e];
                w.csum = w.csum - w.cring[e];
                w.cring[e] = in[e];
                // y = 8 i + 8 entered: output row o = y - 4 = 8 i + 4, i = 8 bb + u - 1; /8 for the window, /64 for the scale
                if (e == 0) bnew = w.csum * (0.125f * 0.015625f);
            }
            if (bb != 0 || u != 0) rph::tail_row(w.tail, bnew, 8 * bb + u - 1, w.lane, w.want_quality);
        }
    }
    w.csum = w.csum - w.cring[0];
    rph::tail_row(w.tail, (w.csum / 7.0f) * 0.015625f, 63, w.lane, w.want_quality);
    wave_lds_fence();
    rph::tail_finish(w.tail, reinterpret_cast<float *>(w.lds), w.lane, hash + (size_t)img * 32, quality ? quality + img : nullptr,
                     coeffs ? coeffs + (size_t)img * 256 : nullptr, dihedral ? dihedral + (size_t)img * 256 : nullptr);
    if (valid && w.lane == 0) valid[img] = 1;
}

}  // namespace

// Low-latency launch: the sample scratch belongs to the caller's stream (ctx->ll_scratch), so launches on different streams run side
// by side; called with ctx->mu held.
int rph_launch_pdq_fused512_ll(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, size_t row_stride, size_t image_stride, uint8_t *d_hash,
                               float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    const uint32_t chunk = n < 256 ? n : 256;  // one image per CU is all that runs at once
    const size_t need = (size_t)chunk * LL_SAMPLE_FLOATS * sizeof(float);
    if (ctx->ll_scratch.size() > 256 && ctx->ll_scratch.find(stream) == ctx->ll_scratch.end()) {  // streams come and go: start over
        RPH_HIP_CHECK(hipDeviceSynchronize());
        for (auto &kv : ctx->ll_scratch) (void)hipFree(kv.second.p);
        ctx->ll_scratch.clear();
    }
    rph_ctx::LLScratch &sc = ctx->ll_scratch[stream];
    if (sc.bytes < need) {
        if (sc.p) {
            RPH_HIP_CHECK(hipStreamSynchronize(stream));  // only this stream's kernels use it
            RPH_HIP_CHECK(hipFree(sc.p));
        }
        sc.p = nullptr;
        sc.bytes = 0;
        RPH_HIP_CHECK(hipMalloc((void **)&sc.p, need));
        sc.bytes = need;
    }
    for (uint32_t first = 0; first < n; first += chunk) {
        const uint32_t m = (n - first) < chunk ? (n - first) : chunk;
        hipLaunchKernelGGL((pdq_fused512_ll_kernel<Geo<64, 3, false>>), dim3(m), dim3(512), 0, stream, d_px + (size_t)first * image_stride, m, row_stride, image_stride,
                           sc.p, d_hash + (size_t)first * 32, d_quality ? d_quality + first : nullptr, d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr,
                           d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr, d_valid ? d_valid + first : nullptr);
        RPH_HIP_CHECK(hipGetLastError());
    }
    return RPH_OK;
}

int rph_launch_pdq_fused512(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, size_t row_stride, size_t image_stride,
                            uint8_t *d_hash, float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid,
                            hipStream_t stream, uint32_t channels)
{
    if (n == 0) return RPH_OK;
    if (channels == 1) {  // Luma8: the one-wave-per-image kernel at every batch size (a third of the bytes, no luma arithmetic)
        hipLaunchKernelGGL((pdq_fused512_kernel<Geo<64, 1>>), dim3(n), dim3(64), 0, stream, d_px, n, row_stride, image_stride, d_hash, d_quality, d_coeffs, d_dihedral,
                           d_valid);
        RPH_HIP_CHECK(hipGetLastError());
        return RPH_OK;
    }
    // 3 = the low-latency kernel, 4 (default) = automatic: below ~3 images per CU the eight-waves-per-image kernel finishes sooner
    // (~60 us against ~300 us), above it the one-wave-per-image kernel has the throughput
    if (ctx->pdq_kernel == 3 || ((ctx->pdq_kernel == 4 || ctx->pdq_kernel == 6) && n < 768))
        return rph_launch_pdq_fused512_ll(ctx, d_px, n, row_stride, image_stride, d_hash, d_quality, d_coeffs, d_dihedral, d_valid, stream);
    if (ctx->pdq_kernel == 2)
        hipLaunchKernelGGL(pdq_fused512_kernel<Geo<128>>, dim3(n), dim3(64), 0, stream, d_px, n, row_stride, image_stride, d_hash,
                           d_quality, d_coeffs, d_dihedral, d_valid);
    else
        hipLaunchKernelGGL(pdq_fused512_kernel<Geo<64>>, dim3(n), dim3(64), 0, stream, d_px, n, row_stride, image_stride, d_hash,
                           d_quality, d_coeffs, d_dihedral, d_valid);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}
