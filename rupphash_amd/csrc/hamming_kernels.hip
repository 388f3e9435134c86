// hamming_kernels.hip -- tiled 256-bit XOR-popcount sweep for gfx950.
//
// Replaces the candidate generation of the reference's groupers:
//   find_groups            /root/reference/src/hamminghash.rs:196-243 (R<=1 MIH probing)
//   group_files_generic    /root/reference/src/scanner.rs:1678-1776   (8 variants, R<=3 probing)
// with an exhaustive sweep over the upper-triangular tile pairs, which is exact
// for every threshold (the reference's probing is exact only up to 31 / 63).
//
// Layout: hashes are 32 B = 8 little-endian dwords, row-major, as in the
// reference's Vec<[u8;32]>.  A workgroup (128 threads = 2 waves) owns one
// (row tile, column tile) pair of T = 1024 files each:
//   - the column tile (32 KiB) is staged once into LDS; every lane reads the
//     same column hash at the same time, i.e. a broadcast ds_read_b128
//   - each lane keeps R = 8 row hashes in VGPRs (first PW dwords for the fast
//     path), so one column read feeds 64 x 8 pair evaluations
//   - per pair the fast path is PW x (v_xor_b32 + v_bcnt_u32_b32): a partial
//     distance over the first PW dwords is a lower bound of the distance, so
//     a pair is only completed (remaining dwords, exact distance, flags) when
//     some lane's partial distance is already <= threshold.  PW is chosen from
//     the threshold so that unrelated (binomial) pairs fail the partial test
//     with ~5 sigma; PW = 8 is the plain full-width sweep.
//   - hits are appended to a global edge list through one atomic cursor.
// The binding roof is integer VALU (2*PW lane-ops per pair), not HBM: the sweep
// moves 64/T bytes per pair.
//
// Default formulation of the same fast path: hamming_mfma_kernel<Format, PW>.  With bits encoded as +-1, a slice of the
// XOR-popcount is a dot product (dot = bits - 2 d), exact in the accumulator, so the matrix pipe evaluates 32 x 32 pairs per
// instruction and the VALU is left with one max-reduction per tile:
//   FmtFp4 (default)  e2m1 codes 0x2 / 0xA, v_mfma_scale_f32_32x32x64_f8f6f4 at unit scales: a 64-bit slice in 32 clk
//   FmtI8             int8 bytes, v_mfma_i32_32x32x32_i8: a 32-bit slice in 32 clk
// (against 64 lanes / 6 clk for v_xor + v_bcnt, a half-rate VALU op on gfx950 -- tools/valu_rate.hip).  Row fragments stay in
// VGPRs, column chunks are expanded once through a 256-entry LUT into LDS.  Candidates (partial distance <= threshold) go through
// the same exact completion as the VALU kernel.  64-bit hashes use the fp4 format with the whole hash as one slice.
#include <hipcub/hipcub.hpp>

#include "rph_internal.h"

namespace {

constexpr int T_FILES = 1024;   // files per tile (rows and columns)
constexpr int BLOCK = 128;      // threads per workgroup
constexpr int R = T_FILES / BLOCK;  // row hashes per lane = 8
// HIP requires gridDim.x * blockDim.x < 2^32 (a larger grid is silently truncated modulo 2^32): 2^22 blocks x <= 1024 threads
constexpr unsigned long long MAX_GRID = 1ull << 22;

struct SweepArgs {
    const uint32_t *rows;      // [n][n_variants][8] dwords (n_variants == 1: may alias cols)
    const uint32_t *cols;      // [n][8] dwords
    const uint8_t *low_conf;   // [n] or nullptr
    const uint8_t *has_features;  // [n] or nullptr: files with 0 only own variant 0 (scanner.rs:1624-1627)
    unsigned long long n;
    uint32_t n_variants;
    uint32_t threshold;
    uint32_t mih_tol;          // find_groups chunk tolerance: 1 if threshold/16 >= 1 else 0
    uint32_t part, nparts;
    unsigned long long n_tile_pairs;
    unsigned long long block0;  // first tile-pair slot of this launch (a launch carries at most MAX_GRID blocks)
    uint32_t seg_tiles;         // int8 MFMA kernel: column tiles per block (a block sweeps one row tile against a segment of column tiles)
    uint32_t n_tiles;
    rph_edge *edges;
    unsigned long long cap;
    unsigned long long *count;
    // popcount-sorted {0,1} formulation (FmtFp4ZO): rows == cols is the hash array sorted by the popcount of its prefix,
    // pcs[k] that popcount (ascending), perm[k] the index of sorted hash k in the caller's array
    const uint32_t *perm;
    const uint16_t *pcs;
};

__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) { return (uint32_t)__builtin_popcount(x) + acc; }

// linear index of the upper-triangular pair -> (I, J), I <= J, row-major in I
__device__ __forceinline__ void tile_pair(unsigned long long p, uint32_t nt, uint32_t &I, uint32_t &J)
{
    const double b = 2.0 * (double)nt + 1.0;
    double disc = b * b - 8.0 * (double)p;
    long long i = (long long)floor((b - sqrt(disc)) * 0.5);
    if (i < 0) i = 0;
    if (i >= (long long)nt) i = nt - 1;
    auto off = [&](long long ii) { return (unsigned long long)ii * nt - (unsigned long long)(ii * (ii - 1) / 2); };
    while (i > 0 && off(i) > p) --i;
    while (i + 1 < (long long)nt && off(i + 1) <= p) ++i;
    I = (uint32_t)i;
    J = (uint32_t)(i + (long long)(p - off(i)));
}

// Segment enumeration of the int8 MFMA kernel: row tile I is swept against segments of S consecutive column tiles
// [J0, J0 + S), J0 = I, I + S, ... ; blocks are numbered row-major in I.  G(M) = sum_{m=1..M} ceil(m / S) blocks cover the
// last M row tiles, so f(I) = G(nt) - G(nt - I) blocks precede row tile I.  (Python mirror: rupphash_amd/dist.py.)
__host__ __device__ __forceinline__ unsigned long long seg_G(unsigned long long M, unsigned long long S)
{
    const unsigned long long q = M / S, r = M % S;
    return S * q * (q + 1) / 2 + (q + 1) * r;
}
__device__ __forceinline__ void seg_block(unsigned long long b, uint32_t nt, uint32_t S, uint32_t &I, uint32_t &J0)
{
    const unsigned long long total = seg_G(nt, S);
    auto f = [&](long long ii) { return total - seg_G((unsigned long long)(nt - ii), S); };
    const double N = (double)nt + 0.5 * (double)S;
    double disc = N * N - 2.0 * (double)S * (double)b;
    disc = disc > 0.0 ? disc : 0.0;
    long long i = (long long)(N - sqrt(disc));
    if (i < 0) i = 0;
    if (i >= (long long)nt) i = nt - 1;
    while (i > 0 && f(i) > b) --i;
    while (i + 1 < (long long)nt && f(i + 1) <= b) ++i;
    I = (uint32_t)i;
    J0 = (uint32_t)(i + (long long)(b - f(i)) * S);
}

// Slow path for one (row, column) pair whose partial distance passed: exact distance,
// owner / limit rules, find_groups reachability flags, append.  In two steps so that a drain can issue the first-half loads of
// several pairs before it looks at any of them.
// Step 1: first half of the two hashes.  Its distance is a lower bound, and most pairs a candidate drags in (the rest of its lane's
// column blocks and rows) fail here, at half the bytes.
__device__ __forceinline__ uint32_t first_half_distance(const uint32_t *__restrict__ rowp, const uint32_t *colp, uint32_t x[4])
{
    const uint4 r0 = *reinterpret_cast<const uint4 *>(rowp), c0 = *reinterpret_cast<const uint4 *>(colp);
    x[0] = r0.x ^ c0.x; x[1] = r0.y ^ c0.y; x[2] = r0.z ^ c0.z; x[3] = r0.w ^ c0.w;
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) d += (uint32_t)__builtin_popcount(x[w]);
    return d;
}

// Step 2, for a pair whose first-half distance d (differences x0) is within the threshold.
__device__ __forceinline__ void complete_rest(const SweepArgs &a, const uint32_t *__restrict__ rowp, const uint32_t *colp, const uint32_t x0[4],
                                              uint32_t d, unsigned long long owner, unsigned long long col, uint32_t variant)
{
    uint32_t x[8];
    {
#pragma unroll
        for (int w = 0; w < 4; w++) x[w] = x0[w];
        const uint4 r1 = *reinterpret_cast<const uint4 *>(rowp + 4), c1 = *reinterpret_cast<const uint4 *>(colp + 4);
        x[4] = r1.x ^ c1.x; x[5] = r1.y ^ c1.y; x[6] = r1.z ^ c1.z; x[7] = r1.w ^ c1.w;
#pragma unroll
        for (int w = 4; w < 8; w++) d += (uint32_t)__builtin_popcount(x[w]);
    }
    if (variant > 0 && a.has_features && !a.has_features[owner]) return;
    if (col >= a.n) return;
    if (col <= owner) return;  // i < j only (scanner.rs:1716 `cand_idx <= i`, hamminghash.rs:216 `dense_idx == i`)
    uint32_t limit = a.threshold;
    if (a.low_conf && (a.low_conf[owner] | a.low_conf[col])) limit = 0;  // scanner.rs:1699,1721
    if (d > limit) return;
    // find_groups reachability (hamminghash.rs:206-238): first chunk k (ascending) whose 16-bit
    // difference has popcount <= tol; slot 0 = exact bucket, 1 + b = flip of bit b.
    uint32_t flags = (variant << RPH_EDGE_VARIANT_SHIFT) & RPH_EDGE_VARIANT_MASK;
#pragma unroll
    for (int k = 15; k >= 0; k--) {  // descending so the smallest qualifying k wins
        const uint32_t c16 = (x[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu;
        const uint32_t pc = (uint32_t)__builtin_popcount(c16);
        if (pc <= a.mih_tol) {
            const uint32_t slot = pc == 0 ? 0u : 1u + (uint32_t)__builtin_ctz(c16);
            flags = (flags & RPH_EDGE_VARIANT_MASK) | RPH_EDGE_MIH_R1 | ((uint32_t)k << 5) | slot;
        }
    }
    const unsigned long long at = atomicAdd(a.count, 1ull);
    if (at < a.cap) {
        rph_edge e;
        e.i = (uint32_t)owner;
        e.j = (uint32_t)col;
        if (a.perm) {  // sorted sweep: back to the caller's indices, i < j (distance and probe key are symmetric)
            const uint32_t pi = a.perm[owner], pj = a.perm[col];
            e.i = pi < pj ? pi : pj;
            e.j = pi < pj ? pj : pi;
        }
        e.d = (uint16_t)d;
        e.flags = (uint16_t)flags;
        a.edges[at] = e;
    }
}

__device__ __forceinline__ void complete_pair(const SweepArgs &a, const uint32_t *__restrict__ rowp, const uint32_t *colp,
                                              unsigned long long owner, unsigned long long col, uint32_t variant)
{
    uint32_t x[4];
    const uint32_t d = first_half_distance(rowp, colp, x);
    if (d <= a.threshold) complete_rest(a, rowp, colp, x, d, owner, col, variant);
}

template <int PW>
__global__ void __launch_bounds__(BLOCK) hamming_sweep_kernel(SweepArgs a)
{
    __shared__ uint4 s_cols[T_FILES * 2];  // [column][2 x uint4] = 32 KiB

    const unsigned long long p = (unsigned long long)a.part + (a.block0 + blockIdx.x) * a.nparts;
    if (p >= a.n_tile_pairs) return;
    uint32_t I, J;
    tile_pair(p, a.n_tiles, I, J);

    const unsigned long long col0 = (unsigned long long)J * T_FILES;
    const unsigned long long row0 = (unsigned long long)I * T_FILES;
    const uint32_t ncols = (uint32_t)((a.n - col0) < (unsigned long long)T_FILES ? (a.n - col0) : T_FILES);

    // stage the column tile: 2048 x 16 B, coalesced
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(a.cols) + col0 * 2;
        for (uint32_t t = threadIdx.x; t < T_FILES * 2; t += BLOCK) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((t >> 1) < ncols) v = g[t];
            s_cols[t] = v;
        }
    }
    __syncthreads();

    const uint32_t nv = a.n_variants;
    for (uint32_t v = 0; v < nv; v++) {
        // this lane's R row hashes: files row0 + r*BLOCK + tid, variant v (first PW dwords kept in VGPRs)
        uint32_t rw[R][PW];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned long long owner = row0 + (unsigned long long)r * BLOCK + threadIdx.x;
            const uint32_t *rp = a.rows + ((owner < a.n ? owner : 0ull) * nv + v) * 8;
            const uint4 lo = *reinterpret_cast<const uint4 *>(rp);
            uint32_t full[8];
            full[0] = lo.x; full[1] = lo.y; full[2] = lo.z; full[3] = lo.w;
            if (PW > 4) {
                const uint4 hi = *reinterpret_cast<const uint4 *>(rp + 4);
                full[4] = hi.x; full[5] = hi.y; full[6] = hi.z; full[7] = hi.w;
            }
#pragma unroll
            for (int w = 0; w < PW; w++) rw[r][w] = full[w];
        }

        for (uint32_t c = 0; c < ncols; c++) {
            const uint4 c0 = s_cols[c * 2];
            uint32_t cw[8];
            cw[0] = c0.x; cw[1] = c0.y; cw[2] = c0.z; cw[3] = c0.w;
            if (PW > 4) {
                const uint4 c1 = s_cols[c * 2 + 1];
                cw[4] = c1.x; cw[5] = c1.y; cw[6] = c1.z; cw[7] = c1.w;
            }
            uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < R; r++) {
                uint32_t d = 0;
#pragma unroll
                for (int w = 0; w < PW; w++) d = bcnt_acc(rw[r][w] ^ cw[w], d);
                dmin = d < dmin ? d : dmin;
            }
            if (dmin <= a.threshold) {  // rare: some row of this lane may pair with column c
                const uint32_t *colp = reinterpret_cast<const uint32_t *>(&s_cols[c * 2]);
#pragma unroll 1
                for (int r = 0; r < R; r++) {
                    const unsigned long long owner = row0 + (unsigned long long)r * BLOCK + threadIdx.x;
                    if (owner < a.n) complete_pair(a, a.rows + (owner * nv + v) * 8, colp, owner, col0 + c, v);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// MFMA formulation of the fast path
// ---------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int MF_BLOCK = 256;   // 4 waves; wave w owns rows 256 w .. 256 w + 255 of the 1024-row tile
#ifndef RPH_SWEEP_NCG
#define RPH_SWEEP_NCG 4
#endif

// byte v -> 8 bytes, byte i = bit i of v ? +1 : -1
__device__ __forceinline__ uint2 expand_byte(uint32_t v)
{
    uint2 r;
    const uint32_t lo = ((v & 0xFu) * 0x00204081u) & 0x01010101u;
    const uint32_t hi = (((v >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
    r.x = 0xFFFFFFFFu ^ (lo * 0xFEu);  // 1 -> 0x01, 0 -> 0xFF, no carries between bytes
    r.y = 0xFFFFFFFFu ^ (hi * 0xFEu);
    return r;
}

__device__ __forceinline__ int max2i(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int max3i(int a, int b, int c)
{
    const int ab = a > b ? a : b;
    return ab > c ? ab : c;
}

// ---- the two matrix formats.  A format says how a packed hash dword becomes MFMA operand bytes (through a 256-entry LUT in LDS),
// how a column's record is laid out in LDS, and which instruction multiplies; everything else (segments, chunks, accumulation
// chains, branch-free screen, candidate queue, exact completion) is the one kernel below.
struct FmtI8 {  // +-1 as int8: v_mfma_i32_32x32x32_i8 evaluates a 32-bit slice of 32 x 32 pairs; one fragment per hash dword
    typedef v16i Acc;
    typedef int Scalar;
    typedef uint2 Lut;  // byte -> 8 int8
    static constexpr int nfrag(int pw) { return pw; }
    static constexpr int pitch(int pw) { return 2 * pw * 16 + 16; }  // [k-half h][dword kb][16 x i8] + pad (conflict-free b128, lane = column)
    static constexpr int chunk(int pw) { return pw <= 4 ? 256 : 128; }
    static constexpr int row_blocks(int pw) { return pw <= 4 ? 8 : 4; }  // A fragments: row_blocks * nfrag * 4 VGPRs
    static constexpr int blocks_per_cu(int) { return 2; }
    static __device__ __forceinline__ Lut lut_entry(uint32_t byte) { return expand_byte(byte); }
    // fragment f of a row for k-half h: the 16 bits [32 f + 16 h, +16) of the row
    static __device__ __forceinline__ v4i a_frag(const Lut *lut, const uint32_t *d, int f, int h)
    {
        const uint32_t hw = (d[f] >> (16 * h)) & 0xFFFFu;
        const uint2 e0 = lut[hw & 0xFFu], e1 = lut[hw >> 8];
        return v4i{(int)e0.x, (int)e0.y, (int)e1.x, (int)e1.y};
    }
    // packed dword kb of a column -> its two 16-byte records (one per k-half)
    template <int PW>
    static __device__ __forceinline__ void expand(const Lut *lut, uint8_t *col, uint32_t kb, uint32_t dw, bool live)
    {
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const uint32_t hw = (dw >> (16 * hh)) & 0xFFFFu;
            uint2 e0 = lut[hw & 0xFFu], e1 = lut[hw >> 8];
            if (!live) e0 = e1 = make_uint2(0, 0);  // zero bytes: dot 0, never a candidate unless every pair is
            *reinterpret_cast<uint4 *>(col + kb * 16 + hh * PW * 16) = make_uint4(e0.x, e0.y, e1.x, e1.y);
        }
    }
    static __device__ __forceinline__ Acc zero() { return Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
    static __device__ __forceinline__ Acc mfma(v4i a, v4i b, Acc c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }
    // the screen compares integer keys: the accumulator itself
    static __device__ __forceinline__ int key(Scalar x) { return x; }
    static __device__ __forceinline__ int thresh_key(int t) { return t; }
};

struct FmtFp4 {  // +-1 as e2m1 (0x2 / 0xA): v_mfma_scale_f32_32x32x64_f8f6f4 at unit scales evaluates a 64-bit slice; f32 accumulation exact
    typedef v16f Acc;
    typedef float Scalar;
    typedef uint32_t Lut;  // byte -> 8 fp4 codes
    static constexpr int nfrag(int pw) { return pw / 2; }
    static constexpr int pitch(int pw) { return pw * 16 + 16; }  // [k-half h][slice][32 x fp4 = 16 B] + pad
    // 128 columns at PW <= 4: with 256 the 54 KB of LDS per workgroup let only two of the three workgroups the register budget
    // allows share a CU (measured: 23.1 -> 24.7 Tpairs/s)
    static constexpr int chunk(int pw) { return pw <= 4 ? 128 : 256; }
    static constexpr int row_blocks(int) { return 8; }
    static constexpr int blocks_per_cu(int pw) { return pw <= 4 ? 3 : 2; }
    static __device__ __forceinline__ Lut lut_entry(uint32_t byte)
    {
        uint32_t e = 0;
        for (int i = 0; i < 8; i++) e |= (((byte >> i) & 1u) ? 0x2u : 0xAu) << (4 * i);
        return e;
    }
    // fragment (slice) f of a row for k-half h: the 32 bits of dword 2 f + h
    static __device__ __forceinline__ v4i a_frag(const Lut *lut, const uint32_t *d, int f, int h)
    {
        const uint32_t dw = h ? d[2 * f + 1] : d[2 * f];
        return v4i{(int)lut[dw & 0xFFu], (int)lut[(dw >> 8) & 0xFFu], (int)lut[(dw >> 16) & 0xFFu], (int)lut[dw >> 24]};
    }
    // packed dword kb of a column -> the 16-byte record of slice kb / 2, k-half kb & 1
    template <int PW>
    static __device__ __forceinline__ void expand(const Lut *lut, uint8_t *col, uint32_t kb, uint32_t dw, bool live)
    {
        uint4 e = make_uint4(lut[dw & 0xFFu], lut[(dw >> 8) & 0xFFu], lut[(dw >> 16) & 0xFFu], lut[dw >> 24]);
        if (!live) e = make_uint4(0, 0, 0, 0);  // fp4 zeros: dot 0, never a candidate unless every pair is
        *reinterpret_cast<uint4 *>(col + ((kb & 1) * (PW / 2) + (kb >> 1)) * 16) = e;
    }
    static __device__ __forceinline__ Acc zero() { return Acc{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }
    static __device__ __forceinline__ Acc mfma(v4i a, v4i b, Acc c)
    {  // cbsz = blgp = 4: fp4 e2m1; E8M0 scale 127 = 2^0
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a[0], a[1], a[2], a[3], 0, 0, 0, 0}, v8i{b[0], b[1], b[2], b[3], 0, 0, 0, 0}, c, 4,
                                                               4, 0, 127, 0, 127);
    }
    // The screen compares integer keys: the f32 bit pattern.  For a positive threshold `x >= t` is the same as
    // `bits(x) >= bits(t)` as signed integers (non-negative floats order like their bit patterns, negative ones are negative
    // integers), and an integer max needs no NaN canonicalisation (a float max compiles to two extra v_max_f32 x, x per tile).
    // t <= 0 (threshold >= 16 PW: every unrelated pair is a candidate anyway) -> INT_MIN: everything goes to the exact completion.
    static __device__ __forceinline__ int key(Scalar x) { return __builtin_bit_cast(int, x); }
    static __device__ __forceinline__ int thresh_key(int t) { return t > 0 ? __builtin_bit_cast(int, (float)t) : (int)0x80000000; }
};

// The same instruction on bits encoded as {0, 1} (e2m1 codes 0x0 / 0x2): dot = popcount(a & b), so
//     d = popcount(a) + popcount(b) - 2 dot      and      d <= threshold  <=>  dot >= (pa + pb - threshold) / 2.
// Three quarters of the products are zero, and the chip -- which runs this kernel against its power limit, not against the matrix
// pipe's cycle count (tools/sweep_loop.hip: in-kernel clock 1.78 GHz on +-1 operands, 1.92-1.98 GHz on {0,1} operands, same cycles)
// -- clocks ~10 % higher.  The price is that the threshold now depends on the pair; with the hashes SORTED by the popcount of their
// prefix (one stable radix sort per call, rph_launch_hamming_sweep) pa is constant over a row block and pb over a column chunk up
// to the few blocks that straddle a step, where the smaller value is used (never misses a pair; the completion is exact).
struct FmtFp4ZO : FmtFp4 {
    static __device__ __forceinline__ Lut lut_entry(uint32_t byte)
    {
        uint32_t e = 0;
        for (int i = 0; i < 8; i++) e |= (((byte >> i) & 1u) ? 0x2u : 0x0u) << (4 * i);
        return e;
    }
};
template <class F>
struct is_zero_one {
    static constexpr bool value = false;
};
template <>
struct is_zero_one<FmtFp4ZO> {
    static constexpr bool value = true;
};

// Exact completion of one u64 pair (impl HammingHash for u64, hamminghash.rs:23-41): distance, i < j, find_groups reachability
// over 8 chunks of 8 bits (first chunk with popcount <= tol; slot 0 = exact bucket, 1 + b = flip of bit b), append.
__device__ __forceinline__ void complete_pair_u64(const SweepArgs &a, unsigned long long owner, unsigned long long col)
{
    if (owner >= a.n || col >= a.n || col <= owner) return;  // i < j only (hamminghash.rs:216)
    const uint2 rv = reinterpret_cast<const uint2 *>(a.rows)[owner], cv = reinterpret_cast<const uint2 *>(a.cols)[col];
    const uint32_t x0 = rv.x ^ cv.x, x1 = rv.y ^ cv.y;
    const uint32_t d = (uint32_t)__builtin_popcount(x0) + (uint32_t)__builtin_popcount(x1);
    if (d > a.threshold) return;
    uint32_t flags = 0;
    for (int k = 7; k >= 0; k--) {
        const uint32_t c8 = ((k < 4 ? x0 : x1) >> ((k & 3) * 8)) & 0xFFu;
        const uint32_t pc = (uint32_t)__builtin_popcount(c8);
        if (pc <= a.mih_tol) flags = RPH_EDGE_MIH_R1 | ((uint32_t)k << 5) | (pc == 0 ? 0u : 1u + (uint32_t)__builtin_ctz(c8));
    }
    const unsigned long long at = atomicAdd(a.count, 1ull);
    if (at < a.cap) {
        rph_edge e;
        e.i = (uint32_t)owner;
        e.j = (uint32_t)col;
        e.d = (uint16_t)d;
        e.flags = (uint16_t)flags;
        a.edges[at] = e;
    }
}

// U64: the hashes are 64-bit (stride 2 dwords, PW = 2: the whole hash is one fp4 MFMA slice) and pairs are completed by complete_pair_u64.
template <class F, int PW, bool U64 = false>
__global__ void __launch_bounds__(MF_BLOCK, F::blocks_per_cu(PW)) hamming_mfma_kernel(SweepArgs a)
{
    constexpr int HS = U64 ? 2 : 8;               // dwords per hash
    constexpr int NF = F::nfrag(PW);              // MFMAs per 32 x 32 tile
    constexpr int PITCH = F::pitch(PW);           // bytes per column record in LDS
    constexpr int CHUNK = F::chunk(PW);           // columns expanded into LDS at a time (two buffers)
    constexpr int QCAP = 128;                     // candidate queue per wave (entries of 8 bytes); overflow falls back to an exhaustive completion
    constexpr int MF_RB = F::row_blocks(PW);      // 32-row blocks per wave and pass
    constexpr int PASS_ROWS = 4 * 32 * MF_RB;     // rows one pass of the 4 waves covers; T_FILES / PASS_ROWS passes per tile
    static_assert(!U64 || PW == 2, "u64 hashes are swept at full width");
    constexpr int NCB = CHUNK / 32;               // 32-column blocks per chunk
    constexpr int NRP = MF_RB / 2;                // row-block pairs per wave and pass
    constexpr int NCG = RPH_SWEEP_NCG < NCB ? RPH_SWEEP_NCG : NCB;  // column-block groups per chunk (cb % NCG): one running maximum per (row-block pair, group)
    static_assert(NCB % NCG == 0 && NRP * NCG <= 32, "candidate bitmap of a chunk is one dword");
    typedef typename F::Acc Acc;
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[2 * CHUNK * PITCH];
    __shared__ typename F::Lut s_lut[256];
    __shared__ uint2 s_q[4][QCAP];      // one queue per wave: filled and drained by the same wave, no barrier needed
    __shared__ uint32_t s_qn[4];

    // one block = row tile I against the column tiles [J0, J0 + seg_tiles): the row fragments are built once, and the wait for
    // the (rare) candidates' exact completion is paid once per segment instead of once per tile
    const unsigned long long p = (unsigned long long)a.part + (a.block0 + blockIdx.x) * a.nparts;
    if (p >= a.n_tile_pairs) return;  // (for this kernel: the number of segment blocks)
    uint32_t I, J;
    seg_block(p, a.n_tiles, a.seg_tiles, I, J);
    const unsigned long long col0 = (unsigned long long)J * T_FILES;
    const unsigned long long row0 = (unsigned long long)I * T_FILES;
    const unsigned long long seg_cols = (unsigned long long)a.seg_tiles * T_FILES;
    const uint32_t ncols = (uint32_t)((a.n - col0) < seg_cols ? (a.n - col0) : seg_cols);  // <= 8192: queue entries keep 16 bits for it

    s_lut[threadIdx.x] = F::lut_entry(threadIdx.x);
    if (threadIdx.x < 4) s_qn[threadIdx.x] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform and known to be: scalar addresses, scalar loads
    const int c32 = lane & 31, h = lane >> 5;
    constexpr bool ZO = is_zero_one<F>::value;
    const int thresh_key = F::thresh_key(32 * PW - 2 * (int)a.threshold);  // +-1 encoding: partial distance <= threshold  <=>  dot >= 32 PW - 2 threshold
    const uint32_t nv = a.n_variants;

    for (uint32_t vp = 0; vp < nv * (T_FILES / PASS_ROWS); vp++) {
        const uint32_t v = vp / (T_FILES / PASS_ROWS);
        auto complete = [&](unsigned long long owner, unsigned long long col) {
            if (U64)
                complete_pair_u64(a, owner, col);
            else
                complete_pair(a, a.rows + (owner * nv + v) * 8, a.cols + col * 8, owner, col, v);
        };
        const uint32_t wrow = (vp % (T_FILES / PASS_ROWS)) * PASS_ROWS + wave * 32 * MF_RB;  // first tile row of this wave in this pass
        // A fragments: this lane holds, for row block rb and fragment f, the operand bytes of row c32 for k-half h
        v4i A[MF_RB][NF];
#pragma unroll
        for (int rb = 0; rb < MF_RB; rb++) {
            const unsigned long long owner = row0 + wrow + 32ull * rb + c32;
            const uint32_t *rp = a.rows + ((owner < a.n ? owner : 0ull) * nv + v) * HS;
            uint32_t d[8];
            if (U64) {
                const uint2 t = *reinterpret_cast<const uint2 *>(rp);
                d[0] = t.x; d[1] = t.y;
            } else {
                const uint4 lo = *reinterpret_cast<const uint4 *>(rp);
                d[0] = lo.x; d[1] = lo.y; d[2] = lo.z; d[3] = lo.w;
            }
            if (PW > 4) {
                const uint4 hi = *reinterpret_cast<const uint4 *>(rp + 4);
                d[4] = hi.x; d[5] = hi.y; d[6] = hi.z; d[7] = hi.w;
            }
#pragma unroll
            for (int f = 0; f < NF; f++) A[rb][f] = F::a_frag(s_lut, d, f, h);
        }

        // {0,1} encoding: smallest prefix popcount of each row-block pair (ascending order: its first row; rows past the end are copies of hash 0)
        int pa_min[NRP];
#pragma unroll
        for (int p = 0; p < NRP; p++) {
            const unsigned long long r_first = row0 + wrow + 64ull * p;
            pa_min[p] = ZO ? (int)a.pcs[(r_first + 63 < a.n) ? r_first : 0ull] : 0;
        }

        // column chunks are double buffered: the packed dwords of chunk i + 1 are fetched into registers while chunk i
        // is swept and expanded into the other LDS buffer afterwards, so global latency never sits between two chunks
        constexpr int PER_THREAD = (CHUNK * PW + MF_BLOCK - 1) / MF_BLOCK;
        uint32_t pre[PER_THREAD];
        auto fetch = [&](uint32_t cbase) {
#pragma unroll
            for (int q = 0; q < PER_THREAD; q++) {
                const uint32_t t = threadIdx.x + q * MF_BLOCK;
                const uint32_t col = t / PW, kb = t % PW;
                pre[q] = (t < CHUNK * PW && cbase + col < ncols) ? a.cols[(col0 + cbase + col) * HS + kb] : 0u;
            }
        };
        auto expand = [&](uint32_t cbase, uint8_t *buf) {
#pragma unroll
            for (int q = 0; q < PER_THREAD; q++) {
                const uint32_t t = threadIdx.x + q * MF_BLOCK;
                if (t >= CHUNK * PW) continue;
                const uint32_t col = t / PW, kb = t % PW;
                F::template expand<PW>(s_lut, buf + col * PITCH, kb, pre[q], cbase + col < ncols);
            }
        };
        fetch(0);
        int which = 0;
        uint32_t undrained = 0;
        for (uint32_t cbase = 0; cbase < ncols; cbase += CHUNK, which ^= 1) {
            uint8_t *s_b = s_buf + which * (CHUNK * PITCH);
            const int pb_min = ZO ? (int)a.pcs[col0 + cbase] : 0;  // smallest prefix popcount of the chunk's columns; issued here, used after the MFMA loop
            expand(cbase, s_b);
            __syncthreads();  // chunk visible (and the queue reset of the previous chunk)
            if (cbase + CHUNK < ncols) fetch(cbase + CHUNK);

            // ---- fast path: MFMA + VALU screen only.  No global memory operation lives in this loop (candidates go to
            // an LDS queue), so the compiler never has to drain vmcnt here and the prefetch above stays in flight.
            // The screen keeps, per row-block pair, the running maximum of the lane's accumulators over ALL column blocks of the
            // chunk: 16 three-input maxima per pair of tiles and nothing else (the compare, select and or of a per-tile test cost
            // another 4 VALU instructions per 4 MFMAs, and the SIMD's issue port is what this loop is short of).  A candidate is
            // then known up to its column block; the (rare) completion examines the lane's column in every block of the chunk.
            int runmax[NCG][NRP];
#pragma unroll
            for (int g = 0; g < NCG; g++)
#pragma unroll
                for (int p = 0; p < NRP; p++) runmax[g][p] = (int)0x80000000;
#pragma unroll 1
            for (int cb0 = 0; cb0 < NCB; cb0 += NCG) {
                if (cbase + cb0 * 32 >= ncols) break;
#pragma unroll
              for (int g = 0; g < NCG; g++) {
                const int cb = cb0 + g;  // (columns past the end of the segment were expanded as zeros: dot 0)
                v4i B[NF];
                const uint8_t *bp = s_b + (cb * 32 + c32) * PITCH + h * NF * 16;
#pragma unroll
                for (int f = 0; f < NF; f++) B[f] = *reinterpret_cast<const v4i *>(bp + f * 16);
                // max of the 32 accumulators of the two tiles and the running maximum in 16 instructions: a tree of three-input
                // maxima (depth 4).  A linear chain of dependent VALU instructions issues at ~9 clk each from one wave
                // (tools/valu_dep.hip), independent ones at ~5.6.
                auto max33 = [&](const Acc &x, const Acc &y, int prev) {
                    auto K = [](typename F::Scalar v) { return F::key(v); };
                    const int t0 = max3i(K(x[0]), K(x[1]), K(x[2])), t1 = max3i(K(x[3]), K(x[4]), K(x[5])), t2 = max3i(K(x[6]), K(x[7]), K(x[8]));
                    const int t3 = max3i(K(x[9]), K(x[10]), K(x[11])), t4 = max3i(K(x[12]), K(x[13]), K(x[14]));
                    const int t5 = max3i(K(y[0]), K(y[1]), K(y[2])), t6 = max3i(K(y[3]), K(y[4]), K(y[5])), t7 = max3i(K(y[6]), K(y[7]), K(y[8]));
                    const int t8 = max3i(K(y[9]), K(y[10]), K(y[11])), t9 = max3i(K(y[12]), K(y[13]), K(y[14]));
                    const int u0 = max3i(t0, t1, t2), u1 = max3i(t3, t4, K(x[15])), u2 = max3i(t5, t6, t7), u3 = max3i(t8, t9, K(y[15]));
                    return max3i(max3i(u0, u1, u2), u3, prev);
                };
                // Two independent accumulation chains are interleaved (a dependent MFMA issues every ~55 clk, an independent one
                // every 32: tools/mfma_rate.hip).  No branch, compare or select in here: a branch, however rare, stalls this
                // wave's MFMA issue and, through the chunk barriers, its three block mates.
#pragma unroll
                for (int rb = 0; rb < MF_RB; rb += 2) {
                    Acc acc0 = F::zero(), acc1 = F::zero();
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        acc0 = F::mfma(A[rb][f], B[f], acc0);
                        acc1 = F::mfma(A[rb + 1][f], B[f], acc1);
                    }
                    runmax[g][rb / 2] = max33(acc0, acc1, runmax[g][rb / 2]);
                }
              }
            }
            uint32_t cand = 0;  // bit g * NRP + p: this lane saw a candidate among its pairs of row blocks 2p, 2p + 1 in a column block cb = g (mod NCG) of the chunk
            if (ZO) {
                // {0,1} encoding on popcount-sorted hashes: dot >= ceil((pa + pb - threshold) / 2) with the smallest pa of the 64 rows and
                // the smallest pb of the chunk's columns
#pragma unroll
                for (int p = 0; p < NRP; p++) {
                    const int need = F::thresh_key((pa_min[p] + pb_min - (int)a.threshold + 1) >> 1);  // arithmetic shift = floor: ceil(x / 2) for any sign
#pragma unroll
                    for (int g = 0; g < NCG; g++) cand |= (runmax[g][p] >= need) ? (1u << (g * NRP + p)) : 0u;
                }
            } else {
#pragma unroll
                for (int g = 0; g < NCG; g++)
#pragma unroll
                    for (int p = 0; p < NRP; p++) cand |= (runmax[g][p] >= thresh_key) ? (1u << (g * NRP + p)) : 0u;
            }

            // ---- lanes with candidates append (bitmap, column) to the wave's queue: slots by ballot rank, no atomics
            // (wave-local: the LDS executes one wave's operations in order, so no barrier is needed between push, read and reset)
            asm volatile("" ::: "memory");
            const unsigned long long vote = __builtin_amdgcn_ballot_w64(cand != 0);
            uint32_t nq = s_qn[wave];
            if (vote != 0) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(vote >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vote, 0u));
                const uint32_t slot = nq + rank;
                if (cand != 0 && slot < QCAP) s_q[wave][slot] = make_uint2(cand, ((uint32_t)h << 16) | (cbase + c32));
                nq += (uint32_t)__builtin_popcountll(vote);
                if (lane == 0) s_qn[wave] = nq;
                asm volatile("" ::: "memory");
            }
            // ---- complete the queued candidates exactly.  A drain costs a global-memory round trip whatever it holds, so it waits
            // until a full wave of entries is queued or the pass ends; `undrained` is the first column of the segment whose
            // candidates may still sit in the queue.
            const uint32_t cend = (cbase + CHUNK) < ncols ? (cbase + CHUNK) : ncols;
            if (nq == 0) {
                undrained = cend;
            } else if (nq >= 64 || cend == ncols) {
                if (nq <= QCAP) {
                    // Each half-wave takes U entries per round; its 32 lanes share the pairs of an entry: for a set bit, the queued
                    // lane's column in the NCB / NCG column blocks of the bit's group against its 32 rows (C/D layout of the two tiles:
                    // row = (r & 3) + 8 (r >> 2) + 4 h).  The first-half loads of a round (2 U entries, first set bit each) are issued
                    // together and looked at afterwards: a drain is a chain of global-memory round trips, and this is what shortens it.
                    // Further set bits of an entry (rare: two candidates of one lane in one chunk) follow one by one.
                    constexpr int U = 2, IPB = NCB / NCG;
                    auto item_of = [&](const uint2 e, uint32_t bit, uint32_t k, unsigned long long &owner, unsigned long long &col) -> bool {
                        const uint32_t r = lane & 31u, eh = e.y >> 16, rb = 2u * (bit % NRP), cb = k * NCG + bit / NRP;
                        const uint32_t cidx = (e.y & 0xFFFFu) + 32u * cb;
                        col = col0 + cidx;
                        owner = row0 + wrow + 32u * (rb + (r >> 4)) + (r & 3u) + 8u * ((r >> 2) & 3u) + 4u * eh;
                        return owner < a.n && cidx < ncols;
                    };
                    for (uint32_t t0 = 0; t0 < nq; t0 += 2 * U) {
                        uint2 e[U];
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const uint32_t t = t0 + 2u * u + (uint32_t)(lane >> 5);
                            e[u] = s_q[wave][t < nq ? t : 0u];  // same address within a half-wave: a broadcast read
                            if (t >= nq) e[u].x = 0;
                        }
                        if (!U64) {
                            uint32_t x[U][IPB][4], d[U][IPB];
#pragma unroll
                            for (int u = 0; u < U; u++)
#pragma unroll
                                for (int k = 0; k < IPB; k++) {
                                    unsigned long long owner, col;
                                    const bool ok = item_of(e[u], e[u].x ? (uint32_t)__builtin_ctz(e[u].x) : 0u, k, owner, col) && e[u].x != 0;
                                    const uint32_t dd = first_half_distance(a.rows + ((ok ? owner : 0ull) * nv + v) * 8, a.cols + (ok ? col : 0ull) * 8, x[u][k]);
                                    d[u][k] = ok ? dd : 0xFFFFFFFFu;
                                }
#pragma unroll
                            for (int u = 0; u < U; u++)
#pragma unroll
                                for (int k = 0; k < IPB; k++)
                                    if (d[u][k] <= a.threshold) {
                                        unsigned long long owner, col;
                                        item_of(e[u], (uint32_t)__builtin_ctz(e[u].x), k, owner, col);
                                        complete_rest(a, a.rows + (owner * nv + v) * 8, a.cols + col * 8, x[u][k], d[u][k], owner, col, v);
                                    }
#pragma unroll
                            for (int u = 0; u < U; u++) e[u].x &= e[u].x - 1;  // (0 stays 0)
                        }
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            uint32_t bm = e[u].x;
                            while (bm != 0) {
                                const uint32_t bit = (uint32_t)__builtin_ctz(bm);
                                bm &= bm - 1;
#pragma unroll
                                for (int k = 0; k < IPB; k++) {
                                    unsigned long long owner, col;
                                    if (item_of(e[u], bit, k, owner, col)) complete(owner, col);
                                }
                            }
                        }
                    }
                } else {
                    // queue overflow (heavily duplicated data): every pair of this wave's rows and the columns since the last
                    // drain is completed exactly
                    const uint32_t ccols = cend - undrained;
                    for (uint32_t t = lane; t < (uint32_t)(32 * MF_RB) * ccols; t += 64) {
                        const unsigned long long owner = row0 + wrow + t / ccols, col = col0 + undrained + t % ccols;
                        if (owner < a.n) complete(owner, col);
                    }
                }
                asm volatile("" ::: "memory");
                if (lane == 0) s_qn[wave] = 0;
                undrained = cend;
            }
        }
        __syncthreads();  // the last chunk's buffers and queue are free before the next pass starts
    }
}

// ---------------------------------------------------------------------------------------------
// 64-bit hashes (impl HammingHash for u64, hamminghash.rs:23-41): the same tiled sweep, 2 dwords per hash
// ---------------------------------------------------------------------------------------------
struct Sweep64Args {
    const uint2 *hashes;  // [n] little-endian u64 as (lo, hi)
    unsigned long long n;
    uint32_t threshold, mih_tol, part, nparts, n_tiles;
    unsigned long long n_tile_pairs, block0;
    rph_edge *edges;
    unsigned long long cap;
    unsigned long long *count;
};

__global__ void __launch_bounds__(256) hamming64_sweep_kernel(Sweep64Args a)
{
    __shared__ uint2 s_cols[T_FILES];  // 8 KiB
    const unsigned long long p = (unsigned long long)a.part + (a.block0 + blockIdx.x) * a.nparts;
    if (p >= a.n_tile_pairs) return;
    uint32_t I, J;
    tile_pair(p, a.n_tiles, I, J);
    const unsigned long long col0 = (unsigned long long)J * T_FILES, row0 = (unsigned long long)I * T_FILES;
    const uint32_t ncols = (uint32_t)((a.n - col0) < (unsigned long long)T_FILES ? (a.n - col0) : T_FILES);
    for (uint32_t t = threadIdx.x; t < T_FILES; t += 256) s_cols[t] = t < ncols ? a.hashes[col0 + t] : make_uint2(0, 0);
    __syncthreads();
    constexpr int R64 = T_FILES / 256;  // 4 rows per lane
    uint2 rw[R64];
#pragma unroll
    for (int r = 0; r < R64; r++) {
        const unsigned long long owner = row0 + (unsigned long long)r * 256 + threadIdx.x;
        rw[r] = a.hashes[owner < a.n ? owner : 0];
    }
    for (uint32_t c = 0; c < ncols; c++) {
        const uint2 cv = s_cols[c];
        uint32_t dmin = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < R64; r++) {
            const uint32_t d = bcnt_acc(rw[r].y ^ cv.y, bcnt_acc(rw[r].x ^ cv.x, 0));
            dmin = d < dmin ? d : dmin;
        }
        if (dmin <= a.threshold) {
#pragma unroll 1
            for (int r = 0; r < R64; r++) {
                const unsigned long long owner = row0 + (unsigned long long)r * 256 + threadIdx.x, col = col0 + c;
                if (owner >= a.n || col <= owner) continue;  // i < j only (hamminghash.rs:216)
                const uint2 rv = a.hashes[owner];
                const uint32_t x0 = rv.x ^ cv.x, x1 = rv.y ^ cv.y;
                const uint32_t d = (uint32_t)__builtin_popcount(x0) + (uint32_t)__builtin_popcount(x1);
                if (d > a.threshold) continue;
                // find_groups reachability for u64: 8 chunks of 8 bits (hamminghash.rs:24-31), first chunk with popcount <= tol
                uint32_t flags = 0;
                for (int k = 7; k >= 0; k--) {
                    const uint32_t c8 = ((k < 4 ? x0 : x1) >> ((k & 3) * 8)) & 0xFFu;
                    const uint32_t pc = (uint32_t)__builtin_popcount(c8);
                    if (pc <= a.mih_tol) flags = RPH_EDGE_MIH_R1 | ((uint32_t)k << 5) | (pc == 0 ? 0u : 1u + (uint32_t)__builtin_ctz(c8));
                }
                const unsigned long long at = atomicAdd(a.count, 1ull);
                if (at < a.cap) {
                    rph_edge e;
                    e.i = (uint32_t)owner;
                    e.j = (uint32_t)col;
                    e.d = (uint16_t)d;
                    e.flags = (uint16_t)flags;
                    a.edges[at] = e;
                }
            }
        }
    }
}

}  // namespace

// Prefix width of the fast path, in dwords.  The partial distance over the first 32 PW bits is a lower bound of the distance, so
// the fast path only has to keep unrelated pairs away from the exact completion: for unrelated hashes it is ~ N(16 PW, 8 PW).
// The smallest PW that leaves >= 4.2 sigma to the threshold lets fewer than ~1.3e-5 of the pairs through (<= 7 M exact
// completions per 5e11 pairs, well under a millisecond); correctness never depends on it (the completion is exact, and a
// flooded candidate queue falls back to exhaustive completion).  kernel 2 (fp4) works on 64-bit slices: PW even.
#ifndef RPH_FP4_SIGMA
#define RPH_FP4_SIGMA 4.0
#endif
extern "C" int rph_hamming_prefix_dwords(uint32_t threshold, int kernel)
{
    int pw = 8;
    for (int cand = 4; cand < 8; cand++) {
        const double mean = 16.0 * cand, sigma = sqrt(8.0 * cand);
        if ((mean - (double)threshold) / sigma >= (kernel >= 2 ? RPH_FP4_SIGMA : 4.2)) {
            pw = cand;
            break;
        }
    }
    if (kernel >= 2 && (pw & 1)) pw++;
    return pw;
}

namespace {
// ---- popcount sort of the hash array (for FmtFp4ZO) ----
__global__ void __launch_bounds__(256) prefix_popcount_kernel(const uint32_t *__restrict__ hashes, unsigned long long n, int pw, uint16_t *keys, uint32_t *ids)
{
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 lo = reinterpret_cast<const uint4 *>(hashes)[2 * i], hi = reinterpret_cast<const uint4 *>(hashes)[2 * i + 1];
        const uint32_t d[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t pc = 0;
#pragma unroll
        for (int w = 0; w < 8; w++) pc += w < pw ? (uint32_t)__builtin_popcount(d[w]) : 0u;
        keys[i] = (uint16_t)pc;
        ids[i] = (uint32_t)i;
    }
}
__global__ void __launch_bounds__(256) gather_hashes_kernel(const uint4 *__restrict__ hashes, const uint32_t *__restrict__ perm, unsigned long long n, uint4 *sorted)
{
    // thread = (sorted position, half of the hash)
    for (unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n; t += (unsigned long long)gridDim.x * blockDim.x)
        sorted[t] = hashes[2ull * perm[t >> 1] + (t & 1)];
}
constexpr unsigned long long ZO_MIN_N = 32768;  // below this the sort's launches cost more than the clock gain returns
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

// Sorted copy of the hashes + permutation + popcounts in the context's sweep scratch (stream ordered).  Stable radix sort: every
// rank of a multi-GPU run builds the SAME order, so the part / nparts shares stay a partition of the pairs.
static int prepare_sorted(rph_ctx *ctx, const uint8_t *d_hashes, uint64_t n, int pw, hipStream_t stream, SweepArgs &a)
{
    size_t temp_bytes = 0;
    RPH_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const uint16_t *)nullptr, (uint16_t *)nullptr, (const uint32_t *)nullptr,
                                                     (uint32_t *)nullptr, (int)n, 0, 9, stream));
    const size_t o_sorted = 0, o_kin = o_sorted + align256(n * 32), o_kout = o_kin + align256(n * 2), o_iin = o_kout + align256(n * 2),
                 o_iout = o_iin + align256(n * 4), o_temp = o_iout + align256(n * 4), need = o_temp + align256(temp_bytes);
    if (ctx->sweep_scratch_bytes < need) {
        RPH_HIP_CHECK(hipDeviceSynchronize());  // kernels of any stream may still be using the old scratch
        if (ctx->sweep_scratch) RPH_HIP_CHECK(hipFree(ctx->sweep_scratch));
        ctx->sweep_scratch = nullptr;
        ctx->sweep_scratch_bytes = 0;
        RPH_HIP_CHECK(hipMalloc(&ctx->sweep_scratch, need + need / 4));
        ctx->sweep_scratch_bytes = need + need / 4;
    }
    if (!ctx->sweep_done) RPH_HIP_CHECK(hipEventCreateWithFlags(&ctx->sweep_done, hipEventDisableTiming));
    if (ctx->sweep_used && ctx->sweep_stream != stream) RPH_HIP_CHECK(hipStreamWaitEvent(stream, ctx->sweep_done, 0));
    uint8_t *base = (uint8_t *)ctx->sweep_scratch;
    uint16_t *kin = (uint16_t *)(base + o_kin), *kout = (uint16_t *)(base + o_kout);
    uint32_t *iin = (uint32_t *)(base + o_iin), *iout = (uint32_t *)(base + o_iout);
    const unsigned grid = (unsigned)std::min<unsigned long long>((n + 255) / 256, 65536);
    hipLaunchKernelGGL(prefix_popcount_kernel, dim3(grid), dim3(256), 0, stream, (const uint32_t *)d_hashes, n, pw, kin, iin);
    RPH_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(base + o_temp, temp_bytes, kin, kout, iin, iout, (int)n, 0, 9, stream));
    hipLaunchKernelGGL(gather_hashes_kernel, dim3((unsigned)std::min<unsigned long long>((2 * n + 255) / 256, 65536)), dim3(256), 0, stream,
                       (const uint4 *)d_hashes, iout, n, (uint4 *)(base + o_sorted));
    RPH_HIP_CHECK(hipGetLastError());
    a.rows = a.cols = (const uint32_t *)(base + o_sorted);
    a.perm = iout;
    a.pcs = kout;
    return RPH_OK;
}

int rph_launch_hamming_sweep(rph_ctx *ctx, const uint8_t *d_rows, uint32_t n_variants, const uint8_t *d_cols, const uint8_t *d_low_conf,
                             const uint8_t *d_has_features, uint64_t n, uint32_t threshold, uint32_t part, uint32_t nparts, rph_edge *d_edges,
                             uint64_t cap, unsigned long long *d_count, hipStream_t stream, int use_mfma)
{
    if (nparts == 0 || part >= nparts || (n_variants != 1 && n_variants != 8) || n > 0xFFFFFFFFull) {
        rph_set_error("hamming sweep: bad arguments (n=%llu variants=%u part=%u/%u)", (unsigned long long)n, n_variants,
                      part, nparts);
        return RPH_ERR_INVALID_ARG;
    }
    if (n < 2) return RPH_OK;
    SweepArgs a;
    a.rows = reinterpret_cast<const uint32_t *>(d_rows);
    a.cols = reinterpret_cast<const uint32_t *>(d_cols);
    a.low_conf = d_low_conf;
    a.has_features = d_has_features;
    a.n = n;
    a.n_variants = n_variants;
    a.threshold = threshold > 256 ? 256 : threshold;
    a.mih_tol = (threshold / 16u) >= 1 ? 1 : 0;
    a.part = part;
    a.nparts = nparts;
    a.n_tiles = (uint32_t)((n + T_FILES - 1) / T_FILES);
    a.n_tile_pairs = (unsigned long long)a.n_tiles * (a.n_tiles + 1ull) / 2ull;
    a.seg_tiles = 1;
    if (use_mfma >= 1) {
        // MFMA kernels: a block sweeps a row tile against a segment of up to 8 column tiles; shorter segments while that
        // would leave this rank fewer than ~8 blocks per resident block slot (2 per CU)
        const unsigned long long want_blocks = 8ull * 512ull * nparts;
        unsigned long long S = a.n_tile_pairs / want_blocks;
        a.seg_tiles = (uint32_t)(S < 1 ? 1 : (S > 8 ? 8 : S));
        a.n_tile_pairs = seg_G(a.n_tiles, a.seg_tiles);  // number of segment blocks
    }
    a.edges = d_edges;
    a.cap = cap;
    a.count = d_count;
    a.perm = nullptr;
    a.pcs = nullptr;
    const unsigned long long mine = (a.n_tile_pairs > part) ? (a.n_tile_pairs - part + nparts - 1) / nparts : 0;
    if (mine == 0) return RPH_OK;
    const int pw = rph_hamming_prefix_dwords(a.threshold, use_mfma);
    // plain all-pairs sweeps (rows are the columns, no per-file rules) run on popcount-sorted {0,1} operands
    const bool zero_one = (use_mfma == 4 || (use_mfma == 2 && n >= ZO_MIN_N)) && ctx && n_variants == 1 && d_rows == d_cols && !d_low_conf && !d_has_features;
    std::unique_lock<std::mutex> scratch_lock;
    if (zero_one) {
        scratch_lock = std::unique_lock<std::mutex>(ctx->mu);
        int rc = prepare_sorted(ctx, d_cols, n, pw, stream, a);
        if (rc != RPH_OK) return rc;
    }
    for (unsigned long long b0 = 0; b0 < mine; b0 += MAX_GRID) {  // one launch carries at most MAX_GRID tile pairs
        a.block0 = b0;
        const dim3 grid((unsigned)((mine - b0) < MAX_GRID ? (mine - b0) : MAX_GRID)), block(BLOCK), mblock(MF_BLOCK);
        if (zero_one) {
            if (pw == 4)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4ZO, 4>), grid, mblock, 0, stream, a);
            else if (pw == 6)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4ZO, 6>), grid, mblock, 0, stream, a);
            else
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4ZO, 8>), grid, mblock, 0, stream, a);
        } else if (use_mfma >= 2) {
            if (pw == 4)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4, 4>), grid, mblock, 0, stream, a);
            else if (pw == 6)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4, 6>), grid, mblock, 0, stream, a);
            else
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4, 8>), grid, mblock, 0, stream, a);
        } else if (use_mfma) {
            if (pw == 4)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtI8, 4>), grid, mblock, 0, stream, a);
            else if (pw == 5)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtI8, 5>), grid, mblock, 0, stream, a);
            else if (pw == 6)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtI8, 6>), grid, mblock, 0, stream, a);
            else if (pw == 7)
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtI8, 7>), grid, mblock, 0, stream, a);
            else
                hipLaunchKernelGGL((hamming_mfma_kernel<FmtI8, 8>), grid, mblock, 0, stream, a);
        } else {
            if (pw == 4)
                hipLaunchKernelGGL(hamming_sweep_kernel<4>, grid, block, 0, stream, a);
            else if (pw == 5)
                hipLaunchKernelGGL(hamming_sweep_kernel<5>, grid, block, 0, stream, a);
            else if (pw == 6)
                hipLaunchKernelGGL(hamming_sweep_kernel<6>, grid, block, 0, stream, a);
            else if (pw == 7)
                hipLaunchKernelGGL(hamming_sweep_kernel<7>, grid, block, 0, stream, a);
            else
                hipLaunchKernelGGL(hamming_sweep_kernel<8>, grid, block, 0, stream, a);
        }
        RPH_HIP_CHECK(hipGetLastError());
    }
    if (zero_one) {  // later users of the sorted scratch on another stream wait for this sweep
        RPH_HIP_CHECK(hipEventRecord(ctx->sweep_done, stream));
        ctx->sweep_stream = stream;
        ctx->sweep_used = true;
    }
    return RPH_OK;
}

int rph_launch_hamming64_sweep(const uint64_t *d_hashes, uint64_t n, uint32_t threshold, uint32_t part, uint32_t nparts,
                               rph_edge *d_edges, uint64_t cap, unsigned long long *d_count, hipStream_t stream, int use_mfma)
{
    if (nparts == 0 || part >= nparts || n > 0xFFFFFFFFull) {
        rph_set_error("hamming64 sweep: bad arguments");
        return RPH_ERR_INVALID_ARG;
    }
    if (n < 2) return RPH_OK;
    if (use_mfma) {
        // fp4 MFMA formulation (the whole 64-bit hash is one slice): same kernel as the 256-bit sweep
        SweepArgs m;
        m.rows = m.cols = reinterpret_cast<const uint32_t *>(d_hashes);
        m.low_conf = nullptr;
        m.has_features = nullptr;
        m.n = n;
        m.n_variants = 1;
        m.threshold = threshold > 64 ? 64 : threshold;
        m.mih_tol = (threshold / 8u) >= 1 ? 1 : 0;  // chunk_tolerance = max_dist / NUM_CHUNKS (hamminghash.rs:193), NUM_CHUNKS = 8
        m.part = part;
        m.nparts = nparts;
        m.n_tiles = (uint32_t)((n + T_FILES - 1) / T_FILES);
        const unsigned long long pairs = (unsigned long long)m.n_tiles * (m.n_tiles + 1ull) / 2ull;
        const unsigned long long S = pairs / (8ull * 768ull * nparts);  // 3 blocks per CU
        m.seg_tiles = (uint32_t)(S < 1 ? 1 : (S > 8 ? 8 : S));
        m.n_tile_pairs = seg_G(m.n_tiles, m.seg_tiles);
        m.edges = d_edges;
        m.cap = cap;
        m.count = d_count;
        const unsigned long long mine = (m.n_tile_pairs > part) ? (m.n_tile_pairs - part + nparts - 1) / nparts : 0;
        for (unsigned long long b0 = 0; b0 < mine; b0 += MAX_GRID) {
            m.block0 = b0;
            hipLaunchKernelGGL((hamming_mfma_kernel<FmtFp4, 2, true>), dim3((unsigned)((mine - b0) < MAX_GRID ? (mine - b0) : MAX_GRID)), dim3(MF_BLOCK), 0,
                               stream, m);
            RPH_HIP_CHECK(hipGetLastError());
        }
        return RPH_OK;
    }
    Sweep64Args a;
    a.hashes = reinterpret_cast<const uint2 *>(d_hashes);
    a.n = n;
    a.threshold = threshold > 64 ? 64 : threshold;
    a.mih_tol = (threshold / 8u) >= 1 ? 1 : 0;  // chunk_tolerance = max_dist / NUM_CHUNKS (hamminghash.rs:193), NUM_CHUNKS = 8
    a.part = part;
    a.nparts = nparts;
    a.n_tiles = (uint32_t)((n + T_FILES - 1) / T_FILES);
    a.n_tile_pairs = (unsigned long long)a.n_tiles * (a.n_tiles + 1ull) / 2ull;
    a.edges = d_edges;
    a.cap = cap;
    a.count = d_count;
    const unsigned long long mine = (a.n_tile_pairs > part) ? (a.n_tile_pairs - part + nparts - 1) / nparts : 0;
    if (mine == 0) return RPH_OK;
    for (unsigned long long b0 = 0; b0 < mine; b0 += MAX_GRID) {
        a.block0 = b0;
        hipLaunchKernelGGL(hamming64_sweep_kernel, dim3((unsigned)((mine - b0) < MAX_GRID ? (mine - b0) : MAX_GRID)), dim3(256), 0, stream, a);
        RPH_HIP_CHECK(hipGetLastError());
    }
    return RPH_OK;
}
