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
#include "rph_internal.h"

namespace {

constexpr int T_FILES = 1024;   // files per tile (rows and columns)
constexpr int BLOCK = 128;      // threads per workgroup
constexpr int R = T_FILES / BLOCK;  // row hashes per lane = 8

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
    uint32_t n_tiles;
    rph_edge *edges;
    unsigned long long cap;
    unsigned long long *count;
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

// Slow path for one (row, column) pair whose partial distance passed: exact distance,
// owner / limit rules, find_groups reachability flags, append.
__device__ __forceinline__ void complete_pair(const SweepArgs &a, const uint32_t *__restrict__ rowp, const uint32_t *colp,
                                              unsigned long long owner, unsigned long long col, uint32_t variant)
{
    uint32_t x[8];
    uint32_t d = 0;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        x[w] = rowp[w] ^ colp[w];
        d += (uint32_t)__builtin_popcount(x[w]);
    }
    if (variant > 0 && a.has_features && !a.has_features[owner]) return;
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
        e.d = (uint16_t)d;
        e.flags = (uint16_t)flags;
        a.edges[at] = e;
    }
}

template <int PW>
__global__ void __launch_bounds__(BLOCK) hamming_sweep_kernel(SweepArgs a)
{
    __shared__ uint4 s_cols[T_FILES * 2];  // [column][2 x uint4] = 32 KiB

    const unsigned long long p = (unsigned long long)a.part + (unsigned long long)blockIdx.x * a.nparts;
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

}  // namespace

int rph_launch_hamming_sweep(const uint8_t *d_rows, uint32_t n_variants, const uint8_t *d_cols, const uint8_t *d_low_conf,
                             const uint8_t *d_has_features, uint64_t n, uint32_t threshold, uint32_t part, uint32_t nparts, rph_edge *d_edges,
                             uint64_t cap, unsigned long long *d_count, hipStream_t stream)
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
    a.edges = d_edges;
    a.cap = cap;
    a.count = d_count;
    const unsigned long long mine = (a.n_tile_pairs > part) ? (a.n_tile_pairs - part + nparts - 1) / nparts : 0;
    if (mine == 0) return RPH_OK;
    if (mine > 0x7FFFFFFFull) {
        rph_set_error("hamming sweep: too many tile pairs for one launch (%llu)", mine);
        return RPH_ERR_INVALID_ARG;
    }
    // Partial-width test: unrelated 256-bit hashes differ in ~16*PW +- sqrt(8*PW) of the first
    // 32*PW bits; keep ~5 sigma between that and the threshold.
    const dim3 grid((unsigned)mine), block(BLOCK);
    if (a.threshold <= 36)
        hipLaunchKernelGGL(hamming_sweep_kernel<4>, grid, block, 0, stream, a);
    else if (a.threshold <= 48)
        hipLaunchKernelGGL(hamming_sweep_kernel<5>, grid, block, 0, stream, a);
    else if (a.threshold <= 60)
        hipLaunchKernelGGL(hamming_sweep_kernel<6>, grid, block, 0, stream, a);
    else if (a.threshold <= 74)
        hipLaunchKernelGGL(hamming_sweep_kernel<7>, grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL(hamming_sweep_kernel<8>, grid, block, 0, stream, a);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

