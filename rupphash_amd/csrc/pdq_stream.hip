// pdq_stream.hip -- streaming single-pass PDQ hash of Luma8 images of any geometry 128..512 x 128..512 for gfx950:
// the hasher behind the pre-downsample (every photo's thumbnail), behind the JPEG decode, and for Luma8 inputs that are not 512x512.
//
// Replaces generate_pdq_from_luma (/root/reference/src/pdqhash.rs:238-262): u8 -> f32 (:244), jarosz_filter_float (2 x rows, cols;
// :341-426), decimate_float (:428-443), and feeds the shared tail (quality :445-460, DCT :306-336, median / hash :59-124; pdq_tail.hpp).
// The image is read once; nothing but the outputs is written (the multi-pass kernels of pdq_kernels.hip move 3.2 MB per 512x344 thumbnail).
//
// One wave owns one image and never synchronises with another wave.  It walks the image in bands of 56 rows x strips of 64 columns:
//   A  pass-1 rows.  The inputs are integers, so every window sum is exact in any order and out = fl(S / n): S comes from the matrix
//      pipe.  v_mfma_i32_32x32x32_i8 with 32 image rows x 64 bytes straight from memory as A (lane = row) and a 0/1 band matrix as B
//      yields S with lane = COLUMN and 16 rows per lane; v_permlane32_swap of two column blocks gives every lane all 32 rows of one
//      column -- the layout the column recurrence wants, with no transposition.  (Bytes are unsigned, the instruction is signed:
//      bytes ^ 0x80, and 128 n comes back through spare K slots: A = 64 there, B = 2 n.)
//   B  pass-1 columns, lane = column: the reference's recurrence step for step (sum += in[t]; sum -= in[t - win]; out = sum / cur),
//      the window's leaving element out of registers (rows of the block above: the first block of a band re-derives the 8 rows
//      above it, so only the running sums persist between bands: 2 KB of LDS).  Outputs go to a 56 x 64 f32 tile in LDS.
//   C  pass-2 rows, lane = row of the band: the tile row comes back as 16 ds_read_b128, the recurrence runs over registers, and only
//      the columns decimate_float keeps are divided and kept (a 32-entry register file, indexed through s_set_gpr_idx).
//   D  pass-2 columns on the kept columns, lane = kept column: the samples are transposed through the (then dead) tile; only the
//      rows decimate_float keeps are divided, and each goes straight into the streaming tail (DCT pass 1 + quality).
// Exactness: every f32 operation of the reference is executed with the same operands in the same order; positions outside the
// image enter the recurrences as +0.0 (x + 0 = x and x - 0 = x exactly; -0.0 cannot arise from sums of non-negative values), and
// sum / cur for cur = 1..8 is Markstein's 1 mul + 2 fma sequence, which IS the IEEE quotient for every normal |sum| < 4096
// (checked exhaustively on the CPU, tools/check_div_small.c).  No FMA contraction anywhere else (-ffp-contract=off).
#include "pdq_tail.hpp"
#include "rph_internal.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

constexpr int ST_BAND = 56;                        // recurrence steps (rows) per band
constexpr int ST_TP = 68;                          // tile pitch in floats: 16-byte rows, conflict-free ds_read_b128 with lane = row
constexpr int ST_TILE_FLOATS = 64 * ST_TP;         // 17 408 B (rows 56..63 are never written)
constexpr int ST_XP = 65;                          // pitch of the sample transposition buffer [32 slots][8 rows above + 56]
constexpr int ST_OFF_SUMS = ST_TILE_FLOATS;        // pass-1 column sums, one per image column
constexpr int ST_OFF_EDGE = ST_OFF_SUMS + 512;     // kept rows of the last lane of a D pass, for the next pass's horizontal gradient
constexpr int ST_OFF_DIV = ST_OFF_EDGE + 32;       // (cur, 1 / cur) of the 56 steps of a band that touches the image's first or last rows
constexpr int ST_LDS_FLOATS = ST_OFF_DIV + 2 * ST_BAND;  // 20 032 B -> 8 waves per CU
static_assert(32 * ST_XP <= ST_TILE_FLOATS, "transposition buffer lives in the dead tile");
static_assert(rph::TAIL_LDS_FLOATS <= ST_TILE_FLOATS, "tail scratch lives in the dead tile");

// (cur, 1 / cur) for cur = 1..8
__constant__ float c_div[9][2] = {{1.0f, 1.0f},        {1.0f, 1.0f},        {2.0f, 0.5f},        {3.0f, 1.0f / 3.0f}, {4.0f, 0.25f},
                                  {5.0f, 1.0f / 5.0f}, {6.0f, 1.0f / 6.0f}, {7.0f, 1.0f / 7.0f}, {8.0f, 0.125f}};

struct StGeo {
    int W, H;
    int win_r, lead_r, a_r;  // window along rows, half - 1, win - half   (pdqhash.rs:246, :351-357)
    int win_c, lead_c;       // the same along columns
    int ns_b;                // strips that hold pixels
    int ns_c;                // strips the row recurrence walks (its last outputs come lead_r steps after the last pixel)
    int nb;                  // bands
};

__device__ __forceinline__ void st_fence() { asm volatile("" ::: "memory"); }  // one wave = one workgroup: the LDS keeps a wave's accesses in order

// IEEE N / d for d = 1..8 and every normal |N| < 4096 (Markstein: q0 = N * (1/d); r = N - q0 d; q = q0 + r * (1/d))
__device__ __forceinline__ float st_div(float N, float d, float dinv)
{
    const float q0 = N * dinv;
    const float r = __builtin_fmaf(-q0, d, N);
    return __builtin_fmaf(r, dinv, q0);
}

// number of samples under the window at recurrence step t of a line of `len` with window `win`
__device__ __forceinline__ int st_count(int t, int len, int win)
{
    const int hi = t < len - 1 ? t : len - 1, lo = t - win + 1 > 0 ? t - win + 1 : 0;
    const int c = hi - lo + 1;
    return c < 1 ? 1 : (c > 8 ? 8 : c);
}

struct StWave {
    float *lds;
    int lane;
    // pass-2 row recurrence (lane = row of the band), reset per band
    float csum, cprev[8];
    // pass-2 column recurrence (lane = kept column), carried over the whole image
    float dsum, dprev[8];
    f32x32 smp;  // pass-2 row outputs at the kept columns since the last D pass
    v4u carry[2];   // the strip's last 32 image bytes per row (two row blocks): they are the next strip's first 32
    v4u ahead[2][2];  // an even strip also fetches the odd strip's 64 new bytes per row: a row's 128 new bytes of a strip pair in one go
    rph::TailAcc tail;
    bool want_quality;
};

// ---- A + B: strip [c0, c0 + 64) of band [T0, T0 + 56) -> tile
template <int WIN_C>
__device__ __forceinline__ void st_ab_stage(StWave &w, const __amdgpu_buffer_rsrc_t rs, const StGeo &g, const int rs32, const int T0, const int c0)
{
    const int lane = w.lane, n = lane & 31, kh = lane >> 5;
    float *tile = w.lds;
    float *sums = w.lds + ST_OFF_SUMS;

    // image bytes: block m holds rows T0 - 8 + 32 m ..; lane (n, kh) fetches bytes [c0 - 16 + 32 c + 16 kh, + 16) of row n of the block.
    // Chunk 0 of a strip is chunk 2 of the strip before it ([c0 - 16, c0 + 16) = [(c0 - 64) + 48, (c0 - 64) + 80)): it stays in registers, and an
    // even strip fetches the 64 new bytes per row of the odd strip behind it as well: 128 new bytes per row and strip pair in one go, no byte of
    // a band requested twice, a 128-byte line touched by at most two fetches.
    v4u ch[2][3];
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const int row = T0 - 8 + 32 * m + n;
        const int off0 = row * rs32 + c0 - 16 + 16 * kh;
        const bool odd_strip = (c0 & 64) != 0;  // (uniform)
#pragma unroll
        for (int c = 0; c < 5; c++) {  // chunks 3 and 4 are the next strip's chunks 1 and 2
            if (c == 0 && c0 != 0) {
                ch[m][0] = w.carry[m];
                continue;
            }
            if (odd_strip) {
                if (c == 1 || c == 2) ch[m][c] = w.ahead[m][c - 1];
                continue;
            }
            const int off = off0 + 32 * c;
            const uint32_t uoff = (row < 0 || row >= g.H || off < 0) ? 0x80000000u : (uint32_t)off;  // outside: the range check returns 0
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rs, uoff, 0, 0);
            if (c < 3)
                ch[m][c] = v;
            else
                w.ahead[m][c - 3] = v;
        }
        w.carry[m] = ch[m][2];
    }
    // band matrices of the two 32-column blocks: slot 32 ks + 16 kh + j <-> source column c0 - 16 + 32 nb + 32 ks + 16 kh + j
    v4i bm[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
        const int xo = c0 + 32 * nb + n;
        const bool ok = xo < g.W;
        const int lo = xo - g.a_r > 0 ? xo - g.a_r : 0, hi = xo + g.lead_r < g.W - 1 ? xo + g.lead_r : g.W - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            const int s0 = c0 - 16 + 32 * nb + 32 * ks + 16 * kh;
            int jlo = lo - s0, jhi = hi - s0;
            jlo = jlo > 0 ? jlo : 0;
            jhi = jhi < 15 ? jhi : 15;
            const uint32_t m16 = (ok && jlo <= jhi) ? (((2u << jhi) - 1u) & ~((1u << jlo) - 1u)) : 0u;
            uint32_t bd[4];
#pragma unroll
            for (int q = 0; q < 4; q++) bd[q] = (((m16 >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u;
            if (ks == 1 && kh == 1) {  // slots 52..63 are never under a window: A = 64 there, B = 2 n in slot 52 -> + 128 n
                bd[1] = ok ? 2u * (uint32_t)(hi - lo + 1) : 0u;
                bd[2] = bd[3] = 0u;
            }
            bm[nb][ks] = v4i{(int)bd[0], (int)bd[1], (int)bd[2], (int)bd[3]};
        }
    }
    // divisor of this lane's column in the row pass
    float dA, dAinv;
    {
        const int x = c0 + lane;
        const int lo = x - g.a_r > 0 ? x - g.a_r : 0, hi = x + g.lead_r < g.W - 1 ? x + g.lead_r : g.W - 1;
        const int nr = x < g.W ? hi - lo + 1 : 1;
        dA = (float)nr;
        dAinv = 1.0f / dA;
    }
    const float sd = c_div[WIN_C][0], sdi = c_div[WIN_C][1];
    const bool steady = T0 >= WIN_C - 1 && T0 + ST_BAND - 1 <= g.H - 1;  // every step of the band divides by the full window
    const float2 *divtab = reinterpret_cast<const float2 *>(w.lds + ST_OFF_DIV);  // else: (cur, 1 / cur) per step, written once per band

    float sum = sums[c0 + lane];
    float keep[8];  // the last 8 row-pass values of block 0
#pragma unroll
    for (int m = 0; m < 2; m++) {
        v16i acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
            acc[nb] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                const v4u raw = ch[m][nb + ks];
                v4i av = v4i{(int)(raw[0] ^ 0x80808080u), (int)(raw[1] ^ 0x80808080u), (int)(raw[2] ^ 0x80808080u), (int)(raw[3] ^ 0x80808080u)};
                if (ks == 1) {
                    av[1] = kh ? 0x40404040 : av[1];
                    av[2] = kh ? 0x40404040 : av[2];
                    av[3] = kh ? 0x40404040 : av[3];
                }
                acc[nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bm[nb][ks], acc[nb], 0, 0, 0);
            }
        }
        // lane l < 32: column l of block 0, l >= 32: column l - 32 of block 1; rows 8 q + i in P[4 q + i], rows 8 q + 4 + i in R[4 q + i]
        float a[32];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const auto sw = __builtin_amdgcn_permlane32_swap(acc[0][i], acc[1][i], false, false);
            const int q = i >> 2, u = i & 3;
            a[8 * q + u] = st_div((float)(int)sw[0], dA, dAinv);
            a[8 * q + 4 + u] = st_div((float)(int)sw[1], dA, dAinv);
        }
#pragma unroll
        for (int r = (m == 0 ? 8 : 0); r < 32; r++) {
            const int tr = m == 0 ? r - 8 : 24 + r;  // step t = T0 + tr, tile row tr
            sum = sum + a[r];
            sum = sum - ((m == 0 || r >= WIN_C) ? a[r - WIN_C >= 0 ? r - WIN_C : 0] : keep[8 + r - WIN_C >= 0 && 8 + r - WIN_C < 8 ? 8 + r - WIN_C : 0]);
            float d = sd, di = sdi;
            if (!steady) {
                const float2 dd = divtab[tr];  // every lane reads the same address: one broadcast
                d = dd.x;
                di = dd.y;
            }
            tile[tr * ST_TP + lane] = st_div(sum, d, di);
        }
        if (m == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) keep[j] = a[24 + j];
        }
    }
    sums[c0 + lane] = sum;
}

// ---- C: the row recurrence over the tile's 64 columns; `em` marks the steps whose output decimate_float keeps
template <int WIN_R>
__device__ __forceinline__ void st_c_stage(StWave &w, const StGeo &g, const int c0, const unsigned long long em, int &cnt)
{
    const float4 *row = reinterpret_cast<const float4 *>(w.lds + w.lane * ST_TP);
    float v[64];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const float4 f = row[q];
        v[4 * q] = f.x;
        v[4 * q + 1] = f.y;
        v[4 * q + 2] = f.z;
        v[4 * q + 3] = f.w;
    }
    const float sd = c_div[WIN_R][0], sdi = c_div[WIN_R][1];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        w.csum = w.csum + v[i];
        w.csum = w.csum - (i >= WIN_R ? v[i >= WIN_R ? i - WIN_R : 0] : w.cprev[i < WIN_R ? 8 + i - WIN_R : 0]);
        if ((em >> i) & 1ull) {
            const int t = c0 + i;
            float d = sd, di = sdi;
            if (t < WIN_R - 1 || t > g.W - 1) {
                const int c = __builtin_amdgcn_readfirstlane(st_count(t, g.W, WIN_R));
                d = c_div[c][0];
                di = c_div[c][1];
            }
            w.smp[__builtin_amdgcn_readfirstlane(cnt)] = st_div(w.csum, d, di);
            cnt++;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) w.cprev[j] = v[56 + j];
}

// steps of strip [c0, c0 + 64) whose output is a kept column: output column o leaves at step o + lead_r; jn = first slot not yet marked
__device__ __forceinline__ unsigned long long st_emit_mask(const StGeo &g, int c0, int &jn)
{
    unsigned long long em = 0;
    while (jn < 64) {
        const int te = ((2 * jn + 1) * g.W) / 128 + g.lead_r;
        if (te >= c0 + 64) break;
        em |= 1ull << (te - c0);
        jn++;
    }
    return em;
}

// ---- D: the column recurrence of kept columns [base, base + cnt) over the band's rows, feeding the tail.  ni / e_idx: next kept row and
// its index within the band, as they stood at the start of the band (every pass of a band walks the same rows).
__device__ __forceinline__ void st_d_pass(StWave &w, const StGeo &g, const int T0, const int base, const int cnt, int ni, const int want_rows)
{
    float *xp = w.lds;
    float *edge = w.lds + ST_OFF_EDGE;
    const int lane = w.lane;
    // samples to [slot][8 + row]; rows outside the image enter the recurrence as 0
    {
        const int o = T0 + lane - g.lead_c;
        const bool inside = o >= 0 && o < g.H;
        if (lane < ST_BAND) {
#pragma unroll
            for (int e = 0; e < 32; e++) xp[e * ST_XP + 8 + lane] = inside ? w.smp[e] : 0.0f;
        }
    }
    st_fence();
    const bool active = lane >= base && lane < base + cnt;
    if (active) {
        float *p = xp + (lane - base) * ST_XP;
#pragma unroll
        for (int j = 0; j < 8; j++) p[j] = w.dprev[j];
        const float *lv = p + 8 - g.win_c;
        int next_ri = ((2 * ni + 1) * g.H) / 128, e_idx = 0;
#pragma unroll 1
        for (int gidx = 0; gidx < ST_BAND / 8; gidx++) {
            float in[8], out[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                in[u] = p[8 + 8 * gidx + u];
                out[u] = lv[8 * gidx + u];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int td = T0 + 8 * gidx + u - g.lead_c;  // step of this recurrence = row of the pass-2 row output
                w.dsum = w.dsum + in[u];
                w.dsum = w.dsum - out[u];
                if (ni < want_rows && td - g.lead_c == next_ri) {
                    const float val = w.dsum / (float)st_count(td, g.H, g.win_c);
                    float left = __shfl_up(val, 1);
                    if (lane == base && base > 0) left = edge[e_idx];
                    if (lane == base + cnt - 1) edge[e_idx] = val;
                    rph::tail_row_left(w.tail, val, left, lane > 0, ni, w.want_quality);
                    e_idx++;
                    ni++;
                    next_ri = ((2 * ni + 1) * g.H) / 128;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) w.dprev[j] = p[8 + ST_BAND - 8 + j];
    }
    st_fence();
}

__global__ void __launch_bounds__(64, 2) pdq_stream_kernel(const uint8_t *__restrict__ px, uint32_t n, uint32_t w_, uint32_t h_, size_t row_stride, size_t image_stride,
                                                           uint8_t *hash, float *quality, float *coeffs, uint8_t *dihedral, uint8_t *valid)
{
    __shared__ __attribute__((aligned(16))) float lds[ST_LDS_FLOATS];
    const uint32_t img = blockIdx.x;
    StGeo g;
    g.W = (int)w_;
    g.H = (int)h_;
    g.win_r = (g.W + 63) / 64;
    g.win_c = (g.H + 63) / 64;
    const int half_r = (g.win_r + 2) / 2, half_c = (g.win_c + 2) / 2;
    g.lead_r = half_r - 1;
    g.a_r = g.win_r - half_r;
    g.lead_c = half_c - 1;
    g.ns_b = (g.W + 63) / 64;
    g.ns_c = ((127 * g.W) / 128 + g.lead_r) / 64 + 1;
    g.nb = (g.H + 2 * g.lead_c + ST_BAND - 1) / ST_BAND;
    const int rs32 = (int)row_stride;

    StWave w;
    w.lds = lds;
    w.lane = threadIdx.x;
    w.dsum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) w.dprev[j] = 0.0f;
    w.smp = 0.0f;
    rph::tail_init(w.tail);
    w.want_quality = quality != nullptr;
#pragma unroll
    for (int i = 0; i < 8; i++) lds[ST_OFF_SUMS + 64 * i + w.lane] = 0.0f;

    // The range check works on whole dwords: the image ends with the aligned dword that holds its last pixel (rows start on dword
    // boundaries, so that dword never leaves the page of the last pixel; what follows the pixel in it is under no window).
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(px + (size_t)img * image_stride), 0,
                                                                        (int)((size_t)(g.H - 1) * row_stride + (((size_t)g.W + 3) & ~(size_t)3)), 0x00027000);

    int ni = 0;  // next kept row (decimate_float's row index)
#pragma unroll 1
    for (int k = 0; k < g.nb; k++) {
        const int T0 = ST_BAND * k;
        w.csum = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; j++) w.cprev[j] = 0.0f;
        int jn = 0, base = 0, cnt = 0;
        unsigned long long em = st_emit_mask(g, 0, jn);
        if (!(T0 >= g.win_c - 1 && T0 + ST_BAND - 1 <= g.H - 1) && w.lane < ST_BAND) {
            const int c = st_count(T0 + w.lane, g.H, g.win_c);
            lds[ST_OFF_DIV + 2 * w.lane] = c_div[c][0];
            lds[ST_OFF_DIV + 2 * w.lane + 1] = c_div[c][1];
        }
        st_fence();
#pragma unroll 1
        for (int s = 0; s < g.ns_c; s++) {
            const int c0 = 64 * s;
            if (s < g.ns_b) {
                switch (g.win_c) {
                case 2: st_ab_stage<2>(w, rs, g, rs32, T0, c0); break;
                case 3: st_ab_stage<3>(w, rs, g, rs32, T0, c0); break;
                case 4: st_ab_stage<4>(w, rs, g, rs32, T0, c0); break;
                case 5: st_ab_stage<5>(w, rs, g, rs32, T0, c0); break;
                case 6: st_ab_stage<6>(w, rs, g, rs32, T0, c0); break;
                case 7: st_ab_stage<7>(w, rs, g, rs32, T0, c0); break;
                default: st_ab_stage<8>(w, rs, g, rs32, T0, c0); break;
                }
            } else {  // past the last pixel: the recurrence only subtracts
#pragma unroll
                for (int r = 0; r < ST_BAND; r++) lds[r * ST_TP + w.lane] = 0.0f;
            }
            st_fence();
            switch (g.win_r) {
            case 2: st_c_stage<2>(w, g, c0, em, cnt); break;
            case 3: st_c_stage<3>(w, g, c0, em, cnt); break;
            case 4: st_c_stage<4>(w, g, c0, em, cnt); break;
            case 5: st_c_stage<5>(w, g, c0, em, cnt); break;
            case 6: st_c_stage<6>(w, g, c0, em, cnt); break;
            case 7: st_c_stage<7>(w, g, c0, em, cnt); break;
            default: st_c_stage<8>(w, g, c0, em, cnt); break;
            }
            st_fence();
            // the next strip's kept columns; the register file holds 32
            em = st_emit_mask(g, c0 + 64, jn);
            const bool last = s == g.ns_c - 1;
            if (cnt > 0 && (last || cnt + __builtin_popcountll(em) > 32)) {
                st_d_pass(w, g, T0, base, cnt, ni, 64);
                base += cnt;
                cnt = 0;
            }
        }
        // kept rows this band has produced: row ri leaves at pass-2 column step ri + lead_c = band row ri + 2 lead_c - T0
        while (ni < 64 && ((2 * ni + 1) * g.H) / 128 + 2 * g.lead_c < T0 + ST_BAND) ni++;
    }
    st_fence();
    rph::tail_finish(w.tail, lds, w.lane, hash + (size_t)img * 32, quality ? quality + img : nullptr, coeffs ? coeffs + (size_t)img * 256 : nullptr,
                     dihedral ? dihedral + (size_t)img * 256 : nullptr);
    if (valid && w.lane == 0) valid[img] = 1;
}

// to_luma601 (pdqhash.rs:268-284) of Rgb8 / Rgba8 pixels into a Luma8 plane with 16-byte aligned rows: four pixels per thread, whole dwords in
// and out.  The streaming kernel takes its A operand -- 16 luma bytes per lane -- straight from memory, so colour inputs pass through this
// plane (read 3 or 4 bytes, write 1, read 1 per pixel); so do Luma8 inputs whose rows do not start on dword boundaries (CH = 1: a copy).
// Source rows of any alignment: aligned dwords + v_alignbyte.
template <int CH>
__global__ void __launch_bounds__(256) st_luma_kernel(const uint8_t *__restrict__ px, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride,
                                                      uint8_t *__restrict__ out, uint32_t out_pitch, size_t out_stride)
{
    const uint32_t quads = (w + 3) / 4;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= quads * h) return;
    const uint32_t y = t / quads, q = t - y * quads;
    const uint8_t *p = px + (size_t)blockIdx.y * image_stride + (size_t)y * row_stride + (size_t)q * 4 * CH;
    uint32_t d[CH];
    if (q * 4 + 4 <= w) {
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u);
        const uint32_t *p4 = reinterpret_cast<const uint32_t *>(p - sh);
        uint32_t raw[CH + 1];
#pragma unroll
        for (int i = 0; i < CH; i++) raw[i] = p4[i];
        raw[CH] = sh ? p4[CH] : 0u;  // (with sh != 0 that dword holds bytes of this quad: it is the image's own; with sh == 0 it is not touched)
#pragma unroll
        for (int i = 0; i < CH; i++) d[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], sh);
    } else {  // the row's last, partial quad: byte by byte, nothing is read behind the row
#pragma unroll
        for (int i = 0; i < CH; i++) d[i] = 0;
        for (uint32_t b = 0; b < (w - q * 4) * CH; b++) d[b >> 2] |= (uint32_t)p[b] << (8 * (b & 3));
    }
    uint32_t o = 0;
    if (CH == 1) {
        o = d[0];
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int B = i * CH;
            const uint32_t r = (d[B >> 2] >> (8 * (B & 3))) & 0xFFu, g = (d[(B + 1) >> 2] >> (8 * ((B + 1) & 3))) & 0xFFu, b = (d[(B + 2) >> 2] >> (8 * ((B + 2) & 3))) & 0xFFu;
            o |= ((299u * r + 587u * g + 114u * b + 500u) / 1000u) << (8 * i);
        }
    }
    reinterpret_cast<uint32_t *>(out + (size_t)blockIdx.y * out_stride + (size_t)y * out_pitch)[q] = o;
}

}  // namespace

bool rph_pdq_stream_supported(const uint8_t *d_px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride)
{
    return channels == 1 && w >= 128 && w <= 512 && h >= 128 && h <= 512 && (row_stride % 4) == 0 && (image_stride % 4) == 0 && ((uintptr_t)d_px % 4) == 0 &&
           (size_t)h * row_stride < ((size_t)1 << 30);
}

// Rgb8 / Rgba8 images of the same geometries, and Luma8 images whose rows do not lie on dword boundaries: a Luma8 plane with aligned rows
// in the context's scratch, then the streaming kernel.  Called with ctx->mu held.
bool rph_pdq_stream_color_supported(const uint8_t *d_px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride)
{
    (void)d_px, (void)row_stride, (void)image_stride;  // any alignment
    return (channels == 1 || channels == 3 || channels == 4) && w >= 128 && w <= 512 && h >= 128 && h <= 512;
}

int rph_launch_pdq_stream_color(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride, size_t image_stride,
                                uint8_t *d_hash, float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    const uint32_t pitch = (w + 15u) & ~15u;
    const size_t plane = (size_t)pitch * h;
    uint32_t chunk = (uint32_t)(((size_t)1 << 30) / plane);
    chunk = chunk > n ? n : (chunk > 65535u ? 65535u : chunk);
    const size_t need = plane * chunk;
    if (ctx->scratch_bytes < need) {
        RPH_HIP_CHECK(hipDeviceSynchronize());  // kernels of any stream may still be using the old scratch
        if (ctx->scratch) RPH_HIP_CHECK(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        RPH_HIP_CHECK(hipMalloc((void **)&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    if (!ctx->scratch_done) RPH_HIP_CHECK(hipEventCreateWithFlags(&ctx->scratch_done, hipEventDisableTiming));
    if (ctx->scratch_used && ctx->scratch_stream != stream) RPH_HIP_CHECK(hipStreamWaitEvent(stream, ctx->scratch_done, 0));
    uint8_t *luma = reinterpret_cast<uint8_t *>(ctx->scratch);
    const uint32_t quads = (w + 3) / 4;
    for (uint32_t first = 0; first < n; first += chunk) {
        const uint32_t m = (n - first) < chunk ? (n - first) : chunk;
        const dim3 grid((quads * h + 255) / 256, m);
        const uint8_t *src = d_px + (size_t)first * image_stride;
        if (channels == 1)
            hipLaunchKernelGGL(st_luma_kernel<1>, grid, dim3(256), 0, stream, src, w, h, row_stride, image_stride, luma, pitch, plane);
        else if (channels == 3)
            hipLaunchKernelGGL(st_luma_kernel<3>, grid, dim3(256), 0, stream, src, w, h, row_stride, image_stride, luma, pitch, plane);
        else
            hipLaunchKernelGGL(st_luma_kernel<4>, grid, dim3(256), 0, stream, src, w, h, row_stride, image_stride, luma, pitch, plane);
        hipLaunchKernelGGL(pdq_stream_kernel, dim3(m), dim3(64), 0, stream, (const uint8_t *)luma, m, w, h, (size_t)pitch, plane, d_hash + (size_t)first * 32,
                           d_quality ? d_quality + first : nullptr, d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr,
                           d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr, d_valid ? d_valid + first : nullptr);
        RPH_HIP_CHECK(hipGetLastError());
    }
    RPH_HIP_CHECK(hipEventRecord(ctx->scratch_done, stream));
    ctx->scratch_stream = stream;
    ctx->scratch_used = true;
    return RPH_OK;
}

int rph_launch_pdq_stream(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, size_t row_stride, size_t image_stride, uint8_t *d_hash,
                          float *d_quality, float *d_coeffs, uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream)
{
    (void)ctx;
    if (n == 0) return RPH_OK;
    hipLaunchKernelGGL(pdq_stream_kernel, dim3(n), dim3(64), 0, stream, d_px, n, w, h, row_stride, image_stride, d_hash, d_quality, d_coeffs, d_dihedral, d_valid);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}
