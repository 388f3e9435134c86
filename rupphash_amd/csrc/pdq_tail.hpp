// pdq_tail.hpp -- wave-level tail of the PDQ pipeline, shared by the generic and the fused kernel.
//
// Input: one wave (64 lanes) per image; lane j holds column j of the decimated 64x64 buffer,
// b[i] = B[i][j] (registers).  Computes, bit-exactly as /root/reference/src/pdqhash.rs:
//   quality   pdq_image_domain_quality_metric  :445-460
//   DCT       dct64_to_16                      :306-336  (mul then add, k ascending, no FMA)
//   median    coefficient_median               :116-124  (128th smallest under total_cmp)
//   hash      bit_rows + pack_bit_rows         :91-106, :155-162
//   dihedral  generate_dihedral_hashes         :71-87, apply_sign :127-137, transpose :140-151
// Compile with -ffp-contract=off (the reference never fuses a*b+c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dct_table.h"

namespace rph {

// D transposed for the device: c_dct_t.v[k*16 + i] = D[i][k], so the 16 frequencies of one
// spatial index k are contiguous (one s_load_dwordx16 in pass 1, one 64-B segment in pass 2).
struct DctT {
    uint32_t v[1024];
};
constexpr DctT make_dct_t()
{
    constexpr uint32_t src[1024] = RPH_DCT_TABLE_INIT;
    DctT t{};
    for (int k = 0; k < 64; k++)
        for (int i = 0; i < 16; i++) t.v[k * 16 + i] = src[i * 64 + k];
    return t;
}
__constant__ DctT c_dct_t = make_dct_t();

constexpr int TAIL_LDS_FLOATS = 16 * 65 + 256;  // T[16][65] + C[256]

// The tail is always run by ONE wave on LDS scratch private to it.  The LDS executes a wave's DS instructions in issue order, so
// between a write and a read of another lane's value only the compiler has to keep the program order; a workgroup barrier here
// would be wrong wherever the wave is part of a larger workgroup whose other waves do not run the tail (the low-latency PDQ kernel).
__device__ __forceinline__ void tail_fence() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ uint32_t total_key(float f)
{
    // f32::total_cmp as an unsigned ascending key
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// 128th smallest (0-based rank 127) of the wave's 256 keys (4 per lane): MSB-first radix select.
__device__ __forceinline__ uint32_t select_rank127(const uint32_t (&key)[4])
{
    uint32_t prefix = 0, mask = 0;
    int k = 127;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t b = 1u << bit;
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) cnt += __popcll(__ballot(((key[m] & mask) == prefix) && !(key[m] & b)));
        if (k >= cnt) {
            k -= cnt;
            prefix |= b;
        }
        mask |= b;
    }
    return prefix;
}

// bits of one 16x16 sign pattern -> 32 hash bytes; lanes 0..31 each store one byte.
// v[m] is the (possibly sign flipped / transposed) coefficient at matrix position
// (r, c) = ((lane>>4) + 4m, lane & 15).
__device__ __forceinline__ void emit_hash(const float (&v)[4], float median, uint8_t *out32, int lane)
{
    unsigned long long bal[4];
#pragma unroll
    for (int m = 0; m < 4; m++) bal[m] = __ballot(v[m] > median);
    if (lane < 32) {
        const int r = (31 - lane) >> 1;
        const unsigned long long bb = (r >> 2) == 0 ? bal[0] : ((r >> 2) == 1 ? bal[1] : ((r >> 2) == 2 ? bal[2] : bal[3]));
        const uint32_t row = (uint32_t)(bb >> (16 * (r & 3))) & 0xFFFFu;
        out32[lane] = (uint8_t)((lane & 1) ? (row & 0xFFu) : (row >> 8));
    }
}

__device__ __forceinline__ float median_of(const float (&v)[4])
{
    uint32_t key[4];
#pragma unroll
    for (int m = 0; m < 4; m++) key[m] = total_key(v[m]);
    return key_to_float(select_rank127(key));
}

// Hash (+ optional 8 dihedral hashes) from the 256 coefficients held as c[m] = C[lane + 64 m].
// lds_c: 256 floats of LDS scratch (only used when dihedral != nullptr).
__device__ __forceinline__ void hashes_from_coeffs(const float (&c)[4], float *lds_c, int lane, uint8_t *hash32,
                                                   uint8_t *dihedral)
{
    const float med0 = median_of(c);
    if (hash32) emit_hash(c, med0, hash32, lane);
    if (!dihedral) return;

    const int cc = lane & 15;  // column of this lane's positions; row = (lane>>4) + 4m
    // transposed values: position (r, c) takes C[c][r]
    tail_fence();
#pragma unroll
    for (int m = 0; m < 4; m++) lds_c[lane + 64 * m] = c[m];
    tail_fence();
    float t[4];
#pragma unroll
    for (int m = 0; m < 4; m++) t[m] = lds_c[16 * cc + ((lane >> 4) + 4 * m)];

    // apply_sign: a row/col is negated when its DCT frequency (index + 1) is odd, i.e. index even
    const bool c_even = (cc & 1) == 0;
    float nr[4], nc[4], nb[4];      // neg_rows / neg_cols / neg_both at (r, c)
    float tnr[4], tnc[4], tnb[4];   // the same patterns evaluated at the transposed source (c, r)
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const bool r_even = ((((lane >> 4) + 4 * m)) & 1) == 0;
        nr[m] = r_even ? -c[m] : c[m];
        nc[m] = c_even ? -c[m] : c[m];
        nb[m] = (r_even != c_even) ? -c[m] : c[m];
        // source position of t[m] is (row = cc, col = r)
        tnr[m] = c_even ? -t[m] : t[m];
        tnc[m] = r_even ? -t[m] : t[m];
        tnb[m] = (r_even != c_even) ? -t[m] : t[m];
    }
    // a transpose only permutes the coefficients, so each pattern shares its median with its transpose
    const float med_nr = median_of(nr), med_nc = median_of(nc), med_nb = median_of(nb);
    emit_hash(c, med0, dihedral + 0 * 32, lane);      // identity
    emit_hash(tnr, med_nr, dihedral + 1 * 32, lane);  // transpose(neg_rows)   rot90
    emit_hash(nb, med_nb, dihedral + 2 * 32, lane);   // neg_both              rot180
    emit_hash(tnc, med_nc, dihedral + 3 * 32, lane);  // transpose(neg_cols)   rot270
    emit_hash(nc, med_nc, dihedral + 4 * 32, lane);   // neg_cols              mirror-x
    emit_hash(nr, med_nr, dihedral + 5 * 32, lane);   // neg_rows              mirror-y
    emit_hash(t, med0, dihedral + 6 * 32, lane);      // transpose(id)
    emit_hash(tnb, med_nb, dihedral + 7 * 32, lane);  // transpose(neg_both)   anti-transpose
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the tail: rows of the decimated buffer are fed in ascending order k = 0..63
// (lane j passes B[k][j]).  DCT pass 1 accumulates T[i][j] += D[i][k] * B[k][j] -- exactly the reference's
// order over k -- so the 64x64 buffer never has to be resident; the quality metric is a sum of exact
// integers and is accumulated on the fly.
// ---------------------------------------------------------------------------------------------
struct TailAcc {
    float t[16];
    float qsum;
    float prev;  // previous row, for the vertical gradient
};

__device__ __forceinline__ void tail_init(TailAcc &a)
{
#pragma unroll
    for (int i = 0; i < 16; i++) a.t[i] = 0.0f;
    a.qsum = 0.0f;
    a.prev = 0.0f;
}

// k is wave-uniform (compile-time or an SGPR): the 16 coefficients of column k come in one scalar load
__device__ __forceinline__ void tail_row(TailAcc &a, float brow, int k, int lane, bool want_quality)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float p = __uint_as_float(c_dct_t.v[k * 16 + i]) * brow;
        a.t[i] = a.t[i] + p;
    }
    if (want_quality) {
        // trunc(|(a - b) * 100 / 255|) over vertical (row k-1 vs k) and horizontal (lane j vs j+1) neighbours
        const float right = __shfl_down(brow, 1);
        const float gh = truncf(fabsf(((brow - right) * 100.0f) / 255.0f));
        const float gv = truncf(fabsf(((a.prev - brow) * 100.0f) / 255.0f));
        a.qsum += (lane < 63) ? gh : 0.0f;
        a.qsum += (k > 0) ? gv : 0.0f;
    }
    a.prev = brow;
}

// The same for a wave whose lanes feed their columns at different times (pdq_stream.hip: groups of kept columns finish a band one after
// the other): the horizontal gradient of the pair (j - 1, j) is taken by lane j from its LEFT neighbour's value -- the same pairs, the
// same expression (a - b) with a the left value, so the same integers.
__device__ __forceinline__ void tail_row_left(TailAcc &a, float brow, float left, bool has_left, int k, bool want_quality)
{
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const float p = __uint_as_float(c_dct_t.v[k * 16 + i]) * brow;
        a.t[i] = a.t[i] + p;
    }
    if (want_quality) {
        const float gh = truncf(fabsf(((left - brow) * 100.0f) / 255.0f));
        const float gv = truncf(fabsf(((a.prev - brow) * 100.0f) / 255.0f));
        a.qsum += has_left ? gh : 0.0f;
        a.qsum += (k > 0) ? gv : 0.0f;
    }
    a.prev = brow;
}

// lds: TAIL_LDS_FLOATS floats private to this wave.  Outputs are per-image pointers (nullable).
__device__ __forceinline__ void tail_finish(TailAcc &a, float *lds, int lane, uint8_t *hash32, float *quality, float *coeffs,
                                            uint8_t *dihedral)
{
    if (quality) {
        float acc = a.qsum;  // every term is an integer <= 100, 8064 terms: exact in f32 in any order
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        const float q = acc / 90.0f;
        if (lane == 0) *quality = q > 1.0f ? 1.0f : q;
    }
    float *lds_t = lds;            // [16][65]
    float *lds_c = lds + 16 * 65;  // [256]
    tail_fence();
#pragma unroll
    for (int i = 0; i < 16; i++) lds_t[i * 65 + lane] = a.t[i];
    tail_fence();

    // ---- DCT pass 2: C[i][j] = sum_k T[i][k] * D[j][k]; this lane: j = lane & 15, i = (lane >> 4) + 4m
    float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int jj = lane & 15, i0 = lane >> 4;
#pragma unroll 8
    for (int k = 0; k < 64; k++) {
        const float dj = __uint_as_float(c_dct_t.v[k * 16 + jj]);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const float p = lds_t[(i0 + 4 * m) * 65 + k] * dj;
            c[m] = c[m] + p;
        }
    }
    if (coeffs) {
#pragma unroll
        for (int m = 0; m < 4; m++) coeffs[lane + 64 * m] = c[m];
    }
    hashes_from_coeffs(c, lds_c, lane, hash32, dihedral);
}

// Whole-buffer form (generic path): lane j holds column j of the 64x64 buffer in registers.
__device__ __forceinline__ void pdq_tail(const float (&b)[64], float *lds, int lane, uint8_t *hash32, float *quality,
                                         float *coeffs, uint8_t *dihedral)
{
    TailAcc a;
    tail_init(a);
#pragma unroll
    for (int k = 0; k < 64; k++) tail_row(a, b[k], k, lane, quality != nullptr);
    tail_finish(a, lds, lane, hash32, quality, coeffs, dihedral);
}

}  // namespace rph
