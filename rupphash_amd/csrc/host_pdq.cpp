// host_pdq.cpp -- host-scalar halves of the PDQ interface: the per-file calls the reference makes on
// features it already holds, for which a kernel launch + PCIe round trip per call would be absurd, and
// the cache record codecs either side of the hash path.
//   PdqFeatures::to_hash                 /root/reference/src/pdqhash.rs:59-61   (scanner.rs:1412, once per file)
//   PdqFeatures::generate_dihedral_hashes /root/reference/src/pdqhash.rs:71-87  (scanner.rs:1622, :2223)
//   hash_db / coeff_db value layouts     /root/reference/src/db.rs:1200-1231 (write), :683-696, :742-755 (read)
// Compare + bit operations only: nothing here touches the GPU.  The batch kernels (pdq_tail.hpp) compute
// the same bits; tests/test_host_pdq.py checks both against the oracle.
#include <algorithm>
#include <cstring>

#include "rph_internal.h"

namespace {

constexpr int N = 16;

// f32::total_cmp as an unsigned ascending key
inline uint32_t total_key(float f)
{
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
inline float key_to_float(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

// apply_sign (pdqhash.rs:127-137): row r is DCT frequency r + 1, a mirror negates the odd frequencies = the even indices
inline float apply_sign(float v, int r, int c, bool neg_rows, bool neg_cols)
{
    const bool flip_r = neg_rows && ((r + 1) % 2 == 1);
    const bool flip_c = neg_cols && ((c + 1) % 2 == 1);
    return (flip_r != flip_c) ? -v : v;
}

// bit_rows (pdqhash.rs:91-106) with coefficient_median (:116-124): 128th smallest of 256 under total_cmp, compare with `>`
void bit_rows(const float *coeffs, bool neg_rows, bool neg_cols, uint16_t (&rows)[N])
{
    float v[256];
    uint32_t keys[256];
    for (int idx = 0; idx < 256; idx++) {
        v[idx] = apply_sign(coeffs[idx], idx / N, idx % N, neg_rows, neg_cols);
        keys[idx] = total_key(v[idx]);
    }
    std::nth_element(keys, keys + 127, keys + 256);  // mid = (256 - 1) / 2
    const float median = key_to_float(keys[127]);
    for (int r = 0; r < N; r++) {
        uint16_t bits = 0;
        for (int c = 0; c < N; c++)
            if (v[r * N + c] > median) bits |= (uint16_t)(1u << c);
        rows[r] = bits;
    }
}

// transpose_bit_rows (pdqhash.rs:140-151)
void transpose_rows(const uint16_t (&in)[N], uint16_t (&out)[N])
{
    for (int r = 0; r < N; r++) out[r] = 0;
    for (int r = 0; r < N; r++)
        for (int c = 0; c < N; c++)
            if (in[r] & (1u << c)) out[c] |= (uint16_t)(1u << r);
}

// pack_bit_rows (pdqhash.rs:155-162): row r -> bytes 31 - 2r (low) and 30 - 2r (high)
void pack_rows(const uint16_t (&rows)[N], uint8_t *hash)
{
    for (int r = 0; r < N; r++) {
        hash[32 - 2 * r - 1] = (uint8_t)(rows[r] & 0xFF);
        hash[32 - 2 * r - 2] = (uint8_t)(rows[r] >> 8);
    }
}

// postcard varint(usize): LEB128, at most 10 bytes
size_t put_varint(uint64_t v, uint8_t *out)
{
    size_t n = 0;
    while (v >= 0x80) {
        out[n++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    out[n++] = (uint8_t)v;
    return n;
}
// returns bytes consumed, 0 on a truncated or overlong varint
size_t get_varint(const uint8_t *p, size_t len, uint64_t *v)
{
    uint64_t acc = 0;
    for (size_t i = 0; i < len && i < 10; i++) {
        const uint8_t b = p[i];
        if (i == 9 && b > 1) return 0;  // would overflow 64 bits (postcard: DeserializeBadVarint)
        acc |= (uint64_t)(b & 0x7F) << (7 * i);
        if (!(b & 0x80)) {
            *v = acc;
            return i + 1;
        }
    }
    return 0;
}

}  // namespace

extern "C" {

void rph_pdq_to_hash(const float *coeffs256, uint8_t *hash32_out)
{
    uint16_t rows[N];
    bit_rows(coeffs256, false, false, rows);
    pack_rows(rows, hash32_out);
}

void rph_pdq_dihedral_one(const float *coeffs256, uint8_t *out8x32)
{
    uint16_t id[N], nc[N], nr[N], nb[N], t[N];
    bit_rows(coeffs256, false, false, id);
    bit_rows(coeffs256, false, true, nc);
    bit_rows(coeffs256, true, false, nr);
    bit_rows(coeffs256, true, true, nb);
    pack_rows(id, out8x32 + 0 * 32);  // identity
    transpose_rows(nr, t);
    pack_rows(t, out8x32 + 1 * 32);   // rot90
    pack_rows(nb, out8x32 + 2 * 32);  // rot180
    transpose_rows(nc, t);
    pack_rows(t, out8x32 + 3 * 32);   // rot270
    pack_rows(nc, out8x32 + 4 * 32);  // mirror-x
    pack_rows(nr, out8x32 + 5 * 32);  // mirror-y
    transpose_rows(id, t);
    pack_rows(t, out8x32 + 6 * 32);   // transpose
    transpose_rows(nb, t);
    pack_rows(t, out8x32 + 7 * 32);   // anti-transpose
}

// ---- hash_db value: [PDQ_ALGO_VERSION || 32-byte hash]  (db.rs:1200-1211 write, :683-696 read) ----
void rph_hash_record_encode(const uint8_t *hash32, uint8_t *out33)
{
    out33[0] = RPH_PDQ_ALGO_VERSION;
    std::memcpy(out33 + 1, hash32, 32);
}

int rph_hash_record_decode(const uint8_t *rec, size_t len, uint8_t *hash32_out)
{
    // `Some((&PDQ_ALGO_VERSION, rest)) if rest.len() == 32` else None: another version or length is a miss, not an error
    if (!rec || !hash32_out || len != RPH_HASH_RECORD_BYTES || rec[0] != RPH_PDQ_ALGO_VERSION) return 0;
    std::memcpy(hash32_out, rec + 1, 32);
    return 1;
}

// ---- coeff_db value: [PDQ_ALGO_VERSION || postcard(CachedCoefficients { coefficients: Vec<f32> })]
//      = [2 || varint(len) || len x f32 little-endian]   (db.rs:217-230, :1221-1231 write, :742-755 read) ----
size_t rph_coeff_record_size(size_t n_coeffs)
{
    uint8_t tmp[10];
    return 1 + put_varint(n_coeffs, tmp) + 4 * n_coeffs;
}

size_t rph_coeff_record_encode(const float *coeffs, size_t n_coeffs, uint8_t *out, size_t cap)
{
    const size_t need = rph_coeff_record_size(n_coeffs);
    if (!out || cap < need || (!coeffs && n_coeffs)) return need;
    size_t at = 0;
    out[at++] = RPH_PDQ_ALGO_VERSION;
    at += put_varint(n_coeffs, out + at);
    for (size_t i = 0; i < n_coeffs; i++) {  // f32::to_le_bytes
        uint32_t b;
        std::memcpy(&b, &coeffs[i], 4);
        out[at++] = (uint8_t)b;
        out[at++] = (uint8_t)(b >> 8);
        out[at++] = (uint8_t)(b >> 16);
        out[at++] = (uint8_t)(b >> 24);
    }
    return need;
}

int rph_coeff_record_decode(const uint8_t *rec, size_t len, float *coeffs_out, size_t cap, size_t *n_out)
{
    if (n_out) *n_out = 0;
    if (!rec || len == 0 || rec[0] != RPH_PDQ_ALGO_VERSION) return 0;  // empty or older algorithm version: absent, not corrupt
    uint64_t n = 0;
    const size_t used = get_varint(rec + 1, len - 1, &n);
    if (used == 0 || n > (len - 1 - used) / 4) {  // postcard error -> lmdb::Error::Corrupted (trailing bytes are ignored by from_bytes)
        rph_set_error("coeff record: truncated or malformed postcard payload (%zu bytes)", len);
        return RPH_ERR_INVALID_ARG;
    }
    if (n_out) *n_out = (size_t)n;
    if (n > cap || (!coeffs_out && n)) {
        rph_set_error("coeff record: %llu coefficients, capacity %zu", (unsigned long long)n, cap);
        return RPH_ERR_CAPACITY;
    }
    const uint8_t *p = rec + 1 + used;
    for (uint64_t i = 0; i < n; i++, p += 4) {
        const uint32_t b = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        std::memcpy(&coeffs_out[i], &b, 4);
    }
    return 1;
}

// Bulk forms for import/export between the engine's flat arrays and a phdupes cache: n fixed-size records each.
void rph_hash_records_encode(const uint8_t *hashes32, size_t n, uint8_t *out33n)
{
    for (size_t i = 0; i < n; i++) rph_hash_record_encode(hashes32 + i * 32, out33n + i * RPH_HASH_RECORD_BYTES);
}

size_t rph_hash_records_decode(const uint8_t *recs33n, size_t n, uint8_t *hashes32_out, uint8_t *present_out)
{
    size_t hits = 0;
    for (size_t i = 0; i < n; i++) {
        const int ok = rph_hash_record_decode(recs33n + i * RPH_HASH_RECORD_BYTES, RPH_HASH_RECORD_BYTES, hashes32_out + i * 32);
        if (!ok) std::memset(hashes32_out + i * 32, 0, 32);
        if (present_out) present_out[i] = (uint8_t)ok;
        hits += (size_t)ok;
    }
    return hits;
}

}  // extern "C"
