/*
 * oracle/hamming_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's Hamming / multi-index-hashing path:
 *   src/hamminghash.rs  (HammingHash, MIHIndex, SparseBitSet, find_groups)
 *   src/scanner.rs:1588-1823 (is_low_pdq_quality, PdqStrategy, group_files_generic)
 *   src/phash.rs:137-255 (64-bit pHash bit operations)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PINNING STATUS: pinned by the reference's own tests re-expressed in
 * tests/test_oracle_hamming.py -- hamminghash.rs:283-332 (u64 {0,0xFFF} at 12,
 * [u8;32] 30 bits at 30), :336-412 (injected 5-cluster among random u64,
 * max_dist 5), and NOTES.txt:64-67 (pHash 0xdeb1e20c136f983c -> rotation
 * invariant 0x8b1bb7a646c5cd96) for the u64 bit operations.
 *
 * The reference runs the per-query loop under rayon; results are order
 * independent (collect() of an indexed parallel iterator keeps index order), so
 * this serial restatement yields the same vectors.  Threaded timing lives in
 * oracle/bench_ref.c.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

#define KIND_U64 0   /* impl HammingHash for u64:      hamminghash.rs:23-41 */
#define KIND_PDQ 1   /* impl HammingHash for [u8; 32]: hamminghash.rs:44-63 */

static inline size_t hsize(int kind) { return kind == KIND_U64 ? 8 : 32; }
static inline int num_chunks(int kind) { return kind == KIND_U64 ? 8 : 16; }
static inline int num_buckets(int kind) { return kind == KIND_U64 ? 256 : 65536; }
static inline int bit_width_per_chunk(int kind) { return kind == KIND_U64 ? 8 : 16; }

/* get_chunk: hamminghash.rs:29-31 (u64: byte k), :50-53 ([u8;32]: LE u16 of bytes 2k,2k+1) */
static inline uint16_t get_chunk(int kind, const uint8_t *h, int k)
{
    if (kind == KIND_U64) {
        uint64_t v;
        memcpy(&v, h, 8);
        return (uint16_t)((v >> (k * 8)) & 0xFF);
    }
    return (uint16_t)(h[2 * k] | ((uint16_t)h[2 * k + 1] << 8));
}

/* hamming_distance: hamminghash.rs:34-36, :56-58 */
uint32_t rph_ref_hamming64(uint64_t a, uint64_t b) { return (uint32_t)__builtin_popcountll(a ^ b); }
uint32_t rph_ref_hamming256(const uint8_t *a, const uint8_t *b)
{
    uint32_t d = 0;
    for (int i = 0; i < 32; i++) d += (uint32_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}
static inline uint32_t hdist(int kind, const uint8_t *a, const uint8_t *b)
{
    if (kind == KIND_U64) {
        uint64_t x, y;
        memcpy(&x, a, 8);
        memcpy(&y, b, 8);
        return rph_ref_hamming64(x, y);
    }
    return rph_ref_hamming256(a, b);
}

/* ---- MIHIndex (CSR): hamminghash.rs:82-149 ---- */

mih_t *rph_ref_mih_new(int kind, const uint8_t *hashes, uint32_t n)
{
    mih_t *m = (mih_t *)calloc(1, sizeof(mih_t));
    size_t hs = hsize(kind);
    int nc = num_chunks(kind), nb = num_buckets(kind);
    size_t total_buckets = (size_t)nc * (size_t)nb;
    m->kind = kind;
    m->n = n;
    m->db_hashes = (uint8_t *)malloc((size_t)n * hs + 1);
    memcpy(m->db_hashes, hashes, (size_t)n * hs);
    m->offsets = (uint32_t *)calloc(total_buckets + 1, sizeof(uint32_t));
    /* count phase */
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < nc; k++) {
            size_t flat = (size_t)k * nb + get_chunk(kind, hashes + (size_t)i * hs, k);
            m->offsets[flat + 1] += 1;
        }
    /* prefix sum */
    for (size_t i = 1; i <= total_buckets; i++) m->offsets[i] += m->offsets[i - 1];
    size_t nvals = m->offsets[total_buckets];
    m->values = (uint32_t *)calloc(nvals + 1, sizeof(uint32_t));
    /* fill phase (ascending id within each bucket) */
    uint32_t *cursor = (uint32_t *)malloc((total_buckets + 1) * sizeof(uint32_t));
    memcpy(cursor, m->offsets, (total_buckets + 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < nc; k++) {
            size_t flat = (size_t)k * nb + get_chunk(kind, hashes + (size_t)i * hs, k);
            m->values[cursor[flat]++] = i;
        }
    free(cursor);
    return m;
}

void rph_ref_mih_free(mih_t *m)
{
    if (!m) return;
    free(m->db_hashes);
    free(m->offsets);
    free(m->values);
    free(m);
}

/* bucket(): hamminghash.rs:133-138.  Returns length, *ids -> first dense id. */
uint32_t rph_ref_mih_bucket(const mih_t *m, int chunk, uint16_t value, const uint32_t **ids)
{
    size_t flat = (size_t)chunk * num_buckets(m->kind) + value;
    *ids = m->values + m->offsets[flat];
    return m->offsets[flat + 1] - m->offsets[flat];
}
const uint32_t *rph_ref_mih_offsets(const mih_t *m) { return m->offsets; }
const uint32_t *rph_ref_mih_values(const mih_t *m) { return m->values; }
uint32_t rph_ref_mih_len(const mih_t *m) { return m->n; }

/* ---- SparseBitSet: hamminghash.rs:152-189 ---- */

static void sbs_init(sbs_t *s, size_t size)
{
    s->data = (uint64_t *)calloc((size + 63) / 64 + 1, sizeof(uint64_t));
    s->cap = 512;
    s->dirty = (size_t *)malloc(s->cap * sizeof(size_t));
    s->ndirty = 0;
}
static void sbs_free(sbs_t *s) { free(s->data); free(s->dirty); }
void rph_ref_sbs_init(sbs_t *s, size_t size) { sbs_init(s, size); }
void rph_ref_sbs_destroy(sbs_t *s) { sbs_free(s); }
/* set(): returns was_set */
static inline int sbs_set(sbs_t *s, size_t idx)
{
    size_t w = idx / 64;
    uint64_t mask = 1ull << (idx % 64);
    int was = (s->data[w] & mask) != 0;
    if (!was) {
        if (s->data[w] == 0) {
            if (s->ndirty == s->cap) {
                s->cap *= 2;
                s->dirty = (size_t *)realloc(s->dirty, s->cap * sizeof(size_t));
            }
            s->dirty[s->ndirty++] = w;
        }
        s->data[w] |= mask;
    }
    return was;
}
static inline void sbs_clear(sbs_t *s)
{
    for (size_t i = 0; i < s->ndirty; i++) s->data[s->dirty[i]] = 0;
    s->ndirty = 0;
}

/* test hooks for SparseBitSet semantics */
void *rph_ref_sbs_new(size_t size) { sbs_t *s = (sbs_t *)malloc(sizeof(sbs_t)); sbs_init(s, size); return s; }
int rph_ref_sbs_set(void *s, size_t idx) { return sbs_set((sbs_t *)s, idx); }
void rph_ref_sbs_clear(void *s) { sbs_clear((sbs_t *)s); }
void rph_ref_sbs_free(void *s) { sbs_free((sbs_t *)s); free(s); }

/* growable u32 vector */
static void vpush(vec_t *v, uint32_t x)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 64;
        v->p = (uint32_t *)realloc(v->p, v->cap * sizeof(uint32_t));
    }
    v->p[v->n++] = x;
}

/* ---- adjacency of one query: the map closure of find_groups, hamminghash.rs:200-241 ---- */
void rph_ref_query_adjacency(const mih_t *m, uint32_t i, uint32_t max_dist, sbs_t *visited, vec_t *results)
{
    int kind = m->kind;
    size_t hs = hsize(kind);
    int nc = num_chunks(kind);
    uint32_t chunk_tolerance = max_dist / (uint32_t)nc;
    int bits_per_chunk = bit_width_per_chunk(kind);
    const uint8_t *q = m->db_hashes + (size_t)i * hs;

    sbs_clear(visited);
    results->n = 0;
    for (int k = 0; k < nc; k++) {
        uint16_t q_chunk = get_chunk(kind, q, k);
        int nprobe = 1 + (chunk_tolerance >= 1 ? bits_per_chunk : 0);
        for (int p = 0; p < nprobe; p++) {
            uint16_t val = p == 0 ? q_chunk : (uint16_t)(q_chunk ^ (1u << (p - 1)));
            if (kind == KIND_U64) val &= 0xFF; /* chunks are 8-bit; flips stay within 8 bits */
            const uint32_t *ids;
            uint32_t len = rph_ref_mih_bucket(m, k, val, &ids);
            for (uint32_t t = 0; t < len; t++) {
                uint32_t d = ids[t];
                if (d == i) continue;
                if (sbs_set(visited, d)) continue;
                if (hdist(kind, q, m->db_hashes + (size_t)d * hs) <= max_dist) vpush(results, d);
            }
        }
    }
}

/* Adjacency list of a single query (test hook, also used by bench_ref.c). */
uint32_t rph_ref_query(const mih_t *m, uint32_t i, uint32_t max_dist, uint32_t *out, uint32_t cap)
{
    sbs_t vis;
    vec_t res = {0};
    sbs_init(&vis, m->n);
    rph_ref_query_adjacency(m, i, max_dist, &vis, &res);
    uint32_t cnt = (uint32_t)res.n;
    for (uint32_t t = 0; t < cnt && t < cap; t++) out[t] = res.p[t];
    free(res.p);
    sbs_free(&vis);
    return cnt;
}

/* ---- find_groups: hamminghash.rs:191-271 ----
 * Output: *members (concatenated groups), *offsets (n_groups+1), returns n_groups.
 * Caller frees both with rph_ref_free. */
uint32_t rph_ref_find_groups(const mih_t *m, uint32_t max_dist, uint32_t **members_out,
                             uint32_t **offsets_out)
{
    uint32_t n = m->n;
    vec_t *adj = (vec_t *)calloc(n ? n : 1, sizeof(vec_t));
    sbs_t vis;
    vec_t res = {0};
    sbs_init(&vis, n);
    for (uint32_t i = 0; i < n; i++) {
        rph_ref_query_adjacency(m, i, max_dist, &vis, &res);
        if (res.n) {
            adj[i].p = (uint32_t *)malloc(res.n * sizeof(uint32_t));
            memcpy(adj[i].p, res.p, res.n * sizeof(uint32_t));
            adj[i].n = adj[i].cap = res.n;
        }
    }
    free(res.p);
    sbs_free(&vis);

    uint32_t ng = rph_ref_greedy_cluster(n, adj, members_out, offsets_out);
    for (uint32_t i = 0; i < n; i++) free(adj[i].p);
    free(adj);
    return ng;
}

/* greedy clustering: hamminghash.rs:245-268 */
uint32_t rph_ref_greedy_cluster(uint32_t n, vec_t *adj, uint32_t **members_out, uint32_t **offsets_out)
{
    uint8_t *visited = (uint8_t *)calloc(n ? n : 1, 1);
    vec_t members = {0}, offsets = {0};
    vpush(&offsets, 0);
    for (uint32_t i = 0; i < n; i++) {
        if (visited[i] || adj[i].n == 0) continue;
        size_t start = members.n;
        vpush(&members, i);
        visited[i] = 1;
        for (size_t t = 0; t < adj[i].n; t++) {
            uint32_t nb = adj[i].p[t];
            if (!visited[nb]) {
                visited[nb] = 1;
                vpush(&members, nb);
            }
        }
        if (members.n - start > 1)
            vpush(&offsets, (uint32_t)members.n);
        else
            members.n = start; /* group of one is dropped (its members stay visited) */
    }
    free(visited);
    *members_out = members.p ? members.p : (uint32_t *)calloc(1, sizeof(uint32_t));
    *offsets_out = offsets.p;
    return (uint32_t)offsets.n - 1;
}

void rph_ref_free(void *p) { free(p); }

/* ---- is_low_pdq_quality: scanner.rs:1588-1594 (quality < 0 encodes None) ---- */
int rph_ref_is_low_pdq_quality(int quality) { return quality >= 0 && quality < 50; }

/* ---- group_files_generic with PdqStrategy: scanner.rs:1607-1637, 1640-1823 ----
 * Inputs (every file has a hash; dense id == sparse id):
 *   hashes   [n][32]   file.pdqhash
 *   variants [n][8][32] generate_dihedral_hashes() of file.pdq_features, or NULL
 *   has_features [n]   (nullable => all 1 when variants != NULL, all 0 otherwise)
 *   quality  [n] int   stored quality 0..100, or -1 for None (nullable => all None)
 * Outputs:
 *   edges (i,j) pairs in the reference's emission order (duplicates across
 *   variants included: comparison_count == n_edges, scanner.rs:1778),
 *   groups = connected components with > 1 member, members ascending, groups
 *   ordered by first member (the reference's HashMap order is unspecified).
 */
uint64_t rph_ref_group_pdq(const uint8_t *hashes, const uint8_t *variants, const uint8_t *has_features,
                           const int32_t *quality, uint32_t n, uint32_t similarity,
                           uint32_t **edges_out, uint32_t **members_out, uint32_t **offsets_out,
                           uint32_t *n_groups_out)
{
    const int kind = KIND_PDQ;
    const int nc = 16, bits = 16;
    mih_t *mih = rph_ref_mih_new(kind, hashes, n);
    uint8_t *low_conf = (uint8_t *)calloc(n ? n : 1, 1);
    for (uint32_t i = 0; i < n; i++) low_conf[i] = quality ? (uint8_t)rph_ref_is_low_pdq_quality(quality[i]) : 0;

    sbs_t visited;
    sbs_init(&visited, n);
    vec_t edges = {0};

    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *hash = hashes + (size_t)i * 32;
        uint8_t vbuf[8][32];
        int count;
        int feat = variants && (has_features ? has_features[i] : 1);
        if (feat) {
            memcpy(vbuf, variants + (size_t)i * 256, 256);
            count = 8;
        } else {
            memcpy(vbuf[0], hash, 32);
            count = 1;
        }
        uint32_t base_limit = low_conf[i] ? 0 : similarity;

        for (int v = 0; v < count; v++) {
            const uint8_t *variant = vbuf[v];
            sbs_clear(&visited);
            for (int k = 0; k < nc; k++) {
                uint16_t q_chunk = get_chunk(kind, variant, k);
#define CHECK_BUCKET(VAL)                                                                   \
    do {                                                                                    \
        const uint32_t *ids;                                                                \
        uint32_t len = rph_ref_mih_bucket(mih, k, (uint16_t)(VAL), &ids);                   \
        for (uint32_t t = 0; t < len; t++) {                                                \
            uint32_t cand = ids[t];                                                         \
            if (cand <= i || sbs_set(&visited, cand)) continue;                             \
            uint32_t limit = low_conf[cand] ? 0 : base_limit;                               \
            if (rph_ref_hamming256(variant, hashes + (size_t)cand * 32) <= limit) {         \
                vpush(&edges, i);                                                           \
                vpush(&edges, cand);                                                        \
            }                                                                               \
        }                                                                                   \
    } while (0)
                CHECK_BUCKET(q_chunk);                                   /* R=0 */
                if (similarity >= (uint32_t)nc)                          /* R=1 */
                    for (int a = 0; a < bits; a++) CHECK_BUCKET(q_chunk ^ (1u << a));
                if (similarity >= (uint32_t)(nc * 2))                    /* R=2 */
                    for (int a = 0; a < bits; a++)
                        for (int b = a + 1; b < bits; b++) CHECK_BUCKET(q_chunk ^ (1u << a) ^ (1u << b));
                if (similarity >= (uint32_t)(nc * 3))                    /* R=3 */
                    for (int a = 0; a < bits; a++)
                        for (int b = a + 1; b < bits; b++)
                            for (int c = b + 1; c < bits; c++)
                                CHECK_BUCKET(q_chunk ^ (1u << a) ^ (1u << b) ^ (1u << c));
#undef CHECK_BUCKET
            }
        }
    }
    sbs_free(&visited);
    uint64_t n_edges = edges.n / 2;

    /* union-find: scanner.rs:1781-1807 */
    uint32_t *parent = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) parent[i] = i;
#define FIND(X, ROOT)                                  \
    do {                                               \
        uint32_t root_ = (X);                          \
        while (root_ != parent[root_]) root_ = parent[root_]; \
        uint32_t cur_ = (X);                           \
        while (cur_ != root_) {                        \
            uint32_t nx_ = parent[cur_];               \
            parent[cur_] = root_;                      \
            cur_ = nx_;                                \
        }                                              \
        (ROOT) = root_;                                \
    } while (0)
    for (uint64_t e = 0; e < n_edges; e++) {
        uint32_t ri, rj;
        FIND(edges.p[2 * e], ri);
        FIND(edges.p[2 * e + 1], rj);
        if (ri != rj) parent[ri] = rj;
    }
    /* groups_map: scanner.rs:1809-1817.  A root is pushed only when it is not the
     * first member seen; a root is never the smallest member of a component of
     * size >= 2 (edges always point i<j and parent[find(i)] = find(j)), so every
     * member ends up listed.  We list members ascending and order groups by
     * first member. */
    uint32_t *root_of = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *count = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r;
        FIND(i, r);
        root_of[i] = r;
        count[r]++;
    }
#undef FIND
    uint32_t *slot = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t)); /* root -> group idx */
    for (uint32_t i = 0; i < n; i++) slot[i] = UINT32_MAX;
    vec_t offsets = {0};
    uint32_t ng = 0, total = 0;
    vpush(&offsets, 0);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = root_of[i];
        if (count[r] > 1 && slot[r] == UINT32_MAX) {
            slot[r] = ng++;
            total += count[r];
            vpush(&offsets, total);
        }
    }
    uint32_t *members = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    uint32_t *fill = (uint32_t *)calloc(ng ? ng : 1, sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) {
        uint32_t r = root_of[i];
        if (count[r] > 1) {
            uint32_t g = slot[r];
            members[offsets.p[g] + fill[g]++] = i;
        }
    }
    free(fill); free(slot); free(count); free(root_of); free(parent); free(low_conf);
    rph_ref_mih_free(mih);

    *edges_out = edges.p ? edges.p : (uint32_t *)calloc(2, sizeof(uint32_t));
    *members_out = members;
    *offsets_out = offsets.p;
    *n_groups_out = ng;
    return n_edges;
}

/* Brute-force all-pairs (i<j, d<=thr) for small n: used by tests to show that
 * the MIH paths above and the GPU sweep agree on the edge set.  Returns the
 * number of edges; writes up to cap triples (i, j, d). */
uint64_t rph_ref_all_pairs256(const uint8_t *hashes, uint32_t n, uint32_t thr, uint32_t *triples, uint64_t cap)
{
    uint64_t cnt = 0;
    for (uint32_t i = 0; i < n; i++)
        for (uint32_t j = i + 1; j < n; j++) {
            uint32_t d = rph_ref_hamming256(hashes + (size_t)i * 32, hashes + (size_t)j * 32);
            if (d <= thr) {
                if (cnt < cap) {
                    triples[3 * cnt] = i;
                    triples[3 * cnt + 1] = j;
                    triples[3 * cnt + 2] = d;
                }
                cnt++;
            }
        }
    return cnt;
}

/* ================= 64-bit pHash bit operations: phash.rs:137-255 ================= */

/* rotate_hash_90: phash.rs:150-171 */
uint64_t rph_ref_rotate_hash_90(uint64_t hash)
{
    uint64_t result = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            int src_idx = 8 * y + x;
            int dst_x = y, dst_y = x;
            int dst_idx = 8 * dst_y + dst_x;
            uint64_t bit = (hash >> (63 - src_idx)) & 1;
            int flip = dst_x % 2 != 0;
            uint64_t final_bit = flip ? (bit ^ 1) : bit;
            result |= final_bit << (63 - dst_idx);
        }
    return result;
}
/* rotate_hash_180: phash.rs:175-188 */
uint64_t rph_ref_rotate_hash_180(uint64_t hash)
{
    uint64_t result = 0;
    for (int i = 0; i < 64; i++) {
        int x = i % 8, y = i / 8;
        int flip = (x + y) % 2 != 0;
        uint64_t bit = (hash >> (63 - i)) & 1;
        uint64_t final_bit = flip ? (bit ^ 1) : bit;
        result |= final_bit << (63 - i);
    }
    return result;
}
/* rotate_hash_270: phash.rs:191-212 */
uint64_t rph_ref_rotate_hash_270(uint64_t hash)
{
    uint64_t result = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) {
            int src_idx = 8 * y + x;
            int dst_x = y, dst_y = x;
            int dst_idx = 8 * dst_y + dst_x;
            uint64_t bit = (hash >> (63 - src_idx)) & 1;
            int flip = dst_y % 2 != 0;
            uint64_t final_bit = flip ? (bit ^ 1) : bit;
            result |= final_bit << (63 - dst_idx);
        }
    return result;
}
/* flip_hash_horizontal: phash.rs:220-230 */
uint64_t rph_ref_flip_hash_horizontal(uint64_t hash)
{
    uint64_t result = 0;
    for (int i = 0; i < 64; i++) {
        int x = i % 8;
        int flip = x % 2 != 0;
        uint64_t bit = (hash >> (63 - i)) & 1;
        uint64_t final_bit = flip ? (bit ^ 1) : bit;
        result |= final_bit << (63 - i);
    }
    return result;
}
/* calculate_rotation_invariant_hash: phash.rs:137-143 */
uint64_t rph_ref_rotation_invariant_hash(uint64_t hash)
{
    uint64_t h90 = rph_ref_rotate_hash_90(hash), h180 = rph_ref_rotate_hash_180(hash),
             h270 = rph_ref_rotate_hash_270(hash);
    uint64_t m = hash;
    if (h90 < m) m = h90;
    if (h180 < m) m = h180;
    if (h270 < m) m = h270;
    return m;
}
/* generate_dihedral_hashes(u64): phash.rs:242-255 */
void rph_ref_phash_dihedral(uint64_t hash, uint64_t out[8])
{
    uint64_t hf = rph_ref_flip_hash_horizontal(hash);
    out[0] = hash;
    out[1] = rph_ref_rotate_hash_90(hash);
    out[2] = rph_ref_rotate_hash_180(hash);
    out[3] = rph_ref_rotate_hash_270(hash);
    out[4] = hf;
    out[5] = rph_ref_rotate_hash_90(hf);
    out[6] = rph_ref_rotate_hash_180(hf);
    out[7] = rph_ref_rotate_hash_270(hf);
}
