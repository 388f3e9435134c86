/*
 * oracle/synth_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU statement of the synthetic workload generators of SURVEY.md section 8(d).
 * These are this repo's own definitions (the reference has no generator); the
 * device generators in rupphash_amd/csrc/synth_kernels.hip implement the same
 * integer functions and tests/ check them byte for byte against this file.
 *
 *   images : 16x16 grid of random block colours + +-8 noise per byte; image k
 *            with k % 1000 == 999 reuses the blocks of image k-1 (near duplicate)
 *   hashes : 4 x splitmix64 words per hash, then n_clusters injected 5-member
 *            clusters (mask popcounts 0,1,2,8,16) and one "2 bits in every
 *            16-bit chunk" pair at distance exactly 32
 */
#include <stdint.h>
#include <string.h>

#define GOLDEN64 0x9E3779B97F4A7C15ull

static inline uint64_t splitmix64(uint64_t x)
{
    x += GOLDEN64;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline uint32_t mix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

uint64_t rph_ref_splitmix64(uint64_t x) { return splitmix64(x); }
uint32_t rph_ref_mix32(uint32_t x) { return mix32(x); }

/* One byte of synthetic image k (global index), channel-interleaved RGB8. */
static inline uint8_t synth_byte(uint32_t seed, uint64_t k, int w, int h, int x, int y, int c)
{
    uint64_t base = (k % 1000ull == 999ull) ? k - 1 : k;
    uint32_t bx = (uint32_t)((x * 16) / w), by = (uint32_t)((y * 16) / h);
    uint32_t bkey = mix32(seed ^ mix32((uint32_t)base * 0x9E3779B1u + (uint32_t)(base >> 32) + 0x1234567u));
    uint32_t col = mix32(bkey + ((by * 16u + bx) * 3u + (uint32_t)c) * 0x85EBCA77u) & 0xFFu;
    uint32_t idx = ((uint32_t)y * (uint32_t)w + (uint32_t)x) * 3u + (uint32_t)c;
    uint32_t nz = mix32((seed ^ 0x5EED5EEDu) + (uint32_t)k * 0x9E3779B1u + (uint32_t)(k >> 32) * 0x7FEB352Du +
                        idx * 0xC2B2AE3Du) & 15u;
    int v = (int)col + (int)nz - 8;
    if (v < 0) v = 0;
    if (v > 255) v = 255;
    return (uint8_t)v;
}

/* n images starting at global index first_k, packed RGB8, row stride 3*w. */
void rph_ref_synth_images(uint8_t *out, uint64_t first_k, uint32_t n, int w, int h, uint32_t seed)
{
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *img = out + (size_t)i * (size_t)w * (size_t)h * 3;
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                for (int c = 0; c < 3; c++)
                    img[((size_t)y * w + x) * 3 + c] = synth_byte(seed, first_k + i, w, h, x, y, c);
    }
}

static inline void hash_words(uint64_t seed, uint64_t i, uint64_t wds[4])
{
    for (int w = 0; w < 4; w++) wds[w] = splitmix64(seed + (4 * i + (uint64_t)w) * GOLDEN64);
}

/* cluster member (c, j): index and value */
static inline uint64_t cluster_index(uint64_t n_total, uint64_t c, int j)
{
    uint64_t M = n_total / 5;
    uint64_t mul = (M % 7919ull == 0) ? 1 : 7919ull;
    return (uint64_t)j * M + (c * mul) % M;
}

static inline void cluster_value(uint64_t seed, uint64_t c, int j, int special, uint64_t wds[4])
{
    static const int POP[5] = {0, 1, 2, 8, 16};
    hash_words(seed ^ 0xC1A57E55EEDull, c, wds);
    if (special) {
        if (j == 1)
            for (int w = 0; w < 4; w++) wds[w] ^= 0x0003000300030003ull; /* 2 bits per 16-bit chunk */
        return;
    }
    for (int t = 0; t < POP[j]; t++) {
        uint32_t pos = (uint32_t)((c * 31 + (uint64_t)j * 11 + (uint64_t)t * 37) & 255);
        wds[pos >> 6] ^= 1ull << (pos & 63);
    }
}

/*
 * Hashes [first, first+count) of a synthetic set of n_total hashes (little
 * endian u64 x 4 per hash).  n_clusters + 1 <= n_total/5 is required (the +1 is
 * the special distance-32 pair, cluster id n_clusters, members j = 0, 1).
 */
void rph_ref_synth_hashes(uint8_t *out, uint64_t first, uint64_t count, uint64_t n_total, uint64_t seed,
                          uint64_t n_clusters)
{
    for (uint64_t i = 0; i < count; i++) {
        uint64_t w[4];
        hash_words(seed, first + i, w);
        memcpy(out + i * 32, w, 32);
    }
    if (n_total < 5) return;
    for (uint64_t c = 0; c <= n_clusters; c++) {
        int special = (c == n_clusters);
        int members = special ? 2 : 5;
        for (int j = 0; j < members; j++) {
            uint64_t idx = cluster_index(n_total, c, j);
            if (idx < first || idx >= first + count) continue;
            uint64_t w[4];
            cluster_value(seed, c, j, special, w);
            memcpy(out + (idx - first) * 32, w, 32);
        }
    }
}

uint64_t rph_ref_synth_cluster_index(uint64_t n_total, uint64_t c, int j) { return cluster_index(n_total, c, j); }
