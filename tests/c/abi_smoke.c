/* Plain C (C99, -pedantic) consumer of include/rupphash.h: proves the header is a C header and that every declared entry point
 * links.  With a GPU it hashes one synthetic image pair and groups three hashes; without one rph_init must fail loudly. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rupphash.h"

int main(void)
{
    rph_ctx *ctx = NULL;
    int rc = rph_init(0, &ctx);
    if (rc != RPH_OK) {
        printf("no device: %s (%s)\n", rph_status_string(rc), rph_last_error());
        return rc == RPH_ERR_NO_DEVICE && ctx == NULL ? 3 : 4; /* 3 = expected outcome on a CPU-only host */
    }
    {
        enum { W = 96, H = 80 };
        static uint8_t img[2][H][W][3];
        uint8_t hash[2][32], valid[2];
        float q[2];
        int x, y, k;
        for (k = 0; k < 2; k++)
            for (y = 0; y < H; y++)
                for (x = 0; x < W; x++) {
                    img[k][y][x][0] = (uint8_t)(x * 2 + k);
                    img[k][y][x][1] = (uint8_t)(y * 3);
                    img[k][y][x][2] = (uint8_t)((x ^ y) & 0xFF);
                }
        rc = rph_pdq_hash_batch(ctx, &img[0][0][0][0], 2, W, H, 3, W * 3, (size_t)W * H * 3, &hash[0][0], q, NULL, NULL, valid);
        if (rc != RPH_OK || !valid[0] || !valid[1]) { printf("hash failed: %s\n", rph_last_error()); return 5; }
        printf("distance between the two images: %u, quality %.2f %.2f\n", rph_hamming_distance256(hash[0], hash[1]), q[0], q[1]);
        {
            uint8_t three[3][32];
            uint32_t members[3], offsets[3], ng = 0;
            memcpy(three[0], hash[0], 32);
            memcpy(three[1], hash[1], 32);
            memset(three[2], 0xFF, 32);
            rc = rph_find_groups256(ctx, &three[0][0], 3, 31, members, offsets, &ng);
            if (rc != RPH_OK) { printf("grouping failed: %s\n", rph_last_error()); return 6; }
            printf("groups at distance <= 31: %u\n", ng);
        }
        /* host-scalar forms (no context): PdqFeatures::to_hash of stored coefficients, cache records */
        {
            float coeffs[256];
            uint8_t h2[32], rec[RPH_HASH_RECORD_BYTES], back[32];
            int i;
            for (i = 0; i < 256; i++) coeffs[i] = (float)((i * 37) % 101) - 50.0f;
            rph_pdq_to_hash(coeffs, h2);
            rph_hash_record_encode(h2, rec);
            if (rec[0] != RPH_PDQ_ALGO_VERSION || rph_hash_record_decode(rec, sizeof rec, back) != 1 || memcmp(back, h2, 32) != 0) return 8;
        }
    }
    if (rph_shutdown(ctx) != RPH_OK) return 7;
    /* Several GPUs under one process (here: the devices listed, one): the same groups as the single-context call */
    {
        enum { N = 600 };
        static uint8_t hashes[N][32];
        static float coeffs[N][256];
        static int32_t quality[N];
        static uint32_t m1[N], o1[N / 2 + 2], m2[N], o2[N / 2 + 2];
        uint32_t g1 = 0, g2 = 0, x = 2463534242u;
        uint64_t c1 = 0, c2 = 0;
        rph_multi *multi = NULL;
        int i, k;
        for (i = 0; i < N; i++) {
            for (k = 0; k < 256; k++) {
                x ^= x << 13; x ^= x >> 17; x ^= x << 5;
                coeffs[i][k] = (i % 3 == 1) ? coeffs[i - 1][k] + (float)(x % 7) * 0.01f : (float)(x % 20001) / 100.0f - 100.0f;
            }
            quality[i] = (i % 50 == 7) ? 10 : 100;
        }
        rc = rph_multi_init(NULL, 1, &multi);
        if (rc != RPH_OK) { printf("rph_multi_init failed: %s\n", rph_last_error()); return 9; }
        if (rph_multi_size(multi) != 1 || rph_multi_ctx(multi, 0) == NULL) return 10;
        for (i = 0; i < N; i++) rph_pdq_to_hash(coeffs[i], hashes[i]);
        rc = rph_group_files_pdq(rph_multi_ctx(multi, 0), &hashes[0][0], &coeffs[0][0], NULL, quality, N, 40, m1, o1, &g1, &c1);
        if (rc != RPH_OK) { printf("rph_group_files_pdq failed: %s\n", rph_last_error()); return 11; }
        rc = rph_multi_group_files_pdq(multi, &hashes[0][0], &coeffs[0][0], NULL, quality, N, 40, m2, o2, &g2, &c2);
        if (rc != RPH_OK) { printf("rph_multi_group_files_pdq failed: %s\n", rph_last_error()); return 12; }
        printf("multi (%d device): %u groups, %lu comparisons; single context: %u groups, %lu comparisons\n", rph_multi_size(multi), g2,
               (unsigned long)c2, g1, (unsigned long)c1);
        if (g1 != g2 || c1 != c2 || g1 == 0 || memcmp(m1, m2, sizeof(uint32_t) * o1[g1]) != 0 || memcmp(o1, o2, sizeof(uint32_t) * (g1 + 1)) != 0) return 13;
        if (rph_multi_shutdown(multi) != RPH_OK) return 14;
    }
    return 0;
}
