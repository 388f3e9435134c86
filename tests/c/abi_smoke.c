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
    }
    return rph_shutdown(ctx) == RPH_OK ? 0 : 7;
}
