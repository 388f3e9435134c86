/*
 * oracle/pdq_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's PDQ path (Safari77/rupphash,
 * src/pdqhash.rs).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (rupphash_amd/) never
 * links, imports or calls it.
 *
 * Every function cites the reference lines it follows.  Arithmetic is f32 with
 * the reference's exact operation order; build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -fno-builtin-cosf
 * so no FMA contraction, no reassociation and no compile-time cosf folding.
 *
 * PINNING STATUS (see DESIGN.md "Oracle"):
 *   - the reference holds NO known-answer PDQ hash; what pins this file are the
 *     reference's own property tests, re-expressed in tests/test_oracle_pdq.py:
 *     pdqhash.rs:548-558 (to_hash / dihedral == naive), :561-570 (8 distinct
 *     variants), :583-628 (physically transformed 64x64 buffer == predicted
 *     dihedral slot, pins DCT + sign parity + slot order + packing),
 *     :631-639 (quality metric), :642-647 (target dimensions).
 *   - rows with no reference fixture (to_luma601, jarosz box filter, decimate)
 *     are line-by-line restatements: "parity unpinned" for those rows.
 *   - the >512 px pre-downsample (fast_image_resize 6.1.0, third party, source
 *     absent from /root/reference) is restated from its published algorithm in
 *     resize_ref.c: "parity unpinned" for every input with a side > 512 px.
 *   - the DCT table depends on the platform cosf (Rust f32::cos -> libm cosf);
 *     this oracle calls glibc cosf at run time like the reference does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIN_HASHABLE_DIM 5          /* pdqhash.rs:17 */
#define PDQ_NUM_JAROSZ_XY_PASSES 2  /* pdqhash.rs:18 */
#define DOWNSAMPLE_DIMS 512         /* pdqhash.rs:19 */
#define BUFFER_W_H 64               /* pdqhash.rs:20 */
#define DCT_OUTPUT_W_H 16           /* pdqhash.rs:21 */
#define DCT_OUTPUT_MATRIX_SIZE 256  /* pdqhash.rs:22 */
#define HASH_LENGTH 32              /* pdqhash.rs:23 */
#define JAROSZ_WINDOW_DIVISOR 64    /* pdqhash.rs:27 */
#define DCT_FREQ_OFFSET 1           /* pdqhash.rs:31 */

#define RPH_REF_OK 0
#define RPH_REF_TOO_SMALL 1     /* generate_pdq_features -> None, pdqhash.rs:167-169 */
#define RPH_REF_NEEDS_RESIZE 2  /* kept for ABI stability; no longer returned */

void rph_ref_resize_box_u8(const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst, uint32_t nw, uint32_t nh); /* resize_ref.c */
void rph_ref_target_dimensions(uint32_t w, uint32_t h, uint32_t max_dim, uint32_t *ow, uint32_t *oh);

/* ---- DCT matrix: pdqhash.rs:287-304 (compute_dct_matrix) ---- */
static float g_dct[DCT_OUTPUT_W_H][BUFFER_W_H];
static int g_dct_ready = 0;

void rph_ref_dct_matrix(float *out /* [16][64] row-major */)
{
    /* std::f32::consts::PI */
    const float PI_F = 3.14159274101257324219f;
    float num_cols = (float)BUFFER_W_H;
    float inv_sqrt_cols = 1.0f / sqrtf(num_cols);
    float sqrt_2 = sqrtf(2.0f);
    for (int i = 0; i < DCT_OUTPUT_W_H; i++) {
        float freq = (float)(i + DCT_FREQ_OFFSET);
        float normalization = (freq == 0.0f) ? inv_sqrt_cols : inv_sqrt_cols * sqrt_2;
        for (int j = 0; j < BUFFER_W_H; j++) {
            /* (PI * freq * (2.0 * j + 1.0)) / (2.0 * num_cols), left to right */
            float angle = (PI_F * freq * (2.0f * (float)j + 1.0f)) / (2.0f * num_cols);
            out[i * BUFFER_W_H + j] = normalization * cosf(angle);
        }
    }
}

static void ensure_dct(void)
{
    if (!g_dct_ready) {
        rph_ref_dct_matrix(&g_dct[0][0]);
        g_dct_ready = 1;
    }
}

/* ---- dct64_to_16: pdqhash.rs:306-336 ---- */
void rph_ref_dct64_to_16(const float *input /* [64][64] */, float *output /* [256] */)
{
    ensure_dct();
    static _Thread_local float intermediate[DCT_OUTPUT_W_H][BUFFER_W_H];
    memset(intermediate, 0, sizeof(intermediate));
    /* Pass 1: i outer, k middle, j inner; accumulate with += from 0.0 */
    for (int i = 0; i < DCT_OUTPUT_W_H; i++) {
        for (int k = 0; k < BUFFER_W_H; k++) {
            float coeff = g_dct[i][k];
            for (int j = 0; j < BUFFER_W_H; j++) {
                float p = coeff * input[k * BUFFER_W_H + j];
                intermediate[i][j] = intermediate[i][j] + p;
            }
        }
    }
    /* Pass 2: scalar sum from 0.0, k ascending */
    for (int i = 0; i < DCT_OUTPUT_W_H; i++) {
        for (int j = 0; j < DCT_OUTPUT_W_H; j++) {
            float sum = 0.0f;
            for (int k = 0; k < BUFFER_W_H; k++) {
                float p = intermediate[i][k] * g_dct[j][k];
                sum = sum + p;
            }
            output[i * DCT_OUTPUT_W_H + j] = sum;
        }
    }
}

/* ---- apply_sign: pdqhash.rs:127-137 ---- */
static float apply_sign(float v, int r, int c, int neg_rows, int neg_cols)
{
    int flip_r = neg_rows && ((r + DCT_FREQ_OFFSET) % 2 == 1);
    int flip_c = neg_cols && ((c + DCT_FREQ_OFFSET) % 2 == 1);
    return (flip_r ^ flip_c) ? -v : v;
}

/* f32::total_cmp order: map the bit pattern to a signed key (Rust std:
 * left ^= ((left >> 31) as u32 >> 1) as i32; compare as i32). */
static int32_t total_key(float f)
{
    int32_t b;
    memcpy(&b, &f, 4);
    b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
    return b;
}

static int cmp_total(const void *a, const void *b)
{
    int32_t ka = total_key(*(const float *)a), kb = total_key(*(const float *)b);
    return (ka > kb) - (ka < kb);
}

/* ---- coefficient_median: pdqhash.rs:116-124.  select_nth_unstable_by(mid,
 * total_cmp) returns the element a full sort would put at index mid; a sort is
 * used here (the reference's own naive test helper does the same, :470-473). */
static float coefficient_median(const float *coeffs, int neg_rows, int neg_cols)
{
    float buffer[DCT_OUTPUT_MATRIX_SIZE];
    for (int idx = 0; idx < DCT_OUTPUT_MATRIX_SIZE; idx++) {
        int r = idx / DCT_OUTPUT_W_H, c = idx % DCT_OUTPUT_W_H;
        buffer[idx] = apply_sign(coeffs[idx], r, c, neg_rows, neg_cols);
    }
    qsort(buffer, DCT_OUTPUT_MATRIX_SIZE, sizeof(float), cmp_total);
    int mid = (DCT_OUTPUT_MATRIX_SIZE - 1) / 2;
    return buffer[mid];
}

/* ---- bit_rows: pdqhash.rs:91-106 ---- */
static void bit_rows(const float *coeffs, int neg_rows, int neg_cols, uint16_t rows[16])
{
    float median = coefficient_median(coeffs, neg_rows, neg_cols);
    for (int r = 0; r < DCT_OUTPUT_W_H; r++) {
        int base = r * DCT_OUTPUT_W_H;
        uint16_t bits = 0;
        for (int c = 0; c < DCT_OUTPUT_W_H; c++) {
            if (apply_sign(coeffs[base + c], r, c, neg_rows, neg_cols) > median)
                bits |= (uint16_t)(1u << c);
        }
        rows[r] = bits;
    }
}

/* ---- transpose_bit_rows: pdqhash.rs:140-151 ---- */
static void transpose_bit_rows(const uint16_t rows[16], uint16_t out[16])
{
    memset(out, 0, 16 * sizeof(uint16_t));
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++)
            if (rows[r] & (1u << c))
                out[c] |= (uint16_t)(1u << r);
}

/* ---- pack_bit_rows: pdqhash.rs:155-162 ---- */
static void pack_bit_rows(const uint16_t rows[16], uint8_t hash[HASH_LENGTH])
{
    for (int r = 0; r < 16; r++) {
        hash[HASH_LENGTH - 2 * r - 1] = (uint8_t)(rows[r] & 0xFF);
        hash[HASH_LENGTH - 2 * r - 2] = (uint8_t)(rows[r] >> 8);
    }
}

/* ---- PdqFeatures::to_hash: pdqhash.rs:59-61 ---- */
void rph_ref_to_hash(const float *coeffs, uint8_t *hash)
{
    uint16_t rows[16];
    bit_rows(coeffs, 0, 0, rows);
    pack_bit_rows(rows, hash);
}

/* ---- PdqFeatures::generate_dihedral_hashes: pdqhash.rs:71-87 ---- */
void rph_ref_dihedral_hashes(const float *coeffs, uint8_t *out /* [8][32] */)
{
    uint16_t id[16], neg_cols[16], neg_rows[16], neg_both[16], t[16];
    bit_rows(coeffs, 0, 0, id);
    bit_rows(coeffs, 0, 1, neg_cols);
    bit_rows(coeffs, 1, 0, neg_rows);
    bit_rows(coeffs, 1, 1, neg_both);

    pack_bit_rows(id, out + 0 * 32);
    transpose_bit_rows(neg_rows, t);  pack_bit_rows(t, out + 1 * 32);
    pack_bit_rows(neg_both, out + 2 * 32);
    transpose_bit_rows(neg_cols, t);  pack_bit_rows(t, out + 3 * 32);
    pack_bit_rows(neg_cols, out + 4 * 32);
    pack_bit_rows(neg_rows, out + 5 * 32);
    transpose_bit_rows(id, t);        pack_bit_rows(t, out + 6 * 32);
    transpose_bit_rows(neg_both, t);  pack_bit_rows(t, out + 7 * 32);
}

/* ---- naive ground truth of the reference's own tests: pdqhash.rs:470-535 ----
 * (sort-based median, explicit 256-float transposes / flips).  Kept so that
 * tests/ can re-run fast_dihedral_matches_naive against this restatement. */
static void naive_to_hash(const float *f, uint8_t *hash)
{
    float buffer[256];
    memcpy(buffer, f, sizeof(buffer));
    qsort(buffer, 256, sizeof(float), cmp_total);
    float median = buffer[(256 - 1) / 2];
    for (int i = 0; i < HASH_LENGTH; i++) {
        uint8_t byte = 0;
        for (int j = 0; j < 8; j++)
            if (f[i * 8 + j] > median)
                byte |= (uint8_t)(1u << j);
        hash[HASH_LENGTH - i - 1] = byte;
    }
}
static void naive_transpose(const float *f, float *o)
{
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++)
            o[c * 16 + r] = f[r * 16 + c];
}
static void naive_flip_x(const float *f, float *o)
{
    memcpy(o, f, 256 * sizeof(float));
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++)
            if ((c + DCT_FREQ_OFFSET) % 2 != 0)
                o[r * 16 + c] = -o[r * 16 + c];
}
static void naive_flip_y(const float *f, float *o)
{
    memcpy(o, f, 256 * sizeof(float));
    for (int r = 0; r < 16; r++)
        if ((r + DCT_FREQ_OFFSET) % 2 != 0)
            for (int c = 0; c < 16; c++)
                o[r * 16 + c] = -o[r * 16 + c];
}
void rph_ref_naive_dihedral(const float *f, uint8_t *out /* [8][32] */)
{
    float a[256], b[256];
    naive_to_hash(f, out + 0 * 32);
    naive_transpose(f, a); naive_flip_x(a, b); naive_to_hash(b, out + 1 * 32);
    naive_flip_x(f, a); naive_flip_y(a, b); naive_to_hash(b, out + 2 * 32);
    naive_transpose(f, a); naive_flip_y(a, b); naive_to_hash(b, out + 3 * 32);
    naive_flip_x(f, a); naive_to_hash(a, out + 4 * 32);
    naive_flip_y(f, a); naive_to_hash(a, out + 5 * 32);
    naive_transpose(f, a); naive_to_hash(a, out + 6 * 32);
    naive_transpose(f, a); naive_flip_x(a, b); naive_flip_y(b, a); naive_to_hash(a, out + 7 * 32);
}

/* ---- to_luma601: pdqhash.rs:268-284 (RGB8 / RGBA8: alpha ignored) ---- */
void rph_ref_luma601(const uint8_t *px, int w, int h, int stride_bytes, int channels, uint8_t *luma)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *row = px + (size_t)y * (size_t)stride_bytes;
        for (int x = 0; x < w; x++) {
            const uint8_t *p = row + (size_t)x * (size_t)channels;
            uint32_t v = (299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u;
            luma[(size_t)y * w + x] = (uint8_t)v;
        }
    }
}

/* ---- box_one_d_float: pdqhash.rs:341-396 ---- */
static void box_one_d_float(const float *invec, size_t in_start, float *outvec, size_t out_start,
                            size_t vec_len, size_t stride, size_t win_size)
{
    size_t maxlen = vec_len > 1 ? vec_len : 1;
    if (win_size < 1) win_size = 1;
    if (win_size > maxlen) win_size = maxlen;
    size_t half_win = (win_size + 2) / 2;

    size_t phase_1 = half_win - 1;
    size_t phase_2 = win_size - half_win + 1;
    size_t phase_3 = vec_len > win_size ? vec_len - win_size : 0;
    size_t phase_4 = half_win - 1;

    size_t li = in_start, ri = in_start, oi = out_start;
    float sum = 0.0f;
    float curr_win = 0.0f;

    for (size_t t = 0; t < phase_1; t++) {
        sum = sum + invec[ri];
        curr_win = curr_win + 1.0f;
        ri += stride;
    }
    for (size_t t = 0; t < phase_2; t++) {
        sum = sum + invec[ri];
        curr_win = curr_win + 1.0f;
        outvec[oi] = sum / curr_win;
        ri += stride;
        oi += stride;
    }
    for (size_t t = 0; t < phase_3; t++) {
        sum = sum + invec[ri];
        sum = sum - invec[li];
        outvec[oi] = sum / curr_win;
        li += stride;
        ri += stride;
        oi += stride;
    }
    for (size_t t = 0; t < phase_4; t++) {
        sum = sum - invec[li];
        curr_win = curr_win - 1.0f;
        outvec[oi] = sum / curr_win;
        li += stride;
        oi += stride;
    }
}

/* ---- box_along_rows_float / box_along_cols_float: pdqhash.rs:398-408 ---- */
static void box_along_rows(const float *in, float *out, size_t rows, size_t cols, size_t win)
{
    for (size_t i = 0; i < rows; i++)
        box_one_d_float(in, i * cols, out, i * cols, cols, 1, win);
}
static void box_along_cols(const float *in, float *out, size_t rows, size_t cols, size_t win)
{
    for (size_t j = 0; j < cols; j++)
        box_one_d_float(in, j, out, j, rows, cols, win);
}

/* ---- jarosz_filter_float: pdqhash.rs:410-426 ---- */
void rph_ref_jarosz(float *buf, float *tmp, int rows, int cols, int w_rows, int w_cols, int nreps)
{
    for (int r = 0; r < nreps; r++) {
        box_along_rows(buf, tmp, (size_t)rows, (size_t)cols, (size_t)w_rows);
        box_along_cols(tmp, buf, (size_t)rows, (size_t)cols, (size_t)w_cols);
    }
}

/* ---- decimate_float::<64,64>: pdqhash.rs:428-443 ---- */
void rph_ref_decimate(const float *input, int in_r, int in_c, float *out /* [64][64] */)
{
    for (int i = 0; i < BUFFER_W_H; i++) {
        size_t ini = ((size_t)(i * 2 + 1) * (size_t)in_r) / (BUFFER_W_H * 2);
        const float *in_row = input + ini * (size_t)in_c;
        for (int j = 0; j < BUFFER_W_H; j++)
            out[i * BUFFER_W_H + j] = in_row[((size_t)(j * 2 + 1) * (size_t)in_c) / (BUFFER_W_H * 2)];
    }
}

/* ---- pdq_image_domain_quality_metric: pdqhash.rs:445-460 (generic R x C) ---- */
float rph_ref_quality(const float *buf, int R, int C)
{
    float sum = 0.0f;
    /* vertical pairs first (row i against row i+1), then horizontal */
    for (int i = 0; i + 1 < R; i++)
        for (int j = 0; j < C; j++) {
            float a = buf[i * C + j], b = buf[(i + 1) * C + j];
            sum = sum + truncf(fabsf(((a - b) * 100.0f) / 255.0f));
        }
    for (int i = 0; i < R; i++)
        for (int j = 0; j + 1 < C; j++) {
            float a = buf[i * C + j], b = buf[i * C + j + 1];
            sum = sum + truncf(fabsf(((a - b) * 100.0f) / 255.0f));
        }
    float q = sum / 90.0f;
    return q > 1.0f ? 1.0f : q;
}

/* ---- calculate_target_dimensions: pdqhash.rs:224-235 ---- */
void rph_ref_target_dimensions(uint32_t w, uint32_t h, uint32_t max_dim, uint32_t *ow, uint32_t *oh)
{
    if (w == 0 || h == 0) {
        *ow = w > 1 ? w : 1;
        *oh = h > 1 ? h : 1;
        return;
    }
    if (w > h) {
        uint64_t nh = (uint64_t)h * (uint64_t)max_dim / (uint64_t)w;
        if (nh < 1) nh = 1;
        *ow = max_dim;
        *oh = (uint32_t)nh;
    } else {
        uint64_t nw = (uint64_t)w * (uint64_t)max_dim / (uint64_t)h;
        if (nw < 1) nw = 1;
        *ow = (uint32_t)nw;
        *oh = max_dim;
    }
}

/* ---- PdqFeatures::new on a raw 64x64 buffer: pdqhash.rs:54-57 ---- */
void rph_ref_features_from_buffer64(const float *buf64, float *coeffs)
{
    rph_ref_dct64_to_16(buf64, coeffs);
}

/* ---- generate_pdq_from_luma: pdqhash.rs:238-262 ----
 * buf64_out (nullable) receives the decimated 64x64 buffer for debugging. */
int rph_ref_pdq_from_luma(const uint8_t *luma, int w, int h, float *coeffs, float *quality,
                          float *buf64_out)
{
    /* no size check here: the reference checks MIN_HASHABLE_DIM once, before the resize (pdqhash.rs:167), so a
     * 4000x5 input is hashed from its 512x1 thumbnail */
    size_t n = (size_t)w * (size_t)h;
    float *buf = (float *)malloc(n * sizeof(float));
    float *tmp = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++) buf[i] = (float)luma[i];

    int win_rows = (w + JAROSZ_WINDOW_DIVISOR - 1) / JAROSZ_WINDOW_DIVISOR; /* cols.div_ceil(64) */
    int win_cols = (h + JAROSZ_WINDOW_DIVISOR - 1) / JAROSZ_WINDOW_DIVISOR; /* rows.div_ceil(64) */
    rph_ref_jarosz(buf, tmp, h, w, win_rows, win_cols, PDQ_NUM_JAROSZ_XY_PASSES);

    float b64[BUFFER_W_H * BUFFER_W_H];
    rph_ref_decimate(buf, h, w, b64);
    *quality = rph_ref_quality(b64, BUFFER_W_H, BUFFER_W_H);
    rph_ref_dct64_to_16(b64, coeffs);
    if (buf64_out) memcpy(buf64_out, b64, sizeof(b64));
    free(buf);
    free(tmp);
    return RPH_REF_OK;
}

/* ---- generate_pdq_features for packed RGB8/RGBA8/Luma8: pdqhash.rs:166-196 ---- */
int rph_ref_pdq_features(const uint8_t *px, int w, int h, int stride_bytes, int channels,
                         float *coeffs, float *quality)
{
    if (w < MIN_HASHABLE_DIM || h < MIN_HASHABLE_DIM) return RPH_REF_TOO_SMALL;
    uint8_t *luma = (uint8_t *)malloc((size_t)w * (size_t)h);
    if (channels == 1) {
        for (int y = 0; y < h; y++)
            memcpy(luma + (size_t)y * w, px + (size_t)y * stride_bytes, (size_t)w);
    } else {
        rph_ref_luma601(px, w, h, stride_bytes, channels, luma);
    }
    /* Resize if larger than 512x512 (pdqhash.rs:181-191): the 1-channel luma image is resized, aspect preserved */
    if (w > DOWNSAMPLE_DIMS || h > DOWNSAMPLE_DIMS) {
        uint32_t nw, nh;
        rph_ref_target_dimensions((uint32_t)w, (uint32_t)h, DOWNSAMPLE_DIMS, &nw, &nh);
        uint8_t *small = (uint8_t *)malloc((size_t)nw * nh);
        rph_ref_resize_box_u8(luma, (uint32_t)w, (uint32_t)h, small, nw, nh);
        free(luma);
        luma = small;
        w = (int)nw;
        h = (int)nh;
    }
    int rc = rph_ref_pdq_from_luma(luma, w, h, coeffs, quality, NULL);
    free(luma);
    return rc;
}

/* ---- generate_pdq: pdqhash.rs:199-201 ---- */
int rph_ref_pdq(const uint8_t *px, int w, int h, int stride_bytes, int channels, uint8_t *hash,
                float *quality)
{
    float coeffs[256];
    int rc = rph_ref_pdq_features(px, w, h, stride_bytes, channels, coeffs, quality);
    if (rc == RPH_REF_OK) rph_ref_to_hash(coeffs, hash);
    return rc;
}

/* Batch helper for the cpu_baseline leg and tests: n contiguous w x h RGB8
 * images, one image per task like the reference's rayon par_iter
 * (scanner.rs:1202).  Threading lives in oracle/bench_ref.c. */
int rph_ref_pdq_batch_rgb(const uint8_t *rgb, int n, int w, int h, uint8_t *hashes, float *quality,
                          float *coeffs /* nullable */)
{
    for (int k = 0; k < n; k++) {
        float c[256], q;
        const uint8_t *img = rgb + (size_t)k * (size_t)w * (size_t)h * 3;
        int rc = rph_ref_pdq_features(img, w, h, w * 3, 3, c, &q);
        if (rc != RPH_REF_OK) return rc;
        rph_ref_to_hash(c, hashes + (size_t)k * 32);
        if (quality) quality[k] = q;
        if (coeffs) memcpy(coeffs + (size_t)k * 256, c, sizeof(c));
    }
    return RPH_REF_OK;
}
