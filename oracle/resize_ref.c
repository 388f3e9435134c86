/*
 * oracle/resize_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference pre-downsamples inputs with a side > 512 px (src/pdqhash.rs:181-220,
 * resize_luma_fast) through the third-party crate fast_image_resize, pinned at 6.1.0
 * (Cargo.lock:1853-1856), ResizeAlg::Convolution(FilterType::Box) on PixelType::U8 (pdqhash.rs:36,210-216).
 * The crate's source is NOT under /root/reference, so this file restates its PUBLISHED algorithm
 * (crate v6, src/convolution: precompute_coefficients, Normalizer16, native u8 convolution; the
 * design is the Pillow ImagingResample family): PARITY UNPINNED -- no reference test or fixture pins a
 * single output of this step, and details recalled from the published source may deviate.
 *
 *   per axis:  scale = in / out;  filter_scale = max(scale, 1);  radius = 0.5 * filter_scale (Box support 0.5)
 *              window = ceil(radius) * 2 + 1
 *              for each output o: center_in = (o + 0.5) * scale
 *                 x_min = max(0, floor(center_in - radius)),  x_max = min(in, ceil(center_in + radius))
 *                 w(x) = box(((x - (center_in - 0.5)) / filter_scale)),  box(t) = 1 if -0.5 < t <= 0.5 else 0
 *                 zero leading / trailing weights are dropped, the rest normalised to sum 1 (f64)
 *   fixed point: precision = largest p in 0..15 with round(max_w * 2^(p+1)) < 2^15 ... (loop below), coef = round(w * 2^p) as i16
 *   pixel:      clip8((2^(p-1) + sum src * coef) >> p)
 *   passes:     horizontal first into a u8 intermediate, then vertical
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t *start, *size;
    int16_t *coef;   /* [out][window] */
    int window, precision;
} axis_t;

static double box_filter(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }

static void axis_build(axis_t *a, uint32_t in_size, uint32_t out_size)
{
    const double scale = (double)in_size / (double)out_size;
    const double filter_scale = scale > 1.0 ? scale : 1.0;
    const double radius = 0.5 * filter_scale;
    const int window = (int)ceil(radius) * 2 + 1;
    const double recip = 1.0 / filter_scale;
    double *w = (double *)calloc((size_t)out_size * window, sizeof(double));
    a->start = (uint32_t *)calloc(out_size, sizeof(uint32_t));
    a->size = (uint32_t *)calloc(out_size, sizeof(uint32_t));
    a->window = window;
    double max_w = 0.0;
    for (uint32_t o = 0; o < out_size; o++) {
        const double in_center = ((double)o + 0.5) * scale;
        double lo = floor(in_center - radius);
        if (lo < 0.0) lo = 0.0;
        double hi = ceil(in_center + radius);
        if (hi > (double)in_size) hi = (double)in_size;
        const uint32_t x_min = (uint32_t)lo, x_max = (uint32_t)hi;
        const double center = in_center - 0.5;
        uint32_t bound_start = x_min, bound_end = x_max;
        double *cw = w + (size_t)o * window;
        int cnt = 0;
        double ww = 0.0;
        for (uint32_t x = x_min; x < x_max; x++) {
            const double v = box_filter(((double)x - center) * recip);
            if (x == bound_start && v == 0.0 && cnt == 0) {
                bound_start++;           /* don't use zero leading coefficients */
            } else {
                cw[cnt++] = v;
                ww += v;
            }
        }
        for (int i = cnt - 1; i >= 0; i--) { /* don't use zero trailing coefficients */
            if (bound_end <= bound_start || cw[i] != 0.0) break;
            bound_end--;
        }
        if (ww != 0.0)
            for (int i = 0; i < cnt; i++) cw[i] /= ww;
        for (int i = 0; i < cnt; i++)
            if (cw[i] > max_w) max_w = cw[i];
        a->start[o] = bound_start;
        a->size[o] = bound_end - bound_start;
    }
    int precision = 0;
    for (int cur = 0; cur < 16; cur++) {
        precision = cur;
        const int32_t next_value = (int32_t)llround(max_w * (double)(1 << (precision + 1)));
        if (next_value >= (1 << 15)) break;
    }
    a->precision = precision;
    a->coef = (int16_t *)calloc((size_t)out_size * window, sizeof(int16_t));
    for (size_t i = 0; i < (size_t)out_size * window; i++) a->coef[i] = (int16_t)llround(w[i] * (double)(1 << precision));
    free(w);
}
static void axis_free(axis_t *a) { free(a->start); free(a->size); free(a->coef); }

static inline uint8_t clip8(int32_t v, int precision)
{
    v >>= precision;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* src w x h u8 -> dst nw x nh u8 (horizontal pass, then vertical) */
void rph_ref_resize_box_u8(const uint8_t *src, uint32_t w, uint32_t h, uint8_t *dst, uint32_t nw, uint32_t nh)
{
    axis_t ax, ay;
    axis_build(&ax, w, nw);
    axis_build(&ay, h, nh);
    uint8_t *tmp = (uint8_t *)malloc((size_t)nw * h);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t o = 0; o < nw; o++) {
            int32_t ss = 1 << (ax.precision - 1);
            const int16_t *k = ax.coef + (size_t)o * ax.window;
            for (uint32_t i = 0; i < ax.size[o]; i++) ss += (int32_t)src[(size_t)y * w + ax.start[o] + i] * (int32_t)k[i];
            tmp[(size_t)y * nw + o] = clip8(ss, ax.precision);
        }
    for (uint32_t o = 0; o < nh; o++)
        for (uint32_t x = 0; x < nw; x++) {
            int32_t ss = 1 << (ay.precision - 1);
            const int16_t *k = ay.coef + (size_t)o * ay.window;
            for (uint32_t i = 0; i < ay.size[o]; i++) ss += (int32_t)tmp[(size_t)(ay.start[o] + i) * nw + x] * (int32_t)k[i];
            dst[(size_t)o * nw + x] = clip8(ss, ay.precision);
        }
    free(tmp);
    axis_free(&ax);
    axis_free(&ay);
}

/* axis facts for the fixture tests: precision, window, and the smallest / largest sum of an output's coefficients
 * (a normalised kernel sums to 1 << precision up to the rounding of its taps) */
void rph_ref_resize_axis_info(uint32_t in_size, uint32_t out_size, int *precision, int *window, int32_t *sum_min, int32_t *sum_max)
{
    axis_t a;
    axis_build(&a, in_size, out_size);
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    for (uint32_t o = 0; o < out_size; o++) {
        int32_t s = 0;
        for (uint32_t i = 0; i < a.size[o]; i++) s += a.coef[(size_t)o * a.window + i];
        if (s < lo) lo = s;
        if (s > hi) hi = s;
    }
    *precision = a.precision;
    *window = a.window;
    *sum_min = lo;
    *sum_max = hi;
    axis_free(&a);
}
