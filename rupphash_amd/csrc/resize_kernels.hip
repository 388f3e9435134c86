// resize_kernels.hip -- the reference's pre-downsample for inputs with a side > 512 px
// (/root/reference/src/pdqhash.rs:181-220: to_luma601 at full resolution, then resize_luma_fast to the
// aspect-preserving thumbnail of calculate_target_dimensions, then generate_pdq_from_luma).
//
// resize_luma_fast is the third-party crate fast_image_resize 6.1.0 (Cargo.lock:1853), Convolution(Box) on U8; its
// source is not part of the reference tree, so this is a statement of its published algorithm (Pillow-style
// two-pass fixed-point convolution: f64 window weights -> i16 coefficients at an adaptive precision ->
// clip8((round + sum) >> precision), horizontal pass into a u8 intermediate, then vertical).  PARITY UNPINNED.
#include <cmath>
#include <vector>

#include "rph_internal.h"

namespace {

struct Axis {
    std::vector<uint32_t> start, size;
    std::vector<int16_t> coef;  // [out][window]
    int window = 0, precision = 0;
};

double box_filter(double x) { return (x > -0.5 && x <= 0.5) ? 1.0 : 0.0; }

// precompute_coefficients + Normalizer16 of the crate (one axis)
Axis build_axis(uint32_t in_size, uint32_t out_size)
{
    Axis a;
    const double scale = (double)in_size / (double)out_size;
    const double filter_scale = scale > 1.0 ? scale : 1.0;
    const double radius = 0.5 * filter_scale;  // Box support 0.5
    a.window = (int)std::ceil(radius) * 2 + 1;
    const double recip = 1.0 / filter_scale;
    std::vector<double> w((size_t)out_size * a.window, 0.0);
    a.start.assign(out_size, 0);
    a.size.assign(out_size, 0);
    double max_w = 0.0;
    for (uint32_t o = 0; o < out_size; o++) {
        const double in_center = ((double)o + 0.5) * scale;
        const uint32_t x_min = (uint32_t)std::max(0.0, std::floor(in_center - radius));
        const uint32_t x_max = (uint32_t)std::min((double)in_size, std::ceil(in_center + radius));
        const double center = in_center - 0.5;
        uint32_t bound_start = x_min, bound_end = x_max;
        double *cw = &w[(size_t)o * a.window];
        int cnt = 0;
        double ww = 0.0;
        for (uint32_t x = x_min; x < x_max; x++) {
            const double v = box_filter(((double)x - center) * recip);
            if (x == bound_start && v == 0.0 && cnt == 0) {
                bound_start++;  // zero leading coefficients are not used
            } else {
                cw[cnt++] = v;
                ww += v;
            }
        }
        for (int i = cnt - 1; i >= 0; i--) {  // nor zero trailing ones
            if (bound_end <= bound_start || cw[i] != 0.0) break;
            bound_end--;
        }
        if (ww != 0.0)
            for (int i = 0; i < cnt; i++) cw[i] /= ww;
        for (int i = 0; i < cnt; i++) max_w = std::max(max_w, cw[i]);
        a.start[o] = bound_start;
        a.size[o] = bound_end - bound_start;
    }
    for (int cur = 0; cur < 16; cur++) {
        a.precision = cur;
        if ((int32_t)std::llround(max_w * (double)(1 << (cur + 1))) >= (1 << 15)) break;
    }
    a.coef.resize(w.size());
    for (size_t i = 0; i < w.size(); i++) a.coef[i] = (int16_t)std::llround(w[i] * (double)(1 << a.precision));
    return a;
}

struct DevAxis {
    const uint32_t *start, *size;
    const int16_t *coef;
    int window, precision;
};

// to_luma601 (pdqhash.rs:268-284) at full resolution, u8 out
__global__ void __launch_bounds__(256) luma_u8_kernel(const uint8_t *__restrict__ px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                                                      size_t row_stride, size_t image_stride, uint8_t *__restrict__ out)
{
    const uint64_t total = (uint64_t)n * w * h;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t img = (uint32_t)(t / ((uint64_t)w * h));
        const uint32_t rem = (uint32_t)(t - (uint64_t)img * w * h);
        const uint32_t y = rem / w, x = rem - y * w;
        const uint8_t *p = px + (size_t)img * image_stride + (size_t)y * row_stride + (size_t)x * channels;
        out[t] = channels == 1 ? p[0] : (uint8_t)((299u * p[0] + 587u * p[1] + 114u * p[2] + 500u) / 1000u);
    }
}

__device__ __forceinline__ uint8_t clip8(int32_t v, int precision)
{
    v >>= precision;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src [n][h][w] -> dst [n][h][nw]
__global__ void __launch_bounds__(256) resize_h_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t n, uint32_t w,
                                                       uint32_t h, uint32_t nw, DevAxis ax)
{
    const uint64_t total = (uint64_t)n * h * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = (uint32_t)(t % nw);
        const uint64_t line = t / nw;  // img * h + y
        const uint8_t *row = src + line * w + ax.start[o];
        const int16_t *k = ax.coef + (size_t)o * ax.window;
        int32_t ss = 1 << (ax.precision - 1);
        for (uint32_t i = 0; i < ax.size[o]; i++) ss += (int32_t)row[i] * (int32_t)k[i];
        dst[t] = clip8(ss, ax.precision);
    }
}

// vertical pass: src [n][h][nw] -> dst [n][nh][nw]
__global__ void __launch_bounds__(256) resize_v_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t n, uint32_t h,
                                                       uint32_t nw, uint32_t nh, DevAxis ay)
{
    const uint64_t total = (uint64_t)n * nh * nw;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)(t % nw);
        const uint32_t o = (uint32_t)((t / nw) % nh);
        const uint32_t img = (uint32_t)(t / ((uint64_t)nw * nh));
        const uint8_t *col = src + ((size_t)img * h + ay.start[o]) * nw + x;
        const int16_t *k = ay.coef + (size_t)o * ay.window;
        int32_t ss = 1 << (ay.precision - 1);
        for (uint32_t i = 0; i < ay.size[o]; i++) ss += (int32_t)col[(size_t)i * nw] * (int32_t)k[i];
        dst[t] = clip8(ss, ay.precision);
    }
}

unsigned grid_for(uint64_t total) { return (unsigned)std::min<uint64_t>((total + 255) / 256, 65536); }

}  // namespace

int rph_launch_pdq_resized(rph_ctx *ctx, const uint8_t *d_px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                           size_t row_stride, size_t image_stride, uint8_t *d_hash, float *d_quality, float *d_coeffs,
                           uint8_t *d_dihedral, uint8_t *d_valid, hipStream_t stream)
{
    if (n == 0) return RPH_OK;
    uint32_t nw, nh;
    rph_pdq_target_dimensions(w, h, RPH_PDQ_MAX_DIM, &nw, &nh);  // pdqhash.rs:183
    const Axis ax = build_axis(w, nw), ay = build_axis(h, nh);

    // device copies of the two coefficient tables + per-chunk planes; this path is not the hot one: allocate per call
    struct Buf {
        void *p = nullptr;
        ~Buf()
        {
            if (p) (void)hipFree(p);
        }
    } b_xs, b_xz, b_xc, b_ys, b_yz, b_yc, b_luma, b_tmp, b_small;
    auto up = [&](Buf &b, const void *src, size_t bytes) -> int {
        RPH_HIP_CHECK(hipMalloc(&b.p, bytes ? bytes : 1));
        RPH_HIP_CHECK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, stream));
        return RPH_OK;
    };
    int rc;
    if ((rc = up(b_xs, ax.start.data(), ax.start.size() * 4)) || (rc = up(b_xz, ax.size.data(), ax.size.size() * 4)) ||
        (rc = up(b_xc, ax.coef.data(), ax.coef.size() * 2)) || (rc = up(b_ys, ay.start.data(), ay.start.size() * 4)) ||
        (rc = up(b_yz, ay.size.data(), ay.size.size() * 4)) || (rc = up(b_yc, ay.coef.data(), ay.coef.size() * 2)))
        return rc;
    const DevAxis dx{(const uint32_t *)b_xs.p, (const uint32_t *)b_xz.p, (const int16_t *)b_xc.p, ax.window, ax.precision};
    const DevAxis dy{(const uint32_t *)b_ys.p, (const uint32_t *)b_yz.p, (const int16_t *)b_yc.p, ay.window, ay.precision};

    const size_t full = (size_t)w * h;
    uint32_t chunk = (uint32_t)std::max<size_t>(1, ((size_t)256 << 20) / full);
    chunk = std::min(chunk, n);
    RPH_HIP_CHECK(hipMalloc(&b_luma.p, full * chunk));
    RPH_HIP_CHECK(hipMalloc(&b_tmp.p, (size_t)nw * h * chunk));
    RPH_HIP_CHECK(hipMalloc(&b_small.p, (size_t)nw * nh * chunk));
    for (uint32_t first = 0; first < n; first += chunk) {
        const uint32_t m = std::min(chunk, n - first);
        hipLaunchKernelGGL(luma_u8_kernel, dim3(grid_for((uint64_t)m * full)), dim3(256), 0, stream, d_px + (size_t)first * image_stride, m, w,
                           h, channels, row_stride, image_stride, (uint8_t *)b_luma.p);
        hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((uint64_t)m * h * nw)), dim3(256), 0, stream, (const uint8_t *)b_luma.p,
                           (uint8_t *)b_tmp.p, m, w, h, nw, dx);
        hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((uint64_t)m * nh * nw)), dim3(256), 0, stream, (const uint8_t *)b_tmp.p,
                           (uint8_t *)b_small.p, m, h, nw, nh, dy);
        RPH_HIP_CHECK(hipGetLastError());
        // generate_pdq_from_luma on the thumbnail (no second size check in the reference: a 4000x5 input is hashed from 512x1)
        rc = rph_launch_pdq_generic(ctx, (const uint8_t *)b_small.p, m, nw, nh, 1, nw, (size_t)nw * nh, d_hash + (size_t)first * 32,
                                    d_quality ? d_quality + first : nullptr, d_coeffs ? d_coeffs + (size_t)first * 256 : nullptr,
                                    d_dihedral ? d_dihedral + (size_t)first * 256 : nullptr, d_valid ? d_valid + first : nullptr, stream);
        if (rc != RPH_OK) return rc;
    }
    RPH_HIP_CHECK(hipStreamSynchronize(stream));  // the per-call buffers above are freed on return
    return RPH_OK;
}
