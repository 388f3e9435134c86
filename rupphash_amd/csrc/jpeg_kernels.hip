// jpeg_kernels.hip -- device half of the JPEG path (row N3 of SURVEY 8f) and its C ABI.
//
// The reference decodes "jpg" | "jpeg" files with zune-jpeg 0.5.15 into Luma8 / Rgb8 and hands the pixels to generate_pdq_features
// (/root/reference/src/scanner.rs:473-508, :1410).  Here the host only undoes the entropy coding (jpeg_host.cpp: a serial bit
// stream per image, one image per host thread); the quantised coefficients cross PCIe once and everything with arithmetic in it
// runs on the device, a whole batch of images per launch:
//   jpeg_idct_kernel   one lane per 8x8 block: dequantise, integer IDCT (columns, rows) entirely in registers, level shift, clamp;
//                      128 B read and 64 B written per block -- HBM-bound, no LDS
//   jpeg_color_kernel  one lane per 4 output pixels: chroma upsampling as a pure function of the position (no intermediate
//                      full-resolution chroma planes), YCbCr -> RGB, packed Rgb8 (or Luma8 for one component) that the PDQ kernels read
// and then the existing PDQ launchers hash the pixels where they lie.  Nothing returns to the host but the hashes.
//
// Two arithmetic flavours (the tests' CPU checker restates the same pair):
//   RPH_JPEG_ZUNE (default)  zune-jpeg as recalled: stb_image's integer IDCT, (3 a + b + 2) >> 2 upsampling per direction, 45/32-style
//                            colour constants.  PARITY UNPINNED: the crate's source is not in the reference tree.
//   RPH_JPEG_LIBJPEG         libjpeg-turbo's defaults (jidctint.c islow, jdsample.c fancy upsampling, jdcolor.c): pinned by the tests
//                            against Pillow's decode of the reference's own JPEG files and of generated ones.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "jpeg_host.h"
#include "rph_internal.h"

namespace {

struct JPlane {            // one per component of each decoded image
    uint64_t first_block;  // in the chunk's coefficient buffer
    uint64_t out_off;      // byte offset of the sample plane in the chunk's plane buffer
    uint32_t blocks_w, blocks_h;
    uint32_t qt;           // index of the plane's 64-entry table in the chunk's table buffer
    uint32_t pitch;        // blocks_w * 8
};
struct JImage {
    uint64_t plane_off[3];  // sample planes (Y, Cb, Cr)
    uint64_t out_off;       // packed pixels
    uint32_t w, h, ncomp;
    uint32_t hs, vs;        // chroma upsampling factors (1 or 2)
    uint32_t pitch[3];
    uint32_t cw, ch;        // chroma samples the upsampler may use: real component samples (libjpeg) or the padded plane (zune)
    uint32_t out_stride;    // bytes per output row: ncomp * align4(w)
    uint32_t pad;
};

#define MUL(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
#define ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define SUB(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))
#define SHL(a, n) ((int32_t)((uint32_t)(a) << (n)))

__device__ __forceinline__ int clamp8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// libjpeg-turbo jidctint.c (jpeg_idct_islow): CONST_BITS 13; `shift` = 11 after the column pass, 18 after the row pass
template <int SHIFT>
__device__ __forceinline__ void idct_islow(int32_t &s0, int32_t &s1, int32_t &s2, int32_t &s3, int32_t &s4, int32_t &s5, int32_t &s6, int32_t &s7)
{
    int32_t z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    z1 = MUL(ADD(s2, s6), 4433);
    tmp2 = ADD(z1, MUL(s6, -15137));
    tmp3 = ADD(z1, MUL(s2, 6270));
    tmp0 = SHL(ADD(s0, s4), 13);
    tmp1 = SHL(SUB(s0, s4), 13);
    tmp10 = ADD(tmp0, tmp3);
    tmp13 = SUB(tmp0, tmp3);
    tmp11 = ADD(tmp1, tmp2);
    tmp12 = SUB(tmp1, tmp2);
    tmp0 = s7;
    tmp1 = s5;
    tmp2 = s3;
    tmp3 = s1;
    z1 = ADD(tmp0, tmp3);
    z2 = ADD(tmp1, tmp2);
    z3 = ADD(tmp0, tmp2);
    z4 = ADD(tmp1, tmp3);
    z5 = MUL(ADD(z3, z4), 9633);
    tmp0 = MUL(tmp0, 2446);
    tmp1 = MUL(tmp1, 16819);
    tmp2 = MUL(tmp2, 25172);
    tmp3 = MUL(tmp3, 12299);
    z1 = MUL(z1, -7373);
    z2 = MUL(z2, -20995);
    z3 = ADD(MUL(z3, -16069), z5);
    z4 = ADD(MUL(z4, -3196), z5);
    tmp0 = ADD(tmp0, ADD(z1, z3));
    tmp1 = ADD(tmp1, ADD(z2, z4));
    tmp2 = ADD(tmp2, ADD(z2, z3));
    tmp3 = ADD(tmp3, ADD(z1, z4));
    constexpr int32_t rnd = 1 << (SHIFT - 1);
    s0 = ADD(ADD(tmp10, tmp3), rnd) >> SHIFT;
    s7 = ADD(SUB(tmp10, tmp3), rnd) >> SHIFT;
    s1 = ADD(ADD(tmp11, tmp2), rnd) >> SHIFT;
    s6 = ADD(SUB(tmp11, tmp2), rnd) >> SHIFT;
    s2 = ADD(ADD(tmp12, tmp1), rnd) >> SHIFT;
    s5 = ADD(SUB(tmp12, tmp1), rnd) >> SHIFT;
    s3 = ADD(ADD(tmp13, tmp0), rnd) >> SHIFT;
    s4 = ADD(SUB(tmp13, tmp0), rnd) >> SHIFT;
}
// stb_image / zune-jpeg integer IDCT: 12-bit constants, BIAS added to the even part, arithmetic shift
template <int32_t BIAS, int SHIFT>
__device__ __forceinline__ void idct_stb(int32_t &s0, int32_t &s1, int32_t &s2, int32_t &s3, int32_t &s4, int32_t &s5, int32_t &s6, int32_t &s7)
{
    int32_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;
    p1 = MUL(ADD(s2, s6), 2217);
    t2 = ADD(p1, MUL(s6, -7567));
    t3 = ADD(p1, MUL(s2, 3135));
    t0 = SHL(ADD(s0, s4), 12);
    t1 = SHL(SUB(s0, s4), 12);
    x0 = ADD(ADD(t0, t3), BIAS);
    x3 = ADD(SUB(t0, t3), BIAS);
    x1 = ADD(ADD(t1, t2), BIAS);
    x2 = ADD(SUB(t1, t2), BIAS);
    t0 = s7;
    t1 = s5;
    t2 = s3;
    t3 = s1;
    p3 = ADD(t0, t2);
    p4 = ADD(t1, t3);
    p1 = ADD(t0, t3);
    p2 = ADD(t1, t2);
    p5 = MUL(ADD(p3, p4), 4816);
    t0 = MUL(t0, 1223);
    t1 = MUL(t1, 8410);
    t2 = MUL(t2, 12586);
    t3 = MUL(t3, 6149);
    p1 = ADD(p5, MUL(p1, -3685));
    p2 = ADD(p5, MUL(p2, -10497));
    p3 = MUL(p3, -8034);
    p4 = MUL(p4, -1597);
    t3 = ADD(t3, ADD(p1, p4));
    t2 = ADD(t2, ADD(p2, p3));
    t1 = ADD(t1, ADD(p2, p4));
    t0 = ADD(t0, ADD(p1, p3));
    s0 = ADD(x0, t3) >> SHIFT;
    s7 = SUB(x0, t3) >> SHIFT;
    s1 = ADD(x1, t2) >> SHIFT;
    s6 = SUB(x1, t2) >> SHIFT;
    s2 = ADD(x2, t1) >> SHIFT;
    s5 = SUB(x2, t1) >> SHIFT;
    s3 = ADD(x3, t0) >> SHIFT;
    s4 = SUB(x3, t0) >> SHIFT;
}

// grid: x = groups of 256 blocks of a plane, y = plane.  Lane = block: neighbouring lanes write neighbouring 8-byte row segments.
template <int FL>
__global__ void __launch_bounds__(256) jpeg_idct_kernel(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qts, const JPlane *__restrict__ planes,
                                                        uint8_t *__restrict__ out)
{
    const JPlane pl = planes[blockIdx.y];
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= pl.blocks_w * pl.blocks_h) return;
    const uint32_t bx = b % pl.blocks_w, by = b / pl.blocks_w;
    const uint4 *src = reinterpret_cast<const uint4 *>(coef + (pl.first_block + b) * 64);
    const uint16_t *qt = qts + (size_t)pl.qt * 64;  // plane-uniform: scalar loads
    int32_t c[64];
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const uint4 r = src[y];
        const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[8 * y + 2 * i] = MUL((int32_t)(int16_t)(u[i] & 0xFFFFu), (int32_t)qt[8 * y + 2 * i]);          // T.81 A.3.4
            c[8 * y + 2 * i + 1] = MUL((int32_t)(int16_t)(u[i] >> 16), (int32_t)qt[8 * y + 2 * i + 1]);
        }
    }
#pragma unroll
    for (int x = 0; x < 8; x++) {
        if (FL == RPH_JPEG_LIBJPEG)
            idct_islow<11>(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
        else
            idct_stb<512, 10>(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
    }
    uint8_t *dst = out + pl.out_off + (size_t)(by * 8) * pl.pitch + bx * 8;
#pragma unroll
    for (int y = 0; y < 8; y++) {
        int32_t *r = c + 8 * y;
        int v[8];
        if (FL == RPH_JPEG_LIBJPEG) {
            idct_islow<18>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
            // range_limit[v & RANGE_MASK], table centred on 128: v modulo 1024 into [-512, 511], then clamp(v + 128)
#pragma unroll
            for (int x = 0; x < 8; x++) v[x] = clamp8((((r[x] + 512) & 1023) - 512) + 128);
        } else {
            idct_stb<65536 + (128 << 17), 17>(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
#pragma unroll
            for (int x = 0; x < 8; x++) {
                // The value is made opaque between the shift and the clamp: ROCm 7.2's compiler otherwise fuses them into gfx950's
                // v_ashr_pk_u8_i32 and ORs the other two bytes into its result as if the instruction cleared bits 16..31 -- it leaves
                // them as they were (measured: bytes 2 and 3 of each dword came out wrong whenever the register's old value was not a byte)
                int t = r[x];
                asm("" : "+v"(t));
                v[x] = clamp8(t);
            }
        }
        uint2 w;
        w.x = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        w.y = (uint32_t)v[4] | ((uint32_t)v[5] << 8) | ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
        *reinterpret_cast<uint2 *>(dst + (size_t)y * pl.pitch) = w;
    }
}

struct PlaneView {
    const uint8_t *p;
    int pitch, w, h;
};
__device__ __forceinline__ int at(const PlaneView &pl, int x, int y)
{
    y = y < 0 ? 0 : (y >= pl.h ? pl.h - 1 : y);  // the edge row repeats above and below (libjpeg: jdmainct.c context rows)
    return pl.p[(size_t)y * pl.pitch + x];
}
// chroma sample of the full-resolution grid at (x, y)
template <int FL>
__device__ __forceinline__ int upsampled(const PlaneView &pl, int x, int y, int hs, int vs)
{
    if (hs == 1 && vs == 1) return at(pl, x, y);
    const int n = pl.w;
    if (FL == RPH_JPEG_LIBJPEG) {
        if (hs == 2 && n <= 2) return at(pl, x >> 1, vs == 2 ? (y >> 1) : y);  // jdsample.c: fancy upsampling needs > 2 columns
        if (hs == 2 && vs == 1) {                                                // h2v1_fancy_upsample
            const int c = x >> 1, v = at(pl, c, y);
            if ((x & 1) == 0) return c == 0 ? v : (3 * v + at(pl, c - 1, y) + 1) >> 2;
            return c == n - 1 ? v : (3 * v + at(pl, c + 1, y) + 2) >> 2;
        }
        if (hs == 1) {  // h1v2_fancy_upsample
            const int r = y >> 1, lower = y & 1;
            return (3 * at(pl, x, r) + at(pl, x, lower ? r + 1 : r - 1) + (lower ? 2 : 1)) >> 2;
        }
        // h2v2_fancy_upsample
        const int r = y >> 1, rr = (y & 1) ? r + 1 : r - 1, c = x >> 1;
        const int cs = 3 * at(pl, c, r) + at(pl, c, rr);
        if ((x & 1) == 0) {
            if (c == 0) return (cs * 4 + 8) >> 4;
            return (3 * cs + (3 * at(pl, c - 1, r) + at(pl, c - 1, rr)) + 8) >> 4;
        }
        if (c == n - 1) return (cs * 4 + 7) >> 4;
        return (3 * cs + (3 * at(pl, c + 1, r) + at(pl, c + 1, rr)) + 7) >> 4;
    }
    // zune-jpeg (recalled): vertical (3 a + b + 2) >> 2, then the same horizontally on the result; edge samples are copied
    const int r = vs == 2 ? (y >> 1) : y, c = hs == 2 ? (x >> 1) : x;
    const int rr = vs == 2 ? ((y & 1) ? r + 1 : r - 1) : r;
    auto zv = [&](int cc) { return vs == 2 ? ((3 * at(pl, cc, r) + at(pl, cc, rr) + 2) >> 2) : at(pl, cc, r); };
    const int v = zv(c);
    if (hs == 1) return v;
    if ((x & 1) == 0) return c == 0 ? v : (3 * v + zv(c - 1) + 2) >> 2;
    return c == n - 1 ? v : (3 * v + zv(c + 1) + 2) >> 2;
}
template <int FL>
__device__ __forceinline__ void ycc_to_rgb(int y, int cb, int cr, int &r, int &g, int &b)
{
    cb -= 128;
    cr -= 128;
    if (FL == RPH_JPEG_LIBJPEG) {  // jdcolor.c, SCALEBITS 16
        r = clamp8(y + ((91881 * cr + 32768) >> 16));
        g = clamp8(y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
        b = clamp8(y + ((116130 * cb + 32768) >> 16));
    } else {
        r = clamp8(y + ((45 * cr) >> 5));
        g = clamp8(y - ((11 * cb + 23 * cr) >> 5));
        b = clamp8(y + ((113 * cb) >> 6));
    }
}

// grid: x = groups of 256 lanes over ceil(w / 4) * h four-pixel groups, y = image
template <int FL>
__global__ void __launch_bounds__(256) jpeg_color_kernel(const uint8_t *__restrict__ planes, const JImage *__restrict__ imgs, uint8_t *__restrict__ out)
{
    const JImage im = imgs[blockIdx.y];
    const uint32_t w4 = (im.w + 3) / 4;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= w4 * im.h) return;
    const int y = (int)(t / w4), x0 = (int)(t % w4) * 4;
    const uint8_t *yrow = planes + im.plane_off[0] + (size_t)y * im.pitch[0];
    uint8_t *dst = out + im.out_off + (size_t)y * im.out_stride;
    if (im.ncomp == 1) {  // Luma8: the plane's own bytes (the plane is padded to whole blocks, so 4 bytes are always there)
        *reinterpret_cast<uint32_t *>(dst + x0) = (uint32_t)yrow[x0] | ((uint32_t)yrow[x0 + 1] << 8) | ((uint32_t)yrow[x0 + 2] << 16) | ((uint32_t)yrow[x0 + 3] << 24);
        return;
    }
    PlaneView cbp{planes + im.plane_off[1], (int)im.pitch[1], (int)im.cw, (int)im.ch};
    PlaneView crp{planes + im.plane_off[2], (int)im.pitch[2], (int)im.cw, (int)im.ch};
    uint32_t px[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int x = min(x0 + i, (int)im.w - 1);  // the tail of the last group repeats the last pixel into the row padding
        int r, g, b;
        ycc_to_rgb<FL>(yrow[x], upsampled<FL>(cbp, x, y, (int)im.hs, (int)im.vs), upsampled<FL>(crp, x, y, (int)im.hs, (int)im.vs), r, g, b);
        px[3 * i] = (uint32_t)r;
        px[3 * i + 1] = (uint32_t)g;
        px[3 * i + 2] = (uint32_t)b;
    }
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + 3 * x0);
    d32[0] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    d32[1] = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
    d32[2] = px[8] | (px[9] << 8) | (px[10] << 16) | (px[11] << 24);
}

#define RPH_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != RPH_OK) return rc_; \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------------------------------------
// The chunk pipeline: two slots (pinned staging + device buffers + stream), so the host threads decode chunk k + 1 while the device
// works on chunk k.  Kept in the context across calls; one JPEG batch call per context at a time (ctx->jpeg_mu).
// ---------------------------------------------------------------------------------------------------------------------------
struct Slot {
    hipStream_t stream = nullptr;
    // host (pinned)
    int16_t *h_coef = nullptr;
    uint8_t *h_desc = nullptr;  // planes | images | tables
    uint8_t *h_res = nullptr;   // hash | quality | coeffs | dihedral | valid
    // device
    int16_t *d_coef = nullptr;
    uint8_t *d_desc = nullptr, *d_planes = nullptr, *d_out = nullptr, *d_res = nullptr;
    size_t coef_bytes = 0, desc_bytes = 0, res_images = 0;
    void release()
    {
        if (stream) (void)hipStreamSynchronize(stream);
        for (void *p : {(void *)h_coef, (void *)h_desc, (void *)h_res})
            if (p) (void)hipHostFree(p);
        for (void *p : {(void *)d_coef, (void *)d_desc, (void *)d_planes, (void *)d_out, (void *)d_res})
            if (p) (void)hipFree(p);
        if (stream) (void)hipStreamDestroy(stream);
        *this = Slot();
    }
    // capacity for `coef_need` bytes of coefficients and `images` images
    int reserve(size_t coef_need, size_t images)
    {
        if (!stream) RPH_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        if (coef_bytes < coef_need) {
            RPH_HIP_CHECK(hipStreamSynchronize(stream));
            for (void *p : {(void *)h_coef})
                if (p) (void)hipHostFree(p);
            for (void *p : {(void *)d_coef, (void *)d_planes, (void *)d_out})
                if (p) (void)hipFree(p);
            h_coef = nullptr, d_coef = nullptr, d_planes = nullptr, d_out = nullptr, coef_bytes = 0;
            RPH_HIP_CHECK(hipHostMalloc((void **)&h_coef, coef_need));
            RPH_HIP_CHECK(hipMalloc((void **)&d_coef, coef_need));
            RPH_HIP_CHECK(hipMalloc((void **)&d_planes, coef_need / 2 + 256));  // 64 bytes of samples per 128 bytes of coefficients
            // packed pixels never exceed the coefficient bytes (4:2:0: both 3 w h; Luma8: w h against 2 w h), plus row / image padding
            RPH_HIP_CHECK(hipMalloc((void **)&d_out, coef_need + coef_need / 8 + 4096));
            coef_bytes = coef_need;
        }
        if (res_images < images) {
            RPH_HIP_CHECK(hipStreamSynchronize(stream));
            for (void *p : {(void *)h_desc, (void *)h_res})
                if (p) (void)hipHostFree(p);
            for (void *p : {(void *)d_desc, (void *)d_res})
                if (p) (void)hipFree(p);
            h_desc = nullptr, h_res = nullptr, d_desc = nullptr, d_res = nullptr, res_images = 0;
            desc_bytes = images * (3 * sizeof(JPlane) + sizeof(JImage) + 3 * 128);
            RPH_HIP_CHECK(hipHostMalloc((void **)&h_desc, desc_bytes));
            RPH_HIP_CHECK(hipMalloc((void **)&d_desc, desc_bytes));
            RPH_HIP_CHECK(hipHostMalloc((void **)&h_res, images * RES_BYTES));
            RPH_HIP_CHECK(hipMalloc((void **)&d_res, images * RES_BYTES));
            res_images = images;
        }
        return RPH_OK;
    }
    static constexpr size_t RES_BYTES = 32 + 4 + 1024 + 256 + 4;  // per image: hash, quality, coefficients, dihedral, valid (padded)
};
struct JpegPipe {
    Slot slot[2];
    void release()
    {
        slot[0].release();
        slot[1].release();
    }
};

constexpr size_t CHUNK_COEF_BYTES = (size_t)192 << 20;   // per slot: ~250 images of 512x512 4:2:0
constexpr size_t MAX_IMAGE_COEF_BYTES = (size_t)3 << 30;  // one image beyond this is refused (RPH_ERR_UNSUPPORTED)
constexpr uint32_t CHUNK_MAX_IMAGES = 4096;

struct Job {
    const uint8_t *data = nullptr;
    size_t len = 0;
    rphj::Frame frame;
    int status = RPH_OK;
    uint64_t first_block = 0;  // within the chunk
};

size_t out_bytes_of(const rphj::Frame &f)
{
    const size_t stride = (size_t)f.ncomp * align_up(f.w, 4);
    return align_up(stride * f.h, 64);
}

// decode jobs [first, last) on `threads` host threads into the slot's staging buffer
void decode_chunk(std::vector<Job> &jobs, size_t first, size_t last, int16_t *h_coef, unsigned threads)
{
    std::atomic<size_t> next{first};
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= last) return;
            Job &j = jobs[i];
            if (j.status != RPH_OK) continue;
            j.status = rphj::decode_coefficients(j.data, j.len, j.frame, h_coef + j.first_block * 64);
        }
    };
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, threads), last - first);
    if (nt <= 1) {
        work();
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t + 1 < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

struct Outputs {
    uint8_t *hash = nullptr;
    float *quality = nullptr;
    float *coeffs = nullptr;
    uint8_t *dihedral = nullptr;
    uint8_t *valid = nullptr;
    int32_t *status = nullptr;
    uint8_t *pixels = nullptr;  // single-image decode: packed w * h * channels
    bool want_hash = true;
};

int run_batch(rph_ctx *ctx, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads, Outputs out)
{
    if (flavour != RPH_JPEG_ZUNE && flavour != RPH_JPEG_LIBJPEG) {
        rph_set_error("rph_jpeg: unknown flavour %d", flavour);
        return RPH_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(ctx->jpeg_mu);
    RPH_HIP_CHECK(hipSetDevice(ctx->device));
    if (!ctx->jpeg) ctx->jpeg = new JpegPipe();
    JpegPipe &P = *static_cast<JpegPipe *>(ctx->jpeg);
    unsigned threads = n_threads ? n_threads : std::max(1u, std::thread::hardware_concurrency());
    threads = std::min(threads, 256u);

    std::vector<Job> jobs(n);
    for (uint32_t i = 0; i < n; i++) {
        Job &j = jobs[i];
        j.data = data[i];
        j.len = len[i];
        j.status = (j.data && j.len) ? rphj::parse_frame(j.data, j.len, j.frame) : RPH_ERR_INVALID_ARG;
        if (j.status == RPH_OK && j.frame.total_blocks * 128 > MAX_IMAGE_COEF_BYTES) j.status = RPH_ERR_UNSUPPORTED;
    }

    struct Pending {
        bool active = false;
        size_t first = 0, last = 0;
    } pend[2];
    auto finish = [&](int b) -> int {
        if (!pend[b].active) return RPH_OK;
        Slot &S = P.slot[b];
        RPH_HIP_CHECK(hipStreamSynchronize(S.stream));
        const size_t m = pend[b].last - pend[b].first, f0 = pend[b].first;
        const uint8_t *r = S.h_res;
        if (out.hash) memcpy(out.hash + f0 * 32, r, m * 32);
        r += S.res_images * 32;
        if (out.quality) memcpy(out.quality + f0, r, m * 4);
        r += S.res_images * 4;
        if (out.coeffs) memcpy(out.coeffs + f0 * 256, r, m * 1024);
        r += S.res_images * 1024;
        if (out.dihedral) memcpy(out.dihedral + f0 * 256, r, m * 256);
        r += S.res_images * 256;
        if (out.valid)
            for (size_t i = 0; i < m; i++) out.valid[f0 + i] = jobs[f0 + i].status == RPH_OK ? r[i] : 0;
        pend[b].active = false;
        return RPH_OK;
    };

    int k = 0;
    for (size_t first = 0; first < n; k++) {
        // ---- the chunk: as many images as fit the staging buffer
        size_t last = first, blocks = 0;
        while (last < n && last - first < CHUNK_MAX_IMAGES) {
            const Job &j = jobs[last];
            const size_t nb = j.status == RPH_OK ? (size_t)j.frame.total_blocks : 0;
            if (last > first && (blocks + nb) * 128 > CHUNK_COEF_BYTES) break;
            blocks += nb;
            last++;
        }
        const int b = k & 1;
        RPH_TRY(finish(b));
        Slot &S = P.slot[b];
        RPH_TRY(S.reserve(std::max(CHUNK_COEF_BYTES, blocks * 128), std::max<size_t>(last - first, std::min<size_t>(n, CHUNK_MAX_IMAGES))));
        {
            uint64_t fb = 0;
            for (size_t i = first; i < last; i++) {
                jobs[i].first_block = fb;
                if (jobs[i].status == RPH_OK) fb += jobs[i].frame.total_blocks;
            }
        }
        decode_chunk(jobs, first, last, S.h_coef, threads);

        // ---- descriptors
        const size_t m = last - first;
        JPlane *hp = reinterpret_cast<JPlane *>(S.h_desc);
        JImage *hi = reinterpret_cast<JImage *>(S.h_desc + m * 3 * sizeof(JPlane));
        uint16_t *hq = reinterpret_cast<uint16_t *>(S.h_desc + m * (3 * sizeof(JPlane) + sizeof(JImage)));
        uint32_t n_planes = 0, n_images = 0, max_blocks = 0, max_groups = 0;
        size_t plane_bytes = 0, out_bytes = 0;
        std::vector<uint32_t> image_of(m, UINT32_MAX);
        std::vector<size_t> out_off(m, 0);
        for (size_t i = first; i < last; i++) {
            Job &j = jobs[i];
            if (j.status != RPH_OK) continue;
            const rphj::Frame &f = j.frame;
            JImage im;
            memset(&im, 0, sizeof im);
            for (int c = 0; c < f.ncomp; c++) {
                const rphj::Comp &kc = f.comp[c];
                JPlane pl;
                pl.first_block = j.first_block + kc.first_block;
                pl.out_off = plane_bytes;
                pl.blocks_w = kc.blocks_w;
                pl.blocks_h = kc.blocks_h;
                pl.qt = n_planes;
                pl.pitch = kc.blocks_w * 8;
                memcpy(hq + (size_t)n_planes * 64, f.qt[kc.tq], 128);
                im.plane_off[c] = plane_bytes;
                im.pitch[c] = pl.pitch;
                plane_bytes += (size_t)pl.pitch * kc.blocks_h * 8;
                max_blocks = std::max(max_blocks, kc.blocks_w * kc.blocks_h);
                hp[n_planes++] = pl;
            }
            im.w = f.w;
            im.h = f.h;
            im.ncomp = (uint32_t)f.ncomp;
            im.hs = im.vs = 1;
            if (f.ncomp == 3) {
                im.hs = f.comp[0].H / f.comp[1].H;
                im.vs = f.comp[0].V / f.comp[1].V;
                im.cw = flavour == RPH_JPEG_LIBJPEG ? f.comp[1].samp_w : f.comp[1].blocks_w * 8;
                im.ch = flavour == RPH_JPEG_LIBJPEG ? f.comp[1].samp_h : f.comp[1].blocks_h * 8;
            }
            im.out_stride = (uint32_t)((size_t)f.ncomp * align_up(f.w, 4));
            im.out_off = out_bytes;
            out_off[i - first] = out_bytes;
            out_bytes += out_bytes_of(f);
            max_groups = std::max<uint32_t>(max_groups, (uint32_t)(((f.w + 3) / 4) * (size_t)f.h));
            image_of[i - first] = n_images;
            hi[n_images++] = im;
        }
        hipStream_t s = S.stream;
        uint8_t *d_hash = S.d_res, *d_q = d_hash + S.res_images * 32, *d_c = d_q + S.res_images * 4, *d_d = d_c + S.res_images * 1024,
                *d_v = d_d + S.res_images * 256;
        RPH_HIP_CHECK(hipMemsetAsync(S.d_res, 0, S.res_images * Slot::RES_BYTES, s));
        if (n_images) {
            RPH_HIP_CHECK(hipMemcpyAsync(S.d_coef, S.h_coef, blocks * 128, hipMemcpyHostToDevice, s));
            RPH_HIP_CHECK(hipMemcpyAsync(S.d_desc, S.h_desc, m * (3 * sizeof(JPlane) + sizeof(JImage) + 3 * 128), hipMemcpyHostToDevice, s));
            const JPlane *dp = reinterpret_cast<const JPlane *>(S.d_desc);
            const JImage *di = reinterpret_cast<const JImage *>(S.d_desc + m * 3 * sizeof(JPlane));
            const uint16_t *dq = reinterpret_cast<const uint16_t *>(S.d_desc + m * (3 * sizeof(JPlane) + sizeof(JImage)));
            const dim3 gi((max_blocks + 255) / 256, n_planes), gc((max_groups + 255) / 256, n_images);
            if (flavour == RPH_JPEG_LIBJPEG) {
                hipLaunchKernelGGL(jpeg_idct_kernel<RPH_JPEG_LIBJPEG>, gi, dim3(256), 0, s, S.d_coef, dq, dp, S.d_planes);
                hipLaunchKernelGGL(jpeg_color_kernel<RPH_JPEG_LIBJPEG>, gc, dim3(256), 0, s, S.d_planes, di, S.d_out);
            } else {
                hipLaunchKernelGGL(jpeg_idct_kernel<RPH_JPEG_ZUNE>, gi, dim3(256), 0, s, S.d_coef, dq, dp, S.d_planes);
                hipLaunchKernelGGL(jpeg_color_kernel<RPH_JPEG_ZUNE>, gc, dim3(256), 0, s, S.d_planes, di, S.d_out);
            }
            RPH_HIP_CHECK(hipGetLastError());
        }
        // ---- hash runs of equal geometry where the pixels lie (generate_pdq_features, scanner.rs:1410)
        if (out.want_hash) {
            for (size_t i = first; i < last;) {
                if (jobs[i].status != RPH_OK) {
                    i++;
                    continue;
                }
                const rphj::Frame &f = jobs[i].frame;
                size_t e = i + 1;
                while (e < last && jobs[e].status == RPH_OK && jobs[e].frame.w == f.w && jobs[e].frame.h == f.h && jobs[e].frame.ncomp == f.ncomp) e++;
                const size_t r0 = i - first;
                RPH_TRY(rph_pdq_hash_batch_dev(ctx, S.d_out + out_off[r0], (uint32_t)(e - i), f.w, f.h, (uint32_t)f.ncomp, (size_t)f.ncomp * align_up(f.w, 4),
                                               out_bytes_of(f), d_hash + r0 * 32, out.quality ? d_q + r0 * 4 : nullptr, out.coeffs ? d_c + r0 * 1024 : nullptr,
                                               out.dihedral ? d_d + r0 * 256 : nullptr, d_v + r0, s));
                i = e;
            }
            RPH_HIP_CHECK(hipMemcpyAsync(S.h_res, S.d_res, S.res_images * Slot::RES_BYTES, hipMemcpyDeviceToHost, s));
        }
        if (out.pixels && m == 1 && jobs[first].status == RPH_OK) {  // single-image decode: rows without their padding
            const rphj::Frame &f = jobs[first].frame;
            const size_t row = (size_t)f.ncomp * f.w, stride = (size_t)f.ncomp * align_up(f.w, 4);
            RPH_HIP_CHECK(hipMemcpy2DAsync(out.pixels, row, S.d_out, stride, row, f.h, hipMemcpyDeviceToHost, s));
        }
        pend[b].active = true;
        pend[b].first = first;
        pend[b].last = last;
        first = last;
    }
    RPH_TRY(finish(k & 1));
    RPH_TRY(finish((k + 1) & 1));
    int worst = RPH_OK;
    for (uint32_t i = 0; i < n; i++) {
        if (out.status) out.status[i] = jobs[i].status;
        if (jobs[i].status != RPH_OK) worst = jobs[i].status;
    }
    if (n == 1 && worst != RPH_OK) rph_set_error("rph_jpeg: not decodable (status %d)", worst);
    return n == 1 ? worst : RPH_OK;  // a batch reports per image (status / valid); one image reports itself
}

}  // namespace

void rph_jpeg_forget(rph_ctx *ctx)
{
    if (ctx->jpeg) {
        JpegPipe *P = static_cast<JpegPipe *>(ctx->jpeg);
        P->release();
        delete P;
        ctx->jpeg = nullptr;
    }
}

extern "C" {

int rph_jpeg_info(const uint8_t *data, size_t len, uint32_t *w, uint32_t *h, uint32_t *channels)
{
    return rph_guarded("rph_jpeg_info", [&]() -> int {
        if (!data || !w || !h || !channels) {
            rph_set_error("rph_jpeg_info: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        rphj::Frame f;
        const int rc = rphj::parse_frame(data, len, f);
        if (rc != RPH_OK) {
            rph_set_error("rph_jpeg_info: %s", rc == RPH_ERR_UNSUPPORTED ? "unsupported kind of JPEG" : "not a JPEG stream");
            return rc;
        }
        *w = f.w;
        *h = f.h;
        *channels = (uint32_t)f.ncomp;
        return RPH_OK;
    });
}

int rph_jpeg_coefficients(const uint8_t *data, size_t len, uint32_t *geometry, uint16_t *qt, int16_t *coef, size_t cap_blocks, uint64_t *total_blocks)
{
    return rph_guarded("rph_jpeg_coefficients", [&]() -> int {
        if (!data || !geometry || !qt || !total_blocks) {
            rph_set_error("rph_jpeg_coefficients: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        rphj::Frame f;
        int rc = rphj::parse_frame(data, len, f);
        if (rc != RPH_OK) return rc;
        *total_blocks = f.total_blocks;
        std::vector<int16_t> tmp;
        int16_t *dst = coef;
        if (!coef) {
            tmp.resize((size_t)f.total_blocks * 64);
            dst = tmp.data();
        } else if (cap_blocks < f.total_blocks) {
            rph_set_error("rph_jpeg_coefficients: %llu blocks, capacity %zu", (unsigned long long)f.total_blocks, cap_blocks);
            return RPH_ERR_CAPACITY;
        }
        rc = rphj::decode_coefficients(data, len, f, dst);
        if (rc != RPH_OK) {
            rph_set_error("rph_jpeg_coefficients: entropy decoding failed (status %d)", rc);
            return rc;
        }
        for (int c = 0; c < f.ncomp; c++) {
            const rphj::Comp &k = f.comp[c];
            uint32_t *g = geometry + 8 * c;
            g[0] = k.blocks_w, g[1] = k.blocks_h, g[2] = k.H, g[3] = k.V, g[4] = k.tq, g[5] = k.samp_w, g[6] = k.samp_h, g[7] = (uint32_t)k.first_block;
        }
        memcpy(qt, f.qt, sizeof f.qt);
        return RPH_OK;
    });
}

int rph_jpeg_decode(rph_ctx *ctx, const uint8_t *data, size_t len, int flavour, uint8_t *pixels_out)
{
    return rph_guarded("rph_jpeg_decode", [&]() -> int {
        if (!ctx || !data || !pixels_out) {
            rph_set_error("rph_jpeg_decode: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        Outputs o;
        o.pixels = pixels_out;
        o.want_hash = false;
        return run_batch(ctx, &data, &len, 1, flavour, 1, o);
    });
}

int rph_jpeg_pdq_hash_batch(rph_ctx *ctx, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads, uint8_t *hash32_out,
                            float *quality_out, float *coeffs_out, uint8_t *dihedral_out, uint8_t *valid_out, int32_t *status_out)
{
    return rph_guarded("rph_jpeg_pdq_hash_batch", [&]() -> int {
        if (!ctx || (n && (!data || !len)) || !hash32_out) {
            rph_set_error("rph_jpeg_pdq_hash_batch: null argument");
            return RPH_ERR_INVALID_ARG;
        }
        if (n == 0) return RPH_OK;
        Outputs o;
        o.hash = hash32_out;
        o.quality = quality_out;
        o.coeffs = coeffs_out;
        o.dihedral = dihedral_out;
        o.valid = valid_out;
        o.status = status_out;
        return run_batch(ctx, data, len, n, flavour, n_threads, o);
    });
}

}  // extern "C"
