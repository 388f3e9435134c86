// jpeg_kernels.hip -- the device kernels of the JPEG path (row N3 of SURVEY 8f); the pipeline around them is jpeg_pipeline.cpp.
//
// The reference decodes "jpg" | "jpeg" files with zune-jpeg 0.5.15 into Luma8 / Rgb8 and hands the pixels to generate_pdq_features
// (/root/reference/src/scanner.rs:473-508, :1410).  Here everything with arithmetic in it runs on the device, a batch of files per launch:
//   jpeg_huff_kernel   the Huffman walk of sequential files, one file -- or one restart interval, or one segment of a stream without
//                      markers (jpeg_sync_kernel) -- per lane: quantised coefficients written de-zigzagged into a dense buffer
//   jpeg_prog_kernel   the same for progressive files, one file per lane through all its scans (small batches of either kind are
//                      decoded by the host threads instead, jpeg_host.cpp)
//   jpeg_idct_kernel   one lane per 8x8 block: dequantise, integer IDCT (columns, rows) entirely in registers, level shift, clamp;
//                      128 B read and 64 B written per block -- HBM-bound, no LDS
//   jpeg_color_kernel  one lane per 8 output pixels: chroma upsampling as a pure function of the position (no intermediate
//                      full-resolution chroma planes), YCbCr -> RGB, then packed Rgb8, or the Rec.601 luma of it when only the hasher
//                      reads the pixels (Luma8 for one component)
// and then the PDQ launchers hash the pixels where they lie.
//
// Two arithmetic flavours (the tests' CPU checker restates the same pair):
//   RPH_JPEG_ZUNE (default)  zune-jpeg as recalled: stb_image's integer IDCT, (3 a + b + 2) >> 2 upsampling per direction, 45/32-style
//                            colour constants.  PARITY UNPINNED: the crate's source is not in the reference tree.
//   RPH_JPEG_LIBJPEG         libjpeg-turbo's defaults (jidctint.c islow, jdsample.c fancy upsampling, jdcolor.c): pinned by the tests
//                            against Pillow's decode of the reference's own JPEG files and of generated ones.

#include <vector>

#include "jpeg_device.h"
#include "rph_internal.h"

namespace {

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

// x's low popcount(m) bits to the set positions of m, the lowest to the lowest (Hacker's Delight 7-5, `expand`)
__device__ __forceinline__ unsigned long long deposit64(unsigned long long x, unsigned long long m)
{
    const unsigned long long m0 = m;
    unsigned long long mk = ~m << 1, mv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        unsigned long long mp = mk ^ (mk << 1);
        mp ^= mp << 2;
        mp ^= mp << 4;
        mp ^= mp << 8;
        mp ^= mp << 16;
        mp ^= mp << 32;
        mv[i] = mp & m;
        m = (m ^ mv[i]) | (mv[i] >> (1 << i));
        mk &= ~mp;
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        const unsigned long long t = x << (1 << i);
        x = (x & ~mv[i]) | (t & mv[i]);
    }
    return x & m0;
}

// zigzag position of the coefficient at natural index n
__device__ constexpr int unzigzag(int n)
{
    constexpr int zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    for (int p = 0; p < 64; p++)
        if (zz[p] == n) return p;
    return 0;
}

// One block: its 64 coefficients (with the refinements of a progressive file's scans added, below) -> 8 rows of 8 samples.
// Planes of progressive files walked on the device carry the records of their AC refinement scans (jpeg_device.h, PCorr): the
// corrections are added here, scan by scan in file order, by T.81 G.1.2.3's rule as libjpeg applies it -- a coefficient with history whose
// correction bit is set moves away from zero by 1 << Al, unless that bit of it is set already (a damaged stream).
template <int FL>
__device__ __forceinline__ void idct_block(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qts, const JPlane &pl, uint32_t b, uint32_t bx, uint32_t by,
                                           const PRef *__restrict__ refs, const PCorr *__restrict__ corr, const uint8_t *__restrict__ dcbits, uint2 (&rows)[8])
{
    const uint4 *src = reinterpret_cast<const uint4 *>(coef + (pl.first_block + b) * 64);
    const uint16_t *qt = qts + (size_t)pl.qt * 64;  // plane-uniform: scalar loads
    int32_t c[64];
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const uint4 r = src[y];
        const uint32_t u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c[8 * y + 2 * i] = (int32_t)(int16_t)(u[i] & 0xFFFFu);
            c[8 * y + 2 * i + 1] = (int32_t)(int16_t)(u[i] >> 16);
        }
    }
    if (pl.ref_count) {
        for (uint32_t q = 0; q < pl.ref_count; q++) {
            const PRef rf = refs[pl.ref_first + q];
            if (rf.al & PREF_DC) {  // a DC refinement scan: its bit of the block, if set, is the coefficient's bit Al (T.81 G.1.2.1)
                if (dcbits[(size_t)rf.corr_first + b]) c[0] |= 1 << (rf.al & 31u);
                continue;
            }
            if (bx >= pl.real_bw || by >= pl.real_bh) continue;
            const PCorr rc = corr[(size_t)rf.corr_first + by * pl.real_bw + bx];
            if (rc.bits == 0) continue;
            const unsigned long long dep = deposit64(rc.bits, rc.history);
            const uint32_t lo = (uint32_t)dep, hi = (uint32_t)(dep >> 32);
            const int32_t p1 = 1 << rf.al;
#pragma unroll
            for (int n = 1; n < 64; n++) {
                const int p = unzigzag(n);
                const uint32_t bit = ((p < 32 ? lo : hi) >> (p & 31)) & 1u;
                const int32_t v = c[n];
                c[n] = (bit && (v & p1) == 0) ? (v >= 0 ? v + p1 : v - p1) : v;
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 64; n++) c[n] = MUL(c[n], (int32_t)qt[n]);  // T.81 A.3.4
#pragma unroll
    for (int x = 0; x < 8; x++) {
        if (FL == RPH_JPEG_LIBJPEG)
            idct_islow<11>(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
        else
            idct_stb<512, 10>(c[x], c[8 + x], c[16 + x], c[24 + x], c[32 + x], c[40 + x], c[48 + x], c[56 + x]);
    }
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
        rows[y].x = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
        rows[y].y = (uint32_t)v[4] | ((uint32_t)v[5] << 8) | ((uint32_t)v[6] << 16) | ((uint32_t)v[7] << 24);
    }
}

// grid: x = groups of 256 blocks of a plane, y = plane.  Lane = block: neighbouring lanes write neighbouring 8-byte row segments.
// (The planes of an image that the fused kernel below takes are skipped.)
template <int FL>
__global__ void __launch_bounds__(256) jpeg_idct_kernel(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qts, const JPlane *__restrict__ planes,
                                                        uint8_t *__restrict__ out, const PRef *__restrict__ refs, const PCorr *__restrict__ corr,
                                                        const uint8_t *__restrict__ dcbits)
{
    const JPlane pl = planes[blockIdx.y];
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (pl.fused || b >= pl.blocks_w * pl.blocks_h) return;
    const uint32_t bx = b % pl.blocks_w, by = b / pl.blocks_w;
    uint2 rows[8];
    idct_block<FL>(coef, qts, pl, b, bx, by, refs, corr, dcbits, rows);
    uint8_t *dst = out + pl.out_off + (size_t)(by * 8) * pl.pitch + bx * 8;
#pragma unroll
    for (int y = 0; y < 8; y++) *reinterpret_cast<uint2 *>(dst + (size_t)y * pl.pitch) = rows[y];
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

__device__ __forceinline__ void bytes4(uint32_t u, int *d)
{
    d[0] = (int)(u & 255u), d[1] = (int)((u >> 8) & 255u), d[2] = (int)((u >> 16) & 255u), d[3] = (int)(u >> 24);
}

// The 8 chroma samples of the full-resolution grid at (x0 .. x0 + 7, y) of one chroma plane (x0 a multiple of 8).  The plane is read
// as aligned words -- the samples of the current and of the neighbouring chroma row under these 8 pixels plus one sample either side
// -- instead of a byte gather per pixel and neighbour (the first form of this kernel was bound by exactly those gathers).
//   libjpeg-turbo (jdsample.c): h2v1 / h2v2 / h1v2 "fancy" triangle filters; columns are replicated instead when the plane has <= 2 of them
//   zune-jpeg (recalled): (3 a + b + 2) >> 2 vertically, then the same horizontally on the result; the first and last column are copied
template <int FL>
__device__ __forceinline__ void chroma8(const uint8_t *__restrict__ plane, int pitch, int n, int rows, int x0, int y, int hs, int vs, int (&out)[8], int cols = 0)
{
    if (cols == 0) cols = pitch;  // samples per row of the plane (the fused kernel reads a tile of it in LDS: another pitch, the same columns)
    int r = vs == 2 ? (y >> 1) : y, rr = vs == 2 ? ((y & 1) ? r + 1 : r - 1) : r;
    r = r >= rows ? rows - 1 : r;  // (only rows of the padding, beyond the image, can exceed the plane's samples)
    rr = rr < 0 ? 0 : (rr >= rows ? rows - 1 : rr);  // the edge row repeats above and below (jdmainct.c context rows)
    const uint8_t *pr = plane + (size_t)r * pitch, *prr = plane + (size_t)rr * pitch;
    const bool lower = (y & 1) != 0;
    if (hs == 1) {  // chroma at full horizontal resolution: 8 samples of the row (pair)
        int a[8], b[8];
        const uint2 ua = *reinterpret_cast<const uint2 *>(pr + x0), ub = *reinterpret_cast<const uint2 *>(prr + x0);
        bytes4(ua.x, a), bytes4(ua.y, a + 4), bytes4(ub.x, b), bytes4(ub.y, b + 4);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (vs == 1)
                out[i] = a[i];
            else if (FL == RPH_JPEG_LIBJPEG)
                out[i] = (3 * a[i] + b[i] + (lower ? 2 : 1)) >> 2;  // h1v2_fancy_upsample
            else
                out[i] = (3 * a[i] + b[i] + 2) >> 2;
        }
        return;
    }
    // hs == 2: chroma columns c0 - 1 .. c0 + 4 under pixels x0 .. x0 + 7
    const int c0 = x0 >> 1;
    int a[6], b[6];
    bytes4(*reinterpret_cast<const uint32_t *>(pr + c0), a + 1);
    bytes4(*reinterpret_cast<const uint32_t *>(prr + c0), b + 1);
    const int cl = c0 > 0 ? c0 - 1 : 0, cr = c0 + 4 < cols ? c0 + 4 : cols - 1;
    a[0] = pr[cl], a[5] = pr[cr], b[0] = prr[cl], b[5] = prr[cr];
    if (FL == RPH_JPEG_LIBJPEG && n <= 2) {  // h2v1_upsample / h2v2_upsample: replication
#pragma unroll
        for (int i = 0; i < 8; i++) out[i] = a[1 + (i >> 1)];
        return;
    }
    int v[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        if (vs == 1)
            v[j] = a[j];
        else if (FL == RPH_JPEG_LIBJPEG)
            v[j] = 3 * a[j] + b[j];  // h2v2_fancy_upsample's column sums (scale 4)
        else
            v[j] = (3 * a[j] + b[j] + 2) >> 2;
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int j = 1 + (i >> 1), c = c0 + (i >> 1);
        const bool even = (i & 1) == 0;
        const bool edge = even ? c == 0 : c == n - 1;
        const int nb = even ? v[j - 1] : v[j + 1];
        if (FL == RPH_JPEG_LIBJPEG) {
            if (vs == 1)
                out[i] = edge ? v[j] : (3 * v[j] + nb + (even ? 1 : 2)) >> 2;                    // h2v1_fancy_upsample
            else
                out[i] = edge ? (v[j] * 4 + (even ? 8 : 7)) >> 4 : (3 * v[j] + nb + (even ? 8 : 7)) >> 4;  // h2v2_fancy_upsample
        } else {
            out[i] = edge ? v[j] : (3 * v[j] + nb + 2) >> 2;
        }
    }
}

// grid: x = groups of 256 lanes over ceil(w / 8) * h eight-pixel groups, y = image.  Packed Rgb8 (Luma8 for one component) with
// rows of ncomp * align8(w) bytes: what the PDQ kernels read.
template <int FL>
__global__ void __launch_bounds__(256) jpeg_color_kernel(const uint8_t *__restrict__ planes, const JImage *__restrict__ imgs, uint8_t *__restrict__ out)
{
    const JImage im = imgs[blockIdx.y];
    const uint32_t w8 = (im.w + 7) / 8;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (im.fused || t >= w8 * im.h) return;
    const int y = (int)(t / w8), x0 = (int)(t % w8) * 8;
    const uint2 yy = *reinterpret_cast<const uint2 *>(planes + im.plane_off[0] + (size_t)y * im.pitch[0] + x0);  // planes are padded to whole blocks
    uint8_t *dst = out + im.out_off + (size_t)y * im.out_stride;
    if (im.ncomp == 1) {
        *reinterpret_cast<uint2 *>(dst + x0) = yy;
        return;
    }
    int ys[8], cb[8], cr[8];
    bytes4(yy.x, ys), bytes4(yy.y, ys + 4);
    chroma8<FL>(planes + im.plane_off[1], (int)im.pitch[1], (int)im.cw, (int)im.ch, x0, y, (int)im.hs, (int)im.vs, cb);
    chroma8<FL>(planes + im.plane_off[2], (int)im.pitch[2], (int)im.cw, (int)im.ch, x0, y, (int)im.hs, (int)im.vs, cr);
    uint32_t px[24];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int r, g, b;
        ycc_to_rgb<FL>(ys[i], cb[i], cr[i], r, g, b);
        px[3 * i] = (uint32_t)r, px[3 * i + 1] = (uint32_t)g, px[3 * i + 2] = (uint32_t)b;
    }
    if (im.luma_out) {  // to_luma601 (pdqhash.rs:268-284) of the pixel just made: the Rgb8 image is never written nor read back
        uint32_t l[8];
#pragma unroll
        for (int i = 0; i < 8; i++) l[i] = (299u * px[3 * i] + 587u * px[3 * i + 1] + 114u * px[3 * i + 2] + 500u) / 1000u;
        *reinterpret_cast<uint2 *>(dst + x0) = make_uint2(l[0] | (l[1] << 8) | (l[2] << 16) | (l[3] << 24), l[4] | (l[5] << 8) | (l[6] << 16) | (l[7] << 24));
        return;
    }
    uint2 *d64 = reinterpret_cast<uint2 *>(dst + 3 * x0);
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const uint32_t *p = px + 8 * q;
        d64[q] = make_uint2(p[0] | (p[1] << 8) | (p[2] << 16) | (p[3] << 24), p[4] | (p[5] << 8) | (p[6] << 16) | (p[7] << 24));
    }
}

// IDCT + chroma upsampling + colour + Rec.601 luma of one tile of a three-component image whose pixels only the hasher reads (luma
// sampled 1 or 2 times the chroma in either direction: 4:2:0, 4:2:2, 4:4:0, 4:4:4).  A tile is 16 x 8 luma blocks = 128 x 64 samples and,
// per chroma plane, the blocks under them with a ring of blocks around (the upsampling filters look one sample beyond the tile; the ring's
// blocks are transformed whole).  The lanes transform one block each (all at once for 4:2:0, in two turns for the finer chroma grids) into
// LDS; then every lane makes 4 x 8 pixels out of LDS with the very functions of the two-kernel path (chroma8, ycc_to_rgb), so the results are
// the same bytes.  The sample planes are never written: 416 -> ~300 bytes of traffic per block at 4:2:0.
constexpr int FT_LBW = 16, FT_LBH = 8;                                // luma blocks per tile
constexpr int FT_LW = FT_LBW * 8, FT_LH = FT_LBH * 8;                 // luma samples
constexpr int FT_CMAX = (FT_LBW + 2) * 8 * (FT_LBH + 2) * 8;          // chroma samples in LDS at most (4:4:4: 18 x 10 blocks)
template <int FL>
__global__ void __launch_bounds__(256) jpeg_fused_kernel(const int16_t *__restrict__ coef, const uint16_t *__restrict__ qts, const JPlane *__restrict__ planes,
                                                         const JImage *__restrict__ imgs, uint8_t *__restrict__ out, const PRef *__restrict__ refs,
                                                         const PCorr *__restrict__ corr, const uint8_t *__restrict__ dcbits)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_y[FT_LW * FT_LH];
    __shared__ __attribute__((aligned(16))) uint8_t s_c[2][FT_CMAX];
    const JImage im = imgs[blockIdx.y];
    if (!im.fused) return;
    if (blockIdx.x >= im.tiles_x * im.tiles_y) return;
    const uint32_t tx = blockIdx.x % im.tiles_x, ty = blockIdx.x / im.tiles_x;
    const uint32_t t = threadIdx.x;
    const uint32_t hs = im.hs, vs = im.vs;
    const uint32_t cbw = FT_LBW / hs + 2, cbh = FT_LBH / vs + 2, cpitch = cbw * 8;  // chroma blocks with the ring; samples per LDS row
    const int cbx0 = (int)(tx * (FT_LBW / hs)) - 1, cby0 = (int)(ty * (FT_LBH / vs)) - 1;  // the chroma grid of LDS begins one block before the tile
    // ---- one block per lane and turn: 16 x 8 luma blocks, then cbw x cbh blocks of Cb, then of Cr
    const uint32_t n_blocks = FT_LBW * FT_LBH + 2 * cbw * cbh;
    for (uint32_t l = t; l < n_blocks; l += 256) {
        uint32_t comp, lbx, lby;  // component; block position in the tile's LDS grid
        if (l < FT_LBW * FT_LBH)
            comp = 0, lbx = l % FT_LBW, lby = l / FT_LBW;
        else {
            const uint32_t j = l - FT_LBW * FT_LBH, k = j % (cbw * cbh);
            comp = 1 + j / (cbw * cbh), lbx = k % cbw, lby = k / cbw;
        }
        const JPlane pl = planes[im.first_plane + comp];
        const int bx = comp == 0 ? (int)(tx * FT_LBW + lbx) : cbx0 + (int)lbx, by = comp == 0 ? (int)(ty * FT_LBH + lby) : cby0 + (int)lby;
        if (bx >= 0 && by >= 0 && bx < (int)pl.blocks_w && by < (int)pl.blocks_h) {
            uint2 rows[8];
            idct_block<FL>(coef, qts, pl, (uint32_t)by * pl.blocks_w + (uint32_t)bx, (uint32_t)bx, (uint32_t)by, refs, corr, dcbits, rows);
            const uint32_t pitch = comp == 0 ? (uint32_t)FT_LW : cpitch;
            uint8_t *dst = comp == 0 ? s_y + (lby * 8) * FT_LW + lbx * 8 : s_c[comp - 1] + (lby * 8) * cpitch + lbx * 8;
#pragma unroll
            for (int y = 0; y < 8; y++) *reinterpret_cast<uint2 *>(dst + y * pitch) = rows[y];
        }
    }
    __syncthreads();
    // ---- 8 pixels of a row per lane and turn, as jpeg_color_kernel makes them; the chroma tiles stand for their planes (a pointer such that
    // the plane's coordinates land in the tile: the filters clamp to the plane's own edges, which lie inside the tile or its ring)
    const int cy0 = cby0 * 8, cx0 = cbx0 * 8;  // plane coordinates of the chroma tiles' first sample
    const uint8_t *vcb = s_c[0] - (cy0 * (int)cpitch + cx0), *vcr = s_c[1] - (cy0 * (int)cpitch + cx0);
    const int cols = (int)im.pitch[1];
    uint8_t *img_out = out + im.out_off;
#pragma unroll
    for (int k = 0; k < (FT_LW / 8) * FT_LH / 256; k++) {
        const uint32_t g = t + 256 * k, row = g / (FT_LW / 8), xg = g % (FT_LW / 8);
        const int y = (int)(ty * FT_LH + row), x0 = (int)(tx * FT_LW + xg * 8);
        if (y >= (int)im.h || x0 >= (int)im.w) continue;
        const uint2 yy = *reinterpret_cast<const uint2 *>(s_y + row * FT_LW + xg * 8);
        int ys[8], cb[8], cr[8];
        bytes4(yy.x, ys), bytes4(yy.y, ys + 4);
        chroma8<FL>(vcb, (int)cpitch, (int)im.cw, (int)im.ch, x0, y, (int)hs, (int)vs, cb, cols);
        chroma8<FL>(vcr, (int)cpitch, (int)im.cw, (int)im.ch, x0, y, (int)hs, (int)vs, cr, cols);
        uint32_t l[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int r, gg, b;
            ycc_to_rgb<FL>(ys[i], cb[i], cr[i], r, gg, b);
            l[i] = (299u * (uint32_t)r + 587u * (uint32_t)gg + 114u * (uint32_t)b + 500u) / 1000u;  // to_luma601 (pdqhash.rs:268-284)
        }
        *reinterpret_cast<uint2 *>(img_out + (size_t)y * im.out_stride + x0) = make_uint2(l[0] | (l[1] << 8) | (l[2] << 16) | (l[3] << 24), l[4] | (l[5] << 8) | (l[6] << 16) | (l[7] << 24));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Entropy decoding on the device: ONE IMAGE PER LANE.
//
// A Huffman-coded scan is a serial bit stream, so a single image cannot be spread over lanes -- but a scan of a photo collection is
// hundreds of thousands of independent streams.  Each lane walks its own image symbol by symbol as a small state machine (DC / AC
// symbol, position in the block, block in the MCU, MCU in the scan); all lanes of a wave execute the same step, so there is no
// divergence beyond the rare long code.  The host only copies the entropy bytes to pinned memory with the byte stuffing undone and
// the RSTn markers dropped (jpeg_host.cpp: memchr + memcpy speed), so what crosses PCIe is the COMPRESSED file (a few tens of KB
// instead of 0.8 MB of coefficients), and the quantised coefficients are born in HBM.  Lanes are ordered by stream length so the
// images of a wave finish together.  This kernel takes sequential (baseline) files; a progressive file revisits every block in up to
// ten scans and has a kernel of its own below.  Latency per image is milliseconds (the walk is serial), so this is the path for large batches
// only (rph_jpeg_set_entropy).
// ---------------------------------------------------------------------------------------------------------------------------
__constant__ uint8_t c_zigzag[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                     6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                     39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct __attribute__((packed, aligned(4))) W2 {
    uint32_t x, y;
};

template <class T>
__device__ __forceinline__ T sel3(uint32_t i, T a, T b, T c)
{
    return i == 0 ? a : (i == 1 ? b : c);
}

// LDS_TABLES > 0: the chunk has at most that many distinct Huffman tables (a photo collection usually has the four Annex K tables) and
// every wave keeps a copy in LDS: the table lookup is on the critical path of every symbol, and an LDS read returns in a tenth of the
// time of a cached global read.
// STAGE_WAVES > 0: the workgroup has that many waves, which share the copy of the tables, and every lane builds its current block in LDS and
// writes it out whole when the block is complete -- eight 16-byte stores of its own 128-byte line instead of a 2-byte store per coefficient
// into 64 different lines per instruction.  That is what the launches with many lanes (segments, restart intervals) are bound by: per
// chunk of 2 M segment lanes the walk takes 27.7 ms, 16.9 ms with the stores removed (and 45 against 10.5 ms at twice the occupancy: the
// partly written lines of 1 024 lanes per CU do not fit the L2).
template <int LDS_TABLES, int STAGE_WAVES>
__global__ void __launch_bounds__(STAGE_WAVES > 0 ? STAGE_WAVES * 64 : 64) jpeg_huff_kernel(const uint8_t *__restrict__ streams, const HImage *__restrict__ imgs,
                                                       const HItem *__restrict__ items,
                                                       const uint32_t *__restrict__ order, uint32_t n_ordered, uint32_t n, const rphj::DeviceLut *__restrict__ g_luts,
                                                       uint32_t n_luts, int16_t *__restrict__ coef, uint8_t *__restrict__ status)
{
    constexpr uint32_t THREADS = STAGE_WAVES > 0 ? STAGE_WAVES * 64 : 64;
    __shared__ uint8_t zz[80];
    __shared__ __attribute__((aligned(16))) rphj::DeviceLut s_luts[LDS_TABLES > 0 ? LDS_TABLES : 1];
    __shared__ uint4 s_blk[STAGE_WAVES > 0 ? 8 * THREADS : 1];  // [row of the block][thread]: the lane's block, natural order
    for (uint32_t t = threadIdx.x; t < 80; t += THREADS) zz[t] = c_zigzag[t];
    if (LDS_TABLES > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(g_luts);
        uint4 *dst = reinterpret_cast<uint4 *>(s_luts);
        const uint32_t words = n_luts * (uint32_t)(sizeof(rphj::DeviceLut) / 16);
        for (uint32_t t = threadIdx.x; t < words; t += THREADS) dst[t] = src[t];
    }
    if (STAGE_WAVES > 0) {
#pragma unroll
        for (int r = 0; r < 8; r++) s_blk[r * THREADS + threadIdx.x] = make_uint4(0, 0, 0, 0);
    }
    // the lane's coefficient n of its current block (natural order): row n >> 3 of s_blk, half-word n & 7
    int16_t *const my_blk = reinterpret_cast<int16_t *>(s_blk + threadIdx.x);
    auto put = [&](uint64_t base, uint32_t nat, int16_t v) {
        if (STAGE_WAVES > 0)
            my_blk[(nat >> 3) * (THREADS * 8) + (nat & 7)] = v;
        else
            coef[base + nat] = v;
    };
    auto flush = [&](uint64_t base) {  // the finished block goes out as one line, and the lane's LDS block is empty again
        if (STAGE_WAVES > 0) {
            uint4 *dst = reinterpret_cast<uint4 *>(coef + base);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                dst[r] = s_blk[r * THREADS + threadIdx.x];
                s_blk[r * THREADS + threadIdx.x] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    const rphj::DeviceLut *luts = LDS_TABLES > 0 ? s_luts : g_luts;
    __syncthreads();
    const uint32_t slot = blockIdx.x * THREADS + threadIdx.x;
    if (slot >= n) return;
    const HItem item = items[slot < n_ordered ? order[slot] : slot];  // the first n_ordered items longest first; the segments' items behind them as they lie
    const uint32_t ii = item.image;
    const HImage *im = imgs + ii;
    const uint64_t img_fb = im->first_block;
    uint32_t bad = 0;
    const bool part = item.scan != HITEM_ALL_SCANS;  // one restart interval of the image's only scan
    const uint32_t sc_end = part ? item.scan + 1 : im->n_scans;
    for (uint32_t sc = part ? item.scan : 0; sc < sc_end && !bad; sc++) {
        const HScan *S = &im->scan[sc];
        const uint32_t ns = S->ns, ri = part ? 0 : S->restart_interval;
        // the scan's components (scan order); a single-component scan walks the component's own block grid (T.81 A.2.2)
        uint32_t H0, H1, H2, V0, V1, V2, BW0, BW1, BW2, FB0, FB1, FB2;
        const rphj::DeviceLut *D0, *D1, *D2, *A0, *A1, *A2;
        {
            const HComp *c0 = &im->comp[S->ci[0]], *c1 = &im->comp[S->ci[ns > 1 ? 1 : 0]], *c2 = &im->comp[S->ci[ns > 2 ? 2 : 0]];
            H0 = ns == 1 ? 1 : c0->H, V0 = ns == 1 ? 1 : c0->V, BW0 = c0->blocks_w, FB0 = c0->first_block;
            H1 = c1->H, V1 = c1->V, BW1 = c1->blocks_w, FB1 = c1->first_block;
            H2 = c2->H, V2 = c2->V, BW2 = c2->blocks_w, FB2 = c2->first_block;
            D0 = luts + S->dc[0], A0 = luts + S->ac[0];
            D1 = luts + S->dc[ns > 1 ? 1 : 0], A1 = luts + S->ac[ns > 1 ? 1 : 0];
            D2 = luts + S->dc[ns > 2 ? 2 : 0], A2 = luts + S->ac[ns > 2 ? 2 : 0];
        }
        const uint32_t MX = ns == 1 ? im->comp[S->ci[0]].real_bw : im->mcus_x, MY = ns == 1 ? im->comp[S->ci[0]].real_bh : im->mcus_y;
        const uint32_t per_mcu = H0 * V0 + (ns > 1 ? H1 * V1 : 0) + (ns > 2 ? H2 * V2 : 0);
        const uint64_t max_it = (part ? (uint64_t)item.mcu_count : (uint64_t)MX * MY) * per_mcu * 65 + 8;  // a block takes at most 1 + 63 symbols: the walk always ends
        // bit reader: `off` bytes of the scan consumed into acc (MSB first), nb valid bits
        const uint32_t skip = part ? item.stream_off : 0;
        const uint8_t *sp = streams + im->stream_base + S->off + skip;
        const uint32_t slen = S->len - (skip < S->len ? skip : S->len);
        // acc: the next bits, MSB first, nb of them valid.  Behind it two 8-byte words: q0 ready, q1 in flight (loaded one word ahead, so a
        // refill never waits for memory), each fetched with ONE load instruction from a 4-byte-aligned address: the per-CU address unit,
        // which takes a fully divergent wave access lane by lane, is what bounds this kernel, so loads per symbol are what matters.
        const uint32_t lead = (uint32_t)((uintptr_t)sp & 3);
        const uint8_t *sbase = sp - lead;
        const uint32_t limit = ((slen + lead + 3) & ~3u) + 8;  // 32 zero bytes lie behind the scan: reading on is safe, and a corrupt stream reads zeros there for ever
        auto load8 = [&](uint32_t at) -> W2 { return *reinterpret_cast<const W2 *>(sbase + (at < limit ? at : limit)); };
        auto be64 = [](W2 v) -> uint64_t { return ((uint64_t)__builtin_bswap32(v.x) << 32) | __builtin_bswap32(v.y); };
        uint64_t acc = (uint64_t)(__builtin_bswap32(*reinterpret_cast<const uint32_t *>(sbase)) << (8 * lead)) << 32;
        int nb = 32 - 8 * (int)lead;
        if (part) {  // a segment of a stream without markers begins mid-byte (at least 8 bits are in the accumulator)
            acc <<= item.bit_skip;
            nb -= (int)item.bit_skip;
        }
        uint64_t q0 = be64(load8(4));
        W2 q1 = load8(12);  // kept as loaded: its bytes are swapped when it becomes q0, a word later, so nothing waits on the load now
        uint32_t q0n = 64, woff = 20;
        // position: component i of the MCU, block (h, v) of the component, MCU (mx, my); k = next coefficient index (zigzag order)
        const uint32_t m0 = part && MX ? item.mcu_first : 0;
        uint32_t i = 0, h = 0, v = 0, mx = MX ? m0 % MX : 0, my = MX ? m0 / MX : 0, k = 0, until = ri, left = part ? item.mcu_count : 0xFFFFFFFFu;
        uint32_t Hc = H0, Vc = V0, BWc = BW0, FBc = FB0;
        const rphj::DeviceLut *DCc = D0, *ACc = A0;
        int p0 = part ? item.dc[0] : 0, p1 = part ? item.dc[1] : 0, p2 = part ? item.dc[2] : 0;
        bool is_dc = true, done = MX == 0 || MY == 0 || my >= MY || left == 0;
        uint64_t base = (img_fb + FBc + (uint64_t)(my * Vc) * BWc + mx * Hc) * 64;
        for (uint64_t it = 0; !done && it < max_it; it++) {
            if (nb < 32) {  // 4 more bytes
                acc |= (q0 >> 32) << (32 - nb);
                nb += 32;
                q0 <<= 32;
                q0n -= 32;
                if (q0n == 0) {
                    q0 = be64(q1);
                    q0n = 64;
                    q1 = load8(woff);
                    woff += 8;
                }
            }
            const rphj::DeviceLut *L = is_dc ? DCc : ACc;
            const uint32_t e = L->look[(uint32_t)(acc >> 54)];
            uint32_t len, sym;
            if (e) {
                len = e >> 8;
                sym = e & 255;
            } else {  // a code longer than 10 bits
                const int32_t win = (int32_t)(acc >> 48);
                uint32_t l = 11;
                while (l <= 16 && win >= L->maxcode[l]) l++;
                if (l > 16) {
                    bad = 1;
                    break;
                }
                len = l;
                sym = L->sym[(uint32_t)((win >> (16 - l)) + L->delta[l]) & 255];
            }
            acc <<= len;
            nb -= (int)len;
            const uint32_t s = is_dc ? sym : (sym & 15), r = is_dc ? 0 : (sym >> 4);
            if (s > 15) {
                bad = 1;
                break;
            }
            int val = 0;
            if (s) {  // s more bits: the value, T.81 F.2.2.1 EXTEND
                const uint32_t raw = (uint32_t)(acc >> (64 - s));
                acc <<= s;
                nb -= (int)s;
                val = raw < (1u << (s - 1)) ? (int)raw - (int)((1u << s) - 1) : (int)raw;
            }
            if (is_dc) {
                const int pv = sel3(i, p0, p1, p2) + val;
                p0 = i == 0 ? pv : p0;
                p1 = i == 1 ? pv : p1;
                p2 = i == 2 ? pv : p2;
                put(base, 0, (int16_t)pv);
                k = 1;
                is_dc = false;
            } else if (s == 0) {
                k = r == 15 ? k + 16 : 64;  // ZRL, or end of block
            } else {
                k += r;
                put(base, zz[k < 79 ? k : 79], (int16_t)val);
                k++;
            }
            if (k > 64) {  // a run or ZRL stepped past coefficient 63: the file is damaged (the host decoder's rule, jpeg_host.cpp: kk > 64)
                bad = 1;
                break;
            }
            if (k >= 64) {  // next block
                flush(base);
                is_dc = true;
                k = 0;
                if (++h == Hc) {
                    h = 0;
                    if (++v == Vc) {
                        v = 0;
                        if (++i == ns) {
                            i = 0;
                            if (ri && --until == 0) {  // restart interval: drop the padding bits, reset the predictions (the marker itself is gone)
                                const int drop = nb & 7;
                                acc <<= drop;
                                nb -= drop;
                                p0 = p1 = p2 = 0;
                                until = ri;
                            }
                            if (++mx == MX) {
                                mx = 0;
                                if (++my == MY) done = true;
                            }
                            if (--left == 0) done = true;
                        }
                        Hc = sel3(i, H0, H1, H2);
                        Vc = sel3(i, V0, V1, V2);
                        BWc = sel3(i, BW0, BW1, BW2);
                        FBc = sel3(i, FB0, FB1, FB2);
                        DCc = sel3(i, D0, D1, D2);
                        ACc = sel3(i, A0, A1, A2);
                    }
                }
                base = (img_fb + FBc + (uint64_t)(my * Vc + v) * BWc + (mx * Hc + h)) * 64;
            }
        }
        if (!done) bad = 1;
    }
    if (bad) status[ii] = 1;  // (the results were zeroed before the launch; several lanes may share an image)
}


// ---------------------------------------------------------------------------------------------------------------------------
// Progressive files (T.81 G.1.2), one SCAN per lane, each over its component's whole block grid -- first DC (difference coding,
// value << Al), DC refinement (one bit per block), first AC of a band (runs, end-of-band runs over blocks), AC refinement (a correction
// bit for every coefficient that is already nonzero, new +-1 values in between).  A scan needs the scans before it that touch the same
// coefficients of the same component, and no others, and it needs them block by block: all scans of a chunk are ONE launch in which a
// scan follows its predecessors a few blocks behind (progress words, jpeg_device.h; the host orders the waves so that a producer
// starts before its consumers).  Nothing here ever waits for a coefficient to come back from memory, and no coefficient is written
// twice.  First scans only write.  Refinement needs to know WHICH coefficients of a block are nonzero, not their values: one 64-bit
// mask per block (zigzag position = bit) is kept beside the coefficients -- first AC scans and the placements of refinement scans OR
// into it with device-scope atomics, a refinement scan reads the words of the next eight blocks (device-scope atomic loads) while it
// walks the current eight out of LDS.  The corrections of an AC refinement scan are collected, not applied: per block the history the
// scan saw and the correction bits; a DC refinement scan leaves its bit per block; the IDCT kernel adds both to the coefficients with a
// lane per block.  Every access stays inside the file's own blocks and records whatever a damaged stream says.
// ---------------------------------------------------------------------------------------------------------------------------
struct LongCodes {  // of one Huffman table: exclusive upper bounds of the codes of 9..16 bits in a 16-bit window; symbol index = (window >> (16 - length)) + delta[length],
    int32_t maxc[8], dlt[8];  // dlt[0] = delta[9] - n_short, dlt[j] = delta[9 + j] - delta[8 + j]: the bounds grow with the length, so the reached ones add up to delta[length] - n_short
    uint32_t n_short;         // codes of up to 8 bits = index of the first longer code's symbol
};
constexpr int PROG_LONG_SYMS = 48;  // symbols of codes longer than 8 bits a progressive lane keeps in LDS
struct BitR {
    const uint8_t *sbase;
    uint32_t limit, q0n, woff;
    uint64_t acc, q0;
    W2 q1;
    int nb;
    __device__ __forceinline__ W2 load8(uint32_t at) const { return *reinterpret_cast<const W2 *>(sbase + (at < limit ? at : limit)); }
    static __device__ __forceinline__ uint64_t be64(W2 v) { return ((uint64_t)__builtin_bswap32(v.x) << 32) | __builtin_bswap32(v.y); }
    __device__ __forceinline__ void init(const uint8_t *sp, uint32_t slen)
    {
        const uint32_t lead = (uint32_t)((uintptr_t)sp & 3);
        sbase = sp - lead;
        limit = ((slen + lead + 3) & ~3u) + 8;  // 32 zero bytes lie behind the scan
        acc = (uint64_t)(__builtin_bswap32(*reinterpret_cast<const uint32_t *>(sbase)) << (8 * lead)) << 32;
        nb = 32 - 8 * (int)lead;
        q0 = be64(load8(4));
        q1 = load8(12);
        q0n = 64;
        woff = 20;
    }
    __device__ __forceinline__ void fill()  // at least 32 valid bits
    {
        if (nb < 32) {
            acc |= (q0 >> 32) << (32 - nb);
            nb += 32;
            q0 <<= 32;
            q0n -= 32;
            if (q0n == 0) {
                q0 = be64(q1);
                q0n = 64;
                q1 = load8(woff);
                woff += 8;
            }
        }
    }
    __device__ __forceinline__ uint32_t take(uint32_t n)  // n in 1..31 bits, MSB first
    {
        const uint32_t v = (uint32_t)(acc >> (64 - n));
        acc <<= n;
        nb -= (int)n;
        return v;
    }
    __device__ __forceinline__ uint32_t take32(uint32_t n)  // n in 0..32 bits
    {
        const uint32_t v = (uint32_t)((acc >> 1) >> (63 - n));
        acc <<= n;
        nb -= (int)n;
        return v;
    }
    // one Huffman symbol; 0x100 = not a code of this table.  The lane's own copy of its scan's table in LDS: look8 = 8-bit lookup
    // ([prefix * 64 + lane]; 0 = longer code) and the symbols (syms: [index * 64 + lane]); for the longer codes the canonical arrays in
    // registers (the bounds grow with the length, so the length is 9 + the number of bounds the window has reached).  With the long codes
    // left in memory nearly every step of a wave paid for one: each lane meets one only every few dozen symbols, but one lane in 64 is enough.
    __device__ __forceinline__ uint32_t symbol8(const uint16_t *look8, const LongCodes &lc, const uint8_t *long_syms, const uint8_t *all_syms, uint32_t lane)
    {
        const uint32_t e = look8[(uint32_t)(acc >> 56) * 64 + lane];
        uint32_t len, sym;
        if (e) {
            len = e >> 8;
            sym = e & 255;
        } else {
            const int32_t win = (int32_t)(acc >> 48);
            uint32_t over = 0;
            int32_t d = lc.dlt[0];
#pragma unroll
            for (int j = 0; j < 8; j++) {  // (dlt[j] for j >= 1: the step from the length before)
                const bool reached = win >= lc.maxc[j];
                over += reached ? 1u : 0u;
                if (j < 7) d += reached ? lc.dlt[j + 1] : 0;
            }
            if (over == 8) return 0x100;
            len = 9 + over;
            // (dlt[0] has the number of shorter codes taken off: the index counts from the first code of nine bits, whose symbols lie in LDS --
            // the first PROG_LONG_SYMS of them, the likeliest; a table with more long codes reads the rest from memory)
            const uint32_t at = (uint32_t)((win >> (16 - len)) + d) & 255u;
            sym = at < (uint32_t)PROG_LONG_SYMS ? long_syms[at * 64 + lane] : all_syms[(at + lc.n_short) & 255u];
        }
        acc <<= len;
        nb -= (int)len;
        return sym;
    }
    __device__ __forceinline__ uint32_t symbol(const rphj::DeviceLut *L)
    {
        const uint32_t e = L->look[(uint32_t)(acc >> 54)];
        uint32_t len, sym;
        if (e) {
            len = e >> 8;
            sym = e & 255;
        } else {
            const int32_t win = (int32_t)(acc >> 48);
            uint32_t l = 11;
            while (l <= 16 && win >= L->maxcode[l]) l++;
            if (l > 16) return 0x100;
            len = l;
            sym = L->sym[(uint32_t)((win >> (16 - l)) + L->delta[l]) & 255];
        }
        acc <<= len;
        nb -= (int)len;
        return sym;
    }
};

constexpr int PROG_GROUP = 8;  // blocks of history an AC refinement lane holds in LDS at a time

// How far a scan has come, for the scans that follow it (jpeg_device.h): published by all lanes of a wave together, every PROG_PUBLISH
// steps.  Everything a following scan reads of this one inside the launch -- the mask words -- is written with device-scope atomics and
// read with device-scope atomic loads, which meet at the device's point of coherence, not in a cache of their own: what has to be
// ordered is only "the atomics before, then the progress word", and waiting for the outstanding memory operations does that (a
// workgroup-scope release: a device-scope one would also write back the whole L2 of the XCD -- at every publication of every wave).
constexpr uint32_t PROG_PUBLISH = 256;
__device__ __forceinline__ void prog_publish(uint32_t *progress, uint32_t me, uint32_t done)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store(progress + me, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// All lanes of the wave wait until the scans they depend on have come at least as far as they need (NONE: no such scan).  The producers
// are in workgroups that started before this one; should one never arrive all the same, the wait ends after a few seconds and the
// file is reported as damaged rather than hanging the device.  (The mask loads that follow are issued after the progress words have
// come back, and they are device-scope atomic loads.)
__device__ __forceinline__ bool prog_wait(const uint32_t *progress, uint32_t dep0, uint32_t dep1, uint32_t need)
{
    bool ok = false;
    for (uint32_t spin = 0; spin < (1u << 22); spin++) {
        ok = (dep0 == PSCAN_NONE || __hip_atomic_load(progress + dep0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) &&
             (dep1 == PSCAN_NONE || __hip_atomic_load(progress + dep1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need);
        if (__all((int)ok)) break;
        __builtin_amdgcn_s_sleep(32);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return ok;
}

template <int LDS_TABLES>
__global__ void __launch_bounds__(64) jpeg_prog_kernel(const uint8_t *__restrict__ streams, const HImage *__restrict__ imgs, const PScan *__restrict__ pscans,
                                                       const uint32_t *__restrict__ items, uint32_t n, const uint32_t *__restrict__ waits,
                                                       const rphj::DeviceLut *__restrict__ g_luts, uint32_t n_luts,
                                                       int16_t *__restrict__ coef, unsigned long long *__restrict__ masks, uint32_t *__restrict__ progress,
                                                       PCorr *__restrict__ corr, uint8_t *__restrict__ dcbits, uint8_t *__restrict__ status)
{
    __shared__ uint8_t zz[80];
    __shared__ __attribute__((aligned(16))) rphj::DeviceLut s_luts[LDS_TABLES > 0 ? LDS_TABLES : 1];
    // the AC table of the lane's scan as an 8-bit lookup of its own: progressive files carry tables optimised per scan, so a chunk has
    // thousands of distinct ones and they stay in global memory -- a probe there is ~1 us on the critical path of every symbol
    __shared__ uint16_t s_look8[256 * 64];
    __shared__ uint8_t s_syms[PROG_LONG_SYMS * 64];  // the symbols of the first codes of 9..16 bits
    __shared__ unsigned long long s_ring[PROG_GROUP * 64];
    for (int t = threadIdx.x; t < 80; t += 64) zz[t] = c_zigzag[t];
    if (LDS_TABLES > 0) {
        const uint4 *src = reinterpret_cast<const uint4 *>(g_luts);
        uint4 *dst = reinterpret_cast<uint4 *>(s_luts);
        const uint32_t words = n_luts * (uint32_t)(sizeof(rphj::DeviceLut) / 16);
        for (uint32_t t = threadIdx.x; t < words; t += 64) dst[t] = src[t];
    }
    const rphj::DeviceLut *luts = LDS_TABLES > 0 ? s_luts : g_luts;
    __syncthreads();
    const uint32_t slot = blockIdx.x * 64 + threadIdx.x, lane = threadIdx.x;
    if (slot >= n) return;
    const uint32_t me = items[slot];
    if (me == PSCAN_NONE) return;  // (a batch of fewer than 64 files)
    const PScan P = pscans[me];
    const uint32_t ii = P.image;
    const HImage *im = imgs + ii;
    const uint64_t img_fb = im->first_block;
    unsigned long long *const my_masks = masks + (size_t)im->mask_first;  // nonzero positions per block of this image
    uint32_t bad = 0;
#ifdef RPH_PROG_TIMING
    const uint64_t t_scan = wall_clock64();
    uint32_t t_steps = 0;
#define RPH_PROG_STEP() t_steps++
#else
#define RPH_PROG_STEP()
#endif
    BitR b;
    b.init(streams + im->stream_base + P.off, P.len);
    const uint32_t al = P.al;
    {  // the scans that must have ended (all lanes of the wave take as many turns as the longest list needs)
        uint32_t w = 0;
        while (__any((int)(w < P.wait_count))) {
            const uint32_t dep = w < P.wait_count ? waits[P.wait_first + w] : PSCAN_NONE;
            if (!prog_wait(progress, dep, PSCAN_NONE, 0xFFFFFFFFu)) bad = 1;
            w++;
        }
    }
    if (P.ss == 0) {
        // ---- DC scan: MCU order (one component: its own block grid, T.81 A.2.2)
        const uint32_t ns = P.ns;
        const HComp *c0 = &im->comp[P.ci[0]], *c1 = &im->comp[P.ci[ns > 1 ? 1 : 0]], *c2 = &im->comp[P.ci[ns > 2 ? 2 : 0]];
        const uint32_t H0 = ns == 1 ? 1 : c0->H, V0 = ns == 1 ? 1 : c0->V, BW0 = c0->blocks_w, FB0 = c0->first_block;
        const uint32_t H1 = c1->H, V1 = c1->V, BW1 = c1->blocks_w, FB1 = c1->first_block;
        const uint32_t H2 = c2->H, V2 = c2->V, BW2 = c2->blocks_w, FB2 = c2->first_block;
        const rphj::DeviceLut *D0 = luts + P.dc[0], *D1 = luts + P.dc[ns > 1 ? 1 : 0], *D2 = luts + P.dc[ns > 2 ? 2 : 0];
        const uint32_t MX = ns == 1 ? c0->real_bw : im->mcus_x, MY = ns == 1 ? c0->real_bh : im->mcus_y;
        uint32_t i = 0, h = 0, v = 0, mx = 0, my = 0;
        int p0 = 0, p1 = 0, p2 = 0;
        bool done = MX == 0 || MY == 0;
        while (!done) {
            RPH_PROG_STEP();
            b.fill();
            const uint32_t Hc = sel3(i, H0, H1, H2), Vc = sel3(i, V0, V1, V2), BWc = sel3(i, BW0, BW1, BW2), FBc = sel3(i, FB0, FB1, FB2);
            const uint64_t base = (img_fb + FBc + (uint64_t)(my * Vc + v) * BWc + (mx * Hc + h)) * 64;
            if (P.ah == 0) {
                const uint32_t s = b.symbol(sel3(i, D0, D1, D2));
                if (s > 15) {
                    bad = 1;
                    break;
                }
                int val = 0;
                if (s) {
                    const uint32_t raw = b.take(s);
                    val = raw < (1u << (s - 1)) ? (int)raw - (int)((1u << s) - 1) : (int)raw;
                }
                const int pv = sel3(i, p0, p1, p2) + val;
                p0 = i == 0 ? pv : p0;
                p1 = i == 1 ? pv : p1;
                p2 = i == 2 ? pv : p2;
                coef[base] = (int16_t)(pv * (1 << al));
            } else if (b.take(1)) {  // (kept beside the coefficients: the IDCT kernel puts the bit in)
                dcbits[(size_t)sel3(i, P.dcb[0], P.dcb[1], P.dcb[2]) + (size_t)(my * Vc + v) * BWc + (mx * Hc + h)] = 1;
            }
            if (++h == Hc) {
                h = 0;
                if (++v == Vc) {
                    v = 0;
                    if (++i == ns) {
                        i = 0;
                        if (++mx == MX) {
                            mx = 0;
                            if (++my == MY) done = true;
                        }
                    }
                }
            }
        }
    } else {
        // ---- AC scan: one component, its own block grid in raster order
        const HComp *c = &im->comp[P.ci[0]];
        const uint32_t MX = c->real_bw, MY = c->real_bh, BW = c->blocks_w, total = MX * MY;
        const uint64_t comp_base = img_fb + c->first_block;
        const rphj::DeviceLut *A = luts + P.ac;
        const uint32_t ss = P.ss, se = P.se;
        for (uint32_t i0 = 0; i0 < 256; i0 += 16) {
            uint16_t e[16];
#pragma unroll
            for (int j = 0; j < 16; j++) e[j] = A->look[(i0 + j) << 2];
#pragma unroll
            for (int j = 0; j < 16; j++) s_look8[(i0 + j) * 64 + lane] = (e[j] >> 8) <= 8 ? e[j] : (uint16_t)0;
        }
        LongCodes lc;  // the canonical arrays for codes of 9..16 bits stay in registers: with 64 lanes nearly every step has a lane that needs them
#pragma unroll
        for (int j = 0; j < 8; j++) {
            lc.maxc[j] = A->maxcode[9 + j];
            lc.dlt[j] = j ? A->delta[9 + j] - A->delta[8 + j] : A->delta[9];
        }
        lc.n_short = (uint32_t)(A->delta[9] + ((A->maxcode[8] >> 8) << 1)) & 255u;  // the first 9-bit code's symbol index = the number of shorter codes
        lc.dlt[0] -= (int32_t)lc.n_short;
        for (uint32_t j0 = 0; j0 < (uint32_t)PROG_LONG_SYMS; j0 += 4) {
            uint8_t w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = A->sym[(lc.n_short + j0 + j) & 255u];
#pragma unroll
            for (int j = 0; j < 4; j++) s_syms[(j0 + j) * 64 + lane] = w[j];
        }
        if (P.ah == 0) {
            // first pass over the band: one symbol per step
            uint32_t bl = 0, k = ss, col = 0;
            uint64_t base = comp_base * 64, row_base = comp_base * 64;
            unsigned long long nz_acc = 0;  // placements in the current block
            const uint64_t max_it = (uint64_t)total * 65 + 8;
            for (uint64_t it = 0; bl < total && it < max_it; it++) {
                RPH_PROG_STEP();
                if (((uint32_t)it & (PROG_PUBLISH - 1)) == PROG_PUBLISH - 1) prog_publish(progress, me, bl);  // (the masks of the blocks before bl are out)
                b.fill();
                const uint32_t rs = b.symbol8(s_look8, lc, s_syms, A->sym, lane);
                if (rs > 255) {
                    bad = 1;
                    break;
                }
                const uint32_t r = rs >> 4, s = rs & 15;
                uint32_t adv = 0;  // blocks to move on by
                if (s == 0) {
                    if (r == 15) {
                        k += 16;
                        if (k > se) adv = 1;
                    } else {  // end of band for this block and the next (1 << r) - 1 + bits
                        uint32_t run = (1u << r) - 1;
                        if (r) run += b.take(r);
                        adv = 1 + run;
                    }
                } else {
                    k += r;
                    if (k > 63) {
                        bad = 1;
                        break;
                    }
                    const uint32_t raw = b.take(s);
                    const int val = raw < (1u << (s - 1)) ? (int)raw - (int)((1u << s) - 1) : (int)raw;
                    coef[base + zz[k]] = (int16_t)(val * (1 << al));
                    nz_acc |= 1ull << k;
                    k++;
                    if (k > se) adv = 1;
                }
                if (adv) {
                    if (nz_acc) {
                        atomicOr(my_masks + (size_t)(base / 64 - img_fb), nz_acc);
                        nz_acc = 0;
                    }
                    bl = adv > total - bl ? total : bl + adv;
                    k = ss;
                    if (adv == 1) {  // (the common case without a division)
                        base += 64;
                        if (++col == MX) {
                            col = 0;
                            row_base += (uint64_t)BW * 64;
                            base = row_base;
                        }
                    } else if (bl < total) {
                        const uint32_t row = bl / MX;
                        col = bl - row * MX;
                        row_base = (comp_base + (uint64_t)row * BW) * 64;
                        base = row_base + (uint64_t)col * 64;
                    }
                }
            }
            if (bl < total) bad = 1;
        } else {
            // Refinement of the band.  Per block: nzb = the band's positions with nonzero history.  A step: a symbol (unless the block
            // lies in an end-of-band run), then the correction bits of the coefficients with history that come before the symbol's own
            // place -- the (r + 1)-th zero-history position from k on -- or, when there is no such place, up to the band's end.  The
            // correction bits are only COLLECTED here (they are the next popcount bits of the stream, taken at once): the block's
            // history and its correction bits go to the scan's record array, and jpeg_prog_apply_kernel adds them to the coefficients
            // with a lane per coefficient.  The history of the next PROG_GROUP blocks lies in LDS (fetched a group ahead), so the lanes
            // of a wave only meet at group boundaries and a slow block of one lane is averaged over the group, not paid by all 64.
            const int p1 = 1 << al;
            const uint64_t band = ((se >= 63 ? 0ull : (1ull << (se + 1))) - 1ull) & ~((1ull << ss) - 1ull);
            PCorr *const rec = corr + P.corr_first;
            uint32_t bl = 0, eobrun = 0, k = ss, col = 0, f_col = 0, f_bl = 0, cn = 0;
            uint64_t row_base = comp_base, f_row_base = comp_base;  // in blocks
            unsigned long long pf[PROG_GROUP];
            auto fetch_group = [&]() {  // the history words of the next PROG_GROUP blocks (raster order over the component's real blocks)
#pragma unroll
                for (int j = 0; j < PROG_GROUP; j++) {
                    pf[j] = 0;
                    if (f_bl < total) {
                        pf[j] = __hip_atomic_load(my_masks + (size_t)(f_row_base + f_col - img_fb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (written in this launch)
                        f_bl++;
                        if (++f_col == MX) {
                            f_col = 0;
                            f_row_base += BW;
                        }
                    }
                }
            };
            auto fetch_when_ready = [&]() {  // the scans this one follows must be past the blocks it is about to read the history of
                const uint32_t need = f_bl + PROG_GROUP < total ? f_bl + PROG_GROUP : total;
                if (!prog_wait(progress, P.chase[0], P.chase[1], f_bl < total ? need : 0u)) bad = 1;
                fetch_group();
            };
            fetch_when_ready();
            uint32_t steps = 0;
            unsigned long long nz_all = 0, nzb = 0, cb = 0, nz_new = 0;
            uint64_t base = 0;
            bool fresh = true;
            for (uint32_t g0 = 0; g0 < total && !bad; g0 += PROG_GROUP) {
#pragma unroll
                for (int j = 0; j < PROG_GROUP; j++) s_ring[j * 64 + lane] = pf[j];
                fetch_when_ready();
                const uint32_t gend = g0 + PROG_GROUP < total ? g0 + PROG_GROUP : total;
                while (bl < gend) {
                    if (fresh) {
                        nz_all = s_ring[(bl & (PROG_GROUP - 1)) * 64 + lane];
                        nzb = nz_all & band;
                        k = ss;
                        cb = 0;
                        cn = 0;
                        nz_new = 0;
                        base = (row_base + col) * 64;
                        fresh = false;
                    }
                    RPH_PROG_STEP();
                    if ((++steps & (PROG_PUBLISH - 1)) == 0) prog_publish(progress, me, bl);  // (the masks of the blocks before bl have their placements)
                    b.fill();
                    uint32_t r = 0;
                    int value = 0;
                    bool place = false;  // a symbol of this block asks for a place; else only corrections are due
                    if (eobrun == 0) {
                        const uint32_t rs = b.symbol8(s_look8, lc, s_syms, A->sym, lane);
                        if (rs > 255) {
                            bad = 1;
                            break;
                        }
                        r = rs >> 4;
                        const uint32_t s = rs & 15;
                        if (s) {
                            if (s != 1) {
                                bad = 1;
                                break;
                            }
                            value = b.take(1) ? p1 : -p1;
                            place = true;
                        } else if (r != 15) {
                            eobrun = 1u << r;
                            if (r) eobrun += b.take(r);
                        } else {
                            place = true;  // sixteen zero-history coefficients to step over, nothing to put down
                        }
                    }
                    const uint64_t from_k = ~0ull << k;  // k <= 63 here
                    uint64_t todo = nzb & from_k;         // coefficients with history that take a correction bit now
                    uint32_t at = 64;                     // the symbol's place
                    if (place) {
                        uint64_t zeros = ~nzb & band & from_k;
                        if ((uint32_t)__builtin_popcountll(zeros) > r) {
                            for (uint32_t q = 0; q < r; q++) zeros &= zeros - 1;
                            at = (uint32_t)__builtin_ctzll(zeros);
                            todo &= (1ull << at) - 1ull;
                        }
                    }
                    uint32_t nc = (uint32_t)__builtin_popcountll(todo);
                    while (nc) {  // (twice when more than 32 are due); the first bit read belongs to the lowest position: it goes lowest
                        b.fill();
                        const uint32_t t = nc < 32 ? nc : 32;
                        cb |= (unsigned long long)(__builtin_bitreverse32(b.take32(t)) >> (32 - t)) << cn;
                        cn += t;
                        nc -= t;
                    }
                    bool block_done = true;  // an end-of-band run, or the band ended inside the symbol's run of zeros
                    if (at < 64) {
                        if (value) {
                            coef[base + zz[at]] = (int16_t)value;
                            nz_new |= 1ull << at;
                        }
                        k = at + 1;
                        block_done = k > se;
                    }
                    if (block_done) {
                        if (nzb) *reinterpret_cast<ulonglong2 *>(rec + bl) = make_ulonglong2(nzb, cb);  // (the records were zeroed)
                        if (nz_new) atomicOr(my_masks + (size_t)(base / 64 - img_fb), nz_new);
                        if (eobrun > 0) eobrun--;
                        bl++;
                        if (++col == MX) {
                            col = 0;
                            row_base += BW;
                        }
                        fresh = true;
                    }
                }
            }
            if (bl < total) bad = 1;
        }
    }
#ifdef RPH_PROG_TIMING
    if (ii == 0) {
        const unsigned long long us = (unsigned long long)(wall_clock64() - t_scan) / 100;
        printf("scan ns=%u ss=%u se=%u ah=%u al=%u len=%u: %llu us, %u steps of this lane, %.2f us per step\n", P.ns, P.ss, P.se, P.ah, P.al, P.len, us, t_steps, (double)us / (t_steps ? t_steps : 1));
    }
#endif
    prog_publish(progress, me, 0xFFFFFFFFu);  // (whatever became of the scan: nobody waits for it any longer)
    if (bad) status[ii] = 1;  // (the results were zeroed before the launch; the scans of a file are different lanes)
}

// ---------------------------------------------------------------------------------------------------------------------------
// Segments of a stream without restart markers (jpeg_device.h): one lane per segment.  mode 0: decode from the segment boundary as if an
// MCU began there, record the MCU starts seen inside the segment (SegTable), report the first one behind it.  mode 1 (validation round
// `round`): the entry is the predecessor's `out` of the previous round; results that began there stand, an entry that round 0 recorded
// takes its results from the record, otherwise the lane decodes from the entry until it falls into step with round 0's record.  mode 2:
// lanes without an entry get empty counts (nothing is decoded any more: every lane with an entry holds results from exactly there).
// Nothing is written to the coefficient buffer here.
// ---------------------------------------------------------------------------------------------------------------------------
struct SegOut2 {
    uint32_t v[2];
};
// What round 0 saw from a segment's boundary on.  A later decode of the segment from its true entry falls into step with round 0's
// after an MCU or two; from the MCU start where they meet, everything round 0 found is the true decode's too.  So round 0 leaves the
// first SEG_MARKS MCU starts it saw inside the segment with the DC sums at each of them: a validation round whose decode reaches one of
// them stops there and takes the rest -- MCU count, DC sums, exit -- from this record; an entry that IS one of them needs no decode at all.
constexpr int SEG_MARKS = 16;
struct SegTable {
    uint32_t n;                   // MCU starts recorded (pos[0] = 0: the boundary itself, where round 0 assumed one)
    uint32_t count;               // MCUs round 0 saw begin in [boundary, out)
    int32_t dc[3];                // sum of their DC differences per component of the scan
    uint32_t out;                 // round 0's exit (SEG_NONE: it broke off)
    uint32_t pos[SEG_MARKS];      // bit offsets from the boundary, ascending; the k-th is where round 0's k-th MCU began
    int32_t sum[SEG_MARKS][3];    // the DC sums when it began
    uint32_t pad[2];
};
static_assert(sizeof(SegTable) == 288, "segment tables are packed");
// LDS_TABLES as in the walk: the table probe is on the critical path of every symbol (from global memory it is a cache round trip per symbol:
// 68 % of this kernel's wave-cycles were spent waiting, profiles/r03_pmc_jpeg_walk_sync.txt).
constexpr int SYNC_BLOCK = 256;
template <int LDS_TABLES, int MODE>
__global__ void __launch_bounds__(SYNC_BLOCK) jpeg_sync_kernel(const uint8_t *__restrict__ streams, const HImage *__restrict__ imgs, const SegFile *__restrict__ files,
                                                       const uint32_t *__restrict__ seg_file, SegState *__restrict__ segs, SegOut2 *__restrict__ outs, uint32_t n_segs,
                                                       uint32_t seg_bytes, SegTable *__restrict__ tables, int round, const rphj::DeviceLut *__restrict__ g_luts,
                                                       uint32_t n_luts)
{
    constexpr int mode = MODE;  // (a kernel per mode: round 0 carries nothing of the validation rounds' comparisons, and the other way round)
    __shared__ __attribute__((aligned(16))) rphj::DeviceLut s_luts[LDS_TABLES > 0 ? LDS_TABLES : 1];
    const rphj::DeviceLut *luts = LDS_TABLES > 0 ? s_luts : g_luts;
    const uint32_t u = blockIdx.x * SYNC_BLOCK + threadIdx.x;
    // what this lane has to do is decided first: in the validation rounds and in the count pass most lanes have nothing to decode, and a
    // workgroup none of whose lanes decodes leaves without copying the tables
    SegFile F = files[seg_file[u < n_segs ? u : n_segs - 1]];
    const uint32_t t = (u < n_segs ? u : n_segs - 1) - F.first_seg;
    const HImage *im = imgs + F.image;
    const HScan *S = &im->scan[0];
    const uint32_t seg_bits = seg_bytes * 8, lo = t * seg_bits, hi = lo + seg_bits, end_bits = S->len * 8;
    SegTable *const tab = tables + (u < n_segs ? u : n_segs - 1);
    SegState st{};
    const int rd = (round + 1) & 1, wr = round & 1;  // validation round r reads the outs of round r - 1 (round 0 wrote slot 0)
    uint32_t start = 0, tab_n = 0;
    uint32_t marks[SEG_MARKS];  // (validation decodes: round 0's MCU starts, as positions of the scan)
    auto prepare = [&]() -> bool {
        if (u >= n_segs) return false;
        st = segs[u];
        if (mode == 0) {
            start = lo;
            st.entry = t == 0 ? 0 : SEG_NONE;
        } else if (mode == 1) {
            const uint32_t own = outs[u].v[rd];
            const uint32_t e = t == 0 ? 0 : outs[u - 1].v[rd];
            outs[u].v[wr] = own;  // unless this round finds another below
            if (e == SEG_NONE) return false;  // the predecessor has nothing to say yet
            st.entry = e;
            if (e >= hi || e >= end_bits) {  // no MCU begins in this segment (or, in a damaged stream, the predecessor ran past the end of the scan: nothing is decoded from there): pass the position on
                st.from = e;
                st.count = 0;
                st.dc[0] = st.dc[1] = st.dc[2] = 0;
                st.out_check = e;
                outs[u].v[wr] = e;
                segs[u] = st;
                return false;
            }
            if (e == st.from) {  // the last decode (or record) began there: its results stand
                segs[u] = st;
                return false;
            }
            // is the entry one of the MCU starts round 0 recorded?  Then round 0's decode went through it: everything from there on is in the record
            const uint32_t n = tab->out == SEG_NONE ? 0u : tab->n;
            tab_n = n < (uint32_t)SEG_MARKS ? n : (uint32_t)SEG_MARKS;
#pragma unroll
            for (int k = 0; k < SEG_MARKS; k++) marks[k] = (uint32_t)k < tab_n ? lo + tab->pos[k] : SEG_NONE;
            uint32_t hit = SEG_NONE;
#pragma unroll
            for (int k = 0; k < SEG_MARKS; k++) hit = marks[k] == e ? (uint32_t)k : hit;
            if (hit != SEG_NONE) {
                st.from = e;
                st.count = tab->count - hit;
                st.dc[0] = tab->dc[0] - tab->sum[hit][0];
                st.dc[1] = tab->dc[1] - tab->sum[hit][1];
                st.dc[2] = tab->dc[2] - tab->sum[hit][2];
                st.out_check = tab->out;
                outs[u].v[wr] = tab->out;
                segs[u] = st;
                return false;
            }
            start = e;
        } else {
            if (st.entry == SEG_NONE || st.entry >= hi || st.entry >= end_bits) {
                st.count = 0;
                st.dc[0] = st.dc[1] = st.dc[2] = 0;
                st.out_check = st.entry;
                segs[u] = st;
                return false;
            }
            // (every lane that was given an entry in a validation round holds the results of a decode, or of a record, from exactly there)
            if (st.from == st.entry) return false;
            start = st.entry;
        }
        return true;
    };
    const bool active = prepare();
    if (!__syncthreads_or(active)) return;
    if (LDS_TABLES > 0) {  // four waves share one copy of the chunk's tables
        const uint4 *src = reinterpret_cast<const uint4 *>(g_luts);
        uint4 *dst = reinterpret_cast<uint4 *>(s_luts);
        const uint32_t words = n_luts * (uint32_t)(sizeof(rphj::DeviceLut) / 16);
        for (uint32_t w = threadIdx.x; w < words; w += SYNC_BLOCK) dst[w] = src[w];
        __syncthreads();
    }
    if (!active) return;
    // ---- decode from bit `start` of the scan, MCU after MCU, until one begins at or behind `hi`
    const uint32_t ns = S->ns;
    uint32_t H0, H1, H2, V0, V1, V2;
    const rphj::DeviceLut *D0, *D1, *D2, *A0, *A1, *A2;
    {
        const HComp *c0 = &im->comp[S->ci[0]], *c1 = &im->comp[S->ci[ns > 1 ? 1 : 0]], *c2 = &im->comp[S->ci[ns > 2 ? 2 : 0]];
        H0 = ns == 1 ? 1 : c0->H, V0 = ns == 1 ? 1 : c0->V, H1 = c1->H, V1 = c1->V, H2 = c2->H, V2 = c2->V;
        D0 = luts + S->dc[0], A0 = luts + S->ac[0];
        D1 = luts + S->dc[ns > 1 ? 1 : 0], A1 = luts + S->ac[ns > 1 ? 1 : 0];
        D2 = luts + S->dc[ns > 2 ? 2 : 0], A2 = luts + S->ac[ns > 2 ? 2 : 0];
    }
    const uint32_t nblk0 = H0 * V0, nblk1 = ns > 1 ? H1 * V1 : 0, nblk2 = ns > 2 ? H2 * V2 : 0;
    const uint8_t *sp = streams + im->stream_base + S->off + (start >> 3);
    const uint32_t slen = S->len - ((start >> 3) < S->len ? (start >> 3) : S->len);
    const uint32_t lead = (uint32_t)((uintptr_t)sp & 3);
    const uint8_t *sbase = sp - lead;
    const uint32_t limit = ((slen + lead + 3) & ~3u) + 8;
    auto load8 = [&](uint32_t at) -> W2 { return *reinterpret_cast<const W2 *>(sbase + (at < limit ? at : limit)); };
    auto be64 = [](W2 v) -> uint64_t { return ((uint64_t)__builtin_bswap32(v.x) << 32) | __builtin_bswap32(v.y); };
    uint64_t acc = (uint64_t)(__builtin_bswap32(*reinterpret_cast<const uint32_t *>(sbase)) << (8 * lead)) << 32;
    int nb = 32 - 8 * (int)lead;
    acc <<= (start & 7);
    nb -= (int)(start & 7);
    uint64_t q0 = be64(load8(4));
    W2 q1 = load8(12);
    uint32_t q0n = 64, woff = 20;
    uint32_t pos = start;       // bit position of the next unread bit
    uint32_t i = 0, b = 0, k = 0;  // component of the MCU, block of the component, next coefficient index
    uint32_t nblk = nblk0;
    const rphj::DeviceLut *DCc = D0, *ACc = A0;
    bool is_dc = true;
    uint32_t count = 0, out = SEG_NONE, n_marks = 0;
    int dc0 = 0, dc1 = 0, dc2 = 0;
    if (mode == 0) {  // the boundary itself: where this decode assumes an MCU
        tab->pos[0] = 0;
        tab->sum[0][0] = tab->sum[0][1] = tab->sum[0][2] = 0;
        n_marks = 1;
    }
    const uint32_t max_it = seg_bits + 70000;  // every symbol takes at least one bit; an MCU is at most 10 blocks of 64 symbols of <= 32 bits
    for (uint32_t it = 0; it < max_it; it++) {
        if (nb < 32) {
            acc |= (q0 >> 32) << (32 - nb);
            nb += 32;
            q0 <<= 32;
            q0n -= 32;
            if (q0n == 0) {
                q0 = be64(q1);
                q0n = 64;
                q1 = load8(woff);
                woff += 8;
            }
        }
        const rphj::DeviceLut *L = is_dc ? DCc : ACc;
        const uint32_t e = L->look[(uint32_t)(acc >> 54)];
        uint32_t len = e >> 8, sym = e & 255;
        if (e == 0) {
            const int32_t win = (int32_t)(acc >> 48);
            uint32_t l = 11;
            while (l <= 16 && win >= L->maxcode[l]) l++;
            if (l > 16) break;  // not a code: this decode was not synchronised (out stays SEG_NONE)
            len = l;
            sym = L->sym[(uint32_t)((win >> (16 - l)) + L->delta[l]) & 255];
        }
        acc <<= len;
        nb -= (int)len;
        const uint32_t sbits = is_dc ? sym : (sym & 15), r = is_dc ? 0 : (sym >> 4);
        if (sbits > 15) break;
        int val = 0;
        if (sbits) {
            const uint32_t raw = (uint32_t)(acc >> (64 - sbits));
            acc <<= sbits;
            nb -= (int)sbits;
            val = raw < (1u << (sbits - 1)) ? (int)raw - (int)((1u << sbits) - 1) : (int)raw;
        }
        pos += len + sbits;
        if (is_dc) {
            dc0 += i == 0 ? val : 0;
            dc1 += i == 1 ? val : 0;
            dc2 += i == 2 ? val : 0;
            k = 1;
            is_dc = false;
        } else if (sbits == 0) {
            k = r == 15 ? k + 16 : 64;
        } else {
            k += r + 1;
        }
        if (mode == 2 && k > 64) break;  // the count pass decodes from verified positions: a run past coefficient 63 there is a damaged file (out stays SEG_NONE, the chain fails, and the whole-file walk reports it like the host decoder); speculative decodes of the earlier rounds start anywhere and may see this before they fall into step
        if (k >= 64) {  // next block
            is_dc = true;
            k = 0;
            if (++b == nblk) {
                b = 0;
                if (++i == ns) {  // the MCU is complete: the next one begins at `pos`
                    i = 0;
                    count++;
                    if (pos >= hi || pos + 8 > end_bits) {  // (less than a byte left: the padding behind the last MCU)
                        out = pos;
                        break;
                    }
                    if (mode == 0) {
                        if (n_marks < (uint32_t)SEG_MARKS) {  // (pos >= lo: the decode began at lo)
                            tab->pos[n_marks] = pos - lo;
                            tab->sum[n_marks][0] = dc0, tab->sum[n_marks][1] = dc1, tab->sum[n_marks][2] = dc2;
                            n_marks++;
                        }
                    } else if (mode == 1) {  // in step with round 0 from here on?  Then the rest is in its record
                        uint32_t hit = SEG_NONE;
#pragma unroll
                        for (int q = 0; q < SEG_MARKS; q++) hit = marks[q] == pos ? (uint32_t)q : hit;
                        if (hit != SEG_NONE) {
                            count += tab->count - hit;
                            dc0 += tab->dc[0] - tab->sum[hit][0];
                            dc1 += tab->dc[1] - tab->sum[hit][1];
                            dc2 += tab->dc[2] - tab->sum[hit][2];
                            out = tab->out;
                            break;
                        }
                    }
                }
                nblk = sel3(i, nblk0, nblk1, nblk2);
                DCc = sel3(i, D0, D1, D2);
                ACc = sel3(i, A0, A1, A2);
            }
        }
    }
    // every decode leaves its counts beside the position it started from: (from, count, dc, out_check) describe one and the same decode
    st.from = start;
    st.count = out == SEG_NONE ? 0 : count;
    st.dc[0] = dc0, st.dc[1] = dc1, st.dc[2] = dc2;
    st.out_check = out;
    if (mode != 2) outs[u].v[mode == 0 ? 0 : wr] = out;
    if (mode == 0) {
        tab->n = n_marks;
        tab->count = count;
        tab->dc[0] = dc0, tab->dc[1] = dc1, tab->dc[2] = dc2;
        tab->out = out;
    }
    segs[u] = st;
}

// one lane per segmented file: which file each segment belongs to
__global__ void __launch_bounds__(64) jpeg_seg_map_kernel(const SegFile *__restrict__ files, uint32_t n_files, uint32_t *__restrict__ seg_file)
{
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= n_files) return;
    const SegFile F = files[f];
    for (uint32_t t = 0; t < F.n_segs; t++) seg_file[F.first_seg + t] = f;
}

// one lane per segmented file: verify the chain, turn the segments into walk items
__global__ void __launch_bounds__(64) jpeg_seg_items_kernel(const SegFile *__restrict__ files, uint32_t n_files, const SegState *__restrict__ segs, const SegOut2 *__restrict__ outs,
                                                            int final_slot, HItem *__restrict__ items)
{
    const uint32_t f = blockIdx.x * 64 + threadIdx.x;
    if (f >= n_files) return;
    const SegFile F = files[f];
    bool ok = true;
    uint32_t running = 0, prev_out = 0;
    int dc0 = 0, dc1 = 0, dc2 = 0;
    for (uint32_t t = 0; t < F.n_segs; t++) {
        const SegState st = segs[F.first_seg + t];
        const uint32_t out = outs[F.first_seg + t].v[final_slot];
        const uint32_t seg_first = running;
        uint32_t cnt = 0;
        if (st.entry != (t == 0 ? 0u : prev_out) || st.entry == SEG_NONE || out == SEG_NONE || st.out_check != out) ok = false;
        const bool behind = st.entry != SEG_NONE && st.entry >= F.scan_bits;  // (a damaged stream: a decode ran past the end of the scan)
        if (behind && st.count != 0) ok = false;                              // no item may start behind the scan's last byte
        if (ok) {
            cnt = st.count < F.total_mcus - running ? st.count : F.total_mcus - running;  // (behind the last MCU a decode sees padding)
            running += cnt;
        }
        HItem it;
        it.image = F.image, it.scan = 0, it.mcu_first = seg_first, it.mcu_count = cnt;
        it.stream_off = (st.entry == SEG_NONE || behind) ? 0 : st.entry >> 3;
        it.bit_skip = (st.entry == SEG_NONE || behind) ? 0 : st.entry & 7;
        it.dc[0] = dc0, it.dc[1] = dc1, it.dc[2] = dc2;
        items[F.first_item + t] = it;
        dc0 += st.dc[0], dc1 += st.dc[1], dc2 += st.dc[2];
        prev_out = out;
    }
    if (!ok || running != F.total_mcus) {  // the chain did not settle (or the file is damaged): one lane walks the whole file, as without segments
        for (uint32_t t = 0; t < F.n_segs; t++) items[F.first_item + t].mcu_count = 0;
        HItem it;
        it.image = F.image, it.scan = HITEM_ALL_SCANS, it.mcu_first = 0, it.mcu_count = 0, it.stream_off = 0, it.bit_skip = 0;
        it.dc[0] = it.dc[1] = it.dc[2] = 0;
        items[F.first_item] = it;
    }
}

}  // namespace

int rph_jpeg_launch_idct(int flavour, uint32_t max_blocks, uint32_t n_planes, hipStream_t stream, const int16_t *d_coef, const uint16_t *d_tables, const JPlane *d_planes,
                         uint8_t *d_samples, const PRef *d_refs, const PCorr *d_corr, const uint8_t *d_dcbits)
{
    const dim3 grid((max_blocks + 255) / 256, n_planes);
    if (flavour == RPH_JPEG_LIBJPEG)
        hipLaunchKernelGGL(jpeg_idct_kernel<RPH_JPEG_LIBJPEG>, grid, dim3(256), 0, stream, d_coef, d_tables, d_planes, d_samples, d_refs, d_corr, d_dcbits);
    else
        hipLaunchKernelGGL(jpeg_idct_kernel<RPH_JPEG_ZUNE>, grid, dim3(256), 0, stream, d_coef, d_tables, d_planes, d_samples, d_refs, d_corr, d_dcbits);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

int rph_jpeg_launch_color(int flavour, uint32_t max_groups, uint32_t n_images, hipStream_t stream, const uint8_t *d_samples, const JImage *d_images, uint8_t *d_pixels)
{
    const dim3 grid((max_groups + 255) / 256, n_images);
    if (flavour == RPH_JPEG_LIBJPEG)
        hipLaunchKernelGGL(jpeg_color_kernel<RPH_JPEG_LIBJPEG>, grid, dim3(256), 0, stream, d_samples, d_images, d_pixels);
    else
        hipLaunchKernelGGL(jpeg_color_kernel<RPH_JPEG_ZUNE>, grid, dim3(256), 0, stream, d_samples, d_images, d_pixels);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

bool rph_jpeg_walk_writes_whole_blocks(uint32_t n_items)
{
    static const int stage_env = getenv("RPH_JPEG_WALK_STAGE") ? atoi(getenv("RPH_JPEG_WALK_STAGE")) : -1;  // experiments: 0 never, 1 always
    return stage_env >= 0 ? stage_env != 0 : n_items >= 131072u;  // (staged: 195 k lanes +6 %, 237 k lanes +9 %; 59 k lanes -5 %)
}

int rph_jpeg_launch_fused(int flavour, uint32_t max_tiles, uint32_t n_images, hipStream_t stream, const int16_t *d_coef, const uint16_t *d_tables, const JPlane *d_planes,
                          const JImage *d_images, uint8_t *d_pixels, const PRef *d_refs, const PCorr *d_corr, const uint8_t *d_dcbits)
{
    if (max_tiles == 0 || n_images == 0) return RPH_OK;
    const dim3 grid(max_tiles, n_images);
    if (flavour == RPH_JPEG_LIBJPEG)
        hipLaunchKernelGGL(jpeg_fused_kernel<RPH_JPEG_LIBJPEG>, grid, dim3(256), 0, stream, d_coef, d_tables, d_planes, d_images, d_pixels, d_refs, d_corr, d_dcbits);
    else
        hipLaunchKernelGGL(jpeg_fused_kernel<RPH_JPEG_ZUNE>, grid, dim3(256), 0, stream, d_coef, d_tables, d_planes, d_images, d_pixels, d_refs, d_corr, d_dcbits);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

int rph_jpeg_launch_walk(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const HItem *d_items, const uint32_t *d_order, uint32_t n_ordered,
                         uint32_t n_items, const rphj::DeviceLut *d_luts, uint32_t n_luts, int16_t *d_coef, uint8_t *d_status)
{
    // Many lanes (segments, restart intervals: the launch is bound by its stores) build their blocks in LDS; a launch of a lane per file
    // is bound by the length of its longest lane and keeps the lighter loop.
    const bool stage = rph_jpeg_walk_writes_whole_blocks(n_items);
    auto launch = [&](auto kernel, uint32_t threads) {
        hipLaunchKernelGGL(kernel, dim3((n_items + threads - 1) / threads), dim3(threads), 0, stream, d_streams, d_images, d_items, d_order, n_ordered, n_items, d_luts, n_luts, d_coef,
                           d_status);
    };
    if (stage) {
        if (n_luts <= 4)
            launch(jpeg_huff_kernel<4, 8>, 512);  // 9.8 + 64 KB of LDS: two workgroups, sixteen waves per CU (75.9 k photos/s; four waves 70.6 k, two 74.0 k)
        else if (n_luts <= (uint32_t)HUFF_LDS_TABLES)
            launch(jpeg_huff_kernel<HUFF_LDS_TABLES, 4>, 256);  // 19.6 + 32 KB: twelve waves per CU (71.3 k with these tables; two waves 68.8 k, six 66.4 k, eight 66.9 k)
        else
            launch(jpeg_huff_kernel<0, 8>, 512);
    } else if (n_luts <= (uint32_t)HUFF_LDS_TABLES) {
        launch(jpeg_huff_kernel<HUFF_LDS_TABLES, 0>, 64);
    } else {
        launch(jpeg_huff_kernel<0, 0>, 64);
    }
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

int rph_jpeg_launch_prog(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const PScan *d_pscans, uint32_t n_pscans, const uint32_t *d_items,
                         uint32_t n_items, const uint32_t *d_waits, const rphj::DeviceLut *d_luts, uint32_t n_luts, int16_t *d_coef, unsigned long long *d_masks,
                         uint32_t *d_progress, PCorr *d_corr, size_t n_corr, uint8_t *d_dcbits, size_t n_dcbits, uint8_t *d_status)
{
    if (n_items == 0) return RPH_OK;
    (void)n_pscans;  // (d_progress: one word per scan, zeroed by the caller with the masks)
    if (n_corr) RPH_HIP_CHECK(hipMemsetAsync(d_corr, 0, n_corr * sizeof(PCorr), stream));
    if (n_dcbits) RPH_HIP_CHECK(hipMemsetAsync(d_dcbits, 0, n_dcbits, stream));
    const dim3 grid((n_items + 63) / 64);
    // (the DC tables stay in memory whatever their number: 40 KB of LDS per wave are four waves per CU, and the launch needs the slots --
    // the scans of a file that follow one another must be resident together)
    hipLaunchKernelGGL(jpeg_prog_kernel<0>, grid, dim3(64), 0, stream, d_streams, d_images, d_pscans, d_items, n_items, d_waits, d_luts, n_luts, d_coef, d_masks, d_progress, d_corr,
                       d_dcbits, d_status);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}

// (RPH_JPEG_TRACE=1) what round 0 left: segments whose decode broke off, and how many MCU starts the others recorded
void rph_jpeg_debug_segment_stats(hipStream_t stream, const void *d_work, uint32_t n_segs)
{
    std::vector<SegTable> t(n_segs);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(t.data(), d_work, (size_t)n_segs * sizeof(SegTable), hipMemcpyDeviceToHost) != hipSuccess) return;
    size_t broke = 0, none = 0, few = 0, full = 0;
    for (const SegTable &x : t) {
        if (x.out == SEG_NONE) broke++;
        else if (x.n <= 1) none++;
        else if (x.n < (uint32_t)SEG_MARKS) few++;
        else full++;
    }
    fprintf(stderr, "[rph_jpeg] round 0 of %u segments: %zu broke off, %zu recorded only the boundary, %zu recorded 2..%d MCU starts, %zu recorded %d\n", n_segs, broke, none, few,
            SEG_MARKS - 1, full, SEG_MARKS);
}

size_t rph_jpeg_segment_work_bytes(uint32_t n_segs) { return (size_t)n_segs * (sizeof(SegTable) + sizeof(SegOut2) + 4) + 64; }

int rph_jpeg_launch_segments(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const SegFile *d_files, uint32_t n_files, SegState *d_segs,
                             uint32_t n_segs, uint32_t seg_bytes, void *d_work, int rounds, const rphj::DeviceLut *d_luts, uint32_t n_luts, HItem *d_items)
{
    if (n_files == 0 || n_segs == 0) return RPH_OK;
    // d_work: n_segs records of round 0 (SegTable), then n_segs double-buffered `out` positions, then the segment -> file map
    SegTable *d_tables = reinterpret_cast<SegTable *>(d_work);
    SegOut2 *d_outs = reinterpret_cast<SegOut2 *>(d_tables + n_segs);
    uint32_t *d_seg_file = reinterpret_cast<uint32_t *>(d_outs + n_segs);
    hipLaunchKernelGGL(jpeg_seg_map_kernel, dim3((n_files + 63) / 64), dim3(64), 0, stream, d_files, n_files, d_seg_file);
    RPH_HIP_CHECK(hipMemsetAsync(d_tables, 0, (size_t)n_segs * sizeof(SegTable), stream));  // (n = 0: a lane that never ran left no record)
    RPH_HIP_CHECK(hipMemsetAsync(d_outs, 0xFF, (size_t)n_segs * sizeof(SegOut2), stream));
    RPH_HIP_CHECK(hipMemsetAsync(d_segs, 0xFF, (size_t)n_segs * sizeof(SegState), stream));
    const dim3 grid((n_segs + SYNC_BLOCK - 1) / SYNC_BLOCK);
    auto sync = [&](int mode, int round) {
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, grid, dim3(SYNC_BLOCK), 0, stream, d_streams, d_images, d_files, d_seg_file, d_segs, d_outs, n_segs, seg_bytes, d_tables, round, d_luts, n_luts);
        };
        if (n_luts <= (uint32_t)HUFF_LDS_TABLES)
            mode == 0 ? launch(jpeg_sync_kernel<HUFF_LDS_TABLES, 0>) : (mode == 1 ? launch(jpeg_sync_kernel<HUFF_LDS_TABLES, 1>) : launch(jpeg_sync_kernel<HUFF_LDS_TABLES, 2>));
        else
            mode == 0 ? launch(jpeg_sync_kernel<0, 0>) : (mode == 1 ? launch(jpeg_sync_kernel<0, 1>) : launch(jpeg_sync_kernel<0, 2>));
    };
    sync(0, 0);
    for (int r = 1; r <= rounds; r++) sync(1, r);
    sync(2, 0);
    hipLaunchKernelGGL(jpeg_seg_items_kernel, dim3((n_files + 63) / 64), dim3(64), 0, stream, d_files, n_files, d_segs, d_outs, rounds & 1, d_items);
    RPH_HIP_CHECK(hipGetLastError());
    return RPH_OK;
}
