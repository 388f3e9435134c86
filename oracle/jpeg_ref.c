// oracle/jpeg_ref.c -- CPU restatement of the JPEG decode that feeds the hasher.  TEST INFRASTRUCTURE ONLY:
// loaded by tests/ and bench.py's cpu_baseline leg, never by rupphash_amd/.
//
// What the reference does (/root/reference/src/scanner.rs:461-551, load_image_fast): "jpg" | "jpeg" bytes go to zune-jpeg 0.5.15
// (Cargo.lock:8870; tier 2 jpeg-decoder 0.3.2 only when zune fails) and come back as Luma8 (1 component), Rgb8 (3) or Rgba8 (4).
// Neither crate's source is in the reference tree, so this file restates the PUBLISHED algorithms:
//   * ITU-T T.81: markers, Huffman entropy coding (sequential SOF0/SOF1 and progressive SOF2, restart intervals, interleaved and
//     non-interleaved scans), dequantisation.  This part has one right answer: the quantised coefficients.
//   * the sample reconstruction, which is decoder specific, in two flavours:
//       RPH_REF_JPEG_LIBJPEG (1)  libjpeg-turbo's defaults: jidctint.c islow IDCT (CONST_BITS 13, PASS1_BITS 2), jdsample.c fancy
//                                 upsampling (h2v1, h2v2, h1v2), jdcolor.c 16-bit fixed-point YCbCr->RGB.  PINNED: Pillow in this image is
//                                 built on libjpeg-turbo, and tests/test_oracle_jpeg.py requires byte equality with Pillow's decode of
//                                 the reference's own JPEG files and of generated ones.
//       RPH_REF_JPEG_ZUNE (0)     zune-jpeg as recalled: the stb_image integer IDCT (12-bit constants, >>10 then >>17), upsampling as
//                                 (3 a + b + 2) >> 2 per direction (vertical, then horizontal for 2x2), YCbCr->RGB as
//                                 y + (45 cr >> 5), y - ((11 cb + 23 cr) >> 5), y + (113 cb >> 6).  PARITY UNPINNED: written from
//                                 recollection of the crate, nothing in the reference tree can confirm it.
// Both flavours share everything up to the dequantised coefficients, so the pin on flavour 1 also pins the entropy decoder, the
// scan bookkeeping and the block geometry that flavour 0 uses.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RPH_REF_JPEG_ZUNE 0
#define RPH_REF_JPEG_LIBJPEG 1

enum { JR_OK = 0, JR_ERR_FORMAT = -1, JR_ERR_UNSUPPORTED = -5, JR_ERR_OOM = -4 };

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    int present;
    uint8_t counts[17];
    uint8_t symbols[256];
    int32_t mincode[17], maxcode[18], valptr[17];
} Huff;

typedef struct {
    int id, H, V, tq;
    int blocks_w, blocks_h;  // MCU-padded block grid (allocation)
    int real_bw, real_bh;    // ceil(component samples / 8): what a non-interleaved scan covers
    int samp_w, samp_h;      // component samples: ceil(w * H / Hmax), ceil(h * V / Vmax)
    size_t first_block;
    int dc_tbl, ac_tbl;
    int32_t pred;
} Comp;

typedef struct {
    const uint8_t *data;
    size_t len, pos;
    int w, h, ncomp, progressive, have_sof;
    int Hmax, Vmax, mcus_x, mcus_y;
    Comp comp[4];
    uint16_t qt[4][64];  // natural order
    int qt_present[4];
    Huff dc[4], ac[4];
    int restart_interval;
    int adobe_transform;  // -1 = no Adobe marker
    int16_t *coef;        // [total_blocks][64], natural order
    size_t total_blocks;
} Dec;

// ------------------------------------------------------------------------------------------------------------------------
// bit reader over one entropy-coded segment (T.81 F.2.2.5: 0xFF00 is a stuffed 0xFF, any other 0xFFxx ends the segment)
// ------------------------------------------------------------------------------------------------------------------------
typedef struct {
    const uint8_t *p, *end;
    uint32_t acc;
    int nbits;
    int hit, marker;  // hit: a marker or the end of the data was reached (zeros are fed from there on)
} BR;

static int br_byte(BR *b)
{
    if (b->hit) return 0;
    if (b->p >= b->end) {
        b->hit = 1;
        return 0;
    }
    int c = *b->p++;
    if (c != 0xFF) return c;
    int d;
    do {
        if (b->p >= b->end) {
            b->hit = 1;
            return 0;
        }
        d = *b->p++;
    } while (d == 0xFF);
    if (d == 0) return 0xFF;
    b->marker = d;
    b->hit = 1;
    return 0;
}
static int br_bit(BR *b)
{
    if (b->nbits == 0) {
        b->acc = (uint32_t)br_byte(b);
        b->nbits = 8;
    }
    b->nbits--;
    return (int)((b->acc >> b->nbits) & 1u);
}
static int br_bits(BR *b, int n)
{
    int v = 0;
    while (n-- > 0) v = (v << 1) | br_bit(b);
    return v;
}
// T.81 F.2.2.1 EXTEND
static int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

static int huff_build(Huff *h)
{
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += h->counts[l];
        k += h->counts[l];
        h->maxcode[l] = h->counts[l] ? code - 1 : -1;
        if (code > (1 << l)) return JR_ERR_FORMAT;
        code <<= 1;
    }
    h->maxcode[17] = 0x7FFFFFFF;
    return JR_OK;
}
// T.81 F.2.2.3 DECODE, bit by bit
static int huff_decode(BR *b, const Huff *h)
{
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br_bit(b);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->symbols[h->valptr[l] + code - h->mincode[l]];
    }
    return -1;
}

// ------------------------------------------------------------------------------------------------------------------------
// markers
// ------------------------------------------------------------------------------------------------------------------------
static int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

static int parse_dqt(Dec *d, const uint8_t *p, int n)
{
    while (n > 0) {
        const int pq = p[0] >> 4, tq = p[0] & 15;
        if (tq > 3 || pq > 1) return JR_ERR_FORMAT;
        const int need = 1 + 64 * (pq + 1);
        if (n < need) return JR_ERR_FORMAT;
        for (int k = 0; k < 64; k++) d->qt[tq][ZIGZAG[k]] = (uint16_t)(pq ? rd16(p + 1 + 2 * k) : p[1 + k]);
        d->qt_present[tq] = 1;
        p += need;
        n -= need;
    }
    return JR_OK;
}
static int parse_dht(Dec *d, const uint8_t *p, int n)
{
    while (n > 0) {
        if (n < 17) return JR_ERR_FORMAT;
        const int tc = p[0] >> 4, th = p[0] & 15;
        if (tc > 1 || th > 3) return JR_ERR_FORMAT;
        Huff *h = tc ? &d->ac[th] : &d->dc[th];
        int total = 0;
        h->counts[0] = 0;
        for (int l = 1; l <= 16; l++) {
            h->counts[l] = p[l];
            total += p[l];
        }
        if (total > 256 || n < 17 + total) return JR_ERR_FORMAT;
        memcpy(h->symbols, p + 17, (size_t)total);
        h->present = 1;
        const int rc = huff_build(h);
        if (rc) return rc;
        p += 17 + total;
        n -= 17 + total;
    }
    return JR_OK;
}
static int parse_sof(Dec *d, const uint8_t *p, int n, int progressive)
{
    if (d->have_sof || n < 6) return JR_ERR_FORMAT;
    if (p[0] != 8) return JR_ERR_UNSUPPORTED;  // 12-bit samples
    d->h = rd16(p + 1);
    d->w = rd16(p + 3);
    d->ncomp = p[5];
    if (d->w == 0 || d->h == 0) return JR_ERR_UNSUPPORTED;  // DNL-defined height
    if (d->ncomp != 1 && d->ncomp != 3) return JR_ERR_UNSUPPORTED;  // CMYK / YCCK: the caller falls back, like a failed tier
    if (n < 6 + 3 * d->ncomp) return JR_ERR_FORMAT;
    d->progressive = progressive;
    d->Hmax = d->Vmax = 1;
    for (int c = 0; c < d->ncomp; c++) {
        Comp *k = &d->comp[c];
        k->id = p[6 + 3 * c];
        k->H = p[7 + 3 * c] >> 4;
        k->V = p[7 + 3 * c] & 15;
        k->tq = p[8 + 3 * c];
        if (k->H < 1 || k->H > 4 || k->V < 1 || k->V > 4 || k->tq > 3) return JR_ERR_FORMAT;
        if (d->ncomp == 1) k->H = k->V = 1;  // one component: the MCU is one block whatever the factors say (T.81 A.2.2)
        if (k->H > d->Hmax) d->Hmax = k->H;
        if (k->V > d->Vmax) d->Vmax = k->V;
    }
    if (d->ncomp == 3) {
        // supported geometries: luma carries the maximum factors (1 or 2 each way), both chroma planes are 1x1 or equal to luma
        const Comp *y = &d->comp[0];
        if (y->H != d->Hmax || y->V != d->Vmax || y->H > 2 || y->V > 2) return JR_ERR_UNSUPPORTED;
        for (int c = 1; c < 3; c++) {
            const Comp *k = &d->comp[c];
            if (!((k->H == 1 && k->V == 1) || (k->H == y->H && k->V == y->V))) return JR_ERR_UNSUPPORTED;
            if (k->H != d->comp[1].H || k->V != d->comp[1].V) return JR_ERR_UNSUPPORTED;
        }
        if (d->comp[0].id == 'R' && d->comp[1].id == 'G' && d->comp[2].id == 'B') return JR_ERR_UNSUPPORTED;  // RGB-coded JPEG
        if (y->H * y->V + 2 * d->comp[1].H * d->comp[1].V > 10) return JR_ERR_FORMAT;  // T.81 B.2.3: at most 10 blocks per MCU
    }
    d->mcus_x = (d->w + 8 * d->Hmax - 1) / (8 * d->Hmax);
    d->mcus_y = (d->h + 8 * d->Vmax - 1) / (8 * d->Vmax);
    size_t total = 0;
    for (int c = 0; c < d->ncomp; c++) {
        Comp *k = &d->comp[c];
        k->blocks_w = d->mcus_x * k->H;
        k->blocks_h = d->mcus_y * k->V;
        k->samp_w = (d->w * k->H + d->Hmax - 1) / d->Hmax;
        k->samp_h = (d->h * k->V + d->Vmax - 1) / d->Vmax;
        k->real_bw = (k->samp_w + 7) / 8;
        k->real_bh = (k->samp_h + 7) / 8;
        k->first_block = total;
        total += (size_t)k->blocks_w * (size_t)k->blocks_h;
    }
    d->total_blocks = total;
    d->have_sof = 1;
    return JR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// scans
// ------------------------------------------------------------------------------------------------------------------------
typedef struct {
    int ns, ci[4];
    int ss, se, ah, al;
    int eobrun;
} Scan;

static int16_t *block_at(Dec *d, const Comp *k, int bx, int by) { return d->coef + (k->first_block + (size_t)by * k->blocks_w + bx) * 64; }

// sequential block (T.81 F.2.2)
static int decode_block_seq(Dec *d, BR *b, Comp *k, int16_t *blk)
{
    const Huff *hd = &d->dc[k->dc_tbl], *ha = &d->ac[k->ac_tbl];
    int s = huff_decode(b, hd);
    if (s < 0 || s > 15) return JR_ERR_FORMAT;
    k->pred += extend(br_bits(b, s), s);
    blk[0] = (int16_t)k->pred;
    for (int kk = 1; kk < 64;) {
        const int rs = huff_decode(b, ha);
        if (rs < 0) return JR_ERR_FORMAT;
        const int r = rs >> 4;
        s = rs & 15;
        if (s == 0) {
            if (r != 15) break;
            kk += 16;
            continue;
        }
        kk += r;
        if (kk > 63) return JR_ERR_FORMAT;
        blk[ZIGZAG[kk]] = (int16_t)extend(br_bits(b, s), s);
        kk++;
    }
    return JR_OK;
}
// progressive (T.81 G.1.2)
static int decode_block_dc_first(Dec *d, BR *b, Comp *k, int16_t *blk, const Scan *sc)
{
    const int s = huff_decode(b, &d->dc[k->dc_tbl]);
    if (s < 0 || s > 15) return JR_ERR_FORMAT;
    k->pred += extend(br_bits(b, s), s);
    blk[0] = (int16_t)(k->pred * (1 << sc->al));
    return JR_OK;
}
static int decode_block_dc_refine(BR *b, int16_t *blk, const Scan *sc)
{
    if (br_bit(b)) blk[0] = (int16_t)(blk[0] | (1 << sc->al));
    return JR_OK;
}
static int decode_block_ac_first(Dec *d, BR *b, Comp *k, int16_t *blk, Scan *sc)
{
    if (sc->eobrun > 0) {
        sc->eobrun--;
        return JR_OK;
    }
    const Huff *ha = &d->ac[k->ac_tbl];
    for (int kk = sc->ss; kk <= sc->se;) {
        const int rs = huff_decode(b, ha);
        if (rs < 0) return JR_ERR_FORMAT;
        const int r = rs >> 4, s = rs & 15;
        if (s == 0) {
            if (r == 15) {
                kk += 16;
                continue;
            }
            sc->eobrun = (1 << r) - 1;
            if (r) sc->eobrun += br_bits(b, r);
            break;
        }
        kk += r;
        if (kk > 63) return JR_ERR_FORMAT;
        blk[ZIGZAG[kk]] = (int16_t)(extend(br_bits(b, s), s) * (1 << sc->al));
        kk++;
    }
    return JR_OK;
}
static int decode_block_ac_refine(Dec *d, BR *b, Comp *k, int16_t *blk, Scan *sc)
{
    const int p1 = 1 << sc->al, m1 = -(1 << sc->al);
    const Huff *ha = &d->ac[k->ac_tbl];
    int kk = sc->ss;
    if (sc->eobrun == 0) {
        for (; kk <= sc->se; kk++) {
            const int rs = huff_decode(b, ha);
            if (rs < 0) return JR_ERR_FORMAT;
            int r = rs >> 4;
            const int s = rs & 15;
            int value = 0;
            if (s) {
                if (s != 1) return JR_ERR_FORMAT;
                value = br_bit(b) ? p1 : m1;
            } else if (r != 15) {
                sc->eobrun = 1 << r;
                if (r) sc->eobrun += br_bits(b, r);
                break;
            }
            // skip r zero-history coefficients, refining the nonzero ones passed on the way
            do {
                int16_t *c = &blk[ZIGZAG[kk]];
                if (*c != 0) {
                    if (br_bit(b)) {
                        if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
                    }
                } else {
                    if (--r < 0) break;
                }
                kk++;
            } while (kk <= sc->se);
            if (value && kk <= sc->se) blk[ZIGZAG[kk]] = (int16_t)value;
        }
    }
    if (sc->eobrun > 0) {
        for (; kk <= sc->se; kk++) {
            int16_t *c = &blk[ZIGZAG[kk]];
            if (*c != 0 && br_bit(b)) {
                if ((*c & p1) == 0) *c = (int16_t)(*c >= 0 ? *c + p1 : *c + m1);
            }
        }
        sc->eobrun--;
    }
    return JR_OK;
}

static int decode_one(Dec *d, BR *b, Comp *k, int16_t *blk, Scan *sc)
{
    if (!d->progressive) return decode_block_seq(d, b, k, blk);
    if (sc->ss == 0) return sc->ah == 0 ? decode_block_dc_first(d, b, k, blk, sc) : decode_block_dc_refine(b, blk, sc);
    return sc->ah == 0 ? decode_block_ac_first(d, b, k, blk, sc) : decode_block_ac_refine(d, b, k, blk, sc);
}

// decodes the entropy-coded data of one scan starting at d->pos; leaves d->pos at the marker that ends it
static int decode_scan(Dec *d, Scan *sc)
{
    BR b;
    memset(&b, 0, sizeof b);
    b.p = d->data + d->pos;
    b.end = d->data + d->len;
    for (int i = 0; i < sc->ns; i++) d->comp[sc->ci[i]].pred = 0;
    sc->eobrun = 0;
    int mcus_x, mcus_y;
    if (sc->ns == 1) {  // non-interleaved: the component's own block grid (T.81 A.2.2)
        mcus_x = d->comp[sc->ci[0]].real_bw;
        mcus_y = d->comp[sc->ci[0]].real_bh;
    } else {
        mcus_x = d->mcus_x;
        mcus_y = d->mcus_y;
    }
    int until_restart = d->restart_interval;
    int16_t dummy[64];
    for (int my = 0; my < mcus_y; my++)
        for (int mx = 0; mx < mcus_x; mx++) {
            if (d->restart_interval && until_restart == 0) {
                // RSTn (T.81 E.2.4): byte-align, take the marker, reset the predictions and the end-of-band run
                b.nbits = 0;
                if (!b.hit) {
                    while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] != 0 && b.p[1] != 0xFF)) b.p++;
                    if (b.p + 1 >= b.end) return JR_ERR_FORMAT;
                    b.marker = b.p[1];
                    b.p += 2;
                }
                if (b.marker < 0xD0 || b.marker > 0xD7) return JR_ERR_FORMAT;
                b.hit = 0;
                b.marker = 0;
                for (int i = 0; i < sc->ns; i++) d->comp[sc->ci[i]].pred = 0;
                sc->eobrun = 0;
                until_restart = d->restart_interval;
            }
            if (sc->ns == 1) {
                Comp *k = &d->comp[sc->ci[0]];
                const int rc = decode_one(d, &b, k, block_at(d, k, mx, my), sc);
                if (rc) return rc;
            } else {
                for (int i = 0; i < sc->ns; i++) {
                    Comp *k = &d->comp[sc->ci[i]];
                    for (int v = 0; v < k->V; v++)
                        for (int h = 0; h < k->H; h++) {
                            const int bx = mx * k->H + h, by = my * k->V + v;
                            int16_t *blk = (bx < k->blocks_w && by < k->blocks_h) ? block_at(d, k, bx, by) : dummy;
                            const int rc = decode_one(d, &b, k, blk, sc);
                            if (rc) return rc;
                        }
                }
            }
            until_restart--;
        }
    // position of the marker that ends the scan
    if (b.hit && b.marker) {
        d->pos = (size_t)(b.p - d->data) - 2;
    } else {
        const uint8_t *p = b.p;
        while (p + 1 < b.end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
        d->pos = (size_t)(p - d->data);
    }
    return JR_OK;
}

static int parse_sos(Dec *d, const uint8_t *p, int n, Scan *sc)
{
    if (!d->have_sof || n < 1) return JR_ERR_FORMAT;
    sc->ns = p[0];
    if (sc->ns < 1 || sc->ns > d->ncomp || n < 1 + 2 * sc->ns + 3) return JR_ERR_FORMAT;
    for (int i = 0; i < sc->ns; i++) {
        const int id = p[1 + 2 * i];
        int c;
        for (c = 0; c < d->ncomp; c++)
            if (d->comp[c].id == id) break;
        if (c == d->ncomp) return JR_ERR_FORMAT;
        sc->ci[i] = c;
        d->comp[c].dc_tbl = p[2 + 2 * i] >> 4;
        d->comp[c].ac_tbl = p[2 + 2 * i] & 15;
        if (d->comp[c].dc_tbl > 3 || d->comp[c].ac_tbl > 3) return JR_ERR_FORMAT;
    }
    const uint8_t *q = p + 1 + 2 * sc->ns;
    sc->ss = q[0];
    sc->se = q[1];
    sc->ah = q[2] >> 4;
    sc->al = q[2] & 15;
    if (d->progressive) {
        if (sc->ss > sc->se || sc->se > 63 || sc->al > 13) return JR_ERR_FORMAT;
        if (sc->ss == 0 && sc->se != 0) return JR_ERR_FORMAT;
        if (sc->ss != 0 && sc->ns != 1) return JR_ERR_FORMAT;
    } else {
        sc->ss = 0;
        sc->se = 63;
        sc->ah = sc->al = 0;
    }
    // tables the scan needs
    for (int i = 0; i < sc->ns; i++) {
        const Comp *k = &d->comp[sc->ci[i]];
        const int need_dc = !d->progressive || (sc->ss == 0 && sc->ah == 0);
        const int need_ac = !d->progressive || sc->ss != 0;
        if (need_dc && !d->dc[k->dc_tbl].present) return JR_ERR_FORMAT;
        if (need_ac && !d->ac[k->ac_tbl].present) return JR_ERR_FORMAT;
    }
    return JR_OK;
}

// header_only: stop after the frame header
static int decode_stream(Dec *d, int header_only)
{
    if (d->len < 4 || d->data[0] != 0xFF || d->data[1] != 0xD8) return JR_ERR_FORMAT;
    d->pos = 2;
    d->adobe_transform = -1;
    int seen_scan = 0;
    for (;;) {
        // next marker
        while (d->pos < d->len && d->data[d->pos] != 0xFF) d->pos++;
        while (d->pos < d->len && d->data[d->pos] == 0xFF) d->pos++;
        if (d->pos >= d->len) break;
        const int m = d->data[d->pos++];
        if (m == 0xD9) break;                                 // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0) continue;  // TEM, stray RSTn
        if (d->pos + 2 > d->len) return seen_scan ? JR_OK : JR_ERR_FORMAT;
        const int n = rd16(d->data + d->pos) - 2;
        const uint8_t *p = d->data + d->pos + 2;
        if (n < 0 || d->pos + 2 + (size_t)n > d->len) return seen_scan ? JR_OK : JR_ERR_FORMAT;
        d->pos += 2 + (size_t)n;
        int rc = JR_OK;
        switch (m) {
        case 0xDB: rc = parse_dqt(d, p, n); break;
        case 0xC4: rc = parse_dht(d, p, n); break;
        case 0xC0:
        case 0xC1: rc = parse_sof(d, p, n, 0); break;
        case 0xC2: rc = parse_sof(d, p, n, 1); break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return JR_ERR_UNSUPPORTED;  // lossless, differential, arithmetic coding
        case 0xDD:
            if (n < 2) return JR_ERR_FORMAT;
            d->restart_interval = rd16(p);
            break;
        case 0xEE:
            if (n >= 12 && memcmp(p, "Adobe", 5) == 0) d->adobe_transform = p[11];
            break;
        case 0xDA: {
            if (header_only) return d->have_sof ? JR_OK : JR_ERR_FORMAT;
            Scan sc;
            rc = parse_sos(d, p, n, &sc);
            if (rc) return rc;
            if (!d->coef) {
                d->coef = (int16_t *)calloc(d->total_blocks * 64, sizeof(int16_t));
                if (!d->coef) return JR_ERR_OOM;
            }
            rc = decode_scan(d, &sc);
            seen_scan = 1;
            break;
        }
        default: break;  // APPn, COM, ...
        }
        if (rc) return rc;
        if (header_only && d->have_sof) return JR_OK;
    }
    if (!d->have_sof) return JR_ERR_FORMAT;
    if (header_only) return JR_OK;
    if (!seen_scan) return JR_ERR_FORMAT;
    if (d->ncomp == 3 && d->adobe_transform == 0) return JR_ERR_UNSUPPORTED;  // Adobe RGB (no colour transform)
    for (int c = 0; c < d->ncomp; c++)
        if (!d->qt_present[d->comp[c].tq]) return JR_ERR_FORMAT;
    return JR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// sample reconstruction
// ------------------------------------------------------------------------------------------------------------------------
static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// 32-bit wrapping arithmetic, as both decoders compute (i32 in Rust release builds wraps on mul via wrapping_*; C int in libjpeg is
// never near overflow on valid data): every product and sum below is done in uint32_t and reinterpreted
#define MUL(a, b) ((int32_t)((uint32_t)(a) * (uint32_t)(b)))
#define ADD(a, b) ((int32_t)((uint32_t)(a) + (uint32_t)(b)))
#define SUB(a, b) ((int32_t)((uint32_t)(a) - (uint32_t)(b)))
#define SHL(a, n) ((int32_t)((uint32_t)(a) << (n)))
static int32_t sra(int32_t a, int n) { return a >= 0 ? a >> n : ~((~a) >> n); }  // arithmetic shift, spelled out

// libjpeg-turbo jidctint.c jpeg_idct_islow (the zero-AC shortcuts of both passes give the same numbers as the full butterfly)
static void idct_islow_1d(const int32_t in[8], int32_t out[8], int shift)
{
    int32_t z1, z2, z3, z4, z5, tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13;
    z2 = in[2];
    z3 = in[6];
    z1 = MUL(ADD(z2, z3), 4433);
    tmp2 = ADD(z1, MUL(z3, -15137));
    tmp3 = ADD(z1, MUL(z2, 6270));
    z2 = in[0];
    z3 = in[4];
    tmp0 = SHL(ADD(z2, z3), 13);
    tmp1 = SHL(SUB(z2, z3), 13);
    tmp10 = ADD(tmp0, tmp3);
    tmp13 = SUB(tmp0, tmp3);
    tmp11 = ADD(tmp1, tmp2);
    tmp12 = SUB(tmp1, tmp2);
    tmp0 = in[7];
    tmp1 = in[5];
    tmp2 = in[3];
    tmp3 = in[1];
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
    z3 = MUL(z3, -16069);
    z4 = MUL(z4, -3196);
    z3 = ADD(z3, z5);
    z4 = ADD(z4, z5);
    tmp0 = ADD(tmp0, ADD(z1, z3));
    tmp1 = ADD(tmp1, ADD(z2, z4));
    tmp2 = ADD(tmp2, ADD(z2, z3));
    tmp3 = ADD(tmp3, ADD(z1, z4));
    const int32_t rnd = SHL(1, shift - 1);
    out[0] = sra(ADD(ADD(tmp10, tmp3), rnd), shift);
    out[7] = sra(ADD(SUB(tmp10, tmp3), rnd), shift);
    out[1] = sra(ADD(ADD(tmp11, tmp2), rnd), shift);
    out[6] = sra(ADD(SUB(tmp11, tmp2), rnd), shift);
    out[2] = sra(ADD(ADD(tmp12, tmp1), rnd), shift);
    out[5] = sra(ADD(SUB(tmp12, tmp1), rnd), shift);
    out[3] = sra(ADD(ADD(tmp13, tmp0), rnd), shift);
    out[4] = sra(ADD(SUB(tmp13, tmp0), rnd), shift);
}
// stb_image / zune-jpeg integer IDCT: 12-bit constants; `bias` is added to the even part before the final shift
static void idct_stb_1d(const int32_t s[8], int32_t out[8], int32_t bias, int shift)
{
    int32_t t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3;
    p2 = s[2];
    p3 = s[6];
    p1 = MUL(ADD(p2, p3), 2217);
    t2 = ADD(p1, MUL(p3, -7567));
    t3 = ADD(p1, MUL(p2, 3135));
    p2 = s[0];
    p3 = s[4];
    t0 = SHL(ADD(p2, p3), 12);
    t1 = SHL(SUB(p2, p3), 12);
    x0 = ADD(t0, t3);
    x3 = SUB(t0, t3);
    x1 = ADD(t1, t2);
    x2 = SUB(t1, t2);
    t0 = s[7];
    t1 = s[5];
    t2 = s[3];
    t3 = s[1];
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
    x0 = ADD(x0, bias);
    x1 = ADD(x1, bias);
    x2 = ADD(x2, bias);
    x3 = ADD(x3, bias);
    out[0] = sra(ADD(x0, t3), shift);
    out[7] = sra(SUB(x0, t3), shift);
    out[1] = sra(ADD(x1, t2), shift);
    out[6] = sra(SUB(x1, t2), shift);
    out[2] = sra(ADD(x2, t1), shift);
    out[5] = sra(SUB(x2, t1), shift);
    out[3] = sra(ADD(x3, t0), shift);
    out[4] = sra(SUB(x3, t0), shift);
}

// one block: dequantise (coefficient * table entry, T.81 A.3.4), columns, rows, level shift, clamp
static void idct_block(const int16_t *coef, const uint16_t *qt, int flavour, uint8_t *out, size_t pitch)
{
    int32_t ws[64], col[8], res[8];
    for (int x = 0; x < 8; x++) {
        for (int y = 0; y < 8; y++) col[y] = MUL((int32_t)coef[8 * y + x], (int32_t)qt[8 * y + x]);
        if (flavour == RPH_REF_JPEG_LIBJPEG)
            idct_islow_1d(col, res, 13 - 2);
        else
            idct_stb_1d(col, res, 512, 10);
        for (int y = 0; y < 8; y++) ws[8 * y + x] = res[y];
    }
    for (int y = 0; y < 8; y++) {
        if (flavour == RPH_REF_JPEG_LIBJPEG) {
            idct_islow_1d(ws + 8 * y, res, 13 + 2 + 3);
            // range_limit[(v) & RANGE_MASK] with the table centred on 128: v is taken modulo 1024 into [-512, 511], then clamp(v + 128)
            for (int x = 0; x < 8; x++) out[y * pitch + x] = clamp8((((res[x] + 512) & 1023) - 512) + 128);
        } else {
            idct_stb_1d(ws + 8 * y, res, 65536 + (128 << 17), 17);
            for (int x = 0; x < 8; x++) out[y * pitch + x] = clamp8(res[x]);
        }
    }
}

typedef struct {
    uint8_t *p;
    int pitch, rows;  // allocation: blocks_w * 8 by blocks_h * 8
    int w, h;         // samples used by the upsampler: flavour 1 = real component samples, flavour 0 = the padded plane
} Plane;

static int plane_at(const Plane *pl, int x, int y)
{
    // rows outside [0, h) repeat the edge row (libjpeg: jdmainct.c context rows at the top and bottom of the image)
    if (y < 0) y = 0;
    if (y >= pl->h) y = pl->h - 1;
    return pl->p[(size_t)y * pl->pitch + x];
}

// chroma sample of the full-resolution grid at (x, y); hs, vs = 1 or 2 = upsampling factor per direction
static int upsampled(const Plane *pl, int x, int y, int hs, int vs, int flavour)
{
    if (hs == 1 && vs == 1) return plane_at(pl, x, y);
    const int n = pl->w;
    if (flavour == RPH_REF_JPEG_LIBJPEG) {
        // jdsample.c: fancy (triangle) upsampling needs more than two columns, otherwise samples are replicated
        if (hs == 2 && n <= 2) return plane_at(pl, x >> 1, vs == 2 ? (y >> 1) : y);
        if (hs == 2 && vs == 1) {  // h2v1_fancy_upsample
            const int c = x >> 1, v = plane_at(pl, c, y);
            if ((x & 1) == 0) return c == 0 ? v : (3 * v + plane_at(pl, c - 1, y) + 1) >> 2;
            return c == n - 1 ? v : (3 * v + plane_at(pl, c + 1, y) + 2) >> 2;
        }
        if (hs == 1 && vs == 2) {  // h1v2_fancy_upsample: bias 1 for the upper output row, 2 for the lower
            const int r = y >> 1, lower = y & 1;
            return (3 * plane_at(pl, x, r) + plane_at(pl, x, lower ? r + 1 : r - 1) + (lower ? 2 : 1)) >> 2;
        }
        // h2v2_fancy_upsample: column sums 3 * near row + far row, then 3:1 across columns with rounding 8 (even) / 7 (odd)
        const int r = y >> 1, rr = (y & 1) ? r + 1 : r - 1, c = x >> 1;
        const int cs = 3 * plane_at(pl, c, r) + plane_at(pl, c, rr);
        if ((x & 1) == 0) {
            if (c == 0) return (cs * 4 + 8) >> 4;
            return (3 * cs + (3 * plane_at(pl, c - 1, r) + plane_at(pl, c - 1, rr)) + 8) >> 4;
        }
        if (c == n - 1) return (cs * 4 + 7) >> 4;
        return (3 * cs + (3 * plane_at(pl, c + 1, r) + plane_at(pl, c + 1, rr)) + 7) >> 4;
    }
    // zune-jpeg (recalled): vertical (3 a + b + 2) >> 2 first, then the same horizontally on the result; edge samples are copied
    const int r = vs == 2 ? (y >> 1) : y, c = hs == 2 ? (x >> 1) : x;
    const int rr = vs == 2 ? ((y & 1) ? r + 1 : r - 1) : r;
#define ZV(cc) (vs == 2 ? ((3 * plane_at(pl, (cc), r) + plane_at(pl, (cc), rr) + 2) >> 2) : plane_at(pl, (cc), r))
    const int v = ZV(c);
    if (hs == 1) return v;
    if ((x & 1) == 0) return c == 0 ? v : (3 * v + ZV(c - 1) + 2) >> 2;
    return c == n - 1 ? v : (3 * v + ZV(c + 1) + 2) >> 2;
#undef ZV
}

static void ycc_to_rgb(int y, int cb, int cr, int flavour, uint8_t *rgb)
{
    cb -= 128;
    cr -= 128;
    if (flavour == RPH_REF_JPEG_LIBJPEG) {  // jdcolor.c build_ycc_rgb_table / ycc_rgb_convert, SCALEBITS 16
        rgb[0] = clamp8(y + sra(91881 * cr + 32768, 16));
        rgb[1] = clamp8(y + sra(-22554 * cb + 32768 - 46802 * cr, 16));
        rgb[2] = clamp8(y + sra(116130 * cb + 32768, 16));
    } else {  // zune-jpeg color_convert (recalled): i16 arithmetic, arithmetic shifts
        rgb[0] = clamp8(y + sra(45 * cr, 5));
        rgb[1] = clamp8(y - sra(11 * cb + 23 * cr, 5));
        rgb[2] = clamp8(y + sra(113 * cb, 6));
    }
}

static void dec_free(Dec *d) { free(d->coef); }

int rph_ref_jpeg_info(const uint8_t *data, size_t len, uint32_t *w, uint32_t *h, uint32_t *channels)
{
    Dec d;
    memset(&d, 0, sizeof d);
    d.data = data;
    d.len = len;
    const int rc = decode_stream(&d, 1);
    if (rc) return rc;
    *w = (uint32_t)d.w;
    *h = (uint32_t)d.h;
    *channels = (uint32_t)d.ncomp;
    return JR_OK;
}

// geometry[c] = {blocks_w, blocks_h, H, V, tq, samp_w, samp_h, first_block} per component; qt = 4 x 64 natural order;
// coef (nullable) receives total_blocks * 64 quantised coefficients in natural order, component-major, raster over the padded grid
int rph_ref_jpeg_coefficients(const uint8_t *data, size_t len, uint32_t *geometry /* 3 x 8 */, uint16_t *qt, int16_t *coef, size_t cap_blocks,
                              uint64_t *total_blocks)
{
    Dec d;
    memset(&d, 0, sizeof d);
    d.data = data;
    d.len = len;
    int rc = decode_stream(&d, 0);
    if (rc == JR_OK) {
        for (int c = 0; c < d.ncomp; c++) {
            const Comp *k = &d.comp[c];
            uint32_t *g = geometry + 8 * c;
            g[0] = (uint32_t)k->blocks_w, g[1] = (uint32_t)k->blocks_h, g[2] = (uint32_t)k->H, g[3] = (uint32_t)k->V, g[4] = (uint32_t)k->tq;
            g[5] = (uint32_t)k->samp_w, g[6] = (uint32_t)k->samp_h, g[7] = (uint32_t)k->first_block;
        }
        memcpy(qt, d.qt, sizeof d.qt);
        *total_blocks = d.total_blocks;
        if (coef) {
            if (cap_blocks < d.total_blocks)
                rc = JR_ERR_OOM;
            else
                memcpy(coef, d.coef, d.total_blocks * 64 * sizeof(int16_t));
        }
    }
    dec_free(&d);
    return rc;
}

// out: w * h * channels bytes, packed (Luma8 or Rgb8, what load_image_fast wraps into a DynamicImage)
int rph_ref_jpeg_decode(const uint8_t *data, size_t len, int flavour, uint8_t *out)
{
    Dec d;
    memset(&d, 0, sizeof d);
    d.data = data;
    d.len = len;
    int rc = decode_stream(&d, 0);
    if (rc) {
        dec_free(&d);
        return rc;
    }
    Plane pl[3];
    memset(pl, 0, sizeof pl);
    for (int c = 0; c < d.ncomp && rc == JR_OK; c++) {
        const Comp *k = &d.comp[c];
        pl[c].pitch = k->blocks_w * 8;
        pl[c].rows = k->blocks_h * 8;
        pl[c].w = flavour == RPH_REF_JPEG_LIBJPEG ? k->samp_w : pl[c].pitch;
        pl[c].h = flavour == RPH_REF_JPEG_LIBJPEG ? k->samp_h : pl[c].rows;
        pl[c].p = (uint8_t *)malloc((size_t)pl[c].pitch * pl[c].rows);
        if (!pl[c].p) {
            rc = JR_ERR_OOM;
            break;
        }
        for (int by = 0; by < k->blocks_h; by++)
            for (int bx = 0; bx < k->blocks_w; bx++)
                idct_block(block_at(&d, k, bx, by), d.qt[k->tq], flavour, pl[c].p + (size_t)by * 8 * pl[c].pitch + bx * 8, (size_t)pl[c].pitch);
    }
    if (rc == JR_OK) {
        if (d.ncomp == 1) {
            for (int y = 0; y < d.h; y++) memcpy(out + (size_t)y * d.w, pl[0].p + (size_t)y * pl[0].pitch, (size_t)d.w);
        } else {
            const int hs = d.comp[0].H / d.comp[1].H, vs = d.comp[0].V / d.comp[1].V;
            for (int y = 0; y < d.h; y++)
                for (int x = 0; x < d.w; x++) {
                    const int yy = pl[0].p[(size_t)y * pl[0].pitch + x];
                    const int cb = upsampled(&pl[1], x, y, hs, vs, flavour), cr = upsampled(&pl[2], x, y, hs, vs, flavour);
                    ycc_to_rgb(yy, cb, cr, flavour, out + ((size_t)y * d.w + x) * 3);
                }
        }
    }
    for (int c = 0; c < 3; c++) free(pl[c].p);
    dec_free(&d);
    return rc;
}
