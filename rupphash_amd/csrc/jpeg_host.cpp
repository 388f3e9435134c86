// jpeg_host.cpp -- markers and Huffman entropy decoding of JPEG files on the host (row N3 of SURVEY 8f; the reference's
// load_image_fast "jpg" | "jpeg" arm, /root/reference/src/scanner.rs:473-551, hands the bytes to zune-jpeg 0.5.15).
//
// The entropy-coded segment is a serial bit stream, so this half stays on the host cores -- one image per thread, the way the
// reference's rayon workers decode (scanner.rs:1202) -- and stops at the QUANTISED coefficients, which have one right answer
// (ITU-T T.81).  Everything decoder specific (IDCT arithmetic, upsampling, colour conversion) is device code.
//
// Design: 64-bit bit accumulator refilled eight bytes at a time while no 0xFF is in sight; 9-bit lookup for the Huffman code, and for
// sequential AC coefficients a second 9-bit table that resolves code + magnitude bits in one probe when both fit; coefficients are
// written de-zigzagged (natural order) straight into the pinned staging buffer the device reads.
#include "jpeg_host.h"

#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "../../include/rupphash.h"

namespace rphj {
namespace {

const uint8_t ZIGZAG[64 + 16] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                 // a corrupt run may step past 63 before it is noticed: keep the lookups in bounds
                                 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

constexpr int LOOK = 9;

struct Huff {
    bool present = false;
    uint16_t look[1 << LOOK];    // (code length << 8) | symbol, 0 = longer than LOOK bits
    int16_t fast_ac[1 << LOOK];  // sequential AC: (value << 8) | (run << 4) | total bits, 0 = not resolvable in LOOK bits
    int32_t maxcode[18];         // largest code of each length, left-aligned to 16 bits (+1), for the slow path
    int32_t delta[17];           // symbol index = (code >> (16 - l)) + delta[l]
    uint8_t sym[256];
};

int build_huff(Huff &h, const uint8_t *counts /* [1..16] */, const uint8_t *symbols, int total)
{
    memset(h.look, 0, sizeof h.look);
    memset(h.fast_ac, 0, sizeof h.fast_ac);
    memcpy(h.sym, symbols, (size_t)total);
    uint8_t size[257];
    uint16_t code_of[256];
    int k = 0;
    for (int l = 1; l <= 16; l++)
        for (int i = 0; i < counts[l]; i++) size[k++] = (uint8_t)l;
    size[k] = 0;
    int code = 0;
    k = 0;
    for (int l = 1; l <= 16; l++) {
        h.delta[l] = k - code;
        if (counts[l]) {
            for (int i = 0; i < counts[l]; i++) code_of[k++] = (uint16_t)code++;
            if (code - 1 >= (1 << l)) return RPH_ERR_INVALID_ARG;
        }
        h.maxcode[l] = code << (16 - l);  // exclusive upper bound of length-l codes in a 16-bit window
        code <<= 1;
    }
    h.maxcode[17] = 0x7FFFFFFF;
    for (int i = 0; i < total; i++) {
        const int l = size[i];
        if (l > LOOK) continue;
        const int first = code_of[i] << (LOOK - l), n = 1 << (LOOK - l);
        for (int j = 0; j < n; j++) h.look[first + j] = (uint16_t)((l << 8) | h.sym[i]);
    }
    h.present = true;
    return RPH_OK;
}

void build_fast_ac(Huff &h)
{
    for (int i = 0; i < (1 << LOOK); i++) {
        const uint16_t e = h.look[i];
        if (!e) continue;
        const int len = e >> 8, rs = e & 255, run = rs >> 4, mag = rs & 15;
        if (mag == 0 || len + mag > LOOK) continue;
        int v = ((i << len) & ((1 << LOOK) - 1)) >> (LOOK - mag);  // the mag bits that follow the code
        if (v < (1 << (mag - 1))) v += (int)((~0u) << mag) + 1;     // EXTEND (T.81 F.2.2.1)
        if (v >= -128 && v <= 127) h.fast_ac[i] = (int16_t)((v * 256) + (run * 16) + (len + mag));
    }
}

// ---- bit reader over one entropy-coded segment: 0xFF00 is a stuffed 0xFF, any other 0xFFxx ends the segment (zeros follow)
struct Bits {
    const uint8_t *p, *end;
    uint64_t acc = 0;  // next bit = MSB
    int n = 0;
    bool hit = false;

    inline void refill()
    {
        if (!hit && end - p >= 8) {
            uint64_t v;
            memcpy(&v, p, 8);
            const uint64_t x = ~v;  // a zero byte in x = a 0xFF byte in the stream
            if (!((x - 0x0101010101010101ULL) & ~x & 0x8080808080808080ULL)) {
                const int k = (64 - n) >> 3;  // whole bytes that fit
                const uint64_t be = __builtin_bswap64(v);
                acc |= k == 8 ? be : ((be >> (64 - 8 * k)) << (64 - n - 8 * k));
                n += 8 * k;
                p += k;
                return;
            }
        }
        while (n <= 56) {
            uint64_t b = 0;
            if (!hit) {
                if (p >= end)
                    hit = true;
                else if (*p != 0xFF)
                    b = *p++;
                else if (p + 1 < end && p[1] == 0) {
                    b = 0xFF;
                    p += 2;
                } else
                    hit = true;  // marker (or a truncated file): p stays on its 0xFF
            }
            acc |= b << (56 - n);
            n += 8;
        }
    }
    inline void need32()
    {
        if (n < 32) refill();
    }
    inline uint32_t peek(int k) const { return (uint32_t)(acc >> (64 - k)); }
    inline void skip(int k)
    {
        acc <<= k;
        n -= k;
    }
    inline int bit()
    {
        if (n < 1) refill();
        const int b = (int)(acc >> 63);
        skip(1);
        return b;
    }
    inline int bits(int k)  // k <= 16
    {
        if (k == 0) return 0;
        if (n < k) refill();
        const int v = (int)peek(k);
        skip(k);
        return v;
    }
};

inline int decode_symbol(Bits &b, const Huff &h)  // caller guarantees >= 16 valid-or-zero bits (need32)
{
    const uint16_t e = h.look[b.peek(LOOK)];
    if (e) {
        b.skip(e >> 8);
        return e & 255;
    }
    const int32_t w = (int32_t)b.peek(16);
    int l = LOOK + 1;
    while (w >= h.maxcode[l]) l++;
    if (l > 16) return -1;
    b.skip(l);
    const int idx = (w >> (16 - l)) + h.delta[l];
    return h.sym[idx & 255];
}
inline int receive_extend(Bits &b, int s)  // s in 1..16, bits available
{
    const int v = (int)b.peek(s);
    b.skip(s);
    return v < (1 << (s - 1)) ? v + (int)((~0u) << s) + 1 : v;
}

struct Scan {
    int ns = 0, ci[3] = {0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0;
    uint32_t eobrun = 0;
};

struct Decoder {
    const uint8_t *data;
    size_t len, pos = 0;
    Frame &f;
    int16_t *coef;
    Huff dc[4], ac[4];

    Decoder(const uint8_t *d, size_t l, Frame &fr, int16_t *c) : data(d), len(l), f(fr), coef(c) {}

    inline int16_t *block_at(const Comp &k, uint32_t bx, uint32_t by) { return coef + (k.first_block + (uint64_t)by * k.blocks_w + bx) * 64; }

    // ---- sequential (T.81 F.2.2)
    inline int block_seq(Bits &b, Comp &k, int16_t *blk)
    {
        const Huff &hd = dc[k.dc_tbl], &ha = ac[k.ac_tbl];
        b.need32();
        int s = decode_symbol(b, hd);
        if (s < 0 || s > 15) return RPH_ERR_INVALID_ARG;
        if (s) {
            b.need32();
            k.pred += receive_extend(b, s);
        }
        blk[0] = (int16_t)k.pred;
        int kk = 1;
        do {
            b.need32();
            const int16_t fa = ha.fast_ac[b.peek(LOOK)];
            if (fa) {  // code and magnitude bits resolved by one probe
                kk += (fa >> 4) & 15;
                b.skip(fa & 15);
                blk[ZIGZAG[kk++]] = (int16_t)(fa >> 8);
                continue;
            }
            const int rs = decode_symbol(b, ha);
            if (rs < 0) return RPH_ERR_INVALID_ARG;
            s = rs & 15;
            const int r = rs >> 4;
            if (s == 0) {
                if (r != 15) break;  // end of block
                kk += 16;
                continue;
            }
            kk += r;
            blk[ZIGZAG[kk++]] = (int16_t)receive_extend(b, s);  // <= 16 + 16 bits since need32
        } while (kk < 64);
        return kk > 64 ? RPH_ERR_INVALID_ARG : RPH_OK;
    }
    // ---- progressive (T.81 G.1.2)
    inline int block_dc_first(Bits &b, Comp &k, int16_t *blk, const Scan &sc)
    {
        b.need32();
        const int s = decode_symbol(b, dc[k.dc_tbl]);
        if (s < 0 || s > 15) return RPH_ERR_INVALID_ARG;
        if (s) {
            b.need32();
            k.pred += receive_extend(b, s);
        }
        blk[0] = (int16_t)(k.pred * (1 << sc.al));
        return RPH_OK;
    }
    inline int block_ac_first(Bits &b, Comp &k, int16_t *blk, Scan &sc)
    {
        if (sc.eobrun) {
            sc.eobrun--;
            return RPH_OK;
        }
        const Huff &ha = ac[k.ac_tbl];
        for (int kk = sc.ss; kk <= sc.se;) {
            b.need32();
            const int rs = decode_symbol(b, ha);
            if (rs < 0) return RPH_ERR_INVALID_ARG;
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r == 15) {
                    kk += 16;
                    continue;
                }
                sc.eobrun = (1u << r) - 1;
                if (r) sc.eobrun += (uint32_t)b.bits(r);
                break;
            }
            kk += r;
            if (kk > 63) return RPH_ERR_INVALID_ARG;
            blk[ZIGZAG[kk++]] = (int16_t)(receive_extend(b, s) * (1 << sc.al));
        }
        return RPH_OK;
    }
    static inline void refine(Bits &b, int16_t &c, int p1, int m1)
    {
        if (b.bit() && (c & p1) == 0) c = (int16_t)(c >= 0 ? c + p1 : c + m1);
    }
    inline int block_ac_refine(Bits &b, Comp &k, int16_t *blk, Scan &sc)
    {
        const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
        const Huff &ha = ac[k.ac_tbl];
        int kk = sc.ss;
        if (sc.eobrun == 0) {
            for (; kk <= sc.se; kk++) {
                b.need32();
                const int rs = decode_symbol(b, ha);
                if (rs < 0) return RPH_ERR_INVALID_ARG;
                int r = rs >> 4;
                const int s = rs & 15;
                int value = 0;
                if (s) {
                    if (s != 1) return RPH_ERR_INVALID_ARG;
                    value = b.bit() ? p1 : m1;
                } else if (r != 15) {
                    sc.eobrun = 1u << r;
                    if (r) sc.eobrun += (uint32_t)b.bits(r);
                    break;
                }
                // step over r coefficients with zero history; the nonzero ones passed on the way take a correction bit each
                do {
                    int16_t &c = blk[ZIGZAG[kk]];
                    if (c != 0)
                        refine(b, c, p1, m1);
                    else if (--r < 0)
                        break;
                    kk++;
                } while (kk <= sc.se);
                if (value && kk <= sc.se) blk[ZIGZAG[kk]] = (int16_t)value;
            }
        }
        if (sc.eobrun > 0) {
            for (; kk <= sc.se; kk++) {
                int16_t &c = blk[ZIGZAG[kk]];
                if (c != 0) refine(b, c, p1, m1);
            }
            sc.eobrun--;
        }
        return RPH_OK;
    }

    template <int MODE>  // 0 sequential, 1 DC first, 2 DC refine, 3 AC first, 4 AC refine
    inline int one(Bits &b, Comp &k, int16_t *blk, Scan &sc)
    {
        if (MODE == 0) return block_seq(b, k, blk);
        if (MODE == 1) return block_dc_first(b, k, blk, sc);
        if (MODE == 2) {
            if (b.bit()) blk[0] = (int16_t)(blk[0] | (1 << sc.al));
            return RPH_OK;
        }
        if (MODE == 3) return block_ac_first(b, k, blk, sc);
        return block_ac_refine(b, k, blk, sc);
    }

    // RSTn (T.81 E.2.4): drop the padding bits, take the marker, reset the predictions and the end-of-band run
    int restart(Bits &b, Scan &sc)
    {
        b.acc = 0;
        b.n = 0;
        const uint8_t *p = b.p;
        while (p + 1 < b.end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF)) p++;
        if (p + 1 >= b.end || p[1] < 0xD0 || p[1] > 0xD7) return RPH_ERR_INVALID_ARG;
        b.p = p + 2;
        b.hit = false;
        for (int i = 0; i < sc.ns; i++) f.comp[sc.ci[i]].pred = 0;
        sc.eobrun = 0;
        return RPH_OK;
    }

    template <int MODE>
    int scan_body(Scan &sc)
    {
        Bits b;
        b.p = data + pos;
        b.end = data + len;
        for (int i = 0; i < sc.ns; i++) f.comp[sc.ci[i]].pred = 0;
        sc.eobrun = 0;
        const bool single = sc.ns == 1;
        const uint32_t mx_n = single ? f.comp[sc.ci[0]].real_bw : f.mcus_x, my_n = single ? f.comp[sc.ci[0]].real_bh : f.mcus_y;
        uint32_t until = f.restart_interval;
        for (uint32_t my = 0; my < my_n; my++)
            for (uint32_t mx = 0; mx < mx_n; mx++) {
                if (f.restart_interval && until == 0) {
                    const int rc = restart(b, sc);
                    if (rc) return rc;
                    until = f.restart_interval;
                }
                if (single) {
                    Comp &k = f.comp[sc.ci[0]];
                    const int rc = one<MODE>(b, k, block_at(k, mx, my), sc);
                    if (rc) return rc;
                } else {
                    for (int i = 0; i < sc.ns; i++) {
                        Comp &k = f.comp[sc.ci[i]];
                        for (uint32_t v = 0; v < k.V; v++)
                            for (uint32_t h = 0; h < k.H; h++) {
                                const int rc = one<MODE>(b, k, block_at(k, mx * k.H + h, my * k.V + v), sc);
                                if (rc) return rc;
                            }
                    }
                }
                until--;
            }
        // the marker that ends the scan: the reader never steps over one, so it is at or after b.p
        const uint8_t *p = b.p;
        while (p + 1 < b.end && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
        pos = (size_t)(p - data);
        return RPH_OK;
    }

    int parse_dht(const uint8_t *p, int n)
    {
        while (n > 0) {
            if (n < 17) return RPH_ERR_INVALID_ARG;
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) return RPH_ERR_INVALID_ARG;
            uint8_t counts[17];
            int total = 0;
            counts[0] = 0;
            for (int l = 1; l <= 16; l++) {
                counts[l] = p[l];
                total += p[l];
            }
            if (total > 256 || n < 17 + total) return RPH_ERR_INVALID_ARG;
            Huff &h = tc ? ac[th] : dc[th];
            const int rc = build_huff(h, counts, p + 17, total);
            if (rc) return rc;
            if (tc) build_fast_ac(h);
            p += 17 + total;
            n -= 17 + total;
        }
        return RPH_OK;
    }

    int parse_sos(const uint8_t *p, int n, Scan &sc)
    {
        if (n < 1) return RPH_ERR_INVALID_ARG;
        sc.ns = p[0];
        if (sc.ns < 1 || sc.ns > f.ncomp || n < 1 + 2 * sc.ns + 3) return RPH_ERR_INVALID_ARG;
        for (int i = 0; i < sc.ns; i++) {
            int c = 0;
            while (c < f.ncomp && f.comp[c].id != p[1 + 2 * i]) c++;
            if (c == f.ncomp) return RPH_ERR_INVALID_ARG;
            for (int j = 0; j < i; j++)
                if (sc.ci[j] == c) return RPH_ERR_INVALID_ARG;
            sc.ci[i] = c;
            f.comp[c].dc_tbl = p[2 + 2 * i] >> 4;
            f.comp[c].ac_tbl = p[2 + 2 * i] & 15;
            if (f.comp[c].dc_tbl > 3 || f.comp[c].ac_tbl > 3) return RPH_ERR_INVALID_ARG;
        }
        const uint8_t *q = p + 1 + 2 * sc.ns;
        if (f.progressive) {
            sc.ss = q[0];
            sc.se = q[1];
            sc.ah = q[2] >> 4;
            sc.al = q[2] & 15;
            if (sc.ss > sc.se || sc.se > 63 || sc.al > 13) return RPH_ERR_INVALID_ARG;
            if (sc.ss == 0 && sc.se != 0) return RPH_ERR_INVALID_ARG;
            if (sc.ss != 0 && sc.ns != 1) return RPH_ERR_INVALID_ARG;
        }
        for (int i = 0; i < sc.ns; i++) {
            const Comp &k = f.comp[sc.ci[i]];
            const bool need_dc = !f.progressive || (sc.ss == 0 && sc.ah == 0), need_ac = !f.progressive || sc.ss != 0;
            if ((need_dc && !dc[k.dc_tbl].present) || (need_ac && !ac[k.ac_tbl].present)) return RPH_ERR_INVALID_ARG;
        }
        return RPH_OK;
    }

    int run_scan(Scan &sc)
    {
        if (!f.progressive) return scan_body<0>(sc);
        if (sc.ss == 0) return sc.ah == 0 ? scan_body<1>(sc) : scan_body<2>(sc);
        return sc.ah == 0 ? scan_body<3>(sc) : scan_body<4>(sc);
    }
};

inline int rd16(const uint8_t *p) { return (p[0] << 8) | p[1]; }

int parse_dqt(Frame &f, const uint8_t *p, int n)
{
    while (n > 0) {
        const int pq = p[0] >> 4, tq = p[0] & 15;
        if (tq > 3 || pq > 1) return RPH_ERR_INVALID_ARG;
        const int need = 1 + 64 * (pq + 1);
        if (n < need) return RPH_ERR_INVALID_ARG;
        for (int k = 0; k < 64; k++) f.qt[tq][ZIGZAG[k]] = (uint16_t)(pq ? rd16(p + 1 + 2 * k) : p[1 + k]);
        f.qt_present[tq] = true;
        p += need;
        n -= need;
    }
    return RPH_OK;
}

int parse_sof(Frame &f, const uint8_t *p, int n, bool progressive)
{
    if (f.have_sof || n < 6) return RPH_ERR_INVALID_ARG;
    if (p[0] != 8) return RPH_ERR_UNSUPPORTED;
    f.h = (uint32_t)rd16(p + 1);
    f.w = (uint32_t)rd16(p + 3);
    f.ncomp = p[5];
    if (f.w == 0 || f.h == 0) return RPH_ERR_UNSUPPORTED;
    if (f.ncomp != 1 && f.ncomp != 3) return RPH_ERR_UNSUPPORTED;
    if (n < 6 + 3 * f.ncomp) return RPH_ERR_INVALID_ARG;
    f.progressive = progressive;
    f.Hmax = f.Vmax = 1;
    for (int c = 0; c < f.ncomp; c++) {
        Comp &k = f.comp[c];
        k.id = p[6 + 3 * c];
        k.H = p[7 + 3 * c] >> 4;
        k.V = p[7 + 3 * c] & 15;
        k.tq = p[8 + 3 * c];
        k.dc_tbl = k.ac_tbl = 0;
        k.pred = 0;
        if (k.H < 1 || k.H > 4 || k.V < 1 || k.V > 4 || k.tq > 3) return RPH_ERR_INVALID_ARG;
        if (f.ncomp == 1) k.H = k.V = 1;  // one component: the MCU is one block (T.81 A.2.2)
        if (k.H > f.Hmax) f.Hmax = k.H;
        if (k.V > f.Vmax) f.Vmax = k.V;
    }
    if (f.ncomp == 3) {
        const Comp &y = f.comp[0];
        if (y.H != f.Hmax || y.V != f.Vmax || y.H > 2 || y.V > 2) return RPH_ERR_UNSUPPORTED;
        for (int c = 1; c < 3; c++) {
            const Comp &k = f.comp[c];
            if (!((k.H == 1 && k.V == 1) || (k.H == y.H && k.V == y.V))) return RPH_ERR_UNSUPPORTED;
            if (k.H != f.comp[1].H || k.V != f.comp[1].V) return RPH_ERR_UNSUPPORTED;
        }
        if (f.comp[0].id == 'R' && f.comp[1].id == 'G' && f.comp[2].id == 'B') return RPH_ERR_UNSUPPORTED;
        if (y.H * y.V + 2 * f.comp[1].H * f.comp[1].V > 10) return RPH_ERR_INVALID_ARG;  // T.81 B.2.3: at most 10 blocks per MCU
    }
    f.mcus_x = (f.w + 8 * f.Hmax - 1) / (8 * f.Hmax);
    f.mcus_y = (f.h + 8 * f.Vmax - 1) / (8 * f.Vmax);
    uint64_t total = 0;
    for (int c = 0; c < f.ncomp; c++) {
        Comp &k = f.comp[c];
        k.blocks_w = f.mcus_x * k.H;
        k.blocks_h = f.mcus_y * k.V;
        k.samp_w = (f.w * k.H + f.Hmax - 1) / f.Hmax;
        k.samp_h = (f.h * k.V + f.Vmax - 1) / f.Vmax;
        k.real_bw = (k.samp_w + 7) / 8;
        k.real_bh = (k.samp_h + 7) / 8;
        k.first_block = total;
        total += (uint64_t)k.blocks_w * k.blocks_h;
    }
    f.total_blocks = total;
    f.have_sof = true;
    return RPH_OK;
}

// One pass over the markers.  dec == nullptr: stop at the first SOS (frame header only).
int walk(const uint8_t *data, size_t len, Frame &f, Decoder *dec)
{
    if (!data || len < 4 || data[0] != 0xFF || data[1] != 0xD8) return RPH_ERR_INVALID_ARG;
    size_t pos = 2;
    bool seen_scan = false;
    for (;;) {
        while (pos < len && data[pos] != 0xFF) pos++;
        while (pos < len && data[pos] == 0xFF) pos++;
        if (pos >= len) break;
        const int m = data[pos++];
        if (m == 0xD9) break;
        if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > len) return seen_scan ? RPH_OK : RPH_ERR_INVALID_ARG;
        const int n = rd16(data + pos) - 2;
        const uint8_t *p = data + pos + 2;
        if (n < 0 || pos + 2 + (size_t)n > len) return seen_scan ? RPH_OK : RPH_ERR_INVALID_ARG;  // truncated tail: keep what was decoded
        pos += 2 + (size_t)n;
        int rc = RPH_OK;
        switch (m) {
        case 0xDB: rc = parse_dqt(f, p, n); break;
        case 0xC4: rc = dec ? dec->parse_dht(p, n) : RPH_OK; break;
        case 0xC0:
        case 0xC1:
        case 0xC2:
            if (dec) {  // the frame is already parsed: a second SOF is an error
                if (seen_scan) return RPH_ERR_INVALID_ARG;
            } else
                rc = parse_sof(f, p, n, m == 0xC2);
            break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return RPH_ERR_UNSUPPORTED;  // lossless, differential, arithmetic coding
        case 0xDD:
            if (n < 2) return RPH_ERR_INVALID_ARG;
            f.restart_interval = (uint32_t)rd16(p);
            break;
        case 0xEE:
            if (n >= 12 && memcmp(p, "Adobe", 5) == 0) f.adobe_transform = p[11];
            break;
        case 0xDA: {
            if (!f.have_sof) return RPH_ERR_INVALID_ARG;
            if (!dec) return RPH_OK;
            Scan sc;
            rc = dec->parse_sos(p, n, sc);
            if (rc) return rc;
            dec->pos = pos;
            rc = dec->run_scan(sc);
            pos = dec->pos;
            seen_scan = true;
            break;
        }
        default: break;  // APPn, COM
        }
        if (rc) return rc;
    }
    if (!f.have_sof) return RPH_ERR_INVALID_ARG;
    if (!dec) return RPH_OK;
    if (!seen_scan) return RPH_ERR_INVALID_ARG;
    if (f.ncomp == 3 && f.adobe_transform == 0) return RPH_ERR_UNSUPPORTED;  // Adobe RGB: no colour transform
    for (int c = 0; c < f.ncomp; c++)
        if (!f.qt_present[f.comp[c].tq]) return RPH_ERR_INVALID_ARG;
    return RPH_OK;
}

}  // namespace

int parse_frame(const uint8_t *data, size_t len, Frame &f)
{
    f = Frame();
    return walk(data, len, f, nullptr);
}

int build_device_lut(const TableSpec &t, DeviceLut &out)
{
    memset(&out, 0, sizeof out);
    memcpy(out.sym, t.symbols, t.total);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        out.delta[l] = k - code;
        for (int i = 0; i < t.counts[l]; i++, k++, code++) {
            if (code >= (1 << l)) return RPH_ERR_INVALID_ARG;
            if (l <= 10) {
                const int first = code << (10 - l), n = 1 << (10 - l);
                for (int j = 0; j < n; j++) out.look[first + j] = (uint16_t)((l << 8) | t.symbols[k]);
            }
        }
        out.maxcode[l] = code << (16 - l);
        code <<= 1;
    }
    out.maxcode[17] = 0x7FFFFFFF;
    return RPH_OK;
}

namespace {
// 32 bytes per step where the CPU has AVX2 (a 0xFF turns up every ~256 bytes of entropy data, so scanning for it and copying whole
// runs is what this pass spends its time in), memchr + memcpy otherwise
#if defined(__x86_64__)
// copies [p, first 0xFF) to out while looking for it: one pass over the bytes instead of memchr + memcpy
__attribute__((target("avx2"))) const uint8_t *copy_until_ff_avx2(const uint8_t *p, const uint8_t *end, uint8_t *&out)
{
    const __m256i ff = _mm256_set1_epi8((char)0xFF);
    uint8_t *o = out;
    while (end - p >= 32) {
        const __m256i v = _mm256_loadu_si256((const __m256i *)p);
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, ff));
        _mm256_storeu_si256((__m256i *)o, v);  // (the caller guarantees 32 bytes of slack behind the bytes that count)
        if (m) {
            const int k = __builtin_ctz(m);
            out = o + k;
            return p + k;
        }
        p += 32;
        o += 32;
    }
    while (p < end && *p != 0xFF) *o++ = *p++;
    out = o;
    return p;
}
#endif

// entropy bytes of one scan, byte stuffing undone and RSTn dropped; returns the position of the marker that ends the scan
const uint8_t *destuff(const uint8_t *p, const uint8_t *end, uint8_t *&out, uint8_t *out_end, std::vector<uint32_t> *marks)
{
    const uint8_t *const scan_out = out;
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
#endif
    while (p < end) {
        const uint8_t *q;
#if defined(__x86_64__)
        if (avx2) {
            // the output never gets ahead of the input (bytes are only ever dropped), and the buffer holds len + 160 bytes: 32 bytes of
            // slack for the whole-vector stores are there as long as 64 bytes remain for the scan's padding, checked here
            if ((size_t)(out_end - out) < (size_t)(end - p) + 64) return nullptr;
            q = copy_until_ff_avx2(p, end, out);
        } else
#endif
        {
            const uint8_t *f = (const uint8_t *)memchr(p, 0xFF, (size_t)(end - p));
            q = f ? f : end;
            const size_t run = (size_t)(q - p);
            if ((size_t)(out_end - out) < run + 1) return nullptr;
            memcpy(out, p, run);
            out += run;
        }
        if (q >= end) return end;
        if (q + 1 >= end) return q;
        const uint8_t m = q[1];
        if (m == 0x00) {
            if (out >= out_end) return nullptr;
            *out++ = 0xFF;
            p = q + 2;
        } else if (m >= 0xD0 && m <= 0xD7) {
            p = q + 2;  // restart marker: the decoder counts MCUs
            if (marks) marks->push_back((uint32_t)(out - scan_out));
        } else if (m == 0xFF) {
            p = q + 1;  // fill byte
        } else {
            return q;  // the marker that ends the scan
        }
    }
    return end;
}
}  // namespace

int prepare_stream(const uint8_t *data, size_t len, Frame &f, StreamPlan &plan, uint8_t *out, size_t cap, size_t *used, InternTable intern, void *store,
                   std::vector<uint32_t> *restart_marks)
{
    if (restart_marks) restart_marks->clear();
    if (!f.have_sof) return RPH_ERR_UNSUPPORTED;
    plan.n_scans = 0;
    plan.prog.clear();
    TableSpec dc[4], ac[4];
    bool dc_present[4] = {false, false, false, false}, ac_present[4] = {false, false, false, false};
    uint32_t dc_id[4], ac_id[4];  // interned lazily, when a scan uses the table
    bool dc_known[4] = {false, false, false, false}, ac_known[4] = {false, false, false, false};
    for (int i = 0; i < 4; i++) f.qt_present[i] = false;
    f.restart_interval = 0;
    f.adobe_transform = -1;
    uint8_t *o = out, *o_end = out + cap;
    size_t pos = 2;
    uint32_t done_mask = 0;
    int8_t coef_bits[4][64];  // progressive files: the Al each coefficient of each component was last coded with (-1: not yet)
    memset(coef_bits, -1, sizeof coef_bits);
    for (;;) {
        while (pos < len && data[pos] != 0xFF) pos++;
        while (pos < len && data[pos] == 0xFF) pos++;
        if (pos >= len) break;
        const int m = data[pos++];
        if (m == 0xD9) break;
        if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (pos + 2 > len) break;
        const int n = rd16(data + pos) - 2;
        const uint8_t *p = data + pos + 2;
        if (n < 0 || pos + 2 + (size_t)n > len) break;
        pos += 2 + (size_t)n;
        switch (m) {
        case 0xDB: {
            const int rc = parse_dqt(f, p, n);
            if (rc) return rc;
            break;
        }
        case 0xC4: {
            int left = n;
            while (left > 0) {
                if (left < 17) return RPH_ERR_INVALID_ARG;
                const int tc = p[0] >> 4, th = p[0] & 15;
                if (tc > 1 || th > 3) return RPH_ERR_INVALID_ARG;
                TableSpec &t = tc ? ac[th] : dc[th];
                int total = 0;
                t.counts[0] = 0;
                for (int l = 1; l <= 16; l++) {
                    t.counts[l] = p[l];
                    total += p[l];
                }
                if (total > 256 || left < 17 + total) return RPH_ERR_INVALID_ARG;
                memset(t.symbols, 0, sizeof t.symbols);
                memcpy(t.symbols, p + 17, (size_t)total);
                t.total = (uint16_t)total;
                (tc ? ac_present : dc_present)[th] = true;
                (tc ? ac_known : dc_known)[th] = false;
                p += 17 + total;
                left -= 17 + total;
            }
            break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            return RPH_ERR_UNSUPPORTED;
        case 0xDD:
            if (n < 2) return RPH_ERR_INVALID_ARG;
            f.restart_interval = (uint32_t)rd16(p);
            break;
        case 0xEE:
            if (n >= 12 && memcmp(p, "Adobe", 5) == 0) f.adobe_transform = p[11];
            break;
        case 0xDA: {
            if (plan.n_scans == (f.progressive ? MAX_PROG_SCANS : 4)) return RPH_ERR_UNSUPPORTED;
            if (f.progressive && f.restart_interval) return RPH_ERR_UNSUPPORTED;  // (rare; the host decoder takes such a file)
            ScanPlan local;
            if (f.progressive) plan.prog.push_back(local);
            ScanPlan &sc = f.progressive ? plan.prog.back() : plan.scan[plan.n_scans];
            sc = local;
            if (n < 1) return RPH_ERR_INVALID_ARG;
            sc.ns = p[0];
            if (sc.ns < 1 || sc.ns > f.ncomp || n < 1 + 2 * sc.ns + 3) return RPH_ERR_INVALID_ARG;
            bool need_dc = true, need_ac = true;
            if (f.progressive) {  // T.81 G.1.1.1
                const uint8_t *q = p + 1 + 2 * sc.ns;
                sc.ss = q[0];
                sc.se = q[1];
                sc.ah = q[2] >> 4;
                sc.al = q[2] & 15;
                if (sc.ss > sc.se || sc.se > 63 || sc.al > 13) return RPH_ERR_INVALID_ARG;
                if (sc.ss == 0 && sc.se != 0) return RPH_ERR_INVALID_ARG;
                if (sc.ss != 0 && sc.ns != 1) return RPH_ERR_INVALID_ARG;
                need_dc = sc.ss == 0 && sc.ah == 0;
                need_ac = sc.ss != 0;
            }
            for (int i = 0; i < sc.ns; i++) {
                int c = 0;
                while (c < f.ncomp && f.comp[c].id != p[1 + 2 * i]) c++;
                if (c == f.ncomp) return RPH_ERR_INVALID_ARG;
                if (f.progressive) {
                    for (int j = 0; j < i; j++)
                        if (sc.ci[j] == c) return RPH_ERR_INVALID_ARG;
                } else if ((done_mask >> c) & 1) {
                    return RPH_ERR_INVALID_ARG;  // a component is coded once in a sequential file
                }
                done_mask |= 1u << c;
                sc.ci[i] = (uint8_t)c;
                const int td = p[2 + 2 * i] >> 4, ta = p[2 + 2 * i] & 15;
                if (td > 3 || ta > 3 || (need_dc && !dc_present[td]) || (need_ac && !ac_present[ta])) return RPH_ERR_INVALID_ARG;
                if (need_dc && !dc_known[td]) {
                    dc_id[td] = intern(store, dc[td]);
                    dc_known[td] = true;
                }
                if (need_ac && !ac_known[ta]) {
                    ac_id[ta] = intern(store, ac[ta]);
                    ac_known[ta] = true;
                }
                if ((need_dc && dc_id[td] == UINT32_MAX) || (need_ac && ac_id[ta] == UINT32_MAX)) return RPH_ERR_INVALID_ARG;
                sc.dc[i] = need_dc ? dc_id[td] : 0;
                sc.ac[i] = need_ac ? ac_id[ta] : 0;
            }
            if (f.progressive) {
                // The device walks the scans of a file side by side and applies the refinements when the coefficients are complete: that
                // equals decoding scan after scan only for a progression as T.81 G.1.1.1.1 describes it -- every coefficient first coded
                // once (Ah = 0), then refined bit by bit (Ah = the Al before).  Anything else (libjpeg: "bogus progression", a warning)
                // stays with the host decoder, which takes the scans in file order whatever they say.
                for (int i = 0; i < sc.ns; i++)
                    for (int k = sc.ss; k <= sc.se; k++) {
                        int8_t &bits = coef_bits[sc.ci[i]][k];
                        if (sc.ah != (bits < 0 ? 0 : bits) || (bits < 0 && sc.ah != 0) || (bits >= 0 && sc.ah == 0)) return RPH_ERR_UNSUPPORTED;
                        bits = (int8_t)sc.al;
                    }
            }
            sc.restart_interval = f.restart_interval;
            sc.stream_off = (uint32_t)(o - out);
            const bool mark = restart_marks && plan.n_scans == 0 && f.restart_interval != 0;
            const uint8_t *stop = destuff(data + pos, data + len, o, o_end, mark ? restart_marks : nullptr);
            if (!stop || (size_t)(o_end - o) < 32) return RPH_ERR_CAPACITY;
            sc.stream_len = (uint32_t)(o - out) - sc.stream_off;
            memset(o, 0, 32);
            o += 32;
            pos = (size_t)(stop - data);
            plan.n_scans++;
            break;
        }
        default: break;
        }
    }
    if (plan.n_scans == 0 || (!f.progressive && done_mask != (1u << f.ncomp) - 1)) return RPH_ERR_INVALID_ARG;
    if (f.ncomp == 3 && f.adobe_transform == 0) return RPH_ERR_UNSUPPORTED;
    for (int c = 0; c < f.ncomp; c++)
        if (!f.qt_present[f.comp[c].tq]) return RPH_ERR_INVALID_ARG;
    if (restart_marks && !restart_marks->empty()) {
        // usable only if the file is one scan and the marks are exactly the interval boundaries
        const ScanPlan &sc = plan.scan[0];
        const uint64_t mcus = sc.ns == 1 ? (uint64_t)f.comp[sc.ci[0]].real_bw * f.comp[sc.ci[0]].real_bh : (uint64_t)f.mcus_x * f.mcus_y;
        const uint64_t intervals = sc.restart_interval ? (mcus + sc.restart_interval - 1) / sc.restart_interval : 0;
        if (plan.n_scans != 1 || intervals == 0 || restart_marks->size() + 1 != intervals) restart_marks->clear();
    }
    *used = (size_t)(o - out);
    return RPH_OK;
}

int decode_coefficients(const uint8_t *data, size_t len, Frame &f, int16_t *coef)
{
    if (!f.have_sof || !coef) return RPH_ERR_INVALID_ARG;
    memset(coef, 0, (size_t)f.total_blocks * 64 * sizeof(int16_t));
    f.restart_interval = 0;
    f.adobe_transform = -1;
    for (int i = 0; i < 4; i++) f.qt_present[i] = false;
    Decoder dec(data, len, f, coef);
    return walk(data, len, f, &dec);
}

}  // namespace rphj
