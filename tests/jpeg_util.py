"""Test helpers for the JPEG path: generated test images, Pillow encodes, and a small baseline JPEG *encoder* for the
sampling layouts Pillow cannot write (4:4:0 = 1x2 luma, all components 2x2, custom restart intervals, 16-bit tables).
Data generation only -- nothing here decodes."""
import io

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,
                   42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])

# ITU-T T.81 Annex K.3 typical Huffman tables
DC_L = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_C = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_L = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
        [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
         0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
         0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
         0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
         0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
         0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
         0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
         0xfa])
AC_C = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
        [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
         0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
         0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
         0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
         0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
         0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
         0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
         0xfa])


def make_image(w, h, mode="RGB", seed=1):
    """smooth structure + noise, so that DC, low and high AC coefficients, EOBs and ZRLs all occur; returns a PIL image"""
    from PIL import Image

    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(np.sin(x / 7.0 + c) + np.cos(y / 5.0 - c)) * 60 + 128 for c in range(3)], -1)
    a = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
    im = Image.fromarray(a, "RGB")
    return im.convert("L") if mode == "L" else im


def pillow_jpeg(im, **kw):
    from PIL import ImageFile

    # optimised / progressive encodes are written in one piece: the encoder's buffer must hold the whole (noisy, hence large) stream
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 8 * im.size[0] * im.size[1] + (1 << 16))
    buf = io.BytesIO()
    im.save(buf, "JPEG", **kw)
    return buf.getvalue()


def pillow_decode(data):
    from PIL import Image

    return np.array(Image.open(io.BytesIO(data)))


def _codes(counts, symbols):
    table, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(counts[length - 1]):
            table[symbols[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return table


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, nbits):
        self.acc = (self.acc << nbits) | (value & ((1 << nbits) - 1))
        self.n += nbits
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _fdct_blocks(plane):
    """plane (multiple of 8 both ways, float) -> (by, bx, 8, 8) DCT-II coefficients, JPEG normalisation"""
    k = np.arange(8)
    c = np.cos((2 * k[None, :] + 1) * k[:, None] * np.pi / 16) * 0.5
    c[0] *= 1 / np.sqrt(2)
    h, w = plane.shape
    b = plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3) - 128.0
    return np.einsum("ux,abxy,vy->abuv", c, b, c)


def encode_baseline(rgb, sampling=((1, 2), (1, 1), (1, 1)), quality_scale=1.0, restart_interval=0, gray=False, sixteen_bit_tables=False, interleaved=True):
    """A plain baseline (SOF0) encoder: rgb (h, w, 3) uint8 [or (h, w) when gray]; sampling = (H, V) per component.
    Chroma is box-averaged down.  interleaved=False writes one scan per component (each over the component's own block grid,
    T.81 A.2.2).  Returns the JPEG byte string."""
    rgb = np.asarray(rgb)
    if gray:
        h, w = rgb.shape
        comps = [rgb.astype(np.float64)]
        sampling = ((1, 1),)
    else:
        h, w, _ = rgb.shape
        r, g, b = [rgb[..., i].astype(np.float64) for i in range(3)]
        comps = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128, 0.5 * r - 0.418688 * g - 0.081312 * b + 128]
    hmax = max(s[0] for s in sampling)
    vmax = max(s[1] for s in sampling)
    mcus_x = -(-w // (8 * hmax))
    mcus_y = -(-h // (8 * vmax))
    ql = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22,
                   37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
    qc = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)
    top = 65535 if sixteen_bit_tables else 255
    qts = [np.clip(np.round(q * quality_scale), 1, top).astype(np.int64) for q in (ql, qc)]
    blocks = []
    for ci, (plane, (H, V)) in enumerate(zip(comps, sampling)):
        fx, fy = hmax // H, vmax // V
        pw, ph = mcus_x * 8 * hmax, mcus_y * 8 * vmax
        full = np.pad(plane, ((0, ph - h), (0, pw - w)), mode="edge")
        small = full.reshape(ph // fy, fy, pw // fx, fx).mean(axis=(1, 3))
        q = qts[0 if ci == 0 else 1].reshape(8, 8)
        blocks.append(np.round(_fdct_blocks(small) / q).astype(np.int64))
    out = bytearray(b"\xff\xd8")

    def seg(marker, payload):
        out.extend(bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload)

    for t, q in enumerate(qts if not gray else qts[:1]):
        zz = q[ZIGZAG]
        seg(0xDB, bytes([(0x10 if sixteen_bit_tables else 0) | t]) + (b"".join(int(v).to_bytes(2, "big") for v in zz) if sixteen_bit_tables else bytes(int(v) for v in zz)))
    sof = bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([len(comps)])
    for ci, (H, V) in enumerate(sampling):
        sof += bytes([ci + 1, (H << 4) | V, 0 if ci == 0 else 1])
    seg(0xC0, sof)
    tables = [(0x00, DC_L), (0x10, AC_L)] + ([] if gray else [(0x01, DC_C), (0x11, AC_C)])
    for tid, (counts, symbols) in tables:
        seg(0xC4, bytes([tid]) + bytes(counts) + bytes(symbols))
    if restart_interval:
        seg(0xDD, restart_interval.to_bytes(2, "big"))
    dc_codes = [_codes(*DC_L), _codes(*DC_C)]
    ac_codes = [_codes(*AC_L), _codes(*AC_C)]
    pred = [0] * len(comps)

    def put_block(bits, blk, ci):
        t = 0 if ci == 0 else 1
        zz = blk.reshape(64)[ZIGZAG]
        diff = int(zz[0]) - pred[ci]
        pred[ci] = int(zz[0])
        s = abs(diff).bit_length()
        bits.put(*dc_codes[t][s])
        if s:
            bits.put(diff if diff >= 0 else diff + (1 << s) - 1, s)
        run = 0
        last = np.nonzero(zz[1:])[0]
        end = (last[-1] + 1) if len(last) else 0
        for k in range(1, end + 1):
            v = int(zz[k])
            if v == 0:
                run += 1
                continue
            while run > 15:
                bits.put(*ac_codes[t][0xF0])
                run -= 16
            s = abs(v).bit_length()
            bits.put(*ac_codes[t][(run << 4) | s])
            bits.put(v if v >= 0 else v + (1 << s) - 1, s)
            run = 0
        if end < 63:
            bits.put(*ac_codes[t][0x00])

    def write_scan(cis):
        nonlocal pred
        sos = bytes([len(cis)])
        for ci in cis:
            sos += bytes([ci + 1, 0x00 if ci == 0 else 0x11])
        seg(0xDA, sos + bytes([0, 63, 0]))
        bits = _Bits()
        pred = [0] * len(comps)
        if len(cis) == 1 and len(comps) > 1:  # the component's own grid: ceil(samples / 8) blocks each way, one block per MCU
            ci = cis[0]
            H, V = sampling[ci]
            bw = -(-(-(-w * H // hmax)) // 8)
            bh = -(-(-(-h * V // vmax)) // 8)
            units = [[(ci, by, bx)] for by in range(bh) for bx in range(bw)]
        else:
            units = [[(ci, my * sampling[ci][1] + v, mx * sampling[ci][0] + hh) for ci in cis for v in range(sampling[ci][1]) for hh in range(sampling[ci][0])]
                     for my in range(mcus_y) for mx in range(mcus_x)]
        rst = 0
        for count, unit in enumerate(units):
            if restart_interval and count and count % restart_interval == 0:
                bits.flush()
                bits.out.extend(bytes([0xFF, 0xD0 + (rst & 7)]))
                rst += 1
                pred = [0] * len(comps)
            for ci, by, bx in unit:
                put_block(bits, blocks[ci][by, bx], ci)
        bits.flush()
        out.extend(bits.out)

    if interleaved or len(comps) == 1:
        write_scan(list(range(len(comps))))
    else:
        for ci in range(len(comps)):
            write_scan([ci])
    out.extend(b"\xff\xd9")
    return bytes(out)


# ---- a progressive (SOF2) encoder with an arbitrary scan script --------------------------------------------------------------------
# Pillow / libjpeg only write libjpeg's default script.  The decoders' hard cases are elsewhere: bands refined in another order than they
# were first coded, deep successive approximation, end-of-band runs that carry correction bits, sixteen zero-history coefficients stepped
# over inside a refinement scan, codes longer than the lookup tables.  T.81 G.1.2 procedures as libjpeg's encoder arranges them (correction
# bits are held back until the symbol they follow is out).

def _quantised_blocks(rgb, sampling, quality_scale, gray):
    rgb = np.asarray(rgb)
    if gray:
        h, w = rgb.shape
        comps = [rgb.astype(np.float64)]
        sampling = ((1, 1),)
    else:
        h, w, _ = rgb.shape
        r, g, b = [rgb[..., i].astype(np.float64) for i in range(3)]
        comps = [0.299 * r + 0.587 * g + 0.114 * b, -0.168736 * r - 0.331264 * g + 0.5 * b + 128, 0.5 * r - 0.418688 * g - 0.081312 * b + 128]
    hmax = max(s[0] for s in sampling)
    vmax = max(s[1] for s in sampling)
    mcus_x = -(-w // (8 * hmax))
    mcus_y = -(-h // (8 * vmax))
    ql = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22,
                   37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99])
    qc = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32)
    qts = [np.clip(np.round(q * quality_scale), 1, 255).astype(np.int64) for q in (ql, qc)]
    blocks = []
    for ci, (plane, (H, V)) in enumerate(zip(comps, sampling)):
        fx, fy = hmax // H, vmax // V
        pw, ph = mcus_x * 8 * hmax, mcus_y * 8 * vmax
        full = np.pad(plane, ((0, ph - h), (0, pw - w)), mode="edge")
        small = full.reshape(ph // fy, fy, pw // fx, fx).mean(axis=(1, 3))
        q = qts[0 if ci == 0 else 1].reshape(8, 8)
        blk = np.round(_fdct_blocks(small) / q).astype(np.int64)
        blocks.append(blk.reshape(blk.shape[0], blk.shape[1], 64)[:, :, ZIGZAG])  # zigzag order
    return h, w, sampling, hmax, vmax, mcus_x, mcus_y, qts, blocks


def _table_for(used, long_codes):
    """a Huffman table (counts[16], symbols) for the symbols in `used`: fixed-length codes (the all-ones code stays free), or -- long_codes --
    the first eight symbols with 2..9 bits and all the others with 16 bits (codes beyond every decoder's lookup table)"""
    used = sorted(used)
    counts = [0] * 16
    if long_codes and len(used) > 8:
        for l in range(2, 10):
            counts[l - 1] = 1
        counts[15] = len(used) - 8
    else:
        length = max(1, int(np.ceil(np.log2(len(used) + 1))))
        counts[length - 1] = len(used)
    return counts, used


def encode_progressive(rgb, script, sampling=((2, 2), (1, 1), (1, 1)), quality_scale=1.0, gray=False, long_codes=False):
    """script: list of (components, Ss, Se, Ah, Al), components a tuple of indices (several only for DC scans).  Every scan gets its own
    Huffman table (written in front of it, slot 0 / 1 alternating so that slots are redefined between scans)."""
    h, w, sampling, hmax, vmax, mcus_x, mcus_y, qts, blocks = _quantised_blocks(rgb, sampling, quality_scale, gray)
    ncomp = len(blocks)
    out = bytearray(b"\xff\xd8")

    def seg(marker, payload):
        out.extend(bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload)

    for t, q in enumerate(qts if ncomp > 1 else qts[:1]):
        seg(0xDB, bytes([t]) + bytes(int(v) for v in q[ZIGZAG]))
    sof = bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([ncomp])
    for ci, (H, V) in enumerate(sampling):
        sof += bytes([ci + 1, (H << 4) | V, 0 if ci == 0 else 1])
    seg(0xC2, sof)

    def units_of(cis):
        if len(cis) == 1:  # the component's own grid (T.81 A.2.2)
            ci = cis[0]
            H, V = sampling[ci]
            bw = -(-(-(-w * H // hmax)) // 8) if ncomp > 1 else -(-w // 8)
            bh = -(-(-(-h * V // vmax)) // 8) if ncomp > 1 else -(-h // 8)
            return [(ci, by, bx) for by in range(bh) for bx in range(bw)]
        return [(ci, my * sampling[ci][1] + v, mx * sampling[ci][0] + hh) for my in range(mcus_y) for mx in range(mcus_x) for ci in cis
                for v in range(sampling[ci][1]) for hh in range(sampling[ci][0])]

    for scan_no, (cis, ss, se, ah, al) in enumerate(script):
        toks = []  # ('s', component, symbol) | ('b', value, nbits)
        if ss == 0:
            pred = [0] * ncomp
            for ci, by, bx in units_of(cis):
                c0 = int(blocks[ci][by, bx, 0])
                if ah == 0:
                    v = c0 >> al  # arithmetic shift: the DC point transform
                    diff = v - pred[ci]
                    pred[ci] = v
                    s = abs(diff).bit_length()
                    toks.append(("s", ci, s))
                    if s:
                        toks.append(("b", diff if diff >= 0 else diff + (1 << s) - 1, s))
                else:
                    toks.append(("b", (c0 >> al) & 1, 1))
        else:
            ci = cis[0]
            eobrun = 0
            be = []  # correction bits waiting behind the end-of-band run

            def flush_eobrun():
                nonlocal eobrun, be
                if eobrun:
                    n = eobrun.bit_length() - 1
                    toks.append(("s", ci, n << 4))
                    if n:
                        toks.append(("b", eobrun - (1 << n), n))
                    eobrun = 0
                for bit in be:
                    toks.append(("b", bit, 1))
                be = []

            for _, by, bx in units_of(cis):
                zz = blocks[ci][by, bx]
                mag = [abs(int(v)) >> al for v in zz]  # magnitudes are shifted, not the signed values
                if ah == 0:
                    r = 0
                    for k in range(ss, se + 1):
                        t = mag[k]
                        if t == 0:
                            r += 1
                            continue
                        flush_eobrun()
                        while r > 15:
                            toks.append(("s", ci, 0xF0))
                            r -= 16
                        s = t.bit_length()
                        toks.append(("s", ci, (r << 4) | s))
                        toks.append(("b", t if zz[k] >= 0 else (~t) & ((1 << s) - 1), s))
                        r = 0
                    if r > 0:
                        eobrun += 1
                        if eobrun == 0x7FFF:
                            flush_eobrun()
                else:
                    eob = 0
                    for k in range(ss, se + 1):
                        if mag[k] == 1:
                            eob = k
                    r = 0
                    br = []
                    for k in range(ss, se + 1):
                        t = mag[k]
                        if t == 0:
                            r += 1
                            continue
                        while r > 15 and k <= eob:
                            flush_eobrun()
                            toks.append(("s", ci, 0xF0))
                            r -= 16
                            toks.extend(("b", bit, 1) for bit in br)
                            br = []
                        if t > 1:
                            br.append(t & 1)
                            continue
                        flush_eobrun()
                        toks.append(("s", ci, (r << 4) | 1))
                        toks.append(("b", 0 if zz[k] < 0 else 1, 1))
                        toks.extend(("b", bit, 1) for bit in br)
                        br = []
                        r = 0
                    if r > 0 or br:
                        eobrun += 1
                        be.extend(br)
                        if eobrun == 0x7FFF or len(be) > 900:
                            flush_eobrun()
            flush_eobrun()
        # tables: one per component of the scan (DC first scans) or one (AC scans); none for DC refinement
        slot = scan_no & 1
        codes = {}
        sos = bytes([len(cis)])
        for n_c, ci in enumerate(cis):
            used = {t[2] for t in toks if t[0] == "s" and t[1] == ci}
            tsel = (slot + n_c) & 3
            if used:
                counts, symbols = _table_for(used, long_codes and (scan_no + ci) % 2 == 0)
                seg(0xC4, bytes([(0x10 if ss else 0x00) | tsel]) + bytes(counts) + bytes(symbols))
                codes[ci] = _codes(counts, symbols)
            sos += bytes([ci + 1, (tsel << 4) if ss == 0 else tsel])
        seg(0xDA, sos + bytes([ss, se, (ah << 4) | al]))
        bits = _Bits()
        for t in toks:
            if t[0] == "s":
                bits.put(*codes[t[1]][t[2]])
            else:
                bits.put(t[1], t[2])
        bits.flush()
        out.extend(bits.out)
    out.extend(b"\xff\xd9")
    return bytes(out)


# scan scripts for encode_progressive (three components unless noted)
SCRIPT_LIBJPEG = [((0, 1, 2), 0, 0, 0, 1), ((0,), 1, 5, 0, 2), ((2,), 1, 63, 0, 1), ((1,), 1, 63, 0, 1), ((0,), 6, 63, 0, 2), ((0,), 1, 63, 2, 1),
                  ((0, 1, 2), 0, 0, 1, 0), ((2,), 1, 63, 1, 0), ((1,), 1, 63, 1, 0), ((0,), 1, 63, 1, 0)]
SCRIPT_SPECTRAL_ONLY = [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)]
SCRIPT_DEEP = [((0,), 0, 0, 0, 3), ((1,), 0, 0, 0, 2), ((2,), 0, 0, 0, 2), ((0,), 1, 2, 0, 3), ((0,), 3, 9, 0, 3), ((0,), 10, 63, 0, 3), ((0,), 1, 63, 3, 2),
               ((0,), 0, 0, 3, 2), ((0,), 1, 63, 2, 1), ((1,), 1, 63, 0, 2), ((1,), 1, 63, 2, 1), ((0,), 0, 0, 2, 1), ((1, 2), 0, 0, 2, 1), ((2,), 1, 63, 0, 0),
               ((0,), 1, 63, 1, 0), ((0, 1, 2), 0, 0, 1, 0), ((1,), 1, 63, 1, 0)]
SCRIPT_REFINE_BEFORE_OTHER_BANDS = [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 8, 0, 1), ((0,), 1, 8, 1, 0), ((0,), 9, 63, 0, 2), ((1,), 1, 63, 0, 1), ((0,), 9, 63, 2, 1),
                                    ((1,), 1, 63, 1, 0), ((2,), 1, 30, 0, 0), ((0,), 9, 63, 1, 0), ((2,), 31, 63, 0, 1), ((2,), 31, 63, 1, 0)]
SCRIPT_MANY_BANDS = [((0, 1, 2), 0, 0, 0, 0)] + [((c,), a, min(a + 2, 63), 0, 0) for a in range(1, 64, 3) for c in (0, 1, 2)]
SCRIPT_GRAY = [((0,), 0, 0, 0, 1), ((0,), 1, 63, 0, 2), ((0,), 1, 63, 2, 1), ((0,), 0, 0, 1, 0), ((0,), 1, 63, 1, 0)]
