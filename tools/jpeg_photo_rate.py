#!/usr/bin/env python3
"""tools/jpeg_photo_rate.py [--n 20000] -- the JPEG path on photo-sized files: tests/golden/bench.jpg (1280x854, the reference's own bench image)
re-coded as baseline 4:2:0 quality 90 (what a camera writes) and as it is (progressive), n copies per call.  These go through the
> 512 px pre-downsample and the generic PDQ kernels (no 512x512 fast path)."""
import argparse
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--seg", default="", help="comma list of segment sizes (bytes) to try on the files without restart markers; 0 = one lane per file")
    a = ap.parse_args()
    from PIL import Image

    from rupphash_amd import Engine

    eng = Engine(0)
    orig = open(os.path.join(ROOT, "tests", "golden", "bench.jpg"), "rb").read()
    im = Image.open(io.BytesIO(orig))
    variants = []
    for k in range(16):  # 16 distinct files: crops of a few pixels change every block
        buf = io.BytesIO()
        im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(buf, "JPEG", quality=90, subsampling=2)
        variants.append(buf.getvalue())
    with_rst = []
    for k in range(16):  # the same pictures with a restart interval per MCU row (what many cameras write): every interval gets a lane
        buf = io.BytesIO()
        im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(buf, "JPEG", quality=90, subsampling=2, restart_marker_rows=1)
        with_rst.append(buf.getvalue())
    prog = []
    for k in range(16):  # and as progressive files (what the web serves): ten scans over every block, one lane per file
        buf = io.BytesIO()
        im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(buf, "JPEG", quality=90, subsampling=2, progressive=True)
        prog.append(buf.getvalue())
    t = time.perf_counter()
    for f in variants:
        np.asarray(Image.open(io.BytesIO(f)))
    dec = (time.perf_counter() - t) / len(variants)
    print(f"libjpeg-turbo (Pillow) full decode, 1 thread: {1 / dec:7.0f} files/s ({dec * 1e3:.2f} ms per ~1265x850 file of {len(variants[0]) / 1e3:.0f} KB)")
    for label, base, mode in [("baseline 4:2:0 q90, device entropy", variants, 1), ("same with restart intervals, device entropy", with_rst, 1),
                              ("same with restart intervals, device, n = 2048", with_rst, 2), ("baseline 4:2:0 q90, device entropy, n = 2048", variants, 2),
                              ("baseline 4:2:0 q90, host entropy", variants, 0), ("bench.jpg as it is (progressive), host entropy", [orig], 0),
                              ("progressive 4:2:0 q90, device entropy", prog, 1), ("progressive 4:2:0 q90, device, n = 2048", prog, 2)]:
        n = a.n if mode == 1 else (2048 if mode == 2 else min(a.n, 4000))
        files = eng.jpeg_file_list([base[k % len(base)] for k in range(n)])
        eng.jpeg_set_entropy(min(mode, 1))
        eng.jpeg_pdq_hash_batch(files, threads=16)
        t = time.perf_counter()
        out = eng.jpeg_pdq_hash_batch(files, threads=16)
        dt = time.perf_counter() - t
        assert out["valid"].all()
        mb = sum(len(base[k % len(base)]) for k in range(n)) / 1e6
        print(f"{label:50s} n={n:6d}: {n / dt:8.0f} files/s  {mb / dt:7.1f} MB/s of JPEG  {n * 1280 * 854 / dt / 1e9:6.2f} Gpixel/s")
    for seg in [int(x) for x in a.seg.split(",") if x]:
        eng.jpeg_set_entropy(1)
        eng.jpeg_set_segments(65536, seg)
        for n in (a.n, 2048):
            files = eng.jpeg_file_list([variants[k % 16] for k in range(n)])
            eng.jpeg_pdq_hash_batch(files, threads=16)
            t = time.perf_counter()
            out = eng.jpeg_pdq_hash_batch(files, threads=16)
            dt = time.perf_counter() - t
            assert out["valid"].all()
            print(f"baseline 4:2:0 q90, segments of {seg:5d} bytes           n={n:6d}: {n / dt:8.0f} files/s")
    eng.close()


if __name__ == "__main__":
    main()
