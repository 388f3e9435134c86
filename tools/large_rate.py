#!/usr/bin/env python3
"""tools/large_rate.py -- images/s of rph_pdq_hash_batch_dev for geometries off the 512x512 fast path, pixels resident in HBM:
photo-sized sources (> 512 px: luma -> box pre-downsample -> PDQ on the thumbnail, pdqhash.rs:181-220) and thumbnails themselves
(<= 512 px: the multi-pass kernels).  Default kernels against the plain baseline (rph_pdq_set_kernel 0); same bits asserted."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rupphash_amd import Engine

eng = Engine(0)
rng = np.random.default_rng(5)
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1  # index of the one geometry to run (profiling)
for case, (w, h, ch, n) in enumerate([(1265, 850, 1, 2000), (1265, 850, 3, 2000), (4000, 3000, 3, 200), (512, 344, 1, 32000), (512, 344, 3, 8000), (344, 512, 1, 32000), (512, 288, 1, 32000), (384, 512, 1, 32000),
                      (256, 256, 3, 20000)]):
    if only >= 0 and case != only:
        continue
    base = rng.integers(0, 256, (8, h, w, ch) if ch > 1 else (8, h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    g = ((xx * 200) // (w - 1) + (yy * 55) // (h - 1)).astype(np.uint8)
    for k in range(8):
        if ch > 1:
            base[k, ..., k % ch] = np.roll(g, 37 * k, axis=1)
        else:
            base[k] = (base[k] // 4 + np.roll(g, 37 * k, axis=1) // 2).astype(np.uint8)
    per = w * h * ch
    d_px = eng.dev_alloc(n * per)
    for k in range(0, n, 8):
        m = min(8, n - k)
        eng.dev_upload(d_px + k * per, base[:m])
    d_hash = eng.dev_alloc(n * 32)
    d_q = eng.dev_alloc(n * 4)
    got = {}
    for which in (0, 5, 4):  # plain multi-pass, multi-pass through LDS tiles, default (streaming single-pass kernel where it applies)
        eng.set_pdq_kernel(which)
        eng.pdq_hash_batch_dev(d_px, n, w, h, ch, d_hash, d_q)
        eng.synchronize()
        t = time.perf_counter()
        for rep in range(3):
            eng.pdq_hash_batch_dev(d_px, n, w, h, ch, d_hash, d_q)
        eng.synchronize()
        dt = (time.perf_counter() - t) / 3
        hs = np.zeros((n, 32), np.uint8)
        eng.dev_download(hs, d_hash)
        got[which] = hs
        print(f"{w:5d} x {h:4d} x {ch}  n={n:6d}  kernel {which}: {n / dt:10.0f} images/s  {dt / n * 1e6:7.2f} us per image  {n * per / dt / 1e9:8.1f} GB/s of pixels")
    assert os.environ.get("RPH_NO_CHECK") or (np.array_equal(got[0], got[4]) and np.array_equal(got[5], got[4]))
    eng.dev_free(d_px)
    eng.dev_free(d_hash)
    eng.dev_free(d_q)
eng.set_pdq_kernel(4)
eng.close()
