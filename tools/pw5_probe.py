#!/usr/bin/env python3
"""tools/pw5_probe.py -- what would a 160-bit prefix (PW = 5: a third fp4 MFMA whose upper half is empty) buy for thresholds 42..54?
The popcount-sorted {0,1} sweep encodes a 0 bit as the fp4 code 0x0, so hashes whose dword 5 (bits 160..191) is zero make the third MFMA of
the PW = 6 kernel exactly such a half-empty instruction: same issue slots, half the non-zero products.  Timing only (the edge set differs)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rupphash_amd import Engine

eng = Engine(0)
n = 1_000_000
rng = np.random.default_rng(1)
base = rng.integers(0, 256, (n, 32), dtype=np.uint8)
cap = 1 << 20
d_e = eng.dev_alloc(cap * 12)
d_c = eng.dev_alloc(8)
d_h = eng.dev_alloc(n * 32)
for label, zero_dwords in (("all 256 bits random", ()), ("dword 5 zero (half-empty third MFMA)", (5,)), ("dwords 4 and 5 zero (empty third MFMA)", (4, 5))):
    h = base.copy()
    for z in zero_dwords:
        h[:, 4 * z:4 * z + 4] = 0
    eng.dev_upload(d_h, h)
    for thr in (32, 48, 63):
        res = []
        for rep in range(4):
            eng.dev_memset(d_c, 0, 8)
            eng.synchronize()
            t = time.perf_counter()
            eng.hamming_all_pairs_dev(d_h, n, thr, d_e, cap, d_c)
            eng.synchronize()
            res.append(time.perf_counter() - t)
        best = min(res[1:])
        print(f"{label:45s} threshold {thr:2d}: {best * 1e3:7.2f} ms  {n * (n - 1) / 2 / best / 1e12:6.2f} Tpairs/s  (PW {eng.L.rph_hamming_prefix_dwords(thr, 2)})")
eng.close()
