#!/usr/bin/env python3
"""tools/jpeg_auto_rate.py -- does the automatic choice between host and device entropy decoding (rph_jpeg_set_entropy 2) pick the quicker
one?  Photo-sized files (tests/golden/bench.jpg re-coded: baseline without restart markers, and progressive), n files per call:
files/s with the host threads, with the device walk, and with the automatic setting."""
import io
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from PIL import Image

from rupphash_amd import Engine

eng = Engine(0)
im = Image.open(os.path.join(ROOT, "tests", "golden", "bench.jpg"))
kinds = {}
for label, kw in (("baseline", {}), ("progressive", {"progressive": True})):
    v = []
    for k in range(16):
        buf = io.BytesIO()
        im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(buf, "JPEG", quality=90, subsampling=2, **kw)
        v.append(buf.getvalue())
    kinds[label] = v
small = []
imgs = eng.synth_images(0, 16)
for label, kw in (("512x512 baseline", {}), ("512x512 progressive", {"progressive": True})):
    v = []
    for k in range(16):
        buf = io.BytesIO()
        Image.fromarray(imgs[k]).save(buf, "JPEG", quality=85, subsampling=2, **kw)
        v.append(buf.getvalue())
    kinds[label] = v
for label, v in kinds.items():
    for n in (8, 32, 64, 128, 512, 2048, 8192):
        files = eng.jpeg_file_list([v[k % 16] for k in range(n)])
        row = []
        for mode in (0, 1, 2):
            eng.jpeg_set_entropy(mode)
            eng.jpeg_pdq_hash_batch(files, threads=16)
            t = time.perf_counter()
            reps = 3 if n <= 512 else 1
            for _ in range(reps):
                eng.jpeg_pdq_hash_batch(files, threads=16)
            row.append(n * reps / (time.perf_counter() - t))
        print(f"{label:22s} n={n:5d}: host {row[0]:9.0f}  device {row[1]:9.0f}  automatic {row[2]:9.0f} files/s   {'ok' if row[2] > 0.8 * max(row[0], row[1]) else 'WRONG CHOICE'}")
eng.jpeg_set_entropy(2)
eng.close()
