#!/usr/bin/env python3
"""tools/jpeg_one_rate.py -- rph_jpeg_pdq_hash_one (one JPEG file per blocking call) from T threads, the reference's scan-loop pattern
(load_image_fast + generate_pdq_features on every rayon worker), for 512x512 files and for photo-sized files (1265x850).  The GIL is
released inside the C call, so Python threads are real concurrent callers here."""
import io
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    from PIL import Image

    from rupphash_amd import Engine

    eng = Engine(0)
    imgs = eng.synth_images(0, 32)
    small = []
    for k in range(32):
        buf = io.BytesIO()
        Image.fromarray(imgs[k]).save(buf, "JPEG", quality=85, subsampling=2)
        small.append(buf.getvalue())
    im = Image.open(os.path.join(ROOT, "tests", "golden", "bench.jpg"))
    photo = []
    for k in range(16):
        buf = io.BytesIO()
        im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(buf, "JPEG", quality=90, subsampling=2)
        photo.append(buf.getvalue())
    for label, files, per_thread in [("512x512 baseline 4:2:0 q85 (29 KB)", small, 400), ("1265x850 baseline 4:2:0 q90 (366 KB)", photo, 60)]:
        for t_n in (1, 4, 8, 16, 32):
            def worker(t):
                for i in range(per_thread):
                    eng.jpeg_pdq_hash_one(files[(t + i) % len(files)], want_coeffs=True)
            th = [threading.Thread(target=worker, args=(t,)) for t in range(t_n)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            print(f"{label:40s} threads={t_n:3d}: {t_n * per_thread / dt:8.0f} files/s")
    eng.close()


if __name__ == "__main__":
    main()
