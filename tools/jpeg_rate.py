#!/usr/bin/env python3
"""tools/jpeg_rate.py [--n 4000] [--size 512] -- files/s of the JPEG path (rph_jpeg_pdq_hash_batch: host entropy decode -> device IDCT /
upsample / colour -> PDQ) by host thread count, next to what the host alone does with the same files:
  * libjpeg-turbo (Pillow) full decode, one thread (the usual CPU decoder; SIMD),
  * the product's host half alone (entropy decoding to coefficients), one thread.
Files: synthetic images of the bench (tools share bench.py's generator) JPEG-coded by Pillow at quality 85, 4:2:0, baseline and progressive."""
import argparse
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4000)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--device-n", type=int, default=100000, help="files per call for the device-entropy rows")
    a = ap.parse_args()
    from PIL import Image

    from rupphash_amd import Engine

    eng = Engine(0)
    imgs = eng.synth_images(0, a.distinct, a.size, a.size)
    for label, kw in [("baseline 4:2:0 q85", dict(quality=85, subsampling=2)), ("progressive 4:2:0 q85", dict(quality=85, subsampling=2, progressive=True)),
                      ("baseline 4:4:4 q95", dict(quality=95, subsampling=0))]:
        base = []
        for k in range(a.distinct):
            buf = io.BytesIO()
            Image.fromarray(imgs[k]).save(buf, "JPEG", **kw)
            base.append(buf.getvalue())
        files = [base[k % a.distinct] for k in range(a.n)]
        mb = sum(len(f) for f in files) / 1e6
        print(f"== {label}: {a.n} files of {a.size}x{a.size}, {mb / a.n * 1e3:.1f} KB each")
        t = time.perf_counter()
        for f in base:
            np.asarray(Image.open(io.BytesIO(f)))
        dt = (time.perf_counter() - t) / len(base)
        print(f"   libjpeg-turbo (Pillow) full decode, 1 thread: {1 / dt:8.0f} files/s  ({dt * 1e3:.3f} ms)")
        t = time.perf_counter()
        for f in base:
            eng.jpeg_coefficients(f)
        dt = (time.perf_counter() - t) / len(base) / 2  # the wrapper decodes twice (size query + data)
        print(f"   host half alone (entropy decode), 1 thread:   {1 / dt:8.0f} files/s  ({dt * 1e3:.3f} ms)")
        eng.jpeg_set_entropy(0)
        ref_hash = eng.jpeg_pdq_hash_batch(files[:256], threads=8)["hash"]  # warm-up: staging buffers
        for t_n in [int(x) for x in a.threads.split(",")]:
            t = time.perf_counter()
            out = eng.jpeg_pdq_hash_batch(files, threads=t_n, want_quality=True)
            dt = time.perf_counter() - t
            assert out["valid"].all()
            print(f"   host entropy   threads={t_n:3d} n={a.n:6d}: {a.n / dt:9.0f} files/s  {mb / dt:7.1f} MB/s of JPEG  {a.n * a.size * a.size * 3 / dt / 1e9:6.2f} GB/s of pixels")
        eng.jpeg_set_entropy(1)
        for n_dev in sorted({a.n, min(a.device_n, 20000), a.device_n}):
            big = [base[k % a.distinct] for k in range(n_dev)]
            for rep in range(2):
                t = time.perf_counter()
                out = eng.jpeg_pdq_hash_batch(big, threads=16, want_quality=True)
                dt = time.perf_counter() - t
                assert out["valid"].all()
            assert np.array_equal(out["hash"][: a.distinct], ref_hash[: a.distinct])
            bmb = sum(len(f) for f in big) / 1e6
            print(f"   device entropy threads= 16 n={n_dev:6d}: {n_dev / dt:9.0f} files/s  {bmb / dt:7.1f} MB/s of JPEG  {n_dev * a.size * a.size * 3 / dt / 1e9:6.2f} GB/s of pixels")
        eng.jpeg_set_entropy(0)
    eng.close()


if __name__ == "__main__":
    main()
