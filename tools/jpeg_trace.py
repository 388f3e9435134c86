"""tools/jpeg_trace.py [--photos] [--n N] [--rst] -- baseline JPEG files through rph_jpeg_pdq_hash_batch with the Huffman streams walked on the
device, three calls: 100 000 files of 512x512 (default), or --photos: 20 000 re-coded copies of tests/golden/bench.jpg (1265x850, no restart
markers unless --rst).  RPH_JPEG_TRACE=1: synchronise after every device phase and print its time; RPH_JPEG_TRACE=2: host-side timestamps of
the chunk pipeline."""
import argparse
import io
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from PIL import Image

from rupphash_amd import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--photos", action="store_true")
ap.add_argument("--rst", action="store_true")
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
eng = Engine(0)
base = []
if a.photos:
    im = Image.open(os.path.join(ROOT, "tests", "golden", "bench.jpg"))
    for k in range(16):
        b = io.BytesIO()
        kw = dict(restart_marker_rows=1) if a.rst else {}
        im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(b, "JPEG", quality=90, subsampling=2, **kw)
        base.append(b.getvalue())
else:
    imgs = eng.synth_images(0, 64, 512, 512)
    for k in range(64):
        b = io.BytesIO()
        Image.fromarray(imgs[k]).save(b, "JPEG", quality=85, subsampling=2)
        base.append(b.getvalue())
n = a.n or (20000 if a.photos else 100000)
files = eng.jpeg_file_list([base[k % len(base)] for k in range(n)])
eng.jpeg_set_entropy(1)
for rep in range(a.reps):
    t = time.perf_counter()
    out = eng.jpeg_pdq_hash_batch(files, threads=16)
    dt = time.perf_counter() - t
    print("total %.1f ms -> %.0f files/s" % (dt * 1e3, n / dt))
eng.close()
