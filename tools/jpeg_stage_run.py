#!/usr/bin/env python3
"""tools/jpeg_stage_run.py small|smallprog|photos|prog [calls] -- the JPEG leg of bench.py as a bare loop of `calls` rph_jpeg_pdq_hash_batch calls, to be run
under rocprofv3 --kernel-trace --stats (tools/jpeg_stage_profile.sh): every kernel launch of the run belongs to the one workload, so a stage's
time per call is its TotalDurationNs / calls.  Prints one JSON line with the counts the per-stage rooflines are priced with."""
import io
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from PIL import Image

from rupphash_amd import Engine

kind = sys.argv[1] if len(sys.argv) > 1 else "small"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = Engine(0)
threads = min(16, len(os.sched_getaffinity(0)))


def enc(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, "JPEG", **kw)
    return b.getvalue()


if kind in ("small", "smallprog"):
    distinct, n, w, h = 4096, 100_000, 512, 512
    base = []
    with ThreadPoolExecutor(threads) as pool:
        for first in range(0, distinct, 256):
            base += list(pool.map(lambda a: enc(a, quality=85, subsampling=2, progressive=(kind == "smallprog")), eng.synth_images(first, 256)))
else:
    distinct, n, w, h = 64, 20_000, 1265, 850
    im = Image.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bench.jpg"))
    im.load()  # (the threads below crop it: decode once, here)
    with ThreadPoolExecutor(threads) as pool:
        base = list(pool.map(lambda k: enc(np.asarray(im.crop((k % 16, k // 16, k % 16 + 1265, k // 16 + 850))), quality=90, subsampling=2, progressive=(kind == "prog")), range(distinct)))
files = eng.jpeg_file_list([base[k % distinct] for k in range(n)])
eng.jpeg_set_entropy(1)
if os.environ.get("RPH_SEG_MIN"):  # experiments: segments from that many entropy bytes on, of RPH_SEG_BYTES each
    eng.jpeg_set_segments(int(os.environ["RPH_SEG_MIN"]), int(os.environ.get("RPH_SEG_BYTES", "1024")))
eng.jpeg_pdq_hash_batch(files, threads=threads)  # buffers
t0 = time.perf_counter()
for _ in range(calls):
    out = eng.jpeg_pdq_hash_batch(files, threads=threads)
dt = (time.perf_counter() - t0) / calls
assert out["valid"].all()
mcu_w, mcu_h = (w + 15) // 16, (h + 15) // 16
print(json.dumps({"kind": kind, "calls_timed": calls, "calls_total": calls + 1, "files_per_call": n, "distinct_files": distinct, "geometry": [w, h],
                  "seconds_per_call": dt, "files_per_s": n / dt, "file_bytes_per_call": sum(len(base[k % distinct]) for k in range(n)),
                  "blocks_per_file": mcu_w * mcu_h * 6, "pixels_per_file": w * h, "padded_pixels_per_file": mcu_w * 16 * mcu_h * 16, "host_threads": threads}))
eng.close()
