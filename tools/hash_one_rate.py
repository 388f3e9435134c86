"""Throughput of the one-image-per-call entry point (rph_pdq_hash_one) from T caller threads: what a scanner that
hashes one decoded file per worker call sees, PCIe and staging copies included.  Run on the GPU box."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd.engine import Engine

eng = Engine(0)
rng = np.random.default_rng(1)
imgs = rng.integers(0, 256, (64, 512, 512, 3), dtype=np.uint8)
for threads, max_batch, wait_us in ((1, 1, 0), (16, 256, 500), (64, 256, 500), (256, 256, 1000)):
    eng.pdq_batcher_config(max_batch, wait_us)
    per = max(4, 4096 // threads)
    def work(t):
        for k in range(per):
            eng.pdq_hash_one(imgs[(t + k) % 64], want_coeffs=False)
    work(0)
    b0, i0 = eng.pdq_batcher_stats()
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    b1, i1 = eng.pdq_batcher_stats()
    print(f"threads={threads:4d} max_batch={max_batch:4d} wait_us={wait_us:5d}: {threads*per/dt:10.0f} hashes/s, "
          f"{(i1-i0)/(b1-b0):6.1f} images/batch", flush=True)
