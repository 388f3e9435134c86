"""Throughput of the one-image-per-call entry point (rph_pdq_hash_one) from T caller threads: what a scanner that
hashes one decoded file per worker call sees, PCIe and staging copies included; and of the host-pointer batch entry point
(rph_pdq_hash_batch) on a large pageable array.  Run on the GPU box."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd.engine import Engine

eng = Engine(0)
rng = np.random.default_rng(1)
imgs = rng.integers(0, 256, (64, 512, 512, 3), dtype=np.uint8)
big = rng.integers(0, 256, (8, 854, 1280, 3), dtype=np.uint8)   # the reference's bench.jpg geometry (pre-downsample path)
mixed = [imgs[k] if k % 3 else big[k % 8] for k in range(64)]


def run(label, pool, threads, total):
    per = max(4, total // threads)
    def work(t):
        for k in range(per):
            eng.pdq_hash_one(pool[(t + k) % len(pool)], want_coeffs=False)
    work(0)
    b0, i0 = eng.pdq_batcher_stats()
    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    b1, i1 = eng.pdq_batcher_stats()
    mb = sum(pool[(t + k) % len(pool)].nbytes for t in range(threads) for k in range(per)) / 1e6
    print(f"{label:28s} threads={threads:4d}: {threads*per/dt:10.0f} hashes/s  {mb/dt/1e3:6.1f} GB/s over PCIe, {(i1-i0)/(b1-b0):6.1f} images/batch", flush=True)


eng.pdq_batcher_config()  # defaults: max_batch 256, no linger
for threads in (1, 4, 16, 32, 64, 256):
    run("512x512 RGB8", imgs, threads, 8192)
for threads in (1, 16, 64):
    run("1280x854 RGB8 (resized path)", big, threads, 2048)
for threads in (16, 64):
    run("mixed 512x512 / 1280x854", mixed, threads, 4096)

# host-pointer batch entry point on a pageable array
n = 8192
arr = np.ascontiguousarray(np.broadcast_to(imgs[None], (n // 64, 64, 512, 512, 3)).reshape(n, 512, 512, 3))
eng.pdq_hash_batch(arr[:256], want_quality=False)
t0 = time.perf_counter()
out = eng.pdq_hash_batch(arr, want_quality=False)
dt = time.perf_counter() - t0
print(f"rph_pdq_hash_batch, {n} images from pageable host memory: {n/dt:10.0f} hashes/s  {arr.nbytes/dt/1e9:6.1f} GB/s", flush=True)
assert (out["hash"][:64] == out["hash"][64:128]).all()
