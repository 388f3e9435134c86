"""Repeat the two hot kernels under sustained load and check that every repetition returns the same bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd import EDGE_DTYPE
from rupphash_amd.engine import Engine

eng = Engine(0)
n = 30_000
d_px = eng.dev_alloc(n * 786432)
eng.synth_images_dev(d_px, 0, n, 512, 512)
d_h = eng.dev_alloc(n * 32)
ref = None
for it in range(150):
    eng.dev_memset(d_h, 0, n * 32)
    eng.pdq_hash_batch_dev(d_px, n, 512, 512, 3, d_h)
    eng.synchronize()
    h = np.zeros((n, 32), np.uint8)
    eng.dev_download(h, d_h)
    if ref is None:
        ref = h
    assert np.array_equal(h, ref), f"PDQ repetition {it} differs"
print("PDQ: 150 repetitions of 30000 images identical")
eng.dev_free(d_px)
m = 1_000_000
d_hh = eng.dev_alloc(m * 32)
eng.synth_hashes_dev(d_hh, 0, m, m, n_clusters=1000)
cap = 1 << 16
d_e, d_c = eng.dev_alloc(cap * 12), eng.dev_alloc(8)
ref = None
for it in range(60):
    eng.dev_memset(d_c, 0, 8)
    eng.hamming_all_pairs_dev(d_hh, m, 32, d_e, cap, d_c)
    eng.synchronize()
    cnt = np.zeros(1, np.uint64)
    eng.dev_download(cnt, d_c)
    e = np.zeros(int(cnt[0]), EDGE_DTYPE)
    eng.dev_download(e, d_e)
    key = np.sort(e.view(np.dtype((np.void, 12))).ravel())
    if ref is None:
        ref = key
    assert len(key) == len(ref) and np.array_equal(key, ref), f"Hamming repetition {it} differs"
print(f"Hamming: 60 repetitions over 1M hashes identical ({len(ref)} edges)")
