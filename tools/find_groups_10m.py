import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd.engine import Engine
eng = Engine(0)
n = 10_000_000
d_h = eng.dev_alloc(n * 32)
eng.synth_hashes_dev(d_h, 0, n, n, n_clusters=1000)
eng.synchronize()
h = np.zeros((n, 32), np.uint8)
eng.dev_download(h, d_h)
eng.dev_free(d_h)
t0 = time.perf_counter()
groups = eng.find_groups256(h, 32)
t1 = time.perf_counter()
sizes = sorted(len(g) for g in groups)
print(f"find_groups256 on {n} hashes, max_dist 32: {t1-t0:.2f} s, {len(groups)} groups, sizes {sizes[0]}..{sizes[-1]}")
t0 = time.perf_counter()
g2, cmp_count = eng.group_files_pdq(h, 32)
t1 = time.perf_counter()
print(f"group_files_pdq (no features, union-find) on {n}: {t1-t0:.2f} s, {len(g2)} groups, comparisons {cmp_count}")
