"""Differential fuzz of the Hamming sweeps: random sizes, thresholds, cluster structures, duplicates and part counts; the fp4 and int8
MFMA kernels must return exactly the VALU kernel's edge multiset (distance and flags included); variant sweeps likewise."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd.engine import Engine

eng = Engine(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)


def key(e):
    a = np.stack([e["i"].astype(np.int64), e["j"].astype(np.int64), e["d"].astype(np.int64), e["flags"].astype(np.int64)], axis=1)
    return a[np.lexsort(a.T[::-1])]


t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    n = int(rng.choice([2, 3, 31, 33, 1023, 1024, 1025, 2047, 2049, int(rng.integers(2, 9000)), int(rng.integers(9000, 40000))]))
    thr = int(rng.choice([0, 1, 5, 16, 31, 32, 33, 40, 41, 53, 54, 63, 66, 67, 80, 81, int(rng.integers(0, 130))]))
    h = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    style = rng.integers(0, 4)
    if style >= 1:  # clusters of near duplicates
        for _ in range(int(rng.integers(1, 40))):
            base = rng.integers(0, 256, 32, dtype=np.uint8)
            for j in rng.choice(n, min(n, int(rng.integers(2, 9))), replace=False):
                v = base.copy()
                for b in rng.choice(256, int(rng.integers(0, min(2 * thr + 4, 120))), replace=False):
                    v[b >> 3] ^= 1 << (b & 7)
                h[j] = v
    if style == 3 and n >= 64:  # a block of exact duplicates: floods the candidate queue
        h[: min(n, int(rng.integers(64, 700)))] = h[0]
    nparts = int(rng.choice([1, 1, 2, 3, 5]))
    res = {}
    t_case = time.time()
    if style == 2:  # skewed bit densities: the popcount-sorted {0,1} form thresholds on popcounts
        dens = rng.choice([0.03, 0.2, 0.5, 0.8, 0.97], n)
        h = np.packbits((rng.random((n, 256)) < dens[:, None]).astype(np.uint8), axis=1)
        for j in range(1, n, 7):
            v = h[j - 1].copy()
            for b in rng.choice(256, int(rng.integers(0, min(thr + 3, 200))), replace=False):
                v[b >> 3] ^= 1 << (b & 7)
            h[j] = v
    for kern in (0, 1, 3, 4):  # VALU, int8 MFMA, fp4 MFMA +-1, fp4 MFMA popcount-sorted {0,1}
        eng.set_hamming_kernel(kern)
        parts = [eng.hamming_all_pairs(h, thr, part=p, nparts=nparts, cap=max(1 << 16, 4 * n)) for p in range(nparts)]
        res[kern] = key(np.concatenate(parts))
    assert all(res[0].shape == res[k].shape and (res[0] == res[k]).all() for k in (1, 3, 4)), (n, thr, style, nparts)
    cases += 1
    if time.time() - t_case > 5.0:  # (edge-heavy cases spend their time in the host-side sort of the comparison)
        print(f"  case {cases - 1}: n {n} thr {thr} style {int(style)} nparts {nparts}: {len(res[0])} edges, {time.time() - t_case:.1f} s", flush=True)
eng.set_hamming_kernel(2)
print(f"seed {seed}: {cases} random cases, all four sweep formulations agree edge for edge")
