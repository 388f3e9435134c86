"""Differential fuzz of the production grouping (8 dihedral variants, low-quality rule, files without features, union-find) and of
find_groups (reference reachability + member order): every sweep formulation must give the same groups and comparison counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rupphash_amd.engine import Engine

eng = Engine(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    n = int(rng.choice([2, 7, 100, 1023, 1025, int(rng.integers(2, 6000))]))
    sim = int(rng.choice([0, 5, 15, 16, 31, 32, 40, 47, 48, 63]))
    coeffs = rng.normal(0, 25, (n, 256)).astype(np.float32)
    for _ in range(int(rng.integers(0, 30))):  # near-duplicate feature vectors, some of them dihedral copies
        a, b = rng.integers(0, n, 2)
        c = coeffs[a].reshape(16, 16).copy()
        t = int(rng.integers(0, 4))
        if t == 1:
            c = c.T.copy()
        if t == 2:
            c = c * np.where((np.arange(16) % 2 == 0)[None, :], -1, 1)
        coeffs[b] = (c + rng.normal(0, 0.3, (16, 16))).reshape(256)
    hashes, _ = eng.pdq_hashes_from_coeffs(coeffs, want_hash=True, want_dihedral=False)
    has = (rng.random(n) > 0.2).astype(np.uint8)
    qual = rng.integers(-1, 101, n).astype(np.int32)
    res = []
    for kern in (0, 1, 2):
        eng.set_hamming_kernel(kern)
        g, cnt = eng.group_files_pdq(hashes, sim, coeffs=coeffs, has_features=has, quality=qual)
        fg = eng.find_groups256(hashes, min(sim, 63))
        res.append((g, cnt, fg))
    assert res[0] == res[1] == res[2], (n, sim)
    cases += 1
eng.set_hamming_kernel(2)
print(f"seed {seed}: {cases} random grouping cases, all three sweep formulations agree")
