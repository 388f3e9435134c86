"""Pins oracle/hamming_ref.c with the reference's own tests (src/hamminghash.rs:273-412,
NOTES.txt:64-67) and cross-checks its two grouping semantics against brute force."""
import numpy as np
import pytest

KIND_U64, KIND_PDQ = 0, 1


def test_high_similarity_support_u64(oracle):
    """hamminghash.rs:286-307: {0, 0xFFF} at distance 12 -> one group [0, 1]."""
    assert bin(0xFFF).count("1") == 12
    groups = oracle.find_groups(KIND_U64, np.array([0, 0xFFF], np.uint64), 12)
    assert groups and set(groups[0]) == {0, 1} and len(groups[0]) == 2
    assert groups == [[0, 1]]  # NOTES.txt:13 prints "Found: [[0, 1]]"


def test_high_similarity_support_pdq(oracle):
    """hamminghash.rs:310-331: first 30 bits set, distance 30 -> group contains 0 and 1."""
    base = np.zeros(32, np.uint8)
    target = np.zeros(32, np.uint8)
    for i in range(30):
        target[i // 8] |= 1 << (i % 8)
    assert oracle.hamming256(base, target) == 30
    groups = oracle.find_groups(KIND_PDQ, np.stack([base, target]), 30)
    assert groups == [[0, 1]]  # NOTES.txt:15


def test_injected_cluster_among_random_u64(oracle):
    """hamminghash.rs:336-412 at 200k (the 1M original runs in bench.py's cpu_baseline leg)."""
    n = 200_000
    rng = np.random.default_rng(12345)
    hashes = rng.integers(0, 2**64, n, dtype=np.uint64)
    target = 0xABCD_1234_5678_90EF
    cluster = [target, target ^ 1, target ^ 2, target ^ 0x8000, target ^ 0x8001]
    idx = rng.choice(n, 5, replace=False)
    for v, i in zip(cluster, idx):
        hashes[i] = v
    groups = oracle.find_groups(KIND_U64, hashes, 5)
    g = [g for g in groups if int(idx[0]) in g]
    assert g, "The injected images were not found in any group!"
    assert set(int(i) for i in idx) <= set(g[0])


def test_phash_bitops_known_answer(oracle):
    """NOTES.txt:64-67: pHash deb1e20c136f983c -> rotation invariant 8b1bb7a646c5cd96."""
    h = 0xDEB1E20C136F983C
    assert oracle.rotation_invariant_hash(h) == 0x8B1BB7A646C5CD96
    r90, r180, r270 = oracle.rotate_hash_90(h), oracle.rotate_hash_180(h), oracle.rotate_hash_270(h)
    assert min(h, r90, r180, r270) == 0x8B1BB7A646C5CD96
    # group structure of the bit ops themselves (phash.rs:150-230)
    assert oracle.rotate_hash_180(r180) == h
    assert oracle.rotate_hash_270(r90) == h and oracle.rotate_hash_90(r270) == h
    assert oracle.rotate_hash_90(r90) == r180
    assert oracle.flip_hash_horizontal(oracle.flip_hash_horizontal(h)) == h
    d = oracle.phash_dihedral(h)
    hf = oracle.flip_hash_horizontal(h)
    assert d == [h, r90, r180, r270, hf, oracle.rotate_hash_90(hf), oracle.rotate_hash_180(hf), oracle.rotate_hash_270(hf)]
    assert len(set(d)) == 8


def test_mih_csr_layout(oracle):
    """hamminghash.rs:89-130: counts, prefix sums, ascending ids per bucket; get_chunk LE u16 (:50-53)."""
    rng = np.random.default_rng(5)
    hashes = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    hashes[17] = hashes[3]
    m = oracle.MIHIndex(KIND_PDQ, hashes)
    off, vals = m.csr()
    assert off[0] == 0 and off[-1] == 16 * 300 and len(vals) == 16 * 300
    for k in (0, 7, 15):
        chunk = hashes[:, 2 * k].astype(np.uint32) | (hashes[:, 2 * k + 1].astype(np.uint32) << 8)
        for v in np.unique(chunk)[:20]:
            ids = m.bucket(k, int(v))
            assert np.array_equal(ids, np.nonzero(chunk == v)[0].astype(np.uint32))
    m64 = oracle.MIHIndex(KIND_U64, np.array([0x0102030405060708, 0x0102030405060708, 0xFF], np.uint64))
    assert list(m64.bucket(0, 0x08)) == [0, 1] and list(m64.bucket(7, 0x01)) == [0, 1] and list(m64.bucket(0, 0xFF)) == [2]


def test_sparse_bitset(oracle):
    """hamminghash.rs:152-189: set() returns was_set; clear() resets only dirty words."""
    L = oracle.lib()
    s = L.rph_ref_sbs_new(1000)
    assert L.rph_ref_sbs_set(s, 5) == 0 and L.rph_ref_sbs_set(s, 5) == 1
    assert L.rph_ref_sbs_set(s, 999) == 0 and L.rph_ref_sbs_set(s, 64) == 0 and L.rph_ref_sbs_set(s, 64) == 1
    L.rph_ref_sbs_clear(s)
    assert L.rph_ref_sbs_set(s, 5) == 0 and L.rph_ref_sbs_set(s, 999) == 0
    L.rph_ref_sbs_free(s)


def _cluster_set(rng, n, n_clusters, max_flip):
    hashes = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for c in range(n_clusters):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        for j in rng.choice(n, 4, replace=False):
            v = base.copy()
            for b in rng.choice(256, rng.integers(0, max_flip + 1), replace=False):
                v[b // 8] ^= 1 << (b % 8)
            hashes[j] = v
    return hashes


@pytest.mark.parametrize("thr", [0, 10, 15, 16, 31])
def test_find_groups_adjacency_is_exact_up_to_31(oracle, thr):
    """Pigeonhole: with 16 chunks, R<=1 probing reaches every pair with d <= 31."""
    rng = np.random.default_rng(100 + thr)
    hashes = _cluster_set(rng, 3000, 40, 20)
    m = oracle.MIHIndex(KIND_PDQ, hashes)
    brute = oracle.all_pairs256(hashes, thr)
    want = {}
    for i, j, d in brute:
        want.setdefault(int(i), set()).add(int(j))
        want.setdefault(int(j), set()).add(int(i))
    for i in range(0, 3000, 7):
        got = set(int(x) for x in m.query(i, thr))
        if thr >= 16:
            assert got == want.get(i, set())
        else:  # chunk_tolerance 0: only pairs sharing an exact 16-bit chunk are reachable
            assert got <= want.get(i, set())


def test_find_groups_misses_unreachable_pairs_above_31(oracle):
    """A d=32 pair with exactly 2 differing bits in every 16-bit chunk is invisible to R<=1 probing
    (find_groups) but found by group_files_generic (R=2 at similarity >= 32)."""
    a = np.zeros(32, np.uint8)
    b = np.zeros(32, np.uint8)
    b[0::2] = 0x03  # chunk k = LE u16 of bytes 2k,2k+1 -> 2 bits per chunk
    assert oracle.hamming256(a, b) == 32
    hashes = np.stack([a, b])
    assert oracle.find_groups(KIND_PDQ, hashes, 32) == []
    edges, groups = oracle.group_pdq(hashes, 32)
    assert edges.tolist() == [[0, 1]] and groups == [[0, 1]]
    edges, groups = oracle.group_pdq(hashes, 31)
    assert edges.tolist() == [] and groups == []


@pytest.mark.parametrize("sim", [0, 15, 16, 31, 32, 40, 47, 48, 63])
def test_group_pdq_edges_equal_brute_force(oracle, sim):
    """scanner.rs:1728-1767: R<=3 probing is exact for similarity <= 63 (no variants, no low-confidence)."""
    rng = np.random.default_rng(200 + sim)
    hashes = _cluster_set(rng, 1500, 30, 70)
    edges, groups = oracle.group_pdq(hashes, sim)
    brute = oracle.all_pairs256(hashes, sim)
    assert sorted(map(tuple, edges.tolist())) == sorted((int(i), int(j)) for i, j, _ in brute)
    # connected components of the brute-force graph
    parent = list(range(1500))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for i, j, _ in brute:
        parent[find(int(i))] = find(int(j))
    comps = {}
    for i in range(1500):
        comps.setdefault(find(i), []).append(i)
    want = sorted(v for v in comps.values() if len(v) > 1)
    assert groups == want


def test_group_pdq_low_confidence_and_variants(oracle):
    """scanner.rs:1699,1721: limit 0 if either file is low quality; :1701-1702 visited cleared per
    variant so an (i,j) pair can be emitted once per matching variant; :1716 cand_idx <= i skipped."""
    a = np.zeros(32, np.uint8)
    b = a.copy(); b[0] = 0x01          # d(a,b) = 1
    c = a.copy()                        # exact duplicate of a
    hashes = np.stack([a, b, c])
    edges, groups = oracle.group_pdq(hashes, 10, quality=[100, 100, 100])
    assert sorted(map(tuple, edges.tolist())) == [(0, 1), (0, 2), (1, 2)] and groups == [[0, 1, 2]]
    edges, groups = oracle.group_pdq(hashes, 10, quality=[100, 49, 100])   # b low quality: only exact matches
    assert sorted(map(tuple, edges.tolist())) == [(0, 2)] and groups == [[0, 2]]
    edges, groups = oracle.group_pdq(hashes, 10, quality=[49, -1, 49])     # both low: exact duplicate still pairs
    assert sorted(map(tuple, edges.tolist())) == [(0, 2)]
    # variants: file 0's variant 3 equals file 1's hash; file 1 has no variant matching file 0 (and i<j only)
    x = np.full(32, 0xAA, np.uint8)
    y = np.full(32, 0x55, np.uint8)
    variants = np.zeros((2, 8, 32), np.uint8)
    variants[0, :] = x; variants[0, 3] = y; variants[0, 5] = y
    variants[1, :] = y
    edges, groups = oracle.group_pdq(np.stack([x, y]), 0, variants=variants)
    assert edges.tolist() == [[0, 1], [0, 1]] and groups == [[0, 1]]   # comparison_count counts both variants
    edges, groups = oracle.group_pdq(np.stack([x, y]), 0, variants=variants, has_features=[0, 1])
    assert edges.tolist() == []


def test_find_groups_greedy_is_star_not_closure(oracle):
    """hamminghash.rs:245-268: a chain 0-1-2 with d(0,2) > max_dist gives [0,1] only; 2 is dropped
    because its single neighbour is already visited."""
    h = np.zeros((3, 32), np.uint8)
    h[1, 0] = 0x0F        # d(0,1)=4
    h[2, 0] = 0xFF        # d(1,2)=4, d(0,2)=8
    assert oracle.find_groups(KIND_PDQ, h, 4) == [[0, 1]]
    assert oracle.group_pdq(h, 4)[1] == [[0, 1, 2]]  # union-find closes the chain


def test_synth_hashes_clusters(oracle):
    n, nc = 5000, 20
    hs = oracle.synth_hashes(0, n, n, n_clusters=nc)
    assert len(np.unique(hs, axis=0)) >= n - nc  # popcount-0 members duplicate their base
    for c in range(nc):
        idx = [oracle.synth_cluster_index(n, c, j) for j in range(5)]
        assert len(set(idx)) == 5
        d = [oracle.hamming256(hs[idx[0]], hs[i]) for i in idx]
        assert d == [0, 1, 2, 8, 16]
    ia, ib = oracle.synth_cluster_index(n, nc, 0), oracle.synth_cluster_index(n, nc, 1)
    assert oracle.hamming256(hs[ia], hs[ib]) == 32
    x = hs[ia] ^ hs[ib]
    assert all(bin(int(x[2 * k]) | int(x[2 * k + 1]) << 8).count("1") == 2 for k in range(16))
    # windowed generation is consistent with whole-set generation
    part = oracle.synth_hashes(1234, 777, n, n_clusters=nc)
    assert np.array_equal(part, hs[1234:1234 + 777])
    edges, groups = oracle.group_pdq(hs, 32)
    assert len(groups) == nc + 1 and sorted(len(g) for g in groups) == [2] + [5] * nc
    fg = oracle.find_groups(KIND_PDQ, hs, 32)
    assert len(fg) == nc  # the special pair is invisible to find_groups
