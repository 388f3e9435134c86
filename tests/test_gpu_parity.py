"""GPU parity tests (-m gpu): every result that crosses the C ABI is compared bit for bit with the
CPU oracle on the same seeded inputs; full-size runs are checked through construction-known answers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from rupphash_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def edge_set(edges):
    return sorted((int(e["i"]), int(e["j"]), int(e["d"])) for e in edges)


def clustered_hashes(rng, n, n_clusters, max_flip, members=4):
    hashes = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for _ in range(n_clusters):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        for j in rng.choice(n, members, replace=False):
            v = base.copy()
            for b in rng.choice(256, rng.integers(0, max_flip + 1), replace=False):
                v[b // 8] ^= 1 << (b % 8)
            hashes[j] = v
    return hashes


# ------------------------------------------------------------------ synthetic generators
def test_synth_images_match_oracle(eng, oracle):
    for first_k, n, w, h in [(0, 2, 512, 512), (998, 3, 512, 512), (5, 3, 100, 36), (2**33 + 7, 2, 64, 64)]:
        got = eng.synth_images(first_k, n, w, h)
        want = oracle.synth_images(first_k, n, w, h)
        assert np.array_equal(got, want), (first_k, n, w, h)
    # k % 1000 == 999 shares its blocks with k - 1 (near duplicate), different noise
    pair = eng.synth_images(998, 2)
    diff = np.abs(pair[0].astype(int) - pair[1].astype(int))
    assert 0 < diff.max() <= 15


def test_synth_hashes_match_oracle(eng, oracle):
    n, nc = 50_000, 100
    assert np.array_equal(eng.synth_hashes(0, n, n, n_clusters=nc), oracle.synth_hashes(0, n, n, n_clusters=nc))
    assert np.array_equal(eng.synth_hashes(12_345, 7_000, n, n_clusters=nc), oracle.synth_hashes(12_345, 7_000, n, n_clusters=nc))
    assert np.array_equal(eng.synth_hashes(0, 3, 3), oracle.synth_hashes(0, 3, 3))


# ------------------------------------------------------------------ PDQ
GEOMS = [(512, 512, 3), (5, 5, 3), (64, 64, 1), (100, 37, 3), (512, 300, 4), (333, 512, 3), (511, 509, 3), (7, 500, 1),
         (129, 65, 3)]


@pytest.mark.parametrize("which", [0, 4, 6])
@pytest.mark.parametrize("w,h,ch", GEOMS)
def test_pdq_generic_kernel_matches_oracle(eng, oracle, w, h, ch, which):
    """which = 0: the plain multi-pass kernels (one thread per line); 4: the default (512x512 fused; elsewhere, for calls this small, rows through
    LDS tiles, the second half of the filter on the 64 kept columns only); 6: the streaming single-pass kernel wherever it applies"""
    rng = np.random.default_rng(w * 1000 + h + ch)
    n = 3
    if ch == 1:
        imgs = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    else:
        imgs = rng.integers(0, 256, (n, h, w, ch), dtype=np.uint8)
        # smooth structure so the hash is not pure noise: add a gradient to image 0
        imgs[0, ..., 0] = (np.arange(w)[None, :] * 255 // max(w - 1, 1)).astype(np.uint8)
    eng.set_pdq_kernel(which)
    out = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    for k in range(n):
        rc, coeffs, q = oracle.pdq_features(imgs[k])
        assert rc == 0 and out["valid"][k] == 1
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), f"coefficients differ for image {k}"
        assert bits(out["quality"][k:k + 1])[0] == bits(np.float32(q))[()]
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs))
        assert np.array_equal(out["dihedral"][k], oracle.dihedral_hashes(coeffs))


def test_pdq_special_images(eng, oracle):
    """flat image (quality 0, all coefficients tie), black, white, checkerboard, single bright pixel."""
    h = w = 96
    flat = np.full((h, w, 3), 128, np.uint8)
    black = np.zeros((h, w, 3), np.uint8)
    white = np.full((h, w, 3), 255, np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    checker = np.repeat((((yy // 8 + xx // 8) % 2) * 255).astype(np.uint8)[..., None], 3, axis=2)
    dot = black.copy()
    dot[40, 50] = 255
    imgs = np.stack([flat, black, white, checker, dot])
    eng.set_pdq_kernel(0)
    out = eng.pdq_hash_batch(imgs, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    for k in range(len(imgs)):
        rc, coeffs, q = oracle.pdq_features(imgs[k])
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), k
        assert out["quality"][k] == np.float32(q)
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs)), k
        assert np.array_equal(out["dihedral"][k], oracle.dihedral_hashes(coeffs)), k
    assert out["quality"][0] == 0.0


def test_pdq_none_and_unsupported(eng):
    from rupphash_amd import RphError, pdqhash

    out = eng.pdq_hash_batch(np.zeros((2, 4, 64, 3), np.uint8))
    assert not out["valid"].any() and not out["hash"].any()
    assert pdqhash.generate_pdq(np.zeros((64, 4, 3), np.uint8), eng) is None        # pdqhash.rs:167-169
    assert pdqhash.generate_pdq_features(np.zeros((5, 5, 3), np.uint8), eng) is not None
    assert eng.pdq_hash_batch(np.zeros((1, 513, 16, 3), np.uint8))["valid"][0] == 1   # > 512: resized, then hashed


RESIZED = [(780, 768, 3), (1280, 854, 3), (513, 512, 3), (512, 513, 1), (4000, 5, 3), (5, 4000, 3), (1024, 1024, 4), (600, 600, 1),
           (2048, 1536, 3)]


@pytest.mark.parametrize("which", [0, 4, 6])
@pytest.mark.parametrize("w,h,ch", RESIZED)
def test_pdq_with_predownsample_matches_oracle(eng, oracle, w, h, ch, which):
    """pdqhash.rs:181-220: sides > 512 px go through luma -> box-convolution thumbnail -> PDQ.  GPU == oracle bit for bit
    (the resize itself is a restatement of third-party code: parity unpinned against the Rust binary)."""
    rng = np.random.default_rng(w * 7 + h)
    n = 2
    shape = (n, h, w) if ch == 1 else (n, h, w, ch)
    imgs = rng.integers(0, 256, shape, dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((xx * 200) // max(w - 1, 1) + (yy * 55) // max(h - 1, 1)).astype(np.uint8)
    if ch == 1:
        imgs[0] = grad
    else:
        imgs[0, ..., 0] = grad
        imgs[0, ..., 1] = grad[::-1]
    eng.set_pdq_kernel(which)  # 0: luma plane + two resize kernels + plain hasher; 4: both resize passes in one kernel (on the matrix pipe for Luma8), tiled hasher; 6: streaming hasher
    out = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    for k in range(n):
        rc, coeffs, q = oracle.pdq_features(imgs[k])
        assert rc == 0 and out["valid"][k] == 1
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), k
        assert bits(out["quality"][k:k + 1])[0] == bits(np.float32(q))[()]
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs))
        assert np.array_equal(out["dihedral"][k], oracle.dihedral_hashes(coeffs))


ODD_LARGE = [(5000, 3000, 3), (8191, 17, 1), (513, 513, 4), (6000, 7, 3), (7, 6000, 1), (2049, 1025, 3), (30000, 20, 3), (20, 30000, 1), (9000, 6000, 1), (1537, 1024, 3),
             (3073, 2049, 1), (4097, 513, 4)]


@pytest.mark.parametrize("w,h,ch", ODD_LARGE)
def test_predownsample_odd_and_very_large_geometries(eng, oracle, w, h, ch):
    """the fused resize kernel's corners: windows of 3 / 5 / 7 and beyond (scale > 6: the general form), staged rows of one to eight
    256-byte pieces and beyond (two-pass fallback), tiles cut by the right and bottom edges, thumbnails one pixel wide or high, sources of
    tens of megapixels.  Default kernels == plain two-pass kernels, and the oracle where it is quick enough"""
    rng = np.random.default_rng(w + 3 * h + ch)
    shape = (2, h, w) if ch == 1 else (2, h, w, ch)
    imgs = rng.integers(0, 256, shape, dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    grad = ((xx * 200) // max(w - 1, 1) + (yy * 55) // max(h - 1, 1)).astype(np.uint8)
    if ch == 1:
        imgs[0] = grad
    else:
        imgs[0, ..., 1] = grad
    outs = []
    for which in (0, 4, 6):
        eng.set_pdq_kernel(which)
        outs.append(eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True))
    eng.set_pdq_kernel(4)
    assert np.array_equal(outs[2]["hash"], outs[1]["hash"]) and np.array_equal(outs[2]["coeffs"].view(np.uint32), outs[1]["coeffs"].view(np.uint32))
    assert np.array_equal(outs[2]["quality"].view(np.uint32), outs[1]["quality"].view(np.uint32)) and np.array_equal(outs[2]["dihedral"], outs[1]["dihedral"])
    assert np.array_equal(outs[0]["valid"], outs[1]["valid"]) and outs[1]["valid"].all()
    assert np.array_equal(outs[0]["hash"], outs[1]["hash"]) and np.array_equal(outs[0]["dihedral"], outs[1]["dihedral"])
    assert np.array_equal(outs[0]["coeffs"].view(np.uint32), outs[1]["coeffs"].view(np.uint32))
    assert np.array_equal(outs[0]["quality"].view(np.uint32), outs[1]["quality"].view(np.uint32))
    if w * h <= 16_000_000:
        rc, coeffs, q = oracle.pdq_features(imgs[1])
        assert rc == 0 and np.array_equal(bits(outs[1]["coeffs"][1]), bits(coeffs)) and np.array_equal(outs[1]["hash"][1], oracle.to_hash(coeffs))


def lcg_features(seed):
    state = np.uint32(seed)
    out = np.zeros(256, np.float32)
    with np.errstate(over="ignore"):
        for i in range(256):
            state = np.uint32(state * np.uint32(1664525) + np.uint32(1013904223))
            out[i] = np.float32(np.float32(int(state) >> 8) / np.float32(65536.0)) - np.float32(128.0)
    return out


def test_hashes_from_coeffs_match_oracle_and_reference_properties(eng, oracle):
    """pdqhash.rs:548-570 on the GPU: to_hash / generate_dihedral_hashes == naive composition; 8 distinct."""
    from rupphash_amd.pdqhash import PdqFeatures

    feats = [lcg_features(s) for s in (1, 42, 0x12345678, 0xDEADBEEF, 7)]
    rng = np.random.default_rng(9)
    feats += [rng.normal(0, 30, 256).astype(np.float32) for _ in range(20)]
    ties = np.zeros(256, np.float32)
    ties[:100] = -0.0
    ties[200:] = 1.0
    feats += [ties, np.zeros(256, np.float32), np.arange(256, dtype=np.float32) - 127.5]
    hashes, dih = eng.pdq_hashes_from_coeffs(np.stack(feats))
    for k, f in enumerate(feats):
        assert np.array_equal(hashes[k], oracle.to_hash(f)), k
        assert np.array_equal(dih[k], oracle.dihedral_hashes(f)), k
        assert np.array_equal(dih[k], oracle.naive_dihedral(f)), k
    assert len({bytes(x) for x in dih[4]}) == 8
    # the per-file (host scalar) forms give the same bits as the batch kernels
    for k, c in enumerate(feats):
        f = PdqFeatures(c)
        assert np.array_equal(f.to_hash(), hashes[k]) and np.array_equal(f.generate_dihedral_hashes(), dih[k]), k


def test_dihedral_hashes_match_physically_transformed_image(eng, oracle):
    """pdqhash.rs:583-628 end to end on the GPU: hashing a transposed / mirrored 64x64 Luma8 image
    (windows are 1, so the 64x64 buffer is the image itself) lands exactly on the predicted slot."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (64, 64), dtype=np.uint8)
    n = 64
    tr = {0: img, 1: img[::-1, :].T, 2: img[::-1, ::-1], 3: img.T[::-1, :], 4: img[:, ::-1], 5: img[::-1, :], 6: img.T,
          7: img[::-1, ::-1].T}
    # the reference's transform(): out[x][y] = in[N-1-y][x] for variant 1, etc.
    assert tr[1][3, 5] == img[n - 1 - 5, 3] and tr[3][3, 5] == img[5, n - 1 - 3] and tr[7][3, 5] == img[n - 1 - 5, n - 1 - 3]
    eng.set_pdq_kernel(0)
    base = eng.pdq_hash_batch(img[None], want_dihedral=True)["dihedral"][0]
    got = eng.pdq_hash_batch(np.stack([np.ascontiguousarray(tr[v]) for v in range(8)]))["hash"]
    eng.set_pdq_kernel(4)
    for v in range(8):
        assert np.array_equal(got[v], base[v]), v


# ------------------------------------------------------------------ Hamming sweep
# 2 = fp4 MFMA fast path (default; +-1 operands below 32768 hashes), 4 = its popcount-sorted {0,1} form forced at every size,
# 3 = fp4 +-1 forced, 1 = int8 MFMA, 0 = VALU xor + popcount
@pytest.mark.parametrize("kernel", [2, 4, 3, 1, 0])
@pytest.mark.parametrize("thr", [0, 10, 31, 32, 36, 40, 41, 48, 53, 54, 60, 66, 67, 74, 80, 81, 100, 200])
def test_all_pairs_matches_brute_force(eng, oracle, thr, kernel):
    rng = np.random.default_rng(300 + thr)
    hashes = clustered_hashes(rng, 2500, 60, min(thr + 20, 120))
    eng.set_hamming_kernel(kernel)
    got = eng.hamming_all_pairs(hashes, thr)
    eng.set_hamming_kernel(2)
    want = oracle.all_pairs256(hashes, thr)
    assert edge_set(got) == sorted(map(tuple, want.tolist()))
    if thr == 32:  # flags (find_groups reachability / probe slot) agree between the two formulations
        eng.set_hamming_kernel(0 if kernel else 1)
        other = eng.hamming_all_pairs(hashes, thr)
        eng.set_hamming_kernel(2)
        key = lambda e: sorted((int(x["i"]), int(x["j"]), int(x["d"]), int(x["flags"])) for x in e)
        assert key(got) == key(other)


@pytest.mark.parametrize("kernel", [2, 4])
def test_all_pairs_sharded_over_parts(eng, oracle, kernel):
    rng = np.random.default_rng(77)
    hashes = clustered_hashes(rng, 5000, 80, 40)
    want = sorted(map(tuple, oracle.all_pairs256(hashes, 32).tolist()))
    eng.set_hamming_kernel(kernel)
    try:
        for nparts in (2, 3, 8):
            parts = [edge_set(eng.hamming_all_pairs(hashes, 32, part=p, nparts=nparts)) for p in range(nparts)]
            merged = sorted(sum(parts, []))
            assert merged == want and sum(len(p) for p in parts) == len(want)
    finally:
        eng.set_hamming_kernel(2)


@pytest.mark.parametrize("thr", [0, 3, 32, 40, 53, 64, 130])
def test_sorted_zero_one_sweep_on_adversarial_popcounts(eng, oracle, thr):
    """The {0,1} formulation thresholds on popcount(a) + popcount(b) - 2 popcount(a & b) with block-wise popcount minima of the
    SORTED hashes: exercise prefixes of extreme and of widely spread popcount (sparse, dense, all-zero, all-one hashes, clusters that
    straddle popcount steps), where the per-block bound is loosest, against the +-1 kernel, the VALU kernel and brute force."""
    rng = np.random.default_rng(900 + thr)
    n = 3000
    dens = rng.choice([0.0, 0.02, 0.1, 0.3, 0.5, 0.7, 0.9, 0.98, 1.0], n)
    bits = (rng.random((n, 256)) < dens[:, None]).astype(np.uint8)
    hashes = np.packbits(bits, axis=1)
    # near duplicates of sparse and dense hashes, and pairs that differ only OUTSIDE the 128-bit prefix
    for k in range(0, 600, 3):
        v = hashes[k].copy()
        for b in rng.choice(256, int(rng.integers(0, thr + 6)), replace=False):
            v[b >> 3] ^= 1 << (b & 7)
        hashes[k + 1] = v
        w = hashes[k].copy()
        for b in rng.choice(np.arange(128, 256), int(rng.integers(0, min(thr + 6, 128))), replace=False):
            w[b >> 3] ^= 1 << (b & 7)
        hashes[k + 2] = w
    want = sorted(map(tuple, oracle.all_pairs256(hashes, thr, cap=1 << 23).tolist()))
    res = {}
    for kernel in (4, 3, 0):
        eng.set_hamming_kernel(kernel)
        res[kernel] = sorted((int(x["i"]), int(x["j"]), int(x["d"]), int(x["flags"])) for x in eng.hamming_all_pairs(hashes, thr, cap=1 << 23))
    eng.set_hamming_kernel(2)
    assert [t[:3] for t in res[4]] == want
    assert res[4] == res[3] == res[0]  # flags (find_groups reachability / probe slot) too


@pytest.mark.parametrize("thr", [0, 40, 70])
def test_all_pairs_heavily_duplicated_data(eng, thr):
    """thousands of identical hashes: every pair of a chunk is a candidate, the MFMA kernel's candidate queue overflows
    and its exhaustive fallback must still report every pair exactly once"""
    n = 2100
    h = np.tile(np.arange(32, dtype=np.uint8) * 7, (n, 1))
    h[n - 100:, 0] ^= 0xFF          # a second cluster at distance 8 from the first
    e = eng.hamming_all_pairs(h, thr, cap=n * n // 2)
    a, b = n - 100, 100
    want = a * (a - 1) // 2 + b * (b - 1) // 2 + (a * b if thr >= 8 else 0)
    assert len(e) == want and (e["i"] < e["j"]).all()
    assert len({(int(x["i"]), int(x["j"])) for x in e}) == want
    assert set(np.unique(e["d"]).tolist()) == ({0, 8} if thr >= 8 else {0})


def test_all_pairs_edge_cases(eng):
    assert len(eng.hamming_all_pairs(np.zeros((0, 32), np.uint8), 10)) == 0
    assert len(eng.hamming_all_pairs(np.zeros((1, 32), np.uint8), 10)) == 0
    same = np.tile(np.arange(32, dtype=np.uint8), (40, 1))
    e = eng.hamming_all_pairs(same, 0)
    assert len(e) == 40 * 39 // 2 and (e["d"] == 0).all() and (e["i"] < e["j"]).all()
    far = np.stack([np.zeros(32, np.uint8), np.full(32, 255, np.uint8)])
    assert edge_set(eng.hamming_all_pairs(far, 256)) == [(0, 1, 256)] and len(eng.hamming_all_pairs(far, 255)) == 0
    # capacity protocol
    import ctypes as C

    from rupphash_amd import EDGE_DTYPE

    buf = np.zeros(10, EDGE_DTYPE)
    found = C.c_uint64()
    rc = eng.L.rph_hamming_all_pairs(eng.ctx, same.ctypes.data, 40, 0, 0, 1, buf.ctypes.data, 10, C.byref(found))
    assert rc == -6 and found.value == 780 and (buf["i"] < buf["j"]).all()


@pytest.mark.parametrize("thr", [0, 5, 15, 16, 20, 31, 32, 40])
def test_find_groups_matches_reference_semantics(eng, oracle, thr):
    """hamminghash.rs:191-271 bit-exact, including group-internal member order and the R<=1 reachability gap."""
    rng = np.random.default_rng(400 + thr)
    hashes = clustered_hashes(rng, 3000, 70, 36, members=5)
    a = np.zeros(32, np.uint8)
    b = np.zeros(32, np.uint8)
    b[0::2] = 0x03  # d = 32, two bits in every 16-bit chunk: unreachable by R<=1 probing
    hashes[100], hashes[2000] = a, b
    assert eng.find_groups256(hashes, thr) == oracle.find_groups(oracle.KIND_PDQ, hashes, thr)


def test_find_groups_reference_tests_on_gpu(eng):
    """hamminghash.rs:310-331 through the mirror module."""
    from rupphash_amd import hamminghash as hh

    base = np.zeros(32, np.uint8)
    target = np.zeros(32, np.uint8)
    for i in range(30):
        target[i // 8] |= 1 << (i % 8)
    idx = hh.MIHIndex(np.stack([base, target]), eng)
    assert hh.find_groups(idx, 30) == [[0, 1]]
    assert idx.len() == 2 and list(idx.bucket(15, 0)) == [0, 1] and list(idx.bucket(0, 0xFFFF)) == [1]


@pytest.mark.parametrize("sim", [0, 16, 31, 40, 63])
def test_group_files_pdq_matches_oracle(eng, oracle, sim):
    """scanner.rs:1640-1823: 8 dihedral variants of file i against hash j > i, low-confidence rule, union-find."""
    rng = np.random.default_rng(500 + sim)
    n = 1500
    coeffs = rng.normal(0, 20, (n, 256)).astype(np.float32)
    # near-duplicate coefficient sets (small perturbations) and a mirrored copy (sign flip on odd-frequency columns)
    for c in range(40):
        src = rng.integers(0, n)
        for j in rng.choice(n, 3, replace=False):
            coeffs[j] = coeffs[src] + rng.normal(0, 0.8, 256).astype(np.float32)
    mirrored = coeffs[10].reshape(16, 16).copy()
    mirrored[:, 0::2] *= -1
    coeffs[1400] = mirrored.ravel()
    hashes, dih = eng.pdq_hashes_from_coeffs(coeffs)
    quality = rng.integers(30, 101, n).astype(np.int32)
    quality[::7] = -1
    has_features = (rng.random(n) > 0.1).astype(np.uint8)
    var = dih.copy()
    want_edges, want_groups = oracle.group_pdq(hashes, sim, variants=var, has_features=has_features, quality=quality)
    groups, cmp_count = eng.group_files_pdq(hashes, sim, coeffs=coeffs, has_features=has_features, quality=quality)
    assert groups == want_groups and cmp_count == len(want_edges)
    # edge multiset through the variant sweep entry point
    v2 = var.copy()
    v2[has_features == 0] = hashes[has_features == 0][:, None, :]
    low = np.array([oracle.lib().rph_ref_is_low_pdq_quality(int(q)) for q in quality], np.uint8)
    e = eng.hamming_variant_pairs(v2, hashes, sim, low_conf=low)
    keep = [(int(x["i"]), int(x["j"])) for x in e if has_features[x["i"]] or ((x["flags"] >> 9) & 7) == 0]
    assert sorted(keep) == sorted(map(tuple, want_edges.tolist()))
    # no features at all: single variant = the hash itself
    g1, c1 = eng.group_files_pdq(hashes, sim, quality=quality)
    we, wg = oracle.group_pdq(hashes, sim, quality=quality)
    assert g1 == wg and c1 == len(we)


def test_group_files_rejects_similarity_above_63(eng):
    from rupphash_amd import RphError

    with pytest.raises(RphError):
        eng.group_files_pdq(np.zeros((4, 32), np.uint8), 64)


def test_mih_build_matches_reference_csr(eng, oracle):
    rng = np.random.default_rng(6)
    hashes = rng.integers(0, 256, (20_000, 32), dtype=np.uint8)
    hashes[:, 4] &= 0x03  # crowded buckets in chunk 2
    hashes[5000:5100] = hashes[0]
    off, vals = eng.mih_build256(hashes)
    woff, wvals = oracle.MIHIndex(oracle.KIND_PDQ, hashes).csr()
    assert np.array_equal(off, woff) and np.array_equal(vals, wvals)


def test_mih_build64_matches_reference_csr(eng, oracle):
    """MIHIndex::<u64>::new (hamminghash.rs:23-41, :89-130): 8 chunks of 8 bits, ids ascending inside a bucket"""
    rng = np.random.default_rng(7)
    hashes = rng.integers(0, 2**64, 50_000, dtype=np.uint64)
    hashes[1000:1200] = hashes[0]                       # a crowded bucket in every chunk
    hashes[2000:3000] &= np.uint64(0xFFFFFFFFFFFF00FF)  # chunk 1 == 0 for a thousand hashes
    off, vals = eng.mih_build64(hashes)
    woff, wvals = oracle.MIHIndex(oracle.KIND_U64, hashes).csr()
    assert np.array_equal(off, woff) and np.array_equal(vals, wvals)
    from rupphash_amd import hamminghash

    idx = hamminghash.MIHIndex64(hashes, engine=eng)
    ref = oracle.MIHIndex(oracle.KIND_U64, hashes)
    for chunk, value in [(0, int(hashes[0]) & 0xFF), (1, 0), (7, int(hashes[5]) >> 56), (3, 255)]:
        assert np.array_equal(idx.bucket(chunk, value), ref.bucket(chunk, value))
    off0, vals0 = eng.mih_build64(np.zeros(0, np.uint64))
    assert not off0.any() and len(vals0) == 0


# ------------------------------------------------------------------ full-size, construction-known answers
@pytest.mark.parametrize("kernel,n", [(2, 1_000_000), (3, 1_000_000), (1, 1_000_000), (0, 1_000_000), (2, 3_100_000), (1, 3_100_000)])
def test_one_million_hashes_threshold_32(eng, oracle, kernel, n):
    """BASELINE config 3: 1M synthetic hashes, 1000 injected 5-member clusters + the distance-32 pair.
    The expected edge set follows from the construction (random 256-bit pairs at d <= 32 have
    probability ~1e-33), so the full-size sweep is checked exactly.  3.1M hashes = 4.6M tile pairs: more than one
    launch can carry (gridDim.x * blockDim.x < 2^32), so the sweep is split across launches."""
    nc = 1000
    d_h = eng.dev_alloc(n * 32)
    cap = 1 << 16
    d_e = eng.dev_alloc(cap * 12)
    d_c = eng.dev_alloc(8)
    try:
        eng.synth_hashes_dev(d_h, 0, n, n, n_clusters=nc)
        eng.dev_memset(d_c, 0, 8)
        eng.set_hamming_kernel(kernel)
        eng.hamming_all_pairs_dev(d_h, n, 32, d_e, cap, d_c)
        eng.synchronize()
        eng.set_hamming_kernel(2)
        cnt = np.zeros(1, np.uint64)
        eng.dev_download(cnt, d_c)
        from rupphash_amd import EDGE_DTYPE

        edges = np.zeros(int(cnt[0]), EDGE_DTYPE)
        eng.dev_download(edges, d_e)
    finally:
        for p in (d_h, d_e, d_c):
            eng.dev_free(p)
    pops = [0, 1, 2, 8, 16]
    want = []
    for c in range(nc):
        idx = [oracle.synth_cluster_index(n, c, j) for j in range(5)]
        masks = []
        for j in range(5):
            m = 0
            for t in range(pops[j]):
                m |= 1 << ((c * 31 + j * 11 + t * 37) & 255)
            masks.append(m)
        for a in range(5):
            for b in range(a + 1, 5):
                i, j = sorted((idx[a], idx[b]))
                want.append((i, j, bin(masks[a] ^ masks[b]).count("1")))
    i, j = sorted((oracle.synth_cluster_index(n, nc, 0), oracle.synth_cluster_index(n, nc, 1)))
    want.append((i, j, 32))
    assert edge_set(edges) == sorted(want)
    special = [e for e in edges if e["d"] == 32 and (int(e["i"]), int(e["j"])) == (i, j)][0]
    assert not (special["flags"] & 0x8000)  # unreachable for find_groups' R<=1 probing


# ------------------------------------------------------------------ fused 512x512 kernel
def _fused_images():
    rng = np.random.default_rng(2026)
    yy, xx = np.mgrid[0:512, 0:512]
    imgs = [rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)]
    ramp = np.zeros((512, 512, 3), np.uint8)
    ramp[..., 0] = (xx * 255) // 511
    ramp[..., 1] = (yy * 255) // 511
    ramp[..., 2] = ((xx + yy) * 255) // 1022
    imgs.append(ramp)
    imgs.append(np.repeat((((yy // 32 + xx // 32) % 2) * 255).astype(np.uint8)[..., None], 3, axis=2))
    imgs.append(np.full((512, 512, 3), 255, np.uint8))      # saturated: largest running sums
    imgs.append(np.full((512, 512, 3), 128, np.uint8))      # flat: quality 0, all coefficients tie
    imgs.append(np.zeros((512, 512, 3), np.uint8))
    tiny = rng.integers(0, 4, (512, 512, 3), dtype=np.uint8)  # tiny values next to a bright frame: finest f32 lattice
    tiny[:, :8] = 255
    tiny[:8, :] = 251
    tiny[:, 505:] = 253
    tiny[506:, :] = 249
    imgs.append(tiny)
    dots = np.zeros((512, 512, 3), np.uint8)
    for (y, x) in [(3, 2), (509, 510), (255, 255), (0, 0), (511, 511), (0, 511), (511, 0), (4, 507), (60, 63), (64, 64), (447, 448)]:
        dots[y, x] = 255
    imgs.append(dots)
    return np.stack(imgs)


@pytest.mark.parametrize("which", [1, 2, 3])  # one wave per image (64- / 128-px strips), eight waves per image
def test_fused512_kernel_matches_oracle_bit_for_bit(eng, oracle, which):
    imgs = _fused_images()
    eng.set_pdq_kernel(which)
    out = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.set_pdq_kernel(4)
    for k in range(len(imgs)):
        rc, coeffs, q = oracle.pdq_features(imgs[k])
        assert rc == 0 and out["valid"][k] == 1
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), f"coefficients differ for image {k}"
        assert bits(out["quality"][k:k + 1])[0] == bits(np.float32(q))[()], k
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs)), k
        assert np.array_equal(out["dihedral"][k], oracle.dihedral_hashes(coeffs)), k


def test_fused512_luma8_input_matches_oracle_and_generic_kernels(eng, oracle):
    """512x512 Luma8 (the reference borrows a Luma8 image as it is, pdqhash.rs:176): the fused kernel's Luma8 form against the oracle on the
    adversarial set (one channel of each image, and its Rec.601 luma), and against the generic kernels on 2000 random gray images"""
    rgb = _fused_images()
    gray = np.concatenate([rgb[..., 1], oracle.luma601(rgb[0])[None], oracle.luma601(rgb[-1])[None]])
    out = eng.pdq_hash_batch(gray, want_quality=True, want_coeffs=True, want_dihedral=True)
    for k in range(len(gray)):
        rc, coeffs, q = oracle.pdq_features(gray[k])
        assert rc == 0 and out["valid"][k] == 1
        assert np.array_equal(bits(out["coeffs"][k]), bits(coeffs)), f"coefficients differ for gray image {k}"
        assert bits(out["quality"][k:k + 1])[0] == bits(np.float32(q))[()], k
        assert np.array_equal(out["hash"][k], oracle.to_hash(coeffs)) and np.array_equal(out["dihedral"][k], oracle.dihedral_hashes(coeffs)), k
    # a Luma8 image hashes like the gray Rgb8 image r = g = b (luma601 of equal channels is the channel: (1000 v + 500) / 1000)
    k = 3
    same = eng.pdq_hash_batch(np.repeat(gray[k][None, :, :, None], 3, axis=3), want_coeffs=True)
    assert np.array_equal(bits(same["coeffs"][0]), bits(out["coeffs"][k]))
    rng = np.random.default_rng(5)
    many = rng.integers(0, 256, (2000, 512, 512), dtype=np.uint8)
    many[::3] = (many[::3] >> 5) * 30  # coarse levels: ties
    fused = eng.pdq_hash_batch(many, want_quality=True, want_coeffs=True)
    eng.set_pdq_kernel(0)
    generic = eng.pdq_hash_batch(many, want_quality=True, want_coeffs=True)
    eng.set_pdq_kernel(4)
    assert np.array_equal(fused["hash"], generic["hash"]) and np.array_equal(bits(fused["coeffs"]), bits(generic["coeffs"]))
    assert np.array_equal(bits(fused["quality"]), bits(generic["quality"]))


@pytest.mark.parametrize("which", [1, 2, 3])
def test_fused512_on_synthetic_bench_images(eng, oracle, which):
    """the bench workload itself: 24 images of the synthetic sequence (incl. a near-duplicate pair) vs the oracle"""
    imgs = eng.synth_images(990, 24)
    eng.set_pdq_kernel(which)
    out = eng.pdq_hash_batch(imgs, want_quality=True, want_coeffs=True)
    eng.set_pdq_kernel(4)
    ref_hash, ref_q, ref_c = oracle.pdq_batch_rgb(imgs, want_coeffs=True)
    assert np.array_equal(out["hash"], ref_hash)
    assert np.array_equal(bits(out["coeffs"]), bits(ref_c)) and np.array_equal(bits(out["quality"]), bits(ref_q))
    assert oracle.hamming256(out["hash"][8], out["hash"][9]) <= 16      # k = 998 / 999: same blocks, different noise
    assert (ref_q == 1.0).all()


def test_fused512_equals_generic_on_4096_images(eng):
    """full-size style check without the oracle: the two independent device paths agree on every hash, quality, coefficient"""
    n = 4096
    d_img = eng.dev_alloc(n * 512 * 512 * 3)
    bufs = [eng.dev_alloc(n * 32), eng.dev_alloc(n * 4), eng.dev_alloc(n * 1024)]
    res = []
    try:
        eng.synth_images_dev(d_img, 123_000, n)
        for which in (1, 2, 3, 0):  # 3: the low-latency kernel in chunks of 1024 images
            eng.set_pdq_kernel(which)
            for p, nb in zip(bufs, (n * 32, n * 4, n * 1024)):
                eng.dev_memset(p, 0xEE, nb)
            eng.pdq_hash_batch_dev(d_img, n, 512, 512, 3, bufs[0], d_quality=bufs[1], d_coeffs=bufs[2])
            eng.synchronize()
            h, q, c = np.zeros((n, 32), np.uint8), np.zeros(n, np.float32), np.zeros((n, 256), np.float32)
            eng.dev_download(h, bufs[0]); eng.dev_download(q, bufs[1]); eng.dev_download(c, bufs[2])
            res.append((h, q, c))
    finally:
        eng.set_pdq_kernel(4)
        for p in [d_img] + bufs:
            eng.dev_free(p)
    for other in (1, 2, 3):
        assert np.array_equal(res[0][0], res[other][0])
        assert np.array_equal(bits(res[0][1]), bits(res[other][1])) and np.array_equal(bits(res[0][2]), bits(res[other][2]))
    # every 1000-image stripe holds one known near-duplicate pair (k % 1000 == 999 shares blocks with k - 1)
    from rupphash_amd import hamminghash as hh

    h = res[0][0]
    for k in range(123_000, 123_000 + n):
        if k % 1000 == 999 and k - 1 >= 123_000:
            assert hh.hamming_distance(h[k - 123_000], h[k - 1 - 123_000]) <= 16


# ------------------------------------------------------------------ 64-bit hashes (impl HammingHash for u64)
def test_u64_reference_tests_on_gpu(eng, oracle):
    """hamminghash.rs:286-307 ({0, 0xFFF} at 12 -> [[0, 1]]) and :336-412 (injected 5-cluster, max_dist 5) at the full 1M."""
    from rupphash_amd import hamminghash as hh

    assert hh.find_groups(hh.MIHIndex64(np.array([0, 0xFFF], np.uint64), eng), 12) == [[0, 1]]
    n = 1_000_000
    rng = np.random.default_rng(2024)
    hashes = rng.integers(0, 2**64, n, dtype=np.uint64)
    target = 0xABCD_1234_5678_90EF
    idx = rng.choice(n, 5, replace=False)
    for v, i in zip([target, target ^ 1, target ^ 2, target ^ 0x8000, target ^ 0x8001], idx):
        hashes[i] = v
    groups = hh.find_groups(hh.MIHIndex64(hashes, eng), 5)
    g = [g for g in groups if int(idx[0]) in g]
    assert g and set(int(i) for i in idx) <= set(g[0])


@pytest.mark.parametrize("kernel", [2, 0])  # 2 = fp4 MFMA sweep (default: the 64-bit hash is one MFMA slice), 0 = VALU xor + popcount
@pytest.mark.parametrize("thr", [0, 5, 7, 8, 12, 15, 20, 31, 40])
def test_u64_find_groups_and_edges_match_oracle(eng, oracle, thr, kernel):
    eng.set_hamming_kernel(kernel)
    try:
        _u64_case(eng, oracle, thr)
    finally:
        eng.set_hamming_kernel(2)


def _u64_case(eng, oracle, thr):
    rng = np.random.default_rng(600 + thr)
    n = 3000
    hashes = rng.integers(0, 2**64, n, dtype=np.uint64)
    for c in range(60):
        base = int(rng.integers(0, 2**63))
        for j in rng.choice(n, 4, replace=False):
            v = base
            for b in rng.choice(64, rng.integers(0, 14), replace=False):
                v ^= 1 << int(b)
            hashes[j] = v
    edges = eng.hamming_all_pairs64(hashes, thr)
    brute = []
    h = hashes
    x = h[:, None] ^ h[None, :]
    pc = np.zeros(x.shape, np.uint8)
    for s in range(64):
        pc += ((x >> np.uint64(s)) & np.uint64(1)).astype(np.uint8)
    ii, jj = np.nonzero(np.triu(pc <= thr, k=1))
    brute = sorted((int(i), int(j), int(pc[i, j])) for i, j in zip(ii, jj))
    assert edge_set(edges) == brute
    assert eng.find_groups64(hashes, thr) == oracle.find_groups(oracle.KIND_U64, hashes, thr)


# ---- batching queue (rph_pdq_hash_one): many threads, one image per call, as scanner.rs:1410 does
def _hash_one_worker(eng, images, out, idx):
    for k in idx:
        out[k] = eng.pdq_hash_one(images[k])


def test_hash_one_threads_coalesce_and_match_oracle(eng, oracle):
    import threading
    rng = np.random.default_rng(77)
    images = [oracle.synth_images(4242 + k, 1)[0] for k in range(96)]                 # 512x512x3
    images += [rng.integers(0, 256, (300, 200, 3), dtype=np.uint8) for _ in range(24)]  # generic geometry, own batches
    images += [rng.integers(0, 256, (40, 4), dtype=np.uint8) for _ in range(4)]         # width 4 < 5 -> None (pdqhash.rs:167-169)
    order = rng.permutation(len(images))
    out = [None] * len(images)
    eng.pdq_batcher_config(max_batch=32, max_wait_us=20000)
    b0, i0 = eng.pdq_batcher_stats()
    nthreads = 12
    threads = [threading.Thread(target=_hash_one_worker, args=(eng, images, out, order[t::nthreads])) for t in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    b1, i1 = eng.pdq_batcher_stats()
    assert i1 - i0 == len(images)
    assert b1 - b0 < len(images) // 2, "concurrent callers were not coalesced"
    for k, img in enumerate(images):
        if img.ndim == 2:
            assert out[k] is None
            continue
        rc, coeffs, q = oracle.pdq_features(img)
        assert rc == 0
        h, gq, gc = out[k]
        assert np.array_equal(gc.view(np.uint32), coeffs.view(np.uint32)), k
        assert np.float32(gq).view(np.uint32) == np.float32(q).view(np.uint32)
        assert np.array_equal(h, oracle.to_hash(coeffs))
    eng.pdq_batcher_config()


def test_hash_one_single_caller_and_bad_args(eng, oracle):
    img = oracle.synth_images(5, 1)[0]
    eng.pdq_batcher_config(max_batch=256, max_wait_us=100)
    h, q, c = eng.pdq_hash_one(img)
    rc, coeffs, oq = oracle.pdq_features(img)
    assert np.array_equal(c.view(np.uint32), coeffs.view(np.uint32))
    h2, _, none = eng.pdq_hash_one(img, want_coeffs=False)
    assert none is None and np.array_equal(h, h2)
    from rupphash_amd import _lib
    hash32 = np.zeros(32, np.uint8)
    assert eng.L.rph_pdq_hash_one(eng.ctx, None, 512, 512, 3, 1536, hash32.ctypes.data, None, None, None) == _lib.RPH_ERR_INVALID_ARG
    assert eng.L.rph_pdq_hash_one(eng.ctx, img.ctypes.data, 512, 512, 2, 1536, hash32.ctypes.data, None, None, None) == _lib.RPH_ERR_INVALID_ARG
    assert eng.L.rph_pdq_hash_one(eng.ctx, img.ctypes.data, 512, 512, 3, 100, hash32.ctypes.data, None, None, None) == _lib.RPH_ERR_INVALID_ARG
    assert eng.L.rph_pdq_batcher_config(eng.ctx, 0, 10) == _lib.RPH_ERR_INVALID_ARG
    eng.pdq_batcher_config()


# ---- strided inputs: row_stride / image_stride are part of the C ABI (a decoder's padded rows, a sub-rectangle of an atlas)
def _strided_batch(rng, n, h, w, ch, row_stride, image_stride):
    """n images embedded in one byte buffer filled with noise; returns (buffer, list of contiguous images)"""
    buf = rng.integers(0, 256, image_stride * (n - 1) + row_stride * (h - 1) + w * ch + 7, dtype=np.uint8)
    imgs = []
    for k in range(n):
        rows = [buf[k * image_stride + y * row_stride: k * image_stride + y * row_stride + w * ch] for y in range(h)]
        img = np.stack(rows)
        imgs.append(img.reshape(h, w, ch) if ch > 1 else img.reshape(h, w))
    return buf, imgs


@pytest.mark.parametrize("w,h,ch,row_pad,img_pad,which", [
    (512, 512, 3, 64, 4096, 1),    # fused strip64 with padded rows and a gap between images
    (512, 512, 3, 128, 0, 2),      # fused strip128, rows padded, images back to back
    (512, 512, 3, 4, 4, 1),        # fused kernel with rows that are only 4-byte aligned (dwordx4 loads at 4-byte alignment)
    (512, 512, 3, 12, 8, 2),
    (512, 512, 3, 1, 3, 1),        # row stride 1537: not a multiple of 4 -> generic kernel must take over
    (100, 37, 3, 5, 11, 1),        # generic kernel
    (64, 64, 1, 3, 0, 1),          # Luma8
    (200, 300, 4, 8, 16, 1),       # RGBA
    (780, 768, 3, 12, 100, 1),     # pre-downsample path
])
def test_pdq_strided_inputs_match_oracle(eng, oracle, w, h, ch, row_pad, img_pad, which):
    rng = np.random.default_rng(w * 7 + h + ch + row_pad)
    n = 3
    row_stride = w * ch + row_pad
    image_stride = row_stride * h + img_pad
    buf, imgs = _strided_batch(rng, n, h, w, ch, row_stride, image_stride)
    from rupphash_amd._lib import check
    hashes = np.zeros((n, 32), np.uint8)
    quality = np.zeros(n, np.float32)
    coeffs = np.zeros((n, 256), np.float32)
    valid = np.zeros(n, np.uint8)
    eng.set_pdq_kernel(which)
    try:
        # host entry point
        check(eng.L.rph_pdq_hash_batch(eng.ctx, buf.ctypes.data, n, w, h, ch, row_stride, image_stride, hashes.ctypes.data,
                                       quality.ctypes.data, coeffs.ctypes.data, None, valid.ctypes.data), "rph_pdq_hash_batch")
        # device entry point on the same bytes
        d_px = eng.dev_alloc(buf.nbytes)
        d_h = eng.dev_alloc(n * 32)
        try:
            eng.dev_upload(d_px, buf)
            eng.pdq_hash_batch_dev(d_px, n, w, h, ch, d_h, row_stride=row_stride, image_stride=image_stride)
            eng.synchronize()
            dev_hashes = np.zeros((n, 32), np.uint8)
            eng.dev_download(dev_hashes, d_h)
        finally:
            eng.dev_free(d_px)
            eng.dev_free(d_h)
    finally:
        eng.set_pdq_kernel(4)
    assert np.array_equal(dev_hashes, hashes)
    for k in range(n):
        rc, c, q = oracle.pdq_features(imgs[k])
        assert rc == 0 and valid[k] == 1
        assert np.array_equal(bits(coeffs[k]), bits(c)), k
        assert bits(quality[k:k + 1])[0] == bits(np.float32(q))[()]
        assert np.array_equal(hashes[k], oracle.to_hash(c))


def test_pdq_bad_strides_are_rejected(eng):
    from rupphash_amd import _lib
    img = np.zeros((64, 64, 3), np.uint8)
    h32 = np.zeros(32, np.uint8)
    f = eng.L.rph_pdq_hash_batch
    assert f(eng.ctx, img.ctypes.data, 1, 64, 64, 3, 191, 64 * 192, h32.ctypes.data, None, None, None, None) == _lib.RPH_ERR_INVALID_ARG  # row_stride < w*ch
    assert f(eng.ctx, img.ctypes.data, 2, 64, 64, 3, 192, 100, h32.ctypes.data, None, None, None, None) == _lib.RPH_ERR_INVALID_ARG      # images overlap
    assert f(eng.ctx, None, 1, 64, 64, 3, 192, 64 * 192, h32.ctypes.data, None, None, None, None) == _lib.RPH_ERR_INVALID_ARG
    assert f(eng.ctx, img.ctypes.data, 1, 64, 64, 2, 128, 64 * 128, h32.ctypes.data, None, None, None, None) in (_lib.RPH_ERR_INVALID_ARG, _lib.RPH_ERR_UNSUPPORTED)
    assert b"" != eng.L.rph_last_error()
    # a single image needs no image_stride at all
    assert f(eng.ctx, img.ctypes.data, 1, 64, 64, 3, 192, 0, h32.ctypes.data, None, None, None, None) == _lib.RPH_OK


def test_pdq_random_geometries_match_oracle(eng, oracle):
    """150 random geometries (5..900 px a side, 1/3/4 channels, padded strides): generic kernel below 513 px, pre-downsample above"""
    rng = np.random.default_rng(20261004)
    from rupphash_amd._lib import check
    for case in range(150):
        w = int(rng.integers(5, 900))
        h = int(rng.integers(5, 900))
        if case % 8 == 0:
            w = int(rng.choice([5, 6, 7, 63, 64, 65, 511, 512, 513]))
        if case % 8 == 1:
            h = int(rng.choice([5, 6, 7, 63, 64, 65, 511, 512, 513]))
        ch = int(rng.choice([1, 3, 4]))
        n = int(rng.integers(1, 4))
        row_stride = w * ch + int(rng.integers(0, 9))
        image_stride = row_stride * h + int(rng.integers(0, 33))
        buf, imgs = _strided_batch(rng, n, h, w, ch, row_stride, image_stride)
        # structure, so that hashes are not pure noise medians
        hashes = np.zeros((n, 32), np.uint8)
        quality = np.zeros(n, np.float32)
        coeffs = np.zeros((n, 256), np.float32)
        dihedral = np.zeros((n, 8, 32), np.uint8)
        valid = np.zeros(n, np.uint8)
        check(eng.L.rph_pdq_hash_batch(eng.ctx, buf.ctypes.data, n, w, h, ch, row_stride, image_stride, hashes.ctypes.data,
                                       quality.ctypes.data, coeffs.ctypes.data, dihedral.ctypes.data, valid.ctypes.data), "rph_pdq_hash_batch")
        for k in range(n):
            rc, c, q = oracle.pdq_features(imgs[k])
            assert rc == 0 and valid[k] == 1, (case, w, h, ch)
            assert np.array_equal(bits(coeffs[k]), bits(c)), (case, w, h, ch, k)
            assert bits(quality[k:k + 1])[0] == bits(np.float32(q))[()], (case, w, h, ch, k)
            assert np.array_equal(hashes[k], oracle.to_hash(c)), (case, w, h, ch, k)
            assert np.array_equal(dihedral[k], oracle.dihedral_hashes(c)), (case, w, h, ch, k)


def test_group_max_dist_matches_reference_rule(eng, oracle):
    """scanner.rs:2214-2241: max over members of the min distance to the pivot's 8 dihedral hashes (or to its plain hash)"""
    from rupphash_amd import scanner
    rng = np.random.default_rng(8)
    n = 40
    coeffs = rng.normal(0, 30, (n, 256)).astype(np.float32)
    hashes = np.stack([oracle.to_hash(c) for c in coeffs])
    # members 1..4 of group 0 are dihedral variants of file 0 with a few bits flipped
    dih0 = oracle.dihedral_hashes(coeffs[0])
    for k, slot in zip((1, 2, 3, 4), (1, 3, 5, 7)):
        h = dih0[slot].copy()
        for b in rng.choice(256, k, replace=False):
            h[b >> 3] ^= 1 << (b & 7)
        hashes[k] = h
    groups = [[0, 1, 2, 3, 4], [10, 11, 12], [20, 21]]
    pivots = [0, 11, None]
    got = scanner.group_max_dist(groups, hashes, pivots, coefficients=coeffs, engine=eng)
    want = []
    for g, p in zip(groups, pivots):
        if p is None:
            want.append(0)
            continue
        var = oracle.dihedral_hashes(coeffs[p])
        want.append(max(min(oracle.hamming256(v, hashes[m]) for v in var) for m in g))
    assert got == want and got[0] == 4
    # without features: plain distance to the pivot's hash
    got2 = scanner.group_max_dist(groups, hashes, pivots, engine=eng)
    assert got2 == [max(oracle.hamming256(hashes[p], hashes[m]) for m in g) if p is not None else 0 for g, p in zip(groups, pivots)]


def test_one_context_from_many_threads(eng, oracle):
    """The C ABI is thread-safe per context (the reference's entry points are called from arbitrary rayon threads): hashing,
    sweeping and grouping run concurrently from 6 threads on one context and every thread gets its own exact answer."""
    import threading
    rng = np.random.default_rng(99)
    imgs = rng.integers(0, 256, (6, 3, 120, 90, 3), dtype=np.uint8)
    hsets = [clustered_hashes(np.random.default_rng(700 + t), 1500, 40, 40) for t in range(6)]
    out = [None] * 6
    errs = []

    def work(t):
        try:
            for _ in range(3):
                r = eng.pdq_hash_batch(imgs[t], want_coeffs=True)
                e = eng.hamming_all_pairs(hsets[t], 32)
                g = eng.find_groups256(hsets[t], 31)
                out[t] = (r, e, g)
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(6):
        r, e, g = out[t]
        for k in range(3):
            rc, c, q = oracle.pdq_features(imgs[t][k])
            assert np.array_equal(bits(r["coeffs"][k]), bits(c)) and np.array_equal(r["hash"][k], oracle.to_hash(c))
        assert edge_set(e) == sorted(map(tuple, oracle.all_pairs256(hsets[t], 32).tolist()))
        assert g == oracle.find_groups(oracle.KIND_PDQ, hsets[t], 31)


def test_generic_path_on_two_caller_streams(eng, oracle):
    """rph_pdq_hash_batch_dev is asynchronous on the caller's stream; the generic kernels share one scratch per context, so launches
    on different streams must be ordered behind each other (event chaining), not race on it."""
    rng = np.random.default_rng(31)
    sets = [rng.integers(0, 256, (24, 200, 160, 3), dtype=np.uint8), rng.integers(0, 256, (24, 200, 160, 3), dtype=np.uint8)]
    streams = [eng.stream_create(), eng.stream_create()]
    d_px = [eng.dev_alloc(s.nbytes) for s in sets]
    d_h = [eng.dev_alloc(24 * 32) for _ in sets]
    try:
        for k in range(2):
            eng.dev_upload(d_px[k], sets[k])
        eng.synchronize()
        for rep in range(20):  # interleave launches on the two streams without any host synchronisation
            for k in range(2):
                eng.pdq_hash_batch_dev(d_px[k], 24, 160, 200, 3, d_h[k], stream=streams[k])
        for st in streams:
            eng.stream_synchronize(st)
        for k in range(2):
            got = np.zeros((24, 32), np.uint8)
            eng.dev_download(got, d_h[k])
            for i in range(24):
                rc, c, _ = oracle.pdq_features(sets[k][i])
                assert np.array_equal(got[i], oracle.to_hash(c)), (k, i)
    finally:
        for p in d_px + d_h:
            eng.dev_free(p)
        for st in streams:
            eng.stream_destroy(st)
