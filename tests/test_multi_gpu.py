"""The in-library multi-GPU form (rph_multi: one context per device + an RCCL communicator, include/rupphash.h) on the devices
this box has: with one device the all-gather is a one-rank collective and the sweep takes part 0 of 1 -- the same code path as
with eight apart from the device count -- and every result must equal the single-context entry points.
Second form: devices = [0, 0, 0], three ranks on the one GPU (the library's rehearsal mode: RCCL refuses two ranks on one device, so
the collective is carried by copies with ncclAllGather's contract): unequal shards padded and compacted, an empty shard, the
part / nparts shares of the sweeps and one host thread per rank all run as they would on a node."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["one device", "three ranks on one device"])
def multi(request):
    from rupphash_amd import MultiEngine

    m = MultiEngine(n_devices=1) if request.param == "one device" else MultiEngine(devices=[0, 0, 0])
    yield m
    m.close()


def test_multi_context_is_a_working_engine(multi, oracle):
    assert multi.size() in (1, 3)
    eng = multi.engine(multi.size() - 1)
    name, cus, _ = eng.device_info()
    assert "gfx950" in name and cus > 0
    h = oracle.synth_hashes(0, 3000, 3000, n_clusters=20)
    got = sorted((int(e["i"]), int(e["j"]), int(e["d"])) for e in eng.hamming_all_pairs(h, 32))
    assert got == sorted(map(tuple, oracle.all_pairs256(h, 32).tolist()))


@pytest.mark.parametrize("n,thr", [(5000, 32), (40001, 40), (2, 10)])
def test_multi_all_pairs_equals_single_context(multi, oracle, n, thr):
    h = oracle.synth_hashes(0, n, n, n_clusters=min(50, n // 10))
    got = sorted((int(e["i"]), int(e["j"]), int(e["d"])) for e in multi.hamming_all_pairs(h, thr))
    assert got == sorted(map(tuple, oracle.all_pairs256(h, thr).tolist()))


def test_multi_hash_and_group_equals_oracle_production_grouping(multi, oracle):
    first, n, sim = 900, 300, 32  # images 998/999 are a near-duplicate pair
    imgs = oracle.synth_images(first, n)
    res = multi.hash_and_group(imgs, sim, want_coeffs=True)
    hashes, qual, coeffs = oracle.pdq_batch_rgb(imgs, want_coeffs=True)
    assert np.array_equal(res["hash"], hashes) and res["valid"].all()
    assert np.array_equal(res["coeffs"].view(np.uint32), coeffs.view(np.uint32))
    assert np.array_equal(res["quality"].view(np.uint32), qual.view(np.uint32))
    dih = np.stack([oracle.dihedral_hashes(c) for c in coeffs])
    stored = np.clip(np.floor(qual * np.float32(100.0) + np.float32(0.5)), 0, 100).astype(np.int32)
    edges, groups = oracle.group_pdq(hashes, sim, variants=dih, quality=stored)
    assert res["groups"] == groups and res["comparison_count"] == len(edges)
    assert [998 - first, 999 - first] in res["groups"]
    # low-quality images pair only at distance 0 (scanner.rs:1699,1721): flat images have quality 0
    flat = np.full((6, 512, 512, 3), 128, np.uint8)
    flat[3:] += 1
    r2 = multi.hash_and_group(flat, 10)
    fh, fq, fc = oracle.pdq_batch_rgb(flat, want_coeffs=True)
    e2, g2 = oracle.group_pdq(fh, 10, variants=np.stack([oracle.dihedral_hashes(c) for c in fc]), quality=np.zeros(6, np.int32))
    assert r2["groups"] == g2 and r2["comparison_count"] == len(e2)
    # too small to hash: every file is None, no groups (pdqhash.rs:167-169)
    r3 = multi.hash_and_group(np.zeros((4, 4, 4, 3), np.uint8), 10)
    assert not r3["valid"].any() and r3["groups"] == []


@pytest.mark.parametrize("sim", [0, 31, 40, 63])
def test_multi_group_files_pdq_equals_single_context(multi, oracle, sim):
    rng = np.random.default_rng(50 + sim)
    n = 4000
    coeffs = rng.normal(0, 20, (n, 256)).astype(np.float32)
    for k in range(0, 600, 3):  # near duplicates and rotated copies
        coeffs[k + 1] = coeffs[k] + rng.normal(0, 0.4, 256).astype(np.float32)
        c = coeffs[k].reshape(16, 16).T.copy()
        coeffs[k + 2] = c.reshape(256)
    has = (rng.random(n) > 0.1).astype(np.uint8)
    quality = rng.choice([-1, 10, 49, 50, 100], n).astype(np.int32)
    eng = multi.engine(0)
    hashes, _ = eng.pdq_hashes_from_coeffs(coeffs, want_hash=True, want_dihedral=False)
    want = eng.group_files_pdq(hashes, sim, coeffs=coeffs, has_features=has, quality=quality)
    assert multi.group_files_pdq(hashes, sim, coeffs=coeffs, has_features=has, quality=quality) == want
    assert multi.group_files_pdq(hashes, sim) == eng.group_files_pdq(hashes, sim)  # hashes only: one variant per file
    dih = np.stack([oracle.dihedral_hashes(c) for c in coeffs])
    e, g = oracle.group_pdq(hashes, sim, variants=dih, has_features=has, quality=quality)
    assert want == (g, len(e))


def test_multi_init_rejects_a_missing_device():
    from rupphash_amd import MultiEngine, RphError

    with pytest.raises(RphError):
        MultiEngine(devices=[0, 99])


def test_multi_jpeg_hash_and_group_equals_decode_then_group_on_the_cpu(multi, oracle):
    """scan-then-group from JPEG FILES: near duplicates (the same picture at two qualities, its rotated copy), unrelated pictures, a file
    that cannot be decoded and one below 5 px; groups are indices into the caller's file list"""
    import jpeg_util as ju
    from PIL import Image

    pics = [ju.make_image(200 + 16 * (k % 4), 160, seed=100 + k) for k in range(12)]
    files = []
    for k, im in enumerate(pics):
        files.append(ju.pillow_jpeg(im, quality=90, subsampling=2))
        if k < 6:
            files.append(ju.pillow_jpeg(im, quality=70, subsampling=0, progressive=bool(k % 2)))   # re-coded
        if k < 3:
            files.append(ju.pillow_jpeg(im.transpose(Image.ROTATE_90), quality=85))                 # rotated: found through the dihedral variants
    files.insert(4, b"\xff\xd8 not a JPEG")
    files.insert(9, ju.pillow_jpeg(ju.make_image(4, 40)))
    sim = 40
    res = multi.jpeg_hash_and_group(files, sim, threads=4)
    dense, hashes, coeffs, stored = [], [], [], []
    for i, f in enumerate(files):
        try:
            rc, c, q = oracle.pdq_features(oracle.jpeg_decode(f, 0))
        except ValueError:
            assert res["status"][i] != 0 and not res["valid"][i]
            continue
        assert bool(res["valid"][i]) == (rc == 0)
        if rc != 0:
            continue
        assert np.array_equal(res["hash"][i], oracle.to_hash(c))
        dense.append(i)
        hashes.append(oracle.to_hash(c))
        coeffs.append(c)
        stored.append(int(np.clip(np.floor(np.float32(q) * np.float32(100.0) + np.float32(0.5)), 0, 100)))
    dih = np.stack([oracle.dihedral_hashes(c) for c in coeffs])
    edges, groups = oracle.group_pdq(np.stack(hashes), sim, variants=dih, quality=np.array(stored, np.int32))
    want = [[dense[m] for m in g] for g in groups]
    assert res["groups"] == want and res["comparison_count"] == len(edges)
    assert len(want) >= 6 and any(len(g) == 3 for g in want)  # picture + re-coded + rotated copy
