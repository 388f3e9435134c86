"""JPEG path on the GPU, through the C ABI: rph_jpeg_decode and rph_jpeg_pdq_hash_batch against the CPU oracle.

Both arithmetic flavours are compared with oracle/jpeg_ref.c byte for byte; the LIBJPEG flavour additionally with Pillow
(libjpeg-turbo) itself, which is the pin.  The ZUNE flavour (what the reference's zune-jpeg does, as recalled) is PARITY UNPINNED
against the Rust binary: GPU == oracle is all that can be asserted.  Hashes: decode with the oracle, hash with the oracle's PDQ,
compare with what the device produced from the file bytes (coefficients and quality bit patterns included)."""
import itertools
import os

import numpy as np
import pytest

import jpeg_util as ju

pytestmark = pytest.mark.gpu
PIL = pytest.importorskip("PIL")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = ["bench.jpg", "Prophecy_Has_Been_Fulfilled_1.jpg", "Prophecy_Has_Been_Fulfilled_2.jpg"]


@pytest.fixture(scope="module")
def eng():
    from rupphash_amd import Engine

    e = Engine(0)
    yield e
    e.close()


def _read(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


def _oracle_hash(oracle, px):
    """generate_pdq_features on decoded pixels: (valid, hash, quality, coeffs)"""
    rc, coeffs, q = oracle.pdq_features(px)
    if rc != 0:
        return False, None, None, None
    return True, oracle.to_hash(coeffs), np.float32(q), coeffs


@pytest.mark.parametrize("name", FILES)
@pytest.mark.parametrize("flavour", [0, 1])
def test_reference_files_decode_like_the_oracle(eng, oracle, name, flavour):
    data = _read(name)
    got = eng.jpeg_decode(data, flavour)
    assert np.array_equal(got, oracle.jpeg_decode(data, flavour))
    if flavour == 1:
        assert np.array_equal(got, ju.pillow_decode(data))  # libjpeg-turbo itself


def test_generated_streams_decode_like_the_oracle(eng, oracle):
    sizes = [(1, 1), (3, 5), (8, 8), (15, 17), (17, 33), (100, 37), (129, 65), (250, 3), (3, 250), (640, 360)]
    n = 0
    for (w, h), (mode, ss), prog, (q, opt, rst) in itertools.product(sizes, [("RGB", 0), ("RGB", 1), ("RGB", 2), ("L", 0)], [False, True],
                                                                     [(30, False, 0), (92, True, 0), (75, False, 3)]):
        kw = dict(quality=q, progressive=prog, optimize=opt)
        if mode == "RGB":
            kw["subsampling"] = ss
        if rst:
            kw["restart_marker_blocks"] = rst
        data = ju.pillow_jpeg(ju.make_image(w, h, mode, seed=n), **kw)
        for flavour in (0, 1):
            got = eng.jpeg_decode(data, flavour)
            assert np.array_equal(got, oracle.jpeg_decode(data, flavour)), ((w, h), mode, ss, prog, q, flavour)
        assert np.array_equal(got, ju.pillow_decode(data))
        n += 1
    layouts = [((2, 1), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((2, 1), (2, 1), (2, 1)), ((1, 2), (1, 2), (1, 2))]
    for (w, h), samp, (rst, qs, t16) in itertools.product([(16, 16), (33, 47), (100, 37), (7, 5), (2, 9)], layouts, [(0, 1.0, False), (5, 3.0, True)]):
        data = ju.encode_baseline(np.array(ju.make_image(w, h)), samp, qs, rst, sixteen_bit_tables=t16)
        for flavour in (0, 1):
            assert np.array_equal(eng.jpeg_decode(data, flavour), oracle.jpeg_decode(data, flavour)), ((w, h), samp, flavour)


@pytest.mark.parametrize("entropy", [0, 1])
@pytest.mark.parametrize("flavour", [0, 1])
def test_hash_batch_of_mixed_files_matches_decode_then_hash_on_the_cpu(eng, oracle, flavour, entropy):
    """files of many geometries, gray and colour, below 5 px, above 512 px, 512x512 (fused kernel), undecodable: one call.
    entropy = 1: the Huffman streams of the sequential files are walked on the device (one file per lane), the progressive ones by the host"""
    eng.jpeg_set_entropy(entropy)
    files = [_read(n) for n in FILES]
    synth = oracle.synth_images(998, 4)  # 512x512 RGB, one near-duplicate pair
    from PIL import Image

    for k in range(4):
        files.append(ju.pillow_jpeg(Image.fromarray(synth[k]), quality=90, subsampling=[0, 2, 2, 1][k]))
    specs = [(64, 64, "RGB", 2), (100, 37, "RGB", 1), (4, 30, "RGB", 0), (30, 4, "L", 0), (5, 5, "L", 0), (333, 512, "RGB", 2), (511, 509, "L", 0),
             (700, 300, "RGB", 2), (129, 65, "RGB", 0), (512, 512, "L", 0)]
    for i, (w, h, mode, ss) in enumerate(specs):
        kw = dict(quality=85, progressive=bool(i & 1))
        if mode == "RGB":
            kw["subsampling"] = ss
        files.append(ju.pillow_jpeg(ju.make_image(w, h, mode, seed=i), **kw))
    # restart intervals, one scan per component, 4:4:0, optimised tables, 16-bit tables: the device walk's special cases
    for i, (w, h, samp, rst, il) in enumerate([(97, 61, ((2, 2), (1, 1), (1, 1)), 3, True), (97, 61, ((2, 1), (1, 1), (1, 1)), 1, False),
                                              (64, 200, ((1, 2), (1, 1), (1, 1)), 0, False), (40, 40, ((1, 1), (1, 1), (1, 1)), 7, False)]):
        files.append(ju.encode_baseline(np.array(ju.make_image(w, h, seed=20 + i)), samp, 0.5 + i, rst, sixteen_bit_tables=(i == 3), interleaved=il))
    files.append(ju.pillow_jpeg(ju.make_image(200, 120, seed=31), quality=97, optimize=True, subsampling=2))
    files.append(ju.pillow_jpeg(ju.make_image(160, 90, seed=32), quality=60, restart_marker_blocks=2))
    files.insert(5, b"\xff\xd8 this is not a JPEG")
    files.append(ju.pillow_jpeg(ju.make_image(16, 16).convert("CMYK")))
    good = ju.pillow_jpeg(ju.make_image(80, 60, seed=33), quality=80)
    files.append(good[: len(good) * 2 // 3])  # truncated entropy segment: zeros are fed behind the end, on the host and on the device
    out = eng.jpeg_pdq_hash_batch(files, flavour=flavour, threads=4, want_quality=True, want_coeffs=True, want_dihedral=True)
    eng.jpeg_set_entropy(2)
    for i, data in enumerate(files):
        try:
            px = oracle.jpeg_decode(data, flavour)
        except ValueError:
            assert out["status"][i] != 0 and out["valid"][i] == 0 and not out["hash"][i].any(), i
            continue
        assert out["status"][i] == 0, i
        ok, h, q, c = _oracle_hash(oracle, px)
        assert bool(out["valid"][i]) == ok, i
        if ok:
            assert np.array_equal(out["hash"][i], h), i
            assert out["quality"][i].view(np.uint32) == q.view(np.uint32), i
            assert np.array_equal(out["coeffs"][i].view(np.uint32), np.asarray(c, np.float32).view(np.uint32)), i
            assert np.array_equal(out["dihedral"][i], oracle.dihedral_hashes(c)), i
    # the near-duplicate pair of the synthetic stripe survives JPEG coding as near duplicates
    assert oracle.hamming256(out["hash"][3], out["hash"][4]) <= 40  # files 3 and 4 are images 998 and 999 of the synthetic set


def test_device_entropy_on_a_few_thousand_files_equals_host_entropy(eng, oracle):
    """the automatic mode takes the device walk from 2048 sequential files: 3000 files (24 distinct, three geometries, progressive ones
    mixed in) give byte for byte what the host threads give, and the oracle's hashes"""
    from PIL import Image

    synth = oracle.synth_images(100, 16)
    base = [ju.pillow_jpeg(Image.fromarray(synth[k]), quality=75 + k, subsampling=2 * (k % 2), progressive=(k % 8 == 7)) for k in range(16)]
    base += [ju.pillow_jpeg(ju.make_image(200 + 8 * k, 120, seed=k), quality=85, restart_marker_blocks=(4 if k % 2 else 0)) for k in range(4)]
    base += [ju.encode_baseline(np.array(ju.make_image(64, 64, seed=40 + k)), ((1, 2), (1, 1), (1, 1)), 1.0, k, interleaved=bool(k % 2)) for k in range(4)]
    files = [base[(k * 7) % 24] for k in range(3000)]
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=16)
    eng.jpeg_set_entropy(2)
    auto = eng.jpeg_pdq_hash_batch(files, threads=16)
    assert host["valid"].all() and auto["valid"].all() and not auto["status"].any()
    assert np.array_equal(host["hash"], auto["hash"]) and np.array_equal(host["quality"].view(np.uint32), auto["quality"].view(np.uint32))
    for k in range(24):
        ok, h, q, _ = _oracle_hash(oracle, oracle.jpeg_decode(base[(k * 7) % 24], 0))
        assert ok and np.array_equal(auto["hash"][k], h)


def test_hash_batch_larger_than_one_chunk_and_thread_counts_agree(eng, oracle):
    """~600 files of 512x512 cross the chunk boundary (double-buffered slots); 1, 3 and 16 host threads give the same bytes"""
    from PIL import Image

    synth = oracle.synth_images(0, 24)
    base = [ju.pillow_jpeg(Image.fromarray(synth[k]), quality=80 + (k % 3) * 5, subsampling=2 * (k % 2), progressive=bool(k % 4 == 3)) for k in range(24)]
    files = [base[k % 24] for k in range(600)]
    ref = eng.jpeg_pdq_hash_batch(files, threads=16)
    assert ref["valid"].all() and not ref["status"].any()
    for k in range(24):
        ok, h, q, _ = _oracle_hash(oracle, oracle.jpeg_decode(base[k], 0))
        assert ok and np.array_equal(ref["hash"][k], h) and ref["quality"][k].view(np.uint32) == q.view(np.uint32)
    for k in range(600):
        assert np.array_equal(ref["hash"][k], ref["hash"][k % 24])
    for t in (1, 3):
        again = eng.jpeg_pdq_hash_batch(files[:100], threads=t)
        assert np.array_equal(again["hash"], ref["hash"][:100])


def test_forty_thousand_tiny_files_on_the_device_walk(eng, oracle):
    """more images than one reconstruction sub-batch may hold (the kernels index planes / images with blockIdx.y)"""
    base = [ju.pillow_jpeg(ju.make_image(24 + (k % 5) * 8, 16 + (k % 3) * 8, seed=k), quality=70 + k % 20, subsampling=k % 3) for k in range(30)]
    files = [base[(k * 11) % 30] for k in range(40000)]
    eng.jpeg_set_entropy(1)
    out = eng.jpeg_pdq_hash_batch(files, threads=16)
    eng.jpeg_set_entropy(2)
    assert out["valid"].all() and not out["status"].any()
    for k in range(30):
        ok, h, q, _ = _oracle_hash(oracle, oracle.jpeg_decode(base[(k * 11) % 30], 0))
        assert ok and np.array_equal(out["hash"][k], h)
    assert np.array_equal(out["hash"][:30 * 1000].reshape(1000, 30, 32), np.broadcast_to(out["hash"][:30], (1000, 30, 32)))


def test_one_file_per_call_from_many_threads(eng, oracle):
    """rph_jpeg_pdq_hash_one from 12 threads at once (the scan loop's pattern): same hashes, qualities and coefficients as the batch call;
    a too-small image gives None, an undecodable file raises"""
    import threading

    from rupphash_amd import RphError

    files = [ju.pillow_jpeg(ju.make_image(64 + 8 * (k % 9), 48 + 8 * (k % 5), "L" if k % 7 == 0 else "RGB", seed=k), quality=60 + k % 35,
                            progressive=bool(k % 4 == 1)) for k in range(96)]
    files[10] = ju.pillow_jpeg(ju.make_image(4, 30))          # below 5 px
    files[20] = b"\xff\xd8 nothing of a JPEG follows"          # not decodable
    ref = eng.jpeg_pdq_hash_batch(files, threads=4, want_coeffs=True)
    got, errors = [None] * len(files), []

    def worker(t):
        for i in range(t, len(files), 12):
            try:
                got[i] = eng.jpeg_pdq_hash_one(files[i], flavour=1 if i == 4 else 0)
            except RphError as e:
                got[i] = e
            except Exception as e:  # noqa: BLE001
                errors.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors
    for i in range(len(files)):
        if i == 10:
            assert got[i] is None and ref["valid"][i] == 0 and ref["status"][i] == 0
        elif i == 20:
            assert isinstance(got[i], RphError) and ref["status"][i] != 0
        elif i == 4:  # asked for the other flavour
            ok, h, _, _ = _oracle_hash(oracle, oracle.jpeg_decode(files[i], 1))
            assert ok and np.array_equal(got[i][0], h)
        else:
            h, q, c = got[i]
            assert np.array_equal(h, ref["hash"][i]) and np.float32(q).view(np.uint32) == ref["quality"][i].view(np.uint32), i
            assert np.array_equal(c.view(np.uint32), ref["coeffs"][i].view(np.uint32)), i


@pytest.mark.parametrize("seg_bytes", [64, 256, 1024])
def test_streams_without_markers_cut_into_segments(eng, oracle, seg_bytes):
    """streams without restart markers walked as segments that synchronise on the device: every sampling layout, gray, tiny and large
    files, optimised tables, noise (long codes) and flat images (MCUs of a few bits), forced on for every file (min 0) at three segment
    sizes; the results must be the host decoder's whatever the segmentation"""
    from PIL import Image

    files = []
    for k, (w, h, mode, ss) in enumerate([(640, 400, "RGB", 2), (333, 517, "RGB", 1), (200, 120, "RGB", 0), (97, 61, "L", 0), (1280, 854, "RGB", 2), (16, 16, "RGB", 2),
                                          (8, 8, "L", 0), (500, 40, "RGB", 2)]):
        kw = dict(quality=[50, 75, 90, 97][k % 4], optimize=bool(k % 3 == 1))
        if mode == "RGB":
            kw["subsampling"] = ss
        files.append(ju.pillow_jpeg(ju.make_image(w, h, mode, seed=90 + k), **kw))
    rng = np.random.default_rng(3)
    files.append(ju.pillow_jpeg(Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)), quality=95, subsampling=0))  # noise: long codes, long MCUs
    files.append(ju.pillow_jpeg(Image.fromarray(np.full((600, 800, 3), 77, np.uint8)), quality=90, subsampling=2))                 # flat: MCUs of ~30 bits
    files.append(ju.encode_baseline(np.array(ju.make_image(240, 160, seed=99)), ((1, 2), (1, 1), (1, 1)), 0.5, 0))                  # 4:4:0
    files.append(ju.encode_baseline(np.array(ju.make_image(240, 160, seed=98)), ((2, 1), (2, 1), (2, 1)), 2.0, 0))                  # chroma at luma resolution
    files = files * 3
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(1)
    eng.jpeg_set_segments(0, seg_bytes)
    dev = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_segments()  # (the defaults)
    eng.jpeg_set_entropy(2)
    assert dev["valid"].all() and not dev["status"].any()
    assert np.array_equal(dev["hash"], host["hash"]) and np.array_equal(dev["coeffs"].view(np.uint32), host["coeffs"].view(np.uint32))
    for k in (0, 4, 8, 9):
        ok, h, _, _ = _oracle_hash(oracle, oracle.jpeg_decode(files[k], 0))
        assert ok and np.array_equal(dev["hash"][k], h)


def test_a_launch_of_many_segment_lanes_builds_its_blocks_in_lds(eng, oracle):
    """from 131 072 lanes per launch the walk stages every block in LDS and writes it out whole, and a chunk whose files all have one scan
    is not zeroed first: 832 photo-sized files without restart markers (about 300 000 segments of 1 KB), twice -- the second call finds
    the first call's coefficients in the buffer -- with a smaller, different set in between; results must be the host decoder's"""
    from PIL import Image

    im = Image.open(os.path.join(GOLDEN, "bench.jpg"))
    im.load()
    crops = [ju.pillow_jpeg(im.crop((k % 8, k // 8, k % 8 + 1265, k // 8 + 850)), quality=90, subsampling=2) for k in range(32)]
    files = [crops[k % 32] for k in range(832)]
    other = [ju.pillow_jpeg(ju.make_image(1100 + 8 * k, 700, "RGB", seed=4000 + k), quality=85, subsampling=[2, 1, 0][k % 3]) for k in range(6)] * 60
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=16)
    host_other = eng.jpeg_pdq_hash_batch(other, threads=16)
    eng.jpeg_set_entropy(1)
    try:
        for batch, want in ((files, host), (other, host_other), (files, host)):
            dev = eng.jpeg_pdq_hash_batch(batch, threads=16)
            assert dev["valid"].all() and not dev["status"].any()
            assert np.array_equal(dev["hash"], want["hash"])
    finally:
        eng.jpeg_set_entropy(2)


def test_progressive_files_walk_on_the_device(eng, oracle):
    """progressive files (T.81 G.1.2: DC first / refinement, AC first with end-of-band runs, AC refinement with correction bits), one per
    lane through all their scans: the reference's own progressive files, every sampling layout, gray, tiny and photo-sized, noise (long
    runs of corrections) and flat images (long end-of-band runs); results must be the host decoder's, and the oracle's where it is asked"""
    from PIL import Image

    files = [_read(n) for n in FILES]
    for k, (w, h, mode, ss) in enumerate([(640, 400, "RGB", 2), (333, 517, "RGB", 1), (200, 120, "RGB", 0), (97, 61, "L", 0), (1280, 854, "RGB", 2), (16, 16, "RGB", 2),
                                          (8, 8, "L", 0), (500, 40, "RGB", 2), (5, 5, "RGB", 2), (1, 1, "L", 0), (17, 9, "RGB", 1), (512, 512, "RGB", 2), (512, 512, "L", 0)]):
        kw = dict(quality=[30, 50, 75, 90, 97][k % 5], progressive=True)
        if mode == "RGB":
            kw["subsampling"] = ss
        files.append(ju.pillow_jpeg(ju.make_image(w, h, mode, seed=190 + k), **kw))
    rng = np.random.default_rng(13)
    files.append(ju.pillow_jpeg(Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)), quality=95, subsampling=0, progressive=True))
    files.append(ju.pillow_jpeg(Image.fromarray(rng.integers(0, 256, (64, 64), dtype=np.uint8)), quality=100, progressive=True))
    files.append(ju.pillow_jpeg(Image.fromarray(np.full((600, 800, 3), 77, np.uint8)), quality=90, subsampling=2, progressive=True))
    files.append(ju.pillow_jpeg(ju.make_image(320, 240, seed=7), quality=85, subsampling=2))  # a sequential file between them
    files = files * 3
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(3)  # device for sequential files only: progressive ones by the host threads
    seq_only = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(1)
    dev = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(2)
    assert not dev["status"].any() and np.array_equal(dev["valid"], host["valid"])
    for other in (seq_only, dev):
        assert np.array_equal(other["hash"], host["hash"]) and np.array_equal(other["coeffs"].view(np.uint32), host["coeffs"].view(np.uint32))
    for k in (0, 1, 2, 3, 7, 16, 18):
        ok, h, _, _ = _oracle_hash(oracle, oracle.jpeg_decode(files[k], 0))
        assert ok and np.array_equal(dev["hash"][k], h), k
    # (the 256 f32 coefficients are compared bit for bit: below 513 px every pixel of the image carries weight in them, so a single wrong
    # DCT coefficient anywhere in such a file shows)


def test_progressive_scans_follow_their_producers_across_many_batches(eng, oracle):
    """all scans of a chunk are one launch in which a scan follows the scans it depends on block by block (progress words): several hundred
    files of libjpeg's script and of a script whose refinements come before other bands' first scans, in sizes from one block to 400 x 300
    -- full batches of 64, part batches, scans of very different lengths side by side -- must come out like the host decoder's, twice
    (the second call reuses every buffer, progress words included)"""
    files = []
    for k in range(150):
        w, h = [(400, 300), (64, 48), (8, 8), (233, 177), (120, 200)][k % 5]
        files.append(ju.pillow_jpeg(ju.make_image(w + k % 7, h + k % 3, "RGB", seed=900 + k), quality=[60, 85, 95][k % 3], subsampling=[2, 1, 0][k % 3], progressive=True))
    for k in range(70):
        files.append(ju.encode_progressive(np.asarray(ju.make_image(96 + k, 80, "RGB", seed=1300 + k)), ju.SCRIPT_REFINE_BEFORE_OTHER_BANDS, quality_scale=1.0))
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(1)
    try:
        for rep in range(2):
            dev = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
            assert not dev["status"].any() and np.array_equal(dev["valid"], host["valid"])
            assert np.array_equal(dev["hash"], host["hash"]) and np.array_equal(dev["coeffs"].view(np.uint32), host["coeffs"].view(np.uint32))
    finally:
        eng.jpeg_set_entropy(2)


def test_progressions_the_standard_does_not_describe_stay_with_the_host_decoder(eng, oracle):
    """the device walks the scans of a file side by side and applies refinements last, which equals scan-after-scan decoding only for
    progressions as T.81 G.1.1.1.1 describes them; a band that is first-coded twice, a first scan after its refinement, a refinement whose
    Ah is not the Al before (libjpeg: "bogus progression", decoded all the same) must give what the host decoder gives -- they are handed to it"""
    bogus = [
        [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 1), ((0,), 1, 63, 1, 0), ((0,), 1, 63, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)],   # luma first-coded again after its refinement
        [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 2), ((0,), 1, 63, 1, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)],                        # Ah = 1 after Al = 2
        [((0, 1, 2), 0, 0, 0, 1), ((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 63, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)],                    # DC first-coded twice
        [((0, 1, 2), 0, 0, 0, 0), ((0,), 1, 20, 0, 0), ((0,), 10, 63, 0, 0), ((1,), 1, 63, 0, 0), ((2,), 1, 63, 0, 0)],                       # overlapping bands
    ]
    files = [ju.encode_progressive(np.asarray(ju.make_image(120 + 8 * k, 88, "RGB", seed=5000 + k)), script) for k, script in enumerate(bogus)]
    files += [ju.encode_progressive(np.asarray(ju.make_image(96, 80, "RGB", seed=5100)), ju.SCRIPT_LIBJPEG)]
    files = files * 20
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(1)
    try:
        dev = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    finally:
        eng.jpeg_set_entropy(2)
    assert np.array_equal(dev["status"], host["status"]) and np.array_equal(dev["valid"], host["valid"])
    assert np.array_equal(dev["hash"], host["hash"]) and np.array_equal(dev["coeffs"].view(np.uint32), host["coeffs"].view(np.uint32))


def test_progressive_scan_scripts_on_the_device(eng, oracle):
    """the device walk of progressive files on what Pillow cannot write (tests/jpeg_util.encode_progressive): bands refined in another order
    than they were first coded (the masks a refinement scan reads were last changed by atomics of an earlier scan), successive approximation
    three bits deep, DC scans per component, twenty bands per component, per-scan tables with 16-bit codes.  The host decoder is the reference
    (itself checked against the oracle and libjpeg-turbo on these streams in test_oracle_jpeg.py); the oracle is asked directly on a sample."""
    import itertools

    scripts = [("libjpeg", ju.SCRIPT_LIBJPEG, False), ("spectral", ju.SCRIPT_SPECTRAL_ONLY, False), ("deep", ju.SCRIPT_DEEP, False),
               ("order", ju.SCRIPT_REFINE_BEFORE_OTHER_BANDS, False), ("many", ju.SCRIPT_MANY_BANDS, False), ("gray", ju.SCRIPT_GRAY, True)]
    files = []
    for (w, h), (name, script, gray), samp, (qs, lc) in itertools.product([(64, 48), (33, 47), (7, 5), (200, 136)], scripts,
                                                                          [((2, 2), (1, 1), (1, 1)), ((1, 1), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1))],
                                                                          [(0.3, False), (1.0, True), (4.0, False)]):
        if gray and samp != ((2, 2), (1, 1), (1, 1)):
            continue
        files.append(ju.encode_progressive(np.array(ju.make_image(w, h, "L" if gray else "RGB", seed=w + h)), script, samp, qs, gray=gray, long_codes=lc))
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    files.append(ju.encode_progressive(noise, ju.SCRIPT_DEEP, ((2, 2), (1, 1), (1, 1)), 0.1, long_codes=True))   # every coefficient nonzero: long correction runs
    files.append(ju.encode_progressive(np.full((96, 128, 3), 200, np.uint8), ju.SCRIPT_LIBJPEG, ((2, 2), (1, 1), (1, 1)), 1.0))  # flat: one end-of-band run per scan
    one_by_one = [((0, 1, 2), 0, 0, 0, 0)] + [((c,), k, k, 0, 0) for k in range(1, 64) for c in (0, 1, 2)]  # 190 scans: more than the device plan holds -> host threads
    files.append(ju.encode_progressive(np.array(ju.make_image(40, 24, seed=77)), one_by_one, ((1, 1), (1, 1), (1, 1)), 0.5))
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(1)
    dev = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(2)
    assert not dev["status"].any() and not host["status"].any() and np.array_equal(dev["valid"], host["valid"])
    assert np.array_equal(dev["hash"], host["hash"]) and np.array_equal(dev["coeffs"].view(np.uint32), host["coeffs"].view(np.uint32))
    for k in range(0, len(files), 17):
        ok, h, _, _ = _oracle_hash(oracle, oracle.jpeg_decode(files[k], 0))
        assert bool(dev["valid"][k]) == ok and (not ok or np.array_equal(dev["hash"][k], h)), k


def test_restart_intervals_get_a_lane_each(eng, oracle):
    """one-scan files with restart markers are walked by one lane per interval; 60 photos with an interval per MCU row are 3 000+
    lanes, so the automatic mode takes the device walk for them; hashes equal the host decoder's and the oracle's"""
    base = [ju.pillow_jpeg(ju.make_image(640 + 16 * k, 400 + 8 * k, seed=70 + k), quality=88, subsampling=2 if k % 2 else 0, restart_marker_rows=1) for k in range(6)]
    base += [ju.pillow_jpeg(ju.make_image(300, 200, "L", seed=80), quality=80, restart_marker_blocks=5)]
    base += [ju.encode_baseline(np.array(ju.make_image(120, 90, seed=81)), ((2, 2), (1, 1), (1, 1)), 1.0, 3)]  # interval = 3 MCUs, the last one short
    files = [base[k % len(base)] for k in range(64)]
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(2)
    auto = eng.jpeg_pdq_hash_batch(files, threads=8, want_coeffs=True)
    assert auto["valid"].all() and np.array_equal(auto["hash"], host["hash"])
    assert np.array_equal(auto["coeffs"].view(np.uint32), host["coeffs"].view(np.uint32))
    for k in range(len(base)):
        ok, h, _, _ = _oracle_hash(oracle, oracle.jpeg_decode(base[k], 0))
        assert ok and np.array_equal(auto["hash"][k], h)


def test_default_threads_and_release(eng, oracle):
    """n_threads = 0 (affinity mask / cgroup quota decide), and the cached buffers can be returned and come back on the next call"""
    files = [ju.pillow_jpeg(ju.make_image(96 + k, 64, seed=k), quality=80) for k in range(40)]
    a = eng.jpeg_pdq_hash_batch(files, threads=0)
    eng.jpeg_release()
    b = eng.jpeg_pdq_hash_batch(files, threads=2)
    assert a["valid"].all() and np.array_equal(a["hash"], b["hash"])
    ok, h, _, _ = _oracle_hash(oracle, oracle.jpeg_decode(files[7], 0))
    assert ok and np.array_equal(a["hash"][7], h)


def test_scanner_load_image_fast_mirror(eng, oracle):
    from rupphash_amd import scanner

    data = _read("bench.jpg")
    img = scanner.load_image_fast("x/y/bench.JPG", data, engine=eng)
    assert img.shape == (854, 1280, 3) and np.array_equal(img, oracle.jpeg_decode(data, 0))
    gray = ju.pillow_jpeg(ju.make_image(40, 30, "L"))
    assert scanner.load_image_fast("g.jpeg", gray, engine=eng).shape == (30, 40)
    with pytest.raises(ValueError):
        scanner.load_image_fast("file.png", data, engine=eng)  # other formats stay with the host's decoders


def test_header_that_announces_more_blocks_than_the_file_can_hold_is_refused(eng):
    """a 20000 x 20000 frame header on a file of a few hundred bytes (header-only decompression bomb): RPH_ERR_UNSUPPORTED before
    anything is allocated, in the batch call and in the one-file call"""
    import jpeg_util as ju

    f = bytearray(ju.pillow_jpeg(ju.make_image(16, 16), quality=50))
    sof = f.find(b"\xff\xc0")
    assert sof > 0
    f[sof + 5:sof + 9] = (20000).to_bytes(2, "big") + (20000).to_bytes(2, "big")  # height, width
    good = ju.pillow_jpeg(ju.make_image(64, 48), quality=80)
    out = eng.jpeg_pdq_hash_batch([bytes(f), good], threads=1)  # (a batch reports per file; a call of ONE file returns the file's status)
    assert not out["valid"][0] and out["status"][0] == -5 and out["valid"][1] and out["status"][1] == 0
    from rupphash_amd import RphError

    with pytest.raises(RphError) as e:
        eng.jpeg_pdq_hash_one(bytes(f))
    assert e.value.status == -5
