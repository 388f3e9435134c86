"""Differential fuzz of the JPEG path: the same random files through rph_jpeg_pdq_hash_batch with the Huffman streams decoded by the
host threads and by the device walk (one file per lane) must give the same bytes (hash, quality bit pattern, 256 coefficients), in both
arithmetic flavours; files with random damage inside their entropy segments, and truncated files, must never hang or fault either path -- nor the
segment synchronisation (rph_jpeg_set_segments(0, 64) / (0, 1024)) -- their results are unspecified.  Files: random sizes 1..700 px, gray and colour, 4:4:4 / 4:2:2 / 4:2:0 from Pillow (baseline and progressive, optimised
tables, restart intervals), 4:4:0 / one scan per component / 16-bit tables from tests/jpeg_util.encode_baseline, progressive files with
scan scripts of their own from tests/jpeg_util.encode_progressive, content from smooth to
pure noise (long Huffman codes, ZRLs)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpeg_util as ju  # noqa: E402
from rupphash_amd.engine import Engine  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
eng = Engine(0)


def random_file():
    from PIL import Image

    w, h = (int(rng.integers(1, 700)), int(rng.integers(1, 700))) if rng.random() < 0.5 else (int(rng.integers(1, 64)), int(rng.integers(1, 64)))
    if w * h > 200_000:
        w, h = w // 2 + 1, h // 2 + 1
    kind = int(rng.integers(0, 4))
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)  # noise: every coefficient class, long codes
    elif kind == 1:
        a = np.array(ju.make_image(w, h, seed=int(rng.integers(1 << 30))))
    elif kind == 2:
        b = int(rng.choice([2, 8, 16, 33]))
        v = rng.integers(0, 256, (h // b + 1, w // b + 1, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.ascontiguousarray(v[yy // b, xx // b])
    else:
        a = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)  # flat: DC only, end-of-block runs
    if rng.random() < 0.3:
        samp = [((1, 1), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1)), ((2, 1), (2, 1), (2, 1))][int(rng.integers(0, 5))]
        return ju.encode_baseline(a, samp, float(rng.choice([0.1, 0.5, 1.0, 4.0])), int(rng.choice([0, 0, 1, 3, 17])), sixteen_bit_tables=bool(rng.random() < 0.2),
                                  interleaved=bool(rng.random() < 0.6))
    if rng.random() < 0.2 and w * h <= 40_000:  # progressive with a scan script of its own (the Python encoder is slow: small files)
        script = [ju.SCRIPT_LIBJPEG, ju.SCRIPT_SPECTRAL_ONLY, ju.SCRIPT_DEEP, ju.SCRIPT_REFINE_BEFORE_OTHER_BANDS, ju.SCRIPT_MANY_BANDS][int(rng.integers(0, 5))]
        samp = [((1, 1), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((2, 2), (1, 1), (1, 1))][int(rng.integers(0, 3))]
        return ju.encode_progressive(a, script, samp, float(rng.choice([0.1, 0.5, 1.0, 4.0])), long_codes=bool(rng.random() < 0.5))
    gray = rng.random() < 0.2
    im = Image.fromarray(a).convert("L") if gray else Image.fromarray(a)
    kw = dict(quality=int(rng.choice([5, 30, 60, 85, 95, 100])), progressive=bool(rng.random() < 0.25))
    if not gray:
        kw["subsampling"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        kw["optimize"] = True
    elif rng.random() < 0.3:
        kw["restart_marker_blocks"] = int(rng.choice([1, 2, 7, 40]))
    return ju.pillow_jpeg(im, **kw)


t_end = time.time() + budget
rounds = files_total = damaged_total = damaged_status_differs = damaged_both_ok = damaged_hash_differs = 0
while time.time() < t_end:
    files = [random_file() for _ in range(160)]
    flavour = rounds & 1
    eng.jpeg_set_entropy(0)
    host = eng.jpeg_pdq_hash_batch(files, flavour=flavour, threads=8, want_coeffs=True)
    eng.jpeg_set_entropy(1)
    dev = eng.jpeg_pdq_hash_batch(files, flavour=flavour, threads=8, want_coeffs=True)
    for key in ("hash", "valid", "status"):
        assert np.array_equal(host[key], dev[key]), (seed, rounds, key, np.argwhere(host[key] != dev[key])[:4].tolist())
    assert np.array_equal(host["quality"].view(np.uint32), dev["quality"].view(np.uint32)), (seed, rounds, "quality")
    assert np.array_equal(host["coeffs"].view(np.uint32), dev["coeffs"].view(np.uint32)), (seed, rounds, "coeffs")
    assert not host["status"].any(), (seed, rounds, "a generated file was refused", host["status"].nonzero()[0][:4].tolist())
    # damage: bytes inside the entropy-coded part (behind the first SOS) overwritten at random
    bad = []
    for f in files[:80]:
        sos = f.find(b"\xff\xda")
        if sos < 0 or len(f) - sos < 40:
            continue
        b = bytearray(f)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(sos + 14, len(b) - 2))] = int(rng.integers(0, 256))
        bad.append(bytes(b))
    if bad:
        res = []
        for mode in (0, 1):
            eng.jpeg_set_entropy(mode)
            out = eng.jpeg_pdq_hash_batch(bad, flavour=flavour, threads=8)
            assert out["hash"].shape == (len(bad), 32)
            res.append(out)
        # how far the two decoders agree on damaged streams (reported, not required: what a decoder makes of a stream that breaks
        # T.81 is its own business, but the fewer differences the better)
        same_status = res[0]["status"] == res[1]["status"]
        both_ok = (res[0]["status"] == 0) & (res[1]["status"] == 0)
        damaged_status_differs += int((~same_status).sum())
        damaged_both_ok += int(both_ok.sum())
        damaged_hash_differs += int((both_ok & (res[0]["hash"] != res[1]["hash"]).any(axis=1)).sum())
        # the same damaged files, and truncated ones, with their streams cut into segments that synchronise on the device (a chain that runs
        # past the end of a scan must neither be trusted nor read behind the stream); then the defaults again
        trunc = []
        for f in files[80:120]:
            sos = f.find(b"\xff\xda")
            if sos >= 0 and len(f) - sos > 60:
                trunc.append(f[: sos + 14 + int(rng.integers(1, len(f) - sos - 14))])
        for seg_bytes in (64, 1024):
            eng.jpeg_set_segments(0, seg_bytes)
            eng.jpeg_set_entropy(1)
            out = eng.jpeg_pdq_hash_batch(bad + trunc, flavour=flavour, threads=8)
            assert out["hash"].shape == (len(bad) + len(trunc), 32)
        eng.jpeg_set_segments()
        damaged_total += 2 * len(trunc)
    files_total += len(files)
    damaged_total += len(bad)
    rounds += 1
eng.jpeg_set_entropy(2)
eng.close()
print(f"fuzz_jpeg seed {seed}: {files_total} random files agree bit for bit between host and device entropy decoding (both flavours); "
      f"{damaged_total} damaged files survived both paths ({damaged_both_ok} of them decodable by both: {damaged_hash_differs} different hashes; {damaged_status_differs} with different status)")
