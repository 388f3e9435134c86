"""tools/soak_jpeg.py [repetitions] -- the JPEG path under repetition: a mixed call (photo-sized baseline files with and without restart markers,
progressive photos, 512x512 baseline and progressive files, small files of odd sizes; ~3 000 files) through the device walks again and again,
from alternating thread counts, must return the same bytes every time, and the host decoder's.  The device side is full of things that could
race if they were wrong: segments that publish their exits for their neighbours, atomics on coefficient dwords and mask words, two chunk
lanes sharing the device."""
import io
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from PIL import Image

import jpeg_util as ju
from rupphash_amd import Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(0)
im = Image.open(os.path.join(ROOT, "tests", "golden", "bench.jpg"))
files = []
for k in range(12):
    crop = im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2)))
    for kw in ({}, {"restart_marker_rows": 1}, {"progressive": True}):
        buf = io.BytesIO()
        crop.save(buf, "JPEG", quality=90, subsampling=2, **kw)
        files.append(buf.getvalue())
imgs = eng.synth_images(0, 12)
for k in range(12):
    for kw in ({}, {"progressive": True}, {"subsampling": 0, "quality": 95}):
        buf = io.BytesIO()
        Image.fromarray(imgs[k]).save(buf, "JPEG", **{"quality": 85, "subsampling": 2, **kw})
        files.append(buf.getvalue())
for k, (w, h) in enumerate([(33, 47), (7, 5), (100, 37), (640, 400), (129, 65), (16, 16)]):
    files.append(ju.pillow_jpeg(ju.make_image(w, h, seed=k), quality=80, progressive=bool(k & 1)))
    files.append(ju.encode_progressive(np.array(ju.make_image(w, h, seed=50 + k)), ju.SCRIPT_REFINE_BEFORE_OTHER_BANDS, ((2, 2), (1, 1), (1, 1)), 0.5, long_codes=True))
n = 3000
batch = eng.jpeg_file_list([files[k % len(files)] for k in range(n)])
eng.jpeg_set_entropy(0)
ref = eng.jpeg_pdq_hash_batch(batch, threads=16, want_quality=True, want_coeffs=True)
assert ref["valid"].all()
eng.jpeg_set_entropy(1)
for it in range(reps):
    eng.jpeg_set_segments(65536 if it % 3 else 0, [1024, 256, 4096][it % 3])
    out = eng.jpeg_pdq_hash_batch(batch, threads=[16, 3, 8][it % 3], want_quality=True, want_coeffs=True)
    assert not out["status"].any() and np.array_equal(out["valid"], ref["valid"]), it
    assert np.array_equal(out["hash"], ref["hash"]), f"repetition {it}: hashes differ"
    assert np.array_equal(out["coeffs"].view(np.uint32), ref["coeffs"].view(np.uint32)), f"repetition {it}: coefficients differ"
    assert np.array_equal(out["quality"].view(np.uint32), ref["quality"].view(np.uint32)), it
eng.jpeg_set_segments(65536, 1024)
eng.jpeg_set_entropy(2)
print(f"JPEG: {reps} repetitions of {n} mixed files ({len(files)} distinct) identical to the host decoder's results")
eng.close()
