"""tools/jpeg_prog_timing.py -- 2 048 progressive photo-sized files (tests/golden/bench.jpg re-coded) through the device walk, two calls.
With a library built with -DRPH_PROG_TIMING (make -C rupphash_amd/csrc OUT=$PWD/gpurun_in/libT.so BUILD=build_T EXTRA_ALL=-DRPH_PROG_TIMING, then
RPH_LIB_PATH=gpurun_in/libT.so) lane 0 prints how long each scan of its file took."""
import io, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from PIL import Image
from rupphash_amd import Engine
eng = Engine(0)
im = Image.open(os.path.join(ROOT, "tests", "golden", "bench.jpg"))
prog = []
for k in range(16):
    buf = io.BytesIO()
    im.crop((k, k // 2, 1280 - (15 - k), 854 - (7 - k // 2))).save(buf, "JPEG", quality=90, subsampling=2, progressive=True)
    prog.append(buf.getvalue())
files = eng.jpeg_file_list([prog[k % 16] for k in range(2048)])
eng.jpeg_set_entropy(1)
for rep in range(2):
    t = time.perf_counter()
    out = eng.jpeg_pdq_hash_batch(files, threads=16)
    print("call: %.1f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
eng.close()
