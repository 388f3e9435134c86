"""tools/jpeg_trace.py -- 100 000 baseline JPEG files through rph_jpeg_pdq_hash_batch with the Huffman streams walked on the device, three calls.
RPH_JPEG_TRACE=1: synchronise after every device phase and print its time; RPH_JPEG_TRACE=2: host-side timestamps of the chunk pipeline."""
import sys, io, time; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from PIL import Image
from rupphash_amd import Engine
eng=Engine(0)
imgs=eng.synth_images(0,64,512,512)
base=[]
for k in range(64):
    b=io.BytesIO(); Image.fromarray(imgs[k]).save(b,'JPEG',quality=85,subsampling=2); base.append(b.getvalue())
files=[base[k%64] for k in range(100000)]
eng.jpeg_set_entropy(1)
for rep in range(3):
    t=time.perf_counter(); out=eng.jpeg_pdq_hash_batch(files,threads=16); dt=time.perf_counter()-t
    print("total %.1f ms -> %.0f files/s"%(dt*1e3, len(files)/dt))
