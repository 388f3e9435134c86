import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle, jpeg_util as ju
from rupphash_amd import Engine
eng=Engine(0)
for mode in ['L','RGB']:
  for fl in (0,1):
    data=ju.pillow_jpeg(ju.make_image(32,16,mode),quality=90,subsampling=0) if mode=='RGB' else ju.pillow_jpeg(ju.make_image(32,16,mode),quality=90)
    got=eng.jpeg_decode(data,fl).astype(int); ref=oracle.jpeg_decode(data,fl).astype(int)
    d=(got!=ref)
    if mode=='RGB': d=d.any(axis=2)
    print(mode,fl,'ndiff',d.sum())
    print(d[:8,:16].astype(int))
    if mode=='L': print((got-ref)[:8,:8])
