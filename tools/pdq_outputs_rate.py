"""Throughput of the fused PDQ kernel by requested outputs (hash only / + quality + coefficients / + 8 dihedral hashes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rupphash_amd.engine import Engine

eng = Engine(0)
n = 50_000
d_px = eng.dev_alloc(n * 786432)
eng.synth_images_dev(d_px, 0, n, 512, 512)
d_h, d_q, d_c, d_d, d_v = (eng.dev_alloc(n * 32), eng.dev_alloc(n * 4), eng.dev_alloc(n * 1024), eng.dev_alloc(n * 256), eng.dev_alloc(n))
for name, kw in (("hash only", {}), ("hash + quality + coefficients + valid", dict(d_quality=d_q, d_coeffs=d_c, d_valid=d_v)),
                 ("all outputs incl. 8 dihedral hashes", dict(d_quality=d_q, d_coeffs=d_c, d_valid=d_v, d_dihedral=d_d))):
    eng.pdq_hash_batch_dev(d_px, n, 512, 512, 3, d_h, **kw)
    a, b = eng.event(), eng.event()
    eng.event_record(a)
    for _ in range(5):
        eng.pdq_hash_batch_dev(d_px, n, 512, 512, 3, d_h, **kw)
    eng.event_record(b)
    eng.synchronize()
    ms = eng.event_elapsed_ms(a, b) / 5
    print(f"{name:45s} {n / ms / 1e3:.3f} M img/s ({ms:.2f} ms per {n})")
