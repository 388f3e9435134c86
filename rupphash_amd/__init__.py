"""rupphash_amd -- MI355X (gfx950) engine for the PDQ-hash + 256-bit Hamming-grouping hot path of
Safari77/rupphash.  The compute lives in librupphash_hip.so (hand-written HIP behind a C ABI,
include/rupphash.h); this package is the Python-side mirror of the reference's module interface:

    rupphash_amd.pdqhash      <->  src/pdqhash.rs
    rupphash_amd.hamminghash  <->  src/hamminghash.rs
    rupphash_amd.phash        <->  src/phash.rs (bit operations)
    rupphash_amd.scanner      <->  src/scanner.rs:1588-1832 (grouping only)
    rupphash_amd.dist         multi-GPU sharding (one process per GPU, RCCL all-gather of hashes)

There is no CPU fallback: every entry point needs the built library and a gfx950 device.
"""
from ._lib import LIB_PATH, RphError, load  # noqa: F401
from .engine import EDGE_DTYPE, Engine, MultiEngine, default_engine  # noqa: F401

__all__ = ["Engine", "MultiEngine", "default_engine", "EDGE_DTYPE", "RphError", "LIB_PATH", "load"]
