"""ctypes binding of librupphash_hip.so (include/rupphash.h).

The library is the product: if it is missing or cannot be loaded, importing a
function from here raises -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPH_LIB_PATH: another build of the same library (A/B timing of kernel variants, tools/ab.sh); never a different implementation
LIB_PATH = os.environ.get("RPH_LIB_PATH") or os.path.join(_HERE, "librupphash_hip.so")

RPH_OK = 0
RPH_ERR_INVALID_ARG = -1
RPH_ERR_NO_DEVICE = -2
RPH_ERR_HIP = -3
RPH_ERR_OOM = -4
RPH_ERR_UNSUPPORTED = -5
RPH_ERR_CAPACITY = -6

RPH_JPEG_ZUNE = 0
RPH_JPEG_LIBJPEG = 1

RPH_EDGE_MIH_R1 = 0x8000
RPH_EDGE_PROBE_MASK = 0x01FF
RPH_EDGE_VARIANT_SHIFT = 9
RPH_EDGE_VARIANT_MASK = 0x0E00


class RphEdge(C.Structure):
    _fields_ = [("i", C.c_uint32), ("j", C.c_uint32), ("d", C.c_uint16), ("flags", C.c_uint16)]


class RphError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__(f"{where}: status {status} ({detail})")
        self.status = status


_vp, _u8p, _f32p, _u32p, _i32p, _u64p = (C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sz = C.c_size_t

# name -> (restype, argtypes); this table is what tests/test_abi.py checks against include/rupphash.h
SIGNATURES = {
    "rph_abi_version": (C.c_int, []),
    "rph_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rph_shutdown": (C.c_int, [_vp]),
    "rph_last_error": (C.c_char_p, []),
    "rph_status_string": (C.c_char_p, [C.c_int]),
    "rph_device_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "rph_synchronize": (C.c_int, [_vp]),
    "rph_pdq_hash_batch": (C.c_int, [_vp, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _sz, _sz, _u8p, _f32p,
                                     _f32p, _u8p, _u8p]),
    "rph_pdq_hash_batch_dev": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _sz, _sz, _vp, _vp, _vp,
                                         _vp, _vp, _vp]),
    "rph_pdq_hash_one": (C.c_int, [_vp, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, _sz, _u8p, _f32p, _f32p, _u8p]),
    "rph_pdq_batcher_config": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rph_pdq_batcher_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rph_pdq_hashes_from_coeffs": (C.c_int, [_vp, _f32p, C.c_uint32, _u8p, _u8p]),
    "rph_pdq_hashes_from_coeffs_dev": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp]),
    "rph_pdq_to_hash": (None, [_f32p, _u8p]),
    "rph_pdq_dihedral_one": (None, [_f32p, _u8p]),
    "rph_pdq_set_kernel": (C.c_int, [_vp, C.c_int]),
    "rph_hamming_set_kernel": (C.c_int, [_vp, C.c_int]),
    "rph_hamming_prefix_dwords": (C.c_int, [C.c_uint32, C.c_int]),
    "rph_pdq_target_dimensions": (None, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rph_hamming_distance256": (C.c_uint32, [_u8p, _u8p]),
    "rph_hamming_distance64": (C.c_uint32, [C.c_uint64, C.c_uint64]),
    "rph_get_chunk256": (C.c_uint16, [_u8p, C.c_uint32]),
    "rph_get_chunk64": (C.c_uint16, [C.c_uint64, C.c_uint32]),
    "rph_hamming_all_pairs": (C.c_int, [_vp, _u8p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64,
                                        C.POINTER(C.c_uint64)]),
    "rph_hamming_all_pairs_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, _vp]),
    "rph_hamming_variant_pairs": (C.c_int, [_vp, _u8p, C.c_uint32, _u8p, _u8p, C.c_uint64, C.c_uint32, C.c_uint32,
                                            C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rph_hamming_variant_pairs_dev": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32,
                                                C.c_uint32, _vp, C.c_uint64, _vp, _vp]),
    "rph_find_groups256": (C.c_int, [_vp, _u8p, C.c_uint64, C.c_uint32, _u32p, _u32p, C.POINTER(C.c_uint32)]),
    "rph_hamming_all_pairs64": (C.c_int, [_vp, _u64p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64,
                                          C.POINTER(C.c_uint64)]),
    "rph_hamming_all_pairs64_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _vp, _vp]),
    "rph_find_groups64": (C.c_int, [_vp, _u64p, C.c_uint64, C.c_uint32, _u32p, _u32p, C.POINTER(C.c_uint32)]),
    "rph_find_groups_from_edges": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _u32p, _u32p, C.POINTER(C.c_uint32)]),
    "rph_group_files_pdq": (C.c_int, [_vp, _u8p, _f32p, _u8p, _i32p, C.c_uint64, C.c_uint32, _u32p, _u32p,
                                      C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "rph_union_find_groups": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _u32p, _u32p, C.POINTER(C.c_uint32)]),
    "rph_is_low_pdq_quality": (C.c_int, [C.c_int32]),
    "rph_mih_build256": (C.c_int, [_vp, _u8p, C.c_uint64, _u32p, _u32p]),
    "rph_mih_build64": (C.c_int, [_vp, _u64p, C.c_uint64, _u32p, _u32p]),
    "rph_multi_init": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rph_multi_shutdown": (C.c_int, [_vp]),
    "rph_multi_size": (C.c_int, [_vp]),
    "rph_multi_ctx": (C.c_void_p, [_vp, C.c_int]),
    "rph_multi_hamming_all_pairs": (C.c_int, [_vp, _u8p, C.c_uint64, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rph_multi_hash_and_group": (C.c_int, [_vp, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _sz, _sz, C.c_uint32, _u8p, _f32p, _f32p, _u8p,
                                           _u32p, _u32p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "rph_multi_jpeg_hash_and_group": (C.c_int, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, _u8p, _f32p, _f32p, _u8p,
                                              _i32p, _u32p, _u32p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "rph_multi_group_files_pdq": (C.c_int, [_vp, _u8p, _f32p, _u8p, _i32p, C.c_uint64, C.c_uint32, _u32p, _u32p, C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_uint64)]),
    "rph_hash_record_encode": (None, [_u8p, _u8p]),
    "rph_hash_record_decode": (C.c_int, [_u8p, _sz, _u8p]),
    "rph_hash_records_encode": (None, [_u8p, _sz, _u8p]),
    "rph_hash_records_decode": (_sz, [_u8p, _sz, _u8p, _u8p]),
    "rph_coeff_record_size": (_sz, [_sz]),
    "rph_coeff_record_encode": (_sz, [_f32p, _sz, _u8p, _sz]),
    "rph_coeff_record_decode": (C.c_int, [_u8p, _sz, _f32p, _sz, C.POINTER(C.c_size_t)]),
    "rph_jpeg_info": (C.c_int, [C.c_char_p, _sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rph_jpeg_coefficients": (C.c_int, [C.c_char_p, _sz, _u32p, _vp, _vp, _sz, C.POINTER(C.c_uint64)]),
    "rph_jpeg_decode": (C.c_int, [_vp, C.c_char_p, _sz, C.c_int, _u8p]),
    "rph_jpeg_pdq_hash_one": (C.c_int, [_vp, C.c_char_p, _sz, C.c_int, _u8p, _f32p, _f32p, _u8p]),
    "rph_jpeg_set_entropy": (C.c_int, [_vp, C.c_int]),
    "rph_jpeg_set_segments": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rph_jpeg_release": (C.c_int, [_vp]),
    "rph_jpeg_pdq_hash_batch": (C.c_int, [_vp, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_uint32, C.c_int, C.c_uint32, _u8p, _f32p, _f32p,
                                          _u8p, _u8p, _i32p]),
    "rph_phash_rotate_90": (C.c_uint64, [C.c_uint64]),
    "rph_phash_rotate_180": (C.c_uint64, [C.c_uint64]),
    "rph_phash_rotate_270": (C.c_uint64, [C.c_uint64]),
    "rph_phash_flip_horizontal": (C.c_uint64, [C.c_uint64]),
    "rph_phash_rotation_invariant": (C.c_uint64, [C.c_uint64]),
    "rph_phash_dihedral": (None, [C.c_uint64, _u64p]),
    "rph_synth_images_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _vp]),
    "rph_synth_hashes_dev": (C.c_int, [_vp, _vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "rph_read_stream_dev": (C.c_int, [_vp, _vp, _sz, _vp]),
    "rph_dev_alloc": (C.c_int, [_vp, _sz, C.POINTER(C.c_void_p)]),
    "rph_dev_free": (C.c_int, [_vp, _vp]),
    "rph_dev_upload": (C.c_int, [_vp, _vp, _vp, _sz]),
    "rph_dev_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "rph_dev_memset": (C.c_int, [_vp, _vp, C.c_int, _sz, _vp]),
    "rph_stream_create": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "rph_stream_synchronize": (C.c_int, [_vp, _vp]),
    "rph_stream_destroy": (C.c_int, [_vp, _vp]),
    "rph_event_create": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "rph_event_record": (C.c_int, [_vp, _vp, _vp]),
    "rph_event_elapsed_ms": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    "rph_event_destroy": (C.c_int, [_vp, _vp]),
    "rph_stream": (C.c_void_p, [_vp]),
}

_lib = None


def _one_hip_runtime():
    """Keep ONE HIP runtime in the process.  librupphash_hip.so needs `libamdhip64.so.7` (resolved from /opt/rocm); a PyTorch
    wheel ships its own copy and asks for it as `libamdhip64.so`, which the loader does not match against an already loaded
    /opt/rocm one -- the second runtime then finds the device taken ("No HIP GPUs are available").  If PyTorch is installed but
    not imported yet, its copy is loaded first (by soname it then also satisfies this library), so a later `import torch`
    shares it.  RPH_HIP_RUNTIME=system skips this (processes that never import torch)."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("RPH_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # fall back to the system runtime


def load():
    """Load librupphash_hip.so (built by `make -C rupphash_amd/csrc` or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C rupphash_amd/csrc` (hipcc, --offload-arch=gfx950). "
                "rupphash_amd has no CPU fallback.")
        _one_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export what the header declares
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(status, where):
    if status != RPH_OK:
        L = load()
        raise RphError(status, where, (L.rph_last_error() or b"").decode() or L.rph_status_string(status).decode())
