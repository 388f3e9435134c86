"""Mirror of the public interface of /root/reference/src/pdqhash.rs on top of the C ABI.

    PdqFeatures { coefficients: [f32; 256] }        pdqhash.rs:48-51
    PdqFeatures::to_hash() -> [u8; 32]              :59-61
    PdqFeatures::generate_dihedral_hashes()         :71-87   (8 x 32 bytes, reference slot order)
    generate_pdq_features(image) -> Option<(PdqFeatures, f32)>   :166-196
    generate_pdq(image) -> Option<([u8; 32], f32)>               :199-201
plus the batch forms the GPU wants.  `image` is a numpy uint8 array: (h, w) = Luma8 (borrowed as
is, :173), (h, w, 3) = Rgb8, (h, w, 4) = Rgba8 (alpha ignored, :279).  None is returned for
w or h < 5 exactly like the reference.  Image hashing runs on the GPU; to_hash / generate_dihedral_hashes of ONE
feature vector are the library's host-scalar functions (compare + bit operations, what the reference's per-file
call sites scanner.rs:1412, :1622 bind); the batch forms go to the GPU.
"""
import numpy as np

from . import _lib
from .engine import default_engine

MIN_HASHABLE_DIM = 5      # pdqhash.rs:17
DOWNSAMPLE_DIMS = 512     # pdqhash.rs:19
HASH_LENGTH = 32          # pdqhash.rs:23


class PdqFeatures:
    __slots__ = ("coefficients",)

    def __init__(self, coefficients):
        c = np.ascontiguousarray(coefficients, np.float32).reshape(256)
        self.coefficients = c

    def to_hash(self):
        out = np.zeros(32, np.uint8)
        _lib.load().rph_pdq_to_hash(self.coefficients.ctypes.data, out.ctypes.data)
        return out

    def generate_dihedral_hashes(self):
        out = np.zeros((8, 32), np.uint8)
        _lib.load().rph_pdq_dihedral_one(self.coefficients.ctypes.data, out.ctypes.data)
        return out


def generate_pdq_features(image, engine=None):
    image = np.asarray(image, np.uint8)
    out = (engine or default_engine()).pdq_hash_batch(image[None], want_quality=True, want_coeffs=True)
    if not out["valid"][0]:
        return None
    return PdqFeatures(out["coeffs"][0]), float(out["quality"][0])


def generate_pdq(image, engine=None):
    image = np.asarray(image, np.uint8)
    out = (engine or default_engine()).pdq_hash_batch(image[None], want_quality=True)
    if not out["valid"][0]:
        return None
    return out["hash"][0], float(out["quality"][0])


def generate_pdq_features_batch(images, engine=None, want_dihedral=False):
    """images: (n, h, w[, c]) uint8 -> dict(hash, quality, coeffs, dihedral, valid)."""
    return (engine or default_engine()).pdq_hash_batch(images, want_quality=True, want_coeffs=True, want_dihedral=want_dihedral)


def calculate_target_dimensions(w, h, max_dim=DOWNSAMPLE_DIMS):
    """pdqhash.rs:224-235"""
    import ctypes as C

    from . import _lib

    nw, nh = C.c_uint32(), C.c_uint32()
    _lib.load().rph_pdq_target_dimensions(w, h, max_dim, C.byref(nw), C.byref(nh))
    return nw.value, nh.value
