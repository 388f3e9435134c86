"""64-bit pHash bit operations of /root/reference/src/phash.rs:137-255 (scalar integer, C ABI).

DctPhash::hash_image itself (rustdct + image::resize, third-party arithmetic) is outside the
hot path and not provided.
"""
import ctypes as C

from . import _lib


def rotate_hash_90(h):
    return int(_lib.load().rph_phash_rotate_90(h))


def rotate_hash_180(h):
    return int(_lib.load().rph_phash_rotate_180(h))


def rotate_hash_270(h):
    return int(_lib.load().rph_phash_rotate_270(h))


def flip_hash_horizontal(h):
    return int(_lib.load().rph_phash_flip_horizontal(h))


def calculate_rotation_invariant_hash(h):
    return int(_lib.load().rph_phash_rotation_invariant(h))


def generate_dihedral_hashes(h):
    out = (C.c_uint64 * 8)()
    _lib.load().rph_phash_dihedral(h, out)
    return [int(x) for x in out]
