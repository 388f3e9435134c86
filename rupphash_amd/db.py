"""Value layouts of the reference's hash cache (/root/reference/src/db.rs), on top of the C ABI's host-scalar codecs.

    hash_db  value = [PDQ_ALGO_VERSION || 32-byte hash]                                    db.rs:1200-1211, read :683-696
    coeff_db value = [PDQ_ALGO_VERSION || postcard(CachedCoefficients{coefficients})]      db.rs:217-230, :1221-1231, read :742-755
                   = [2 || varint(len) || len x f32 little-endian]
For bulk import/export between an existing phdupes cache and the engine's flat arrays.  The XChaCha20-Poly1305 envelope
(db.rs:634-673) and LMDB itself stay with the host application.
"""
import ctypes as C

import numpy as np

from . import _lib

PDQ_ALGO_VERSION = 2
HASH_RECORD_BYTES = 33
COEFF_RECORD_BYTES = 1027


class Corrupted(ValueError):
    """the reference's lmdb::Error::Corrupted: a current-version record whose postcard payload does not parse"""


def encode_pdqhash(hash32):
    h = np.ascontiguousarray(hash32, np.uint8).reshape(32)
    out = np.zeros(HASH_RECORD_BYTES, np.uint8)
    _lib.load().rph_hash_record_encode(h.ctypes.data, out.ctypes.data)
    return out.tobytes()


def decode_pdqhash(record):
    """get_pdqhash's match: the 32-byte hash, or None for another algorithm version / length"""
    rec = np.frombuffer(bytes(record), np.uint8)
    out = np.zeros(32, np.uint8)
    ok = _lib.load().rph_hash_record_decode(rec.ctypes.data if len(rec) else None, len(rec), out.ctypes.data)
    return out if ok else None


def encode_pdqhashes(hashes):
    h = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
    out = np.zeros((len(h), HASH_RECORD_BYTES), np.uint8)
    _lib.load().rph_hash_records_encode(h.ctypes.data, len(h), out.ctypes.data)
    return out


def decode_pdqhashes(records):
    """records: (n, 33) uint8 -> (hashes (n, 32), present (n,) bool)"""
    r = np.ascontiguousarray(records, np.uint8).reshape(-1, HASH_RECORD_BYTES)
    hashes = np.zeros((len(r), 32), np.uint8)
    present = np.zeros(len(r), np.uint8)
    _lib.load().rph_hash_records_decode(r.ctypes.data, len(r), hashes.ctypes.data, present.ctypes.data)
    return hashes, present.astype(bool)


def encode_coefficients(coeffs):
    c = np.ascontiguousarray(coeffs, np.float32).reshape(-1)
    L = _lib.load()
    out = np.zeros(L.rph_coeff_record_size(len(c)), np.uint8)
    L.rph_coeff_record_encode(c.ctypes.data if len(c) else None, len(c), out.ctypes.data, len(out))
    return out.tobytes()


def decode_coefficients(record):
    """get_coefficients' match: float32 array (any length: the scanner keeps it only if len == 256), None for an empty
    record or another algorithm version; raises Corrupted where the reference returns lmdb::Error::Corrupted"""
    rec = np.frombuffer(bytes(record), np.uint8)
    cap = max(len(rec) // 4, 1)
    out = np.zeros(cap, np.float32)
    n = C.c_size_t()
    rc = _lib.load().rph_coeff_record_decode(rec.ctypes.data if len(rec) else None, len(rec), out.ctypes.data, cap, C.byref(n))
    if rc == 0:
        return None
    if rc != 1:
        raise Corrupted(_lib.load().rph_last_error().decode())
    return out[: n.value].copy()
