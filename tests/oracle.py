"""ctypes view of oracle/liboracle_ref.so (CPU restatement of the reference).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Never imported by rupphash_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liboracle_ref.so")

KIND_U64 = 0
KIND_PDQ = 1
REF_OK, REF_TOO_SMALL, REF_NEEDS_RESIZE = 0, 1, 2


def build(force=False):
    """(Re)build the oracle when missing or stale.  With RPH_ORACLE_NO_BUILD=1 (bench.py sets it) nothing is ever spawned:
    the library __graft_entry__.build() produced is used as it is, and a missing one is an error -- a `make` child of a process
    that holds the GPU, or that runs under a profiler's preload, would be an exec from a GPU-initialised process."""
    if os.environ.get("RPH_ORACLE_NO_BUILD") == "1" and not force:
        if not os.path.exists(_SO):
            raise ImportError(f"{_SO} is missing: run `python __graft_entry__.py` (build()) first; bench.py never builds the oracle itself")
        return _SO
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, f32p, u32p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32))
        L.rph_ref_hamming64.restype = C.c_uint32
        L.rph_ref_hamming64.argtypes = [C.c_uint64, C.c_uint64]
        L.rph_ref_hamming256.restype = C.c_uint32
        L.rph_ref_quality.restype = C.c_float
        L.rph_ref_mih_new.restype = C.c_void_p
        L.rph_ref_mih_new.argtypes = [C.c_int, C.c_void_p, C.c_uint32]
        L.rph_ref_mih_free.argtypes = [C.c_void_p]
        L.rph_ref_mih_bucket.restype = C.c_uint32
        L.rph_ref_mih_bucket.argtypes = [C.c_void_p, C.c_int, C.c_uint16, C.POINTER(u32p)]
        L.rph_ref_mih_offsets.restype = u32p
        L.rph_ref_mih_offsets.argtypes = [C.c_void_p]
        L.rph_ref_mih_values.restype = u32p
        L.rph_ref_mih_values.argtypes = [C.c_void_p]
        L.rph_ref_find_groups.restype = C.c_uint32
        L.rph_ref_find_groups.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(u32p), C.POINTER(u32p)]
        L.rph_ref_query.restype = C.c_uint32
        L.rph_ref_query.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.rph_ref_free.argtypes = [C.c_void_p]
        L.rph_ref_group_pdq.restype = C.c_uint64
        L.rph_ref_group_pdq.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.POINTER(u32p), C.POINTER(u32p), C.POINTER(u32p), u32p]
        L.rph_ref_all_pairs256.restype = C.c_uint64
        L.rph_ref_all_pairs256.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
        for name in ("rph_ref_rotate_hash_90", "rph_ref_rotate_hash_180", "rph_ref_rotate_hash_270",
                     "rph_ref_flip_hash_horizontal", "rph_ref_rotation_invariant_hash"):
            f = getattr(L, name)
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint64]
        L.rph_ref_phash_dihedral.argtypes = [C.c_uint64, C.c_void_p]
        L.rph_ref_splitmix64.restype = C.c_uint64
        L.rph_ref_splitmix64.argtypes = [C.c_uint64]
        L.rph_ref_mix32.restype = C.c_uint32
        L.rph_ref_mix32.argtypes = [C.c_uint32]
        L.rph_ref_synth_images.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_uint32]
        L.rph_ref_synth_hashes.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        L.rph_ref_synth_cluster_index.restype = C.c_uint64
        L.rph_ref_synth_cluster_index.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.rph_ref_bench_pdq.restype = C.c_double
        L.rph_ref_bench_pdq.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rph_ref_bench_all_pairs256.restype = C.c_double
        L.rph_ref_bench_all_pairs256.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
        L.rph_ref_bench_find_groups.restype = C.c_uint32
        L.rph_ref_bench_find_groups.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                                C.POINTER(C.c_double), C.POINTER(u32p), C.POINTER(u32p)]
        L.rph_ref_sbs_new.restype = C.c_void_p
        L.rph_ref_sbs_new.argtypes = [C.c_size_t]
        L.rph_ref_sbs_set.argtypes = [C.c_void_p, C.c_size_t]
        L.rph_ref_sbs_clear.argtypes = [C.c_void_p]
        L.rph_ref_sbs_free.argtypes = [C.c_void_p]
        L.rph_ref_target_dimensions.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
        L.rph_ref_jpeg_info.argtypes = [C.c_char_p, C.c_size_t, u32p, u32p, u32p]
        L.rph_ref_jpeg_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p]
        L.rph_ref_jpeg_coefficients.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------- PDQ ----------------
def dct_matrix():
    out = np.zeros((16, 64), np.float32)
    lib().rph_ref_dct_matrix(_p(out))
    return out


def features_from_buffer64(buf64):
    buf64 = np.ascontiguousarray(buf64, np.float32).reshape(64, 64)
    out = np.zeros(256, np.float32)
    lib().rph_ref_features_from_buffer64(_p(buf64), _p(out))
    return out


def to_hash(coeffs):
    coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(256)
    out = np.zeros(32, np.uint8)
    lib().rph_ref_to_hash(_p(coeffs), _p(out))
    return out


def dihedral_hashes(coeffs):
    coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(256)
    out = np.zeros((8, 32), np.uint8)
    lib().rph_ref_dihedral_hashes(_p(coeffs), _p(out))
    return out


def naive_dihedral(coeffs):
    coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(256)
    out = np.zeros((8, 32), np.uint8)
    lib().rph_ref_naive_dihedral(_p(coeffs), _p(out))
    return out


def quality(buf):
    buf = np.ascontiguousarray(buf, np.float32)
    r, c = buf.shape
    return float(lib().rph_ref_quality(_p(buf), C.c_int(r), C.c_int(c)))


def target_dimensions(w, h, max_dim=512):
    ow, oh = C.c_uint32(), C.c_uint32()
    lib().rph_ref_target_dimensions(w, h, max_dim, C.byref(ow), C.byref(oh))
    return ow.value, oh.value


def luma601(img):
    """img: (h, w, 3|4) uint8 -> (h, w) uint8"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, ch = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().rph_ref_luma601(_p(img), C.c_int(w), C.c_int(h), C.c_int(w * ch), C.c_int(ch), _p(out))
    return out


def pdq_from_luma(luma, want_buf64=False):
    luma = np.ascontiguousarray(luma, np.uint8)
    h, w = luma.shape
    coeffs = np.zeros(256, np.float32)
    q = C.c_float()
    b64 = np.zeros((64, 64), np.float32)
    rc = lib().rph_ref_pdq_from_luma(_p(luma), C.c_int(w), C.c_int(h), _p(coeffs), C.byref(q), _p(b64))
    if want_buf64:
        return rc, coeffs, q.value, b64
    return rc, coeffs, q.value


def pdq_features(img):
    """img: (h, w, 3|4) or (h, w) uint8 -> (rc, coeffs[256], quality)"""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        h, w = img.shape
        ch = 1
    else:
        h, w, ch = img.shape
    coeffs = np.zeros(256, np.float32)
    q = C.c_float()
    rc = lib().rph_ref_pdq_features(_p(img), C.c_int(w), C.c_int(h), C.c_int(w * ch), C.c_int(ch), _p(coeffs), C.byref(q))
    return rc, coeffs, q.value


def pdq_batch_rgb(imgs, want_coeffs=False):
    """imgs: (n, h, w, 3) uint8 -> hashes (n,32), quality (n,), [coeffs (n,256)]"""
    imgs = np.ascontiguousarray(imgs, np.uint8)
    n, h, w, _ = imgs.shape
    hashes = np.zeros((n, 32), np.uint8)
    qual = np.zeros(n, np.float32)
    coeffs = np.zeros((n, 256), np.float32)
    rc = lib().rph_ref_pdq_batch_rgb(_p(imgs), C.c_int(n), C.c_int(w), C.c_int(h), _p(hashes), _p(qual), _p(coeffs))
    assert rc == 0, rc
    return (hashes, qual, coeffs) if want_coeffs else (hashes, qual)


def resize_box_u8(luma, nw, nh):
    luma = np.ascontiguousarray(luma, np.uint8)
    h, w = luma.shape
    out = np.zeros((nh, nw), np.uint8)
    lib().rph_ref_resize_box_u8(_p(luma), C.c_uint32(w), C.c_uint32(h), _p(out), C.c_uint32(nw), C.c_uint32(nh))
    return out


def resize_axis_info(in_size, out_size):
    """(precision, window, min coefficient sum, max coefficient sum) of one axis of the box pre-downsample"""
    p, w, lo, hi = C.c_int(), C.c_int(), C.c_int32(), C.c_int32()
    lib().rph_ref_resize_axis_info(C.c_uint32(in_size), C.c_uint32(out_size), C.byref(p), C.byref(w), C.byref(lo), C.byref(hi))
    return p.value, w.value, lo.value, hi.value


def jarosz(plane, w_rows, w_cols, nreps=2):
    buf = np.array(plane, np.float32, copy=True, order="C")
    tmp = np.zeros_like(buf)
    rows, cols = buf.shape
    lib().rph_ref_jarosz(_p(buf), _p(tmp), C.c_int(rows), C.c_int(cols), C.c_int(w_rows), C.c_int(w_cols), C.c_int(nreps))
    return buf


def decimate(plane):
    plane = np.ascontiguousarray(plane, np.float32)
    r, c = plane.shape
    out = np.zeros((64, 64), np.float32)
    lib().rph_ref_decimate(_p(plane), C.c_int(r), C.c_int(c), _p(out))
    return out


# ---------------- Hamming / MIH ----------------
def hamming256(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().rph_ref_hamming256(_p(a), _p(b)))


def hamming64(a, b):
    return int(lib().rph_ref_hamming64(int(a), int(b)))


class MIHIndex:
    """MIHIndex<H> (hamminghash.rs:82-149).  kind: KIND_U64 (hashes: uint64[n]) or KIND_PDQ (uint8[n,32])."""

    def __init__(self, kind, hashes):
        self.kind = kind
        if kind == KIND_U64:
            self.hashes = np.ascontiguousarray(hashes, np.uint64)
        else:
            self.hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        self.n = len(self.hashes)
        self.h = lib().rph_ref_mih_new(kind, _p(self.hashes), self.n)

    def __del__(self):
        if getattr(self, "h", None):
            lib().rph_ref_mih_free(self.h)
            self.h = None

    def bucket(self, chunk, value):
        ids = C.POINTER(C.c_uint32)()
        n = lib().rph_ref_mih_bucket(self.h, chunk, value, C.byref(ids))
        return np.array([ids[i] for i in range(n)], np.uint32)

    def csr(self):
        nb = (8 * 256 if self.kind == KIND_U64 else 16 * 65536) + 1
        off = np.ctypeslib.as_array(lib().rph_ref_mih_offsets(self.h), shape=(nb,)).copy()
        vals = np.ctypeslib.as_array(lib().rph_ref_mih_values(self.h), shape=(int(off[-1]),)).copy() if off[-1] else np.zeros(0, np.uint32)
        return off, vals

    def query(self, i, max_dist, cap=1 << 16):
        out = np.zeros(cap, np.uint32)
        n = lib().rph_ref_query(self.h, i, max_dist, _p(out), cap)
        return out[:n].copy()

    def find_groups(self, max_dist):
        mem = C.POINTER(C.c_uint32)()
        off = C.POINTER(C.c_uint32)()
        ng = lib().rph_ref_find_groups(self.h, max_dist, C.byref(mem), C.byref(off))
        offs = [off[i] for i in range(ng + 1)]
        groups = [[mem[t] for t in range(offs[g], offs[g + 1])] for g in range(ng)]
        lib().rph_ref_free(mem)
        lib().rph_ref_free(off)
        return groups


def find_groups(kind, hashes, max_dist):
    return MIHIndex(kind, hashes).find_groups(max_dist)


def group_pdq(hashes, similarity, variants=None, has_features=None, quality=None):
    """group_files_generic + PdqStrategy (scanner.rs:1640-1823).

    Returns (edges[(i,j)...] in emission order incl. duplicates, groups[list of ascending members])."""
    hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
    n = len(hashes)
    v = None if variants is None else np.ascontiguousarray(variants, np.uint8).reshape(n, 8, 32)
    hf = None if has_features is None else np.ascontiguousarray(has_features, np.uint8)
    q = None if quality is None else np.ascontiguousarray(quality, np.int32)
    e = C.POINTER(C.c_uint32)()
    mem = C.POINTER(C.c_uint32)()
    off = C.POINTER(C.c_uint32)()
    ng = C.c_uint32()
    ne = lib().rph_ref_group_pdq(_p(hashes), None if v is None else _p(v), None if hf is None else _p(hf),
                                 None if q is None else _p(q), n, similarity, C.byref(e), C.byref(mem), C.byref(off),
                                 C.byref(ng))
    edges = np.ctypeslib.as_array(e, shape=(max(int(ne), 1), 2))[: int(ne)].copy()
    offs = [off[i] for i in range(ng.value + 1)]
    groups = [[mem[t] for t in range(offs[g], offs[g + 1])] for g in range(ng.value)]
    for ptr in (e, mem, off):
        lib().rph_ref_free(ptr)
    return edges, groups


def all_pairs256(hashes, thr, cap=1 << 22):
    hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
    out = np.zeros((cap, 3), np.uint32)
    n = lib().rph_ref_all_pairs256(_p(hashes), len(hashes), thr, _p(out), cap)
    assert n <= cap
    return out[: int(n)].copy()


# ---------------- pHash u64 bit ops ----------------
def rotate_hash_90(h):
    return int(lib().rph_ref_rotate_hash_90(h))


def rotate_hash_180(h):
    return int(lib().rph_ref_rotate_hash_180(h))


def rotate_hash_270(h):
    return int(lib().rph_ref_rotate_hash_270(h))


def flip_hash_horizontal(h):
    return int(lib().rph_ref_flip_hash_horizontal(h))


def rotation_invariant_hash(h):
    return int(lib().rph_ref_rotation_invariant_hash(h))


def phash_dihedral(h):
    out = np.zeros(8, np.uint64)
    lib().rph_ref_phash_dihedral(h, _p(out))
    return [int(x) for x in out]


# ---------------- synthetic workloads ----------------
def synth_images(first_k, n, w=512, h=512, seed=0x5EED2026):
    out = np.zeros((n, h, w, 3), np.uint8)
    lib().rph_ref_synth_images(_p(out), first_k, n, w, h, seed)
    return out


def synth_hashes(first, count, n_total, seed=0xC0FFEE, n_clusters=0):
    out = np.zeros((count, 32), np.uint8)
    lib().rph_ref_synth_hashes(_p(out), first, count, n_total, seed, n_clusters)
    return out


def synth_cluster_index(n_total, c, j):
    return int(lib().rph_ref_synth_cluster_index(n_total, c, j))


# ---------------- threaded CPU baseline (bench.py only) ----------------
def bench_pdq(imgs, nthreads):
    imgs = np.ascontiguousarray(imgs, np.uint8)
    n, h, w, _ = imgs.shape
    hashes = np.zeros((n, 32), np.uint8)
    qual = np.zeros(n, np.float32)
    secs = lib().rph_ref_bench_pdq(_p(imgs), n, w, h, nthreads, _p(hashes), _p(qual))
    return secs, hashes, qual


def bench_all_pairs256(hashes, thr, nthreads):
    hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
    hits = C.c_uint64()
    secs = lib().rph_ref_bench_all_pairs256(_p(hashes), len(hashes), thr, nthreads, C.byref(hits))
    return secs, hits.value


def bench_find_groups(kind, hashes, max_dist, nthreads, q_limit=0):
    if kind == KIND_U64:
        hashes = np.ascontiguousarray(hashes, np.uint64)
    else:
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
    n = len(hashes)
    times = (C.c_double * 3)()
    mem = C.POINTER(C.c_uint32)()
    off = C.POINTER(C.c_uint32)()
    full = q_limit in (0, n)
    ng = lib().rph_ref_bench_find_groups(kind, _p(hashes), n, max_dist, nthreads, q_limit, times,
                                         C.byref(mem) if full else None, C.byref(off) if full else None)
    groups = None
    if full:
        offs = [off[i] for i in range(ng + 1)]
        groups = [[mem[t] for t in range(offs[g], offs[g + 1])] for g in range(ng)]
        lib().rph_ref_free(mem)
        lib().rph_ref_free(off)
    return list(times), groups


# ---------------- JPEG decode (row N3; oracle/jpeg_ref.c) ----------------
JPEG_ZUNE, JPEG_LIBJPEG = 0, 1


def jpeg_info(data):
    """(w, h, channels) of a JPEG byte string, or raises ValueError(status)"""
    w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib().rph_ref_jpeg_info(data, len(data), C.byref(w), C.byref(h), C.byref(c))
    if rc:
        raise ValueError(rc)
    return w.value, h.value, c.value


def jpeg_decode(data, flavour=JPEG_ZUNE):
    """decoded pixels: (h, w) uint8 for one component, (h, w, 3) for three"""
    w, h, c = jpeg_info(data)
    out = np.zeros((h, w, c) if c == 3 else (h, w), np.uint8)
    rc = lib().rph_ref_jpeg_decode(data, len(data), int(flavour), _p(out))
    if rc:
        raise ValueError(rc)
    return out


def jpeg_coefficients(data):
    """(geometry[ncomp][8], qt[4][64], coef[total_blocks][64]) -- quantised coefficients, natural order, component-major"""
    _, _, c = jpeg_info(data)
    geo = np.zeros((3, 8), np.uint32)
    qt = np.zeros((4, 64), np.uint16)
    total = C.c_uint64()
    rc = lib().rph_ref_jpeg_coefficients(data, len(data), _p(geo), _p(qt), None, 0, C.byref(total))
    if rc:
        raise ValueError(rc)
    coef = np.zeros((total.value, 64), np.int16)
    rc = lib().rph_ref_jpeg_coefficients(data, len(data), _p(geo), _p(qt), _p(coef), total.value, C.byref(total))
    if rc:
        raise ValueError(rc)
    return geo[:c], qt, coef
