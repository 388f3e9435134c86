"""Engine: one rph_ctx (one GPU) with numpy-friendly wrappers over the C ABI.

Everything here is argument marshalling; all arithmetic happens inside
librupphash_hip.so on the GPU (or, for the reference's serial host logic such
as union-find, in the library's C++).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RphEdge, check

EDGE_DTYPE = np.dtype([("i", np.uint32), ("j", np.uint32), ("d", np.uint16), ("flags", np.uint16)])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Engine:
    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        check(self.L.rph_init(device, C.byref(h)), "rph_init")
        self.ctx = h
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.L.rph_shutdown(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- info / sync ----
    def device_info(self):
        name = C.create_string_buffer(64)
        cus = C.c_int()
        mem = C.c_uint64()
        check(self.L.rph_device_info(self.ctx, name, C.byref(cus), C.byref(mem)), "rph_device_info")
        return name.value.decode(), cus.value, mem.value

    def synchronize(self):
        check(self.L.rph_synchronize(self.ctx), "rph_synchronize")

    @property
    def stream(self):
        return self.L.rph_stream(self.ctx)

    def set_pdq_kernel(self, which):
        check(self.L.rph_pdq_set_kernel(self.ctx, which), "rph_pdq_set_kernel")

    def set_hamming_kernel(self, which):
        check(self.L.rph_hamming_set_kernel(self.ctx, which), "rph_hamming_set_kernel")

    # ---- PDQ ----
    def pdq_hash_batch(self, images, want_quality=True, want_coeffs=False, want_dihedral=False):
        """images: uint8 (n,h,w,3|4) or (n,h,w) [Luma8].  Returns dict(hash, quality, coeffs, dihedral, valid)."""
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 3:
            n, h, w = images.shape
            ch = 1
        else:
            n, h, w, ch = images.shape
        out = {
            "hash": np.zeros((n, 32), np.uint8),
            "quality": np.zeros(n, np.float32) if want_quality else None,
            "coeffs": np.zeros((n, 256), np.float32) if want_coeffs else None,
            "dihedral": np.zeros((n, 8, 32), np.uint8) if want_dihedral else None,
            "valid": np.zeros(n, np.uint8),
        }
        check(self.L.rph_pdq_hash_batch(self.ctx, _ptr(images), n, w, h, ch, w * ch, w * h * ch, _ptr(out["hash"]),
                                        _ptr(out["quality"]), _ptr(out["coeffs"]), _ptr(out["dihedral"]), _ptr(out["valid"])),
              "rph_pdq_hash_batch")
        return out

    def pdq_hash_one(self, image, want_coeffs=True):
        """One image through the batching queue (thread-safe; concurrent callers share a GPU batch).
        Returns (hash, quality, coeffs or None) or None when the image is too small (pdqhash.rs:167-169)."""
        image = np.ascontiguousarray(image, np.uint8)
        if image.ndim == 2:
            h, w = image.shape
            ch = 1
        else:
            h, w, ch = image.shape
        hash32 = np.zeros(32, np.uint8)
        q = C.c_float()
        coeffs = np.zeros(256, np.float32) if want_coeffs else None
        valid = C.c_uint8()
        check(self.L.rph_pdq_hash_one(self.ctx, _ptr(image), w, h, ch, w * ch, _ptr(hash32), C.cast(C.byref(q), C.c_void_p), _ptr(coeffs),
                                      C.cast(C.byref(valid), C.c_void_p)), "rph_pdq_hash_one")
        if not valid.value:
            return None
        return hash32, q.value, coeffs

    # ---- JPEG (scanner.rs:461-508) ----
    @staticmethod
    def jpeg_info(data):
        """(w, h, channels) from the frame header (host code), or raises RphError (RPH_ERR_UNSUPPORTED / RPH_ERR_INVALID_ARG)."""
        w, h, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        check(_lib.load().rph_jpeg_info(data, len(data), C.byref(w), C.byref(h), C.byref(c)), "rph_jpeg_info")
        return w.value, h.value, c.value

    @staticmethod
    def jpeg_coefficients(data):
        """The host half alone: (geometry[ncomp][8], qt[4][64], coef[total_blocks][64]) -- quantised coefficients in natural order."""
        L = _lib.load()
        _, _, c = Engine.jpeg_info(data)
        geo = np.zeros((3, 8), np.uint32)
        qt = np.zeros((4, 64), np.uint16)
        total = C.c_uint64()
        check(L.rph_jpeg_coefficients(data, len(data), _ptr(geo), _ptr(qt), None, 0, C.byref(total)), "rph_jpeg_coefficients")
        coef = np.zeros((total.value, 64), np.int16)
        check(L.rph_jpeg_coefficients(data, len(data), _ptr(geo), _ptr(qt), _ptr(coef), total.value, C.byref(total)), "rph_jpeg_coefficients")
        return geo[:c], qt, coef

    def jpeg_set_entropy(self, where):
        """0 = Huffman streams decoded by host threads, 1 = on the device, 2 = automatic (default), 3 = device for sequential files only"""
        check(self.L.rph_jpeg_set_entropy(self.ctx, int(where)), "rph_jpeg_set_entropy")

    def jpeg_pdq_hash_one(self, data, flavour=0, want_coeffs=True):
        """One JPEG file per call, thread-safe (concurrent callers share a batch): (hash, quality, coeffs or None), None when the image is
        below 5 px; raises RphError for a file the library does not decode."""
        hash32 = np.zeros(32, np.uint8)
        q = C.c_float()
        coeffs = np.zeros(256, np.float32) if want_coeffs else None
        valid = C.c_uint8()
        check(self.L.rph_jpeg_pdq_hash_one(self.ctx, data, len(data), int(flavour), _ptr(hash32), C.cast(C.byref(q), C.c_void_p), _ptr(coeffs),
                                           C.cast(C.byref(valid), C.c_void_p)), "rph_jpeg_pdq_hash_one")
        return (hash32, q.value, coeffs) if valid.value else None

    def jpeg_set_segments(self, min_stream_bytes=8192, segment_bytes=1024):
        """device walk of streams without restart markers: cut into segments from min_stream_bytes of entropy data (segment_bytes = 0: never)"""
        check(self.L.rph_jpeg_set_segments(self.ctx, int(min_stream_bytes), int(segment_bytes)), "rph_jpeg_set_segments")

    def jpeg_release(self):
        """give the JPEG path's cached staging / device buffers back"""
        check(self.L.rph_jpeg_release(self.ctx), "rph_jpeg_release")

    def jpeg_decode(self, data, flavour=0):
        """load_image_fast for one JPEG byte string: (h, w) uint8 [Luma8] or (h, w, 3) [Rgb8], decoded on the device."""
        w, h, c = self.jpeg_info(data)
        out = np.zeros((h, w, 3) if c == 3 else (h, w), np.uint8)
        check(self.L.rph_jpeg_decode(self.ctx, data, len(data), int(flavour), _ptr(out)), "rph_jpeg_decode")
        return out

    @staticmethod
    def jpeg_file_list(files):
        """The (pointer array, length array, n) a C caller would hold: build it once when the same files are hashed repeatedly"""
        n = len(files)
        return (C.c_char_p * n)(*files), (C.c_size_t * n)(*[len(f) for f in files]), n

    def jpeg_pdq_hash_batch(self, files, flavour=0, threads=0, want_quality=True, want_coeffs=False, want_dihedral=False):
        """files: list of JPEG byte strings (any mix of sizes), or the tuple jpeg_file_list() made of one.  Returns dict(hash, quality,
        coeffs, dihedral, valid, status): valid[i] = 0 with status[i] != 0 for a file that cannot be decoded, valid[i] = 0 with status 0
        for an image below 5 px."""
        arr, lens, n = files if isinstance(files, tuple) else self.jpeg_file_list(files)
        out = {
            "hash": np.zeros((n, 32), np.uint8),
            "quality": np.zeros(n, np.float32) if want_quality else None,
            "coeffs": np.zeros((n, 256), np.float32) if want_coeffs else None,
            "dihedral": np.zeros((n, 8, 32), np.uint8) if want_dihedral else None,
            "valid": np.zeros(n, np.uint8),
            "status": np.zeros(n, np.int32),
        }
        check(self.L.rph_jpeg_pdq_hash_batch(self.ctx, arr, lens, n, int(flavour), int(threads), _ptr(out["hash"]), _ptr(out["quality"]),
                                             _ptr(out["coeffs"]), _ptr(out["dihedral"]), _ptr(out["valid"]), _ptr(out["status"])),
              "rph_jpeg_pdq_hash_batch")
        return out

    def pdq_batcher_config(self, max_batch=256, max_wait_us=0):
        check(self.L.rph_pdq_batcher_config(self.ctx, max_batch, max_wait_us), "rph_pdq_batcher_config")

    def pdq_batcher_stats(self):
        nb, ni = C.c_uint64(), C.c_uint64()
        check(self.L.rph_pdq_batcher_stats(self.ctx, C.byref(nb), C.byref(ni)), "rph_pdq_batcher_stats")
        return nb.value, ni.value

    def pdq_hash_batch_dev(self, d_px, n, w, h, channels, d_hash, d_quality=None, d_coeffs=None, d_dihedral=None,
                           d_valid=None, row_stride=None, image_stride=None, stream=None):
        row_stride = w * channels if row_stride is None else row_stride
        image_stride = row_stride * h if image_stride is None else image_stride
        check(self.L.rph_pdq_hash_batch_dev(self.ctx, d_px, n, w, h, channels, row_stride, image_stride, d_hash, d_quality,
                                            d_coeffs, d_dihedral, d_valid, stream), "rph_pdq_hash_batch_dev")

    def pdq_hashes_from_coeffs(self, coeffs, want_hash=True, want_dihedral=True):
        coeffs = np.ascontiguousarray(coeffs, np.float32).reshape(-1, 256)
        n = len(coeffs)
        hashes = np.zeros((n, 32), np.uint8) if want_hash else None
        dih = np.zeros((n, 8, 32), np.uint8) if want_dihedral else None
        check(self.L.rph_pdq_hashes_from_coeffs(self.ctx, _ptr(coeffs), n, _ptr(hashes), _ptr(dih)), "rph_pdq_hashes_from_coeffs")
        return hashes, dih

    def pdq_hashes_from_coeffs_dev(self, d_coeffs, n, d_hash=None, d_dihedral=None, stream=None):
        check(self.L.rph_pdq_hashes_from_coeffs_dev(self.ctx, d_coeffs, n, d_hash, d_dihedral, stream),
              "rph_pdq_hashes_from_coeffs_dev")

    # ---- Hamming ----
    def hamming_all_pairs(self, hashes, threshold, part=0, nparts=1, cap=None):
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        cap = max(1 << 16, 4 * n) if cap is None else cap
        while True:
            edges = np.zeros(cap, EDGE_DTYPE)
            found = C.c_uint64()
            rc = self.L.rph_hamming_all_pairs(self.ctx, _ptr(hashes), n, threshold, part, nparts, _ptr(edges), cap, C.byref(found))
            if rc == _lib.RPH_ERR_CAPACITY:
                cap = int(found.value) + 1024
                continue
            check(rc, "rph_hamming_all_pairs")
            return edges[: found.value]

    def hamming_all_pairs_dev(self, d_hashes, n, threshold, d_edges, cap, d_count, part=0, nparts=1, stream=None):
        check(self.L.rph_hamming_all_pairs_dev(self.ctx, d_hashes, n, threshold, part, nparts, d_edges, cap, d_count, stream),
              "rph_hamming_all_pairs_dev")

    def hamming_variant_pairs(self, variants, hashes, similarity, low_conf=None, part=0, nparts=1, cap=None):
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        variants = np.ascontiguousarray(variants, np.uint8).reshape(n, -1, 32)
        nv = variants.shape[1]
        lc = None if low_conf is None else np.ascontiguousarray(low_conf, np.uint8)
        cap = max(1 << 16, 8 * n) if cap is None else cap
        while True:
            edges = np.zeros(cap, EDGE_DTYPE)
            found = C.c_uint64()
            rc = self.L.rph_hamming_variant_pairs(self.ctx, _ptr(variants), nv, _ptr(hashes), _ptr(lc), n, similarity, part,
                                                  nparts, _ptr(edges), cap, C.byref(found))
            if rc == _lib.RPH_ERR_CAPACITY:
                cap = int(found.value) + 1024
                continue
            check(rc, "rph_hamming_variant_pairs")
            return edges[: found.value]

    def hamming_variant_pairs_dev(self, d_variants, n_variants, d_hashes, n, similarity, d_edges, cap, d_count, d_low_conf=None,
                                  part=0, nparts=1, stream=None):
        check(self.L.rph_hamming_variant_pairs_dev(self.ctx, d_variants, n_variants, d_hashes, d_low_conf, n, similarity, part,
                                                   nparts, d_edges, cap, d_count, stream), "rph_hamming_variant_pairs_dev")

    @staticmethod
    def _groups(members, offsets, ng):
        return [members[offsets[g]:offsets[g + 1]].tolist() for g in range(ng)]

    def find_groups256(self, hashes, max_dist):
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        check(self.L.rph_find_groups256(self.ctx, _ptr(hashes), n, max_dist, _ptr(members), _ptr(offsets), C.byref(ng)),
              "rph_find_groups256")
        return self._groups(members, offsets, ng.value)

    def hamming_all_pairs64(self, hashes, threshold, part=0, nparts=1, cap=None):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        n = len(hashes)
        cap = max(1 << 16, 4 * n) if cap is None else cap
        while True:
            edges = np.zeros(cap, EDGE_DTYPE)
            found = C.c_uint64()
            rc = self.L.rph_hamming_all_pairs64(self.ctx, _ptr(hashes), n, threshold, part, nparts, _ptr(edges), cap, C.byref(found))
            if rc == _lib.RPH_ERR_CAPACITY:
                cap = int(found.value) + 1024
                continue
            check(rc, "rph_hamming_all_pairs64")
            return edges[: found.value]

    def hamming_all_pairs64_dev(self, d_hashes, n, threshold, d_edges, cap, d_count, part=0, nparts=1, stream=None):
        check(self.L.rph_hamming_all_pairs64_dev(self.ctx, d_hashes, n, threshold, part, nparts, d_edges, cap, d_count, stream),
              "rph_hamming_all_pairs64_dev")

    def find_groups64(self, hashes, max_dist):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        n = len(hashes)
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        check(self.L.rph_find_groups64(self.ctx, _ptr(hashes), n, max_dist, _ptr(members), _ptr(offsets), C.byref(ng)),
              "rph_find_groups64")
        return self._groups(members, offsets, ng.value)

    def find_groups_from_edges(self, edges, n):
        edges = np.ascontiguousarray(edges, EDGE_DTYPE)
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        check(self.L.rph_find_groups_from_edges(_ptr(edges), len(edges), n, _ptr(members), _ptr(offsets), C.byref(ng)),
              "rph_find_groups_from_edges")
        return self._groups(members, offsets, ng.value)

    def union_find_groups(self, edges, n):
        edges = np.ascontiguousarray(edges, EDGE_DTYPE)
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        check(self.L.rph_union_find_groups(_ptr(edges), len(edges), n, _ptr(members), _ptr(offsets), C.byref(ng)),
              "rph_union_find_groups")
        return self._groups(members, offsets, ng.value)

    def group_files_pdq(self, hashes, similarity, coeffs=None, has_features=None, quality=None):
        """group_files_generic + PdqStrategy up to the union-find: returns (groups, comparison_count)."""
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        c = None if coeffs is None else np.ascontiguousarray(coeffs, np.float32).reshape(n, 256)
        hf = None if has_features is None else np.ascontiguousarray(has_features, np.uint8)
        q = None if quality is None else np.ascontiguousarray(quality, np.int32)
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        cmp_count = C.c_uint64()
        check(self.L.rph_group_files_pdq(self.ctx, _ptr(hashes), _ptr(c), _ptr(hf), _ptr(q), n, similarity, _ptr(members),
                                         _ptr(offsets), C.byref(ng), C.byref(cmp_count)), "rph_group_files_pdq")
        return self._groups(members, offsets, ng.value), cmp_count.value

    def mih_build256(self, hashes):
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        offsets = np.zeros(16 * 65536 + 1, np.uint32)
        values = np.zeros(max(16 * n, 1), np.uint32)
        check(self.L.rph_mih_build256(self.ctx, _ptr(hashes), n, _ptr(offsets), _ptr(values)), "rph_mih_build256")
        return offsets, values[: 16 * n]

    def mih_build64(self, hashes):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        n = len(hashes)
        offsets = np.zeros(8 * 256 + 1, np.uint32)
        values = np.zeros(max(8 * n, 1), np.uint32)
        check(self.L.rph_mih_build64(self.ctx, _ptr(hashes), n, _ptr(offsets), _ptr(values)), "rph_mih_build64")
        return offsets, values[: 8 * n]

    # ---- synthetic workloads ----
    def synth_images_dev(self, d_out, first_k, n, w=512, h=512, seed=0x5EED2026, stream=None):
        check(self.L.rph_synth_images_dev(self.ctx, d_out, first_k, n, w, h, seed, stream), "rph_synth_images_dev")

    def synth_hashes_dev(self, d_out, first, count, n_total, seed=0xC0FFEE, n_clusters=0, stream=None):
        check(self.L.rph_synth_hashes_dev(self.ctx, d_out, first, count, n_total, seed, n_clusters, stream), "rph_synth_hashes_dev")

    def synth_images(self, first_k, n, w=512, h=512, seed=0x5EED2026):
        d = self.dev_alloc(n * w * h * 3)
        try:
            self.synth_images_dev(d, first_k, n, w, h, seed)
            out = np.zeros((n, h, w, 3), np.uint8)
            self.dev_download(out, d)
        finally:
            self.dev_free(d)
        return out

    def synth_hashes(self, first, count, n_total, seed=0xC0FFEE, n_clusters=0):
        d = self.dev_alloc(count * 32)
        try:
            self.synth_hashes_dev(d, first, count, n_total, seed, n_clusters)
            out = np.zeros((count, 32), np.uint8)
            self.dev_download(out, d)
        finally:
            self.dev_free(d)
        return out

    # ---- device memory / events ----
    def read_stream_dev(self, d_buf, nbytes, stream=None):
        """one pure read pass over a device buffer (bench.py times it: the read bandwidth a streaming kernel can get)"""
        check(self.L.rph_read_stream_dev(self.ctx, d_buf, nbytes, stream), "rph_read_stream_dev")

    def stream_create(self):
        s = C.c_void_p()
        check(self.L.rph_stream_create(self.ctx, C.byref(s)), "rph_stream_create")
        return s

    def stream_synchronize(self, stream=None):
        check(self.L.rph_stream_synchronize(self.ctx, stream), "rph_stream_synchronize")

    def stream_destroy(self, stream):
        check(self.L.rph_stream_destroy(self.ctx, stream), "rph_stream_destroy")

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self.L.rph_dev_alloc(self.ctx, nbytes, C.byref(p)), "rph_dev_alloc")
        return p.value

    def dev_free(self, p):
        check(self.L.rph_dev_free(self.ctx, p), "rph_dev_free")

    def dev_upload(self, d_dst, arr):
        arr = np.ascontiguousarray(arr)
        check(self.L.rph_dev_upload(self.ctx, d_dst, _ptr(arr), arr.nbytes), "rph_dev_upload")

    def dev_download(self, arr, d_src, nbytes=None):
        assert arr.flags["C_CONTIGUOUS"]
        check(self.L.rph_dev_download(self.ctx, _ptr(arr), d_src, arr.nbytes if nbytes is None else nbytes), "rph_dev_download")

    def dev_memset(self, d_dst, value, nbytes, stream=None):
        check(self.L.rph_dev_memset(self.ctx, d_dst, value, nbytes, stream), "rph_dev_memset")

    def event(self):
        e = C.c_void_p()
        check(self.L.rph_event_create(self.ctx, C.byref(e)), "rph_event_create")
        return e.value

    def event_record(self, e, stream=None):
        check(self.L.rph_event_record(self.ctx, e, stream), "rph_event_record")

    def event_elapsed_ms(self, start, stop):
        ms = C.c_float()
        check(self.L.rph_event_elapsed_ms(self.ctx, start, stop, C.byref(ms)), "rph_event_elapsed_ms")
        return ms.value

    def event_destroy(self, e):
        check(self.L.rph_event_destroy(self.ctx, e), "rph_event_destroy")


_default = None


def default_engine():
    """Process-wide engine on HIP device LOCAL_RANK (or 0)."""
    global _default
    if _default is None:
        import os

        _default = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    return _default


class MultiEngine:
    """Several GPUs under this one process: an rph_multi (one rph_ctx per device + an RCCL communicator inside the library).
    The sharded forms of the path (BASELINE configs 4 and 5) for a single-process host like the reference's scanner."""

    def __init__(self, devices=None, n_devices=None):
        self.L = _lib.load()
        if devices is None:
            devices = list(range(n_devices or 1))
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(self.L.rph_multi_init(arr, len(devices), C.byref(h)), "rph_multi_init")
        self.m = h
        self.devices = list(devices)

    def close(self):
        if getattr(self, "m", None):
            self.L.rph_multi_shutdown(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self.L.rph_multi_size(self.m)

    def engine(self, index):
        """Engine view of device `index`'s context (owned by the multi: do not close it)"""
        e = Engine.__new__(Engine)
        e.L = self.L
        e.ctx = None
        ctx = self.L.rph_multi_ctx(self.m, index)
        assert ctx
        e.__dict__["ctx"] = C.c_void_p(ctx)
        e.device = self.devices[index]
        e.close = lambda: None
        return e

    def hamming_all_pairs(self, hashes, threshold, cap=None):
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        cap = max(1 << 16, 4 * n) if cap is None else cap
        while True:
            edges = np.zeros(cap, EDGE_DTYPE)
            found = C.c_uint64()
            rc = self.L.rph_multi_hamming_all_pairs(self.m, _ptr(hashes), n, threshold, _ptr(edges), cap, C.byref(found))
            if rc == _lib.RPH_ERR_CAPACITY:
                cap = int(found.value) + 1024
                continue
            check(rc, "rph_multi_hamming_all_pairs")
            return edges[: found.value]

    def hash_and_group(self, images, similarity, want_coeffs=False):
        """images: uint8 (n,h,w,3|4) or (n,h,w).  Returns dict(hash, quality, coeffs, valid, groups, comparison_count)."""
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim == 3:
            n, h, w = images.shape
            ch = 1
        else:
            n, h, w, ch = images.shape
        out = {"hash": np.zeros((n, 32), np.uint8), "quality": np.zeros(n, np.float32),
               "coeffs": np.zeros((n, 256), np.float32) if want_coeffs else None, "valid": np.zeros(n, np.uint8)}
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        cmp_count = C.c_uint64()
        check(self.L.rph_multi_hash_and_group(self.m, _ptr(images), n, w, h, ch, w * ch, w * h * ch, similarity, _ptr(out["hash"]), _ptr(out["quality"]),
                                              _ptr(out["coeffs"]), _ptr(out["valid"]), _ptr(members), _ptr(offsets), C.byref(ng), C.byref(cmp_count)),
              "rph_multi_hash_and_group")
        out["groups"] = Engine._groups(members, offsets, ng.value)
        out["comparison_count"] = cmp_count.value
        return out

    def jpeg_hash_and_group(self, files, similarity, flavour=0, threads=0):
        """files: list of JPEG byte strings.  Returns dict(hash, quality, coeffs, valid, status, groups, comparison_count); groups hold
        indices into `files` (only files that produced a hash are grouped)."""
        arr, lens, n = files if isinstance(files, tuple) else Engine.jpeg_file_list(files)
        out = {"hash": np.zeros((n, 32), np.uint8), "quality": np.zeros(n, np.float32), "coeffs": np.zeros((n, 256), np.float32),
               "valid": np.zeros(n, np.uint8), "status": np.zeros(n, np.int32)}
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        cmp_count = C.c_uint64()
        check(self.L.rph_multi_jpeg_hash_and_group(self.m, arr, lens, n, int(flavour), int(threads), similarity, _ptr(out["hash"]), _ptr(out["quality"]),
                                                   _ptr(out["coeffs"]), _ptr(out["valid"]), _ptr(out["status"]), _ptr(members), _ptr(offsets), C.byref(ng),
                                                   C.byref(cmp_count)), "rph_multi_jpeg_hash_and_group")
        out["groups"] = Engine._groups(members, offsets, ng.value)
        out["comparison_count"] = cmp_count.value
        return out

    def group_files_pdq(self, hashes, similarity, coeffs=None, has_features=None, quality=None):
        hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        n = len(hashes)
        c = None if coeffs is None else np.ascontiguousarray(coeffs, np.float32).reshape(n, 256)
        hf = None if has_features is None else np.ascontiguousarray(has_features, np.uint8)
        q = None if quality is None else np.ascontiguousarray(quality, np.int32)
        members = np.zeros(max(n, 1), np.uint32)
        offsets = np.zeros(n // 2 + 2, np.uint32)
        ng = C.c_uint32()
        cmp_count = C.c_uint64()
        check(self.L.rph_multi_group_files_pdq(self.m, _ptr(hashes), _ptr(c), _ptr(hf), _ptr(q), n, similarity, _ptr(members), _ptr(offsets),
                                               C.byref(ng), C.byref(cmp_count)), "rph_multi_group_files_pdq")
        return Engine._groups(members, offsets, ng.value), cmp_count.value
