"""Mirror of the public interface of /root/reference/src/hamminghash.rs on top of the C ABI.

    MAX_SIMILARITY_64 / MAX_SIMILARITY_256                 hamminghash.rs:5,8
    HammingHash for [u8;32] and u64: get_chunk, hamming_distance, bit_width_per_chunk  :11-63
    MIHIndex<[u8;32]>::{new, bucket, hash, len}            :82-149 (CSR built on the GPU)
    SparseBitSet::{new, set, clear}                        :152-189
    find_groups(&MIHIndex, max_dist) -> Vec<Vec<u32>>      :191-271 (GPU all-pairs sweep + the
                                                           reference's serial greedy clustering)
"""
import numpy as np

from . import _lib
from .engine import default_engine

MAX_SIMILARITY_64 = 15
MAX_SIMILARITY_256 = 63


class HammingHash256:
    NUM_CHUNKS = 16
    NUM_BUCKETS = 65536
    MAX_DIST = MAX_SIMILARITY_256

    @staticmethod
    def get_chunk(h, chunk_idx):
        h = np.ascontiguousarray(h, np.uint8)
        return int(_lib.load().rph_get_chunk256(h.ctypes.data, chunk_idx))

    @staticmethod
    def hamming_distance(a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return int(_lib.load().rph_hamming_distance256(a.ctypes.data, b.ctypes.data))

    @staticmethod
    def bit_width_per_chunk():
        return 16


class HammingHash64:
    NUM_CHUNKS = 8
    NUM_BUCKETS = 256
    MAX_DIST = MAX_SIMILARITY_64

    @staticmethod
    def get_chunk(h, chunk_idx):
        return int(_lib.load().rph_get_chunk64(int(h), chunk_idx))

    @staticmethod
    def hamming_distance(a, b):
        return int(_lib.load().rph_hamming_distance64(int(a), int(b)))

    @staticmethod
    def bit_width_per_chunk():
        return 8


def hamming_distance(a, b):
    if np.ndim(a) == 0:
        return HammingHash64.hamming_distance(a, b)
    return HammingHash256.hamming_distance(a, b)


class MIHIndex:
    """MIHIndex<[u8; 32]>: CSR multi-index built on the GPU (rph_mih_build256)."""

    def __init__(self, hashes, engine=None):
        self.engine = engine or default_engine()
        self.db_hashes = np.ascontiguousarray(hashes, np.uint8).reshape(-1, 32)
        self.offsets, self.values = self.engine.mih_build256(self.db_hashes)

    def bucket(self, chunk, value):
        flat = chunk * HammingHash256.NUM_BUCKETS + int(value)
        return self.values[self.offsets[flat]:self.offsets[flat + 1]]

    def hash(self, dense_id):
        return self.db_hashes[int(dense_id)]

    def len(self):
        return len(self.db_hashes)

    __len__ = len


class SparseBitSet:
    """hamminghash.rs:152-189 (host bookkeeping type of the reference's probe loop)."""

    def __init__(self, size):
        self.data = np.zeros((size + 63) // 64, np.uint64)
        self.dirty = []

    def set(self, idx):
        w, mask = idx // 64, np.uint64(1) << np.uint64(idx % 64)
        was_set = bool(self.data[w] & mask)
        if not was_set:
            if self.data[w] == 0:
                self.dirty.append(w)
            self.data[w] |= mask
        return was_set

    def clear(self):
        for w in self.dirty:
            self.data[w] = 0
        self.dirty.clear()


class MIHIndex64:
    """MIHIndex<u64>: CSR multi-index (8 chunks of 8 bits) built on the GPU (rph_mih_build64), lazily: find_groups::<u64>
    itself only needs the hashes (the sweep replaces the probe tables)."""

    def __init__(self, hashes, engine=None):
        self.engine = engine or default_engine()
        self.db_hashes = np.ascontiguousarray(hashes, np.uint64)
        self._csr = None

    def bucket(self, chunk, value):
        if self._csr is None:
            self._csr = self.engine.mih_build64(self.db_hashes)
        offsets, values = self._csr
        flat = chunk * HammingHash64.NUM_BUCKETS + int(value)
        return values[offsets[flat]:offsets[flat + 1]]

    def hash(self, dense_id):
        return int(self.db_hashes[int(dense_id)])

    def len(self):
        return len(self.db_hashes)

    __len__ = len


def find_groups(index, max_dist):
    """find_groups::<[u8;32]> / find_groups::<u64>, bit-exact including member order."""
    if isinstance(index, MIHIndex64):
        return index.engine.find_groups64(index.db_hashes, max_dist)
    return index.engine.find_groups256(index.db_hashes, max_dist)
