// rust/hamminghash.rs -- drop-in for the reference's src/hamminghash.rs: the same public items (hamminghash.rs:5-271), the index
// construction and the grouping done by librupphash_hip.so.
//
//   MAX_SIMILARITY_64, MAX_SIMILARITY_256                                           (:5, :8)
//   trait HammingHash { NUM_CHUNKS, NUM_BUCKETS, MAX_DIST, get_chunk, hamming_distance, bit_width_per_chunk }   (:11-20)
//   impl HammingHash for u64, for [u8; 32]                                          (:23-63)
//   DenseId(u32)::index(), BucketId(u32)                                            (:67-81)
//   MIHIndex<H>::{new(Vec<H>), bucket(usize, u16) -> &[DenseId], hash(DenseId) -> &H, len()}   (:82-149)
//   SparseBitSet::{new(usize), set(usize) -> bool, clear()}                         (:152-189)
//   find_groups<H>(&MIHIndex<H>, u32) -> Vec<Vec<u32>>                              (:191-271)
//
// src/scanner.rs (which probes the index itself: :1673-1776) compiles against this file unchanged: the CSR arrays built on the GPU are
// the arrays MIHIndex::new builds (same offsets, same ascending ids per bucket).  Needs `mod rph_ffi;` (rust/rph_ffi.rs).
// NOT COMPILED HERE: the build image has no Rust toolchain.  tests/cpp/reference_tests.cpp runs the reference's own unit tests of this
// module through the C++ twin of this file (include/rupphash.hpp) on the GPU.
use crate::rph_ffi as ffi;

#[allow(unused)]
pub const MAX_SIMILARITY_64: u32 = ffi::RPH_MAX_SIMILARITY_64;
pub const MAX_SIMILARITY_256: u32 = ffi::RPH_MAX_SIMILARITY_256;

/// What an index is built over; `mih_build` / `find_groups_ffi` are this shim's additions (they pick the C entry point of the width).
pub trait HammingHash: Copy + Send + Sync + 'static {
    const NUM_CHUNKS: usize;
    const NUM_BUCKETS: usize;
    #[allow(dead_code)]
    const MAX_DIST: u32;

    fn get_chunk(&self, chunk_idx: usize) -> u16;
    fn hamming_distance(&self, other: &Self) -> u32;
    fn bit_width_per_chunk() -> usize;

    #[doc(hidden)]
    fn mih_build(hashes: &[Self], offsets: &mut [u32], values: &mut [u32]) -> i32;
    #[doc(hidden)]
    fn find_groups_ffi(hashes: &[Self], max_dist: u32, members: &mut [u32], offsets: &mut [u32], n_groups: &mut u32) -> i32;
}

impl HammingHash for u64 {
    const NUM_CHUNKS: usize = 8;
    const NUM_BUCKETS: usize = 256;
    const MAX_DIST: u32 = MAX_SIMILARITY_64;

    #[inline(always)]
    fn get_chunk(&self, chunk_idx: usize) -> u16 {
        unsafe { ffi::rph_get_chunk64(*self, chunk_idx as u32) }
    }
    #[inline(always)]
    fn hamming_distance(&self, other: &Self) -> u32 {
        unsafe { ffi::rph_hamming_distance64(*self, *other) }
    }
    fn bit_width_per_chunk() -> usize {
        8
    }
    fn mih_build(hashes: &[Self], offsets: &mut [u32], values: &mut [u32]) -> i32 {
        unsafe { ffi::rph_mih_build64(ffi::ctx(), hashes.as_ptr(), hashes.len() as u64, offsets.as_mut_ptr(), values.as_mut_ptr()) }
    }
    fn find_groups_ffi(hashes: &[Self], max_dist: u32, members: &mut [u32], offsets: &mut [u32], n_groups: &mut u32) -> i32 {
        unsafe { ffi::rph_find_groups64(ffi::ctx(), hashes.as_ptr(), hashes.len() as u64, max_dist, members.as_mut_ptr(), offsets.as_mut_ptr(), n_groups) }
    }
}

impl HammingHash for [u8; 32] {
    const NUM_CHUNKS: usize = 16;
    const NUM_BUCKETS: usize = 65536;
    const MAX_DIST: u32 = MAX_SIMILARITY_256;

    #[inline(always)]
    fn get_chunk(&self, chunk_idx: usize) -> u16 {
        unsafe { ffi::rph_get_chunk256(self.as_ptr(), chunk_idx as u32) }
    }
    #[inline(always)]
    fn hamming_distance(&self, other: &Self) -> u32 {
        unsafe { ffi::rph_hamming_distance256(self.as_ptr(), other.as_ptr()) }
    }
    fn bit_width_per_chunk() -> usize {
        16
    }
    fn mih_build(hashes: &[Self], offsets: &mut [u32], values: &mut [u32]) -> i32 {
        unsafe { ffi::rph_mih_build256(ffi::ctx(), hashes.as_ptr() as *const u8, hashes.len() as u64, offsets.as_mut_ptr(), values.as_mut_ptr()) }
    }
    fn find_groups_ffi(hashes: &[Self], max_dist: u32, members: &mut [u32], offsets: &mut [u32], n_groups: &mut u32) -> i32 {
        unsafe {
            ffi::rph_find_groups256(ffi::ctx(), hashes.as_ptr() as *const u8, hashes.len() as u64, max_dist, members.as_mut_ptr(), offsets.as_mut_ptr(), n_groups)
        }
    }
}

#[repr(transparent)]
#[derive(Copy, Clone, Debug, Eq, PartialEq)]
pub struct DenseId(u32);

impl DenseId {
    #[inline(always)]
    pub fn index(self) -> usize {
        self.0 as usize
    }
}

#[repr(transparent)]
#[derive(Copy, Clone, Debug, Eq, PartialEq)]
pub struct BucketId(u32);

/// Multi-index hashing table in CSR form: bucket (chunk k, value v) = values[offsets[k * NUM_BUCKETS + v] .. offsets[k * NUM_BUCKETS + v + 1]],
/// ids ascending inside a bucket (hamminghash.rs:82-131).
pub struct MIHIndex<H: HammingHash> {
    db_hashes: Box<[H]>,
    offsets: Box<[u32]>,
    values: Box<[DenseId]>,
}

impl<H: HammingHash> MIHIndex<H> {
    /// Histogram, prefix sum and stable fill run on the GPU (rph_mih_build256 / rph_mih_build64).
    pub fn new(hashes: Vec<H>) -> Self {
        let n = hashes.len();
        let mut offsets = vec![0u32; H::NUM_CHUNKS * H::NUM_BUCKETS + 1];
        let mut values = vec![0u32; H::NUM_CHUNKS * n];
        let rc = H::mih_build(&hashes, &mut offsets, &mut values);
        if rc != ffi::RPH_OK {
            panic!("rph_mih_build failed: {}", ffi::last_error());
        }
        // DenseId is repr(transparent) over u32: the vector is reinterpreted, not copied
        let values: Vec<DenseId> = {
            let mut v = std::mem::ManuallyDrop::new(values);
            unsafe { Vec::from_raw_parts(v.as_mut_ptr() as *mut DenseId, v.len(), v.capacity()) }
        };
        Self { db_hashes: hashes.into_boxed_slice(), offsets: offsets.into_boxed_slice(), values: values.into_boxed_slice() }
    }

    #[inline(always)]
    pub fn bucket(&self, chunk: usize, value: u16) -> &[DenseId] {
        let flat = chunk * H::NUM_BUCKETS + value as usize;
        &self.values[self.offsets[flat] as usize..self.offsets[flat + 1] as usize]
    }

    #[inline(always)]
    pub fn hash(&self, id: DenseId) -> &H {
        &self.db_hashes[id.index()]
    }

    #[inline(always)]
    pub fn len(&self) -> usize {
        self.db_hashes.len()
    }
}

/// Bit set that remembers which words it touched, so that clearing costs what was set (hamminghash.rs:152-189); scanner.rs keeps one per
/// worker thread (:1682).  Pure host data structure: nothing to offload.
pub struct SparseBitSet {
    words: Vec<u64>,
    touched: Vec<usize>,
}

impl SparseBitSet {
    pub fn new(size: usize) -> Self {
        Self { words: vec![0; (size + 63) / 64], touched: Vec::with_capacity(512) }
    }

    /// Sets bit `idx`; returns whether it was set before.
    #[inline(always)]
    pub fn set(&mut self, idx: usize) -> bool {
        let (w, mask) = (idx >> 6, 1u64 << (idx & 63));
        let before = self.words[w];
        if before & mask != 0 {
            return true;
        }
        if before == 0 {
            self.touched.push(w);
        }
        self.words[w] = before | mask;
        false
    }

    #[inline(always)]
    pub fn clear(&mut self) {
        for w in self.touched.drain(..) {
            self.words[w] = 0;
        }
    }
}

/// Groups of near-identical hashes with the reference's semantics, member order included (hamminghash.rs:191-271): neighbours are the
/// pairs within max_dist that R <= 1 probing reaches, in first-seen order; then the greedy star clustering in ascending id order.
/// The all-pairs sweep runs on the GPU, the serial clustering in the library's host code.
pub fn find_groups<H: HammingHash>(index: &MIHIndex<H>, max_dist: u32) -> Vec<Vec<u32>> {
    let n = index.len();
    let mut members = vec![0u32; n.max(1)];
    let mut offsets = vec![0u32; n / 2 + 2];
    let mut n_groups = 0u32;
    let rc = H::find_groups_ffi(&index.db_hashes, max_dist, &mut members, &mut offsets, &mut n_groups);
    if rc != ffi::RPH_OK {
        panic!("rph_find_groups failed: {}", ffi::last_error());
    }
    ffi::groups_from_csr(&members, &offsets, n_groups)
}
