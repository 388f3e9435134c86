// rust/pdqhash.rs -- drop-in for the reference's src/pdqhash.rs: the same public items, the work done by librupphash_hip.so.
//
//   pub use image;                                                                  (pdqhash.rs:11)
//   pub struct PdqFeatures { pub coefficients: [f32; 256] }                         (pdqhash.rs:48-51; the layout is public API: scanner.rs:1270
//                                                                                    builds it from 256 cached floats)
//   PdqFeatures::to_hash(&self) -> [u8; 32]                                         (pdqhash.rs:59)
//   PdqFeatures::generate_dihedral_hashes(&self) -> [[u8; 32]; 8]                   (pdqhash.rs:71)
//   generate_pdq_features(&image::DynamicImage) -> Option<(PdqFeatures, f32)>       (pdqhash.rs:166)
//   generate_pdq(&image::DynamicImage) -> Option<([u8; 32], f32)>                   (pdqhash.rs:199)
//
// src/scanner.rs and src/image_features.rs compile against this file unchanged.  Needs `mod rph_ffi;` (rust/rph_ffi.rs) in the crate.
// NOT COMPILED HERE: the build image has no Rust toolchain.  The same calls, in the same order, are exercised through the C ABI by
// tests/cpp/reference_tests.cpp and tests/test_gpu_parity.py.
use crate::rph_ffi as ffi;
pub use image;
use std::borrow::Cow;

pub const HASH_LENGTH: usize = 32;
const NUM_COEFFICIENTS: usize = 256;

#[derive(Clone, Debug)]
pub struct PdqFeatures {
    pub coefficients: [f32; NUM_COEFFICIENTS],
}

impl PdqFeatures {
    /// 256-bit hash of the coefficients: bit (r, c) = coefficient > median (pdqhash.rs:59-61).  One vector per call is a host-scalar
    /// function of the library (compare + bit operations, ~3 us): no kernel launch, no context.
    pub fn to_hash(&self) -> [u8; HASH_LENGTH] {
        let mut hash = [0u8; HASH_LENGTH];
        unsafe { ffi::rph_pdq_to_hash(self.coefficients.as_ptr(), hash.as_mut_ptr()) };
        hash
    }

    /// The hashes of the image under the eight dihedral transforms, in the reference's order: identity, rot90, rot180, rot270,
    /// mirror-x, mirror-y, transpose, anti-transpose (pdqhash.rs:71-87).
    pub fn generate_dihedral_hashes(&self) -> [[u8; HASH_LENGTH]; 8] {
        let mut out = [[0u8; HASH_LENGTH]; 8];
        unsafe { ffi::rph_pdq_dihedral_one(self.coefficients.as_ptr(), out.as_mut_ptr() as *mut u8) };
        out
    }
}

/// Pixels of the image in a layout the library takes as it is: Luma8 is borrowed (pdqhash.rs:173-176), Rgb8 / Rgba8 go down unchanged
/// (to_luma601 ignores alpha, pdqhash.rs:276-279), every other format through to_rgb8() as in the reference (pdqhash.rs:281).
fn packed_pixels(image: &image::DynamicImage) -> (Cow<'_, [u8]>, u32) {
    match image {
        image::DynamicImage::ImageLuma8(buf) => (Cow::Borrowed(buf.as_raw().as_slice()), 1),
        image::DynamicImage::ImageRgb8(buf) => (Cow::Borrowed(buf.as_raw().as_slice()), 3),
        image::DynamicImage::ImageRgba8(buf) => (Cow::Borrowed(buf.as_raw().as_slice()), 4),
        other => (Cow::Owned(other.to_rgb8().into_raw()), 3),
    }
}

/// One image through rph_pdq_hash_one: blocking, callable from every rayon worker at once (the library coalesces concurrent callers into
/// GPU batches, csrc/batcher.cpp).  Sides > 512 px are pre-downsampled on the GPU (resize_luma_fast, pdqhash.rs:181-220).
/// None: a side below 5 px (pdqhash.rs:167-169), or a library error (logged by the caller's policy; the reference has no error path here).
fn hash_one(image: &image::DynamicImage, want_coefficients: bool) -> Option<([u8; HASH_LENGTH], Option<PdqFeatures>, f32)> {
    let (pixels, channels) = packed_pixels(image);
    let (w, h) = (image.width(), image.height());
    let mut hash = [0u8; HASH_LENGTH];
    let mut quality = 0f32;
    let mut valid = 0u8;
    let mut features = PdqFeatures { coefficients: [0.0; NUM_COEFFICIENTS] };
    let coeffs_ptr = if want_coefficients { features.coefficients.as_mut_ptr() } else { std::ptr::null_mut() };
    let rc = unsafe {
        ffi::rph_pdq_hash_one(ffi::ctx(), pixels.as_ptr(), w, h, channels, (w as usize) * (channels as usize), hash.as_mut_ptr(), &mut quality, coeffs_ptr, &mut valid)
    };
    if rc != ffi::RPH_OK || valid == 0 {
        return None;
    }
    Some((hash, if want_coefficients { Some(features) } else { None }, quality))
}

/// The 256 DCT coefficients and the quality metric in [0, 1] of an image (pdqhash.rs:166-197).
pub fn generate_pdq_features(image: &image::DynamicImage) -> Option<(PdqFeatures, f32)> {
    hash_one(image, true).map(|(_, features, quality)| (features.expect("coefficients were requested"), quality))
}

/// The 256-bit hash and the quality metric of an image (pdqhash.rs:199-201).  The kernel produces the hash beside the coefficients, so
/// nothing is derived a second time.
pub fn generate_pdq(image: &image::DynamicImage) -> Option<([u8; HASH_LENGTH], f32)> {
    hash_one(image, false).map(|(hash, _, quality)| (hash, quality))
}

/// Many decoded images at once (not part of the reference's API; for callers that hold a batch): equal geometry, packed pixels.
/// Returns per image None / Some((hash, features, quality)) exactly as generate_pdq_features would.
pub fn generate_pdq_features_batch(pixels: &[u8], n: usize, w: u32, h: u32, channels: u32) -> Vec<Option<([u8; HASH_LENGTH], PdqFeatures, f32)>> {
    let row = (w as usize) * (channels as usize);
    assert!(pixels.len() >= n * row * h as usize);
    let mut hashes = vec![0u8; n * HASH_LENGTH];
    let mut quality = vec![0f32; n];
    let mut coeffs = vec![0f32; n * NUM_COEFFICIENTS];
    let mut valid = vec![0u8; n];
    let rc = unsafe {
        ffi::rph_pdq_hash_batch(ffi::ctx(), pixels.as_ptr(), n as u32, w, h, channels, row, row * h as usize, hashes.as_mut_ptr(), quality.as_mut_ptr(),
                                coeffs.as_mut_ptr(), std::ptr::null_mut(), valid.as_mut_ptr())
    };
    (0..n)
        .map(|k| {
            if rc != ffi::RPH_OK || valid[k] == 0 {
                return None;
            }
            let mut hash = [0u8; HASH_LENGTH];
            hash.copy_from_slice(&hashes[k * HASH_LENGTH..(k + 1) * HASH_LENGTH]);
            let mut features = PdqFeatures { coefficients: [0.0; NUM_COEFFICIENTS] };
            features.coefficients.copy_from_slice(&coeffs[k * NUM_COEFFICIENTS..(k + 1) * NUM_COEFFICIENTS]);
            Some((hash, features, quality[k]))
        })
        .collect()
}
