/*
 * rupphash.h -- C ABI of librupphash_hip.so: the MI355X (gfx950) engine for the
 * PDQ-hash + 256-bit Hamming-grouping hot path of Safari77/rupphash (phdupes).
 *
 * Each entry point names the reference interface it replaces (file:line in the
 * reference tree).  The library is a drop-in for that path only; the reference's
 * Rust modules keep their public signatures and call these functions through an
 * `extern "C"` block (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 (RPH_OK) or a negative rph_status; nothing aborts
 *   - plain pointers and sizes only; the caller owns every buffer
 *   - one rph_ctx per GPU (one process per GPU with rupphash_amd/dist.py, or several
 *     contexts in one process under an rph_multi, below); a context is thread-safe:
 *     concurrent calls are safe, their GPU work is ordered on the stream each one uses
 *     (the context's own stream for the host-pointer entry points), and the library's
 *     shared scratch buffers are handed from one stream to the next with events
 *   - `*_dev` twins take DEVICE pointers and a hipStream_t (as void*) and enqueue
 *     asynchronously on it (work given to different streams may overlap; the library
 *     orders its own shared scratch between them).  They do not synchronise, except when
 *     a scratch buffer has to grow or a source geometry larger than 512 px is met for
 *     the first time (its resize tables are then uploaded once).  Host-pointer
 *     versions stage through device memory and return when the result is in the
 *     caller's buffer
 *   - there is no CPU fallback: if no gfx950 device is usable, rph_init fails
 */
#ifndef RUPPHASH_H
#define RUPPHASH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPH_ABI_VERSION 1

typedef enum rph_status {
    RPH_OK = 0,
    RPH_ERR_INVALID_ARG = -1,
    RPH_ERR_NO_DEVICE = -2,     /* no HIP device / not gfx950 */
    RPH_ERR_HIP = -3,           /* a HIP runtime call failed; rph_last_error() has the text */
    RPH_ERR_OOM = -4,
    RPH_ERR_UNSUPPORTED = -5,   /* an input this library does not take (e.g. a CMYK or arithmetic-coded JPEG): the caller keeps its own path for it */
    RPH_ERR_CAPACITY = -6       /* output capacity too small; the required count is still reported */
} rph_status;

typedef struct rph_ctx rph_ctx;

/* ---- constants of the reference ---- */
#define RPH_MAX_SIMILARITY_64 15u    /* hamminghash.rs:5  MAX_SIMILARITY_64  */
#define RPH_MAX_SIMILARITY_256 63u   /* hamminghash.rs:8  MAX_SIMILARITY_256 */
#define RPH_PDQ_MIN_QUALITY 50       /* scanner.rs:1588   PDQ_MIN_QUALITY    */
#define RPH_PDQ_MIN_DIM 5u           /* pdqhash.rs:17     MIN_HASHABLE_DIM   */
#define RPH_PDQ_MAX_DIM 512u         /* pdqhash.rs:19     DOWNSAMPLE_DIMS    */

/* ---- lifecycle ---- */
int rph_abi_version(void);
/* Bind to HIP device `device` (ordinal within HIP_VISIBLE_DEVICES).  Fails with
 * RPH_ERR_NO_DEVICE when the device is missing or is not gfx950. */
int rph_init(int device, rph_ctx **ctx_out);
int rph_shutdown(rph_ctx *ctx);
/* Text of the last error raised on this thread (never NULL). */
const char *rph_last_error(void);
const char *rph_status_string(int status);
/* name[64], compute-unit count, bytes of device memory */
int rph_device_info(rph_ctx *ctx, char *name64, int *compute_units, uint64_t *total_mem);
/* Block until all work queued on the context's own stream is done. */
int rph_synchronize(rph_ctx *ctx);

/* =====================================================================
 * PDQ hashing  (reference: src/pdqhash.rs)
 * ===================================================================== */

/*
 * Batch form of generate_pdq_features / generate_pdq (pdqhash.rs:166-201) +
 * PdqFeatures::to_hash (:59-61) + generate_dihedral_hashes (:71-87).
 *
 * px: n images of w x h pixels, `channels` interleaved u8 samples per pixel
 *     (1 = Luma8, borrowed as is pdqhash.rs:173; 3 = RGB8; 4 = RGBA8, alpha
 *     ignored :279), row_stride bytes between rows, image_stride bytes between
 *     images.  All images of a call share one geometry.
 * Outputs (each nullable except hash32_out):
 *   hash32_out   n x 32 bytes           to_hash() of the features
 *   quality_out  n floats in [0,1]      second member of the reference's tuple
 *   coeffs_out   n x 256 floats         PdqFeatures.coefficients, row-major i*16+j
 *   dihedral_out n x 8 x 32 bytes       generate_dihedral_hashes(), reference slot order
 *   valid_out    n bytes                1 = Some(..), 0 = None (w or h < 5, :167-169)
 * w or h > 512: the images are first converted to luma and pre-downsampled to the aspect-preserving <= 512 px
 * thumbnail exactly where the reference does it (:181-220, calculate_target_dimensions + resize_luma_fast).  The
 * resize is the third-party fast_image_resize 6.1.0 Convolution(Box)/U8, restated from its published algorithm:
 * parity with the Rust binary is unpinned for such inputs.
 */
int rph_pdq_hash_batch(rph_ctx *ctx, const uint8_t *px, uint32_t n, uint32_t w, uint32_t h,
                       uint32_t channels, size_t row_stride, size_t image_stride,
                       uint8_t *hash32_out, float *quality_out, float *coeffs_out,
                       uint8_t *dihedral_out, uint8_t *valid_out);
int rph_pdq_hash_batch_dev(rph_ctx *ctx, const void *d_px, uint32_t n, uint32_t w, uint32_t h,
                           uint32_t channels, size_t row_stride, size_t image_stride,
                           void *d_hash32, void *d_quality, void *d_coeffs, void *d_dihedral,
                           void *d_valid, void *stream);

/*
 * generate_pdq_features for ONE image, as scanner.rs:1410 calls it from many rayon workers at once: thread-safe and
 * blocking; concurrent callers are coalesced into GPU batches (images of any mix of sizes) whose transfers are pipelined
 * over three slots.  A batch goes as soon as nobody is still copying into it and a pipeline slot is free, so its size
 * follows the load; `max_batch` (default 256) bounds it, `max_wait_us` (default 0) optionally holds a non-full batch back
 * for more callers.  Same outputs as rph_pdq_hash_batch with n = 1.
 */
int rph_pdq_hash_one(rph_ctx *ctx, const uint8_t *px, uint32_t w, uint32_t h, uint32_t channels, size_t row_stride,
                     uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *valid_out);
int rph_pdq_batcher_config(rph_ctx *ctx, uint32_t max_batch, uint32_t max_wait_us);
int rph_pdq_batcher_stats(rph_ctx *ctx, uint64_t *n_batches_out, uint64_t *n_images_out);

/* PdqFeatures::to_hash (pdqhash.rs:59-61) and generate_dihedral_hashes (:71-87)
 * for n stored coefficient vectors (n x 256 floats), e.g. features read back
 * from the cache (scanner.rs:1270-1272).  hash32_out / dihedral_out nullable. */
int rph_pdq_hashes_from_coeffs(rph_ctx *ctx, const float *coeffs, uint32_t n, uint8_t *hash32_out,
                               uint8_t *dihedral_out);
int rph_pdq_hashes_from_coeffs_dev(rph_ctx *ctx, const void *d_coeffs, uint32_t n, void *d_hash32,
                                   void *d_dihedral, void *stream);

/* PdqFeatures::to_hash (pdqhash.rs:59-61) and generate_dihedral_hashes (:71-87) for ONE coefficient vector, on the host:
 * compare + bit operations only, no GPU call, no context.  This is what the reference's per-file call sites bind
 * (scanner.rs:1412 `f.to_hash()`, :1622 and :2223 `features.generate_dihedral_hashes()`); same bits as the batch kernels. */
void rph_pdq_to_hash(const float *coeffs256, uint8_t *hash32_out);
void rph_pdq_dihedral_one(const float *coeffs256, uint8_t *out8x32);

/* Which PDQ kernels a context uses.  512x512 RGB8 / Luma8: 0 = generic multi-pass (any geometry), 1 = fused single-pass, one wave
 * per image, 64-px strips (8 waves per CU: the throughput kernel, ~0.3 ms per image however few there are), 2 = the same
 * with 128-px strips (cache-line aligned loads, 6 waves per CU), 3 = fused low-latency form (eight waves share an image:
 * ~60 us, one image per CU), 4 = automatic (default): 3 below 768 images per call, 1 from there.  Every other geometry of
 * 128..512 px (thumbnails behind the pre-downsample included): the streaming single-pass kernel (one wave per image) from 384
 * images per call, below that the multi-pass kernels (rows through LDS tiles); 6 = as 4 with the streaming kernel at every
 * batch size (tests); 5 = no single-pass kernel at all: the multi-pass kernels (0 = their plain one-thread-per-line form,
 * which also selects the two-pass pre-downsample).  All produce identical bits.  Debug/bench. */
int rph_pdq_set_kernel(rph_ctx *ctx, int which);

/* calculate_target_dimensions (pdqhash.rs:224-235): integer geometry, host. */
void rph_pdq_target_dimensions(uint32_t w, uint32_t h, uint32_t max_dim, uint32_t *new_w, uint32_t *new_h);

/* =====================================================================
 * Hamming distance, all-pairs sweep, grouping  (reference: src/hamminghash.rs,
 * src/scanner.rs:1588-1823)
 * ===================================================================== */

/* HammingHash::hamming_distance for [u8;32] (hamminghash.rs:56-58) and u64 (:34-36);
 * HammingHash::get_chunk (:29-31, :50-53).  Scalar integer helpers, host. */
uint32_t rph_hamming_distance256(const uint8_t *a32, const uint8_t *b32);
uint32_t rph_hamming_distance64(uint64_t a, uint64_t b);
uint16_t rph_get_chunk256(const uint8_t *h32, uint32_t chunk_idx);
uint16_t rph_get_chunk64(uint64_t h, uint32_t chunk_idx);

/* Which formulation the sweep's fast path uses: 2 = fp4 MFMA (default: bits as e2m1 +-1; plain all-pairs sweeps of >= 32768
 * hashes run on a popcount-sorted copy with bits as {0, 1}, which the power-limited chip clocks ~10 % higher), 3 = fp4 MFMA with
 * +-1 operands everywhere, 4 = the sorted {0, 1} form at every size (tests), 1 = int8 MFMA, 0 = VALU xor + popcount.  All feed the same exact completion and report the same edge
 * set.  Debug/bench. */
int rph_hamming_set_kernel(rph_ctx *ctx, int which);
/* Width (in 32-bit words, 4..8) of the hash prefix the sweep's fast path examines for `threshold` under formulation `kernel`
 * (as in rph_hamming_set_kernel).  Informational (bench.py prices the fast path with it): results never depend on it. */
int rph_hamming_prefix_dwords(uint32_t threshold, int kernel);

/* One reported pair of the all-pairs sweep. */
typedef struct rph_edge {
    uint32_t i, j;  /* i < j (indices into the hash array; for variant sweeps i is the owning file) */
    uint16_t d;     /* Hamming distance, <= threshold */
    uint16_t flags; /* RPH_EDGE_* */
} rph_edge;
#define RPH_EDGE_MIH_R1 0x8000u      /* pair is reachable by find_groups' R<=1 probing (hamminghash.rs:206-238) */
#define RPH_EDGE_PROBE_MASK 0x01FFu  /* (first chunk k << 5) | probe slot: 0 = exact bucket, 1+b = flip of bit b */
#define RPH_EDGE_VARIANT_SHIFT 9     /* variant sweeps: bits 9..11 = dihedral slot that matched */
#define RPH_EDGE_VARIANT_MASK 0x0E00u

/*
 * All-pairs 256-bit sweep: every pair i<j of `hashes32` (n x 32 bytes) with
 * hamming_distance <= threshold (0..256), the exact counterpart of the
 * candidate generation of group_files_generic (scanner.rs:1704-1767, exact for
 * threshold <= 63) and, with RPH_EDGE_MIH_R1, of find_groups (hamminghash.rs:206-238).
 * Edges are written unordered; *n_edges_out always receives the total found; at
 * most `cap` are stored (RPH_ERR_CAPACITY if more were found).
 * part/nparts shard the upper-triangular tile pairs round-robin (multi-GPU:
 * rank r of N passes part=r, nparts=N; single GPU 0,1).
 */
int rph_hamming_all_pairs(rph_ctx *ctx, const uint8_t *hashes32, uint64_t n, uint32_t threshold,
                          uint32_t part, uint32_t nparts, rph_edge *edges, uint64_t cap,
                          uint64_t *n_edges_out);
/* d_count: device uint64 cursor, zeroed by the caller before the first shard. */
int rph_hamming_all_pairs_dev(rph_ctx *ctx, const void *d_hashes32, uint64_t n, uint32_t threshold,
                              uint32_t part, uint32_t nparts, void *d_edges, uint64_t cap,
                              void *d_count, void *stream);

/*
 * Variant sweep of group_files_generic + PdqStrategy (scanner.rs:1607-1637,
 * 1678-1776): rows are the 8 dihedral hashes of every file (n x 8 x 32 bytes,
 * or n x 1 x 32 when n_variants == 1), columns the plain hashes; an edge
 * (i, j, d, variant) is reported for every variant v of file i and every j > i
 * with hamming_distance(variant_v(i), hash(j)) <= limit(i, j), where
 * limit = 0 if low_conf[i] or low_conf[j] (scanner.rs:1699,1721) else `similarity`.
 * low_conf nullable (all 0).  The multiset of (i, j) equals the reference's edge
 * list (its comparison_count, scanner.rs:1778).
 */
int rph_hamming_variant_pairs(rph_ctx *ctx, const uint8_t *variants, uint32_t n_variants,
                              const uint8_t *hashes32, const uint8_t *low_conf, uint64_t n,
                              uint32_t similarity, uint32_t part, uint32_t nparts, rph_edge *edges,
                              uint64_t cap, uint64_t *n_edges_out);
int rph_hamming_variant_pairs_dev(rph_ctx *ctx, const void *d_variants, uint32_t n_variants,
                                  const void *d_hashes32, const void *d_low_conf, uint64_t n,
                                  uint32_t similarity, uint32_t part, uint32_t nparts, void *d_edges,
                                  uint64_t cap, void *d_count, void *stream);

/*
 * find_groups::<[u8;32]> (hamminghash.rs:191-271), bit-exact including member
 * order: adjacency = pairs with d <= max_dist that R<=1 probing reaches, in
 * first-seen order, then the serial greedy star clustering.
 * members: capacity n; offsets: capacity n/2 + 2; group g = members[offsets[g] .. offsets[g+1]).
 */
int rph_find_groups256(rph_ctx *ctx, const uint8_t *hashes32, uint64_t n, uint32_t max_dist,
                       uint32_t *members, uint32_t *offsets, uint32_t *n_groups_out);
/* impl HammingHash for u64 (hamminghash.rs:23-41): all-pairs sweep over 64-bit hashes (pHash) and find_groups::<u64>
 * (8 chunks of 8 bits, chunk tolerance max_dist / 8), same edge / group conventions as the 256-bit forms. */
int rph_hamming_all_pairs64(rph_ctx *ctx, const uint64_t *hashes64, uint64_t n, uint32_t threshold, uint32_t part,
                            uint32_t nparts, rph_edge *edges, uint64_t cap, uint64_t *n_edges_out);
int rph_hamming_all_pairs64_dev(rph_ctx *ctx, const void *d_hashes64, uint64_t n, uint32_t threshold, uint32_t part,
                                uint32_t nparts, void *d_edges, uint64_t cap, void *d_count, void *stream);
int rph_find_groups64(rph_ctx *ctx, const uint64_t *hashes64, uint64_t n, uint32_t max_dist, uint32_t *members,
                      uint32_t *offsets, uint32_t *n_groups_out);
/* Same from a precomputed edge list (e.g. gathered from several GPUs). */
int rph_find_groups_from_edges(const rph_edge *edges, uint64_t n_edges, uint64_t n,
                               uint32_t *members, uint32_t *offsets, uint32_t *n_groups_out);

/*
 * group_files_generic with PdqStrategy (scanner.rs:1640-1823) up to and
 * including the union-find (merge_groups_by_stem / process_raw_groups are
 * file-name logic and stay in the caller).
 *   hashes32 n x 32; coeffs n x 256 floats or NULL (then every file has the one
 *   variant out[0] = hash, scanner.rs:1624-1627); has_features n bytes or NULL
 *   (all 1 when coeffs != NULL); quality n x int32, <0 = None (scanner.rs:1592), or NULL.
 * similarity must be <= RPH_MAX_SIMILARITY_256 (scanner.rs:1650-1655) else RPH_ERR_INVALID_ARG.
 * Outputs: connected components with > 1 member: members ascending inside a
 * group (scanner.rs:1810-1814), groups ordered by first member (the reference's
 * HashMap order is unspecified); *comparison_count_out = number of edges
 * (scanner.rs:1778).  members capacity n, offsets capacity n/2 + 2.
 */
int rph_group_files_pdq(rph_ctx *ctx, const uint8_t *hashes32, const float *coeffs,
                        const uint8_t *has_features, const int32_t *quality, uint64_t n,
                        uint32_t similarity, uint32_t *members, uint32_t *offsets,
                        uint32_t *n_groups_out, uint64_t *comparison_count_out);
/* Union-find part alone (scanner.rs:1781-1817) over an edge list. */
int rph_union_find_groups(const rph_edge *edges, uint64_t n_edges, uint64_t n, uint32_t *members,
                          uint32_t *offsets, uint32_t *n_groups_out);

/* is_low_pdq_quality (scanner.rs:1592-1594); quality < 0 encodes None. */
int rph_is_low_pdq_quality(int32_t quality);

/* MIHIndex::new (hamminghash.rs:89-130) for [u8;32]: CSR arrays built on the
 * device.  offsets: 16*65536+1 u32, values: 16*n u32 (ascending id per bucket). */
int rph_mih_build256(rph_ctx *ctx, const uint8_t *hashes32, uint64_t n, uint32_t *offsets, uint32_t *values);
/* MIHIndex::<u64>::new (hamminghash.rs:23-41, :89-130): 8 chunks of 8 bits.  offsets: 8*256+1 u32, values: 8*n u32. */
int rph_mih_build64(rph_ctx *ctx, const uint64_t *hashes64, uint64_t n, uint32_t *offsets, uint32_t *values);

/* =====================================================================
 * Several GPUs under ONE host process  
 *
 * phdupes is a single process (scanner.rs:1146: one call hashes every file, then groups them), so the multi-GPU form of the
 * path is offered inside the library: an rph_multi owns one rph_ctx per device and an RCCL communicator over them
 * (ncclCommInitAll; RCCL is loaded at run time).  Its entry points shard exactly as rupphash_amd/dist.py does across processes:
 * files are dealt to the devices in contiguous ranges (no communication), ONE ncclAllGather completes the per-file hash
 * blocks on every device (the only exchange step of the path), every device sweeps the blocks p == i (mod n_devices) of the
 * sweep's enumeration, the few edges go to the host for the serial union-find.  Results equal the single-context entry
 * points for every n_devices.  One multi-device call at a time per rph_multi.
 * ===================================================================== */
typedef struct rph_multi rph_multi;
/* devices: n_devices HIP ordinals, or NULL for 0 .. n_devices-1.  RPH_ERR_UNSUPPORTED when RCCL cannot be loaded. */
int rph_multi_init(const int *devices, int n_devices, rph_multi **multi_out);
int rph_multi_shutdown(rph_multi *multi);
int rph_multi_size(rph_multi *multi);
rph_ctx *rph_multi_ctx(rph_multi *multi, int index);  /* the context of device `index`: any single-device entry point works on it */
/* rph_hamming_all_pairs across the devices (BASELINE config 5): hashes from the host, edges (unordered) back. */
int rph_multi_hamming_all_pairs(rph_multi *multi, const uint8_t *hashes32, uint64_t n, uint32_t threshold, rph_edge *edges,
                                uint64_t cap, uint64_t *n_edges_out);
/*
 * The reference's scan-then-group call (scanner.rs:1146-1551 + group_files_generic with PdqStrategy, :1640-1823) across the
 * devices (BASELINE config 4): n images of one geometry in host memory -> PDQ hash, quality, coefficients (as
 * rph_pdq_hash_batch; coeffs_out / quality_out / valid_out nullable) -> all-gather of the 8 dihedral hashes and the
 * low-confidence flag of every file (quality = round(q * 100) < 50, scanner.rs:1416-1417, :1588-1594) -> variant sweep ->
 * connected components as rph_group_files_pdq reports them.
 */
int rph_multi_hash_and_group(rph_multi *multi, const uint8_t *px, uint32_t n, uint32_t w, uint32_t h, uint32_t channels,
                             size_t row_stride, size_t image_stride, uint32_t similarity, uint8_t *hash32_out, float *quality_out,
                             float *coeffs_out, uint8_t *valid_out, uint32_t *members, uint32_t *offsets, uint32_t *n_groups_out,
                             uint64_t *comparison_count_out);
/* The same from JPEG FILES in host memory (rows N3 + B6 in one call): device i decodes and hashes files [lo_i, hi_i) as
 * rph_jpeg_pdq_hash_batch does (n_threads host threads in all, 0 = what the process may use per device), then the files that
 * produced a hash are grouped as above; members are indices into the caller's file list.  quality_out / coeffs_out / valid_out /
 * status_out are nullable. */
int rph_multi_jpeg_hash_and_group(rph_multi *multi, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads,
                                  uint32_t similarity, uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *valid_out,
                                  int32_t *status_out, uint32_t *members, uint32_t *offsets, uint32_t *n_groups_out,
                                  uint64_t *comparison_count_out);
/* rph_group_files_pdq across the devices: hashes / coefficients / quality from the cache (same arguments and results). */
int rph_multi_group_files_pdq(rph_multi *multi, const uint8_t *hashes32, const float *coeffs, const uint8_t *has_features,
                              const int32_t *quality, uint64_t n, uint32_t similarity, uint32_t *members, uint32_t *offsets,
                              uint32_t *n_groups_out, uint64_t *comparison_count_out);

/* =====================================================================
 * Cache record codecs (reference: src/db.rs), host scalar.  The layouts of the VALUES phdupes keeps per content hash,
 * for bulk import/export between an existing cache and the engine's flat arrays.  The XChaCha20-Poly1305 envelope
 * around them (db.rs:634-673) is the host application's business and is not touched here.
 * ===================================================================== */
#define RPH_PDQ_ALGO_VERSION 2u        /* db.rs:47 */
#define RPH_HASH_RECORD_BYTES 33u      /* hash_db value: [PDQ_ALGO_VERSION || 32-byte hash], db.rs:1200-1211 */
#define RPH_COEFF_RECORD_BYTES 1027u   /* coeff_db value of a 256-coefficient vector: [2 || 0x80 0x02 || 256 x f32 LE] */
void rph_hash_record_encode(const uint8_t *hash32, uint8_t *out33);
/* get_pdqhash's match (db.rs:683-696): 1 = Some(hash); 0 = None (another algorithm version or a length != 33: a miss, not an error) */
int rph_hash_record_decode(const uint8_t *rec, size_t len, uint8_t *hash32_out);
/* n records of 33 bytes each; decode returns the number of hits, present_out[i] (nullable) = 1/0, missing hashes zeroed */
void rph_hash_records_encode(const uint8_t *hashes32, size_t n, uint8_t *out33n);
size_t rph_hash_records_decode(const uint8_t *recs33n, size_t n, uint8_t *hashes32_out, uint8_t *present_out);
/* coeff_db value: [PDQ_ALGO_VERSION || postcard(CachedCoefficients { coefficients: Vec<f32> })] = [2 || varint(len) || len x f32 LE]
 * (db.rs:217-230, :1221-1231).  encode returns the record size (always; nothing is written if cap is smaller). */
size_t rph_coeff_record_size(size_t n_coeffs);
size_t rph_coeff_record_encode(const float *coeffs, size_t n_coeffs, uint8_t *out, size_t cap);
/* get_coefficients' match (db.rs:742-755): 1 = Some (n_out coefficients written; the caller keeps them only if n_out == 256,
 * scanner.rs:1265-1267); 0 = None (empty or another algorithm version); RPH_ERR_INVALID_ARG = the reference's
 * lmdb::Error::Corrupted (malformed postcard payload); RPH_ERR_CAPACITY = more than `cap` coefficients (n_out still set). */
int rph_coeff_record_decode(const uint8_t *rec, size_t len, float *coeffs_out, size_t cap, size_t *n_out);

/* =====================================================================
 * JPEG decode feeding the hasher.  Replaces the "jpg" | "jpeg" arm of load_image_fast
 * (reference: src/scanner.rs:461-508: zune-jpeg 0.5.15 -> Luma8 for one component, Rgb8 for three) followed by
 * generate_pdq_features (scanner.rs:1410).  The entropy coding is undone on the device too when a call brings enough
 * files (rph_jpeg_set_entropy; the host then only strips the byte stuffing), else by the host threads, one image each like
 * the reference's rayon workers (scanner.rs:1202); dequantisation, IDCT, chroma upsampling, colour conversion, luma and
 * the hash run on the device, a batch of files per launch, and only the hashes come back.
 * Supported: baseline / extended sequential / progressive Huffman JPEG, 8 bit, 1 or 3 components, luma sampling
 * 1x1, 2x1, 1x2, 2x2 with 1x1 chroma, restart intervals.  Anything else (CMYK, arithmetic coding, 12 bit, lossless)
 * returns RPH_ERR_UNSUPPORTED and a corrupt stream RPH_ERR_INVALID_ARG -- the caller falls through to its next
 * decoder exactly as the reference falls from tier 1 to tier 2 (scanner.rs:510-551).
 * `flavour`: the sample arithmetic behind the (decoder independent) coefficients.
 *   RPH_JPEG_ZUNE     what zune-jpeg does, as far as it can be restated without its source (absent from the reference
 *                     tree): PARITY UNPINNED against the Rust binary.
 *   RPH_JPEG_LIBJPEG  libjpeg-turbo's default arithmetic (islow IDCT, fancy upsampling): byte-identical to
 *                     libjpeg-turbo / Pillow, which is what the tests pin it with.
 * ===================================================================== */
#define RPH_JPEG_ZUNE 0
#define RPH_JPEG_LIBJPEG 1
/* Frame header only, host code: width, height and 1 or 3 channels (what zune's info() gives load_image_fast, scanner.rs:477-481). */
int rph_jpeg_info(const uint8_t *data, size_t len, uint32_t *w, uint32_t *h, uint32_t *channels);
/* The host half alone (tests, tools): quantised coefficients in natural order, component-major, raster over each component's
 * MCU-padded block grid.  geometry: 8 words per component {blocks_w, blocks_h, H, V, table, samples_w, samples_h, first_block};
 * qt: 4 x 64 quantisation tables in natural order; coef may be NULL to ask for *total_blocks only. */
int rph_jpeg_coefficients(const uint8_t *data, size_t len, uint32_t *geometry, uint16_t *qt, int16_t *coef, size_t cap_blocks,
                          uint64_t *total_blocks);
/* load_image_fast for one JPEG: pixels_out receives w * h * channels bytes, packed rows (Luma8 or Rgb8). */
int rph_jpeg_decode(rph_ctx *ctx, const uint8_t *data, size_t len, int flavour, uint8_t *pixels_out);
/* n files -> n hashes (optional quality / 256 coefficients / 8 dihedral hashes per file, as rph_pdq_hash_batch).
 * n_threads host threads undo the entropy coding (0 = as many as the process may use: affinity mask and cgroup CPU quota).  valid_out[i] = 0 and status_out[i] != RPH_OK
 * for a file that cannot be decoded (the call itself still returns RPH_OK); valid_out[i] = 0 with status RPH_OK for an
 * image below 5 px (generate_pdq_features' None, pdqhash.rs:167-169).  Files of any mix of sizes share a call. */
int rph_jpeg_pdq_hash_batch(rph_ctx *ctx, const uint8_t *const *data, const size_t *len, uint32_t n, int flavour, uint32_t n_threads,
                            uint8_t *hash32_out, float *quality_out, float *coeffs_out, uint8_t *dihedral_out, uint8_t *valid_out,
                            int32_t *status_out);

/* The same for ONE file per call, from any number of threads at once (the reference's scan loop as it stands: load_image_fast +
 * generate_pdq_features on every rayon worker, scanner.rs:1202, :1410): blocking.  The calling thread undoes the entropy coding of
 * its file itself (into a pinned buffer it keeps for its lifetime); callers that arrive while a batch is on the device leave
 * together as the next batch.  Returns the file's status (RPH_OK with *valid_out = 0: below 5 px); quality_out, coeffs_out
 * (256 floats) and valid_out may be NULL. */
int rph_jpeg_pdq_hash_one(rph_ctx *ctx, const uint8_t *data, size_t len, int flavour, uint8_t *hash32_out, float *quality_out, float *coeffs_out,
                          uint8_t *valid_out);

/* Where rph_jpeg_pdq_hash_batch decodes the Huffman streams:
 *   RPH_JPEG_ENTROPY_HOST (0)    n_threads host threads, one file each; the coefficients cross PCIe (0.8 MB per 512x512 file);
 *   RPH_JPEG_ENTROPY_DEVICE (1)  on the device, one file per lane: the host only copies the entropy bytes (stuffing undone) and the
 *                                compressed bytes cross PCIe; the walk of one file is serial (milliseconds), so this pays from
 *                                thousands of files per call;
 *   RPH_JPEG_ENTROPY_AUTO (2)    default: sequential files on the device from 512 lanes per call (a file is one lane, or one per
 *                                restart interval when it has restart markers, or one per 8 KB of a stream without them that is
 *                                long enough for segments), host below; progressive files on the device when the host threads
 *                                would need longer for all of them than the device for the longest (~0.6 us per byte);
 *   RPH_JPEG_ENTROPY_DEVICE_SEQUENTIAL (3)  as DEVICE, but progressive files stay with the host threads (tests, A/B timing).
 * Same results either way. */
#define RPH_JPEG_ENTROPY_HOST 0
#define RPH_JPEG_ENTROPY_DEVICE 1
#define RPH_JPEG_ENTROPY_AUTO 2
#define RPH_JPEG_ENTROPY_DEVICE_SEQUENTIAL 3
int rph_jpeg_set_entropy(rph_ctx *ctx, int where);
/* Tuning of the device walk for streams WITHOUT restart markers: from min_stream_bytes of entropy-coded data (default 8192) a
 * stream is cut into segments of segment_bytes (default 1024; a multiple of 4 in 64 .. 65536; 0 = never) that find their
 * entry points on the device (Huffman streams re-synchronise; the chain of entries is verified, a file that does not verify is
 * walked by one lane) and are then walked side by side -- unless, for a chunk of the call, one lane per file is estimated to be
 * quicker (many shorter files).  min_stream_bytes = 0 forces segments for every such file (tests).  Same results whatever the setting. */
int rph_jpeg_set_segments(rph_ctx *ctx, uint32_t min_stream_bytes, uint32_t segment_bytes);
/* The JPEG path keeps its staging and device buffers in the context between calls (for a large call up to half of the free device
 * memory for the coefficients of the files in flight); this returns them.  The next call allocates again. */
int rph_jpeg_release(rph_ctx *ctx);

/* =====================================================================
 * 64-bit pHash bit operations (reference: src/phash.rs:137-255), host scalar
 * ===================================================================== */
uint64_t rph_phash_rotate_90(uint64_t hash);            /* phash.rs:150 */
uint64_t rph_phash_rotate_180(uint64_t hash);           /* phash.rs:175 */
uint64_t rph_phash_rotate_270(uint64_t hash);           /* phash.rs:191 */
uint64_t rph_phash_flip_horizontal(uint64_t hash);      /* phash.rs:220 */
uint64_t rph_phash_rotation_invariant(uint64_t hash);   /* phash.rs:137 */
void rph_phash_dihedral(uint64_t hash, uint64_t out8[8]); /* phash.rs:242 */

/* =====================================================================
 * Synthetic workloads, generated on the device
 * ===================================================================== */
/* n RGB8 images w x h, global indices first_k.., packed (row stride 3*w). */
int rph_synth_images_dev(rph_ctx *ctx, void *d_out, uint64_t first_k, uint32_t n, uint32_t w, uint32_t h,
                         uint32_t seed, void *stream);
/* hashes [first, first+count) of a synthetic set of n_total (with n_clusters
 * injected 5-member clusters and one distance-32 "2 bits per chunk" pair). */
int rph_synth_hashes_dev(rph_ctx *ctx, void *d_out, uint64_t first, uint64_t count, uint64_t n_total,
                         uint64_t seed, uint64_t n_clusters, void *stream);

/* Measurement aid for bench.py: one pure read pass over a resident device buffer (16-byte aligned), on `stream`.
 * Timed with the event hooks below it gives the read bandwidth this GPU actually delivers to a streaming kernel --
 * the practical ceiling of the PDQ kernel, which reads every image byte exactly once and writes 32 bytes. */
int rph_read_stream_dev(rph_ctx *ctx, const void *d_buf, size_t bytes, void *stream);

/* Device memory helpers for callers without a HIP binding (tests, bench). */
int rph_dev_alloc(rph_ctx *ctx, size_t bytes, void **d_ptr_out);
int rph_dev_free(rph_ctx *ctx, void *d_ptr);
int rph_dev_upload(rph_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int rph_dev_download(rph_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int rph_dev_memset(rph_ctx *ctx, void *d_dst, int value, size_t bytes, void *stream);

/* Extra streams for callers without a HIP binding: the _dev entry points are asynchronous on the stream they are given, and
 * work given to different streams may overlap (the library orders its own shared scratch behind the previous user). */
int rph_stream_create(rph_ctx *ctx, void **stream_out);
int rph_stream_synchronize(rph_ctx *ctx, void *stream);
int rph_stream_destroy(rph_ctx *ctx, void *stream);

/* Timing hooks used by bench.py: HIP events recorded on `stream` (NULL = the
 * context's stream), so the measured interval is the kernels' own stream time. */
int rph_event_create(rph_ctx *ctx, void **event_out);
int rph_event_record(rph_ctx *ctx, void *event, void *stream);
int rph_event_elapsed_ms(rph_ctx *ctx, void *start, void *stop, float *ms_out); /* synchronises on stop */
int rph_event_destroy(rph_ctx *ctx, void *event);
/* The context's own stream (hipStream_t as void*). */
void *rph_stream(rph_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* RUPPHASH_H */
