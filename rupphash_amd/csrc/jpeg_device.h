// jpeg_device.h -- records the JPEG pipeline (jpeg_pipeline.cpp, host) hands to the JPEG kernels (jpeg_kernels.hip, device), and the
// launchers between them.  Internal to librupphash_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jpeg_host.h"

struct JPlane {            // one per component of each decoded image
    uint64_t first_block;  // in the chunk's coefficient buffer
    uint64_t out_off;      // byte offset of the sample plane in the chunk's plane buffer
    uint32_t blocks_w, blocks_h;
    uint32_t qt;           // index of the plane's 64-entry table in the chunk's table buffer
    uint32_t pitch;        // blocks_w * 8
    uint32_t real_bw, real_bh;      // progressive files walked on the device: the component's real blocks (its AC scans cover those) ...
    uint32_t ref_first, ref_count;  // ... and its AC refinement scans, file order, in the chunk's PRef array: their corrections are added here
    uint32_t fused, pad_;           // the image goes through jpeg_fused_kernel (IDCT + upsampling + colour in one): the plane kernels skip it
};
struct PRef {  // a refinement scan of a plane and its bit.  AC: its correction records (PCorr, one per real block, raster order).  DC
    uint32_t corr_first, al;  // (al has PREF_DC set): its bits, one byte per block of the padded grid (the walk never touches a coefficient twice)
};
constexpr uint32_t PREF_DC = 0x100u;
struct JImage {
    uint64_t plane_off[3];  // sample planes (Y, Cb, Cr)
    uint64_t out_off;       // packed pixels
    uint32_t w, h, ncomp;
    uint32_t hs, vs;        // chroma upsampling factors (1 or 2)
    uint32_t pitch[3];
    uint32_t cw, ch;        // chroma samples the upsampler may use: real component samples (libjpeg) or the padded plane (zune)
    uint32_t out_stride;    // bytes per output row: channels * align8(w)
    uint32_t luma_out;      // three components, but the hasher is the only reader: write Rec.601 luma (what to_luma601 makes of the RGB) instead of Rgb8
    uint32_t fused;         // three components with luma_out: jpeg_fused_kernel takes the image (its three planes are first_plane ..)
    uint32_t first_plane, tiles_x, tiles_y;  // tiles of 16 x 8 luma blocks
};

struct HComp {
    uint32_t blocks_w, real_bw, real_bh, first_block;
    uint32_t H, V;
};
struct HScan {
    uint32_t off, len, restart_interval, ns;
    uint32_t ci[3];
    uint32_t dc[3], ac[3];  // indices into the chunk's table array
};
struct HImage {
    uint64_t first_block;  // of the image in the chunk's coefficient buffer
    uint64_t stream_base;  // of the image's de-stuffed entropy bytes in the chunk's stream buffer
    uint32_t mcus_x, mcus_y, n_scans, ncomp;
    HComp comp[3];
    HScan scan[4];
    uint32_t pscan_first, pscan_count;  // progressive files: their scans in the chunk's PScan array (n_scans = 0)
    uint32_t mask_first, pad_;          // progressive files: index of the file's first block in the chunk's mask array (one 64-bit word per block)
};
constexpr uint32_t PSCAN_NONE = 0xFFFFFFFFu;
struct PScan {  // one scan of a progressive file (T.81 G.1): DC scans (ss == 0) may interleave components, AC scans have one
    uint32_t off, len;       // de-stuffed entropy bytes, from the image's stream_base
    uint32_t ns, ss, se, ah, al;
    uint32_t ci[3], dc[3];   // components (indices into HImage::comp) and their DC tables (first DC scans only)
    uint32_t ac;             // AC table (AC scans)
    uint32_t image;          // index of the file's HImage
    uint32_t chase[2];       // AC scans: up to two earlier AC scans of the component (indices into the chunk's PScan array, PSCAN_NONE: none)
                             // that this one follows block by block -- it may read the history of a block once they are past it
    uint32_t wait_first, wait_count;  // the scans that must have ended before this one begins (a range of the chunk's wait list)
    uint32_t corr_first;     // AC refinement scans: index of the scan's first record in the chunk's correction array (one per block of the
                             // component, raster order over its real blocks)
    uint32_t dcb[3];         // DC refinement scans: per component of the scan, where its bits go in the chunk's DC bit array (one byte per
                             // block, the component's padded grid in raster order)
};
// What an AC refinement scan leaves per block instead of touching the coefficients with history: which positions of the band (zigzag
// order = bit number) had history when the scan ran, and their correction bits, the bit of the lowest position lowest.  The IDCT kernel
// adds them to the coefficients as it loads them.
struct PCorr {
    unsigned long long history, bits;
};

// One lane's work: a whole image (scan == HITEM_ALL_SCANS: its scans one after the other, restart intervals handled in the walk), or ONE
// restart interval of a single-scan image: mcu_count MCUs from mcu_first, whose bits begin stream_off bytes into the scan -- every
// restart interval is an independent stream (predictions reset, byte aligned), so a file with restart markers is walked by as many
// lanes as it has intervals.
constexpr uint32_t HITEM_ALL_SCANS = 0xFFFFFFFFu;
struct HItem {
    uint32_t image, scan, mcu_first, mcu_count, stream_off;
    uint32_t bit_skip;  // bits to drop behind stream_off before the first symbol (segments of a stream without markers begin mid-byte)
    int32_t dc[3];      // predictions at the item's first MCU, in the scan's component order (0 at a restart interval)
};

// ---- a stream WITHOUT restart markers, cut into segments of SEG_BYTES that are walked side by side (jpeg_sync_kernel) ------------------
// A lane cannot know where a symbol begins in the middle of a stream, but Huffman streams re-synchronise: a lane that starts at a
// segment boundary in an assumed state (an MCU begins here) soon parses the same symbols as the true decoder.  Round 0: every lane
// decodes from its segment boundary, records the first MCU starts it saw inside its segment with the DC sums at each of them
// (SegTable, jpeg_kernels.hip) and reports the first MCU start behind its segment (`out`).  Validation rounds: the entry of segment t is
// the `out` of segment t - 1.  If that is one of the recorded MCU starts, round 0's decode went through it and everything from there on --
// MCU count, DC sums, exit -- is in the record; otherwise the lane decodes from the entry until it reaches a recorded MCU start (it is in
// step with round 0 from there: the rest comes from the record) or the end of the segment.  The prefix kernel then checks
// entry[t] == out[t - 1] along each file -- true for segment 0 by construction, hence for all of them -- and turns the segments into walk
// items (first MCU, DC predictions); a file that did not settle is walked by one lane as before.  Nothing here is trusted: the chain is
// verified, or not used.
constexpr uint32_t SEG_NONE = 0xFFFFFFFFu;
struct SegFile {           // one per segmented file of the chunk
    uint32_t image;        // index of its HImage
    uint32_t first_seg, n_segs;
    uint32_t first_item;   // its walk items: first_item .. first_item + n_segs - 1
    uint32_t total_mcus;
    uint32_t scan_bits;    // length of the scan's entropy data in bits: no position a chain hands on may lie at or behind it
};
struct SegState {          // one per segment
    uint32_t entry;        // bit position (from the scan's first bit) of the first MCU that begins in or behind the segment; SEG_NONE: unknown
    uint32_t out;          // first MCU start at or behind the end of the segment, as this lane's last decode saw it
    uint32_t count;        // MCUs that begin in [entry, out)
    int32_t dc[3];         // sum of their DC differences per component of the scan
    uint32_t from;         // the position the results below were decoded (or taken from round 0's record) from
    uint32_t out_check;    // count pass: `out` as decoded from `entry` (must equal out)
};


constexpr int HUFF_LDS_TABLES = 8;  // the walk keeps the chunk's Huffman tables in LDS when there are at most this many

// jpeg_kernels.hip: all asynchronous on `stream`; flavour = RPH_JPEG_ZUNE / RPH_JPEG_LIBJPEG
int rph_jpeg_launch_idct(int flavour, uint32_t max_blocks, uint32_t n_planes, hipStream_t stream, const int16_t *d_coef, const uint16_t *d_tables, const JPlane *d_planes,
                         uint8_t *d_samples, const PRef *d_refs = nullptr, const PCorr *d_corr = nullptr, const uint8_t *d_dcbits = nullptr);
int rph_jpeg_launch_color(int flavour, uint32_t max_groups, uint32_t n_images, hipStream_t stream, const uint8_t *d_samples, const JImage *d_images, uint8_t *d_pixels);
// Images marked `fused` (three components, only the hasher reads them): IDCT of a tile of 16 x 8 luma blocks and of the chroma blocks under
// them with a ring of blocks around into LDS, upsampling + colour + Rec.601 luma out of LDS -- the sample planes never exist in memory.
// max_tiles = the largest tiles_x * tiles_y of the images; d_planes / d_images as for the two kernels above (same sub-batch).
int rph_jpeg_launch_fused(int flavour, uint32_t max_tiles, uint32_t n_images, hipStream_t stream, const int16_t *d_coef, const uint16_t *d_tables, const JPlane *d_planes,
                          const JImage *d_images, uint8_t *d_pixels, const PRef *d_refs = nullptr, const PCorr *d_corr = nullptr, const uint8_t *d_dcbits = nullptr);
// d_order: the first n_ordered items in the order the lanes take them (longest first); items n_ordered .. n_items - 1 are taken as they lie
// A launch of that many items builds every block in LDS and writes it out whole (zeros included): blocks the walk covers need no zeroing before.
bool rph_jpeg_walk_writes_whole_blocks(uint32_t n_items);
int rph_jpeg_launch_walk(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const HItem *d_items, const uint32_t *d_order, uint32_t n_ordered,
                         uint32_t n_items, const rphj::DeviceLut *d_luts, uint32_t n_luts, int16_t *d_coef, uint8_t *d_status);
// Progressive files, one SCAN per lane, all scans of the chunk in one launch.  A scan depends on the earlier scans of its file that touch
// the same coefficients of the same component.  d_items: indices into d_pscans (PSCAN_NONE: idle lane), 64 per wave, a scan's producers in
// earlier waves than the scan itself: workgroups start in the order of the grid, so a producer is running (or done) whenever its
// consumer waits for it.  Every scan publishes how far it has come (d_progress: one zeroed word per scan: blocks done for AC scans, all ones
// at the end), and a consumer waits on those words -- at its start for the scans of its wait
// list, before every group of blocks for the scans it follows.
// d_masks: one zeroed 64-bit word per block of the progressive files (HImage::mask_first): which coefficients are nonzero.
// d_corr: the correction records of the AC refinement scans (n_corr of them, zeroed here), which rph_jpeg_launch_idct applies.
int rph_jpeg_launch_prog(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const PScan *d_pscans, uint32_t n_pscans, const uint32_t *d_items,
                         uint32_t n_items, const uint32_t *d_waits, const rphj::DeviceLut *d_luts, uint32_t n_luts, int16_t *d_coef, unsigned long long *d_masks,
                         uint32_t *d_progress, PCorr *d_corr, size_t n_corr, uint8_t *d_dcbits, size_t n_dcbits, uint8_t *d_status);
// Segment synchronisation of the files in d_files (see above): round 0, `rounds` validation rounds, the count pass and the prefix
// kernel, which writes the files' walk items (d_items[first_item ..]) -- n_segs of them per file, or one whole-file item and empty ones
// when the file's chain did not verify.  d_work: rph_jpeg_segment_work_bytes(n_segs) (round 0's records, two `out` slots per segment, the segment -> file map).
int rph_jpeg_launch_segments(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const SegFile *d_files, uint32_t n_files, SegState *d_segs,
                             uint32_t n_segs, uint32_t seg_bytes, void *d_work, int rounds, const rphj::DeviceLut *d_luts, uint32_t n_luts, HItem *d_items);
size_t rph_jpeg_segment_work_bytes(uint32_t n_segs);  // of d_work
void rph_jpeg_debug_segment_stats(hipStream_t stream, const void *d_work, uint32_t n_segs);  // (RPH_JPEG_TRACE=1)
