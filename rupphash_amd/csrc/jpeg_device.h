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
};
struct JImage {
    uint64_t plane_off[3];  // sample planes (Y, Cb, Cr)
    uint64_t out_off;       // packed pixels
    uint32_t w, h, ncomp;
    uint32_t hs, vs;        // chroma upsampling factors (1 or 2)
    uint32_t pitch[3];
    uint32_t cw, ch;        // chroma samples the upsampler may use: real component samples (libjpeg) or the padded plane (zune)
    uint32_t out_stride;    // bytes per output row: channels * align8(w)
    uint32_t luma_out;      // three components, but the hasher is the only reader: write Rec.601 luma (what to_luma601 makes of the RGB) instead of Rgb8
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
};

// One lane's work: a whole image (scan == HITEM_ALL_SCANS: its scans one after the other, restart intervals handled in the walk), or ONE
// restart interval of a single-scan image: mcu_count MCUs from mcu_first, whose bits begin stream_off bytes into the scan -- every
// restart interval is an independent stream (predictions reset, byte aligned), so a file with restart markers is walked by as many
// lanes as it has intervals.
constexpr uint32_t HITEM_ALL_SCANS = 0xFFFFFFFFu;
struct HItem {
    uint32_t image, scan, mcu_first, mcu_count, stream_off;
};


constexpr int HUFF_LDS_TABLES = 8;  // the walk keeps the chunk's Huffman tables in LDS when there are at most this many

// jpeg_kernels.hip: all asynchronous on `stream`; flavour = RPH_JPEG_ZUNE / RPH_JPEG_LIBJPEG
int rph_jpeg_launch_idct(int flavour, uint32_t max_blocks, uint32_t n_planes, hipStream_t stream, const int16_t *d_coef, const uint16_t *d_tables, const JPlane *d_planes,
                         uint8_t *d_samples);
int rph_jpeg_launch_color(int flavour, uint32_t max_groups, uint32_t n_images, hipStream_t stream, const uint8_t *d_samples, const JImage *d_images, uint8_t *d_pixels);
int rph_jpeg_launch_walk(hipStream_t stream, const uint8_t *d_streams, const HImage *d_images, const HItem *d_items, const uint32_t *d_order, uint32_t n_items,
                         const rphj::DeviceLut *d_luts, uint32_t n_luts, int16_t *d_coef, uint8_t *d_status);
