// jpeg_host.h -- host half of the JPEG path (row N3): marker parsing and Huffman entropy decoding down to quantised DCT
// coefficients.  Everything after the coefficients (dequantisation, IDCT, upsampling, colour conversion, luma, hash) runs on the
// device (jpeg_kernels.hip).  Internal to librupphash_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rphj {

struct Comp {
    uint8_t id, H, V, tq, dc_tbl, ac_tbl;
    uint32_t blocks_w, blocks_h;  // MCU-padded block grid: the layout of the coefficient buffer and of the sample plane
    uint32_t real_bw, real_bh;    // ceil(samples / 8): what a non-interleaved scan covers (T.81 A.2.2)
    uint32_t samp_w, samp_h;      // ceil(w * H / Hmax), ceil(h * V / Vmax)
    uint64_t first_block;         // index of block (0, 0) in the image's coefficient buffer (component-major, raster)
    int32_t pred;
};

struct Frame {
    uint32_t w = 0, h = 0;
    int ncomp = 0;
    bool progressive = false, have_sof = false;
    int Hmax = 1, Vmax = 1;
    uint32_t mcus_x = 0, mcus_y = 0;
    Comp comp[3];
    uint16_t qt[4][64];  // natural (row-major) order
    bool qt_present[4] = {false, false, false, false};
    uint64_t total_blocks = 0;
    int adobe_transform = -1;
    uint32_t restart_interval = 0;
};

// Status codes are the library's (include/rupphash.h): RPH_OK, RPH_ERR_INVALID_ARG (not a JPEG / corrupt), RPH_ERR_UNSUPPORTED
// (arithmetic coding, lossless, 12-bit, CMYK, sampling factors outside {1,2} -- the caller falls back the way the reference falls to its next tier).

// Frame header only (stops at the first SOS): dimensions, components, block geometry.
int parse_frame(const uint8_t *data, size_t len, Frame &f);

// All scans.  `coef` receives f.total_blocks * 64 int16 in natural order (zeroed here first); `f` must come from parse_frame on
// the same bytes (its quantisation tables are completed here: a DQT may follow the frame header).
int decode_coefficients(const uint8_t *data, size_t len, Frame &f, int16_t *coef);

}  // namespace rphj
