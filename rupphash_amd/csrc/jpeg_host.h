// jpeg_host.h -- host half of the JPEG path (row N3): marker parsing and Huffman entropy decoding down to quantised DCT
// coefficients.  Everything after the coefficients (dequantisation, IDCT, upsampling, colour conversion, luma, hash) runs on the
// device (jpeg_kernels.hip).  Internal to librupphash_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

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
    Frame() {}  // user-provided: the quantisation tables (512 bytes) are not cleared per file; qt_present says which ones are there
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

// ---- device entropy decoding (jpeg_kernels.hip: one image per lane): what the host prepares for it -------------------------------
// A Huffman table as the device reads it: 10-bit lookup (code length << 8 | symbol, 0 = longer code), and the canonical-code arrays
// for the rare longer codes.
struct DeviceLut {
    uint16_t look[1024];
    int32_t maxcode[18];  // exclusive upper bound of the codes of each length in a 16-bit window
    int32_t delta[17];    // symbol index = (window >> (16 - length)) + delta[length]
    uint8_t sym[256];
    uint8_t pad[4];
};
static_assert(sizeof(DeviceLut) % 16 == 0, "tables are packed in one array");

struct TableSpec {  // a DHT entry as it stands in the file: the key tables are de-duplicated by
    uint8_t counts[17];
    uint8_t symbols[256];
    uint16_t total = 0;
};
struct ScanPlan {
    uint32_t stream_off = 0, stream_len = 0;  // de-stuffed entropy bytes of the scan in the job's stream buffer
    uint32_t restart_interval = 0;            // as defined when the scan starts
    uint8_t ns = 0, ci[3] = {0, 0, 0};
    uint32_t dc[3] = {0, 0, 0}, ac[3] = {0, 0, 0};  // table ids (from the caller's interning function) per component of the scan
    uint8_t ss = 0, se = 63, ah = 0, al = 0;        // progressive scans: spectral selection, successive approximation (T.81 G.1)
};
constexpr int MAX_PROG_SCANS = 64;
struct StreamPlan {
    int n_scans = 0;
    ScanPlan scan[4];            // sequential files
    std::vector<ScanPlan> prog;  // progressive files: all their scans, in file order (n_scans of them)
};
// Gives a table an id; equal tables get equal ids (the device keeps one lookup table per id).  UINT32_MAX = not a valid Huffman table.
typedef uint32_t (*InternTable)(void *store, const TableSpec &t);

// Walks the markers of a file whose frame `f` came from parse_frame, copies the entropy-coded
// bytes of each scan to `out` with the byte stuffing undone (0xFF00 -> 0xFF) and the RSTn markers dropped (the decoder byte-aligns
// every restart_interval MCUs instead), 32 zero bytes after each scan; at most `cap` bytes (len + 160 always suffices for a sequential
// file, len + 160 + 32 * MAX_PROG_SCANS for a progressive one).
// RPH_ERR_UNSUPPORTED: a sequential file of more than 4 scans, a progressive one of more than MAX_PROG_SCANS or with restart intervals
// -- the caller uses decode_coefficients for that file.
// `restart_marks` (nullable): for a file of ONE scan with a restart interval, the offsets (from the scan's first byte in `out`) at which
// the restart intervals after the first begin -- every interval is an independent bit stream (predictions reset, byte aligned), so
// the device can give each a lane of its own.  Left empty when the marks do not add up to ceil(MCUs / interval) - 1.
int prepare_stream(const uint8_t *data, size_t len, Frame &f, StreamPlan &plan, uint8_t *out, size_t cap, size_t *used, InternTable intern, void *store,
                   std::vector<uint32_t> *restart_marks = nullptr);
int build_device_lut(const TableSpec &t, DeviceLut &out);

}  // namespace rphj
