#!/usr/bin/env python3
"""tools/jpeg_stage_roofline.py DIR -- {small,photos,prog}_kernel_stats.csv + {small,photos,prog}_counts.json (tools/jpeg_stage_profile.sh) -> per-stage
rooflines of the JPEG path as JSON: algorithmic bytes per call / the stage's kernel time per call / 8 TB/s for the HBM-bound stages (IDCT,
upsampling + colour, pre-downsample, hash), entropy bytes per second for the Huffman walk (latency-bound: one serial bit stream per lane)."""
import csv
import json
import os
import sys

HBM = 8000.0e9
d = sys.argv[1]
out = {}
for kind in ("small", "photos", "prog"):
    cf = os.path.join(d, f"{kind}_counts.json")
    if not os.path.exists(cf):
        continue
    c = json.load(open(cf))
    calls = c["calls_total"]
    n = c["files_per_call"]
    ms = {}
    for r in csv.DictReader(open(os.path.join(d, f"{kind}_kernel_stats.csv"))):
        name = r["Name"]
        for key in ("jpeg_huff_kernel", "jpeg_prog_kernel", "jpeg_sync_kernel", "jpeg_seg_items_kernel", "jpeg_seg_map_kernel", "jpeg_idct_kernel", "jpeg_color_kernel", "jpeg_fused_kernel",
                    "pdq_fused512_kernel", "pdq_stream_kernel", "resize_mfma_kernel", "resize_fused_kernel", "fillBuffer", "copyBuffer"):
            if key in name:
                ms[key] = ms.get(key, 0.0) + float(r["TotalDurationNs"]) / 1e6 / calls
    blocks, px, ppx = c["blocks_per_file"] * n, c["pixels_per_file"] * n, c["padded_pixels_per_file"] * n
    w, h = c["geometry"]
    stages = {}

    def hbm_stage(label, kernel, nbytes, what):
        if kernel in ms and ms[kernel] > 0:
            gbs = nbytes / (ms[kernel] * 1e-3) / 1e9
            stages[label] = {"kernel": kernel, "ms_per_call": round(ms[kernel], 3), "algorithmic_bytes_per_call": int(nbytes), "bound": "hbm", "achieved_GBs": round(gbs, 1),
                             "peak_GBs": HBM / 1e9, "frac": round(gbs * 1e9 / HBM, 4), "bytes": what}

    hbm_stage("reconstruction", "jpeg_fused_kernel", blocks * 128 + px * 1.0,
              "IDCT + upsampling + colour + Rec.601 luma in one kernel: 128 B of coefficients read per 8x8 block, 1 B of luma written per pixel (the ring of chroma blocks a tile "
              "reads beyond its own is not counted)")
    hbm_stage("idct", "jpeg_idct_kernel", blocks * 192, "128 B of coefficients read + 64 B of samples written per 8x8 block")
    hbm_stage("upsampling_and_colour", "jpeg_color_kernel", ppx * 1.5 + px * 1.0, "Y + Cb + Cr samples read (1.5 B per pixel at 4:2:0), Rec.601 luma written (1 B per pixel: the hasher is the only reader)")
    if w > 512 or h > 512:
        nw, nh = (512, max(1, 512 * h // w)) if w >= h else (max(1, 512 * w // h), 512)
        hbm_stage("pre_downsample", "resize_mfma_kernel", px * 1.0 + n * nw * nh, "luma read once (1 B per pixel), thumbnail written")
        hbm_stage("hash", "pdq_stream_kernel", n * (nw * nh + 32), "thumbnail read once, 32-byte hash written")
    else:
        hbm_stage("hash", "pdq_fused512_kernel", n * (w * h + 32), "Luma8 pixels read once, 32-byte hash written")
    walk_ms = ms.get("jpeg_huff_kernel", 0.0)
    sync_ms = ms.get("jpeg_sync_kernel", 0.0) + ms.get("jpeg_seg_items_kernel", 0.0) + ms.get("jpeg_seg_map_kernel", 0.0)
    if walk_ms:
        stages["huffman_walk"] = {"kernel": "jpeg_huff_kernel", "ms_per_call": round(walk_ms, 3), "bound": "latency (one serial bit stream per lane, ~0.4 us per symbol)",
                                  "entropy_GB_per_s": round(c["file_bytes_per_call"] / (walk_ms * 1e-3) / 1e9, 2),
                                  "blocks_per_s": round(blocks / (walk_ms * 1e-3) / 1e9, 3), "blocks_per_s_unit": "G blocks/s",
                                  "coefficient_bytes_zeroed_and_written_per_call": int(blocks * 128)}
    if ms.get("jpeg_prog_kernel"):
        pm = ms["jpeg_prog_kernel"]
        stages["progressive_walk"] = {"kernel": "jpeg_prog_kernel (one lane per scan; one launch per depth of the scans' dependency order, three for libjpeg's script)",
                                      "ms_per_call": round(pm, 3), "bound": "latency (a lane's dependent instruction chain: ~138 instructions per symbol step, 8.7 clocks each when a wave is alone on its SIMD)",
                                      "entropy_GB_per_s": round(c["file_bytes_per_call"] / (pm * 1e-3) / 1e9, 2),
                                      "note": "kernel time summed over the chunks' launches, which overlap pairwise"}
    if sync_ms:
        stages["segment_synchronisation"] = {"kernels": "jpeg_sync_kernel (round 0, validation rounds, count pass) + jpeg_seg_items_kernel", "ms_per_call": round(sync_ms, 3),
                                             "entropy_GB_per_s": round(c["file_bytes_per_call"] / (sync_ms * 1e-3) / 1e9, 2)}
    if "fillBuffer" in ms:
        frac = blocks * 128 / (ms["fillBuffer"] * 1e-3) / HBM
        if frac < 1.0:
            stages["zeroing"] = {"kernel": "fillBuffer (hipMemsetAsync of the coefficient buffer)", "ms_per_call": round(ms["fillBuffer"], 3), "bound": "hbm",
                                 "achieved_GBs": round(blocks * 128 / (ms["fillBuffer"] * 1e-3) / 1e9, 1), "frac": round(frac, 4)}
        else:  # (the staged walk writes whole blocks: the coefficient buffer is not zeroed, what is left are the segment records)
            stages["zeroing"] = {"kernel": "fillBuffer (records of the segment synchronisation only: a walk that writes whole blocks needs no zeroed coefficients)",
                                 "ms_per_call": round(ms["fillBuffer"], 3)}
    dev_ms = sum(v for k, v in ms.items() if k not in ("copyBuffer",))
    out[kind] = {"workload": f"{n} files of {w}x{h} per call ({c['distinct_files']} distinct), device entropy decoding", "files_per_s": round(c["files_per_s"]),
                 "seconds_per_call": round(c["seconds_per_call"], 4), "sum_of_kernel_ms_per_call": round(dev_ms, 2), "pcie_bytes_per_call": c["file_bytes_per_call"],
                 "stages": stages}
out["source"] = "rocprofv3 --kernel-trace --stats over tools/jpeg_stage_run.py (tools/jpeg_stage_profile.sh); kernel time per call = TotalDurationNs / calls"
print(json.dumps(out, indent=1))
