#!/usr/bin/env python3
"""profiles/rNN_pmc_summary.txt and profiles/pdq_traffic.json from the raw per-kernel counter sums of tools/run_pmc.sh
   usage: make_pmc_report.py gpurun_out/pmc_r02/raw.txt r02"""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw, tag = sys.argv[1], sys.argv[2]


def parse(path):
    d = {}
    cur = None
    for line in open(path):
        if not line.startswith(" "):
            m = re.match(r"(DURATION_NS_PASS7 )?(.*?)\s+\((\d+) dispatches\)", line.strip())
            name = ("DUR " if m.group(1) else "") + m.group(2)
            cur = d.setdefault(name, {"_dispatches": int(m.group(3))})
        else:
            k, v = line.split()
            cur[k] = float(v)
    return d


d = parse(raw)
pick = lambda sub, dur=False: next(v for k, v in d.items() if sub in k and k.startswith("DUR ") == dur)
pdq, ham, rs = pick("pdq_fused512"), pick("hamming_mfma"), pick("read_stream")
N_IMG, IMG, PASSES = 30000, 786432, 7
per = lambda k, c: k[c] / (k["_dispatches"] / PASSES)  # per launch: every pass runs the same launches, each counter lives in one pass
out = [f"rocprofv3 --pmc passes ({tag}; tools/run_pmc.sh: one counter group per run, --kernel-trace only), command per pass:",
       "  python3 bench.py --steps 2 --warmup 1 --images 30000 --hashes 1000000 --no-e2e --no-jpeg --hashes-strong 0 --no-cpu-baseline --no-reference-cases",
       f"Raw per-kernel sums: {tag}_pmc_raw.txt.  Per pass: 3 launches of the PDQ kernel (30000 images = 23.59 GB algorithmic read each), 3 sweeps of",
       "1 000 448 hashes at threshold 32, 4 launches of the read-stream kernel over the same 23.59 GB.  SQ_* cycle counters are in units of 4 clocks.", ""]
# calibration of FETCH_SIZE on a known byte count
rs_bytes = per(rs, "FETCH_SIZE") * 1024
cal = N_IMG * IMG / rs_bytes
out += ["read_stream_kernel (calibration: a pure 16 B/lane read of 30000 x 786432 B)",
        f"  FETCH_SIZE {per(rs, 'FETCH_SIZE'):.0f} KB per launch = {rs_bytes / 1e9:.3f} GB against {N_IMG * IMG / 1e9:.3f} GB read -> factor {cal:.3f}",
        "  (MI355X_MICROARCH.md: gfx950 reports half the bytes of a wide coalesced stream); TCC hit rate "
        f"{rs['TCC_HIT_sum'] / rs['TCC_REQ_sum'] * 100:.1f} %, TCC_MISS_sum x 128 B = {per(rs, 'TCC_MISS_sum') * 128 / 1e9:.2f} GB", ""]
fetch = per(pdq, "FETCH_SIZE") * 1024 * cal
dur = lambda k: pick(k, True)["total_ns"] / pick(k, True)["_dispatches"]
clk = lambda c, k: per(c, "GRBM_GUI_ACTIVE") / 8 / dur(k)
out += ["pdq_fused512_kernel<Geo<64>> (bench default)",
        f"  FETCH_SIZE {per(pdq, 'FETCH_SIZE'):.0f} KB per launch x 1024 x {cal:.3f} (calibration above) = {fetch / 1e9:.2f} GB = {fetch / N_IMG:.0f} B/image = "
        f"{fetch / N_IMG / IMG:.3f} x algorithmic",
        f"    (L2 -> fabric requests, Infinity-Cache hits included: a 192-B strip row straddles 128-B lines); TCC_MISS_sum x 128 B = "
        f"{per(pdq, 'TCC_MISS_sum') * 128 / 1e9:.2f} GB agrees; TCC hit rate {pdq['TCC_HIT_sum'] / pdq['TCC_REQ_sum'] * 100:.0f} %",
        f"  WRITE_SIZE {per(pdq, 'WRITE_SIZE'):.0f} KB per launch = {per(pdq, 'WRITE_SIZE') * 1024 / N_IMG:.0f} B/image (the 32-byte hashes: no VGPR is spilled since the column",
        "    recurrences' states wait in LDS between strips, round 3; round 2 wrote 5 152 B per image of scratch)",
        f"  SQ_INSTS_VALU {per(pdq, 'SQ_INSTS_VALU') / N_IMG:.0f} per image (round 1: 64 784, round 2: 62 680), SQ_INSTS_LDS {per(pdq, 'SQ_INSTS_LDS') / N_IMG:.0f}, SQ_INSTS_SALU "
        f"{per(pdq, 'SQ_INSTS_SALU') / N_IMG:.0f}, SQ_INSTS_VMEM_RD {per(pdq, 'SQ_INSTS_VMEM_RD') / N_IMG:.0f}",
        f"  wave-cycles: executing VALU {pdq['SQ_ACTIVE_INST_VALU'] / pdq['SQ_WAVE_CYCLES'] * 100:.0f} % (two waves per SIMD: the SIMD's VALU is busy ~{2 * pdq['SQ_ACTIVE_INST_VALU'] / pdq['SQ_WAVE_CYCLES'] * 100:.0f} % of the time), "
        f"SQ_WAIT_INST_ANY {pdq['SQ_WAIT_INST_ANY'] / pdq['SQ_WAVE_CYCLES'] * 100:.0f} % (LDS {pdq['SQ_WAIT_INST_LDS'] / pdq['SQ_WAVE_CYCLES'] * 100:.1f} %), SQ_WAIT_ANY {pdq['SQ_WAIT_ANY'] / pdq['SQ_WAVE_CYCLES'] * 100:.0f} %",
        f"  LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {pdq['SQ_LDS_BANK_CONFLICT'] / pdq['SQ_LDS_IDX_ACTIVE'] * 100:.0f} %",
        f"  effective clock (GRBM_GUI_ACTIVE / 8 / kernel time, profiled run): {clk(pdq, 'pdq_fused512'):.2f} GHz; the read-stream kernel of the same run: "
        f"{clk(rs, 'read_stream'):.2f} GHz -> the PDQ kernel is held under 2.4 GHz by power management, the memory stream alone is not", ""]
n_mfma = per(ham, "SQ_INSTS_MFMA")
hfetch = per(ham, "FETCH_SIZE") * 1024
out += ["hamming_mfma_kernel<FmtFp4ZO, 4> (1 000 448 hashes, threshold 32: 5.0045e11 pairs per sweep)",
        f"  SQ_INSTS_MFMA {n_mfma:.4g} per sweep (= 2 per 1024 pairs), SQ_INSTS_VALU {per(ham, 'SQ_INSTS_VALU'):.4g} = {per(ham, 'SQ_INSTS_VALU') / n_mfma:.2f} per MFMA (round 1: 7.4, round 2: 6.41)",
        f"  SQ_VALU_MFMA_BUSY_CYCLES {per(ham, 'SQ_VALU_MFMA_BUSY_CYCLES'):.4g} (= 32 x MFMAs) / (GRBM_GUI_ACTIVE {per(ham, 'GRBM_GUI_ACTIVE'):.4g} / 8 XCDs x 1024 SIMDs) = "
        f"{per(ham, 'SQ_VALU_MFMA_BUSY_CYCLES') / (per(ham, 'GRBM_GUI_ACTIVE') / 8 * 1024) * 100:.0f} % of the cycles the matrix pipe is busy",
        f"  effective clock (GRBM_GUI_ACTIVE / 8 / kernel time, profiled run): {clk(ham, 'hamming_mfma'):.2f} GHz; in-kernel s_memtime / s_memrealtime of the same loop (tools/sweep_loop.hip): 1.78 GHz",
        "    on +-1 operands, 1.92-1.98 GHz on {0,1} operands -- the sweep runs against the power limit, not the pipe's cycle count",
        f"  FETCH_SIZE {per(ham, 'FETCH_SIZE'):.0f} KB per sweep = {hfetch / 1e9:.2f} GB uncorrected, {hfetch * cal / 1e9:.2f} GB with the wide-stream factor (TCC_MISS_sum x 128 B = "
        f"{per(ham, 'TCC_MISS_sum') * 128 / 1e9:.2f} GB agrees with the latter): {hfetch * cal / dur('hamming_mfma'):.0f} GB/s achieved from HBM/fabric",
        f"    against 31.3 GB of algorithmic tile bytes (64/T B per pair, T = 1024): row fragments are loaded once per 8 column tiles and half the requests hit L2 ({ham['TCC_HIT_sum'] / ham['TCC_REQ_sum'] * 100:.0f} %)",
        f"  wave-cycles: executing VALU {ham['SQ_ACTIVE_INST_VALU'] / ham['SQ_WAVE_CYCLES'] * 100:.0f} %, SQ_WAIT_INST_ANY {ham['SQ_WAIT_INST_ANY'] / ham['SQ_WAVE_CYCLES'] * 100:.0f} % (LDS {ham['SQ_WAIT_INST_LDS'] / ham['SQ_WAVE_CYCLES'] * 100:.1f} %), "
        f"SQ_WAIT_ANY {ham['SQ_WAIT_ANY'] / ham['SQ_WAVE_CYCLES'] * 100:.0f} %",
        f"  LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {ham['SQ_LDS_BANK_CONFLICT'] / ham['SQ_LDS_IDX_ACTIVE'] * 100:.0f} % (the 256-entry code LUT is read at random addresses while a chunk is expanded; the B-fragment reads of the MFMA loop are conflict-free)"]
open(f"{R}/profiles/{tag}_pmc_summary.txt", "w").write("\n".join(out) + "\n")
j = json.load(open(f"{R}/profiles/pdq_traffic.json"))
j["fused512_strip64"] = round(fetch / N_IMG)
j["source"] = (f"profiles/{tag}_pmc_summary.txt: rocprofv3 --pmc FETCH_SIZE of pdq_fused512_kernel over 30000 images, x1024 B, x{cal:.3f} (calibrated on the read-stream kernel of the "
               "same run; MI355X_MICROARCH.md: gfx950 reports half the bytes of a wide coalesced stream); TCC_MISS_sum x 128 B agrees")
j["counts"] = "L2 -> fabric read requests: Infinity-Cache hits are included, so this can exceed what HBM itself delivered"
json.dump(j, open(f"{R}/profiles/pdq_traffic.json", "w"), indent=1)
print("\n".join(out))
