#!/usr/bin/env python3
"""profiles/r01_pmc_summary.txt (and the strip64 entry of pdq_traffic.json) from the raw per-kernel counter sums"""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def parse(path):
    d = {}; cur = None
    for l in open(path):
        if not l.startswith(' '):
            cur = l.strip(); d[cur] = {}
        else:
            k, v = l.split(); d[cur][k] = float(v)
    return d
a = parse(f'{R}/profiles/r01_pmc_strip64_raw.txt'); b = parse(f'{R}/profiles/r01_pmc_strip128_raw.txt')
def pick(d, sub):
    for k in d:
        if sub in k: return k, d[k]
_, p64 = pick(a, 'pdq_fused512'); _, p128 = pick(b, 'pdq_fused512'); hk, hm = pick(a, 'hamming_fp4_kernel<4') if pick(a, 'hamming_fp4_kernel<4') else pick(a, 'FmtFp4, 4')
n = 30000
out = ["rocprofv3 --pmc passes (tools/run_pmc.sh: one counter group per run, --kernel-trace only), command per pass:",
       "  python3 bench.py --steps 1 --warmup 0 --images 30000 --hashes 1000000 --no-cpu-baseline [--pdq-kernel 2]",
       "Raw per-kernel sums: r01_pmc_strip64_raw.txt (default run: PDQ strip64 + fp4 Hamming), r01_pmc_strip128_raw.txt (--pdq-kernel 2, taken",
       "before the Hamming kernels were restructured; its PDQ rows are current).  The PDQ kernel runs once per pass (30000 images =",
       "23.59 GB algorithmic read); SQ_* cycle counters are in units of 4 clocks.", ""]
for name, p in (("pdq_fused512_kernel<Geo<64>> (default)", p64), ("pdq_fused512_kernel<Geo<128>>", p128)):
    fetch = p['FETCH_SIZE'] * 1024 * 2
    out += [name,
            f"  FETCH_SIZE {p['FETCH_SIZE']:.0f} KB-units -> x1024 x2 (gfx950 reports half the bytes of a wide coalesced stream, MI355X_MICROARCH.md) = {fetch/1e9:.2f} GB",
            f"    = {fetch/n:.0f} B/image = {fetch/n/786432:.3f} x algorithmic; TCC_MISS_sum x 128 B = {p['TCC_MISS_sum']*128/1e9:.2f} GB agrees; TCC hit rate {p['TCC_HIT_sum']/p['TCC_REQ_sum']*100:.0f} %",
            f"  WRITE_SIZE {p['WRITE_SIZE']:.0f} KB-units = {p['WRITE_SIZE']*1024/1e6:.1f} MB (32-byte hashes" + (" plus the 32 B/lane of scratch the first and last band of this build spill: 11 KB/image)" if p['WRITE_SIZE'] > 10000 else "; this build has no scratch)"),
            f"  SQ_INSTS_VALU {p['SQ_INSTS_VALU']/n:.0f} per image, SQ_INSTS_LDS {p['SQ_INSTS_LDS']/n:.0f}, SQ_INSTS_SALU {p['SQ_INSTS_SALU']/n:.0f}, SQ_INSTS_VMEM_RD {p['SQ_INSTS_VMEM_RD']/n:.0f}",
            f"  wave-cycles: SQ_WAVE_CYCLES {p['SQ_WAVE_CYCLES']:.3g}; executing VALU {p['SQ_ACTIVE_INST_VALU']/p['SQ_WAVE_CYCLES']*100:.0f} %, SQ_WAIT_INST_ANY {p['SQ_WAIT_INST_ANY']/p['SQ_WAVE_CYCLES']*100:.0f} % of which LDS {p['SQ_WAIT_INST_LDS']/p['SQ_WAVE_CYCLES']*100:.1f} %",
            f"  LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {p['SQ_LDS_BANK_CONFLICT']/p['SQ_LDS_IDX_ACTIVE']*100:.0f} %", ""]
nd = int(re.search(r'\((\d+) dispatches\)', hk).group(1))
out += [f"fp4 MFMA sweep, PW = 4 (1M hashes, threshold 32; {nd} dispatches over the 7 passes: warm-up, the timed launch, and the 500k-file reference case in each)",
        f"  SQ_INSTS_MFMA {hm.get('SQ_INSTS_MFMA', 0):.4g}, SQ_INSTS_VALU {hm['SQ_INSTS_VALU']:.4g}",
        f"  SQ_VALU_MFMA_BUSY_CYCLES {hm['SQ_VALU_MFMA_BUSY_CYCLES']:.4g} / (GRBM_GUI_ACTIVE {hm['GRBM_GUI_ACTIVE']:.4g} summed over 8 XCDs / 8 x 1024 SIMDs) = {hm['SQ_VALU_MFMA_BUSY_CYCLES']/(hm['GRBM_GUI_ACTIVE']/8*1024)*100:.0f} % matrix-pipe busy",
        f"  wave-cycles: executing VALU {hm['SQ_ACTIVE_INST_VALU']/hm['SQ_WAVE_CYCLES']*100:.0f} %, SQ_WAIT_INST_ANY {hm['SQ_WAIT_INST_ANY']/hm['SQ_WAVE_CYCLES']*100:.0f} % (LDS {hm['SQ_WAIT_INST_LDS']/hm['SQ_WAVE_CYCLES']*100:.1f} %)",
        f"  LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {hm['SQ_LDS_BANK_CONFLICT']/hm['SQ_LDS_IDX_ACTIVE']*100:.0f} % (the 256-entry +-1 LUT is read at random addresses)"]
open(f'{R}/profiles/r01_pmc_summary.txt', 'w').write('\n'.join(out) + '\n')
j = json.load(open(f'{R}/profiles/pdq_traffic.json'))
j['fused512_strip64'] = round(p64['FETCH_SIZE'] * 1024 * 2 / n)
json.dump(j, open(f'{R}/profiles/pdq_traffic.json', 'w'), indent=1)
print('\n'.join(out[-6:]))
