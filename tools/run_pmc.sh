#!/bin/bash
# tools/run_pmc.sh [TAG] -- rocprofv3 PMC passes over one bench.py step (run on the GPU box; counters in their own runs, with
# --kernel-trace only, one counter group per run as MI355X_MICROARCH.md prescribes).  The read-stream kernel of the same run (a known
# byte count) calibrates FETCH_SIZE.  Raw per-kernel sums -> gpurun_out/pmc_TAG/raw.txt (tools/pmc_summary.py).
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
while read -r p; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $p -d $OUT/pass$i -o run -- python3 $R/bench.py --steps 2 --warmup 1 --images 30000 --hashes 1000000 --no-e2e --no-jpeg --hashes-strong 0 --no-cpu-baseline --no-reference-cases > $OUT/pass$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH_LEVEL SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
LIST
python3 $R/tools/pmc_summary.py $OUT pdq_fused512 hamming_mfma read_stream > $OUT/raw.txt
# kernel durations of the GRBM pass (for effective clocks)
python3 - "$OUT" <<'PY' >> $OUT/raw.txt
import csv, glob, sys, collections
root = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(root + "/pass7/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("pdq_fused512", "hamming_mfma", "read_stream")):
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    print(f"DURATION_NS_PASS7 {k[:90]}  ({len(v)} dispatches)\n    total_ns {sum(v)}")
PY
cat $OUT/raw.txt | head -80
