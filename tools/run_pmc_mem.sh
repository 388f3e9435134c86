#!/bin/bash
# tools/run_pmc_mem.sh TAG "kernel-substring ..." SCRIPT [ARGS] -- memory-side PMC passes (HBM bytes, L2 hits, L1 traffic) over a python script of
# this repo, one counter group per run, summed per kernel -> gpurun_out/pmc_TAG/raw.txt (run on the GPU box)
TAG=$1; FILT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
while read -r p; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $p -d $OUT/pass$i -o run -- python3 $R/"$@" > $OUT/pass$i.log 2>&1 || echo "pass $i ($p) failed"
done <<'LIST'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
GRBM_GUI_ACTIVE
LIST
python3 $R/tools/pmc_summary.py $OUT $FILT > $OUT/raw.txt
cat $OUT/raw.txt
