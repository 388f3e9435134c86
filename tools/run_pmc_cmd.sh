#!/bin/bash
# tools/run_pmc_cmd.sh TAG "kernel-substring ..." SCRIPT [ARGS] -- the first three PMC passes of tools/run_pmc.sh (SQ busy / wait / LDS counters) over
# any python script of this repo, summed per kernel whose name contains one of the substrings -> gpurun_out/pmc_TAG/raw.txt (run on the GPU box)
TAG=$1; FILT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
while read -r p; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $p -d $OUT/pass$i -o run -- python3 $R/"$@" > $OUT/pass$i.log 2>&1 || exit 1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
LIST
python3 $R/tools/pmc_summary.py $OUT $FILT > $OUT/raw.txt
cat $OUT/raw.txt
