#!/bin/bash
# tools/run_pmc_fetch.sh TAG "kernel-substring ..." SCRIPT [ARGS] -- FETCH_SIZE and WRITE_SIZE only (two passes) over a python script of this repo,
# summed per kernel -> gpurun_out/pmc_TAG/raw.txt (run on the GPU box)
TAG=$1; FILT=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
i=0
for p in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $p -d $OUT/pass$i -o run -- python3 $R/"$@" > $OUT/pass$i.log 2>&1 || echo "pass $i ($p) failed"
done
python3 $R/tools/pmc_summary.py $OUT $FILT > $OUT/raw.txt
cat $OUT/raw.txt
