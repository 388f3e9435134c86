#!/bin/bash
# tools/prof_cmd.sh TAG SCRIPT [ARGS] -- rocprofv3 --kernel-trace --stats over any python script of this repo (run on the GPU box); per-kernel
# totals -> gpurun_out/prof_TAG/kernel_stats.csv and the top of it on stdout
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -o run -- python3 $R/"$@" > $OUT/stdout.txt 2> $OUT/stderr.txt || { tail -5 $OUT/stderr.txt; exit 1; }
cp $(find $OUT/run -name "run_kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cut -c1-180 $OUT/kernel_stats.csv | head -14
cat $OUT/stdout.txt | tail -12
