#!/bin/bash
# tools/run_profile.sh -- rocprofv3 kernel-trace + stats of the default bench.py command (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_final
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
head -12 $OUT/run_kernel_stats.csv
tail -1 $OUT/bench.json | head -c 600
