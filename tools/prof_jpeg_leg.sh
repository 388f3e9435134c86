#!/bin/bash
# tools/prof_jpeg_leg.sh -- the `jpeg` workload of tools/run_profile.sh alone (GPU box): bench.py --only jpeg under rocprofv3 --kernel-trace --stats
# -> gpurun_out/prof_r03/jpeg.json, gpurun_out/prof_r03/jpeg/**/run_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof_r03 && rm -rf $R/gpurun_out/prof_r03/jpeg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03/jpeg -o run -- python3 $R/bench.py --no-cpu-baseline --no-reference-cases --only jpeg > $R/gpurun_out/prof_r03/jpeg.json 2> $R/gpurun_out/prof_r03/jpeg.err
python3 $R/tools/show_bench.py $R/gpurun_out/prof_r03/jpeg.json
