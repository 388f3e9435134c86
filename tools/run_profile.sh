#!/bin/bash
# tools/run_profile.sh [TAG] -- rocprofv3 kernel-trace + stats on the GPU box, one command per workload so that the average duration of a
# kernel in each CSV is THAT workload only (no CPU baseline, no reference cases: nothing under the profiler spawns a child):
#   gpurun_out/prof_TAG/default/   python3 bench.py --no-e2e --no-jpeg --no-cpu-baseline --no-reference-cases   (BASELINE configs 2 and 3: the two headline kernels)
#   gpurun_out/prof_TAG/e2e/       python3 bench.py --only e2e ...                                  (config 4 on one GPU; its PDQ launches also write
#                                                                                                    quality + 8 dihedral hashes and take longer)
#   gpurun_out/prof_TAG/thr40/     python3 bench.py --only hamming --threshold 40 ...              (the scanner's default similarity)
#   gpurun_out/prof_TAG/thr63/     python3 bench.py --only hamming --threshold 63 ...              (MAX_SIMILARITY_256)
#   gpurun_out/prof_TAG/config5/   python3 bench.py --only hamming_10m ...                         (BASELINE config 5: 10 000 000 hashes, on the GPUs given)
#   gpurun_out/prof_TAG/jpeg/      python3 bench.py --only jpeg ...                                (row N3: JPEG files -> hashes; 5 calls of 100 000 files
#                                                                                                    with the Huffman walk on the device + 2 of 8 000 with host entropy decoding)
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
run() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -o run -- python3 $R/bench.py --no-cpu-baseline --no-reference-cases "$@" > $OUT/$name.json 2> $OUT/$name.err
  python3 $R/tools/show_bench.py $OUT/$name.json
  grep -E "pdq_fused512|hamming_mfma|read_stream|jpeg_" $OUT/$name/run_kernel_stats.csv | cut -c1-200
}
run default --no-e2e --no-jpeg --hashes-strong 0
run e2e --only e2e
run thr40 --only hamming --threshold 40
run thr63 --only hamming --threshold 63
run config5 --only hamming_10m
run jpeg --only jpeg
