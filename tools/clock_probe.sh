#!/bin/bash
# tools/clock_probe.sh [extra bench args] -- sample rocm-smi (power, sclk) while bench.py runs the PDQ kernel back to back
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/bench.py --no-cpu-baseline --hashes 65536 --steps 600 --warmup 2 "$@" > $R/gpurun_out/clock_bench.json 2>/dev/null &
PID=$!
: > $R/gpurun_out/clock_samples.txt
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ' >> $R/gpurun_out/clock_samples.txt
  echo >> $R/gpurun_out/clock_samples.txt
done
wait $PID
sort -t'(' -k2 -n $R/gpurun_out/clock_samples.txt | uniq -c | sort -rn | head -12
python3 $R/tools/show_bench.py $R/gpurun_out/clock_bench.json "600 steps"
