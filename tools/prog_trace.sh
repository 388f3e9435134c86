#!/bin/bash
# tools/prog_trace.sh -- kernel trace of one run of tools/jpeg_stage_run.py prog (GPU box): per-launch durations of the progressive walk's
# levels and of the pass that applies the corrections -> gpurun_out/prog_trace/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prog_trace
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -o run -- python3 $R/tools/jpeg_stage_run.py prog 2 > $OUT/stdout.txt 2> $OUT/stderr.txt || { tail -5 $OUT/stderr.txt; exit 1; }
cp $(find $OUT/run -name "run_kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/run/**/run_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "jpeg_prog" in r["Kernel_Name"]]
t0=min(int(r["Start_Timestamp"]) for r in rows)
for r in rows[-16:]:
    print(r["Kernel_Name"][:60], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "start %.1f ms"%((int(r["Start_Timestamp"])-t0)/1e6), "dur %.1f ms"%((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
cut -c1-60,200- $OUT/kernel_stats.csv | head -12
