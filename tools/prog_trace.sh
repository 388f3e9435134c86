#!/bin/bash
# tools/prog_trace.sh [parts] -- kernel trace of one run of tools/jpeg_stage_run.py prog (GPU box): when each launch of the progressive walk
# started and how long it ran, with the host's timeline (RPH_JPEG_TRACE=2) -> gpurun_out/prog_trace/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prog_trace
rm -rf $OUT && mkdir -p $OUT
[ -n "$1" ] && export RPH_JPEG_PARTS=$1 RPH_JPEG_CHUNK_GB=4
export RPH_JPEG_TRACE=2
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -o run -- python3 $R/tools/jpeg_stage_run.py prog 1 > $OUT/stdout.txt 2> $OUT/stderr.txt || { tail -5 $OUT/stderr.txt; exit 1; }
cp $(find $OUT/run -name "run_kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/run/**/run_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=[i for i,r in enumerate(rows) if "jpeg_prog" in r["Kernel_Name"]]
# the last call: launches after the largest gap
rows=rows[last[len(last)//2]-3:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"]
    if any(k in n for k in ("jpeg_prog","idct","fillBuffer")):
        print("%-28s queue %s start %7.1f ms dur %6.1f ms"%(n.split("(")[0][-28:], r.get("Queue_Id"), (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
grep "rph_jpeg" $OUT/stderr.txt | tail -40
