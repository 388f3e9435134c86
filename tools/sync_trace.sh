#!/bin/bash
# tools/sync_trace.sh -- kernel trace of tools/jpeg_stage_run.py photos (GPU box): the duration of every launch of the segment synchronisation
# (round 0, validation rounds, count pass) and of the walk for the last chunks of the run -> gpurun_out/sync_trace/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/sync_trace
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/run -o run -- python3 $R/tools/jpeg_stage_run.py photos 1 > $OUT/stdout.txt 2> $OUT/stderr.txt || { tail -5 $OUT/stderr.txt; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/run/**/run_kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "jpeg_sync" in r["Kernel_Name"] or "jpeg_huff" in r["Kernel_Name"] or "seg_items" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[-48]["Start_Timestamp"])
for r in rows[-48:]:
    print("%-22s queue %s start %7.2f ms  dur %6.2f ms"%(r["Kernel_Name"].split("(")[0][-22:], r["Queue_Id"], (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
