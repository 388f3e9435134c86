#!/bin/bash
# tools/jpeg_stage_profile.sh TAG -- per-stage rooflines of the JPEG path (run on the GPU box): rocprofv3 kernel trace + stats over
# tools/jpeg_stage_run.py for 100 000 small files, 20 000 photos and the same photos as progressive files -> gpurun_out/jpegstage_TAG/{small,photos,prog}_kernel_stats.csv,
# {small,photos,prog}_counts.json and stage_roofline.json (tools/jpeg_stage_roofline.py)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/jpegstage_$TAG
rm -rf $OUT && mkdir -p $OUT
for kind in small photos prog; do
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$kind -o run -- python3 $R/tools/jpeg_stage_run.py $kind 4 > $OUT/${kind}_stdout.txt 2> $OUT/${kind}_stderr.txt || { tail -5 $OUT/${kind}_stderr.txt; exit 1; }
  grep '^{' $OUT/${kind}_stdout.txt | tail -1 > $OUT/${kind}_counts.json
  cp $(find $OUT/$kind -name "run_kernel_stats.csv" | head -1) $OUT/${kind}_kernel_stats.csv
done
python3 $R/tools/jpeg_stage_roofline.py $OUT > $OUT/stage_roofline.json
python3 -c "import json,sys; d=json.load(open('$OUT/stage_roofline.json')); [print(k, json.dumps(v)) for k,v in d.items()]"
