#!/bin/bash
# round 4: kernels of a value refresh (tools/update_values_time.py [rmat]) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04s_upd; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ka -- python3 $ROOT/tools/update_values_time.py "$@" > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if ('spb::' in n) and int(r['Calls'])>=10 and float(r['TotalDurationNs'])>100e3:
        print(f"{n[:80]:80s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us")
PY
