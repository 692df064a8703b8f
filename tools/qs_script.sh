#!/bin/bash
# tools/qs_script.sh <tag> <script.py> [args]: kernel-trace stats of one python script -> top kernels
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/qs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o qs -- python3 $ROOT/"$@" > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i>=${TOPN:-10}: break
    print(f"{r['Name'].split('(')[0][-70:]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
tail -2 $OUT/run.log
