#!/bin/bash
# tools/kall.sh <tag> [ENV=VAL ...] -- [bench args]: every kernel above 20 us of one bench run (rocprofv3 kernel trace)
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ka_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "${ENVS[@]}"; do export "$e"; done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ka -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
cd $ROOT
tail -1 $OUT/run.log | cut -c1-1500
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    if ('spb::' in n or 'spmv' in n or 'pb_' in n) and float(r['TotalDurationNs'])>20e3:
        print(f"{n[:70]:70s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us")
PY
