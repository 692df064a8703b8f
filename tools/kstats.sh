#!/bin/bash
# tools/kstats.sh <tag> [ENV=VAL ...] -- [bench args]: per-kernel averages (rocprofv3 kernel trace) of the sliced
# launch pair under the given environment; one line.  A/B comparisons inside ONE gpurun call only.
TAG=$1; shift
ENVS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ks_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for e in "${ENVS[@]}"; do export "$e"; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
cd $ROOT
python3 - "$TAG" "${ENVS[*]}" <<PY
import csv,glob,sys
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)
if not f: print(sys.argv[1], "no stats"); sys.exit()
t={}
for r in csv.DictReader(open(f[0])):
    n=r['Name']
    for k in ("pb_expand","pb_reduce","pb_combine","pb_hot_rows","pb_hot_fixup","pb_presum","pb_split","pb_empty","spmv_rowblock","spmm_","spg_"):
        if k in n: t[k]=t.get(k,0)+float(r['AverageNs'])/1e3
print(f"{sys.argv[1]:28s} " + "  ".join(f"{k}={v:7.1f}us" for k,v in t.items()) + f"   [{sys.argv[2]}]")
PY
