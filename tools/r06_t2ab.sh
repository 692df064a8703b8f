#!/bin/bash
# A/B timing of the two-pass op = T scatter kernel: tools/ab/lib<variant>.so built beforehand (tools/build_variant.sh)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "$@"; do
  export SPBLAS_GFX950_LIB=$GRAFT_REPO_ROOT/tools/ab/lib$v.so
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t2ab_$v -o t2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload csc_spmv --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
  python3 - "$v" <<'PY'
import csv,glob,os,sys
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/t2ab_"+sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "t2_" in r["Name"]: print(sys.argv[1], r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
