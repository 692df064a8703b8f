#!/bin/bash
# cfg4's reduce under the kernel's timing-experiment switch (results wrong): SPBLAS_GFX950_PB_DBG 1 = no atomic path for the
# flagged entries, 2 = the stream alone (no LDS traffic); kernel times from rocprofv3 --stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in 0 1 2; do
  export SPBLAS_GFX950_PB_DBG=$d
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cfg4dbg_$d -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --workload spmv_rmat1 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
  python3 - $d <<'PY'
import csv,glob,os,sys
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/cfg4dbg_"+sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("pb_reduce","pb_expand","pb_hot_rows","pb_split","pb_combine","pb_empty")): print("DBG="+sys.argv[1], r["Name"][:48], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
