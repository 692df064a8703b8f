#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_spmv.py -q -x -k "two_pass_form or csc_and_transposed" 2>&1 | tail -8
python bench.py --workload csc_spmv --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('csc', round(d['ms_per_step'],3), d['parity_check'], d['parity'])"
python - <<'PY'
import json
d=json.load(open("bench_secondary_csc_spmv.json")); c=d["config"]; print({k:c[k] for k in ("uninspected_ms_per_step","uninspected_roofline_frac")})
PY
SPBLAS_GFX950_SPMV_T2=0 python bench.py --workload csc_spmv --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null >/dev/null; python - <<'PY'
import json
d=json.load(open("bench_secondary_csc_spmv.json")); c=d["config"]; print("T2=0", {k:c[k] for k in ("uninspected_ms_per_step","uninspected_roofline_frac")})
PY
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t2stats -o t2 -- python3 $GRAFT_REPO_ROOT/bench.py --workload csc_spmv --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/t2stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "t2_" in r["Name"] or "plan_window_rows" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
