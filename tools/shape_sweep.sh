#!/bin/bash
# tools/shape_sweep.sh: reduce-kernel shapes of the sliced plan on cfg2 (wave-bins per workgroup x LDS per workgroup x
# bins x batch depth), one bench.py run each, plus kernel-trace stats of the default shape.  A/B inside ONE gpurun call.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-sweep}
mkdir -p $OUT
cd $ROOT
run() {  # label, env...
  local label=$1; shift
  local line=$(env "$@" timeout 300 python3 bench.py --full-line --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
  python3 - "$label" <<PY "$line"
import json,sys
try:
    d=json.loads(sys.argv[2]); p=d["config"]["plan"]
    print(f"{sys.argv[1]:44s} {d['roofline']['kernel_avg_ms']*1e3:8.1f} us  {d['value']:7.1f} GF  H={p['rows_per_bin']} S={p['n_slices']} bytes={p['device_bytes']/1e9:.2f}GB")
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
}
run "default"                                   X=1
run "RW=8 LDS=160 (H~4.9k, 8 waves/CU)"         SPBLAS_GFX950_PB_RWAVES=8 SPBLAS_GFX950_PB_RLDS_KB=160
run "RW=4 LDS=160 BINS=1024 (H~9.8k)"           SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024
run "RW=8 LDS=160 BINS=4096 (H~2.4k,16 w/CU)"   SPBLAS_GFX950_PB_RWAVES=8 SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=4096
run "RW=8 LDS=80 BINS=4096 (H~2.4k)"            SPBLAS_GFX950_PB_RWAVES=8 SPBLAS_GFX950_PB_BINS=4096
run "RW=4 LDS=40 BINS=4096 (H~2.4k, 4 WG/CU)"   SPBLAS_GFX950_PB_RLDS_KB=40 SPBLAS_GFX950_PB_BINS=4096
run "default RBATCH=2"                          SPBLAS_GFX950_PB_RBATCH=2
run "XLDS=80 (489 slices)"                      SPBLAS_GFX950_PB_XLDS_KB=80
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/qs -o qs -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/qs.log 2>&1
cd $ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/qs/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i>=10: break
    print(f"{r['Name'].split('(')[0][-70:]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
