#!/bin/bash
# cfg4 (R-MAT scale 24, fp64) under the plan's test knobs, one at a time: ms per multiply + parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { env "$@" python bench.py --workload spmv_rmat1 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('$*', round(d['ms_per_step'],3), 'ms', d['parity_check'], 'plan_bytes', c.get('plan_bytes'), 'inspect', c.get('inspect_ms'))"; }
one X=0
one SPBLAS_GFX950_PB_ENC8=2
one SPBLAS_GFX950_PB_HOT_DEPTH=2
one SPBLAS_GFX950_PB_HOT_DEPTH=3
one SPBLAS_GFX950_PB_XITEM_DIV=1
one SPBLAS_GFX950_PB_XITEM_DIV=3
one SPBLAS_GFX950_PB_XITEM_DIV=4
one SPBLAS_GFX950_PB_RWAVES=8
one SPBLAS_GFX950_PB_RBATCH=2
one SPBLAS_GFX950_PB_RBATCH=8
one SPBLAS_GFX950_PB_RLDS_KB=80
one SPBLAS_GFX950_PB_XLDS_KB=80
one SPBLAS_GFX950_PB_HOT_MIN_PCT=5
one SPBLAS_GFX950_PB_NT=0
one SPBLAS_GFX950_PB_NT=1
one SPBLAS_GFX950_PB_BINS=2048
one SPBLAS_GFX950_PB_BINS=8192
one SPBLAS_GFX950_PB_KSPLIT=0
one SPBLAS_GFX950_PB_RITEMS=2
one X=1
