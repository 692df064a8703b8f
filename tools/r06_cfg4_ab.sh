#!/bin/bash
# cfg4 (and cfg2) under A/B builds of the library: tools/ab/lib<variant>.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { python bench.py --workload $2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', round(d['ms_per_step'],4), 'ms')"; }
for r in 1 2; do for v in "$@"; do SPBLAS_GFX950_LIB=$GRAFT_REPO_ROOT/tools/ab/lib$v.so one $v spmv_rmat1; SPBLAS_GFX950_LIB=$GRAFT_REPO_ROOT/tools/ab/lib$v.so one $v spmv; done; done
