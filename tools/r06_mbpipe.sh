#!/bin/bash
# A/B of the band kernel's entry loop: tools/ab/lib<variant>.so (tools/build_variant.sh), banded bench, twice each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', d['parity_check'])"; }
for r in 1 2; do for v in "$@"; do SPBLAS_GFX950_LIB=$GRAFT_REPO_ROOT/tools/ab/lib$v.so one "$v"; done; done
