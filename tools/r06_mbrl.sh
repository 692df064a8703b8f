#!/bin/bash
# A/B of the band kernel's entry delivery: MB_RL of 8 entries per batch by v_readlane (tools/ab/libmb_rl<k>.so)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
one() { python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', d['parity_check'])"; }
for v in "$@"; do SPBLAS_GFX950_LIB=$GRAFT_REPO_ROOT/tools/ab/libmb_rl$v.so one "MB_RL=$v"; done
for v in "$@"; do SPBLAS_GFX950_LIB=$GRAFT_REPO_ROOT/tools/ab/libmb_rl$v.so one "MB_RL=$v"; done
