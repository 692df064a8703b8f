#!/bin/bash
# banded SpMM: where the panel kernel's time goes (SPBLAS_GFX950_SPMM_DBG bit 1 = no MFMA, 2 = no A scatter, 4 = no B staging; results wrong)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in 0 1 2 4 3 5 6 7; do
  SPBLAS_GFX950_SPMM_DBG=$d python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg=$d', round(d['ms_per_step'],3), 'ms', d['parity_check'])"
done
