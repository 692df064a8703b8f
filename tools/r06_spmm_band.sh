#!/bin/bash
# banded SpMM: the band kernel (vector FMAs over the stored entries) against the matrix-core kernel; phase experiments
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_spmm.py -q -x 2>&1 | tail -2
one() { python bench.py --workload spmm_banded --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), 'ms', d['parity_check'])"; }
SPBLAS_GFX950_SPMM_BAND=0 one "mfma panel"
one "band ch152 w8"
SPBLAS_GFX950_SPMM_BAND_WAVES=16 one "band ch152 w16"
SPBLAS_GFX950_SPMM_DBG=1 one "band w8 no contraction"
SPBLAS_GFX950_SPMM_DBG=4 one "band w8 no staging"
SPBLAS_GFX950_SPMM_DBG=5 one "band w8 neither"
SPBLAS_GFX950_SPMM_BAND_WAVES=16 SPBLAS_GFX950_SPMM_DBG=1 one "band w16 no contraction"
SPBLAS_GFX950_SPMM_BAND_WAVES=16 SPBLAS_GFX950_SPMM_DBG=4 one "band w16 no staging"
SPBLAS_GFX950_SPMM_BAND_WAVES=16 SPBLAS_GFX950_SPMM_DBG=5 one "band w16 neither"
