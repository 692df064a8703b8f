#!/bin/bash
# round 6: the fuzzers of round 5 over the kernels this round changed (band SpMM kernels -- default, matrix-core sibling, small
# chunks --, SpGEMM with and without addend incl. big shapes, SpMV incl. value-free plans, transpose, triangular solve)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 900 python tools/fuzz_spmm.py 150 600 2>&1 | tail -2
  SPBLAS_GFX950_SPMM_PANEL_MIN=32 timeout 900 python tools/fuzz_spmm.py 120 900 2>&1 | tail -2
  SPBLAS_GFX950_SPMM_PANEL_MIN=32 SPBLAS_GFX950_SPMM_BAND_DENSE=0 timeout 900 python tools/fuzz_spmm.py 120 1200 2>&1 | tail -2
  SPBLAS_GFX950_SPMM_PANEL_MIN=32 SPBLAS_GFX950_SPMM_BAND_CH=64 SPBLAS_GFX950_SPMM_BAND_WAVES=4 timeout 900 python tools/fuzz_spmm.py 100 1500 2>&1 | tail -2
  timeout 900 python tools/fuzz_spgemm.py 150 300 2>&1 | tail -2
  FUZZ_BIG=1 timeout 900 python tools/fuzz_spgemm.py 25 700 2>&1 | tail -2
  timeout 900 python tools/fuzz_spmv.py 200 5000 2>&1 | tail -2
  timeout 600 python tools/fuzz_transpose.py 60 100 2>&1 | tail -2
  timeout 600 python tools/fuzz_sptrsv.py 40 100 2>&1 | tail -2 ) > gpurun_out/r06_fuzz.log 2>&1
cat gpurun_out/r06_fuzz.log
