#!/bin/bash
# round 6, end: a long campaign with fresh seeds over every fuzzer (about 30 minutes of GPU time); failures, if any, are listed
# with their seeds in gpurun_out/r06_fuzz_long.log
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=${1:-0}  # seed offset: a second campaign takes other cases
L=gpurun_out/r06_fuzz_long.log; : > $L
run() { echo "== $*" >> $L; ( "$@" 2>&1 | grep -v "^ok\|^SKIP\|amdgpu.ids" | tail -12 ) >> $L; }
run timeout 420 python tools/fuzz_spmv.py 2500 $((20000 + O))
FUZZ_FORCE_VFREE=1 run timeout 300 python tools/fuzz_spmv.py 1200 $((30000 + O))
FUZZ_FORCE_HOT=1 run timeout 300 python tools/fuzz_spmv.py 1200 $((40000 + O))
run timeout 240 python tools/fuzz_spmv_t.py 4000 $((50000 + O))
run timeout 300 python tools/fuzz_spmm.py 600 $((60000 + O))
SPBLAS_GFX950_SPMM_PANEL_MIN=32 run timeout 300 python tools/fuzz_spmm.py 500 $((61000 + O))
SPBLAS_GFX950_SPMM_PANEL_MIN=32 SPBLAS_GFX950_SPMM_BAND_DENSE=0 run timeout 240 python tools/fuzz_spmm.py 400 $((62000 + O))
SPBLAS_GFX950_SPMM_PANEL_MIN=32 SPBLAS_GFX950_SPMM_BAND=0 run timeout 240 python tools/fuzz_spmm.py 300 $((63000 + O))
run timeout 360 python tools/fuzz_spgemm.py 700 $((70000 + O))
FUZZ_BIG=1 run timeout 600 python tools/fuzz_spgemm.py 12 $((71000 + O))
run timeout 240 python tools/fuzz_transpose.py 250 $((80000 + O))
run timeout 240 python tools/fuzz_sptrsv.py 150 $((90000 + O))
cat $L
