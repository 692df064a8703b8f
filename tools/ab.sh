#!/bin/bash
# A/B on one box: lib/variants/libHEAD.so (previous kernels) vs the current build under several reduce shapes
L=spblas-reference_amd/lib
cp $L/libspblas_gfx950.so $L/variants/libNEW.so
run() { python bench.py --no-cpu-baseline --steps 300 --warmup 50 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['kernel_min_ms'])"; }
for rep in 1 2; do
cp $L/variants/libHEAD.so $L/libspblas_gfx950.so; echo -n "HEAD: "; run
cp $L/variants/libNEW.so $L/libspblas_gfx950.so
for cfg in ${SWEEP:-"8 1 1" "4 2 1" "4 1 2" "4 2 2" "8 1 2"}; do set -- $cfg; echo -n "NEW RW=$1 C=$2 E=$3: "; SPBLAS_GFX950_PB_RWAVES=$1 SPBLAS_GFX950_PB_RCHUNKS=$2 SPBLAS_GFX950_PB_RENTRIES=$3 run; done
done
