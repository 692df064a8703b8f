#!/bin/bash
W="--workload spmv_rmat --alg sliced"
tools/kstats.sh x2 -- $W
tools/kstats.sh x4 SPBLAS_GFX950_PB_XITEM_DIV=4 -- $W
tools/kstats.sh x8 SPBLAS_GFX950_PB_XITEM_DIV=8 -- $W
tools/kstats.sh x16 SPBLAS_GFX950_PB_XITEM_DIV=16 -- $W
tools/kstats.sh r2 SPBLAS_GFX950_PB_RITEMS=2 -- $W
tools/kstats.sh r4 SPBLAS_GFX950_PB_RITEMS=4 -- $W
tools/kstats.sh b4096 SPBLAS_GFX950_PB_BINS=4096 -- $W
