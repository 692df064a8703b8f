#!/bin/bash
# r03: what would one byte of row stream per entry and 160 KiB reduce workgroups buy? (timing only; u8rows results are wrong)
L=$PWD/tools/ab/libu8rows.so
tools/kstats.sh base
tools/kstats.sh u8rows SPBLAS_GFX950_LIB=$L
tools/kstats.sh r160 SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024
tools/kstats.sh r160b8 SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024 SPBLAS_GFX950_PB_RBATCH=8
tools/kstats.sh r160u8 SPBLAS_GFX950_LIB=$L SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024
tools/kstats.sh r160u8b8 SPBLAS_GFX950_LIB=$L SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024 SPBLAS_GFX950_PB_RBATCH=8
tools/kstats.sh base2
