#!/bin/bash
# r03: enc8 -- where the extra inspect time goes, reduce batch depth
SPBLAS_GFX950_TRACE_INSPECT=1 python tools/inspect_repeat.py 2>&1 | tail -45
echo ---- enc0
SPBLAS_GFX950_PB_ENC8=0 SPBLAS_GFX950_TRACE_INSPECT=1 python tools/inspect_repeat.py 2>&1 | tail -16
tools/kstats.sh e8b2 SPBLAS_GFX950_PB_RBATCH=2
tools/kstats.sh e8b4
tools/kstats.sh e8b8 SPBLAS_GFX950_PB_RBATCH=8
tools/kstats.sh e8r160 SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024
