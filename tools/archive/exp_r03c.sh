#!/bin/bash
tools/kstats.sh e8b1 SPBLAS_GFX950_PB_RBATCH=1
tools/kstats.sh e8b2
tools/kstats.sh e8b4 SPBLAS_GFX950_PB_RBATCH=4
tools/kstats.sh e0b1 SPBLAS_GFX950_PB_ENC8=0 SPBLAS_GFX950_PB_RBATCH=1
tools/kstats.sh e0b2 SPBLAS_GFX950_PB_ENC8=0 SPBLAS_GFX950_PB_RBATCH=2
tools/kstats.sh e0b4 SPBLAS_GFX950_PB_ENC8=0
tools/kstats.sh e8b2w8 SPBLAS_GFX950_PB_RWAVES=8
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"; done
