#!/bin/bash
# r03 cfg4: work lists ordered heaviest first (expand items, reduce bin groups)
W="--workload spmv_rmat --alg sliced"
tools/kstats.sh lpt1 -- $W
tools/kstats.sh lpt0 SPBLAS_GFX950_PB_LPT=0 -- $W
tools/kstats.sh lpt1b -- $W
for l in 1 0; do SPBLAS_GFX950_PB_LPT=$l python bench.py $W --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('LPT=$l', d['ms_per_step'], d['roofline']['frac'], d['config']['plan']['expand_items'], d['config']['plan']['reduce_items'])"; done
