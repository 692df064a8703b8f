#!/bin/bash
# r03 cfg4 (fp64 R-MAT scale 24): per-kernel times, forced one-byte codes, batch depth, issue-side and fabric counters
W="--workload spmv_rmat --alg sliced"
tools/kstats.sh rm0 -- $W
tools/kstats.sh rm_e8 SPBLAS_GFX950_PB_ENC8=2 -- $W
tools/kstats.sh rm_b2 SPBLAS_GFX950_PB_RBATCH=2 -- $W
tools/kstats.sh rm_b8 SPBLAS_GFX950_PB_RBATCH=8 -- $W
python bench.py --workload spmv_rmat --alg sliced --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['plan'])"
tools/pmc_sq.sh rmat $W 2>&1 | grep -A30 "pb_expand\|pb_reduce" | head -80
tools/pmc_one.sh rmat3 $W 2>&1 | tail -8
