#!/bin/bash
# round 4: expand lanes past a run's end neither load nor store (default) against the previous behaviour (libnopad), same box
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3; do
for v in default nopad; do
  L=""; [ $v != default ] && L=$PWD/tools/ab/lib$v.so
  SPBLAS_GFX950_LIB=$L timeout 300 python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $v', round(d['ms_per_step'],4), 'ms', d.get('parity_check'))"
  SPBLAS_GFX950_LIB=$L timeout 300 python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 $v', round(d['ms_per_step'],4), 'ms', d.get('parity_check'))"
done
done
