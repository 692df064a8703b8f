#!/bin/bash
# round 4: SpGEMM in-register direct path -- merged rank loop (default) against one loop per product (librm0), same box
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_spgemm.py tests/test_gpu_add.py -x -q 2>&1 | tail -2
for rep in 1 2; do
for v in default "$@"; do
  L=""; [ $v != default ] && L=$PWD/tools/ab/lib$v.so
  SPBLAS_GFX950_LIB=$L python bench.py --workload spgemm --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['parity_check'], round(d['config']['repeated_fills']['ms_per_fill'],4))"
done
done
