#!/bin/bash
# round 4: SpGEMM one-shot fill at cfg5 -- rank-sort buckets, packed B (same box)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_spgemm.py tests/test_gpu_add.py -x -q 2>&1 | tail -2
for v in default nbk64 nbk32; do
 for PACK in 1 0; do
  L=""; [ $v != default ] && L=$PWD/tools/ab/lib$v.so
  SPBLAS_GFX950_SPG_PACK=$PACK SPBLAS_GFX950_LIB=$L python bench.py --workload spgemm --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v pack=$PACK', round(d['ms_per_step'],4), d['parity_check'], round(d['config']['repeated_fills']['ms_per_fill'],4))"
 done
done
