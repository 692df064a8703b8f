#!/bin/bash
# r03p: the same box runs cfg2 in a fast (~290 us) or a slow (~312 us) mode from process to process.  Does it depend on
# where the plan's arrays come from (stream-ordered pool vs plain hipMalloc)?  Alternating runs, plain stores forced.
cd ${GRAFT_REPO_ROOT:-.}
export SPBLAS_GFX950_PB_NT=0
for rep in 1 2 3 4 5; do
for np in 0 1; do
  if [ $np == 1 ]; then export SPBLAS_GFX950_NO_POOL=1; else unset SPBLAS_GFX950_NO_POOL; fi
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('no_pool=$np', round(d['ms_per_step']*1e3,1), 'us')"
done
done
