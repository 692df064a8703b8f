#!/bin/bash
# r03q: does choosing the product workspace among 8 allocations by a write test at inspect pay?  bench.py with the search on
# (default) and off (SPBLAS_GFX950_PB_PLACE=1), interleaved, four times each on one box; first the candidates' speeds once.
cd ${GRAFT_REPO_ROOT:-.}
SPBLAS_GFX950_TRACE_INSPECT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | grep "candidate" | cut -c1-110
for rep in 1 2 3 4; do
for pl in 8 1; do
  SPBLAS_GFX950_PB_PLACE=$pl python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['config']['plan']['sliced']; print('place=$pl', round(d['ms_per_step']*1e3,1), 'us  nt', s.get('nt_product_stores'), s.get('store_trial_ns'), ' candidates', s.get('workspace_candidates'))"
done
done
