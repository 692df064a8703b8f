#!/bin/bash
# r03q: does timing up to three placements of the product workspace at inspect pay?  bench.py with the search on (default)
# and off (SPBLAS_GFX950_PB_PLACE=1), interleaved, four times each on one box.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3 4; do
for pl in 3 1; do
  SPBLAS_GFX950_PB_PLACE=$pl python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['config']['plan']['sliced']; print('place=$pl', round(d['ms_per_step']*1e3,1), 'us  nt', s.get('nt_product_stores'), ' timed', s.get('workspace_placements_timed'))"
done
done
