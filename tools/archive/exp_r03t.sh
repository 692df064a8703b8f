#!/bin/bash
# r03t: workspace search on (4 candidates) / off, as shipped otherwise (store trial on), three interleaved runs each.
cd ${GRAFT_REPO_ROOT:-.}
unset SPBLAS_GFX950_PB_VMM
for rep in 1 2 3; do
for pl in 4 1; do
  SPBLAS_GFX950_PB_PLACE=$pl python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['config']['plan']['sliced']; print('place=$pl', round(d['ms_per_step']*1e3,1), 'nt', s.get('nt_product_stores'))"
done
done | tr "\n" ";"
echo
