#!/bin/bash
# r03r: inspect cost with and without the workspace search (bench.py's inspect_ms_untimed, second inspect of the process
# would be cheaper: this is the first, with the store trial), and the SpMV time, six runs each, interleaved.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3 4 5 6; do
for pl in 8 4 1; do
  SPBLAS_GFX950_PB_PLACE=$pl python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['config']['plan']['sliced']; print('place=$pl', round(d['ms_per_step']*1e3,1), 'us  nt', s.get('nt_product_stores'), ' inspect', round(d['config']['inspect_ms_untimed'],1), 'ms')"
done
done
