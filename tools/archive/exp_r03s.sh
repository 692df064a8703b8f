#!/bin/bash
# r03s: the product workspace through the virtual-memory API (hipMemCreate + hipMemMap) against the search among four pool
# allocations and against the first pool allocation; bench.py, interleaved, five times each on one box.
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2 3 4 5; do
for mode in vmm place4 place1; do
  case $mode in
    vmm) export SPBLAS_GFX950_PB_VMM=1; unset SPBLAS_GFX950_PB_PLACE;;
    place4) unset SPBLAS_GFX950_PB_VMM; export SPBLAS_GFX950_PB_PLACE=4;;
    place1) unset SPBLAS_GFX950_PB_VMM; export SPBLAS_GFX950_PB_PLACE=1;;
  esac
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d['config']['plan']['sliced']; print('$mode', round(d['ms_per_step']*1e3,1), 'us  frac', round(d['roofline']['frac'],4), ' nt', s.get('nt_product_stores'), s.get('store_trial_ns'), d['parity_check'])"
done
done
