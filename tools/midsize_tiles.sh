#!/bin/bash
# slice count at mid sizes: rounded up to a multiple of 512 (default) vs as few full-width slices as fit
# (SPBLAS_GFX950_PB_XROUND=1), us per SpMV
for n in 5000000 6000000 7000000 8000000 10000000; do
  for w in 512 1; do
    SPBLAS_GFX950_PB_XROUND=$w timeout 120 python bench.py --full-line --no-cpu-baseline --steps 100 --warmup 10 --rows $n --alg sliced 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print('$n', 'xround=$w', 'S', p['n_slices'], 'H', p['rows_per_bin'], round(d['ms_per_step']*1e3,1), 'us', round(d['ms_per_step']*1e6/d['config']['nnz'],3), 'ns/nnz')"
  done
done
