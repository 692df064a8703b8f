#!/bin/bash
# minimum run length (blocks per (slice, bin) tile) on square cfg2-like matrices of several sizes, forced SLICED plan
for n in 1000000 2000000 3000000 4000000 6000000; do
  for r in 128 96; do
    echo -n "n=$n RUN_MIN=$r: "
    SPBLAS_GFX950_PB_RUN_MIN=$r timeout 300 python bench.py --full-line --no-cpu-baseline --steps 100 --warmup 10 --rows $n --alg sliced 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us bins', p['sliced']['n_bins'], 'slices', p['n_slices'], 'ksplit', p['sliced']['ksplit'])"
  done
done
