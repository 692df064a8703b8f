#!/bin/bash
# 80 vs 160 KiB x slices on square cfg2-like matrices below the n = 8 M switch point, forced SLICED plan
for n in ${SIZES:-3000000 4000000 5000000 6000000 7000000 7900000}; do
  for x in 80 160; do
    echo -n "n=$n XLDS=$x: "
    SPBLAS_GFX950_PB_XLDS_KB=$x timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --rows $n --alg sliced 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us bins', p['sliced']['n_bins'], 'slices', p['n_slices'], 'xitems', p['expand_items'])"
  done
done
