#!/bin/bash
one() { timeout 300 python bench.py --rows $ROWS --cols 10000000 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us bins', p['sliced']['n_bins'], 'ksplit', p['sliced']['ksplit'], 'u8', p['sliced']['row_code_u8'])"; }
for ROWS in 10000000 5000000 2500000 1250000 625000; do
  export ROWS
  echo "== rows $ROWS"
  echo -n "default: "; one
  echo -n "RBATCH=2: "; SPBLAS_GFX950_PB_RBATCH=2 one
  echo -n "RBATCH=4: "; SPBLAS_GFX950_PB_RBATCH=4 one
done
