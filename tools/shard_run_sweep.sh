#!/bin/bash
one() { timeout 300 python bench.py --full-line --rows $ROWS --cols 10000000 --no-cpu-baseline --steps 200 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us slices', p['n_slices'], 'bins', p['sliced']['n_bins'], 'ksplit', p['sliced']['ksplit'], 'eblocks', p['sliced']['expand_blocks'], 'rows/bin', p['rows_per_bin'])"; }
for ROWS in 2500000 1250000 625000; do
  export ROWS
  echo "== rows $ROWS"
  for r in 128 96 64 48; do echo -n "RUN_MIN=$r: "; SPBLAS_GFX950_PB_RUN_MIN=$r one; done
done
