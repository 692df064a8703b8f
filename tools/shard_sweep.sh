#!/bin/bash
# plan-shape knobs on row shards of cfg2 (N = 2 / 4 / 8 / 16: 5 M / 2.5 M / 1.25 M / 0.625 M rows x 10 M columns),
# emulated on one GPU: tools/shard_sweep.sh [quick]
one() { timeout 300 python bench.py --full-line --rows $ROWS --cols 10000000 --no-cpu-baseline --steps 200 --warmup 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us slices', p['n_slices'], 'bins', p['sliced']['n_bins'], 'ksplit', p['sliced']['ksplit'], 'eblocks', p['sliced']['expand_blocks'], 'rows/bin', p['rows_per_bin'])"; }
for ROWS in 10000000 5000000 2500000 1250000 625000; do
  export ROWS
  echo "== rows $ROWS"
  echo -n "default: "; one
  if [ "${1:-}" != quick ]; then
    for x in 80 160; do for k in 1 2 4 8; do echo -n "XLDS=$x KSPLIT=$k: "; SPBLAS_GFX950_PB_XLDS_KB=$x SPBLAS_GFX950_PB_KSPLIT=$k one; done; done
  else
    for k in 1 2; do echo -n "KSPLIT=$k: "; SPBLAS_GFX950_PB_KSPLIT=$k one; done
  fi
done
