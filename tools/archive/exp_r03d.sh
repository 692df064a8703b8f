#!/bin/bash
# r03: row shards of cfg2 with the one-byte row codes: K split of the reduce, per-kernel times, fused-step overhead
one() { timeout 300 python bench.py --rows $ROWS --cols 10000000 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us slices', p['n_slices'], 'bins', p['sliced']['n_bins'], 'ksplit', p['sliced']['ksplit'], 'u8', p['sliced']['row_code_u8'], 'rows/bin', p['rows_per_bin'])"; }
for ROWS in 5000000 2500000 1250000 625000; do
  export ROWS
  echo "== rows $ROWS"
  echo -n "default: "; one
  echo -n "ENC8=0: "; SPBLAS_GFX950_PB_ENC8=0 one
  for k in 1 2 4; do echo -n "KSPLIT=$k: "; SPBLAS_GFX950_PB_KSPLIT=$k one; done
  echo -n "KSPLIT=1 BINS=1024: "; SPBLAS_GFX950_PB_KSPLIT=1 SPBLAS_GFX950_PB_BINS=1024 SPBLAS_GFX950_PB_RUN_MIN=32 one
done
tools/kstats.sh sh125 -- --rows 1250000 --cols 10000000
tools/kstats.sh sh125k1 SPBLAS_GFX950_PB_KSPLIT=1 -- --rows 1250000 --cols 10000000
tools/kstats.sh sh250 -- --rows 2500000 --cols 10000000
python tools/fused_overhead.py 1250000 2>/dev/null
python tools/fused_overhead.py 2500000 2>/dev/null
