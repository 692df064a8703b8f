#!/bin/bash
# reduce-side shape knobs on the full cfg2 matrix: LDS per reduce workgroup, wave-bins per workgroup, bin count
one() { timeout 300 python bench.py --full-line --no-cpu-baseline --steps 100 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['config']['plan']; print(round(d['ms_per_step']*1e3,1), 'us bins', p['sliced']['n_bins'], 'rows/bin', p['rows_per_bin'], 'eblocks', p['sliced']['expand_blocks'], 'rblocks', p['sliced']['reduce_blocks'])"; }
echo -n "default: "; one
for b in 1024 1280 1536; do echo -n "RLDS=160 BINS=$b: "; SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=$b one; done
for b in 1536 2048 3072; do echo -n "RLDS=120 BINS=$b: "; SPBLAS_GFX950_PB_RLDS_KB=120 SPBLAS_GFX950_PB_BINS=$b one; done
for b in 3072 4096; do echo -n "RLDS=80 BINS=$b: "; SPBLAS_GFX950_PB_BINS=$b one; done
for b in 2048 4096; do echo -n "RLDS=80 RWAVES=8 BINS=$b: "; SPBLAS_GFX950_PB_RWAVES=8 SPBLAS_GFX950_PB_BINS=$b one; done
for r in 2 8; do echo -n "RBATCH=$r: "; SPBLAS_GFX950_PB_RBATCH=$r one; done
echo -n "default again: "; one
