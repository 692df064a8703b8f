#!/bin/bash
# round 4: fp64 plans with 64-byte blocks (8 entries) against 128-byte blocks (16): padding, bytes, time at cfg4; parity first
cd ${GRAFT_REPO_ROOT:-.}
V=${1:-blk8}
SPBLAS_GFX950_LIB=$PWD/tools/ab/lib$V.so python -m pytest tests/test_gpu_spmv.py -x -q -k "float64 or f64 or hot or rmat or ragged" 2>&1 | tail -2
for rep in 1 2 3; do
for v in default $V; do
  L=""; [ $v != default ] && L=$PWD/tools/ab/lib$v.so
  SPBLAS_GFX950_LIB=$L timeout 300 python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pl=d['config']['plan']; si=pl['sliced']
print('$v', round(d['ms_per_step'],4), 'ms', d.get('parity_check'), 'expand_blocks', si['expand_blocks'], 'reduce_blocks', si['reduce_blocks'], 'placed', si['placed_entries'], 'bytes', pl['device_bytes'])"
done
done
