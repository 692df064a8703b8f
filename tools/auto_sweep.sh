#!/bin/bash
# does multiply_inspect(AUTO) pick the faster plan?  square cfg2-like matrices of several sizes, every algorithm
for n in 500000 1000000 2000000 3000000 4000000 6000000; do
  for alg in auto rowblock sliced; do
    python bench.py --full-line --no-cpu-baseline --steps 100 --warmup 10 --rows $n --alg $alg 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '$alg', 'plan', d['config']['plan'].get('alg'), round(d['value'],1), 'GFLOP/s', round(d['ms_per_step']*1e3,1), 'us')"
  done
done
