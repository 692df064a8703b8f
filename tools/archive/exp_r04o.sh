#!/bin/bash
# round 4: cfg4 -- expand work-list granularity (items per CU) and reduce work list, same box
cd ${GRAFT_REPO_ROOT:-.}
run() {
  env "$@" timeout 300 python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pl=d['config']['plan']
print('$*', round(d['ms_per_step'],4), 'ms', d.get('parity_check'), 'xitems', pl.get('expand_items'), 'ritems', pl.get('reduce_items'))"
}
run A=0
for v in 1 3 4 6 8; do run SPBLAS_GFX950_PB_XITEM_DIV=$v; done
run A=0
for v in 1 2 3; do run SPBLAS_GFX950_PB_RITEMS=$v; done
run A=0
