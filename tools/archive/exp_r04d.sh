#!/bin/bash
# round 4: cfg4 with the hot split -- bin shapes of the tiled remainder (same box)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  env "$@" python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pl=d['config']['plan']; si=pl.get('sliced',{})
print('$*', round(d['ms_per_step'],4), 'ms  bins', si.get('n_bins'), 'pad', round(si.get('reduce_blocks',0)*16/max(1,si.get('placed_entries',1)),4), 'inspect', round(d['config']['inspect_ms_untimed'],1), 'bytes', pl['device_bytes'], 'hot', (si.get('hot_split') or {}).get('hot_entries'))"
}
run A=0
run SPBLAS_GFX950_PB_RLDS_KB=160
run SPBLAS_GFX950_PB_BINS=1024
run SPBLAS_GFX950_PB_BINS=4096
run SPBLAS_GFX950_PB_RLDS_KB=160 SPBLAS_GFX950_PB_BINS=1024
run SPBLAS_GFX950_PB_SPLIT_LEN=4096
run SPBLAS_GFX950_PB_SPLIT_LEN=1024
run SPBLAS_GFX950_PB_HOT=0
run A=1
