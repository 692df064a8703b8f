#!/bin/bash
# round 4: cfg4, the remainder of the hot split split again (same box)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  env "$@" python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pl=d['config']['plan']; si=pl.get('sliced',{})
print('$*', round(d['ms_per_step'],4), 'ms', d.get('parity_check'), 'inspect', round(d['config']['inspect_ms_untimed'],1), 'bytes', pl['device_bytes'], 'hot', (si.get('hot_split') or {}).get('hot_entries'), 'tiled', (si.get('hot_split') or {}).get('tiled_entries'))"
}
run SPBLAS_GFX950_PB_HOT_DEPTH=1
run SPBLAS_GFX950_PB_HOT_DEPTH=2
run SPBLAS_GFX950_PB_HOT_DEPTH=3
run SPBLAS_GFX950_PB_HOT_DEPTH=4 SPBLAS_GFX950_PB_HOT_MIN_PCT2=5
run SPBLAS_GFX950_PB_HOT_DEPTH=1
