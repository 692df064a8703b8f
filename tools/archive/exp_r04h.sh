#!/bin/bash
# round 4: pre-summing plan on cfg4 against the hot-column split and the plain tiles (same box)
cd ${GRAFT_REPO_ROOT:-.}
run() {
  env "$@" timeout 300 python bench.py --workload spmv_rmat1 --steps 20 --warmup 5 2>gpurun_out/r04h.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); pl=d['config']['plan']; si=pl.get('sliced',{})
print('$*', round(d['ms_per_step'],4), 'ms', d.get('parity_check'), 'inspect', round(d['config']['inspect_ms_untimed'],1), 'bytes', pl['device_bytes'], 'alg', pl['alg'], si.get('hot_split'))" || tail -5 gpurun_out/r04h.err
}
run SPBLAS_GFX950_PB_PS=1
run SPBLAS_GFX950_PB_PS=0
run SPBLAS_GFX950_PB_PS=1
