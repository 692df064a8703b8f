#!/bin/bash
# tools/rmat_try.sh "ENV=VAL ..." ...: cfg4 forced SLICED under each environment; ms per step + padding
for e in "$@"; do
  env $e timeout 600 python bench.py --full-line --workload spmv_rmat --alg ${RMAT_ALG:-sliced} --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/rmat_try.json
  python3 - "$e" <<'PY'
import json,sys
try:
    d=json.loads(open('/tmp/rmat_try.json').read())
    s=d['config']['plan'].get('sliced',{})
    blk=16 if d['dtype']=='f64' else 32
    pad=s.get('expand_blocks',0)*blk/max(1,s.get('placed_entries',1))
    print(f"{sys.argv[1]:50s} {d['ms_per_step']:.3f} ms  alg {d['config']['plan']['alg']} inspect {d['config']['inspect_ms_untimed']:.1f} ms bins {s.get('n_bins')} H {d['config']['plan']['rows_per_bin']} pad {pad:.3f} hub {s.get('hub_rows')} bytes {d['config']['plan']['device_bytes']/1e9:.2f} GB")
except Exception as ex:
    print(sys.argv[1], "failed", ex)
PY
done
