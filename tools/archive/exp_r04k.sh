#!/bin/bash
# round 4: fabric read requests of the SpGEMM direct kernel with plain / non-temporal B gathers (variant library as $1)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for v in default "$@"; do
  L=""; [ $v != default ] && L=$ROOT/tools/ab/lib$v.so
  OUT=$ROOT/gpurun_out/r04k_$v; rm -rf $OUT; mkdir -p $OUT
  export SPBLAS_GFX950_LIB=$L
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT -o pmc -- python3 $ROOT/bench.py --workload spgemm --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log 2>&1
  tail -1 $OUT/log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['parity_check'])"
  python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'spg_direct_kernel' not in r['Kernel_Name']:
            continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for c, (v, n) in acc.items():
        print('  ', c, round(v / max(n, 1)))
PY
done
