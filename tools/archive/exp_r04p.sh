#!/bin/bash
# round 4: fabric bytes of the pipelined XCD-local hand-off microbenchmark (tools/ubench/l2_handoff2_<cfg>, built by hand:
# hipcc -DH2_THREADS= -DH2_R= -DH2_DEPTH= -DH2_BATCH=)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for c in "$@"; do
  OUT=$ROOT/gpurun_out/r04p_$c; rm -rf $OUT; mkdir -p $OUT
  timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT -o pmc -- $ROOT/tools/ubench/l2_handoff2_$c > $OUT/log 2>&1
  timeout 120 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b -o pmc -- $ROOT/tools/ubench/l2_handoff2_$c > $OUT/log2 2>&1
  grep "^mode" $OUT/log | sed -n "2p;5p" | cut -c1-110
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'handoff2' in r['Kernel_Name']:
            acc[(r['Counter_Name'])].append(float(r['Counter_Value']))
# six dispatches per run: 0-2 hand-off only, 3-5 with streams
for k, v in sorted(acc.items()):
    if len(v) >= 6:
        print('  ', k, 'hand-off only', round(sum(v[0:3]) / 3), ' with streams', round(sum(v[3:6]) / 3))
PY
done
