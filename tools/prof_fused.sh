#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_fused; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pf -- python3 $ROOT/tools/fused_overhead.py 1250000 > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i>=10: break
    print(f"{r['Name'].split('(')[0][-70:]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
