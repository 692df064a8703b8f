#!/bin/bash
# round 4: HIP API trace of the first multiply_inspect of a process at cfg2 (which host calls make it 8.6 - 10 ms?)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04n; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --hip-trace --kernel-trace --output-format csv -d $OUT -o tr -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob
api = list(csv.DictReader(open(glob.glob('gpurun_out/r04n/**/*hip_api_trace.csv', recursive=True)[0])))
ker = list(csv.DictReader(open(glob.glob('gpurun_out/r04n/**/*kernel_trace.csv', recursive=True)[0])))
ker.sort(key=lambda r: int(r['Start_Timestamp']))
t_stats = [int(r['Start_Timestamp']) for r in ker if 'plan_row_stats' in r['Kernel_Name']][0]
t_scatter = [int(r['End_Timestamp']) for r in ker if 'pb_flag_dups8' in r['Kernel_Name']][0]
api.sort(key=lambda r: int(r['Start_Timestamp']))
print('window: 3 ms before the first plan kernel .. end of the first flag kernel; HIP calls > 50 us')
for r in api:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > t_stats - 3_000_000 and s < t_scatter + 1_500_000 and e - s > 50_000:
        print(f"{(s - t_stats) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  {r['Function']}")
PY
