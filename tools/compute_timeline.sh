#!/bin/bash
# tools/compute_timeline.sh: kernel timeline (rocprofv3 kernel trace) of the LAST multiply_compute of `bench.py --workload spgemm`:
# start / duration of every kernel between spg_adesc_kernel and spg_direct_lists_kernel, and the gaps between them
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ctl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o ctl -- python3 $ROOT/bench.py --full-line --workload spgemm --steps 5 --warmup 2 --no-cpu-baseline > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-48:]) for r in csv.DictReader(open(f))]
rows.sort()
starts=[i for i,r in enumerate(rows) if 'spg_adesc' in r[2] or 'spg_bound' in r[2]]
# last compute: from the last adesc (or bound) backwards to the previous end
i0=[i for i,r in enumerate(rows) if 'spg_bound' in r[2]][-1]
i1=[i for i,r in enumerate(rows) if 'spg_direct_lists' in r[2] and i>i0]
i1=i1[0] if i1 else min(len(rows)-1,i0+20)
t0=rows[i0][0]
prev=t0
for s,e,n in rows[i0:i1+1]:
    print(f"{(s-t0)/1e3:9.1f} us  +gap {(s-prev)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {n}")
    prev=e
print(f"total {(rows[i1][1]-t0)/1e3:.1f} us")
print(open("$OUT/run.log").read()[-600:][:0])
PY
grep "^{" $OUT/run.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); c=d['config']; print('compute warm ms', c['multiply_compute_ms_untimed'], 'first', c['multiply_compute_first_call_ms'], 'fill ms', d['ms_per_step'])"
