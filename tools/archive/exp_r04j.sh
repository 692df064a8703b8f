#!/bin/bash
# round 4: instruction / stall / address-unit counters of one kernel:  tools/exp_r04j.sh [kernel-name-part [bench args]]
# (default: the SpGEMM direct kernel at cfg5)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04j; rm -rf $OUT; mkdir -p $OUT
i=0
KERNEL=${1:-spg_direct_kernel}; shift
BARGS=${@:---workload spgemm}
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY" \
           "TA_BUSY_avr TA_BUSY_max TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_WRITE_sum TCP_TOTAL_READ_sum" \
           "TCC_BUSY_avr TCC_REQ_sum TCC_WRITE_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o pmc -- python3 $ROOT/bench.py $BARGS --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/pmc$i.log 2>&1
done
cd $ROOT
KERNEL=$KERNEL python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r04j/pmc*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        import os
        if os.environ.get('KERNEL', 'spg_direct_kernel') not in k:
            continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for c, (v, n) in acc.items():
        print(c, round(v / max(n, 1)), 'per dispatch,', n, 'dispatches')
PY
