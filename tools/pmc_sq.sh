#!/bin/bash
# Issue-side counters (SQ / TA / TCP) of the kernels of one bench workload: tools/pmc_sq.sh <tag> <bench args...>
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sq_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o pmc -- $BENCH > $OUT/p$i.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if not any(x in k for x in ("spb::pb_", "spb::spmv", "spb::spmm", "spb::spg_", "spb::spt_", "spb::trsv")): continue
    print("==", k)
    for n in sorted(c):
        v = c[n]; print(f"   {n:40s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
