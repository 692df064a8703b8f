#!/bin/bash
# HBM-side traffic of the kernels of one bench workload: tools/pmc_one.sh <tag> <bench args...>
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline $*"
i=0
for SET in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
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
    if not any(x in k for x in ("spb::spmv", "spb::pb_", "spb::spmm", "spb::spg", "spb::trsv", "spb::spt", "spb::scan")): continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    w64 = m.get("TCC_EA0_WRREQ_64B_sum", 0); wr = 64 * w64 + 32 * max(m.get("TCC_EA0_WRREQ_sum", 0) - w64, 0)
    hit = m.get("TCC_HIT_sum", 0) / max(m.get("TCC_REQ_sum", 1), 1)
    print(f"{k:60s} read {rd/1e9:7.3f} GB  write {wr/1e9:7.3f} GB  L2 hit {hit:.2f}  (n={len(next(iter(c.values())))})")
PY
