#!/bin/bash
# r03: SpMM cfg3 -- B-row gathers vs a column-sliced variant (B window in L2, C read-modify-write), time and fabric bytes
tools/ubench/spmm_colslice
cd /tmp; export TMPDIR=/tmp
for SET in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_colslice/$(echo $SET | cut -c1-20 | tr ' ' _) -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/spmm_colslice > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("gpurun_out/pmc_colslice/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in acc.items():
    if "kernel" not in k: continue
    rd = 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0)
    w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0); wr = 64 * w64 + 32 * max(c.get("TCC_EA0_WRREQ_sum", 0) - w64, 0)
    print(f"{k:24s} launches {n[k].get('TCC_REQ_sum', 0):5d}  total over all launches of the run: read {rd/1e9:8.2f} GB  write {wr/1e9:8.2f} GB  L2 hit {c.get('TCC_HIT_sum',0)/max(c.get('TCC_REQ_sum',1),1):.2f}")
PY
python bench.py --workload spmm --no-cpu-baseline --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 bench', d['ms_per_step'], d['value'])"
