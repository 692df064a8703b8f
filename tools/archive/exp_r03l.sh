#!/bin/bash
# fabric bytes of the XCD-local hand-off microbenchmark (400 MB of products pass from producers to consumers per launch)
cd /tmp; export TMPDIR=/tmp
for SET in "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" "TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  timeout 120 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_handoff/$(echo $SET | cut -c1-20 | tr ' ' _) -o pmc -- $GRAFT_REPO_ROOT/tools/ubench/l2_handoff > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_handoff/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "handoff" in r["Kernel_Name"]: per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items(): acc[c]["all"].append(v)
m = {c: v["all"] for c, v in acc.items()}
n = len(m.get("TCC_REQ_sum", []))
for i in range(n):
    g = lambda c: m.get(c, [0]*n)[i] if i < len(m.get(c, [])) else 0
    rd = 32*g("TCC_EA0_RDREQ_32B_sum") + 64*g("TCC_EA0_RDREQ_64B_sum") + 128*g("TCC_EA0_RDREQ_128B_sum")
    w64 = g("TCC_EA0_WRREQ_64B_sum"); wr = 64*w64 + 32*max(g("TCC_EA0_WRREQ_sum") - w64, 0)
    print(f"launch {i} ({'hand-off only' if i < 3 else 'with streams (3.0 GB of streamed loads)'}): fabric read {rd/1e9:6.3f} GB  write {wr/1e9:6.3f} GB  L2 hit {g('TCC_HIT_sum')/max(g('TCC_REQ_sum'),1):.2f}")
PY
