#!/bin/bash
# tools/prof_spmm.sh <tag>: rocprofv3 evidence for SpMM (north_star: "MFMA utilisation for SpMM"): kernel-trace stats and
# PMC passes (matrix-core counters, L2 hit rate, fabric read/write requests) for cfg3 (uniform random columns) and for
# its banded variant (the panel kernel).  --pmc passes carry --kernel-trace only.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/spmm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counters.txt
for W in spmm spmm_banded; do
  BENCH="python3 $ROOT/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -o s -- $BENCH > $OUT/${W}_stats.log 2>&1
  i=0
  for SET in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAVES" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum" \
             "TCC_EA0_WRREQ_64B_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_DRAM_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/${W}_pmc$i -o p -- $BENCH > $OUT/${W}_pmc$i.log 2>&1
  done
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json
out = {}
for W in ("spmm", "spmm_banded"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s_pmc*/**/*counter_collection.csv" % W, recursive=True):
        per = collections.defaultdict(float); names = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (d, c), v in per.items():
            acc[names[d].split("(")[0][-60:]][c].append(v)
    st = {}
    for f in glob.glob("$OUT/%s_stats/**/*kernel_stats.csv" % W, recursive=True):
        for r in csv.DictReader(open(f)):
            if "spb::spmm" in r["Name"]:
                st[r["Name"].split("(")[0][-60:]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
    out[W] = {"kernel_stats": st, "pmc_mean_per_dispatch": {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if "spb::spmm" in k}}
    for k, cs in out[W]["pmc_mean_per_dispatch"].items():
        need = ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")
        if all(n in cs for n in need):
            cs["derived_fabric_read_bytes"] = 32 * cs[need[0]] + 64 * cs[need[1]] + 128 * cs[need[2]]
            cs["derived_fabric_write_bytes"] = 64 * cs[need[4]] + 32 * (cs[need[3]] - cs[need[4]])
        if "TCC_HIT_sum" in cs:
            cs["derived_l2_hit_rate"] = cs["TCC_HIT_sum"] / max(1.0, cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
print(open("$OUT/mfma_counters.txt").read())
PY
