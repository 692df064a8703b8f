"""tools/mfma_summary.py <tag> [dir]: condenses the passes of tools/prof_mfma.sh (gpurun_out/mfma_<tag>) into
profiles/<tag>_mfma.json and stamps profiles/pmc_traffic.json (run it where profiles/ is tracked: the GPU box only returns gpurun_out/)."""
import csv, glob, collections, hashlib, json, os, sys
tag = sys.argv[1]
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out", "mfma_" + sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per.items():
        acc[names[d].split("(")[0][-60:]][c].append(v)
stats = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0][-60:]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3}
# (round 6: the tile kernel of rounds 3 - 5 with SPBLAS_GFX950_SPMM_BAND=0, the dense-window kernel with _BAND_DENSE=250;
# the default band kernel issues no MFMA -- whichever ran is found by its counters)
k = [n for n in acc if "spmm_panel_kernel" in n or "spmm_band_mfma_kernel" in n]
if not k:
    print("no matrix-core SpMM dispatch found"); sys.exit(1)
k = k[0]
key = "spmm_banded_mfma" if "spmm_panel_kernel" in k else "spmm_banded_mfma_window"
c = {n: sum(v) / len(v) for n, v in acc[k].items()}
cycles = c["SQ_BUSY_CYCLES"] / 32.0
res = {"kernel": k, "avg_us": stats.get(k, {}).get("avg_us"), "counters_mean_per_dispatch": c,
       "kernel_cycles": cycles, "mfma_busy_cycles_per_simd": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0,
       "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cycles if cycles else None,
       "mfma_instructions": c.get("SQ_INSTS_MFMA"), "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
       "definition": "SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (SQ_BUSY_CYCLES / 32 SQ instances), rocprofv3 --pmc, mean per dispatch"}
json.dump(res, open(os.path.join(root, "profiles", f"{tag}_mfma.json"), "w"), indent=1)
p = os.path.join(root, "profiles", "pmc_traffic.json")
d = json.load(open(p))
d[key] = {kk: res[kk] for kk in ("mfma_util", "mfma_busy_cycles_per_simd", "kernel_cycles", "mfma_instructions", "avg_us", "definition")}
h = hashlib.sha256(open(os.path.join(root, "spblas-reference_amd", "csrc", "spmm.hip"), "rb").read()).hexdigest()[:16]
d["_stamp"][key] = {"profile": tag, "source_file": "spmm.hip", "spmm_hip_sha256_16": h,
                                   "note": f"measured on exactly this file (tools/prof_mfma.sh {tag})"}
json.dump(d, open(p, "w"), indent=1)
print(json.dumps(res, indent=1)[:1500])
