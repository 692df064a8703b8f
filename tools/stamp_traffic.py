"""After `tools/profile.sh <tag>` + `tools/pmc_summary.py gpurun_out/prof_<tag> <tag>`: put the cfg2 launch pair's HBM bytes
of profiles/<tag>_traffic.json into profiles/pmc_traffic.json (the `roofline.traffic` bench.py reports) and stamp it with
the profile tag and the hash of csrc/spmv_sliced.hip the profile was taken on -- run it on the same tree as the profile.
    python tools/stamp_traffic.py r03e"""
import hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "spmv_cfg2"  # "spmv_cfg2_nt": the profile was taken with SPBLAS_GFX950_PB_NT=1
t = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json")))
e = [k for k in t if "pb_expand_kernel<float" in k][0]
r = [k for k in t if "pb_reduce_kernel<float" in k][0]
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
det = d["_detail"]
prev = (d["_stamp"].get(key) or {}).get("profile")
if prev and key in d:
    det[f"previous {key} ({prev})"] = {key: d[key]}
d[key] = t[e]["hbm_bytes"] + t[r]["hbm_bytes"]
det[f"{key}: source"] = (f"profiles/{tag}_summary.md (rocprofv3 --pmc, separate passes, tools/profile.sh {tag}; bytes from the "
                         "request-size counters, see 'source')")
det[f"{key}: expand read / write"] = [t[e]["read_bytes"], t[e]["write_bytes"]]
det[f"{key}: reduce read / write"] = [t[r]["read_bytes"], t[r]["write_bytes"]]
h = hashlib.sha256(open(os.path.join(ROOT, "spblas-reference_amd", "csrc", "spmv_sliced.hip"), "rb").read()).hexdigest()[:16]
d["_stamp"][key] = {"profile": tag, "spmv_sliced_hip_sha256_16": h,
                    "note": f"measured on exactly this file (tools/profile.sh {tag})"}
json.dump(d, open(p, "w"), indent=1)
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "stats", "stats_kernel_stats.csv")
if os.path.exists(src):
    shutil.copy(src, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
print(f"{key} = {d[key]:.0f} B per SpMV ({tag}, {h})")
