"""After `tools/profile.sh <tag> [bench args]` (which ends in tools/pmc_summary.py): put the HBM bytes per step of one
workload into profiles/pmc_traffic.json (the `roofline.traffic` bench.py reports) and stamp them with the profile tag and
the hash of the kernel source the profile was taken on -- run it on the same tree as the profile.

    python tools/stamp_traffic.py r04b                      # cfg2: pb_expand<float + pb_reduce<float of spmv_sliced.hip
    python tools/stamp_traffic.py r04c spmv_cfg2_nt         # the same with SPBLAS_GFX950_PB_NT=1
    python tools/stamp_traffic.py r04d spmv_rmat spmv_sliced.hip 'pb_expand_kernel<double' 'pb_reduce_kernel<double' \
        'pb_split_finish_kernel<double' 'pb_empty_rows_kernel<double' 'pb_hot_rows_kernel<double'
    python tools/stamp_traffic.py r04e spmm_cfg3 spmm.hip 'spmm_rowgroup_kernel<float'
    python tools/stamp_traffic.py r04f spgemm_cfg5 spgemm.hip 'spg_hash_kernel<float'

Each named kernel counts ONCE per step with its mean bytes per dispatch unless the pattern ends in `*N` (N launches per step:
`'spt_count_kernel*3'`); a pattern that matches nothing is reported and skipped."""
import hashlib, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "spmv_cfg2"  # "spmv_cfg2_nt": the profile was taken with SPBLAS_GFX950_PB_NT=1
src = sys.argv[3] if len(sys.argv) > 3 else "spmv_sliced.hip"
pats = sys.argv[4:] or ["pb_expand_kernel<float", "pb_reduce_kernel<float"]
t = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json")))
p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
d = json.load(open(p))
det = d["_detail"]
prev = (d["_stamp"].get(key) or {}).get("profile")
if prev and key in d:
    det[f"previous {key} ({prev})"] = {key: d[key]}
total, parts = 0.0, {}
for pat in pats:
    mult = 1  # 'pattern*N': the kernel is launched N times per step (the radix passes of the transpose)
    if "*" in pat and pat.rsplit("*", 1)[1].isdigit():
        pat, mult = pat.rsplit("*", 1)[0], int(pat.rsplit("*", 1)[1])
    hits = [k for k in t if pat in k]
    if not hits:
        print(f"  (no kernel matches {pat!r})")
        continue
    for k in hits:
        total += mult * t[k]["hbm_bytes"]
        parts[k + (f" x{mult}" if mult != 1 else "")] = [mult * t[k]["read_bytes"], mult * t[k]["write_bytes"]]
d[key] = total
det[f"{key}: source"] = (f"profiles/{tag}_summary.md (rocprofv3 --pmc, separate passes, tools/profile.sh {tag}; bytes from the "
                         "request-size counters, see 'source')")
det[f"{key}: read / write per kernel"] = parts
h = hashlib.sha256(open(os.path.join(ROOT, "spblas-reference_amd", "csrc", src), "rb").read()).hexdigest()[:16]
d["_stamp"][key] = {"profile": tag, "source_file": src, src.replace(".", "_") + "_sha256_16": h,
                    "note": f"measured on exactly this file (tools/profile.sh {tag})"}
json.dump(d, open(p, "w"), indent=1)
stats = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "stats", "stats_kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
print(f"{key} = {d[key]:.0f} B per step ({tag}, {src} {h})")
