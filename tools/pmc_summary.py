"""Condenses rocprofv3 CSV output (tools/profile.sh) into profiles/<tag>_summary.md and
profiles/pmc_traffic.json.  Per kernel: average duration (kernel-trace stats) and the mean
per-dispatch value of every collected counter; FETCH_SIZE / WRITE_SIZE are converted to
bytes (they are in KiB... as reported by rocprofv3: 1 unit = 1 KB? calibrated below against
the known-byte copy kernel and the gfx950 2x FETCH correction of MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(pattern):
    return sorted(glob.glob(os.path.join(out_dir, pattern), recursive=True))


def short(name):
    n = name.split("(")[0]
    return n[-90:]


lines = [f"# rocprofv3 summary `{tag}`", ""]
# ---- kernel stats
stats = find("stats/**/*kernel_stats.csv")
if stats:
    lines += ["## kernel-trace --stats (top kernels)", "", "| kernel | calls | avg us | total % |", "|---|---|---|---|"]
    with open(stats[0]) as f:
        for i, row in enumerate(csv.DictReader(f)):
            if i >= 12:
                break
            lines.append(f"| `{short(row['Name'])}` | {row['Calls']} | {float(row['AverageNs']) / 1e3:.1f} | {row['Percentage']} |")
    lines.append("")


def counters(prefix):
    """kernel -> counter -> mean per-dispatch value"""
    acc = defaultdict(lambda: defaultdict(list))
    for path in find(f"{prefix}*/**/*counter_collection.csv"):
        with open(path) as f:
            per_dispatch = defaultdict(float)
            names = {}
            for row in csv.DictReader(f):
                key = (row["Dispatch_Id"], row["Counter_Name"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
            for (did, cname), val in per_dispatch.items():
                acc[short(names[did])][cname].append(val)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


bench = counters("pmc")
calib = counters("cal")
traffic = {}
for title, table in (("bench", bench), ("calibration (tools/calib_copy.py)", calib)):
    lines += [f"## PMC, mean per dispatch: {title}", ""]
    for k, cs in sorted(table.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0)):
        if max(cs.values(), default=0) < 1e3:
            continue
        lines.append(f"* `{k}`")
        for c, v in sorted(cs.items()):
            lines.append(f"    * {c}: {v:,.0f}")
    lines.append("")

# calibration factors: the clone kernel reads 1 GiB and writes 1 GiB
copy = next((cs for k, cs in calib.items() if "elementwise" in k.lower() or "copy" in k.lower()), None)
fetch_unit = write_unit = None
if copy and "FETCH_SIZE" in copy and "WRITE_SIZE" in copy:
    fetch_unit = (1 << 30) / copy["FETCH_SIZE"]   # true bytes per FETCH_SIZE unit on a wide stream
    write_unit = (1 << 30) / copy["WRITE_SIZE"]
    lines += [f"Calibration: 1 GiB copy -> FETCH_SIZE={copy['FETCH_SIZE']:,.0f}, WRITE_SIZE={copy['WRITE_SIZE']:,.0f} "
              f"=> {fetch_unit:.1f} B per FETCH unit, {write_unit:.1f} B per WRITE unit on coalesced streams.", ""]
for k, cs in bench.items():
    if "spmv" in k or "spmm" in k or "spg" in k:
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs and fetch_unit:
            traffic[k] = {"fetch_bytes": cs["FETCH_SIZE"] * fetch_unit, "write_bytes": cs["WRITE_SIZE"] * write_unit,
                          "hbm_bytes": cs["FETCH_SIZE"] * fetch_unit + cs["WRITE_SIZE"] * write_unit}
lines += ["## derived HBM-side traffic per launch (calibrated)", "", "```", json.dumps(traffic, indent=1), "```"]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write("\n".join(lines) + "\n")
with open(os.path.join(ROOT, "gpurun_out", f"{tag}_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print("\n".join(lines))
