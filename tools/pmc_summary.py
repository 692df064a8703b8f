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

# HBM-side bytes per launch from the request-size counters (exact, no unit guessing):
#   read  = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B
#   write = 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B)
# MI355X_MICROARCH.md: FETCH_SIZE (KiB) tallies 128-B requests at 64 B, i.e. reads 1/2 of a wide
# stream -- the table above shows FETCH_SIZE*1024 == 64*RDREQ; WRITE_SIZE*1024 == the write bytes.
def req_bytes(cs):
    need = ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum",
            "TCC_EA0_WRREQ_64B_sum")
    if not all(k in cs for k in need):
        return None
    rd = 32 * cs[need[0]] + 64 * cs[need[1]] + 128 * cs[need[2]]
    wr = 64 * cs[need[4]] + 32 * (cs[need[3]] - cs[need[4]])
    return {"read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr,
            "fetch_size_kib": cs.get("FETCH_SIZE"), "write_size_kib": cs.get("WRITE_SIZE"),
            "l2_hit_rate": cs["TCC_HIT_sum"] / max(1.0, cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"]) if "TCC_HIT_sum" in cs else None}


for k, cs in bench.items():
    if "spb::" in k:
        t = req_bytes(cs)
        if t:
            traffic[k.replace("void ", "").strip()] = t
lines += ["## derived HBM-side traffic per launch (request-size counters)", "", "```", json.dumps(traffic, indent=1), "```"]
cal = {k: req_bytes(cs) for k, cs in calib.items() if req_bytes(cs) and req_bytes(cs)["hbm_bytes"] > 1e8}
lines += ["", "## calibration kernels (known byte counts: 1 GiB clone = 1.07e9 B read + 1.07e9 B written; "
          "gather = 8e8 B index read + 4e8 B written + gathered lines)", "", "```", json.dumps(cal, indent=1), "```"]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
with open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w") as f:
    f.write("\n".join(lines) + "\n")
with open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print("\n".join(lines[-40:]))
