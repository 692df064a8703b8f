"""Print the backend kernels of a rocprofv3 --stats csv directory: kstat_csv.py <dir> [min_total_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
for r in csv.DictReader(open(f[0])):
    n = r["Name"]
    if "spb::" in n and float(r["TotalDurationNs"]) > lim * 1e3:
        print(f"{n[:64]:64s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us")
