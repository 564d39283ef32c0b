"""Side by side per-kernel totals of two rocprofv3 kernel-stats CSVs: cmp_stats.py a.csv b.csv [min_us]"""
import csv, sys
a, b = ({r["Name"]: r for r in csv.DictReader(open(f))} for f in sys.argv[1:3])
ta = sum(float(r["TotalDurationNs"]) for r in a.values()); tb = sum(float(r["TotalDurationNs"]) for r in b.values())
print(f"total {ta/1e6:.3f} ms -> {tb/1e6:.3f} ms")
rows = []
for n in set(a) | set(b):
    x = float(a[n]["TotalDurationNs"]) if n in a else 0.0
    y = float(b[n]["TotalDurationNs"]) if n in b else 0.0
    rows.append((y - x, n, x, y, int(a[n]["Calls"]) if n in a else 0, int(b[n]["Calls"]) if n in b else 0))
for dlt, n, x, y, ca, cb in sorted(rows, key=lambda r: -abs(r[0]))[:14]:
    print(f"{n[:70]:70s} {ca:5d} {x/1e6:8.3f} ms -> {cb:5d} {y/1e6:8.3f} ms  ({dlt/1e6:+.3f})")
