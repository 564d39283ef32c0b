"""Per-kernel averages of the counters of scripts/pmc_any.sh passes: pmc_summary.py <dir> <kernel substring>"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    byc = collections.defaultdict(list)
    for (disp, c), v in per.items():
        byc[c].append(v)
    for c, v in byc.items():
        acc[c] = v
for c in sorted(acc):
    v = acc[c]
    print(f"{c:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
