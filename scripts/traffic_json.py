"""profiles/dominant_kernel_traffic.json from the PMC passes of scripts/pmc_any.sh over
scripts/micro_line_x3.py (both widths of line_x3r_kernel):
    traffic_json.py gpurun_out/pmc_x3r_64 gpurun_out/pmc_x3r_128 > profiles/dominant_kernel_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
16-byte-per-lane streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported."""
import collections
import csv
import glob
import json
import sys

GEOM = {64: (6, 128, 256, 1), 128: (6, 64, 128, 2)}
out = {"kernels": {}}
for d in sys.argv[1:]:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    name = None
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "line_x3r" in r["Kernel_Name"]:
                name = r["Kernel_Name"]
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    mean = {c: sum(v.values()) / len(v) for c, v in per.items()}
    c = 64 if "<64" in name else 128
    n, h, w, dil = GEOM[c]
    hbm = int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)
    out["kernels"][name.replace("void ", "").replace("(LineDesc)", "")] = {
        "command": f"scripts/pmc_any.sh x3r_{c} scripts/micro_line_x3.py --c {c} --h {h} --w {w} --d {dil} --reps 20 "
                   f"((3,1) convolution, dilation {dil}, bias + ReLU, {n}x{h}x{w} pixels); raw passes: "
                   f"profiles/r3_pmc/line_x3r_{c}_pass1-5.csv",
        "launches_sampled": len(per["FETCH_SIZE"]),
        "FETCH_SIZE_KB_raw": round(mean["FETCH_SIZE"], 1), "WRITE_SIZE_KB_raw": round(mean["WRITE_SIZE"], 1),
        "correction": "gfx950: FETCH_SIZE counts 128-B requests of 16-B/lane streaming reads as 64 B -> x2; WRITE_SIZE as reported",
        "hbm_bytes_per_launch": hbm,
        "algorithmic_bytes_per_launch": 2 * n * h * w * c * 4,
        "note": "reads above the input's size: the lines either side of a block's range of output lines (read by two "
                "blocks) and the weight fragments every block loads (mostly L2 hits)",
        "counters": {k: round(v, 1) for k, v in sorted(mean.items())},
    }
json.dump(out, sys.stdout, indent=1)
print()
