"""Condense the output of x3_stamps: per configuration the prologue, the mean tap / barrier / epilogue
parts and the total, in ticks."""
import re, sys
cfg = None
rows = {}
for line in sys.stdin:
    if line.startswith("NC"):
        cfg = line.split(";")[0]; rows[cfg] = []
    m = re.match(r"\s+stamp\s+(\d+): median\s+(\d+) \(\+\s*(-?\d+)\)", line)
    if m and cfg: rows[cfg].append((int(m.group(1)), int(m.group(2)), int(m.group(3))))
for cfg, r in rows.items():
    d = [x[2] for x in r]
    pro = d[0] + d[1]
    body = d[2:]
    print(f"{cfg}: prologue {pro}; total {r[-1][1]}; deltas after prologue: {' '.join(str(x) for x in body)}")
