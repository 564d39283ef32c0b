"""GPU idle time inside the timed steps of a rocprofv3 kernel trace: usage gap_report.py <kernel_trace.csv> [marker]
Steps are delimited by the first kernel of each forward (marker, default first_down_fwd_kernel)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "first_down_fwd_kernel"
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
starts = [i for i, e in enumerate(ev) if marker in e[2]]
print("steps found:", len(starts))
for a, b in list(zip(starts, starts[1:]))[-6:]:
    sel = ev[a:b]
    span = ev[b][0] - sel[0][0]
    busy = 0; cs, ce = sel[0][0], sel[0][1]; gaps = []
    for s, e, n in sel[1:] + [(ev[b][0], ev[b][0], "next step")]:
        if s > ce:
            busy += ce - cs; gaps.append(((s - ce) / 1e3, n[:40])); cs, ce = s, e
        else:
            ce = max(ce, e)
    gaps.sort(reverse=True)
    print(f"step {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {100*(1-busy/span):.1f} %, kernels {len(sel)}; top gaps us: " +
          ", ".join(f"{g:.0f}@{n}" for g, n in gaps[:5]))
