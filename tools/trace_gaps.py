"""Where a substep's wall time goes that is not kernel time: python tools/trace_gaps.py <kernel_trace.csv> [min gap us]
Takes the last complete substep of the trace (from one k_bin_count launch to the next) and lists the idle gaps between consecutive kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if e[2].startswith("k_bin_count")]
a, b = marks[-2], marks[-1]
sub = ev[a:b]
span = (sub[-1][1] - sub[0][0]) / 1e3
busy = sum(e[1] - e[0] for e in sub) / 1e3
print("substep: %d kernels, span %.1f us, kernel time %.1f us, idle %.1f us" % (len(sub), span, busy, span - busy))
gaps = []
for p, q in zip(sub[:-1], sub[1:]):
    g = (q[0] - p[1]) / 1e3
    if g >= mingap:
        gaps.append((g, p[2][:60], q[2][:60], (p[1] - sub[0][0]) / 1e3))
print("gaps >= %.0f us: %d, total %.1f us" % (mingap, len(gaps), sum(g[0] for g in gaps)))
for g in sorted(gaps, key=lambda x: x[3]):
    print("  at %8.1f us: %7.1f us idle between %s -> %s" % (g[3], g[0], g[1], g[2]))
small = sum((q[0] - p[1]) / 1e3 for p, q in zip(sub[:-1], sub[1:]) if 0 < (q[0] - p[1]) / 1e3 < mingap)
print("gaps below that: %.1f us in total over %d boundaries" % (small, len(sub) - 1 - len(gaps)))
