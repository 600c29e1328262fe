"""memset / copy kernels of the last complete substep of a rocprofv3 kernel trace, by duration: python tools/trace_fills.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)) for r in rows), key=lambda e: e[0])
marks = [i for i, e in enumerate(ev) if e[2].startswith("k_bin_count")]
sub = ev[marks[-2]:marks[-1]]
for key in ("fillBuffer", "copyBuffer"):
    sel = [(e[1] - e[0], e[3], i) for i, e in enumerate(sub) if key in e[2]]
    print("%s: %d calls, %.1f us in total" % (key, len(sel), sum(s[0] for s in sel) / 1e3))
    for d, g, i in sorted(sel, reverse=True)[:14]:
        prev = next((sub[j][2] for j in range(i - 1, -1, -1) if "Buffer" not in sub[j][2]), "?")
        nxt = next((sub[j][2] for j in range(i + 1, len(sub)) if "Buffer" not in sub[j][2]), "?")
        print("   %7.1f us  grid %9d   after %-40.40s before %-40.40s" % (d / 1e3, g, prev.replace("void ", "").replace("(anonymous namespace)::", ""), nxt.replace("void ", "").replace("(anonymous namespace)::", "")))
