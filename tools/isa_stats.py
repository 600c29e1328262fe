"""instruction mix of one kernel in a hipcc -S listing: python tools/isa_stats.py file.s <mangled-name-substring>"""
import collections
import re
import sys
lines = open(sys.argv[1]).read().split("\n")
start = [n for n, l in enumerate(lines) if re.match(r"^_Z.*" + re.escape(sys.argv[2]) + r".*:", l)][0]
ins = []
for l in lines[start + 1:]:
    if l.startswith(".Lfunc_end"):
        break
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith((".", ";")):
        ins.append(t.split()[0])
c = collections.Counter(ins)
g = collections.Counter()
for k, v in c.items():
    if "f64" in k: g["fp64"] += v
    elif "load" in k: g["load"] += v
    elif "store" in k: g["store"] += v
    elif k.startswith("ds_") or "dpp" in k or "permute" in k or "readlane" in k: g["cross-lane/lds"] += v
    elif k.startswith("s_"): g["salu"] += v
    else: g["valu"] += v
print("instructions", len(ins), dict(g))
print(c.most_common(30))
