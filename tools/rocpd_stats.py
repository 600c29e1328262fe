"""per-kernel statistics from a rocprofv3 rocpd database (the default output format of this rocprofv3 when --output-format is not given):
   python tools/rocpd_stats.py results.db [name-substring ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
name = "kernel_name" if "kernel_name" in cols else ("display_name" if "display_name" in cols else cols[-1])
rows = cur.execute("select s.%s, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start), avg(d.grid_size_x*d.grid_size_y*d.grid_size_z) "
                   "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 6 desc" % (name, kd, ks, name)).fetchall()
tot = sum(r[5] for r in rows)
keys = sys.argv[2:]
print("%-90s %7s %10s %10s %10s %7s %10s" % ("kernel", "calls", "avg us", "min us", "max us", "%", "threads"))
for r in rows:
    if keys and not any(k in r[0] for k in keys):
        continue
    print("%-90s %7d %10.1f %10.1f %10.1f %7.2f %10.0f" % (r[0].replace("void ", "")[:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, 100.0 * r[5] / tot, r[6]))
print("total kernel time %.2f ms" % (tot / 1e6))
