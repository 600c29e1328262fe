"""The reference itself (oracle/_ref) timed on the HEADLINE configuration -- 256^3 bunny drop, viscosity 5, one thread --
through bench.py's own cpu_baseline leg; writes profiles/r2/cpu_baseline_256.json (about six minutes of CPU).
    python tools/cpu_baseline_fullsize.py [size]"""
import json
import os
import platform
import sys
sys.path.insert(0, os.getcwd())
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = bench.cpu_baseline(5.0, N)
cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
out["host"] = {"cpu_model": cpu[0] if cpu else platform.processor(), "logical_cpus": os.cpu_count()}
path = os.path.join("profiles", "r2", "cpu_baseline_%d.json" % N)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
