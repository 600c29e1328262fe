#!/bin/bash
# round 6: A/B of library builds (tools/ab_lib.sh build ...) on the bench line: headline, sparse-scene SpMV, filled-box SpMVs.   bash tools/r6_ab_bench.sh name1 name2 ...
cs=flipviscosity3d_amd/csrc
mkdir -p gpurun_out/r6
for name in "$@"; do
  FLIPV_LIB=$PWD/$cs/build/variants/$name.so python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict --dense-size ${DENSE:-512} > gpurun_out/r6/ab_$name.json 2> gpurun_out/r6/ab_$name.err
  python3 - $name <<'P'
import json, sys
n = sys.argv[1]
d = json.loads(open("gpurun_out/r6/ab_%s.json" % n).read().strip().splitlines()[-1])
def fr(k):
    r = d.get(k) or {}
    return " ".join("%s %.3f" % (kk.replace("viscosity_spmv", "v").replace("pressure_spmv", "p").replace("_multigrid_loop", "mg"), vv["frac"]) for kk, vv in r.items() if isinstance(vv, dict) and "frac" in vv)
print("%-10s %.1f MCells/s %.2f ms | visc %.2f project %.2f | its %.1f | sparse spmv %.2f us frac %.3f | dense256: %s | dense512: %s" % (
    n, d["value"], d["ms_per_step"], d["phase_ms"]["viscosity"], d["phase_ms"]["project"], d["viscosity_iterations"]["mean"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], fr("roofline_dense"), fr("roofline_dense_512")))
P
done
