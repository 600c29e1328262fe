"""research: the 256^3 bench scene over n substeps, one line per viscosity solve; verbose output of the library from substep `vfrom` on"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_workload
from flipviscosity3d_amd.capi import Context
N = int(sys.argv[1]); n = int(sys.argv[2]); vfrom = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
NU = float(os.environ.get("R5_NU", "5.0"))
kw = {}
for a in sys.argv[4:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(NU)
if kw: c.set_params(**kw)
c.particles = P
tot = 0.0
for t in range(n):
    if t == vfrom: c.set_params(verbose=1)
    st = c.substep(min(c.cfl(), 0.01)); v = st["viscosity"]; tot += st["total_ms"]
    print("substep %3d: %6.2f ms  visc its %3d corr %3d status %d corrstatus %d refinements %d defect %.2e step %.1e | pressure %d" % (t, st["total_ms"], v["iterations"], v["correction_iterations"], v["status"], v["correction_status"], v["refinements"], v["defect_residual"] / max(v["rhs_norm"], 1e-300), v["velocity_step"], st["pressure"]["iterations"]), flush=True)
print("mean %.2f ms" % (tot / n))
