"""default-parameter run of a bench workload with solver diagnostics: python tools/r3_scene.py workload size viscosity substeps [verbose]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd.capi import Context
from bench import build_workload
wl, N, nu, nsub = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
I, J, K, dx, solid, P = build_workload(wl, N, on_device=True)
c = Context(I, J, K, dx)
c.set_solid_sdf(solid); c.set_viscosity(nu)
c.set_params(verbose=int(sys.argv[5]) if len(sys.argv) > 5 else 1)
c.particles = P
for t in range(nsub):
    st = c.substep(min(c.cfl(), 0.01))
    v = st["viscosity"]
    print("substep %d: total %.2f ms visc %.2f ms | its %d status %d prec %d refinements %d residual/rhs %.2e" % (t, st["total_ms"], st["phase_ms"]["viscosity"], v["iterations"], v["status"], v["preconditioner"], v["refinements"], v["residual"] / max(v["rhs_norm"], 1e-300)), flush=True)
