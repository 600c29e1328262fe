"""default-parameter run of the bench scene with solver diagnostics: python tools/r3_soak.py size substeps [verbose]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd.capi import Context
from bench import build_workload
N, nsub = int(sys.argv[1]), int(sys.argv[2])
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
c = Context(I, J, K, dx)
c.set_solid_sdf(solid); c.set_viscosity(5.0)
c.set_params(verbose=int(sys.argv[3]) if len(sys.argv) > 3 else 1, exact_viscosity_operator=int(sys.argv[4]) if len(sys.argv) > 4 else 0,
             viscosity_layout=int(sys.argv[5]) if len(sys.argv) > 5 else 0)
c.particles = P
t0 = time.perf_counter()
for t in range(nsub):
    st = c.substep(min(c.cfl(), 0.01))
    v = st["viscosity"]
    print("substep %d: dt %.5f total %.2f ms visc %.2f ms | its %d status %d prec %d refinements %d" % (t, st["dt"], st["total_ms"], st["phase_ms"]["viscosity"], v["iterations"], v["status"], v["preconditioner"], v["refinements"]), flush=True)
c.synchronize()
print("mean %.2f ms per substep" % ((time.perf_counter() - t0) * 1e3 / nsub))
