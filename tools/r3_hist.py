"""residual histories of the viscosity solves (flipv_params.verbose = 2):
   python tools/r3_hist.py <fixture | bunnyN> layout precond replacement [exact] [substeps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd import capi
from flipviscosity3d_amd.capi import Context
name, lay, pre, rep = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
exact = int(sys.argv[5]) if len(sys.argv) > 5 else 0
nsub = int(sys.argv[6]) if len(sys.argv) > 6 else 1
if name.startswith("bunny") and name[5:].isdigit():
    from bench import build_workload
    I, J, K, dx, solid, P = build_workload("bunny", int(name[5:]), on_device=True)
    c = Context(I, J, K, dx)
    c.set_solid_sdf(solid); c.set_viscosity(5.0)
    dt = 0.01
else:
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    I, J, K = int(z["I"]), int(z["J"]), int(z["K"])
    c = Context(I, J, K, float(z["dx"]))
    c.set_solid_sdf(z["solid"]); c.set_viscosity(z["viscosity"]); c.set_gravity(*[float(v) for v in z["gravity"]])
    P, dt = z["particles0"], float(z["dt"])
c.set_params(viscosity_layout=lay, viscosity_preconditioner=pre, verbose=2, exact_viscosity_operator=exact)
c.particles = P
for t in range(nsub):
    st = c.substep(min(c.cfl(), dt))
    print(st["viscosity"])
