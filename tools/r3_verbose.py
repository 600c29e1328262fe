import os, sys
sys.path.insert(0, os.getcwd())
from flipviscosity3d_amd.capi import Context
from bench import build_workload
I, J, K, dx, solid, P = build_workload("bunny", 256, on_device=True)
c = Context(I, J, K, dx)
c.set_solid_sdf(solid); c.set_viscosity(float(sys.argv[1]))
c.set_params(verbose=1)
c.particles = P
for t in range(int(sys.argv[2])):
    st = c.substep(min(c.cfl(), 0.01))
    v = st["viscosity"]
    print("substep", t, v["iterations"], v["status"], v["preconditioner"], v["residual"], flush=True)
