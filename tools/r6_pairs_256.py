"""research (round 6): the pairs the 256^3 bench scene lists during the fall"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_workload
from flipviscosity3d_amd.capi import Context
I, J, K, dx, solid, P = build_workload("bunny", 256, on_device=True)
for prm in (dict(), dict(viscosity_pair_correction=-1), dict(viscosity_pair_lambda_floor=1.0)):
    c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = P
    sys.stderr.write("==== %s\n" % prm); sys.stderr.flush()
    for t in range(2):
        c.set_params(verbose=1, **prm)
        st = c.substep(min(c.cfl(), 0.01))
        sys.stderr.write("substep %d: %d iterations (%d correction)\n" % (t, st["viscosity"]["iterations"], st["viscosity"]["correction_iterations"])); sys.stderr.flush()
    c.close()
