"""research (round 6): what the strongly coupled pairs' correction does to the stages of the default solve (verbose histories, pairs on / off)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from flipviscosity3d_amd.capi import Context
from test_oracle_compact_golden import build_host_scene
N = 64
dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
y = (np.arange(N + 1) * dx)[None, :, None]
nu0 = np.ascontiguousarray(np.broadcast_to(np.where(y < 0.42, 0.0, 200.0), (N + 1, N + 1, N + 1)), np.float32)
for label, nu, prm in (("nu 0|200", nu0, {}), ("nu 3000 two stages", 3000.0, dict(viscosity_stage2_rounds=2)), ("nu 200 fp64", 200.0, dict(precision=1)), ("nu 200", 200.0, {}), ("nu 5", 5.0, {})):
    for pc in (0, -1):
        print("==== %s, viscosity_pair_correction %d" % (label, pc), flush=True)
        sys.stderr.write("==== %s, viscosity_pair_correction %d\n" % (label, pc)); sys.stderr.flush()
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_params(verbose=2, viscosity_pair_correction=pc, **prm)
        c.particles = P
        st = c.substep(0.01)
        v = st["viscosity"]
        print("   %d iterations (%d correction, status %d), status %d, residual %.2e defect %.2e step %.1e" % (v["iterations"], v["correction_iterations"], v["correction_status"], v["status"], v["residual"], v["defect_residual"], v["velocity_step"]), flush=True)
        c.close()
