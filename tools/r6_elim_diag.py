"""research (round 6): does taking the repeated rows out of the system change the second correction stage at nu = 3000 (64^3 bunny from rest)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from flipviscosity3d_amd.capi import Context
from test_oracle_compact_golden import build_host_scene
N = 64
dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
for prm in (dict(), dict(viscosity_massless_polish=-1), dict(viscosity_pair_correction=-1), dict(viscosity_pair_correction=-1, viscosity_massless_polish=-1), dict(stall_guard_ratio=1000.0),
            dict(stall_guard_ratio=1000.0, viscosity_pair_correction=-1, viscosity_massless_polish=-1)):
    for rep in range(2):
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(3000.0); c.set_params(viscosity_stage2_rounds=2, **prm)
        c.particles = P
        v = c.substep(0.01)["viscosity"]
        print("%s: %d iterations (%d correction, status %d), status %d, residual %.2e defect %.2e step %.1e" % (prm, v["iterations"], v["correction_iterations"], v["correction_status"], v["status"], v["residual"], v["defect_residual"], v["velocity_step"]), flush=True)
        c.close()
