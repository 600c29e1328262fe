"""research: which solve the default's distance to the tightened run comes from (256^3 bench scene, late states)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_workload
from flipviscosity3d_amd.capi import Context
N = int(sys.argv[1]); at = [int(x) for x in sys.argv[2].split(",")]
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = P
states = {}
for t in range(max(at) + 1):
    if t in at: states[t] = c.particles.copy()
    c.substep(0.01)
c.close()
TV = dict(precision=1, viscosity_tolerance=1e-9, viscosity_max_iterations=5000, viscosity_velocity_tolerance=1e-6)
VAR = [("tight both", dict(TV, pressure_rel_tolerance=1e-9)), ("default", {}), ("tight pressure only", dict(pressure_rel_tolerance=1e-9)), ("tight viscosity only", dict(TV)),
       ("pressure 1e-7", dict(pressure_rel_tolerance=1e-7)), ("pressure 1e-8", dict(pressure_rel_tolerance=1e-8)),
       ("vtol 1e-7 off", dict(viscosity_tolerance=1e-7, viscosity_velocity_tolerance=-1.0)), ("vtol 1e-7 off + pressure 1e-8", dict(viscosity_tolerance=1e-7, viscosity_velocity_tolerance=-1.0, pressure_rel_tolerance=1e-8)),
       ("default + pressure 1e-8", dict(pressure_rel_tolerance=1e-8)), ("crit 1e-4 + pressure 1e-8", dict(pressure_rel_tolerance=1e-8, viscosity_velocity_tolerance=1e-4))]
for t in at:
    ref = None
    for name, kw in VAR:
        c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0)
        if kw: c.set_params(**kw)
        c.particles = states[t]
        c.particle_sdf(); c.advect_velocity_field(); c.body_force(0.01)
        v = c.viscosity_solve(0.01); gv = [c.grid(n) for n in "UVW"]
        c.compute_weights(); p = c.pressure_solve(0.01); c.apply_pressure(0.01); c.extrapolate(); c.constrain()
        g = [c.grid(n) for n in "UVW"]
        c.close()
        if ref is None:
            ref, refv = g, gv; den = max(np.abs(a).max() for a in ref)
            print("after %d substeps, max|u| %.3f; tight: viscosity %d its, pressure %d its" % (t, den, v["iterations"], p["iterations"]), flush=True); continue
        def err(A, B):
            e = [np.abs(a.astype(np.float64) - b) / den for a, b in zip(A, B)]
            return max(x.max() for x in e), sum(int((x > 1e-4).sum()) for x in e)
        ev, nv = err(gv, refv); ef, nf = err(g, ref)
        print("   %-32s viscosity its %3d pressure its %3d: after viscosity %.2e (%d faces > 1e-4)   end of substep %.2e (%d)" % (name, v["iterations"], p["iterations"], ev, nv, ef, nf), flush=True)
