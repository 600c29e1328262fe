"""repro of test_liquid_box_restriction_over_a_long_run with solver diagnostics"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from flipviscosity3d_amd import capi, hostapi as H
from helpers import rel_maxnorm3
mesh = os.path.join(ROOT, "tests", "golden", "meshes")
N = 64
dx = float(np.float32(1.0 / N))
s = H.FluidSimulation(); s.initialize(N, N, N, dx)
s.addBoundary(H.load_ply(os.path.join(mesh, "sphere_large.ply")), True)
s.setSeeding(H.FluidSimulation.SEED_COUNTER, 2)
s.addLiquid(H.load_ply(os.path.join(mesh, "stanford_bunny.ply")))
solid, P = s.solid_sdf(), s.particles
s.close()
P[:, 3:] = np.array([0.9, -2.5, 0.6], np.float32)
lay = int(sys.argv[1]) if len(sys.argv) > 1 else 0
a = capi.Context(N, N, N, dx); b = capi.Context(N, N, N, dx)
b.set_params(no_liquid_box=1)
for c in (a, b):
    c.set_solid_sdf(solid); c.set_viscosity(0.5)
    c.set_params(viscosity_max_iterations=5000, viscosity_tolerance=1e-7, pressure_rel_tolerance=1e-7, viscosity_preconditioner=capi.PRECOND_DIAGONAL, viscosity_layout=lay, verbose=1)
a.particles = P
for t in range(6):
    b.particles = a.particles
    for n in "UVW":
        b.set_grid(n, a.grid(n))
    dt = min(a.cfl(), 0.01)
    sa, sb = a.substep(dt), b.substep(dt)
    print(t, "visc a", {k: sa["viscosity"][k] for k in ("iterations", "status", "refinements", "rows", "active_tiles")}, "b", {k: sb["viscosity"][k] for k in ("iterations", "status", "refinements", "rows", "active_tiles")},
          "vel diff %.3e" % rel_maxnorm3([a.grid(n) for n in "UVW"], [b.grid(n) for n in "UVW"]), flush=True)
