"""Where do the 3e-4 of the rod + sheet scene at nu dt/dx^2 = 1.3e5 come from?  The viscosity solve alone, GPU against the oracle on IDENTICAL inputs
(the GPU's own phi and velocities after the body force), with the control volumes compared lattice by lattice."""
import os
import sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import Golden
from test_oracle_compact_golden import build_host_scene, STIFF
from flipviscosity3d_amd import capi
from oracle import oraclebind as O

name = sys.argv[1] if len(sys.argv) > 1 else "honey96_nu1422"
name, N, boundary, liquids = [s for s in STIFF if s[0] == name][0]
g = Golden(name)
dx, solid, P = build_host_scene(N, boundary, liquids)
nu = float(g["nu"])
visc = np.full((N + 1, N + 1, N + 1), nu, np.float32)


def prep(c):
    c.set_solid_sdf(solid); c.set_viscosity(nu); c.particles = P
    c.particle_sdf(); c.advect_velocity_field(); c.body_force(g.dt)
    return c.grid("LIQUID_PHI"), [c.grid(n) for n in "UVW"]


c = capi.Context(N, N, N, dx)
phi, uvw = prep(c)
(Uo, Vo, Wo), info = O.viscosity_solve(N, N, N, dx, g.dt, uvw[0], uvw[1], uvw[2], phi, solid, visc, maxiter=400000)
print("oracle viscosity solve: %d iterations, status %d" % (info["iterations"], info["status"]))
ref = [Uo, Vo, Wo]
den = max(np.abs(a).max() for a in ref)
vo = O.viscosity_volumes(N, N, N, dx, phi)


def report(tag, c):
    got = [c.grid(n) for n in "UVW"]
    errs = [np.abs(a.astype(np.float64) - b).max() / den for a, b in zip(got, ref)]
    m = int(np.argmax(errs))
    idx = np.unravel_index(np.argmax(np.abs(got[m].astype(np.float64) - ref[m])), ref[m].shape)
    print("   %-60s error %.2e (U %.1e V %.1e W %.1e), worst face %s%s: gpu %.6f oracle %.6f" % (tag, max(errs), errs[0], errs[1], errs[2], "UVW"[m], idx[::-1], got[m][idx], ref[m][idx]))
    return got


st = c.viscosity_solve(g.dt)
print("   default:", {k: st[k] for k in ("iterations", "correction_iterations", "status", "correction_status", "rows")})
report("default", c)
for nm in ["center", "U", "V", "W", "edgeU", "edgeV", "edgeW"]:
    a, b = c.viscosity_volume(nm), vo[nm]
    d = np.abs(a - b)
    w = np.unravel_index(np.argmax(d), d.shape)
    print("   volume %-6s max abs diff %.2e at %s (gpu %.6f oracle %.6f); entries differing: %d of %d non-trivial" % (nm, d.max(), w[::-1], a[w], b[w], int((a != b).sum()), int(((b > 0) & (b < 1)).sum())))
c.close()
for tag, prm in [("3 rounds", dict(viscosity_stage2_rounds=3)),
                 ("diagonal, cap lifted, tol 1e-7", dict(viscosity_preconditioner=capi.PRECOND_DIAGONAL, viscosity_max_iterations=400000, viscosity_tolerance=1e-7)),
                 ("fp64 vectors, diagonal, cap lifted, tol 1e-8", dict(precision=1, viscosity_max_iterations=400000, viscosity_tolerance=1e-8)),
                 ("exact operator, fp64, diagonal, tol 1e-8", dict(precision=1, exact_viscosity_operator=1, viscosity_max_iterations=400000, viscosity_tolerance=1e-8))]:
    c = capi.Context(N, N, N, dx)
    c.set_params(**prm)
    prep(c)
    st = c.viscosity_solve(g.dt)
    print("   %s:" % tag, {k: st[k] for k in ("iterations", "status", "residual", "rhs_norm")})
    report(tag, c)
    c.close()
