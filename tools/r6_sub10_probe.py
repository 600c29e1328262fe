"""research (round 6): the mid-fall state of the headline scene (the compiled reference's particles after 10 of its substeps: tests/golden/_big/bunny256_nu5_sub10_state.npy) --
the GPU default against the GPU's own tightened solves (fp64 vectors, 1e-9; the exact-residual criterion), while the compiled reference's converged answer from that state is
still being computed on the CPU (three hours of one core at 1e-10 and counting).  NOT a parity claim (the tight run is the library's own): a screen."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context
name = sys.argv[1] if len(sys.argv) > 1 else "bunny256_nu5_sub10"
nu = 200.0 if "nu200" in name else 5.0
S = np.load(os.path.join(ROOT, "tests", "golden", "_big", name + "_state.npy"))
N = 256
dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])


def run(**prm):
    c = Context(N, N, N, dx); c.set_solid_sdf(solid); c.set_viscosity(nu)
    if prm: c.set_params(**prm)
    c.particles = S
    st = c.substep(0.01)
    out = [c.grid(n).astype(np.float64) for n in "UVW"]
    c.close()
    return out, st["viscosity"]


V = {}
for label, prm in (("default", {}), ("default again", {}), ("strict (stage 1 to 1e-6)", dict(viscosity_stage1_factor=1.0)),
                   ("two correction stages", dict(viscosity_stage2_rounds=2)),
                   ("tight: fp64 vectors, 1e-9, cap 5000", dict(precision=1, viscosity_tolerance=1e-9, pressure_rel_tolerance=1e-9, viscosity_max_iterations=5000)),
                   ("tight: fp64 vectors, 1e-10, cap 20000, diagonal", dict(precision=1, viscosity_tolerance=1e-10, pressure_rel_tolerance=1e-9, viscosity_max_iterations=20000, viscosity_preconditioner=1)),
                   ("round 4's rule", dict(viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1, viscosity_pair_correction=-1))):
    V[label] = run(**prm)
    v = V[label][1]
    print("%-50s %4d iterations (%d correction), status %d, residual %.2e, defect %.2e, step %.1e" % (label, v["iterations"], v["correction_iterations"], v["status"], v["residual"], v["defect_residual"], v["velocity_step"]), flush=True)
for ref in ("tight: fp64 vectors, 1e-9, cap 5000", "tight: fp64 vectors, 1e-10, cap 20000, diagonal"):
    den = max(np.abs(a).max() for a in V[ref][0])
    print("against '%s' (max|u| %.3f):" % (ref, den))
    for label in V:
        if label == ref: continue
        e = max(np.abs(a - b).max() for a, b in zip(V[label][0], V[ref][0])) / den
        nb = sum(int((np.abs(a - b) / den > 1e-4).sum()) for a, b in zip(V[label][0], V[ref][0]))
        print("   %-50s %.2e (%d faces beyond 1e-4)" % (label, e, nb))
