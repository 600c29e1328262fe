"""research (round 6): the 256^3 scene at nu = 200 (nu dt/dx^2 = 131 072) 25 substeps in -- tests/golden/bunny256_nu200_sub25_tol8: the compiled reference from its own state, converged to
1e-8 in 15 148 iterations (at its defaults it stops at the cap of 700, 0.82 max|u| from that).  What does the GPU need to get there?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context
import flipviscosity3d_amd.capi as capi
name = "bunny256_nu200_sub25"
g = np.load(os.path.join(ROOT, "tests", "golden", name + "_tol8.npz"))
S = np.load(os.path.join(ROOT, "tests", "golden", "_big", name + "_state.npy"))
N = 256
dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
den = float(g["maxabs"])


def run(label, **prm):
    c = Context(N, N, N, dx); c.set_solid_sdf(solid); c.set_viscosity(200.0)
    if prm: c.set_params(**prm)
    c.particles = S
    st = c.substep(0.01)
    worst, n4 = 0.0, 0
    for cc in "UVW":
        e = np.abs(c.grid(cc).reshape(-1)[g["probe_idx_" + cc]].astype(np.float64) - g["probe_val_" + cc]) / den
        worst = max(worst, float(e.max())); n4 += int((e > 1e-4).sum())
    c.close()
    v = st["viscosity"]
    print("%-64s %.2e (%7d probes beyond 1e-4) | %5d iterations (%d correction, status %d), status %d, prec %d, residual %.2e of rhs %.2e, defect %.2e, step %.1e, %.0f ms" % (
        label, worst, n4, v["iterations"], v["correction_iterations"], v["correction_status"], v["status"], v["preconditioner"], v["residual"], v["rhs_norm"], v["defect_residual"], v["velocity_step"], st["phase_ms"]["viscosity"]), flush=True)


run("default")
run("cap 3000", viscosity_max_iterations=3000)
run("cap 20000", viscosity_max_iterations=20000)
run("cap 20000, two correction rounds", viscosity_max_iterations=20000, viscosity_stage2_rounds=2)
run("cap 20000, stage 1 to 1e-6", viscosity_max_iterations=20000, viscosity_stage1_factor=1.0)
run("cap 20000, fp64 vectors, 1e-8", viscosity_max_iterations=20000, precision=1, viscosity_tolerance=1e-8)
run("cap 20000, diagonal, fp64 vectors, 1e-8", viscosity_max_iterations=20000, precision=1, viscosity_tolerance=1e-8, viscosity_preconditioner=capi.PRECOND_DIAGONAL if hasattr(capi, "PRECOND_DIAGONAL") else 1)
