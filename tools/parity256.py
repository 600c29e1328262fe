"""Every solver variant against the two 256^3 reference dumps (tests/golden/bunny256_nu5_converged.npz: reference cap lifted, its
tolerance 1e-6; bunny256_nu5_tight.npz: 1e-8): preconditioner x vector precision x tolerance x operator (exact / the reference's
float-rounded diagonal, flipv_params.reference_diagonal).  Prints iterations, status, solve time and the velocity error at the
goldens' probe faces.   python tools/parity256.py > profiles/r2/parity256_variants.log"""
import os
import sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import Golden
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context, PRECOND_DIAGONAL, PRECOND_MULTIGRID

dx, solid, P = build_host_scene(256, ("sphere_large.ply", True), ["stanford_bunny.ply"])
variants = [  # (preconditioner, fp64 vectors, tolerance, reference_diagonal)
    (PRECOND_DIAGONAL, 0, 1e-6, 0), (PRECOND_DIAGONAL, 0, 1e-7, 0), (PRECOND_DIAGONAL, 1, 1e-7, 0), (PRECOND_MULTIGRID, 0, 1e-6, 0), (PRECOND_MULTIGRID, 0, 1e-7, 0),
    (PRECOND_DIAGONAL, 0, 1e-6, 1), (PRECOND_DIAGONAL, 1, 1e-6, 1), (PRECOND_DIAGONAL, 1, 1e-8, 1)]
for name in ("bunny256_nu5_tight", "bunny256_nu5_converged"):
    g = Golden(name)
    print("== %s: reference viscosity iterations %s" % (name, [int(g["s%d_visc_iters" % t]) for t in range(g.nsub)]), flush=True)
    for prec, fp64, tol, ref in variants:
        c = Context(256, 256, 256, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0)
        c.set_params(viscosity_preconditioner=prec, viscosity_tolerance=tol, viscosity_max_iterations=60000, precision=fp64, exact_viscosity_operator=0 if ref else 1)
        c.particles = P
        for t in range(g.nsub):
            st = c.substep(g.dt)
            num = den = 0.0
            for n in "UVW":
                a = c.grid(n).reshape(-1)
                idx, val = g["s%d_probe_idx_%s" % (t, n)], g["s%d_probe_val_%s" % (t, n)]
                num = max(num, float(np.abs(a[idx].astype(np.float64) - val).max()))
                den = max(den, float(g["s%d_maxabs_%s" % (t, n)]))
            v = st["viscosity"]
            print("%-9s %s tol %g %-18s substep %d: %5d iterations, status %d, %6.1f ms, velocity error %.3e" % (
                "multigrid" if prec == PRECOND_MULTIGRID else "diagonal", "fp64" if fp64 else "fp32", tol, "reference operator" if ref else "exact operator", t,
                v["iterations"], v["status"], st["phase_ms"]["viscosity"], num / den), flush=True)
        c.close()
