"""Stage 1 of the two-stage viscosity solve: how early may it stop?  flipv_params.viscosity_stage1_factor in {300 (the default), 1000, 3000} on the converged goldens --
the 128^3 and 256^3 bunny (two CHAINED substeps, as the parity tests run them), the two stiff fixtures (unchained) -- and the bench's 20 substeps.
    python tools/r4_stage1_scan.py > profiles/r4/stage1_factor_scan.log"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import Golden
from test_oracle_compact_golden import build_host_scene, STIFF
from flipviscosity3d_amd.capi import Context


def probe_err(c, g, t):
    num = den = 0.0
    for n in "UVW":
        a = c.grid(n).reshape(-1)
        num = max(num, float(np.abs(a[g["s%d_probe_idx_%s" % (t, n)]].astype(np.float64) - g["s%d_probe_val_%s" % (t, n)]).max()))
        den = max(den, float(g["s%d_maxabs_%s" % (t, n)]))
    return num / den


factors = [300.0, 1000.0, 3000.0]
for name, N in (("bunny128_nu5_converged", 128), ("bunny256_nu5_converged", 256)):
    g = Golden(name)
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    for f in factors:
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"])); c.set_params(viscosity_stage1_factor=f); c.particles = P
        out = []
        for t in range(g.nsub):
            st = c.substep(g.dt)
            out.append("%.2e (%d its, status %d)" % (probe_err(c, g, t), st["viscosity"]["iterations"], st["viscosity"]["status"]))
        print("%-26s stage1_factor %6g: %s" % (name, f, "  ".join(out)), flush=True)
        c.close()
    if N == 256:
        for f in factors:
            c = Context(N, N, N, dx)
            c.set_solid_sdf(solid); c.set_viscosity(5.0); c.set_params(viscosity_stage1_factor=f); c.particles = P
            for _ in range(5):
                c.substep(min(c.cfl(), 0.01))
            c.synchronize()
            t0 = time.perf_counter()
            sts = [c.substep(min(c.cfl(), 0.01)) for _ in range(20)]
            c.synchronize()
            el = time.perf_counter() - t0
            print("bench window (5 + 20 substeps) stage1_factor %6g: %.1f MCells/s, %.2f ms per substep, mean %.1f iterations, statuses %s" % (
                f, N ** 3 / 1e6 / (el / 20), el * 50, np.mean([s["viscosity"]["iterations"] for s in sts]), sorted(set(s["viscosity"]["status"] for s in sts))), flush=True)
            c.close()
for name, N, boundary, liquids in STIFF:
    g = Golden(name)
    dx, solid, P = build_host_scene(N, boundary, liquids)
    for f in factors:
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"])); c.set_params(viscosity_stage1_factor=f)
        out = []
        for t in range(g.nsub):
            c.particles = P if t == 0 else g["s%d_particles" % (t - 1)]
            st = c.substep(g.dt)
            out.append("%.2e (%d its, status %d)" % (probe_err(c, g, t), st["viscosity"]["iterations"], st["viscosity"]["status"]))
        print("%-26s stage1_factor %6g: %s" % (name, f, "  ".join(out)), flush=True)
        c.close()
