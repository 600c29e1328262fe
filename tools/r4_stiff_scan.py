"""Scan of the two-stage viscosity solve's constants (flipv_params.viscosity_stage*) against the goldens cut at BASELINE config 4's stiffness
(tests/golden/bunny64_nu3000, honey96_nu1422: nu dt/dx^2 = 1.2e5 ... 1.3e5; the reference with its cap lifted).  Every substep is UNCHAINED:
substep t starts from the reference's own particles.  Output: velocity error at the fixture's probe faces (relative max-norm), iterations
(of which in correction stages), correction status, delivered fp64 residual on the reference's operator.

    python tools/r4_stiff_scan.py [fixture ...] > profiles/r4/stiff_regime_scan.log"""
import os
import sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import Golden
from test_oracle_compact_golden import build_host_scene, STIFF
from flipviscosity3d_amd.capi import Context

COMBOS = [dict(),   # the library's defaults
          dict(viscosity_stage2_factor=1e-2, viscosity_stage2_max_iterations=48),   # round 3's rule
          dict(viscosity_stage2_factor=1e-2), dict(viscosity_stage2_factor=3e-3), dict(viscosity_stage2_factor=1e-3),
          dict(viscosity_stage2_rounds=2, viscosity_stage2_factor=1e-2), dict(viscosity_stage1_factor=1.0), dict(exact_viscosity_operator=1)]

names = sys.argv[1:] or [s[0] for s in STIFF]
for name, N, boundary, liquids in STIFF:
    if name not in names:
        continue
    g = Golden(name)
    dx, solid, P = build_host_scene(N, boundary, liquids)
    nu = float(g["nu"])
    print("== %s: %d^3, nu %g, nu dt/dx^2 = %.0f, reference iterations %d / %d" % (name, N, nu, nu * g.dt / g.dx ** 2, int(g["s0_visc_iters"]), int(g["s1_visc_iters"])))
    for prm in COMBOS:
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid)
        c.set_viscosity(nu)
        if prm:
            c.set_params(**prm)
        for t in range(g.nsub):
            c.particles = P if t == 0 else g["s%d_particles" % (t - 1)]
            st = c.substep(g.dt)
            v = st["viscosity"]
            num = den = 0.0
            for n in "UVW":
                a = c.grid(n).reshape(-1)
                num = max(num, float(np.abs(a[g["s%d_probe_idx_%s" % (t, n)]].astype(np.float64) - g["s%d_probe_val_%s" % (t, n)]).max()))
                den = max(den, float(g["s%d_maxabs_%s" % (t, n)]))
            print("   %-90s substep %d: error %.2e  iterations %3d (%3d in corrections, status %d / %d)  residual %.1e  defect %.1e  %.1f ms" % (
                prm if prm else "DEFAULT", t, num / den, v["iterations"], v["correction_iterations"], v["status"], v["correction_status"],
                v["residual"] / v["rhs_norm"], v["defect_residual"] / v["rhs_norm"], st["phase_ms"]["viscosity"]), flush=True)
        c.close()
