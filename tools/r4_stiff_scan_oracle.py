"""How many correction stages does the two-stage viscosity solve need BETWEEN the stiffness of the headline (3e3) and config 4's (1.3e5)?
Config 1's scene at 64^3 with nu = 200 ... 2000, two UNCHAINED substeps (both sides start every substep from the oracle's particles), GPU against
the oracle run to convergence (its cap lifted).     python tools/r4_stiff_scan_oracle.py > profiles/r4/stiff_regime_scan_oracle.log"""
import os
import sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context
from oracle import oraclebind as O

N = 64
dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
for nu in [float(v) for v in (sys.argv[1:] or ["200", "500", "800", "1280", "2000"])]:
    o = O.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=400000)
    o.particles = P
    starts, refs, oits = [], [], []
    for t in range(2):
        starts.append(o.particles.copy())
        sec, vi, pi = o.substep(0.01)
        refs.append([o.grid(n).astype(np.float64) for n in "UVW"])
        oits.append(vi["iterations"])
    o.close()
    print("== 64^3 nu %g (nu dt/dx^2 = %.0f), oracle iterations %s" % (nu, nu * 0.01 / dx ** 2, oits), flush=True)
    for prm in [dict(), dict(viscosity_stage1_factor=1000.0), dict(viscosity_stage1_factor=3000.0), dict(viscosity_stage1_factor=10000.0)]:
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_params(**prm)
        for t in range(2):
            c.particles = starts[t]
            st = c.substep(0.01)
            v = st["viscosity"]
            num = max(np.abs(c.grid(n).astype(np.float64) - r).max() for n, r in zip("UVW", refs[t]))
            den = max(np.abs(r).max() for r in refs[t])
            print("   %-70s substep %d: error %.2e  iterations %3d (%3d in corrections, status %d / %d)  %.1f ms" % (
                prm, t, num / den, v["iterations"], v["correction_iterations"], v["status"], v["correction_status"], st["phase_ms"]["viscosity"]), flush=True)
        c.close()
