"""Round-3 bring-up probe: the viscosity solve in every layout x preconditioner against the oracle on the small fixtures, then the
64^3 bunny scene's per-solve cost in each.  python tools/r3_layouts.py [size]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd import capi  # noqa: E402
from flipviscosity3d_amd.capi import Context  # noqa: E402
from oracle import oraclebind as O  # noqa: E402

LAY = {"plane": capi.LAYOUT_SWIZZLED, "brick": capi.LAYOUT_BRICK}
PRE = {"diag": capi.PRECOND_DIAGONAL, "mg": capi.PRECOND_MULTIGRID}


def fixture(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    I, J, K = int(z["I"]), int(z["J"]), int(z["K"])
    dx, dt = float(z["dx"]), float(z["dt"])
    g = [float(v) for v in z["gravity"]]
    s = O.OracleSim(I, J, K, dx)
    s.set_solid(z["solid"]); s.set_viscosity(z["viscosity"]); s.set_gravity(*g)
    s.particles = z["particles0"]
    ref = []
    for t in range(2):
        s.substep(dt)
        ref.append([s.grid(n).copy() for n in "UVW"])
    s.close()
    for ln, lv in LAY.items():
        for pn, pv in PRE.items():
            for rep in (0, -1):
                c = Context(I, J, K, dx)
                c.set_solid_sdf(z["solid"]); c.set_viscosity(z["viscosity"]); c.set_gravity(*g)
                c.set_params(viscosity_layout=lv, viscosity_preconditioner=pv, exact_viscosity_operator=1, check_every=4)
                c.particles = z["particles0"]
                errs = []
                for t in range(2):
                    st = c.substep(dt)
                    num = max(np.abs(c.grid(n).astype(np.float64) - ref[t][q]).max() for q, n in enumerate("UVW"))
                    den = max(np.abs(ref[t][q]).max() for q in range(3))
                    errs.append(num / den)
                v = st["viscosity"]
                print("%-18s %-6s %-5s repl %2d: err %.2e %.2e | its %4d status %d layout %d prec %d tiles %d" %
                      (name, ln, pn, rep, errs[0], errs[1], v["iterations"], v["status"], v["layout"], v["preconditioner"], v["active_tiles"]), flush=True)
                c.close()


def bunny(N):
    from bench import build_workload
    I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
    for ln, lv in LAY.items():
        for pn, pv in PRE.items():
            for exact in ((0, 1) if N <= 64 else (0,)):
                c = Context(I, J, K, dx)
                c.set_solid_sdf(solid); c.set_viscosity(5.0)
                c.set_params(viscosity_layout=lv, viscosity_preconditioner=pv, exact_viscosity_operator=exact, viscosity_max_iterations=20000)
                c.particles = P
                out = []
                for t in range(4):
                    st = c.substep(min(c.cfl(), 0.01))
                    v = st["viscosity"]
                    out.append("%d its %.2f ms st %d" % (v["iterations"], st["phase_ms"]["viscosity"], v["status"]))
                ms, cells = c.bench_spmv(1, 200)
                print("bunny%d %-6s %-5s exact %d: %s | spmv %.1f us, rows %d, tiles %d" % (N, ln, pn, exact, " ; ".join(out), ms * 1e3, v["rows"], v["active_tiles"]), flush=True)
                c.close()


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    if N <= 64:
        for f in ("twobody20_varvisc", "bunny32_viscous"):
            fixture(f)
    bunny(N)
