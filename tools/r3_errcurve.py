"""velocity error against the oracle as a function of the viscosity iteration cap (multigrid, twobody fixture): which residual norm
tracks the error?  python tools/r3_errcurve.py [fixture] [precond]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd import capi
from flipviscosity3d_amd.capi import Context
from oracle import oraclebind as O
name = sys.argv[1] if len(sys.argv) > 1 else "twobody20_varvisc"
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 2
z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
I, J, K = int(z["I"]), int(z["J"]), int(z["K"])
dx, dt = float(z["dx"]), float(z["dt"])
g = [float(v) for v in z["gravity"]]
s = O.OracleSim(I, J, K, dx)
s.set_solid(z["solid"]); s.set_viscosity(z["viscosity"]); s.set_gravity(*g)
s.set_solver_limits(vmaxiter=20000)
s.particles = z["particles0"]
s.substep(dt)
ref = [s.grid(n).copy() for n in "UVW"]
den = max(np.abs(r).max() for r in ref)
for tol in (1e-3, 3e-4, 1e-4, 3e-5, 1e-5, 3e-6, 1e-6, 1e-7):
    c = Context(I, J, K, dx)
    c.set_solid_sdf(z["solid"]); c.set_viscosity(z["viscosity"]); c.set_gravity(*g)
    c.set_params(viscosity_preconditioner=pre, viscosity_max_iterations=2000, viscosity_tolerance=tol, check_every=1, exact_viscosity_operator=1)
    c.particles = z["particles0"]
    st = c.substep(dt)
    v = st["viscosity"]
    err = max(np.abs(c.grid(n).astype(np.float64) - ref[q]).max() for q, n in enumerate("UVW")) / den
    print("tol %.0e: its %3d status %d residual/rhs %.2e  velocity error %.2e" % (tol, v["iterations"], v["status"], v["residual"] / v["rhs_norm"], err), flush=True)
    c.close()
