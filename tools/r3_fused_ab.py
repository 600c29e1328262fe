"""A/B of the one-kernel Jacobi-PCG iteration against SpMV + update (brick layout, diagonal preconditioner, cap 700 like the reference):
   python tools/r3_fused_ab.py [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from flipviscosity3d_amd import capi
from flipviscosity3d_amd.capi import Context
from bench import build_workload
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
res = {}
for two in (1, 0):
    c = Context(I, J, K, dx)
    c.set_solid_sdf(solid); c.set_viscosity(5.0)
    c.set_params(viscosity_preconditioner=capi.PRECOND_DIAGONAL, two_kernel_pcg=two, exact_viscosity_operator=1)
    c.particles = P
    out = []
    for t in range(4):
        st = c.substep(0.01)
        v = st["viscosity"]
        out.append((v["iterations"], v["status"], st["phase_ms"]["viscosity"], v["residual"] / v["rhs_norm"]))
    res[two] = [c.grid(n) for n in "UVW"]
    print("two_kernel_pcg=%d:" % two, " ; ".join("%d its st %d %.2f ms res %.2e" % o for o in out), flush=True)
    c.close()
num = max(np.abs(a.astype(np.float64) - b).max() for a, b in zip(res[0], res[1]))
den = max(np.abs(a).max() for a in res[1])
print("velocity difference fused vs two-kernel after 4 capped substeps: %.3e" % (num / den))
