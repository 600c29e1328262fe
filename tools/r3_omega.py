"""scan of the viscosity multigrid's smoother weights (and other parameters) on a bench workload:
    python tools/r3_omega.py workload size viscosity substeps w0,w1[,key=value,...] [...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd.capi import Context
from bench import build_workload
wl, N, nu, nsub = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
I, J, K, dx, solid, P = build_workload(wl, N, on_device=True)
for pair in sys.argv[5:]:
    parts = pair.split(",")
    w0, w1 = float(parts[0]), float(parts[1])
    extra = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in parts[2:]}
    c = Context(I, J, K, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu)
    c.set_params(viscosity_mg_omega_first=w0, viscosity_mg_omega_second=w1, **extra)
    c.particles = P
    its, ms, st_ = [], [], []
    for t in range(nsub):
        st = c.substep(min(c.cfl(), 0.01))
        v = st["viscosity"]
        its.append(v["iterations"]); ms.append(st["phase_ms"]["viscosity"]); st_.append(v["status"])
    print("w %.3f %.3f %s: its %s | sum %d | visc ms sum %.1f (last half %.1f) | status %s" % (w0, w1, extra, its, sum(its), sum(ms), sum(ms[nsub // 2:]), "".join(str(x) for x in st_)), flush=True)
    c.close()
