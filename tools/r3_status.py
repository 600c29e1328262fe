"""long default-parameter run with a summary of the viscosity solves: python tools/r3_status.py workload size viscosity substeps [flipv_params field=value ...]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd.capi import Context
from bench import build_workload
wl, N, nu, nsub = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
I, J, K, dx, solid, P = build_workload(wl, N, on_device=True)
c = Context(I, J, K, dx)
c.set_solid_sdf(solid); c.set_viscosity(nu)
extra = {kv.split("=")[0]: (float(kv.split("=")[1]) if "." in kv.split("=")[1] or "e" in kv.split("=")[1] else int(kv.split("=")[1])) for kv in sys.argv[5:] if "=" in kv}   # flipv_params fields
if extra:
    c.set_params(**extra)
c.particles = P
status, prec, its, ms = collections.Counter(), collections.Counter(), [], []
pstatus, pits = collections.Counter(), []
t0 = time.perf_counter()
for t in range(nsub):
    st = c.substep(min(c.cfl(), 0.01))
    v = st["viscosity"]
    pstatus[st["pressure"]["status"]] += 1; pits.append(st["pressure"]["iterations"])
    status[v["status"]] += 1; prec[v["preconditioner"]] += 1; its.append(v["iterations"]); ms.append(st["total_ms"])
    if v["status"] != 0:
        print("substep %d: status %d after %d iterations (%d in correction stages, correction status %d), residual/rhs %.2e, defect residual/rhs %.2e, preconditioner %d" % (
            t, v["status"], v["iterations"], v["correction_iterations"], v["correction_status"], v["residual"] / max(v["rhs_norm"], 1e-300), v["defect_residual"] / max(v["rhs_norm"], 1e-300), v["preconditioner"]), flush=True)
c.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / nsub
print("%s %dx%dx%d nu %g, %d substeps: %.2f ms per substep (wall), GPU mean %.2f; viscosity status %s, preconditioner %s, iterations min/mean/max %d/%.1f/%d; last 5: %s; pressure status %s, iterations min/mean/max %d/%.1f/%d" % (
    wl, I, J, K, nu, nsub, wall, sum(ms) / nsub, dict(status), dict(prec), min(its), sum(its) / nsub, max(its), its[-5:], dict(pstatus), min(pits), sum(pits) / nsub, max(pits)))
