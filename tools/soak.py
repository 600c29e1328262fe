"""Soak run: the bench scene at N^3 for many substeps (bunny drop, splash, settling) -- watches for NaNs, lost particles,
solver failures and tile-geometry switches.   python tools/soak.py [N=128] [substeps=200] [auto|mg|diag] [bunny|honey|sheet] [viscosity=5]
(auto = the library default; FLIPV_VISC_AUTO=0 makes it the diagonal alone)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import build_workload
from flipviscosity3d_amd.capi import Context, PRECOND_DIAGONAL, PRECOND_MULTIGRID

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "auto"
workload = sys.argv[4] if len(sys.argv) > 4 else "bunny"
nu = float(sys.argv[5]) if len(sys.argv) > 5 else 5.0
I, J, K, dx, solid, P = build_workload(workload, N, on_device=True)
c = Context(I, J, K, dx)
c.set_solid_sdf(solid)
c.set_viscosity(nu)
if mode in ("mg", "diag"):
    c.set_params(viscosity_preconditioner=PRECOND_MULTIGRID if mode == "mg" else PRECOND_DIAGONAL)
c.particles = P
n0 = c.num_particles
last = None
worst = 0.0
total = 0.0
precs = []
phases = {}
wall = []
for t in range(steps):
    dt = min(c.cfl(), 0.01)
    t0_ = time.perf_counter()
    st = c.substep(dt)
    wall.append((time.perf_counter() - t0_) * 1e3)
    geo = (st["pressure"]["total_tiles"], st["viscosity"]["total_tiles"])
    if geo != last or t % 25 == 0 or st["rc"] not in (0, 1):
        print("substep %4d dt %.4f rc %d  visc %4d its st %d  pres %3d its st %d  tiles %s active %d/%d  %.2f ms" % (
            t, dt, st["rc"], st["viscosity"]["iterations"], st["viscosity"]["status"], st["pressure"]["iterations"], st["pressure"]["status"],
            geo, st["pressure"]["active_tiles"], st["viscosity"]["active_tiles"], st["total_ms"]), flush=True)
        last = geo
    worst = max(worst, st["total_ms"])
    total += st["total_ms"]
    precs.append(st["viscosity"]["preconditioner"])
    if t >= steps - 50:   # phase times of the last 50 substeps
        phases["total_ms"] = phases.get("total_ms", 0.0) + st["total_ms"] / min(50, steps)
        phases["wall_ms"] = phases.get("wall_ms", 0.0) + wall[-1] / min(50, steps)
        for k_, v_ in st["phase_ms"].items():
            phases[k_] = phases.get(k_, 0.0) + v_ / min(50, steps)
    assert st["rc"] >= 0, st
    if t % 25 == 24 or t == steps - 1:
        Q = c.particles
        assert np.isfinite(Q).all(), "non-finite particle state at substep %d" % t
        assert len(Q) == n0
        lo, hi = Q[:, :3].min(), Q[:, :3].max()
        assert lo >= 0.0 and hi <= max(I, J, K) * dx, (lo, hi)
        print("   particles ok: y range %.3f..%.3f, max speed %.3f" % (Q[:, 1].min(), Q[:, 1].max(), np.abs(Q[:, 3:]).max()), flush=True)
print("done: %d substeps, %.1f ms in all (%.2f ms per substep), worst %.2f ms; multigrid-preconditioned viscosity solves: %d" % (steps, total, total / steps, worst, sum(precs)))
print("phase ms over the last %d substeps: %s" % (min(50, steps), ", ".join("%s %.2f" % kv for kv in phases.items())))
c.close()
