"""GPU against the oracle run to 1e-10 / 1e-13 -- the SOLUTION of the reference's linear systems, not its 1e-6 iterate -- on the first substep of five scenes (sparse and dense,
uniform and variable viscosity): with default parameters, and with the solves tightened (precision = 1, tolerances 1e-9).  What is left with tight solves on both sides is what
the two implementations differ by outside their stop criteria.   python tools/r4_tight_oracle_scan.py > profiles/r4/tight_oracle_scan.log"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from test_gpu_stiff_regime import dense_scene
from flipviscosity3d_amd.capi import Context
from oracle import oraclebind as O
def run(name, N, dx, solid, P, nu, dims=None):
    I, J, K = dims or (N, N, N)
    o = O.OracleSim(I, J, K, dx); o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=3000000, vtol=1e-10, ptol=1e-13)
    o.particles = P
    sec, vi, pi = o.substep(0.01)
    ref = [o.grid(n) for n in "UVW"]; den = max(np.abs(r).max() for r in ref)
    for label, kw in (("default", {}), ("tight", dict(precision=1, viscosity_tolerance=1e-9, pressure_rel_tolerance=1e-9, viscosity_max_iterations=5000))):
        c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(nu)
        if kw: c.set_params(**kw)
        c.particles = P
        st = c.substep(0.01); v = st["viscosity"]
        err = max(np.abs(c.grid(n).astype(np.float64) - r).max() for n, r in zip("UVW", ref)) / den
        perr = np.abs(c.particles[:, :3] - o.particles[:, :3]).max()
        print("%-28s %-8s its %4d status %d layout %d  velocity err %.2e  particle pos %.1e  (oracle %d its)" % (name, label, v["iterations"], v["status"], v["layout"], err, perr, vi["iterations"]), flush=True)
        c.close()
    o.close()
N = 64
dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
y = (np.arange(N + 1) * dx)[None, :, None]
nuv = np.broadcast_to(5.0 + 195.0 * np.clip((0.42 + 2.0 * dx - y) / (4.0 * dx), 0.0, 1.0), (N + 1, N + 1, N + 1)).astype(np.float32).copy()
run("bunny64 variable nu", N, dx, solid, P, nuv)
dxh, solidh, Ph = build_host_scene(64, None, ["rod.ply", "sheet.ply"])
run("honey64 nu 50", 64, dxh, solidh, Ph, 50.0)
run("honey64 nu 800", 64, dxh, solidh, Ph, 800.0)
Nd = 48
dxd, solidd, Pd = dense_scene(Nd, 0.66)
Pd = Pd.copy(); Pd[:, 3] = 0.8 * np.sin(7.0 * Pd[:, 1]) * np.cos(5.0 * Pd[:, 2]); Pd[:, 4] = -0.3 * np.cos(6.0 * Pd[:, 0]); Pd[:, 5] = 0.5 * np.sin(4.0 * Pd[:, 0] + 3.0 * Pd[:, 1])
run("dense48 nu 150", Nd, dxd, solidd, Pd, 150.0)
x = (np.arange(Nd + 1) * dxd)[None, None, :]
nud = np.broadcast_to(10.0 + 290.0 * np.clip((x - 0.3) / 0.4, 0.0, 1.0), (Nd + 1, Nd + 1, Nd + 1)).astype(np.float32).copy()
run("dense48 variable nu", Nd, dxd, solidd, Pd, nud)
