"""The defect predictor of the two-stage viscosity solve (flipv_params.viscosity_defect_predictor; stage 1 solves A x = b - E u_old): velocity error against the
reference's converged goldens, iterations (of which correction stage), and the bench window.
    python tools/r4_predict_scan.py <stage1_factor or 0> <bench: 0/1> <predictor: 0 on / -1 off>"""
import os
import sys
import time
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
from helpers import Golden
from test_oracle_compact_golden import build_host_scene, STIFF
from flipviscosity3d_amd.capi import Context
from r4_stage1_scan_lib import probe_err

f = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
pred = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tag = "predictor=%d f1=%g" % (pred, f)
for name, N in (("bunny128_nu5_converged", 128), ("bunny256_nu5_converged", 256)):
    g = Golden(name)
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"])); c.particles = P
    c.set_params(viscosity_stage1_factor=f, viscosity_defect_predictor=pred, verbose=int(os.environ.get('SCAN_VERBOSE', '0')), viscosity_stage2_factor=float(os.environ.get('SCAN_F2', '0')))
    out = []
    for t in range(g.nsub):
        st = c.substep(g.dt)
        v = st["viscosity"]
        out.append("%.2e (%d its, %d corr, status %d, defect %.2e)" % (probe_err(c, g, t), v["iterations"], v["correction_iterations"], v["status"], v["defect_residual"]))
    print("%-26s %s: %s" % (name, tag, "  ".join(out)), flush=True)
    c.close()
    if N == 256 and len(sys.argv) > 2 and int(sys.argv[2]):
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = P
        c.set_params(viscosity_stage1_factor=f, viscosity_defect_predictor=pred, verbose=int(os.environ.get('SCAN_VERBOSE', '0')), viscosity_stage2_factor=float(os.environ.get('SCAN_F2', '0')))
        for _ in range(5):
            c.substep(min(c.cfl(), 0.01))
        c.synchronize()
        t0 = time.perf_counter()
        sts = [c.substep(min(c.cfl(), 0.01)) for _ in range(20)]
        c.synchronize()
        el = time.perf_counter() - t0
        print("bench window %s: %.1f MCells/s, mean %.1f iterations (%.1f correction), statuses %s" % (
            tag, N ** 3 / 1e6 / (el / 20), np.mean([s["viscosity"]["iterations"] for s in sts]), np.mean([s["viscosity"]["correction_iterations"] for s in sts]),
            sorted(set(s["viscosity"]["status"] for s in sts))), flush=True)
        c.close()
for name, N, boundary, liquids in STIFF:
    g = Golden(name)
    dx, solid, P = build_host_scene(N, boundary, liquids)
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"]))
    c.set_params(viscosity_stage1_factor=f, viscosity_defect_predictor=pred, verbose=int(os.environ.get('SCAN_VERBOSE', '0')), viscosity_stage2_factor=float(os.environ.get('SCAN_F2', '0')))
    out = []
    for t in range(g.nsub):
        c.particles = P if t == 0 else g["s%d_particles" % (t - 1)]
        st = c.substep(g.dt)
        v = st["viscosity"]
        out.append("%.2e (%d its, %d corr, status %d, defect %.2e)" % (probe_err(c, g, t), v["iterations"], v["correction_iterations"], v["status"], v["defect_residual"]))
    print("%-26s %s: %s" % (name, tag, "  ".join(out)), flush=True)
    c.close()
