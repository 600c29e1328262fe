"""research: what the velocity criterion buys and costs at a size where no converged reference is affordable.  The bench scene runs to the listed substeps at the
library's defaults; from each of those states one substep is taken with the criterion at several thresholds, and with the solves tightened to the point where they
no longer matter (precision = 1, tolerances 1e-9, criterion 1e-6): the distance to THAT is the error a threshold leaves.
    python tools/r5_eta_scan.py 256 30,50,70"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_workload
from flipviscosity3d_amd.capi import Context
N = int(sys.argv[1]); at = [int(x) for x in sys.argv[2].split(",")]
scene = sys.argv[3] if len(sys.argv) > 3 else "bunny"; nu = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0
I, J, K, dx, solid, P = build_workload(scene, N, on_device=True)
c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(nu); c.particles = P
states = {}
for t in range(max(at) + 1):
    if t in at: states[t] = c.particles.copy()
    c.substep(0.01)
c.close()
VAR = [("tight", dict(precision=1, viscosity_tolerance=1e-9, pressure_rel_tolerance=1e-9, viscosity_max_iterations=5000, viscosity_velocity_tolerance=1e-6)),
       ("round 4's rule", dict(viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1)),
       ("clusters + mass scale, no criterion", dict(viscosity_velocity_tolerance=-1.0)),
       ("no cluster solve", dict(viscosity_massless_polish=-1)),
       ("velocity criterion only", dict(viscosity_mass_scale=-1.0)),
       ("mass scale only", dict(viscosity_velocity_tolerance=-1.0)),
       ("default (3e-5, 100)", {}),
       ("3e-5, mass scale 30", dict(viscosity_mass_scale=30.0)),
       ("3e-5, mass scale 10", dict(viscosity_mass_scale=10.0)),
       ("1e-4, mass scale 100", dict(viscosity_velocity_tolerance=1e-4)),
       ("mass floor 0.1", dict(viscosity_mass_floor=0.1)), ("mass floor 0.3", dict(viscosity_mass_floor=0.3))]
if os.environ.get("R5_FLOORS_ONLY"):
    VAR = [v for v in VAR if v[0] in ("tight", "round 4's rule", "default (3e-5, 100)", "clusters + mass scale, no criterion", "no cluster solve")]
print("# %s %d^3 nu %g: one substep from the state after k default substeps; error = distance to the 'tight' run of the same state, relative max-norm over all faces" % (scene, N, nu))
for t in at:
    ref = None
    for name, kw in VAR:
        c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(nu)
        if kw: c.set_params(**kw)
        c.particles = states[t]
        st = c.substep(0.01); v = st["viscosity"]
        g = [c.grid(n) for n in "UVW"]
        c.close()
        if ref is None:
            ref = g; den = max(np.abs(a).max() for a in ref)
            print("after %3d substeps (max|u| %.3f): tight run %d viscosity iterations, status %d" % (t, den, v["iterations"], v["status"]), flush=True)
            continue
        e = [np.abs(a.astype(np.float64) - b) / den for a, b in zip(g, ref)]
        err = max(x.max() for x in e); n4 = sum(int((x > 1e-4).sum()) for x in e); n5 = sum(int((x > 1e-5).sum()) for x in e)
        print("   %-24s its %3d (corr %3d) status %d step %.1e  viscosity %.2f ms: error %.2e  (%d faces > 1e-4, %d > 1e-5)" % (name, v["iterations"], v["correction_iterations"], v["status"], v["velocity_step"], st["phase_ms"]["viscosity"], err, n4, n5), flush=True)
