"""research: the GPU's velocities after every phase of a substep on stored late states, for offline analysis against the oracle"""
import os, sys, glob
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context
N = 64
dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
os.makedirs("gpurun_out/r5", exist_ok=True)
for f in sorted(glob.glob("tools/r5_states/*.npz")):
    z = np.load(f); nu = float(z["nu"])
    out = {}
    c = Context(N, N, N, dx); c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_params(verbose=1)
    c.particles = z["particles"]
    c.particle_sdf(); c.advect_velocity_field(); c.body_force(0.01)
    out["phi"] = c.grid("LIQUID_PHI")
    for n in "UVW": out["pre_" + n] = c.grid(n)
    v = c.viscosity_solve(0.01)
    for n in "UVW": out["visc_" + n] = c.grid(n)
    c.compute_weights(); p = c.pressure_solve(0.01); c.apply_pressure(0.01)
    for n in "UVW": out["proj_" + n] = c.grid(n); out["valid_" + n] = c.grid("VALID_" + n)
    c.extrapolate(); c.constrain()
    for n in "UVW": out["final_" + n] = c.grid(n)
    print(os.path.basename(f), v, p["iterations"], flush=True)
    c.close()
    np.savez_compressed("gpurun_out/r5/gpudump2_" + os.path.basename(f), **out)
