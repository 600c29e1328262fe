"""research (round 6): the GPU's post-viscosity velocities at the probe rows WITHOUT own volume of the mid-fall system fixture, written out for a look on the CPU
(which of the 496 beyond 1e-4 are rows held at 0, their partners, rows the independent solution itself had not settled)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_oracle_compact_golden import build_host_scene
from flipviscosity3d_amd.capi import Context
g = np.load(os.path.join(ROOT, "tests", "golden", "bunny256_nu5_sub10_system.npz"))
S = np.load(os.path.join(ROOT, "tests", "golden", "_big", "bunny256_nu5_sub10_state.npy"))
N = 256
dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
c = Context(N, N, N, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = S
c.particle_sdf(); c.advect_velocity_field(); c.body_force(0.01)
v = c.viscosity_solve(0.01)
out = {}
for k in "UVW":
    ml = g["massless_" + k]
    out["idx_" + k] = g["idx_" + k][ml]
    out["gpu_" + k] = c.grid(k).reshape(-1)[g["idx_" + k][ml]]
c.close()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r6"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "r6", "midfall_massless_gpu.npz"), **out)
print("written", v["iterations"], v["eliminated_rows"])
