"""where is the velocity error of the multigrid-preconditioned solve on the twobody fixture?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd import capi
from flipviscosity3d_amd.capi import Context
from oracle import oraclebind as O
z = np.load(os.path.join(ROOT, "tests", "golden", "twobody20_varvisc.npz"))
I, J, K = int(z["I"]), int(z["J"]), int(z["K"])
dx, dt = float(z["dx"]), float(z["dt"])
g = [float(v) for v in z["gravity"]]
s = O.OracleSim(I, J, K, dx)
s.set_solid(z["solid"]); s.set_viscosity(z["viscosity"]); s.set_gravity(*g)
s.particles = z["particles0"]
c = Context(I, J, K, dx)
c.set_solid_sdf(z["solid"]); c.set_viscosity(z["viscosity"]); c.set_gravity(*g)
c.set_params(viscosity_preconditioner=int(sys.argv[1]) if len(sys.argv) > 1 else 2, verbose=2)
c.particles = z["particles0"]
# phase by phase
s.particle_sdf() if hasattr(s, "particle_sdf") else None
print("viscosity nodes: min %.3g max %.3g" % (z["viscosity"].min(), z["viscosity"].max()))
for obj in (c,):
    obj.particle_sdf(); obj.advect_velocity_field(); obj.body_force(dt)
pre = [c.grid(n).copy() for n in "UVW"]
vi = c.viscosity_solve(dt)
print(vi)
post = [c.grid(n).copy() for n in "UVW"]
c.set_params(viscosity_preconditioner=1, viscosity_tolerance=1e-9, viscosity_max_iterations=5000)
for q, n in enumerate("UVW"):
    c.set_grid(n, pre[q])
vi2 = c.viscosity_solve(dt)
print(vi2)
ref = [c.grid(n).copy() for n in "UVW"]
den = max(np.abs(r).max() for r in ref)
for q, n in enumerate("UVW"):
    d = np.abs(post[q].astype(np.float64) - ref[q])
    idx = np.unravel_index(np.argmax(d), d.shape)
    print(n, "max err %.3e at %s (k,j,i), value %.4g ref %.4g, |err|>1e-4*den at %d faces" % (d.max() / den, idx, post[q][idx], ref[q][idx], int((d > 1e-4 * den).sum())))
    kk, jj, ii = idx
    print("   viscosity at that node: %.3g ; neighbours err: %s" % (z["viscosity"][kk, jj, ii], np.array2string(d[max(kk-1,0):kk+2, jj, ii] / den, precision=2)))
