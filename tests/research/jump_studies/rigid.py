import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/research")
import numpy as np
from jump_proto import load
from bench import build_workload
from oracle import oraclebind as O
N = int(sys.argv[1]); nsub = int(sys.argv[2])
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=False)
o = O.OracleSim(I, J, K, dx); o.set_solid(solid); o.set_viscosity(5.0); o.particles = P
o.set_solver_limits(vmaxiter=700)
path = "/root/repo/scratch/jump/rigid_%d.vdump" % N
for t in range(nsub):
    if t == nsub - 1: O.lib().oracle_viscosity_dump_to(path.encode())
    o.substep(0.01)
O.lib().oracle_viscosity_dump_to(None)
A, b, dgx, vol, table = load(path)
n = A.shape[0]
nu_, nv_ = (I + 1) * J * K, I * (J + 1) * K
comp = np.empty(n, int)
for c, (lo, hi) in enumerate(((0, nu_), (nu_, nu_ + nv_), (nu_ + nv_, len(table)))):
    t = table[lo:hi]; comp[t[t >= 0]] = c
x = np.fromfile(path + ".x", np.float64)
print("rows", n, "max|b|", np.abs(b).max(), "max|x|", np.abs(x).max())
c0 = np.zeros(n)
for c in range(3):
    m = comp == c
    u = np.where(vol[m] > 0, b[m] / np.maximum(vol[m], 1e-300), 0.0)   # incoming velocity of rows with own volume (b = vol u away from solids)
    mean = (vol[m] * u).sum() / vol[m].sum()
    c0[m] = mean
    print("comp", c, "mean incoming velocity", mean)
r0 = b - A @ c0
print("max|b - A c| / max|b| = %.3e ; rows beyond 1e-2: %d, 1e-3: %d" % (np.abs(r0).max() / np.abs(b).max(), (np.abs(r0) > 1e-2 * np.abs(b).max()).sum(), (np.abs(r0) > 1e-3 * np.abs(b).max()).sum()))
print("max|x - c|/max|x| = %.3e" % (np.abs(x - c0).max() / np.abs(x).max()))
