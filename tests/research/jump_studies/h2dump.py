import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests/research")
import numpy as np
import holdout_sweep2 as H
from oracle import oraclebind as O
i = int(sys.argv[1])
d = [x for x in H.draws() if x["id"] == i][0]
z = np.load("/root/repo/tools/holdout2_cache/draw_%02d.npz" % i)
I, J, K, dx, solid, P = H.build_scene(d["scene"], d["N"])
nu = H.viscosity_of(d["visc"], I, J, K, dx)
o = O.OracleSim(I, J, K, dx); o.set_solid(solid); o.set_viscosity(nu); o.set_gravity(*d["gravity"])
o.set_solver_limits(vmaxiter=200000, vtol=1e-13, pmaxiter=0)
o.particles = z["state"]
path = "/root/repo/scratch/jump/h2_%02d.vdump" % i
O.lib().oracle_viscosity_dump_to(path.encode())
o.substep(float(z["dt"]))
O.lib().oracle_viscosity_dump_to(None)
print(H.describe(d), "->", path)
