"""Manual parity run at a size of your choice (not collected by pytest): HIP path against the oracle on the bunny scene.
    python tests/parity_vs_oracle.py <N> <substeps> [precision]
Lives under tests/ because it calls the oracle (test infrastructure)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np
from flipviscosity3d_amd import hostapi as H
from flipviscosity3d_amd.capi import Context
from oracle import oraclebind as O
N=int(sys.argv[1]); nsub=int(sys.argv[2])
M='tests/golden/meshes/'
dx=float(np.float32(1.0/N))
s=H.FluidSimulation(); s.initialize(N,N,N,dx); s.addBoundary(H.load_ply(M+'sphere_large.ply'),True); s.setSeeding(1,0); s.addLiquid(H.load_ply(M+'stanford_bunny.ply'))
solid=s.solid_sdf(); P=s.particles; s.close()
c=Context(N,N,N,dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles=P
prec=int(sys.argv[3]) if len(sys.argv)>3 else 0
c.set_params(precision=prec, viscosity_max_iterations=20000, viscosity_tolerance=1e-7 if prec else 1e-6, pressure_rel_tolerance=1e-7)
o=O.OracleSim(N,N,N,dx); o.set_solid(solid); o.set_viscosity(5.0); o.particles=P
o.set_solver_limits(vmaxiter=20000)
for t in range(nsub):
    st=c.substep(0.01); t0=time.time(); sec,vi,pi=o.substep(0.01)
    num=max(np.abs(c.grid(n).astype(np.float64)-o.grid(n)).max() for n in 'UVW'); den=max(np.abs(o.grid(n)).max() for n in 'UVW')
    print(t,'relerr',num/den,'gpu visc',st['viscosity']['iterations'],st['viscosity']['residual'],'pres',st['pressure']['iterations'],st['pressure']['rhs_norm'],'| oracle',vi['iterations'],pi['iterations'], 'cpu s',time.time()-t0, 'gpu ms', st['total_ms'])
