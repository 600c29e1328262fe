import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_scene
from flipviscosity3d_amd.capi import Context
N=int(sys.argv[1]); prec=int(sys.argv[2]); cap=int(sys.argv[3])
dx, solid, P = build_scene(N, 5.0)
c=Context(N,N,N,dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles=P
c.set_params(precision=prec, viscosity_max_iterations=cap)
for t in range(4):
    st=c.substep(min(c.cfl(),0.01))
    U=c.grid('U'); V=c.grid('V'); W=c.grid('W')
    print(t, 'visc', st['viscosity']['iterations'], '%.3e'%st['viscosity']['residual'], 'rhs %.4f'%st['viscosity']['rhs_norm'], '| pres', st['pressure']['iterations'], 'rhs %.4f'%st['pressure']['rhs_norm'], '%.2e'%st['pressure']['residual'], '| max|U,V,W|', abs(U).max(), abs(V).max(), abs(W).max(), 'ms %.1f'%st['total_ms'])
