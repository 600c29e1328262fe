import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_scene
from flipviscosity3d_amd.capi import Context
N=int(sys.argv[1]); prec=int(sys.argv[2]); cap=700
dx, solid, P = build_scene(N, 5.0)
c=Context(N,N,N,dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles=P
c.set_params(precision=prec, viscosity_max_iterations=cap)
for t in range(4):
    dt=min(c.cfl(),0.01)
    c.particle_sdf(); c.advect_velocity_field(); c.body_force(dt)
    vi=c.viscosity_solve(dt)
    out=np.zeros((5,cap+1)); 
    c.L.fvdbg_pcg_scalars(c.h, cap, cap+1, out.ctypes.data_as(C.POINTER(C.c_double)))
    sig,a,b,cc,rmax=out
    alpha=sig[:cap]/a[:cap]; est=sig[:cap]-2*alpha*b[:cap]+alpha**2*cc[:cap]; true=sig[1:cap+1]
    rel=np.abs(est-true)/np.abs(true)
    print(t,'visc its',vi['iterations'],'res',vi['residual'],'max rel dev est vs true sigma',rel.max(),'at',rel.argmax(),'median',np.median(rel), 'sigma range',sig.min(),sig.max(), 'rmax last', rmax[cap-1], 'min rmax', rmax[:cap].min())
    c.compute_weights(); pi=c.pressure_solve(dt); print('   pres rhs',pi['rhs_norm'],pi['iterations'])
    c.apply_pressure(dt); c.extrapolate(); c.constrain(); c.advect_particles(dt)
