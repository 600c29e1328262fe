import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_scene
from flipviscosity3d_amd.capi import Context
N=256
dx, solid, P = build_scene(N, 5.0)
for nog in (1,0,1,0):
    c=Context(N,N,N,dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles=P
    p=c.get_params(); p.reserved[0]=nog; 
    import ctypes as C
    c._chk(c.L.flipv_set_params(c.h, C.byref(p)),'set')
    for t in range(3):
        st=c.substep(min(c.cfl(),0.01))
    print('nograph' if nog else 'graph  ', {k: round(v,2) for k,v in st['phase_ms'].items()}, 'total %.1f'%st['total_ms'], st['viscosity']['iterations'], st['pressure']['iterations'])
    c.close()
