"""PCG kernel time against the grid cap (flipv_params.reserved[2]) on the 256^3 bench scene: phase times of the third substep."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
from bench import build_scene
from flipviscosity3d_amd.capi import Context

N = 256
dx, solid, P = build_scene(N, 5.0)
for cap in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]:
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(5.0)
    c.particles = P
    p = c.get_params()
    p.reserved[2] = cap
    c._chk(c.L.flipv_set_params(c.h, C.byref(p)), "set")
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
    print("grid cap %5d" % cap, "viscosity %.2f ms (%d its)  project %.2f ms (%d its)  total %.2f" % (
        st["phase_ms"]["viscosity"], st["viscosity"]["iterations"], st["phase_ms"]["project"], st["pressure"]["iterations"], st["total_ms"]))
    c.close()
