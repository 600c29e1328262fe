"""A/B: LDS-binned particle scatters (default) against the un-binned global-atomic kernels (flipv_params.unbinned_scatter=1)
on the 256^3 bench scene; prints the sdf / p2g phase times of the third substep."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
from bench import build_scene
from flipviscosity3d_amd.capi import Context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dx, solid, P = build_scene(N, 5.0)
for unbinned in (1, 0, 1, 0):
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(0.0)   # viscosity off: only the particle phases matter here
    c.particles = P
    c.set_params(unbinned_scatter=unbinned)
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
    print("global atomics" if unbinned else "LDS bins      ", {k: round(v, 3) for k, v in st["phase_ms"].items()}, "total %.2f" % st["total_ms"])
    c.close()
