"""A/B over one measurement switch of flipv_params.reserved[] on the 256^3 bench scene:
    python tools/ab_switch.py <index> <value> [<value> ...]      e.g.  2 512 1024 2048  (grid cap)   3 2 4  (lane width)
prints the phase times of the third substep."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
from bench import build_scene
from flipviscosity3d_amd.capi import Context

N = int(os.environ.get("AB_N", "256"))
idx = int(sys.argv[1])
dx, solid, P = build_scene(N, 5.0)
for val in [int(a) for a in sys.argv[2:]]:
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(5.0)
    c.particles = P
    p = c.get_params()
    p.reserved[idx] = val
    c._chk(c.L.flipv_set_params(c.h, C.byref(p)), "set")
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
    print("reserved[%d] = %5d" % (idx, val), "viscosity %.2f ms (%d its, res %.3g)  project %.2f ms (%d its)  total %.2f" % (
        st["phase_ms"]["viscosity"], st["viscosity"]["iterations"], st["viscosity"]["residual"], st["phase_ms"]["project"],
        st["pressure"]["iterations"], st["total_ms"]))
    c.close()
