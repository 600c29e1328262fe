"""A/B over one measurement switch of flipv_params (by field name) on the 256^3 bench scene:
    python tools/ab_switch.py <field> <value> [<value> ...]      e.g.  grid_cap 512 1024 2048     viscosity_lane_width 2 4
prints the phase times of the third substep."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
from bench import build_scene
from flipviscosity3d_amd.capi import Context

N = int(os.environ.get("AB_N", "256"))
field = sys.argv[1]
dx, solid, P = build_scene(N, 5.0)
for val in [int(a) for a in sys.argv[2:]]:
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(5.0)
    c.particles = P
    c.set_params(**{field: val})
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
    print("%s = %5d" % (field, val), "viscosity %.2f ms (%d its, res %.3g)  project %.2f ms (%d its)  total %.2f" % (
        st["phase_ms"]["viscosity"], st["viscosity"]["iterations"], st["viscosity"]["residual"], st["phase_ms"]["project"],
        st["pressure"]["iterations"], st["total_ms"]))
    c.close()
