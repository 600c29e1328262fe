"""back-to-back launch time of the brick SpMV on the bench scene against the grid cap (flipv_params.viscosity_spmv_grid_cap):
    python tools/r3_gridcap.py cap [cap ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from flipviscosity3d_amd.capi import Context
from bench import build_workload
I, J, K, dx, solid, P = build_workload("bunny", 256, on_device=True)
for cap in [int(a) for a in sys.argv[1:]]:
    c = Context(I, J, K, dx)
    c.set_solid_sdf(solid); c.set_viscosity(5.0)
    c.set_params(viscosity_spmv_grid_cap=cap)
    c.particles = P
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
    ms = min(c.bench_spmv(1, 300)[0] for _ in range(3))
    print("grid cap %5d: SpMV back to back %.2f us, viscosity phase %.2f ms (%d iterations)" % (cap, ms * 1e3, st["phase_ms"]["viscosity"], st["viscosity"]["iterations"]), flush=True)
    c.close()
