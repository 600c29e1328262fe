"""filled-box SpMV roofline with a pinned tile geometry: python tools/r3_dense.py size tile_rows [run_length]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from flipviscosity3d_amd import hostapi as H
from flipviscosity3d_amd.capi import Context
N, rowl = int(sys.argv[1]), int(sys.argv[2])
rl = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dx = float(np.float32(1.0 / N))
sim = H.FluidSimulation(); sim.initialize(N, N, N, dx); solid = sim.solid_sdf(); sim.close()
c = Context(N, N, N, dx)
c.set_solid_sdf(solid); c.set_viscosity(5.0)
c.set_params(pressure_max_iterations=4, viscosity_max_iterations=4, check_every=4, tile_rows=rowl, spmv_run_length=rl, viscosity_preconditioner=1)
rng = np.random.default_rng(0)
c.set_grid("LIQUID_PHI", np.full((N, N, N), -0.5 * dx, np.float32))
for n, shp in (("U", (N, N, N + 1)), ("V", (N, N + 1, N)), ("W", (N + 1, N, N))):
    c.set_grid(n, rng.uniform(-1, 1, shp).astype(np.float32))
c.compute_weights()
vi = c.viscosity_solve(0.01); pi = c.pressure_solve(0.01)
for which, name, b, units in ((0, "pressure", 24, float(pi["rows"])), (1, "viscosity", 52, vi["rows"] / 3.0)):
    ms, swept = c.bench_spmv(which, 30)
    print("%d^3 tile_rows %d run %d %-9s: %.1f us, %.0f GB/s algorithmic = %.3f of 8 TB/s (swept %.1f M, units %.1f M, layout %d)" %
          (N, rowl, rl, name, ms * 1e3, b * units / (ms * 1e-3) / 1e9, b * units / (ms * 1e-3) / 8e12, swept / 1e6, units / 1e6, vi["layout"]))
c.close()
