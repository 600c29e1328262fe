"""research: the pressure SpMV kernels back to back on a filled box (flipv_bench_spmv), per kernel choice"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from flipviscosity3d_amd import hostapi as H
from flipviscosity3d_amd.capi import Context
for N in [int(x) for x in sys.argv[1].split(",")]:
    dx = float(np.float32(1.0 / N))
    sim = H.FluidSimulation(); sim.initialize(N, N, N, dx); solid = sim.solid_sdf(); sim.close()
    for rl, name, var in ((0, "default", 0), (64, "march 64", 64), (-1, "tiles", 0)):
        c = Context(N, N, N, dx); c.set_solid_sdf(solid); c.set_viscosity(0.0)
        c.set_params(pressure_max_iterations=4, check_every=4, spmv_run_length=rl, tile_rows=var)
        rng = np.random.default_rng(0)
        c.set_grid("LIQUID_PHI", np.full((N, N, N), -0.5 * dx, np.float32))
        for n, shp in (("U", (N, N, N + 1)), ("V", (N, N + 1, N)), ("W", (N + 1, N, N))): c.set_grid(n, rng.uniform(-1, 1, shp).astype(np.float32))
        c.compute_weights(); pi = c.pressure_solve(0.01)
        ms, swept = c.bench_spmv(0, 20)
        print("L: PX %d PY %d" % (0, 0)) if False else None
        print("%d^3 %-12s: %8.1f us  %.3f of 8 TB/s (%d cells, %d swept)" % (N, name, ms * 1e3, 24.0 * pi["rows"] / (ms * 1e-3) / 8e12, pi["rows"], swept), flush=True)
        c.close()
