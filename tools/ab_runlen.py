"""Scan of the k-marching run length (flipv_params.spmv_run_length; -1 = tile-at-a-time kernels) and the tile geometry:
    python tools/ab_runlen.py bench|dense [N] [--tile-rows 16|64] -- <runlen> [<runlen> ...]
bench: the 256^3 bunny scene (sparse liquid), third substep's solve times + back-to-back SpMV launch times.
dense: filled N^3 box, back-to-back SpMV launch times and algorithmic GB/s (24 B / 52 B per unit)."""
import os
import sys
sys.path.insert(0, os.getcwd())
import numpy as np
from flipviscosity3d_amd.capi import Context

mode = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "--" else 256
vals = [int(a) for a in sys.argv[sys.argv.index("--") + 1:]]
ROWL = int(sys.argv[sys.argv.index("--tile-rows") + 1]) if "--tile-rows" in sys.argv else 0
tag = "tile_rows=%s" % (ROWL or "auto")
if mode == "bench":
    from bench import build_scene
    dx, solid, P = build_scene(N, 5.0)
    for rl in vals:
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = P
        c.set_params(spmv_run_length=rl, tile_rows=ROWL)
        for t in range(3):
            st = c.substep(min(c.cfl(), 0.01))
        vms, _ = c.bench_spmv(1, 200)
        pms, _ = c.bench_spmv(0, 200)
        print(tag, "runlen %3d: viscosity %.2f ms (%d its, res %.3g) project %.2f ms (%d its) total %.2f | SpMV b2b visc %.2f us pres %.2f us" % (
            rl, st["phase_ms"]["viscosity"], st["viscosity"]["iterations"], st["viscosity"]["residual"], st["phase_ms"]["project"],
            st["pressure"]["iterations"], st["total_ms"], vms * 1e3, pms * 1e3), flush=True)
        c.close()
else:
    from flipviscosity3d_amd import hostapi as H
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation(); s.initialize(N, N, N, dx); solid = s.solid_sdf(); s.close()
    rng = np.random.default_rng(0)
    uvw = {n: rng.uniform(-1, 1, shp).astype(np.float32) for n, shp in (("U", (N, N, N + 1)), ("V", (N, N + 1, N)), ("W", (N + 1, N, N)))}
    for rl in vals:
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(5.0)
        c.set_params(pressure_max_iterations=4, viscosity_max_iterations=4, check_every=4, spmv_run_length=rl, tile_rows=ROWL)
        c.set_grid("LIQUID_PHI", np.full((N, N, N), -0.5 * dx, np.float32))
        for n in "UVW":
            c.set_grid(n, uvw[n])
        c.compute_weights()
        vi = c.viscosity_solve(0.01); pi = c.pressure_solve(0.01)
        vms, _ = c.bench_spmv(1, 50)
        pms, _ = c.bench_spmv(0, 50)
        print(tag, "tiles v %d p %d" % (vi["total_tiles"], pi["total_tiles"]), "runlen %3d: visc SpMV %.1f us = %.0f GB/s (%.3f of 8 TB/s) | pressure SpMV %.1f us = %.0f GB/s (%.3f)" % (
            rl, vms * 1e3, 52 * vi["rows"] / 3 / vms / 1e6, 52 * vi["rows"] / 3 / vms / 1e6 / 8000,
            pms * 1e3, 24 * pi["rows"] / pms / 1e6, 24 * pi["rows"] / pms / 1e6 / 8000), flush=True)
        c.close()
