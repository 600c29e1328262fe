#!/usr/bin/env python3
"""SpMV roofline on a fully-filled box ('for pure kernel roofline runs use a fully-filled box so
that active ~ swept'): every interior cell liquid, default box boundary, random velocities.  No particles are
needed: the liquid SDF is written directly.  Prints one JSON line per size."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK = 8000.0


def run(N, precision, reps, viscosity=5.0, gridcap=0):
    from flipviscosity3d_amd import hostapi as H
    from flipviscosity3d_amd.capi import Context
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    solid = s.solid_sdf()
    s.close()
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(viscosity)
    c.set_params(precision=precision, pressure_max_iterations=4, viscosity_max_iterations=4, check_every=4)
    if gridcap:
        c.set_params(grid_cap=gridcap)
    rng = np.random.default_rng(0)
    c.set_grid("LIQUID_PHI", np.full((N, N, N), -0.5 * dx, np.float32))
    for n in "UVW":
        shp = {"U": (N, N, N + 1), "V": (N, N + 1, N), "W": (N + 1, N, N)}[n]
        c.set_grid(n, rng.uniform(-1, 1, shp).astype(np.float32))
    c.compute_weights()
    vi = c.viscosity_solve(0.01)
    pi = c.pressure_solve(0.01)
    out = {"gridcap": gridcap, "size": N, "precision": "f32" if precision == 0 else "f64", "device": c.device_name(),
           "copy_GBs": c.bench_copy(1 << 30, 10)}
    bpc = {0: (24, 52), 1: (32, 76)}[precision]  # fp64 vectors: s,z (pressure) / x,y (viscosity) double
    for which, name, b, info in ((0, "pressure_spmv", bpc[0], pi), (1, "viscosity_spmv", bpc[1], vi)):
        ms, cells = c.bench_spmv(which, reps)
        gbs = b * cells / (ms * 1e-3) / 1e9
        out[name] = {"avg_launch_us": ms * 1e3, "swept": cells, "active_tiles": info["active_tiles"],
                     "total_tiles": info["total_tiles"], "bytes_per_unit": b, "achieved_GBs": gbs, "frac_of_8TBs": gbs / PEAK}
    c.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[256, 384])
    ap.add_argument("--precision", type=int, default=0)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--gridcap", type=int, default=0, help="cap of the PCG kernels' grids (flipv_params.grid_cap); 0 = library default")
    a = ap.parse_args()
    for N in a.sizes:
        print(json.dumps(run(N, a.precision, a.reps, gridcap=a.gridcap)), flush=True)
