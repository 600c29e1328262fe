"""Filled-box SpMV roofline against the tile geometry (flipv_params.tile_rows: 0 = the library's choice, 16 = 64-wide rows, 64 = 256-wide rows) per extent.
    python tools/r4_dense_geo_scan.py 320,384,448,512 > profiles/r4/dense_geometry_scan.log"""
import os
import sys
sys.path.insert(0, os.getcwd())
import bench

sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "320,384,448,512").split(",")]
for N in sizes:
    for rows in (0, 16, 64):
        r = bench.dense_roofline(N, 0, reps=10, tile_rows=rows)
        print("%4d^3 tile_rows %2d: pressure %.3f of peak (%.0f us, swept %.1fM for %.1fM units)  viscosity %.3f (%.0f us)  viscosity, multigrid loop's %.3f (%.0f us, swept %.1fM for %.1fM)" % (
            N, rows, r["pressure_spmv"]["frac"], r["pressure_spmv"]["avg_launch_us"], r["pressure_spmv"]["swept_indices_per_launch"] / 1e6, r["pressure_spmv"]["units_per_launch"] / 1e6,
            r["viscosity_spmv"]["frac"], r["viscosity_spmv"]["avg_launch_us"], r["viscosity_spmv_multigrid_loop"]["frac"], r["viscosity_spmv_multigrid_loop"]["avg_launch_us"],
            r["viscosity_spmv_multigrid_loop"]["swept_indices_per_launch"] / 1e6, r["viscosity_spmv_multigrid_loop"]["units_per_launch"] / 1e6), flush=True)
