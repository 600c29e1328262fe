"""research (round 6): the filled-box viscosity SpMV -- sweep-ordered tile list (spmv_run_length 0 / -2) against the k-marching kernel (a run length) and the tile list in address order (-1)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import dense_roofline
for N in [int(a) for a in sys.argv[1:]] or [256, 384, 512]:
    for rl, label in ((0, "default"), (-2, "sweep order forced"), (32, "march, runs of 32"), (-1, "tile at a time, address order")):
        try:
            d = dense_roofline(N, 0, reps=30, spmv_run_length=rl)
            print("%d^3 %-32s viscosity: diagonal loop %.3f (%.0f us)  multigrid loop %.3f (%.0f us) | pressure %.3f | mix_10to3 %.0f GB/s" % (
                N, label, d["viscosity_spmv"]["frac"], d["viscosity_spmv"]["avg_launch_us"], d["viscosity_spmv_multigrid_loop"]["frac"], d["viscosity_spmv_multigrid_loop"]["avg_launch_us"],
                d["pressure_spmv"]["frac"], d["attainable_GBs"]["mix_10to3"]), flush=True)
        except Exception as e:
            print(N, label, "failed:", e, flush=True)
