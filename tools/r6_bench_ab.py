"""research (round 6): the bench window (256^3 bunny, 5 + 20 substeps) under parameter variations: mean viscosity iterations and ms per substep"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import build_workload
from flipviscosity3d_amd.capi import Context
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
def run(label, **prm):
    c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = P
    if prm: c.set_params(**prm)
    for _ in range(5): c.substep(min(c.cfl(), 0.01))
    c.synchronize(); t0 = time.perf_counter(); its = []; corr = []; st_ = []
    for _ in range(20):
        st = c.substep(min(c.cfl(), 0.01)); its.append(st["viscosity"]["iterations"]); corr.append(st["viscosity"]["correction_iterations"]); st_.append(st["viscosity"]["status"])
    c.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 20
    print("%-60s %.2f ms/substep %7.1f MCells/s  viscosity iterations mean %.1f (correction %.1f) %s status!=0: %d" % (label, ms, I * J * K / 1e3 / ms, np.mean(its), np.mean(corr), its, sum(1 for s in st_ if s)), flush=True)
    c.close()
run("default")
run("default (again)")
run("elimination / polish off", viscosity_massless_polish=-1)
run("pairs forced on", viscosity_pair_correction=1)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from r6_jump_diag import run as jrun
jrun(9, reps=8)
jrun(11, reps=3)
