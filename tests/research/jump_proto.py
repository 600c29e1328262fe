#!/usr/bin/env python3
"""jump_proto.py -- scipy study of the viscosity systems the default GPU solve did not get through in round 5's holdout sweep (draws 9 and 11: viscosity FIELDS with a
jump across the liquid).  Solver research, CPU only; drives the oracle (test infrastructure), not collected by pytest.

    python tests/research/jump_proto.py dump 9          # the oracle's assembled system of the compared substep -> scratch/jump/draw_09.vdump
    python tests/research/jump_proto.py study 9
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
OUT = os.path.join(ROOT, "scratch", "jump")


def dump(i, hold="holdout_sweep", cache=None, vmaxiter=3000000, vtol=1e-13):
    H = __import__(hold)
    from oracle import oraclebind as O
    d = [x for x in H.draws() if x["id"] == i][0]
    cache = cache or os.path.join(ROOT, "tests", "golden", "holdout")
    z = np.load(os.path.join(cache, "draw_%02d.npz" % i))
    I, J, K, dx, solid, P, g = H.build_scene(d["scene"], d["N"])[:7]
    nu = H.viscosity_of(d["visc"], I, J, K, dx)
    o = O.OracleSim(I, J, K, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_gravity(*g)
    o.set_solver_limits(vmaxiter=vmaxiter, vtol=vtol, pmaxiter=0)
    o.particles = z["state"]
    path = os.path.join(OUT, "draw_%02d.vdump" % i)
    O.lib().oracle_viscosity_dump_to(path.encode())
    o.substep(float(np.float32(d["dt"])))
    O.lib().oracle_viscosity_dump_to(None)
    o.close()
    print(H.describe(d), "->", path)


def load(path):
    with open(path, "rb") as f:
        n, cap, dim, ext = np.fromfile(f, np.int64, 4)
        cnt = np.fromfile(f, np.int32, n)
        col = np.fromfile(f, np.uint32, n * cap).reshape(n, cap)
        val = np.fromfile(f, np.float64, n * cap).reshape(n, cap)
        rhs = np.fromfile(f, np.float64, n)
        table = np.fromfile(f, np.int32, dim)
        dgx = np.fromfile(f, np.float64, n)
        vol = np.fromfile(f, np.float64, n)
    m = np.arange(cap)[None, :] < cnt[:, None]
    rows = np.repeat(np.arange(n), cnt)
    A = sp.csr_matrix((val[m], (rows, col[m].astype(np.int64))), shape=(n, n))
    return A, rhs, dgx, vol, table


if __name__ == "__main__":
    cmd, i = sys.argv[1], int(sys.argv[2])
    os.makedirs(OUT, exist_ok=True)
    if cmd == "dump":
        dump(i)
