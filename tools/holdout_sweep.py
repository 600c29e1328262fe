#!/usr/bin/env python3
"""Holdout sweep for the viscosity solve's rule: the rule's constants (flipv.h: the two-stage solve, the velocity criterion) were scanned on
the bunny / honey / dense-box scenes at dt = 0.01 that the parity tests then assert.  This sweep draws scenes, sizes, time steps, viscosities and start states
that none of those scans used, from a FIXED seed, and compares the GPU's DEFAULT substep (no field of flipv_params set) with the solution of the reference's
linear systems (the oracle run to 1e-13 / 1e-13), on every face.

    python tools/holdout_sweep.py prepare [--cache DIR] [--workers 6]     # CPU: states + converged references (the oracle; minutes)
    python tools/holdout_sweep.py run [--cache DIR] [--only 3,7,...]      # GPU: the sweep; one line per draw
    python tools/holdout_sweep.py list                                    # the draws

A draw = (scene, N, dt, viscosity, k):  scene in bunny-in-sphere / honey (rod + sheet) / thin sheet (N x N/2 x N/2) / dense box (N x 3N/4 x N/2) / two bodies
with tilted gravity;  N in {40, 56, 72, 96};  dt uniform in [0.002, 0.01];  viscosity  (i) log-uniform in [1e-3, 5e3],  (ii) 5 % either side of every threshold of
the rule (nu dt/dx^2 = 64, 1 000, 2e4, 2e5),  (iii) a FIELD: a jump 1e-4 | 200, a jump 0 | 3 000, smooth 1 ... 1 000;  k in {0, 5, 40} oracle substeps (at the
reference's defaults) before the substep that is compared.  The oracle is test infrastructure; this tool lives with the other measurement scripts."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SEED = 5502026          # never used by any scan or test of rounds 1-5
THRESHOLDS = (64.0, 1000.0, 2.0e4, 2.0e5)
MESH = os.path.join(ROOT, "tests", "golden", "meshes")


def box_mesh(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y0, z1], [x0, y0, z1], [x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3], [4, 7, 6], [4, 6, 5], [0, 3, 7], [0, 7, 4], [1, 5, 6], [1, 6, 2], [0, 4, 5], [0, 5, 1], [3, 2, 6], [3, 6, 7]], np.int32)
    return v, t


def dims_of(scene, N):
    if scene == "sheet":
        return N, N // 2, N // 2
    if scene == "dense":
        return N, (3 * N) // 4, N // 2
    return N, N, N


def build_scene(scene, N):
    """-> I, J, K, dx, solid SDF, particles, gravity (host library: bit-exact level sets, tests/test_host_setup.py)"""
    import ctypes
    from flipviscosity3d_amd import hostapi as H
    I, J, K = dims_of(scene, N)
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(I, J, K, dx)
    g = (0.0, -9.81, 0.0)
    if scene == "bunny":
        s.addBoundary(H.load_ply(os.path.join(MESH, "sphere_large.ply")), True)
        ctypes.CDLL(None).srand(1)
        s.addLiquid(H.load_ply(os.path.join(MESH, "stanford_bunny.ply")))
    elif scene == "honey":
        ctypes.CDLL(None).srand(1)
        for m in ("rod.ply", "sheet.ply"):
            s.addLiquid(H.load_ply(os.path.join(MESH, m)))
    elif scene == "twobody":
        ctypes.CDLL(None).srand(1)
        for m in ("sphere_small.ply", "cone.ply"):
            s.addLiquid(H.load_ply(os.path.join(MESH, m)))
        g = (1.5, -9.81, 0.7)
    elif scene == "sheet":
        s.setSeeding(H.FluidSimulation.SEED_COUNTER, 11)
        s.addLiquid(box_mesh((0.05, 0.30, 0.05), (0.95, 0.30 + 2.1 * dx, 0.45)))
    elif scene == "dense":
        s.setSeeding(H.FluidSimulation.SEED_COUNTER, 12)
        s.addLiquid(box_mesh((0.06, 0.06, 0.06), (0.94, 0.45, 0.44)))
    else:
        raise SystemExit(scene)
    solid, P = s.solid_sdf(), s.particles
    s.close()
    return I, J, K, dx, solid, P, g


def viscosity_of(spec, I, J, K, dx):
    """a scalar, or the node field (K+1, J+1, I+1)"""
    kind = spec[0]
    if kind == "uniform":
        return float(spec[1])
    y = (np.arange(J + 1) * dx)[None, :, None]
    x = (np.arange(I + 1) * dx)[None, None, :]
    shape = (K + 1, J + 1, I + 1)
    if kind == "jump":            # lo below the plane y = 0.30 (it cuts every scene's liquid), hi above
        lo, hi = spec[1], spec[2]
        return np.broadcast_to(np.where(y < 0.30, lo, hi), shape).astype(np.float32).copy()
    if kind == "smooth":          # log-linear along x from lo to hi
        lo, hi = spec[1], spec[2]
        return np.broadcast_to(np.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * np.clip(x / (I * dx), 0.0, 1.0)) + 0.0 * y, shape).astype(np.float32).copy()
    raise SystemExit(kind)


def draws():
    rng = np.random.default_rng(SEED)
    scenes = ["bunny", "honey", "sheet", "dense", "twobody"]
    sizes = [40, 56, 72, 96]
    out = []

    def pick(cost_cap=None):
        while True:
            sc, N, k = scenes[rng.integers(5)], sizes[rng.integers(4)], (0, 5, 40)[rng.integers(3)]
            if cost_cap and N * N * N * (k + 3) > cost_cap:
                continue
            return sc, int(N), int(k)
    # (ii) the thresholds of the rule, 5 % either side
    for T in THRESHOLDS:
        for side in (0.95, 1.05):
            sc, N, k = pick(cost_cap=72 ** 3 * 10 if T >= 2e4 else None)
            dt = float(rng.uniform(0.002, 0.01))
            dx = float(np.float32(1.0 / N))
            out.append(dict(scene=sc, N=N, k=k, dt=dt, visc=("uniform", side * T * dx * dx / dt), why="nu dt/dx^2 = %.2f x %g" % (side, T)))
    # (iii) viscosity fields
    for spec in (("jump", 1e-4, 200.0), ("jump", 0.0, 3000.0), ("smooth", 1.0, 1000.0)):
        for _ in range(3):
            sc, N, k = pick(cost_cap=72 ** 3 * 10)
            out.append(dict(scene=sc, N=N, k=k, dt=float(rng.uniform(0.002, 0.01)), visc=spec, why="field %s %g | %g" % spec))
    # (i) log-uniform viscosities
    while len(out) < 48:
        sc, N, k = pick()
        nu = float(np.exp(rng.uniform(np.log(1e-3), np.log(5e3))))
        dt = float(rng.uniform(0.002, 0.01))
        if nu * dt * N * N > 3e4 and N * N * N * (k + 3) > 72 ** 3 * 10:     # (the oracle's converged solve of a stiff 96^3 chain: hours)
            continue
        out.append(dict(scene=sc, N=N, k=k, dt=dt, visc=("uniform", nu), why="log-uniform"))
    for i, d in enumerate(out):
        d["id"] = i
    return out


def describe(d):
    I, J, K = dims_of(d["scene"], d["N"])
    dx = float(np.float32(1.0 / d["N"]))
    v = d["visc"]
    numax = v[1] if v[0] == "uniform" else v[2]
    vs = "nu %.4g" % v[1] if v[0] == "uniform" else "%s %g | %g" % v
    return "%2d %-7s %3dx%3dx%3d dt %.4f k %2d  %-22s nu dt/dx^2 <= %9.1f  (%s)" % (d["id"], d["scene"], I, J, K, d["dt"], d["k"], vs, numax * d["dt"] / dx ** 2, d["why"])


def prepare_one(args):
    d, cache = args
    from oracle import oraclebind as O
    path = os.path.join(cache, "draw_%02d.npz" % d["id"])
    if os.path.exists(path):
        return d["id"], 0.0
    t0 = time.time()
    I, J, K, dx, solid, P, g = build_scene(d["scene"], d["N"])
    nu = viscosity_of(d["visc"], I, J, K, dx)
    dt = float(np.float32(d["dt"]))

    def sim(limits=None):
        o = O.OracleSim(I, J, K, dx)
        o.set_solid(solid); o.set_viscosity(nu); o.set_gravity(*g)
        if limits:
            o.set_solver_limits(**limits)
        return o
    o = sim()
    o.particles = P
    for _ in range(d["k"]):
        o.substep(dt)
    state = o.particles.copy()
    _, vd, pd = o.substep(dt)
    dflt = [o.grid(n) for n in "UVW"]
    o.close()
    q = sim(dict(vmaxiter=3000000, vtol=1e-13, ptol=1e-13))
    q.particles = state
    _, vq, pq = q.substep(dt)
    ref = [q.grid(n) for n in "UVW"]
    q.close()
    den = max(np.abs(r).max() for r in ref)
    err_ref = max(np.abs(a.astype(np.float64) - b).max() for a, b in zip(dflt, ref)) / den if den > 0 else 0.0
    out = dict(state=state, den=np.float64(den), err_ref_defaults=np.float64(err_ref), its_defaults=vd["iterations"], its_converged=vq["iterations"],
               pits_defaults=pd["iterations"], pits_converged=pq["iterations"], state_sum=np.float64(state.astype(np.float64).sum()))
    for n, r in zip("UVW", ref):
        nz = np.flatnonzero(r)
        out["idx_" + n] = nz.astype(np.uint32)
        out["val_" + n] = r.reshape(-1)[nz]
    np.savez_compressed(path, **out)
    return d["id"], time.time() - t0


def run(cache, only, params=None):
    from flipviscosity3d_amd.capi import Context
    worst = 0.0
    fails = []
    print("# holdout sweep, seed %d: GPU default substep against the oracle at 1e-13 / 1e-13, every face; 'reference' = the oracle at the reference's defaults against the same" % SEED)
    for d in draws():
        if only and d["id"] not in only:
            continue
        path = os.path.join(cache, "draw_%02d.npz" % d["id"])
        if not os.path.exists(path):
            print(describe(d), " -- not prepared")
            continue
        z = np.load(path)
        I, J, K, dx, solid, P, g = build_scene(d["scene"], d["N"])
        nu = viscosity_of(d["visc"], I, J, K, dx)
        c = Context(I, J, K, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_gravity(*g)
        if params:
            c.set_params(**params)   # (exploration only: the sweep proper runs with NO parameter set)
        c.particles = z["state"]
        st = c.substep(float(np.float32(d["dt"])))
        den = float(z["den"])
        err, nbad = 0.0, 0
        for n in "UVW":
            a = c.grid(n).reshape(-1).astype(np.float64)
            r = np.zeros_like(a)
            r[z["idx_" + n]] = z["val_" + n]
            e = np.abs(a - r) / den if den > 0 else np.abs(a - r)
            err = max(err, float(e.max()))
            nbad += int((e > 1e-4).sum())
        c.close()
        v = st["viscosity"]
        print("%s | GPU %.2e (%d faces > 1e-4) its %3d corr %3d prec %d status %d step %.1e pressure %3d | reference %.2e its %d (converged %d)%s" % (
            describe(d), err, nbad, v["iterations"], v["correction_iterations"], v["preconditioner"], v["status"], v["velocity_step"], st["pressure"]["iterations"],
            float(z["err_ref_defaults"]), int(z["its_defaults"]), int(z["its_converged"]),
            "   (the oracle ran into its cap of 3 000 000 iterations: NO converged reference for this draw)" if int(z["its_converged"]) >= 3000000 else ("   <-- FAIL" if err > 1e-4 else "")), flush=True)
        if int(z["its_converged"]) >= 3000000:
            continue
        worst = max(worst, err)
        if err > 1e-4:
            fails.append(d["id"])
    print("# worst %.2e; draws over 1e-4: %s" % (worst, fails))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=["prepare", "run", "list"])
    ap.add_argument("--cache", default=os.path.join(ROOT, "tools", "holdout_cache"))
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--params", default="", help="exploration: k=v,k=v parameter overrides (the sweep proper sets none)")
    a = ap.parse_args()
    only = [int(x) for x in a.only.split(",")] if a.only else []
    if a.cmd == "list":
        for d in draws():
            print(describe(d))
        return
    os.makedirs(a.cache, exist_ok=True)
    if a.cmd == "prepare":
        import multiprocessing as mp
        todo = [(d, a.cache) for d in draws() if not only or d["id"] in only]
        todo.sort(key=lambda t: -(t[0]["N"] ** 3) * (t[0]["k"] + 3))
        with mp.Pool(a.workers) as pool:
            for i, sec in pool.imap_unordered(prepare_one, todo):
                print("draw %d prepared in %.0f s" % (i, sec), flush=True)
        return
    prm = {}
    for kv in a.params.split(",") if a.params else []:
        k, v = kv.split("=")
        prm[k] = float(v) if any(ch in v for ch in ".e") else int(v)
    if prm:
        print("# EXPLORATION with", prm)
    run(a.cache, only, prm)


if __name__ == "__main__":
    main()
