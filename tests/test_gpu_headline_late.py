"""GPU (-m gpu): the HEADLINE configuration -- BASELINE configs[2], the 256^3 bunny drop -- in the states bench.py's timed window and a long run actually spend their time in,
against the COMPILED REFERENCE (round 5 compared these states with the GPU's own tightened solve).

tests/golden/bunny256_nu5_sub10 / _sub25 (and bunny256_nu200_sub25: the last test) (tests/golden/make_golden.py K): the reference carried the scene at ITS defaults (dt = 0.01, viscosity
cap 700 -- at 256^3 every carried solve ends at the cap) through 10 substeps (mid-fall, inside the window bench.py times: substeps 5 ... 24), 20 (just landed, inside the window
too) and 25 (the liquid on the container wall), and at nu = 200 (nu dt/dx^2 = 131 072) through 25; the particles it then holds are the state.  From the state ONE substep of the reference with its viscosity cap lifted and the
tolerance at 1e-13: ~305 000 probe faces per component (300 000 seeded among the faces that carry a velocity, the 5 000 of largest |u|, every 4th face within one cell of the free
surface) and per-octant particle checksums.

The state itself (4.7 million particles, 113 MB) is NOT in git: tests/golden/_big/<name>_state.npy, written by make_golden.py and checked here by its sha256 from the committed
fixture.  Where the file is missing the test regenerates it -- oracle/_ref where it is built, else the bit-pinned oracle, through the same substeps -- and caches it there: by itself
up to 10 substeps (8 minutes of one host core), beyond that (17 / 21 minutes) only with FLIPV_REGENERATE_BIG=1 in the environment of the TEST, else it skips and says so.

GPU with NO field of flipv_params set: <= 1e-4 relative max-norm on every probe; round 4's rule and bench.py's strict mode (stage 1 to 1e-6) are printed beside it.  Also on 2 x 2 x 2 blocks."""
import hashlib
import os

import numpy as np
import pytest

from helpers import GOLDEN
from test_oracle_compact_golden import build_host_scene

pytestmark = pytest.mark.gpu
BIG = os.path.join(GOLDEN, "_big")
VEL_TOL = 1e-4
NAMES = ["bunny256_nu5_sub10", "bunny256_nu5_sub20", "bunny256_nu5_sub25"]   # (a state at 35 substeps was carried too; three 113 MB states are what a gpurun snapshot of 512 MiB holds)


def headline_state(g, name, P0, solid):
    """the reference's particles after g['nsub_before'] of its own substeps: the cached file if its sha256 is the fixture's, else regenerated (and cached)"""
    path = os.path.join(BIG, name + "_state.npy")
    if os.path.exists(path):
        S = np.load(path)
        if hashlib.sha256(np.ascontiguousarray(S).tobytes()).hexdigest() == str(g["state_sha256"]):
            return S
    N, dx, nu, k = int(g["I"]), float(g["dx"]), float(g["nu"]), int(g["nsub_before"])
    if k > 10 and os.environ.get("FLIPV_REGENERATE_BIG", "") != "1":     # (a switch of the TEST, not of the library: 20 / 25 substeps are 17 / 21 minutes of one host core each)
        pytest.skip("%s is not in this tree (113 MB, git-ignored) and regenerating it takes %d substeps of the reference at 256^3 (~1 minute each): set FLIPV_REGENERATE_BIG=1, "
                    "or run tests/golden/make_golden.py carry256_nu5" % (path, k))
    print("regenerating %s: %d substeps of the 256^3 scene on one host core ..." % (path, k), flush=True)
    from oracle import oraclebind as O, refbind as R
    if R.available():      # the compiled reference (build container, or a box that received oracle/_ref)
        from flipviscosity3d_amd.plyio import load_ply
        mesh = os.path.join(GOLDEN, "meshes")
        s = R.RefSim(N, N, N, dx)
        s.add_boundary(*load_ply(os.path.join(mesh, "sphere_large.ply")), True)
        R.lib().ref_srand(1)
        s.add_liquid(*load_ply(os.path.join(mesh, "stanford_bunny.ply")))
        s.set_viscosity(nu)
        for t in range(k):
            s.substep(float(g["dt"]))
        S = s.particles
        s.close()
    else:                  # the oracle, pinned to the reference iteration for iteration (tests/test_oracle_vs_reference.py)
        o = O.OracleSim(N, N, N, dx)
        o.set_solid(solid); o.set_viscosity(nu)
        o.particles = P0
        for t in range(k):
            o.substep(float(g["dt"]))
        S = o.particles.copy()
        o.close()
    assert hashlib.sha256(np.ascontiguousarray(S).tobytes()).hexdigest() == str(g["state_sha256"]), "the regenerated state is not the fixture's"
    os.makedirs(BIG, exist_ok=True)
    np.save(path, S)
    return S


def probe_error(g, uvw):
    den = float(g["maxabs"])
    worst, beyond, n = 0.0, 0, 0
    for c, a in zip("UVW", uvw):
        e = np.abs(a.reshape(-1)[g["probe_idx_" + c]].astype(np.float64) - g["probe_val_" + c]) / den
        worst = max(worst, float(e.max())); beyond += int((e > 1e-5).sum()); n += len(e)
    return worst, beyond, n


def load(name):
    # the fixture cut at 1e-13 where that run has finished, else the one at 1e-10 / 1e-8 from the same state (make_golden.py: ..._tol10, ..._tol8; the tolerance is printed)
    path = os.path.join(GOLDEN, name + ".npz")
    for suffix in ("_tol10", "_tol8"):
        if not os.path.exists(path):
            path = os.path.join(GOLDEN, name + suffix + ".npz")
    if not os.path.exists(path):
        pytest.skip("fixture %s not built (make_golden.py K: hours of one core)" % name)
    g = np.load(path)
    N = int(g["I"])
    dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"] and len(P0) == int(g["nparticles"])
    S = headline_state(g, name, P0, solid)
    assert np.array_equal(S.astype(np.float64).sum(axis=0), g["state_sum"])
    return g, N, dx, solid, S


@pytest.mark.parametrize("name", NAMES)
def test_headline_late_state_default_parameters_against_the_reference(name):
    from flipviscosity3d_amd.capi import Context
    g, N, dx, solid, S = load(name)

    def gpu(**prm):
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"]))
        if prm:
            c.set_params(**prm)
        c.particles = S
        st = c.substep(float(g["dt"]))
        out = [c.grid(n) for n in "UVW"], st, c.particles
        c.close()
        return out
    uvw, st, Pn = gpu()
    err, beyond, n = probe_error(g, uvw)
    old, st_old, _ = gpu(viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1, viscosity_pair_correction=-1)
    err_old, beyond_old, _ = probe_error(g, old)
    strict, st_s, _ = gpu(viscosity_stage1_factor=1.0)
    err_s, beyond_s, _ = probe_error(g, strict)
    v = st["viscosity"]
    print("%s (%d probe faces; the reference at %g: %d iterations, at its defaults %d iterations and %.2e from that on %d faces beyond 1e-4):\n"
          "   GPU default %.2e [%d probes beyond 1e-5] in %d viscosity iterations (status %d, velocity step %.1e, %d rows eliminated) | round 4's rule %.2e [%d] in %d | stage 1 to 1e-6 %.2e [%d] in %d" % (
              name, n, float(g["vtol"]), int(g["visc_iters"]), int(g["defaults_visc_iters"]), float(g["defaults_vs_converged"]), int(g["defaults_faces_beyond_1e-4"]),
              err, beyond, v["iterations"], v["status"], v["velocity_step"], v["eliminated_rows"], err_old, beyond_old, st_old["viscosity"]["iterations"], err_s, beyond_s, st_s["viscosity"]["iterations"]))
    assert v["status"] == 0 and st["pressure"]["status"] == 0, st
    assert err <= VEL_TOL, err
    oc = (Pn[:, 0] > 0.5).astype(int) + 2 * (Pn[:, 1] > 0.25).astype(int) + 4 * (Pn[:, 2] > 0.5).astype(int)
    sums = np.stack([Pn[oc == o].astype(np.float64).sum(axis=0) if (oc == o).any() else np.zeros(6) for o in range(8)])
    d = np.abs(sums - g["particles_octant_sum"]) / len(Pn)
    assert d[:, :3].max() <= 1e-6 and d[:, 3:].max() <= 1e-5, d      # (a particle that changes octant between the two runs would show as ~1/n per particle: none does)


@pytest.mark.parametrize("name", ["bunny256_nu5_sub10", "bunny256_nu5_sub20", "bunny256_nu5_sub25"])
def test_headline_late_state_default_blocks_against_the_reference(name):
    from test_gpu_multirank import assemble, run_ranks
    from test_gpu_multirank_default import assert_same_solve_on_every_rank, make_blocks
    g, N, dx, solid, S = load(name)
    ctxs = make_blocks(N, dx, solid, S, float(g["nu"]), (2, 2, 2))
    sts = run_ranks(ctxs, lambda r, c: c.substep(float(g["dt"])))
    assert_same_solve_on_every_rank(sts)
    err, beyond, n = probe_error(g, [assemble(ctxs, c) for c in "UVW"])
    v = sts[0]["viscosity"]
    print("%s on 2 x 2 x 2 blocks: %.2e [%d of %d probes beyond 1e-5] in %d viscosity iterations, status %d" % (name, err, beyond, n, v["iterations"], v["status"]))
    assert v["status"] == 0 and err <= VEL_TOL, (err, v)
    for c in ctxs:
        c.close()


def test_mid_fall_viscosity_solve_against_an_independent_solve_of_the_references_system():
    """The mid-fall state (10 substeps in, inside bench.py's timed window) is the one state of the headline run the compiled reference cannot be run to convergence from: its
    MIC(0)-PCG wanders for 19 000 iterations and then diverges (profiles/r6/sub10_reference_residual_history.log).  What can be pinned there is the viscosity SOLVE: the system the
    reference assembles from that state -- float-rounded diagonal, right-hand side and all, written by the bit-pinned oracle at assembly (oracle_viscosity_dump_to) -- solved by an
    INDEPENDENT method, fp64 diagonal-PCG in scipy (tests/golden/make_golden.py bunny256_nu5_sub10_system; the fixture states its true residual), against the face velocities the GPU
    holds after flipv_viscosity_solve with NO parameter set.  Bar: <= 1e-4 max|x| on every probe row that HAS own volume (~300 000 seeded rows per component, the 5 000 largest).
    Rows WITHOUT own volume (up to 100 000 probes per component) are printed, not asserted: the independent solution itself is not settled on them -- between true residuals of
    1.8e-7 and 7.8e-9 max|rhs| (9 850 and 36 400 iterations) 114 such rows, diagonals 4e-7 ... 44 beside a median of 26 000, still moved by up to 0.38 max|x| while every row with
    own volume had moved by < 1e-5 --, and where they repeat one equation the reference's matrix is singular (any split is a solution; the library holds the later rows at 0)."""
    from flipviscosity3d_amd.capi import Context
    path = os.path.join(GOLDEN, "bunny256_nu5_sub10_system.npz")
    spath = os.path.join(BIG, "bunny256_nu5_sub10_state.npy")
    if not (os.path.exists(path) and os.path.exists(spath)):
        pytest.skip("fixture or state not present (make_golden.py carry256_nu5, bunny256_nu5_sub10_system)")
    g = np.load(path)
    S = np.load(spath)
    assert hashlib.sha256(np.ascontiguousarray(S).tobytes()).hexdigest() == str(g["state_sha256"])
    N = int(g["I"])
    dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"]))
    c.particles = S
    dt = float(g["dt"])
    c.particle_sdf(); c.advect_velocity_field(); c.body_force(dt)      # the substep up to the viscosity solve (fluidsimulation.cpp:145-153)
    v = c.viscosity_solve(dt)
    den = float(g["maxabs"])
    worst, worst_ml, nbad, nheld, n, nml = 0.0, 0.0, 0, 0, 0, 0
    for k in "UVW":
        a = c.grid(k).reshape(-1)[g["idx_" + k]].astype(np.float64)
        ref = g["val_" + k]
        ml = g["massless_" + k]
        held = ml & (a == 0.0) & (ref != 0.0)          # rows held at 0 (a massless row that repeats another row's equation)
        e = np.abs(a - ref) / den
        worst = max(worst, float(e[~ml].max())); worst_ml = max(worst_ml, float(e[ml].max()) if ml.any() else 0.0)
        nbad += int((e[ml] > VEL_TOL).sum()); nheld += int(held.sum()); n += int((~ml).sum()); nml += int(ml.sum())
    c.close()
    print("mid-fall state, the viscosity solve alone (the reference's system solved by fp64 diagonal-PCG: %d iterations, true residual %.1e max|rhs|): GPU default %.2e on %d probe rows "
          "with own volume | %d probe rows without: worst %.2e, %d beyond 1e-4, %d of them held at 0 (%d rows eliminated in all) | %d iterations, status %d, velocity step %.1e" % (
              int(g["iterations"]), float(g["true_residual"]), worst, n, nml, worst_ml, nbad, nheld, v["eliminated_rows"], v["iterations"], v["status"], v["velocity_step"]))
    assert v["status"] == 0, v
    assert worst <= VEL_TOL, worst


def test_headline_size_at_nu_200_is_not_pinned_and_the_solve_says_so():
    """nu = 200 at 256^3 (nu dt/dx^2 = 131 072), 25 substeps in: the compiled reference needs 15 148 iterations for 1e-8 from its own state (at its defaults it stops at the cap of
    700, 0.82 max|u| from that), and 1e-8 is NOT converged in the velocities there: GPU runs at 6e-8 ... 2e-8 relative residual (stage 1 to 1e-6 under a lifted cap; 19 355 fp64
    diagonal-PCG iterations) are 0.19 / 0.52 from the fixture and as far from each other (profiles/r6/nu200_256_late_probe.log); the reference's run at 1e-10 had not ended after
    four hours of one core.  So this state pins nothing to 1e-4.  What IS asserted: the default solve does not claim what it has not got -- flipv_solve_info.status != 0 (a stage
    short of its target inside the reference's cap) whenever it is beyond the bar against the best fixture there is -- and it is no further from that fixture than the reference at
    its own defaults."""
    from flipviscosity3d_amd.capi import Context
    if not os.path.exists(os.path.join(BIG, "bunny256_nu200_sub25_state.npy")):
        pytest.skip("the state (113 MB) is not in this tree: a gpurun snapshot holds three of them and this one pins nothing (make_golden.py carry256_nu200 writes it)")
    g, N, dx, solid, S = load("bunny256_nu200_sub25")
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(float(g["nu"]))
    c.particles = S
    st = c.substep(float(g["dt"]))
    err, beyond, n = probe_error(g, [c.grid(k) for k in "UVW"])
    c.close()
    v = st["viscosity"]
    print("bunny256_nu200_sub25 (the reference at %g: %d iterations; at its defaults %.2e from that): GPU default %.2e in %d viscosity iterations, status %d (correction stage %d), velocity step %.1e" % (
        float(g["vtol"]), int(g["visc_iters"]), float(g["defaults_vs_converged"]), err, v["iterations"], v["status"], v["correction_status"], v["velocity_step"]))
    assert err <= VEL_TOL or v["status"] != 0, (err, v)
    assert err <= float(g["defaults_vs_converged"]), (err, float(g["defaults_vs_converged"]))
