"""CPU: the C restatement against the two compact reference dumps at BASELINE sizes (tests/golden/make_golden.py D, E).

 * honey64_nu50 -- config #4 in miniature (rod.ply + sheet.ply through two addLiquid calls, nu = 50, 64^3) with the
   viscosity cap lifted in the reference harness: the oracle must land on the reference's converged answer bit for bit,
   iteration for iteration.
 * bunny128_nu5_converged -- config #3's scene at the largest size where the reference converges (128^3): the scene the
   host library builds must BE the reference's (count and checksums); the oracle's first substep is compared at the
   fixture's probe faces (one substep: ~30 s of the CPU suite's budget).
"""
import ctypes
import os

import numpy as np
import pytest

from helpers import GOLDEN, Golden

MESH = os.path.join(GOLDEN, "meshes")


def build_host_scene(N, boundary, liquids):
    """the reference's scene through the host library with the reference's libc rand() stream (bit-exact:
    tests/test_host_setup.py)"""
    from flipviscosity3d_amd import hostapi as H
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    if boundary:
        s.addBoundary(H.load_ply(os.path.join(MESH, boundary[0])), boundary[1])
    ctypes.CDLL(None).srand(1)
    for m in liquids:
        s.addLiquid(H.load_ply(os.path.join(MESH, m)))
    solid, P = s.solid_sdf(), s.particles
    s.close()
    return dx, solid, P


def test_honey_scene_setup_matches_reference():
    g = Golden("honey64_nu50")
    dx, solid, P = build_host_scene(64, None, ["rod.ply", "sheet.ply"])
    assert np.array_equal(solid, g["solid"])
    assert np.array_equal(P, g["particles0"])          # the second addLiquid appends (fluidsimulation.cpp:89)


def test_oracle_honey64_converged(oracle):
    g = Golden("honey64_nu50")
    I, J, K = g.dims()
    s = oracle.OracleSim(I, J, K, g.dx)
    s.set_solid(g["solid"])
    s.set_viscosity(float(g["nu"]))
    s.set_solver_limits(vmaxiter=int(g["vcap"]))
    s.particles = g["particles0"]
    for t in range(g.nsub):
        sec, vi, pi = s.substep(g.dt)
        assert vi["iterations"] == int(g["s%d_visc_iters" % t]) and vi["iterations"] > 700   # beyond the stock cap
        assert pi["iterations"] == int(g["s%d_pres_iters" % t])
        for c in "UVW":
            assert np.array_equal(s.grid(c), g["s%d_final_%s" % (t, c)])
        assert np.array_equal(s.particles, g["s%d_particles" % t])
    s.close()


def test_oracle_bunny128_first_substep_at_probes(oracle):
    g = Golden("bunny128_nu5_converged")
    dx, solid, P = build_host_scene(128, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    assert len(P) == int(g["nparticles"])
    assert np.array_equal(P.astype(np.float64).sum(axis=0), g["particles0_sum"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"]
    s = oracle.OracleSim(128, 128, 128, dx)
    s.set_solid(solid)
    s.set_viscosity(float(g["nu"]))
    s.set_solver_limits(vmaxiter=int(g["vcap"]))
    s.particles = P
    sec, vi, pi = s.substep(g.dt)
    assert vi["iterations"] == int(g["s0_visc_iters"]) and pi["iterations"] == int(g["s0_pres_iters"])
    for c in "UVW":
        a = s.grid(c).reshape(-1)
        assert np.array_equal(a[g["s0_probe_idx_" + c]], g["s0_probe_val_" + c])
        assert np.float32(np.abs(a).max()) == g["s0_maxabs_" + c]
    assert np.array_equal(s.particles.astype(np.float64).sum(axis=0), g["s0_particles_sum"])
    s.close()


def test_bunny256_scene_setup_matches_reference():
    """bunny256_nu5_converged (the reference's converged answer at the HEADLINE size, cap lifted: ~25 minutes per substep, so neither the
    reference nor the oracle is re-run here): the scene the GPU test builds with the host library must be the reference's -- particle
    count, particle checksum, solid-SDF checksum -- and the fixture must say what it is"""
    if not os.path.exists(os.path.join(GOLDEN, "bunny256_nu5_converged.npz")):
        pytest.skip("fixture not built")
    g = Golden("bunny256_nu5_converged")
    assert g.dims() == (256, 256, 256) and int(g["vcap"]) >= 20000 and float(g["nu"]) == 5.0
    assert int(g["s0_visc_iters"]) > 700                     # far beyond the stock cap: the dump is the converged answer
    dx, solid, P = build_host_scene(256, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    assert len(P) == int(g["nparticles"])
    assert np.array_equal(P.astype(np.float64).sum(axis=0), g["particles0_sum"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"]


def test_bunny256_whole_field_fixture_is_the_same_reference_run():
    """bunny256_nu5_converged_wide (round 5: ~337 000 probe faces per component -- 200 000 seeded, the 5 000 of largest |u|, every 4th face within one cell of the
    free surface -- and per-octant particle checksums) is the SAME reference run as bunny256_nu5_converged: same scene, same iteration counts, and bit-identical
    velocities on every probe face the two fixtures share; its probes reach the top of the velocity range."""
    if not (os.path.exists(os.path.join(GOLDEN, "bunny256_nu5_converged.npz")) and os.path.exists(os.path.join(GOLDEN, "bunny256_nu5_converged_wide.npz"))):
        pytest.skip("fixtures not built")
    a, b = Golden("bunny256_nu5_converged"), Golden("bunny256_nu5_converged_wide")
    assert b.dims() == (256, 256, 256) and np.array_equal(a["particles0_sum"], b["particles0_sum"]) and a["solid_sum"] == b["solid_sum"]
    for t in range(2):
        assert int(a["s%d_visc_iters" % t]) == int(b["s%d_visc_iters" % t]) and int(a["s%d_pres_iters" % t]) == int(b["s%d_pres_iters" % t])
        assert np.array_equal(a["s%d_particles_sum" % t], b["s%d_particles_sum" % t])
        assert np.allclose(b["s%d_particles_octant_sum" % t].sum(axis=0), b["s%d_particles_sum" % t], rtol=1e-12)
        for c in "UVW":
            ia, ib = a["s%d_probe_idx_%s" % (t, c)], b["s%d_probe_idx_%s" % (t, c)]
            assert len(ib) >= 200000 and len(np.unique(ib)) == len(ib)
            common, pa, pb = np.intersect1d(ia, ib, return_indices=True)
            assert len(common) > 50
            assert np.array_equal(a["s%d_probe_val_%s" % (t, c)][pa], b["s%d_probe_val_%s" % (t, c)][pb])
            assert float(np.abs(b["s%d_probe_val_%s" % (t, c)]).max()) == float(b["s%d_maxabs_%s" % (t, c)])   # (the largest |u| IS among the probes)


STIFF = [("bunny64_nu3000", 64, ("sphere_large.ply", True), ["stanford_bunny.ply"]),
         ("honey96_nu1422", 96, None, ["rod.ply", "sheet.ply"])]


@pytest.mark.parametrize("name,N,boundary,liquids", STIFF)
def test_oracle_stiff_regime_goldens(oracle, name, N, boundary, liquids):
    """BASELINE config #4's stiffness regime (nu dt/dx^2 = 1.2e5 ... 1.3e5; make_golden.py H, I): the host library's scene is the
    reference's (count, checksums), and the oracle with its cap lifted lands on the reference's converged answer iteration for iteration
    and bit for bit at the probe faces -- on BOTH substeps, the second started from the reference's own particles (stored in the fixture)."""
    g = Golden(name)
    assert g.dims() == (N, N, N) and int(g["vcap"]) >= 100000
    stiff = float(g["nu"]) * g.dt / g.dx ** 2
    assert 1.2e5 <= stiff <= 1.35e5, stiff
    dx, solid, P = build_host_scene(N, boundary, liquids)
    assert len(P) == int(g["nparticles"])
    assert np.array_equal(P.astype(np.float64).sum(axis=0), g["particles0_sum"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"]
    s = oracle.OracleSim(N, N, N, dx)
    s.set_solid(solid)
    s.set_viscosity(float(g["nu"]))
    s.set_solver_limits(vmaxiter=int(g["vcap"]))
    s.particles = P
    for t in range(g.nsub):
        sec, vi, pi = s.substep(g.dt)
        assert vi["iterations"] == int(g["s%d_visc_iters" % t]) and vi["iterations"] > 700 and vi["status"] == 0
        assert pi["iterations"] == int(g["s%d_pres_iters" % t])
        for c in "UVW":
            a = s.grid(c).reshape(-1)
            assert np.array_equal(a[g["s%d_probe_idx_%s" % (t, c)]], g["s%d_probe_val_%s" % (t, c)])
            assert np.float32(np.abs(a).max()) == g["s%d_maxabs_%s" % (t, c)]
        assert np.array_equal(s.particles.astype(np.float64).sum(axis=0), g["s%d_particles_sum" % t])
        if t == 0:
            assert np.array_equal(s.particles, g["s0_particles"])
    s.close()


def test_oracle_late_state_golden_128(oracle):
    """bunny128_nu200_late (make_golden.py J, round 5): the reference carried config 3's scene at nu = 200 through 45 substeps at ITS defaults -- the bunny lies on the container
    wall --, the particles it then holds are the fixture's `state`; from them ONE substep with the viscosity tolerance at 1e-13 and the cap lifted (2 870 iterations; at its
    stock 1e-6 the reference is 4.1e-4 of max|u| away from that).  The oracle must reproduce it from the same state iteration for iteration and bit for bit at the 100 000
    probe faces per component (+ the 5 000 of largest |u|) and in the per-octant particle checksums.  tests/test_gpu_late_states.py holds the GPU against the same fixture."""
    if not os.path.exists(os.path.join(GOLDEN, "bunny128_nu200_late.npz")):
        pytest.skip("fixture not built")
    g = np.load(os.path.join(GOLDEN, "bunny128_nu200_late.npz"))
    N = int(g["I"])
    dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"]
    assert len(g["state"]) == len(P0) and np.float64(g["state"].astype(np.float64).sum()) == g["state_sum"]
    assert float(g["defaults_vs_converged"]) > 1e-4          # what the fixture is for: the reference's own defaults miss the bar in this state
    s = oracle.OracleSim(N, N, N, dx)
    s.set_solid(solid)
    s.set_viscosity(float(g["nu"]))
    s.set_solver_limits(vmaxiter=3000000, vtol=float(g["vtol"]))
    s.particles = g["state"]
    sec, vi, pi = s.substep(float(g["dt"]))
    assert vi["iterations"] == int(g["visc_iters"]) and vi["status"] == 0 and pi["iterations"] == int(g["pres_iters"])
    for c in "UVW":
        a = s.grid(c).reshape(-1)
        assert np.array_equal(a[g["probe_idx_" + c]], g["probe_val_" + c])
    Pn = s.particles
    oct_ = (Pn[:, 0] > 0.5).astype(int) + 2 * (Pn[:, 1] > 0.25).astype(int) + 4 * (Pn[:, 2] > 0.5).astype(int)
    sums = np.stack([Pn[oct_ == o].astype(np.float64).sum(axis=0) if (oct_ == o).any() else np.zeros(6) for o in range(8)])
    assert np.array_equal(sums, g["particles_octant_sum"])
    s.close()


@pytest.mark.parametrize("state", ["bunny256_nu5_sub20", "bunny256_nu5_sub25"])
def test_oracle_headline_late_state_256(oracle, state):
    """bunny256_nu5_sub25(_tol10) (make_golden.py K, round 6): the compiled reference carried BASELINE configs[2] itself -- the 256^3 bunny drop, nu = 5 -- through 25 of its own
    substeps (every carried viscosity solve ends at or near its cap of 700), and its particles are the fixture's state (tests/golden/_big/, 113 MB, not in git: sha256 in the fixture).
    The reference's converged answer from there (1e-10: 12 212 iterations) is hours of one core and is not re-run here; what IS re-run is the substep at the reference's DEFAULTS from that
    state: the oracle must take the same number of viscosity iterations (558) and end at the same residual (to the six digits the reference prints) -- the pin of oracle against reference ON this state (the GPU is held
    against the converged answer in tests/test_gpu_headline_late.py).  ~3 minutes of one core; skipped where the state file has not been generated."""
    import hashlib
    name = state if os.path.exists(os.path.join(GOLDEN, state + ".npz")) else state + "_tol10"      # (sub20: 20 substeps in, inside bench.py's timed window: 584 iterations at the defaults, 11 073 to 1e-10)
    spath = os.path.join(GOLDEN, "_big", state + "_state.npy")
    if not (os.path.exists(os.path.join(GOLDEN, name + ".npz")) and os.path.exists(spath)):
        pytest.skip("fixture or its state not present (make_golden.py carry256_nu5)")
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    S = np.load(spath)
    assert hashlib.sha256(np.ascontiguousarray(S).tobytes()).hexdigest() == str(g["state_sha256"])
    N = int(g["I"])
    dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"] and len(P0) == int(g["nparticles"])
    assert float(g["defaults_vs_converged"]) > 1e-4          # the reference at its own defaults misses the bar in this state
    s = oracle.OracleSim(N, N, N, dx)
    s.set_solid(solid)
    s.set_viscosity(float(g["nu"]))
    s.particles = S
    sec, vi, pi = s.substep(float(g["dt"]))
    assert vi["iterations"] == int(g["defaults_visc_iters"]), (vi, int(g["defaults_visc_iters"]))
    assert abs(vi["residual"] - float(g["defaults_visc_err"])) <= 5e-6 * float(g["defaults_visc_err"]), (vi["residual"], float(g["defaults_visc_err"]))   # (the harness reads the residual the reference PRINTS: six digits)
    s.close()
