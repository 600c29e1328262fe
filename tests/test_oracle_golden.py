"""CPU: the C restatement (oracle/flip_oracle.c) against the committed reference dumps, bit for bit.

The fixtures were produced by tests/golden/make_golden.py from the unmodified reference; this pins the
oracle wherever the repository travels (the reference sources themselves do not).
"""
import numpy as np
import pytest

from helpers import SCENES, Golden


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_reference_phase_by_phase(name, oracle):
    O = oracle
    g = Golden(name)
    I, J, K = g.dims()
    dx, dt = g.dx, g.dt
    solid, visc = g["solid"], g["viscosity"]
    layers = 7
    U = np.zeros((K, J, I + 1), np.float32)
    V = np.zeros((K, J + 1, I), np.float32)
    W = np.zeros((K + 1, J, I), np.float32)
    for t in range(g.nsub):
        P = g.particles_before(t)
        assert np.float32(O.cfl(I, J, K, dx, U, V, W)) == g["s%d_cfl" % t] or (
            np.isinf(g["s%d_cfl" % t]) and np.isinf(O.cfl(I, J, K, dx, U, V, W)))
        phi = O.particle_sdf(I, J, K, dx, P, solid)
        assert np.array_equal(phi, g["s%d_phi" % t])
        (U, V, W), val = O.p2g(I, J, K, dx, P, phi)
        U, V, W = [O.extrapolate_grid(a, m, layers) for a, m in zip((U, V, W), val)]
        for a, b in zip((U, V, W), g.uvw(t, "adv")):
            assert np.array_equal(a, b)
        for a, b in zip(val, g.valid(t, "adv")):
            assert np.array_equal(a, b)
        sU, sV, sW = U.copy(), V.copy(), W.copy()
        U, V, W = O.body_force(I, J, K, phi, U, V, W, g.gravity, dt)
        for a, b in zip((U, V, W), g.uvw(t, "force")):
            assert np.array_equal(a, b)
        (U, V, W), info = O.viscosity_solve(I, J, K, dx, dt, U, V, W, phi, solid, visc)
        if visc.max() > 0:
            assert info["iterations"] == int(g["s%d_visc_iters" % t])
            assert info["residual"] == pytest.approx(float(g["s%d_visc_err" % t]), rel=1e-5)
        for a, b in zip((U, V, W), g.uvw(t, "visc")):
            assert np.array_equal(a, b)
        w = O.compute_weights(I, J, K, solid)
        for a, c in zip(w, "UVW"):
            assert np.array_equal(a, g["s%d_weight_%s" % (t, c)])
        p, info = O.pressure_solve(I, J, K, dx, dt, U, V, W, *w, phi)
        if int(g["s%d_pres_iters" % t]) >= 0:
            assert info["iterations"] == int(g["s%d_pres_iters" % t])
        assert np.array_equal(p, g["s%d_pressure" % t])
        (U, V, W), val = O.apply_pressure(I, J, K, dx, dt, p, phi, *w, U, V, W)
        for a, b in zip((U, V, W), g.uvw(t, "proj")):
            assert np.array_equal(a, b)
        for a, b in zip(val, g.valid(t, "proj")):
            assert np.array_equal(a, b)
        U, V, W = [O.extrapolate_grid(a, m, layers) for a, m in zip((U, V, W), val)]
        U, V, W, sU, sV, sW = O.constrain(I, J, K, *w, U, V, W, sU, sV, sW)
        for a, b in zip((U, V, W), g.uvw(t, "final")):
            assert np.array_equal(a, b)
        for a, b in zip((sU, sV, sW), g.uvw(t, "saved")):
            assert np.array_equal(a, b)
        P2 = O.advect_particles(I, J, K, dx, dt, P, U, V, W, sU, sV, sW, solid)
        assert np.array_equal(P2, g["s%d_particles" % t])


@pytest.mark.parametrize("name", SCENES)
def test_oracle_sim_substeps_match_reference(name, oracle):
    """the whole-simulation wrapper (used as CPU baseline) lands on the same particles"""
    g = Golden(name)
    I, J, K = g.dims()
    s = oracle.OracleSim(I, J, K, g.dx)
    s.set_solid(g["solid"])
    s.set_viscosity(g["viscosity"])
    s.set_gravity(*g.gravity)
    s.particles = g["particles0"]
    for t in range(g.nsub):
        sec, vi, pi = s.substep(g.dt)
        assert np.array_equal(s.particles, g["s%d_particles" % t])
        for c, b in zip("UVW", g.uvw(t, "final")):
            assert np.array_equal(s.grid(c), b)
    s.close()


def test_level_set_fraction_known_answers(oracle):
    L = oracle.lib()
    # LevelsetUtils::fractionInside (levelsetutils.cpp:15-27)
    assert L.oracle_fraction_inside2(-1.0, -2.0) == 1.0
    assert L.oracle_fraction_inside2(1.0, 2.0) == 0.0
    assert L.oracle_fraction_inside2(-1.0, 1.0) == 0.5
    assert L.oracle_fraction_inside2(3.0, -1.0) == 0.25
    # marching squares (levelsetutils.cpp:38-119): half plane, corner, full, empty
    assert L.oracle_fraction_inside4(-1.0, -1.0, 1.0, 1.0) == 0.5
    assert L.oracle_fraction_inside4(-1.0, 1.0, 1.0, 1.0) == pytest.approx(0.125)
    assert L.oracle_fraction_inside4(-1.0, -1.0, -1.0, -1.0) == 1.0
    assert L.oracle_fraction_inside4(1.0, 1.0, 1.0, 1.0) == 0.0
    # cube volume fraction (levelsetutils.cpp:219-235): plane x = 0.5 cuts the cube in half
    import ctypes as C
    a = (C.c_float * 8)(-1, 1, -1, 1, -1, 1, -1, 1)
    assert L.oracle_volume_fraction8(a) == pytest.approx(0.5, abs=1e-6)
