"""CPU (build container or any box that carries oracle/_ref): the C restatement against the compiled
reference, live, on a scene that is NOT among the committed fixtures -- different size, anisotropic grid
count, shifted liquid -- so the pin does not depend on the fixtures alone."""
import os

import numpy as np
import pytest

from helpers import GOLDEN

MESH = os.path.join(GOLDEN, "meshes")


def test_oracle_matches_reference_live(oracle):
    from oracle import refbind as R
    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    from flipviscosity3d_amd.plyio import load_ply
    O = oracle
    I, J, K = 26, 22, 30
    dx = float(np.float32(1.0 / 30))
    r = R.RefSim(I, J, K, dx)
    v, t = load_ply(os.path.join(MESH, "cone.ply"))
    R.lib().ref_srand(12345)
    r.add_liquid(v * np.float32(0.8) + np.array([0.05, 0.02, 0.1], np.float32), t)
    r.set_viscosity(2.5)
    r.set_gravity(0.5, -9.81, -0.25)
    s = O.OracleSim(I, J, K, dx)
    s.set_solid(r.grid("SOLID_PHI"))
    s.set_viscosity(2.5)
    s.set_gravity(0.5, -9.81, -0.25)
    s.particles = r.particles
    assert len(r.particles) > 1000
    for step in range(3):
        r.substep(0.008)
        _, vi, pi = s.substep(0.008)
        st = r.solver_stats()
        assert vi["iterations"] == st["visc_iters"] and pi["iterations"] == st["pres_iters"]
        for name in ("U", "V", "W", "SAVED_U", "LIQUID_PHI", "PRESSURE", "WEIGHT_V", "VALID_W"):
            assert np.array_equal(s.grid(name), r.grid(name)), (step, name)
        assert np.array_equal(s.particles, r.particles)
    # the CFL loop of advance(): same number of substeps, same end state
    n_ref = r.advance(0.05)
    n_or = s.advance(0.05)
    assert n_ref == n_or
    assert np.array_equal(s.particles, r.particles)
    r.close()
    s.close()


def test_fraction_helpers_match_reference_on_random_inputs(oracle):
    from oracle import refbind as R
    if not R.available():
        pytest.skip("oracle/_ref not built")
    import ctypes as C
    rng = np.random.default_rng(0)
    L, Q = oracle.lib(), R.lib()
    vals = rng.uniform(-1, 1, size=(4000, 8)).astype(np.float32)
    vals[::7, 3] = 0.0
    vals[::11] = np.abs(vals[::11])
    for row in vals:
        a, b, c, d = [float(x) for x in row[:4]]
        assert L.oracle_fraction_inside2(a, b) == Q.ref_fraction_inside2(a, b)
        assert L.oracle_fraction_inside4(a, b, c, d) == Q.ref_fraction_inside4(a, b, c, d)
        arr = (C.c_float * 8)(*[float(x) for x in row])
        assert L.oracle_volume_fraction8(arr) == Q.ref_volume_fraction8(arr)
