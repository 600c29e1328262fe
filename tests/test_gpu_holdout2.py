"""GPU (-m gpu): the SECOND holdout sweep (tools/holdout_sweep2.py) as a test -- its FIRST 12 draw ids, a list fixed before anything was run (the sweep's
order is a seeded permutation, so the twelve hold viscosity fields and uniform viscosities, all five scenes and all three sizes).  Seed 6602026; every scene has an INTERIOR solid
(FluidSimulation::addBoundary(mesh, false), fluidsimulation.cpp:45-58; the union of meshlevelset.cpp:152-184), gravity off-axis, dt from the CFL split of FluidSimulation::advance,
N in {48, 80, 112}, viscosity log-uniform in [1e-3, 5e3] or a field with a jump 1e-2 | 50 / 1 | 1e4 or smooth, start states after 10 / 60 oracle substeps.

The sweep was run ONCE against the library of commit 1a3c98c with no parameter set: profiles/r6/holdout2_sweep.log -- 40 of 48 within 1e-4.  The test asserts what THAT run
delivered: the nine of the twelve that were within the bar must be; the three that were not are listed with what is known about them (profiles/r6/holdout2_misses.log)
and must not get worse than the frozen run by more than its run-to-run spread.

Fixtures: tests/golden/holdout2/draw_NN.npz (the tool's `prepare` output; for the three large draws a compact form -- 60 000 probe faces per component incl. the 2 000 of largest
|u|, and the start state regenerated here by carrying the oracle through the stored CFL time steps, checked by its sha256)."""
import hashlib
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu
HOLD = os.path.join(GOLDEN, "holdout2")
FIRST = list(range(12))
# the frozen run's misses among them: (error of that run, what is known)
MISSES = {0: (4.6e-1, "honey + sphere 48^3, nu = 1214 at dt = 1/30 (nu dt/dx^2 = 93 000), 10 substeps in: the multigrid loops end far from their targets (status 1 -- the solve says so); "
                      "the reference at its own defaults is 0.98 off as well (525 of the 2 408 iterations its converged answer takes)"),
          7: (9.4e-3, "honey + sphere 80^3, nu = 3e-3 (nu dt/dx^2 = 0.3: the diagonal loop) after 60 substeps: the residual passes at 4.7e-7 max|rhs| while CG still moves massless faces of "
                      "the splashed sheet by 2e-3 max|u| per window; the velocity criterion's patience of 48 iterations runs out (flipv_solve_info.velocity_step says so; status 0); 9e-3 ... 4e-2 "
                      "on ~270 faces from run to run; the reference at its defaults: 1.1e-1"),
          9: (1.09e-4, "honey + sphere 80^3, nu = 91 at dt = 1/30 (nu dt/dx^2 = 19 459): 1.09e-4 on 630 faces -- 9 % over the bar; with stage 1 to 1e-6 8.9e-5")}


def draw_and_fixture(i):
    import holdout_sweep2 as H
    d = [x for x in H.draws() if x["id"] == i][0]
    z = dict(np.load(os.path.join(HOLD, "draw_%02d.npz" % i)))
    if "state" not in z:      # a compact fixture: the state = the oracle carried through the stored time steps from the scene's seeding (deterministic, bit-pinned to the reference)
        big = os.path.join(GOLDEN, "_big", "holdout2_draw_%02d_state.npy" % i)      # (git-ignored like the 256^3 states: written by the tool's machine, regenerated where missing)
        cache = os.path.join(ROOT, "tools", "holdout2_cache", "draw_%02d.npz" % i)
        if os.path.exists(big):
            z["state"] = np.load(big)
        elif os.path.exists(cache):
            z["state"] = np.load(cache)["state"]
        else:
            from oracle import oraclebind as O
            I, J, K, dx, solid, P = H.build_scene(d["scene"], d["N"])
            o = O.OracleSim(I, J, K, dx)
            o.set_solid(solid); o.set_viscosity(H.viscosity_of(d["visc"], I, J, K, dx)); o.set_gravity(*d["gravity"])
            o.particles = P
            for dt in z["dts"]:
                o.substep(float(dt))
            z["state"] = o.particles.copy()
            o.close()
        assert hashlib.sha256(np.ascontiguousarray(z["state"]).tobytes()).hexdigest() == str(z["state_sha"])
        if not os.path.exists(big):
            os.makedirs(os.path.dirname(big), exist_ok=True)
            np.save(big, z["state"])
    return H, d, z


@pytest.mark.parametrize("i", FIRST)
def test_second_holdout_first_twelve_draws_default_parameters(i):
    H, d, z = draw_and_fixture(i)
    err, nbad, st = H.run_draw(d, z)
    v = st["viscosity"]
    print("%s dt %.5f | GPU %.2e (%d faces > 1e-4), %d viscosity iterations, status %d, %d rows eliminated, velocity step %.1e | the reference at its defaults %.2e" % (
        H.describe(d), float(z["dt"]), err, nbad, v["iterations"], v["status"], v["eliminated_rows"], v["velocity_step"], float(z["err_ref_defaults"])))
    if i in MISSES:
        frozen, why = MISSES[i]
        print("   a miss of the frozen run (%.2e): %s" % (frozen, why))
        assert err <= max(5.0 * frozen, 2e-4), (err, frozen)      # (not worse than that run beyond its run-to-run spread; if it comes within the bar, move it)
        return
    assert v["status"] in (0, 3) and st["pressure"]["status"] in (0, 3), st
    assert err <= 1e-4, err


def test_interior_solid_is_part_of_every_scene():
    """what the first sweep lacked: every scene of this one has solid nodes strictly inside the box walls, and liquid that meets them within the carried substeps"""
    import holdout_sweep2 as H
    for sc in H.SCENES:
        I, J, K, dx, solid, P = H.build_scene(sc, 48)
        assert (solid[4:-4, 4:-4, 4:-4] < 0).sum() > 100, sc
