"""GPU (-m gpu): a subset of the holdout sweep (tools/holdout_sweep.py) as a test -- 13 draws of its fixed seed with their states and the oracle's
converged answers (1e-13 / 1e-13) committed under tests/golden/holdout/ (the tool's `prepare` step wrote them; the draws themselves are regenerated from the seed).
Scenes, sizes (non-cubic included), time steps, viscosities (5 % either side of the rule's thresholds among them) and viscosity FIELDS that none of the scans behind the
rule's constants used.  The GPU runs with NO field of flipv_params set; bar: <= 1e-4 relative max-norm on EVERY face.
The full sweep (47 draws): profiles/r5/holdout_sweep.log.  Its failures are fixtures too -- see the second test."""
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu
HOLD = os.path.join(GOLDEN, "holdout")
PASSING = [0, 6, 7, 8, 13, 15, 21, 23, 26, 29, 36, 39, 40]
# draws the default solve does NOT bring within 1e-4 (profiles/r5/holdout_sweep.log), kept as fixtures with what is known about them
KNOWN = {9: "honey 40^3, viscosity 1e-4 | 200 (contrast 2e6) after 40 substeps: the reference's operator is indefinite on the sliver rows at the jump (own volume + rounding "
            "defect < 0); the fp32 correction stage stalls at 1e-3 max|rhs| and the solve SAYS so (status 1) -- the reference's own MIC(0) PCG gets through in 310 iterations",
         11: "honey 40^3, viscosity 0 | 3 000 after 5 substeps: 600+ iterations, within 1e-4 ... 8e-4 on a few faces from run to run (the correction stages end short: status 1)"}


def run_draw(i):
    import holdout_sweep as H
    from flipviscosity3d_amd.capi import Context
    d = [x for x in H.draws() if x["id"] == i][0]
    z = np.load(os.path.join(HOLD, "draw_%02d.npz" % i))
    I, J, K, dx, solid, P, g = H.build_scene(d["scene"], d["N"])
    nu = H.viscosity_of(d["visc"], I, J, K, dx)
    c = Context(I, J, K, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_gravity(*g)
    c.particles = z["state"]
    st = c.substep(float(np.float32(d["dt"])))
    den = float(z["den"])
    err, nbad = 0.0, 0
    for n in "UVW":
        a = c.grid(n).reshape(-1).astype(np.float64)
        r = np.zeros_like(a)
        r[z["idx_" + n]] = z["val_" + n]
        e = np.abs(a - r) / den
        err = max(err, float(e.max())); nbad += int((e > 1e-4).sum())
    c.close()
    print("%s | GPU %.2e (%d faces > 1e-4), %d viscosity iterations, status %d | the reference at its defaults %.2e" % (
        H.describe(d), err, nbad, st["viscosity"]["iterations"], st["viscosity"]["status"], float(z["err_ref_defaults"])))
    return err, nbad, st


@pytest.mark.parametrize("i", PASSING)
def test_holdout_draw_default_parameters(i):
    err, nbad, st = run_draw(i)
    assert st["viscosity"]["status"] in (0, 3) and st["pressure"]["status"] in (0, 3), st
    assert err <= 1e-4, err


@pytest.mark.parametrize("i", sorted(KNOWN))
def test_holdout_known_failures_say_so(i):
    """the draws of the sweep that miss the bar: either they have come within it, or the solve reports that it did not converge (status 1) -- never a silent miss"""
    err, nbad, st = run_draw(i)
    print("known:", KNOWN[i])
    assert err <= 1e-4 or st["viscosity"]["status"] == 1, (err, st["viscosity"])
