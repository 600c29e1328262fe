"""research (round 6): the holdout draws the default solve misses (9, 11): error against the committed converged reference, where it sits, and what the solve reports"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import holdout_sweep as H
from flipviscosity3d_amd.capi import Context
HOLD = os.path.join(ROOT, "tests", "golden", "holdout")


def run(i, reps=1, **prm):
    d = [x for x in H.draws() if x["id"] == i][0]
    z = np.load(os.path.join(HOLD, "draw_%02d.npz" % i))
    I, J, K, dx, solid, P, g = H.build_scene(d["scene"], d["N"])
    nu = H.viscosity_of(d["visc"], I, J, K, dx)
    for rep in range(reps):
        c = Context(I, J, K, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_gravity(*g)
        if prm:
            c.set_params(**prm)
        c.particles = z["state"]
        st = c.substep(float(np.float32(d["dt"])))
        den = float(z["den"])
        worst = []
        err = 0.0; nbad = 0
        for n in "UVW":
            a = c.grid(n).astype(np.float64)
            r = np.zeros(a.size); r[z["idx_" + n]] = z["val_" + n]; r = r.reshape(a.shape)
            e = np.abs(a - r) / den
            err = max(err, float(e.max())); nbad += int((e > 1e-4).sum())
            k, j, i_ = np.unravel_index(np.argmax(e), e.shape)
            worst.append("%s(%d,%d,%d) %.2e gpu %.5f ref %.5f vol %.3g" % (n, i_, j, k, e.max(), a[k, j, i_], r[k, j, i_], c.viscosity_volume(n)[k, j, i_]))
        c.close()
        v = st["viscosity"]
        print("draw %d %s rep %d: err %.2e (%d faces > 1e-4) its %d corr %d (status %d) status %d residual %.2e defect %.2e step %.1e | %s" % (
            i, prm, rep, err, nbad, v["iterations"], v["correction_iterations"], v["correction_status"], v["status"], v["residual"], v["defect_residual"], v["velocity_step"], "; ".join(worst)), flush=True)


if __name__ == "__main__":
    run(9, reps=3)
    run(9, verbose=1)
    run(9, viscosity_massless_polish=-1)
    run(11, reps=3)
    run(11, verbose=1)
    run(11, viscosity_stage2_rounds=3)
    run(11, precision=1)
    run(11, viscosity_preconditioner=1, viscosity_max_iterations=20000)
