"""research (round 6): the misses of the second holdout sweep -- where the error sits and what moves it (AFTER the one frozen run: profiles/r6/holdout2_sweep.log)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import holdout_sweep2 as H
from flipviscosity3d_amd.capi import Context
CACHE = os.path.join(ROOT, "tools", "holdout2_cache")


def run(i, reps=1, **prm):
    d = [x for x in H.draws() if x["id"] == i][0]
    z = np.load(os.path.join(CACHE, "draw_%02d.npz" % i))
    I, J, K, dx, solid, P = H.build_scene(d["scene"], d["N"])
    nu = H.viscosity_of(d["visc"], I, J, K, dx)
    for rep in range(reps):
        c = Context(I, J, K, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_gravity(*d["gravity"])
        if prm:
            c.set_params(**prm)
        c.particles = z["state"]
        st = c.substep(float(z["dt"]))
        den = float(z["den"]); worst = []; err = 0.0; nbad = 0
        for n in "UVW":
            a = c.grid(n).astype(np.float64)
            r = np.zeros(a.size); r[z["idx_" + n]] = z["val_" + n]; r = r.reshape(a.shape)
            e = np.abs(a - r) / den
            err = max(err, float(e.max())); nbad += int((e > 1e-4).sum())
            k, j, i_ = np.unravel_index(np.argmax(e), e.shape)
            worst.append("%s(%d,%d,%d) %.1e gpu %.4f ref %.4f vol %.2g" % (n, i_, j, k, e.max(), a[k, j, i_], r[k, j, i_], c.viscosity_volume(n)[k, j, i_]))
        c.close()
        v = st["viscosity"]
        print("draw %d %s: err %.2e (%d) its %d corr %d (%d) prec %d status %d elim %d res %.1e defect %.1e step %.1e | %s" % (
            i, prm, err, nbad, v["iterations"], v["correction_iterations"], v["correction_status"], v["preconditioner"], v["status"], v["eliminated_rows"], v["residual"], v["defect_residual"], v["velocity_step"], "; ".join(worst)), flush=True)


if __name__ == "__main__":
    from flipviscosity3d_amd import capi
    for i in [int(a) for a in sys.argv[1:]] or [7, 22, 25, 1, 3, 28, 36, 37, 39, 44]:
        run(i, viscosity_preconditioner=capi.PRECOND_MULTIGRID, viscosity_pair_correction=1, stall_guard_ratio=1000.0, reps=3 if i in (7, 22, 25) else 1)
        run(i, viscosity_preconditioner=capi.PRECOND_MULTIGRID, viscosity_pair_correction=1)
        run(i, viscosity_preconditioner=capi.PRECOND_MULTIGRID)
