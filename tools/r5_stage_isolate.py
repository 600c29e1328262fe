"""research: a holdout draw that misses the bar -- which operator of the substep carries the difference to the oracle?  Every GPU operator's output against the oracle's
operator applied to THE GPU'S OWN inputs (viscosity and pressure solved to 1e-13), plus the chained end-of-substep difference.
    python tools/r5_stage_isolate.py <draw id> [--cache DIR] [k=v ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import holdout_sweep as H
from oracle import oraclebind as O
from flipviscosity3d_amd.capi import Context

did = int(sys.argv[1])
cache = sys.argv[sys.argv.index("--cache") + 1] if "--cache" in sys.argv else os.path.join(ROOT, "tools", "holdout_cache")
extra = {kv.split("=")[0]: float(kv.split("=")[1]) if "." in kv or "e" in kv.split("=")[1] else int(kv.split("=")[1]) for kv in sys.argv[2:] if "=" in kv}
d = [x for x in H.draws() if x["id"] == did][0]
z = np.load(os.path.join(cache, "draw_%02d.npz" % did))
I, J, K, dx, solid, P, g = H.build_scene(d["scene"], d["N"])
nu = H.viscosity_of(d["visc"], I, J, K, dx)
dt = float(np.float32(d["dt"]))
print(H.describe(d))
c = Context(I, J, K, dx)
c.set_solid_sdf(solid); c.set_viscosity(nu); c.set_gravity(*g)
if extra:
    c.set_params(**extra)
c.particles = z["state"]


def diff(name, A, B, den):
    e = [np.abs(a.astype(np.float64) - b.astype(np.float64)) / den for a, b in zip(A, B)]
    print("   %-44s %.2e  (entries > 1e-4: %d, > 1e-5: %d)" % (name, max(x.max() for x in e), sum(int((x > 1e-4).sum()) for x in e), sum(int((x > 1e-5).sum()) for x in e)), flush=True)


state = z["state"]
c.particle_sdf()
phi_g = c.grid("LIQUID_PHI")
phi_o = O.particle_sdf(I, J, K, dx, state, solid)
diff("liquid SDF", [phi_g], [phi_o], dx)
c.advect_velocity_field()
uvw_g = [c.grid(n) for n in "UVW"]
(po, vo) = O.p2g(I, J, K, dx, state, phi_g)
po = [O.extrapolate_grid(a, v, 10) for a, v in zip(po, vo)] if False else po
den = max(np.abs(a).max() for a in uvw_g)
val_g = [c.grid("VALID_" + n) for n in "UVW"]
print("   P2G valid masks equal: %s" % [bool(np.array_equal(a != 0, b != 0)) for a, b in zip(val_g, vo)])
diff("P2G on the valid faces", [a * (v != 0) for a, v in zip(uvw_g, vo)], [a * (v != 0) for a, v in zip(po, vo)], den)
c.body_force(dt)
pre = [c.grid(n) for n in "UVW"]
vi = c.viscosity_solve(dt)
post_g = [c.grid(n) for n in "UVW"]
numat = nu if isinstance(nu, np.ndarray) else np.full((K + 1, J + 1, I + 1), nu, np.float32)
post_o, io = O.viscosity_solve(I, J, K, dx, dt, pre[0], pre[1], pre[2], phi_g, solid, numat, tol=1e-13, maxiter=3000000, accept=1e30)
den = max(np.abs(a).max() for a in post_o)
print("   viscosity: GPU %d iterations (status %d, preconditioner %d), oracle on the GPU's input %d iterations" % (vi["iterations"], vi["status"], vi["preconditioner"], io["iterations"]))
diff("viscosity solve (same input)", post_g, post_o, den)
for n, a, b, p0 in zip("UVW", post_g, post_o, pre):
    x = np.abs(a.astype(np.float64) - b) / den
    idx = np.argwhere(x > 1e-5)
    if len(idx):
        print("   after viscosity, %s: %d entries > 1e-5 in k %d..%d j %d..%d i %d..%d" % (n, len(idx), idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max(), idx[:, 2].min(), idx[:, 2].max()))
        order = np.argsort(-x[tuple(idx.T)])[:6]
        for q in order:
            kk, jj, ii = idx[q]
            print("      (i %d, j %d, k %d): before %.6g GPU %.6g oracle %.6g   phi/dx at the two cells %.3g %.3g" % (ii, jj, kk, p0[kk, jj, ii], a[kk, jj, ii], b[kk, jj, ii],
                  phi_g[min(kk, K - 1), min(jj, J - 1), min(ii, I - 1)] / dx, phi_g[max(kk - (n == "W"), 0), max(jj - (n == "V"), 0), max(ii - (n == "U"), 0)] / dx))
c.compute_weights()
w_g = [c.grid("WEIGHT_" + n) for n in "UVW"] if False else None
pi = c.pressure_solve(dt)
p_g = c.grid("PRESSURE")
wo = O.compute_weights(I, J, K, solid)
p_o, ipo = O.pressure_solve(I, J, K, dx, dt, post_g[0], post_g[1], post_g[2], wo[0], wo[1], wo[2], phi_g, tol=1e-13, maxiter=100000)
diff("pressure (same input; relative to max|p|)", [p_g], [p_o], max(np.abs(p_o).max(), 1e-300))
c.apply_pressure(dt)
ap_g = [c.grid(n) for n in "UVW"]
(ap_o, apv) = O.apply_pressure(I, J, K, dx, dt, p_g, phi_g, wo[0], wo[1], wo[2], post_g[0], post_g[1], post_g[2])
den = max(np.abs(a).max() for a in ap_o)
diff("pressure gradient (same pressure), valid faces", [a * (v != 0) for a, v in zip(ap_g, apv)], [a * (v != 0) for a, v in zip(ap_o, apv)], den)
c.extrapolate(); c.constrain()
fin = [c.grid(n) for n in "UVW"]
ref = []
for n, a in zip("UVW", fin):
    r = np.zeros(a.size, np.float32); r[z["idx_" + n]] = z["val_" + n]; ref.append(r.reshape(a.shape))
diff("END OF SUBSTEP against the cached oracle", fin, ref, float(z["den"]))
# where the end-of-substep difference sits
e = [np.abs(a.astype(np.float64) - b) / float(z["den"]) for a, b in zip(fin, ref)]
for n, x, a, b in zip("UVW", e, fin, ref):
    idx = np.argwhere(x > 1e-4)
    if len(idx):
        kk, jj, ii = idx[np.argmax(x[tuple(idx.T)])]
        print("   %s: %d faces > 1e-4, box k %d..%d j %d..%d i %d..%d; worst at (i %d, j %d, k %d): GPU %.6g oracle %.6g, phi there %.3g dx" % (
            n, len(idx), idx[:, 0].min(), idx[:, 0].max(), idx[:, 1].min(), idx[:, 1].max(), idx[:, 2].min(), idx[:, 2].max(), ii, jj, kk, a[kk, jj, ii], b[kk, jj, ii],
            phi_g[min(kk, K - 1), min(jj, J - 1), min(ii, I - 1)] / dx))

# ---- the two chains side by side: where does the end-of-substep difference first appear?
print("cumulative: the GPU's chain against the oracle's chain (its own inputs throughout)")
(uo, vo2) = O.p2g(I, J, K, dx, state, phi_o)
uo = [O.extrapolate_grid(a, v, 7) for a, v in zip(uo, vo2)]
saved_o = [a.copy() for a in uo]
den0 = max(np.abs(a).max() for a in uo)
diff("P2G + extrapolation, every face", uvw_g, uo, den0)
uo = list(O.body_force(I, J, K, phi_o, uo[0], uo[1], uo[2], g, dt))
diff("... + body force, every face", pre, uo, den0)
post_oo, _ = O.viscosity_solve(I, J, K, dx, dt, uo[0], uo[1], uo[2], phi_o, solid, numat, tol=1e-13, maxiter=3000000, accept=1e30)
p_oo, _ = O.pressure_solve(I, J, K, dx, dt, post_oo[0], post_oo[1], post_oo[2], wo[0], wo[1], wo[2], phi_o, tol=1e-13, maxiter=100000)
(ap_oo, apv_oo) = O.apply_pressure(I, J, K, dx, dt, p_oo, phi_o, wo[0], wo[1], wo[2], post_oo[0], post_oo[1], post_oo[2])
denf = float(z["den"])
diff("after viscosity, on the faces valid after the projection", [a * (v != 0) for a, v in zip(post_g, apv_oo)], [a * (v != 0) for a, v in zip(post_oo, apv_oo)], denf)
vols = O.viscosity_volumes(I, J, K, dx, phi_o)
for vn, ref_v in vols.items():
    gv = c.viscosity_volume(vn)
    bad = np.argwhere(gv != ref_v)
    print("   control volume %-7s GPU == oracle: %s%s" % (vn, len(bad) == 0, "" if len(bad) == 0 else "  (%d entries differ, max %.3g, first at k j i %s: GPU %.9g oracle %.9g)" % (len(bad), np.abs(gv - ref_v).max(), bad[0], gv[tuple(bad[0])], ref_v[tuple(bad[0])])))
for n, a, b, p0, v in zip("UVW", post_g, post_oo, pre, apv_oo):
    x = np.abs(a.astype(np.float64) - b) / denf * (v != 0)
    for kk, jj, ii in np.argwhere(x > 1e-5):
        lo = (kk - (n == "W"), jj - (n == "V"), ii - (n == "U"))
        print("   %s (i %d, j %d, k %d): before %.7g | GPU after %.7g | oracle after %.7g; own volume %.3g; phi/dx of its two cells %.4g %.4g" % (
            n, ii, jj, kk, p0[kk, jj, ii], a[kk, jj, ii], b[kk, jj, ii], vols[n][kk, jj, ii], phi_o[min(kk, K - 1), min(jj, J - 1), min(ii, I - 1)] / dx, phi_o[max(lo[0], 0), max(lo[1], 0), max(lo[2], 0)] / dx))
        sl = (slice(max(kk - 1, 0), kk + 2), slice(max(jj - 1, 0), jj + 2), slice(max(ii - 1, 0), ii + 2))
        print("      volumes around it:", {q: float(np.abs(vols[q][sl]).max()) for q in ("center", "U", "V", "W", "edgeU", "edgeV", "edgeW")})
diff("pressure (relative to max|p|)", [p_g], [p_oo], max(np.abs(p_oo).max(), 1e-300))
val2_g = [c2 for c2 in apv]
print("   valid masks after the projection equal (oracle on GPU input vs oracle chain): %s" % [bool(np.array_equal(a != 0, b != 0)) for a, b in zip(apv, apv_oo)])
diff("after the pressure gradient, valid faces", [a * (v != 0) for a, v in zip(ap_g, apv_oo)], [a * (v != 0) for a, v in zip(ap_oo, apv_oo)], denf)
ex_oo = [O.extrapolate_grid(a, v, 7) for a, v in zip(ap_oo, apv_oo)]
ex_g_by_oracle = [O.extrapolate_grid(a, v, 7) for a, v in zip(ap_g, apv_oo)]
diff("oracle extrapolation of the GPU's projected field vs the oracle chain's", ex_g_by_oracle, ex_oo, denf)
fin_oo = O.constrain(I, J, K, wo[0], wo[1], wo[2], ex_oo[0], ex_oo[1], ex_oo[2], saved_o[0], saved_o[1], saved_o[2])[:3]
diff("end of substep: GPU vs the oracle chain", fin, fin_oo, denf)
fin_mix = O.constrain(I, J, K, wo[0], wo[1], wo[2], ex_g_by_oracle[0], ex_g_by_oracle[1], ex_g_by_oracle[2], saved_o[0], saved_o[1], saved_o[2])[:3]
diff("end of substep: GPU's projected field + ORACLE extrapolation and constrain vs the GPU's own", fin, fin_mix, denf)
c.close()
