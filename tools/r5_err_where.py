"""research: WHERE the default solve's error sits in a late state of the 256^3 bench scene (against the tightened run of the same state)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy import ndimage
from bench import build_workload
from flipviscosity3d_amd.capi import Context
N = int(sys.argv[1]); at = int(sys.argv[2])
I, J, K, dx, solid, P = build_workload("bunny", N, on_device=True)
c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0); c.particles = P
for t in range(at): c.substep(0.01)
S = c.particles.copy(); c.close()
def run(**kw):
    c = Context(I, J, K, dx); c.set_solid_sdf(solid); c.set_viscosity(5.0)
    if kw: c.set_params(**kw)
    c.particles = S
    st = c.substep(0.01)
    out = dict(g=[c.grid(n) for n in "UVW"], vol=[c.viscosity_volume(n) for n in "UVW"], phi=c.grid("LIQUID_PHI"), valid=[c.grid("VALID_" + n) for n in "UVW"], st=st)
    c.close(); return out
T = run(precision=1, viscosity_tolerance=1e-9, pressure_rel_tolerance=1e-9, viscosity_max_iterations=5000, viscosity_velocity_tolerance=1e-6)
D = run()
den = max(np.abs(a).max() for a in T["g"])
liq = T["phi"] < 0
lab, nl = ndimage.label(liq)
sizes = ndimage.sum(liq, lab, range(1, nl + 1)).astype(int)
order = np.argsort(sizes)[::-1]
print("liquid cells %d in %d components; largest: %s" % (liq.sum(), nl, sizes[order[:8]]))
for n, a, b, vol in zip("UVW", D["g"], T["g"], D["vol"]):
    e = np.abs(a.astype(np.float64) - b) / den
    bad = e > 1e-4
    print("%s: max err %.2e; faces > 1e-4: %d; of those with own volume 0: %d, < 0.1: %d, >= 0.5: %d" % (n, e.max(), bad.sum(), (bad & (vol == 0)).sum(), (bad & (vol < 0.1)).sum(), (bad & (vol >= 0.5)).sum()))
    # which liquid component do the bad faces touch (cell at the face's own index, else the one before it along the normal)
    ax = {"U": 2, "V": 1, "W": 0}[n]
    sl = [slice(0, K), slice(0, J), slice(0, I)]
    eb = bad[tuple(sl)]
    labs = lab[eb]
    cnt = np.bincount(labs, minlength=nl + 1)
    top = np.argsort(cnt)[::-1][:6]
    print("   bad faces by liquid component of the cell at the face index (0 = not liquid): " + ", ".join("comp %d (size %d): %d" % (t, sizes[t - 1] if t else 0, cnt[t]) for t in top if cnt[t]))
    bl, nb = ndimage.label(bad)
    bs = ndimage.sum(bad, bl, range(1, nb + 1)).astype(int)
    print("   connected groups of bad faces: %d, largest %s" % (nb, np.sort(bs)[::-1][:8]))
    k, j, i = np.unravel_index(np.argmax(e), e.shape)
    print("   worst face (%d,%d,%d): default %.5f tight %.5f vol %.3f phi %.3f solid %.3f" % (i, j, k, a[k, j, i], b[k, j, i], vol[k, j, i], T["phi"][min(k, K - 1), min(j, J - 1), min(i, I - 1)], solid[k, j, i]))
