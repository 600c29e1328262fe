import sys, os
import numpy as np, scipy.sparse as sp
sys.path.insert(0, "/root/repo/scratch/jump")
exec(open("/root/repo/scratch/jump/emu5_head.py").read())
m = (S.row < S.col) & (np.abs(S.data) >= 0.7)
rows, cols, s = S.row[m], S.col[m], S.data[m]
lam = np.maximum(1 - np.abs(s), 1e-9); sg = -np.sign(s)
Sc = S.tocsr()
onorm = []
for r_, c_, g in zip(rows, cols, sg):
    v = np.zeros(n); v[r_] = 1 / np.sqrt(2); v[c_] = g / np.sqrt(2)
    w = Sc @ v; w[r_] = 0; w[c_] = 0
    onorm.append(np.linalg.norm(w))
onorm = np.array(onorm)
order = np.argsort(lam)
print("pairs: lambda1 (block), |off-block part of S v| :")
print(" ".join("(%.1e,%.1e)" % (lam[i], onorm[i]) for i in order))
def make_sel(sel, mode="poly"):
    R, C, L, G = rows[sel], cols[sel], lam[sel], sg[sel]
    gain = ((1 - w0 * L) * (1 - w1 * L)) ** 2 / np.maximum(L, 1e-5)
    def Minv(r):
        z = smooth(r)
        t = 0.5 * gain * (r[R] / np.sqrt(d[R]) + G * r[C] / np.sqrt(d[C]))
        np.add.at(z, R, t / np.sqrt(d[R])); np.add.at(z, C, G * t / np.sqrt(d[C]))
        return z
    return Minv
print("smoother only:", pcg(smooth, 3000, 1e-6)[0])
for bound in (0.01, 0.03, 0.1, 0.3, 10.0):
    for lb in (0.03, 0.3):
        sel = (onorm <= bound) & (lam <= lb)
        print("off-block <= %.2f, lambda <= %.2f: %3d pairs -> %d iterations" % (bound, lb, sel.sum(), pcg(make_sel(sel), 3000, 1e-6)[0]))
# relative criterion: off-block <= k * lambda
for k in (0.3, 1.0, 3.0):
    sel = onorm <= k * lam
    print("off-block <= %.1f lambda: %3d pairs -> %d iterations" % (k, sel.sum(), pcg(make_sel(sel), 3000, 1e-6)[0]))
