import sys, os
sys.path.insert(0, "/root/repo/tests/research")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from jump_proto import load
i = int(sys.argv[1])
P = "/root/repo/scratch/jump/draw_%02d.vdump" % i
A, b, dgx, vol, table = load(P)
xr = np.fromfile(P + ".x", np.float64)
keep = np.flatnonzero(A.diagonal() > 0)
A = A[keep][:, keep].tocsr(); b = b[keep]; xr = xr[keep]; vol = vol[keep]
n = A.shape[0]; d = A.diagonal()
Dh = sp.diags(1 / np.sqrt(d)); S = (Dh @ A @ Dh).tocoo()
def make(theta, mode):
    m = (S.row < S.col) & (np.abs(S.data) >= theta)
    blocks = [np.array(p) for p in zip(S.row[m], S.col[m])]
    Cs = []
    for c in blocks:
        B = A[c][:, c].toarray(); B = (B + B.T) / 2
        if mode == "full":
            Cs.append(np.linalg.pinv(B, rcond=1e-9))
        else:
            # generalized weak mode: smallest eigenvalue of D^-1/2 B D^-1/2
            dh = 1 / np.sqrt(d[c]); Bs = B * np.outer(dh, dh)
            w, v = np.linalg.eigh(Bs)
            v1 = dh * v[:, 0]              # B-orthogonal direction in unscaled variables: v1^T B v1 = w0
            lam = max(w[0], 1e-5)
            if mode == "rank1": Cs.append(np.outer(v1, v1) / lam)
            if mode == "rank1m": Cs.append(np.outer(v1, v1) * (1 / lam - 1.0))
    def Minv(r):
        z = r / d
        for c, C in zip(blocks, Cs): z[c] += C @ r[c]
        return z
    return Minv, len(blocks)
def pcg(Minv, T, iters, tol):
    A_ = A.astype(T); b_ = b.astype(T)
    x = np.zeros(n, T); r = b_.copy(); z = Minv(r.astype(np.float64)).astype(T); p = z.copy()
    rz = np.dot(r.astype(np.float64), z.astype(np.float64)); bn = np.abs(b).max()
    for it in range(iters):
        q = (A_ @ p).astype(T)
        al = rz / np.dot(p.astype(np.float64), q.astype(np.float64))
        x = (x + T(al) * p).astype(T); r = (r - T(al) * q).astype(T)
        z = Minv(r.astype(np.float64)).astype(T)
        rz2 = np.dot(r.astype(np.float64), z.astype(np.float64))
        p = (z + T(rz2 / rz) * p).astype(T); rz = rz2
        if np.abs(r).max() <= tol * bn: break
    e = np.abs(x - xr) / np.abs(xr).max()
    return it + 1, np.sort(e)[-3:]
T = np.float64
print("plain Jacobi:", pcg(lambda r: r / d, T, 4000, 1e-6))
for theta in (0.7, 0.9):
    for mode in ("full", "rank1", "rank1m"):
        M, nb = make(theta, mode)
        print("theta %.2f %-7s %3d blocks:" % (theta, mode, nb), pcg(M, T, 4000, 1e-6))
