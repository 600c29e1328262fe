import sys, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
def loadany(path):
    with open(path, "rb") as f:
        n, cap, dim, ext = np.fromfile(f, np.int64, 4)
        cnt = np.fromfile(f, np.int32, n)
        col = np.fromfile(f, np.uint32, n * cap).reshape(n, cap)
        val = np.fromfile(f, np.float64, n * cap).reshape(n, cap)
        rhs = np.fromfile(f, np.float64, n)
    m = np.arange(cap)[None, :] < cnt[:, None]
    rows = np.repeat(np.arange(n), cnt)
    return sp.csr_matrix((val[m], (rows, col[m].astype(np.int64))), shape=(n, n)), rhs
A, b = loadany(sys.argv[1])
keep = np.flatnonzero(A.diagonal() > 0)
A = A[keep][:, keep].tocsr(); b = b[keep]
A = ((A + A.T) / 2).tocsr()
n = A.shape[0]; d = A.diagonal()
Dh = sp.diags(1 / np.sqrt(d)); S = (Dh @ A @ Dh).tocoo()
w0, w1 = 1.317, 0.382
def smooth(r):   # V(2,2) without a coarse grid: four Chebyshev-weighted Jacobi sweeps from zero (pre w0, w1; post w0, w1)
    z = w0 * r / d
    for w in (w1, w0, w1):
        z = z + w * (r - A @ z) / d
    return z
def make(theta, mode):
    m = (S.row < S.col) & (np.abs(S.data) >= theta)
    rows, cols, s = S.row[m], S.col[m], S.data[m]
    lam = np.maximum(1 - np.abs(s), 1e-5)
    if mode == "inv": gain = 1 / lam
    elif mode == "poly": gain = ((1 - w0 * lam) * (1 - w1 * lam)) ** 2 / lam
    sg = -np.sign(s)
    def Minv(r):
        z = smooth(r)
        t = 0.5 * gain * (r[rows] / np.sqrt(d[rows]) + sg * r[cols] / np.sqrt(d[cols]))     # v^T r / lambda, v = D^-1/2 (1, sg)/sqrt 2
        np.add.at(z, rows, t / np.sqrt(d[rows])); np.add.at(z, cols, sg * t / np.sqrt(d[cols]))
        return z
    return Minv, len(rows)
def pcg(Minv, iters, tol):
    x = np.zeros(n); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; bn = np.abs(b).max(); h = []
    for it in range(iters):
        q = A @ p; al = rz / (p @ q); x += al * p; r -= al * q
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
        h.append(np.abs(r).max() / bn)
        if h[-1] <= tol: break
    return it + 1, ["%.1e" % v for v in h[9::10]][:8]
print("smoother only:", pcg(smooth, 3000, 1e-6))
for theta in (0.7, 0.9, 0.97):
    for mode in ("inv", "poly"):
        M, nb = make(theta, mode)
        print("theta %.2f %-5s %3d pairs:" % (theta, mode, nb), pcg(M, 3000, 1e-6))
