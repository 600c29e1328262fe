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
def pcg(Minv, iters, tol):
    x = np.zeros(n); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; bn = np.abs(b).max(); h = []
    for it in range(iters):
        q = A @ p; al = rz / (p @ q); x += al * p; r -= al * q
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
        h.append(np.abs(r).max() / bn)
        if h[-1] <= tol: break
    return it + 1, ["%.1e" % v for v in h[9::10]][:8]
