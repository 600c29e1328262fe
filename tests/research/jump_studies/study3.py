import sys, os
sys.path.insert(0, "/root/repo/tests/research")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from jump_proto import load
i = int(sys.argv[1])
P = "/root/repo/scratch/jump/draw_%02d.vdump" % i
A, b, dgx, vol, table = load(P)
xr = np.fromfile(P + ".x", np.float64)
n = A.shape[0]; d = A.diagonal(); A = A.tocsr()
print("n", n, "reference iterate: residual", np.abs(b - A @ xr).max() / np.abs(b).max(), "max|x|", np.abs(xr).max())
# Jacobi-scaled regularised direct solve: (S + eps I) y = D b
D = 1 / np.sqrt(d); S = (sp.diags(D) @ A @ sp.diags(D)).tocsc()
for eps in (1e-9, 1e-11):
    y = spl.spsolve(S + eps * sp.eye(n, format="csc"), D * b)
    x = D * y
    e = np.abs(x - xr) / np.abs(xr).max()
    print("eps %.0e: residual %.2e  max diff to the reference iterate %.3e, rows beyond 1e-4: %d, 1e-6: %d" % (eps, np.abs(b - A @ x).max() / np.abs(b).max(), e.max(), (e > 1e-4).sum(), (e > 1e-6).sum()))
bad = np.argsort(e)[::-1][:10]
for r in bad:
    cols = A.indices[A.indptr[r]:A.indptr[r+1]]; vals = A.data[A.indptr[r]:A.indptr[r+1]]
    print(" row %d diff %.3e x_ref %.4f x %.4f diag %.3e vol %.2e | " % (r, e[r], xr[r], x[r], d[r], vol[r]), [(int(c), "%.2e" % (v / d[r])) for c, v in zip(cols, vals) if c != r and v != 0])
