import sys, os
sys.path.insert(0, "/root/repo/tests/research")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from jump_proto import load
i = int(sys.argv[1])
P = "/root/repo/scratch/jump/draw_%02d.vdump" % i
A, b, dgx, vol, table = load(P)
xr = np.fromfile(P + ".x", np.float64)
n = A.shape[0]; d = A.diagonal(); A = A.tocsr()
z = np.flatnonzero(d == 0)
print("n", n, "rows with zero diagonal:", len(z), " their b:", np.abs(b[z]).max() if len(z) else None, " x_ref there:", np.abs(xr[z]).max() if len(z) else None, "vol", vol[z].max() if len(z) else None)
absA = abs(A)
print("  max |offdiag| in those rows", absA[z].max() if len(z) else None, " max |A[:, z]|", absA[:, z].max() if len(z) else None)
keep = np.flatnonzero(d > 0)
A2 = A[keep][:, keep]; b2 = b[keep]; x2r = xr[keep]; d2 = d[keep]; vol2 = vol[keep]
D = 1 / np.sqrt(d2); S = (sp.diags(D) @ A2 @ sp.diags(D)).tocsc()
w, v = spl.eigsh(S, k=10, sigma=-1e-4, which="LM")
print("Jacobi-scaled eigenvalues nearest 0:", np.sort(w))
eps = 1e-12
y = spl.spsolve(S + eps * sp.eye(len(keep), format="csc"), D * b2); x = D * y
e = np.abs(x - x2r) / np.abs(x2r).max()
print("regularised direct solve: residual %.2e, diff to reference iterate %.3e; rows > 1e-4: %d, > 1e-6: %d" % (np.abs(b2 - A2 @ x).max() / np.abs(b2).max(), e.max(), (e > 1e-4).sum(), (e > 1e-6).sum()))
for r in np.argsort(e)[::-1][:8]:
    cols = A2.indices[A2.indptr[r]:A2.indptr[r+1]]; vals = A2.data[A2.indptr[r]:A2.indptr[r+1]]
    print(" row %d(%d) diff %.3e x_ref %.5f x %.5f diag %.3e vol %.2e | " % (r, keep[r], e[r], x2r[r], x[r], d2[r], vol2[r]), [(int(c), "%.2e" % (v / d2[r])) for c, v in zip(cols, vals) if c != r and v != 0])
# identical-row massless clusters: rows with vol == 0 whose off-diagonals are all +-diag
