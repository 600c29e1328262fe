import sys, os
sys.path.insert(0, "/root/repo/tests/research")
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from jump_proto import load
i = int(sys.argv[1])
A, b, dgx, vol, table = load("/root/repo/scratch/jump/draw_%02d.vdump" % i)
n = A.shape[0]
d = A.diagonal()
A = A.tocsr()
print("smallest diagonals:")
for r in np.argsort(d)[:12]:
    cols = A.indices[A.indptr[r]:A.indptr[r+1]]; vals = A.data[A.indptr[r]:A.indptr[r+1]]
    print(" row %d diag %.3e vol %.3e b %.3e  offdiag:" % (r, d[r], vol[r], b[r]), [(int(c), "%.3e" % v, "d=%.2e" % d[c]) for c, v in zip(cols, vals) if c != r])
# scale spread
print("b max", np.abs(b).max(), "median diag", np.median(d))
# connected components of the graph
nc, lab = sp.csgraph.connected_components(A, directed=False)
sizes = np.bincount(lab)
print("components:", nc, "sizes (sorted)", np.sort(sizes)[::-1][:15])
# for each small component: is its block singular?
for c in np.argsort(sizes)[:min(nc-1, 20)]:
    rows = np.flatnonzero(lab == c)
    B = A[rows][:, rows].toarray()
    w = np.linalg.eigvalsh((B + B.T) / 2)
    print(" comp %d rows %s eig %s  b %s  vol %s" % (c, rows, w, b[rows], vol[rows]))
