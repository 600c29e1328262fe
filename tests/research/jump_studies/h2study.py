import sys
sys.path.insert(0, "/root/repo/tests/research")
import numpy as np, scipy.sparse as sp
from jump_proto import load
i = int(sys.argv[1])
P = "/root/repo/scratch/jump/h2_%02d.vdump" % i
A, b, dgx, vol, table = load(P)
x = np.fromfile(P + ".x", np.float64)
n = A.shape[0]; A = A.tocsr(); d = A.diagonal()
print("n", n, "zero-diagonal rows", (d == 0).sum(), "max|x|", np.abs(x).max(), "residual", np.abs(b - A @ x).max() / np.abs(b).max())
# rows that repeat another row's equation: massless, a single off-diagonal magnitude == diag on all entries
S = A.copy(); S.setdiag(0); S.eliminate_zeros()
cand = []
for r in np.flatnonzero((vol == 0) & (d > 0)):
    v = S.data[S.indptr[r]:S.indptr[r+1]]
    if len(v) and np.allclose(np.abs(v), d[r], rtol=1e-6): cand.append(r)
cand = np.array(cand)
print("single-factor massless rows:", len(cand))
cs = set(cand.tolist())
groups = {}
for r in cand:
    cols = S.indices[S.indptr[r]:S.indptr[r+1]]
    key = tuple(sorted([r] + cols.tolist()))
    groups.setdefault(key, []).append(r)
bad = 0
for key, rows in groups.items():
    if len(rows) < 2: continue
    rows = sorted(rows)
    print(" cluster rows %s: x_ref %s  b %s diag %s" % (rows, ["%.5f" % x[r] for r in rows], ["%.2e" % b[r] for r in rows], ["%.3e" % d[r] for r in rows]))
    if any(abs(x[r]) > 1e-6 for r in rows[1:]): bad += 1
print("clusters whose LATER rows are not zero in the reference's iterate:", bad)
