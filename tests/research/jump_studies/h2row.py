import sys
sys.path.insert(0, "/root/repo/tests/research")
import numpy as np
from jump_proto import load
i = int(sys.argv[1]); N = int(sys.argv[2]); faces = [tuple(a.split(",")) for a in sys.argv[3:]]
P = "/root/repo/scratch/jump/h2_%02d.vdump" % i
A, b, dgx, vol, table = load(P)
x = np.fromfile(P + ".x", np.float64)
I = J = K = N
nu_, nv_ = (I + 1) * J * K, I * (J + 1) * K
A = A.tocsr(); d = A.diagonal()
def where(r):
    f = np.flatnonzero(table == r)[0]
    if f < nu_: c, w, h = "U", I + 1, J
    elif f < nu_ + nv_: c, w, h, f = "V", I, J + 1, f - nu_
    else: c, w, h, f = "W", I, J, f - nu_ - nv_
    return "%s(%d,%d,%d)" % (c, f % w, (f // w) % h, f // (w * h))
def row_of(c, i, j, k):
    if c == "U": return table[i + (I + 1) * (j + J * k)]
    if c == "V": return table[nu_ + i + I * (j + (J + 1) * k)]
    return table[nu_ + nv_ + i + I * (j + J * k)]
for c, i_, j_, k_ in faces:
    r = row_of(c, int(i_), int(j_), int(k_))
    if r < 0: print(c, i_, j_, k_, "is not a row"); continue
    cols = A.indices[A.indptr[r]:A.indptr[r+1]]; vals = A.data[A.indptr[r]:A.indptr[r+1]]
    print("row %d %s diag %.4e vol %.3e b %.3e x_ref %.5f | " % (r, where(r), d[r], vol[r], b[r], x[r]), [(where(cc), "%.3e" % (v / d[r]), "x=%.4f" % x[cc]) for cc, v in zip(cols, vals) if cc != r and v != 0])
