import numpy as np


def probe_err(c, g, t):
    num = den = 0.0
    for n in "UVW":
        a = c.grid(n).reshape(-1)
        num = max(num, float(np.abs(a[g["s%d_probe_idx_%s" % (t, n)]].astype(np.float64) - g["s%d_probe_val_%s" % (t, n)]).max()))
        den = max(den, float(g["s%d_maxabs_%s" % (t, n)]))
    return num / den
