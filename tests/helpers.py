"""Shared helpers for the parity tests: golden fixtures and comparison metrics."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = ["cube24_inviscid", "bunny32_viscous", "twobody20_varvisc"]


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.I, self.J, self.K = int(self.z["I"]), int(self.z["J"]), int(self.z["K"])
        self.dx = float(self.z["dx"])
        self.dt = float(self.z["dt"])
        self.gravity = tuple(float(g) for g in self.z["gravity"])
        self.nsub = int(self.z["nsub"])

    def __getitem__(self, k):
        return self.z[k]

    def dims(self):
        return self.I, self.J, self.K

    def particles_before(self, t):
        return self.z["particles0"] if t == 0 else self.z["s%d_particles" % (t - 1)]

    def uvw(self, t, stage):
        return [self.z["s%d_%s_%s" % (t, stage, c)] for c in "UVW"]

    def valid(self, t, stage):
        return [self.z["s%d_%s_valid_%s" % (t, stage, c)] for c in "UVW"]


def rel_maxnorm(a, b):
    """max|a-b| / max|b| -- the 'relative max-norm' of BASELINE.json's north_star."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.abs(b).max()
    if den == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / den)


def rel_maxnorm3(A, B):
    """relative max-norm over the three MAC components together"""
    num = max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() for a, b in zip(A, B))
    den = max(np.abs(np.asarray(b, np.float64)).max() for b in B)
    return float(num / den) if den else float(num)


def fluid_face_masks(phi):
    """faces bordering a phi<0 cell (Grid3d::isFaceBorderingValueU/V/W, reference grid3d.h:496-530)"""
    f = phi < 0
    K, J, I = f.shape
    mu = np.zeros((K, J, I + 1), bool)
    mu[:, :, :-1] |= f
    mu[:, :, 1:] |= f
    mv = np.zeros((K, J + 1, I), bool)
    mv[:, :-1, :] |= f
    mv[:, 1:, :] |= f
    mw = np.zeros((K + 1, J, I), bool)
    mw[:-1, :, :] |= f
    mw[1:, :, :] |= f
    return mu, mv, mw


def assert_same_particle_set(a, b, tol=1e-5):
    """two particle arrays hold the same particles up to `tol` in position, in any order: every particle of `a` has a partner in
    `b` within tol, and the pairing is one to one (sorting both and comparing row by row mispairs particles whose leading
    coordinate differs in the last bit)"""
    from scipy.spatial import cKDTree
    a = np.asarray(a)[:, :3].astype(np.float64)
    b = np.asarray(b)[:, :3].astype(np.float64)
    assert len(a) == len(b), (len(a), len(b))
    if len(a) == 0:
        return
    dist, idx = cKDTree(b).query(a, k=1)
    assert dist.max() <= tol, dist.max()
    # one to one, unless two particles of b coincide within tol (then any of them is a valid partner)
    if len(np.unique(idx)) != len(idx):
        dist2, _ = cKDTree(a).query(b, k=1)
        assert dist2.max() <= tol, dist2.max()
