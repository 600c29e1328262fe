#!/usr/bin/env python3
"""vmg_proto.py -- scipy prototype of the viscosity multigrid on the oracle's assembled systems (solver research, CPU only; lives under
tests/ because it drives the oracle, which is test infrastructure; not collected by pytest).

    python tests/research/vmg_proto.py dump 128            # bench scene at 128^3 -> /tmp/visc_128.vdump (oracle_viscosity_dump_to)
    python tests/research/vmg_proto.py run 128 [options]   # PCG iteration counts: diagonal vs Galerkin multigrid variants

The transfer is the one of k_viscosity_mg.hip: per component, linear along the face normal, piecewise constant across.
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump(N, path, viscosity=5.0, nsub=1):
    from bench import build_workload
    from oracle import oraclebind as O
    I, J, K, dx, solid, P = build_workload("bunny", N, on_device=False)
    o = O.OracleSim(I, J, K, dx)
    o.set_solid(solid)
    o.set_viscosity(viscosity)
    o.particles = P
    o.set_solver_limits(vmaxiter=1 if nsub == 1 else 700, pmaxiter=0)
    for t in range(nsub):
        if t == nsub - 1:
            O.lib().oracle_viscosity_dump_to(path.encode())
        dt = 0.01
        o.substep(dt)
    O.lib().oracle_viscosity_dump_to(None)
    o.close()


def load(path, N):
    with open(path, "rb") as f:
        n, cap, dim, _ = np.fromfile(f, np.int64, 4)
        cnt = np.fromfile(f, np.int32, n)
        col = np.fromfile(f, np.uint32, n * cap).reshape(n, cap)
        val = np.fromfile(f, np.float64, n * cap).reshape(n, cap)
        rhs = np.fromfile(f, np.float64, n)
        table = np.fromfile(f, np.int32, dim)
    m = np.arange(cap)[None, :] < cnt[:, None]
    rows = np.repeat(np.arange(n), cnt)
    A = sp.csr_matrix((val[m], (rows, col[m].astype(np.int64))), shape=(n, n))
    I = J = K = N
    nu, nv = (I + 1) * J * K, I * (J + 1) * K
    comp = np.empty(n, np.int8)
    ijk = np.empty((n, 3), np.int32)
    for c, (off, w, h, cntc) in enumerate([(0, I + 1, J, nu), (nu, I, J + 1, nv), (nu + nv, I, J, dim - nu - nv)]):
        t = table[off:off + cntc]
        flat = np.nonzero(t >= 0)[0]
        r = t[flat]
        comp[r] = c
        ijk[r, 0] = flat % w
        ijk[r, 1] = (flat // w) % h
        ijk[r, 2] = flat // (w * h)
    return A, rhs, comp, ijk


def transfer_tri(comp, ijk, dims, tdirs=(True, True, True)):
    """per-component trilinear interpolation on the MAC lattices: along the normal 1 or (1/2, 1/2), across (3/4, 1/4) towards the
    nearer / farther coarse centre; a parent outside the lattice gives its weight to the one inside (constant extrapolation)"""
    n = len(comp)
    CI, CJ, CK = [(d + 1) // 2 for d in dims]
    stride = np.int64(4 * (max(CI, CJ, CK) + 2))
    rows, keys, w = [], [], []
    for c in range(3):
        sel = np.nonzero(comp == c)[0]
        p = ijk[sel].astype(np.int64)
        ext = [CI + (c == 0), CJ + (c == 1), CK + (c == 2)]
        # per axis: two parents and weights
        par, wt = [], []
        for a in range(3):
            if a == c:
                odd = (p[:, a] & 1) == 1
                p0 = np.where(odd, (p[:, a] - 1) >> 1, p[:, a] >> 1)
                p1 = np.where(odd, (p[:, a] + 1) >> 1, p[:, a] >> 1)
                w0 = np.where(odd, 0.5, 1.0); w1 = np.where(odd, 0.5, 0.0)
            else:
                p0 = p[:, a] >> 1
                p1 = np.where((p[:, a] & 1) == 1, p0 + 1, p0 - 1)
                if tdirs[a]:
                    w0 = np.full(len(p), 0.75); w1 = np.full(len(p), 0.25)
                else:
                    w0 = np.full(len(p), 1.0); w1 = np.zeros(len(p))
                out = (p1 < 0) | (p1 >= ext[a])
                w0 = np.where(out, w0 + w1, w0); w1 = np.where(out, 0.0, w1); p1 = np.where(out, p0, p1)
            par.append((p0, p1)); wt.append((w0, w1))
        for bx in range(2):
            for by in range(2):
                for bz in range(2):
                    ww = wt[0][bx] * wt[1][by] * wt[2][bz]
                    m = ww > 0
                    Q = [par[0][bx][m], par[1][by][m], par[2][bz][m]]
                    rows.append(sel[m]); w.append(ww[m])
                    keys.append(((c * stride + Q[2]) * stride + Q[1]) * stride + Q[0])
    rows = np.concatenate(rows); keys = np.concatenate(keys); w = np.concatenate(w)
    uk, inv = np.unique(keys, return_inverse=True)
    P = sp.csr_matrix((w, (rows, inv)), shape=(n, len(uk)))
    P.sum_duplicates()
    ccomp = (uk // stride ** 3).astype(np.int8)
    cijk = np.stack([uk % stride, (uk // stride) % stride, (uk // stride ** 2) % stride], 1).astype(np.int32)
    return P, ccomp, cijk, (CI, CJ, CK)


def transfer(comp, ijk, dims, mode="lin"):
    """P (fine dofs x coarse dofs) and the coarse dofs' (comp, ijk); only coarse dofs with a fine child exist"""
    if mode == "tri":
        return transfer_tri(comp, ijk, dims)
    n = len(comp)
    CI, CJ, CK = [(d + 1) // 2 for d in dims]
    ext = lambda c: (CI + (c == 0), CJ + (c == 1), CK + (c == 2))
    rows, keys, w = [], [], []
    stride = np.int64(4 * (max(CI, CJ, CK) + 2))
    for c in range(3):
        sel = np.nonzero(comp == c)[0]
        p = ijk[sel].astype(np.int64)
        base = p >> 1
        odd = (p[:, c] & 1) == 1
        if mode == "const":
            odd = np.zeros_like(odd)
        P0 = base.copy()
        P0[odd, c] = (p[odd, c] - 1) >> 1
        P1 = base.copy()
        P1[odd, c] = (p[odd, c] + 1) >> 1
        e = ext(c)
        def key(Q):
            return ((c * stride + Q[:, 2]) * stride + Q[:, 1]) * stride + Q[:, 0]
        rows.append(sel); keys.append(key(P0)); w.append(np.where(odd, 0.5, 1.0))
        rows.append(sel[odd]); keys.append(key(P1[odd])); w.append(np.full(odd.sum(), 0.5))
    rows = np.concatenate(rows); keys = np.concatenate(keys); w = np.concatenate(w)
    uk, inv = np.unique(keys, return_inverse=True)
    P = sp.csr_matrix((w, (rows, inv)), shape=(n, len(uk)))
    ccomp = (uk // stride ** 3).astype(np.int8)
    cijk = np.stack([uk % stride, (uk // stride) % stride, (uk // stride ** 2) % stride], 1).astype(np.int32)
    return P, ccomp, cijk, (CI, CJ, CK)


class MG:
    def __init__(self, A, comp, ijk, dims, nu1=2, nu2=2, omega=0.6, min_dim=4, smoother="jacobi", cheb_deg=2, f32=False, mode="lin", coarse_sweeps=8,
                 cheb_lo=0.25, alpha=1.0, gamma=1, lam_its=20, nu_fine=0, l1=0.0, nu_coarse=0, additive=0, gamma_at=-1, omega_coarse=0.0, nu_deep=0, deep_from=2, skip=-1, papp="",
                 papp_from=0):
        self.skip = skip
        self.papp = papp   # "tri": the CYCLE interpolates trilinearly while the coarse operators stay the Galerkin products of `mode` (round 6: is the cheap half of the trilinear hierarchy worth anything?)
        self.lev = []
        self.nu1, self.nu2, self.omega, self.smoother, self.cheb_deg, self.coarse_sweeps, self.cheb_lo = nu1, nu2, omega, smoother, cheb_deg, coarse_sweeps, cheb_lo
        self.alpha, self.gamma, self.nu_fine, self.nu_coarse = alpha, gamma, nu_fine, nu_coarse
        self.additive = additive
        self.gamma_at, self.omega_coarse = gamma_at, omega_coarse
        self.nu_deep, self.deep_from = nu_deep, deep_from   # levels >= deep_from smooth nu_deep times (0: like the others)   # gamma_at = l: only level l visits its coarser level gamma times
        while True:
            d = A.diagonal()
            if l1:   # l1-Jacobi: the smoother's diagonal is the row's absolute sum (times l1)
                d = np.asarray(abs(A).sum(axis=1)).ravel() * l1
            lam = None
            if smoother == "cheb" or omega < 0:
                lam = self.lmax(A, d, its=lam_its)
                gersh = (abs(A).sum(axis=1).A1 / d).max()
                print("  level %d: lambda_max ~ %.3f, Gershgorin %.3f" % (len(self.lev), lam / 1.1, gersh), flush=True)
            self.lev.append(dict(A=A, d=d, lam=lam))
            if max(dims) <= min_dim or A.shape[0] < 30:
                break
            comp_f, ijk_f, dims_f = comp, ijk, dims
            P, comp, ijk, dims = transfer(comp, ijk, dims, mode)
            self.lev[-1]["P"] = P
            self.lev[-1]["Pa"] = P
            if papp and len(self.lev) - 1 >= papp_from:
                # the application transfer: `papp`'s weights, restricted to the coarse dofs the Galerkin hierarchy has, rows rescaled to their original sums
                Pt, ct, it_, _ = transfer(comp_f, ijk_f, dims_f, papp)
                stride = np.int64(4 * (max(dims) + 2))
                key = lambda c_, q: ((c_.astype(np.int64) * stride + q[:, 2]) * stride + q[:, 1]) * stride + q[:, 0]
                kg, kt = key(comp, ijk.astype(np.int64)), key(ct, it_.astype(np.int64))
                order = np.argsort(kg); pos = np.searchsorted(kg[order], kt)
                ok = (pos < len(kg)) & (kg[order][np.minimum(pos, len(kg) - 1)] == kt)
                col = np.where(ok, order[np.minimum(pos, len(kg) - 1)], -1)
                Pt = Pt.tocoo()
                keep = col[Pt.col] >= 0
                Pa = sp.csr_matrix((Pt.data[keep], (Pt.row[keep], col[Pt.col[keep]])), shape=P.shape)
                rs_full = np.asarray(Pt.tocsr().sum(axis=1)).ravel(); rs = np.asarray(Pa.sum(axis=1)).ravel()
                scale = np.where(rs > 0, rs_full / np.where(rs > 0, rs, 1.0), 0.0)
                Pa = sp.diags(scale) @ Pa
                self.lev[-1]["Pa"] = Pa.tocsr()
                print("  level %d: application transfer %s: %.1f entries per fine row (Galerkin transfer %.1f), %d of %d of its coarse dofs exist" % (
                    len(self.lev) - 1, papp, Pa.nnz / Pa.shape[0], P.nnz / P.shape[0], int(ok.sum()), len(kt)), flush=True)
            A = (P.T @ A @ P).tocsr()
            if f32:
                A = A.astype(np.float32).astype(np.float64)
        print("levels:", [l["A"].shape[0] for l in self.lev], "nnz/row:", ["%.1f" % (l["A"].nnz / l["A"].shape[0]) for l in self.lev], flush=True)
        if smoother == "cheb":
            print("lambda_max(D^-1 A):", ["%.2f" % l["lam"] for l in self.lev])

    @staticmethod
    def lmax(A, d, its=20):
        rng = np.random.default_rng(0)
        v = rng.standard_normal(A.shape[0])
        lam = 1.0
        for _ in range(its):
            v /= np.linalg.norm(v)
            w = (A @ v) / d
            lam = np.linalg.norm(w)
            v = w
        return 1.1 * lam

    def smooth(self, l, x, b, n):
        L = self.lev[l]
        A, d = L["A"], L["d"]
        if self.smoother == "jacobi":
            om = self.omega if self.omega > 0 else -self.omega / (L["lam"] / 1.1)   # omega < 0: |omega| / lambda_max of the level
            if l > 0 and self.omega_coarse:
                om = self.omega_coarse
            for _ in range(n):
                x = x + om * (b - A @ x) / d if x is not None else om * b / d
            return x
        # Chebyshev of degree n*cheb_deg on [lam*lo, lam] for D^-1 A
        deg = n * self.cheb_deg
        lam = L["lam"]
        lo, hi = lam * self.cheb_lo, lam
        theta, delta = 0.5 * (hi + lo), 0.5 * (hi - lo)
        sigma = theta / delta
        rho = 1.0 / sigma
        r = b - A @ x if x is not None else b.copy()
        dvec = (r / d) / theta
        x = dvec.copy() if x is None else x + dvec
        for _ in range(deg - 1):
            rho_n = 1.0 / (2 * sigma - rho)
            r = b - A @ x
            dvec = rho_n * rho * dvec + (2 * rho_n / delta) * (r / d)
            x = x + dvec
            rho = rho_n
        return x

    def cycle(self, l, b):
        L = self.lev[l]
        if l == len(self.lev) - 1:
            if self.coarse_sweeps == 0:   # exact
                if "lu" not in L:
                    import scipy.sparse.linalg as sla
                    L["lu"] = sla.splu((L["A"] + 1e-9 * sp.diags(L["d"])).tocsc())
                return L["lu"].solve(b)
            if self.smoother == "jacobi":
                return self.smooth(l, None, b, self.coarse_sweeps)
            return self.smooth(l, None, b, max(1, self.coarse_sweeps // self.cheb_deg))
        if l == 0 and self.additive:
            # top level additive: z = S r + P Mc P^T r, S = `additive` symmetric Jacobi sweeps from zero; the two terms are independent
            xs = self.smooth(0, None, b, self.additive)
            return xs + self.alpha * (L["P"] @ self.cycle(1, L["P"].T @ b))
        if l == self.skip:   # this level only passes through (a 4:1 transfer between its neighbours: the composite of the two 2:1 ones)
            return L["P"] @ self.cycle(l + 1, L["P"].T @ b)
        nu1, nu2 = (self.nu_fine, self.nu_fine) if (l == 0 and self.nu_fine) else (self.nu1, self.nu2)
        if l > 0 and self.nu_coarse:
            nu1 = nu2 = self.nu_coarse
        if self.nu_deep and l >= self.deep_from:
            nu1 = nu2 = self.nu_deep
        x = self.smooth(l, None, b, nu1)
        r = b - L["A"] @ x
        bc = L["Pa"].T @ r
        xc = self.cycle(l + 1, bc)
        for _ in range((self.gamma if self.gamma_at in (-1, l) else 1) - 1):   # W-cycle: second visit, as a correction on the coarse level
            Ac = self.lev[l + 1]["A"]
            xc = xc + self.cycle(l + 1, bc - Ac @ xc)
        x = x + self.alpha * (L["Pa"] @ xc)
        return self.smooth(l, x, b, nu2)

    def __call__(self, r):
        return self.cycle(0, r)


class Affine:
    """global affine velocity fields (12 vectors: each component = a + b.x) as an extra coarse space around a preconditioner M"""
    def __init__(self, A, comp, ijk, M, how="add", blocks=1, N=1):
        n = len(comp)
        pos = ijk.astype(np.float64) + 0.5
        for c in range(3):
            pos[comp == c, c] -= 0.5
        pos -= pos.mean(axis=0)
        pos /= np.abs(pos).max()
        # blocks^3 boxes, each with its own 12 modes
        bid = np.zeros(n, np.int64)
        if blocks > 1:
            q = np.minimum((ijk.astype(np.int64) * blocks) // N, blocks - 1)
            bid = (q[:, 2] * blocks + q[:, 1]) * blocks + q[:, 0]
        ub, bid = np.unique(bid, return_inverse=True)
        cols, rows, vals = [], [], []
        for c in range(3):
            sel = np.nonzero(comp == c)[0]
            for m in range(4):
                v = np.ones(len(sel)) if m == 0 else pos[sel, m - 1]
                rows.append(sel); cols.append(bid[sel] * 12 + c * 4 + m); vals.append(v)
        Z = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, 12 * len(ub)))
        self.Z, self.A, self.M, self.how = Z, A, M, how
        E = (Z.T @ A @ Z).toarray()
        self.Einv = np.linalg.pinv(E)
        print("affine modes:", Z.shape[1], flush=True)

    def cz(self, r):
        return self.Z @ (self.Einv @ (self.Z.T @ r))

    def __call__(self, r):
        if self.how == "add":
            return self.M(r) + self.cz(r)
        x = self.cz(r)
        x = x + self.M(r - self.A @ x)
        return x + self.cz(r - self.A @ x)


def pcg(A, b, M, tol=1e-6, cap=20000, f32=False):
    """the reference's loop (pcgsolver.h:241-295): stop on max|r| <= tol * max|b|"""
    x = np.zeros_like(b)
    r = b.copy()
    tolabs = tol * np.abs(b).max()
    z = M(r)
    s = z.copy()
    rho = r @ z
    for it in range(cap):
        q = A @ s
        alpha = rho / (s @ q)
        x += alpha * s
        r -= alpha * q
        if f32:
            r = r.astype(np.float32).astype(np.float64)
        if np.abs(r).max() <= tolabs:
            return x, it + 1
        z = M(r)
        if f32:
            z = z.astype(np.float32).astype(np.float64)
        rho_n = r @ z
        s = z + (rho_n / rho) * s
        rho = rho_n
    return x, cap


def main():
    cmd, N = sys.argv[1], int(sys.argv[2])
    path = "/tmp/visc_%d.vdump" % N
    if cmd == "dump":
        nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 1
        t = time.time()
        dump(N, path, nsub=nsub)
        print("dumped", path, "in %.1f s" % (time.time() - t))
        return
    A, rhs, comp, ijk = load(path, N)
    print("rows", A.shape[0], "nnz", A.nnz, flush=True)
    variants = sys.argv[3:] or ["jacobi", "v22", "v11"]
    for v in variants:
        t = time.time()
        if v == "jacobi":
            d = A.diagonal()
            _, its = pcg(A, rhs, lambda r: r / d)
        else:
            # e.g. v22, v11, c2 (Chebyshev degree 2 pre/post), c3, v22f (fp32 coarse operators), v22c (piecewise-constant P)
            kw = {}
            name = v
            if ":" in v:
                name, opts = v.split(":", 1)
                for o in opts.split(","):
                    k_, v_ = o.split("=")
                    kw[k_] = v_ if v_.isalpha() else (float(v_) if "." in v_ else int(v_))
            if "f" in name[1:]:
                kw["f32"] = True; name = name.replace("f", "")
            if name.endswith("k"):
                kw["mode"] = "const"; name = name[:-1]
            if name[0] == "v":
                kw.update(nu1=int(name[1]), nu2=int(name[2]))
                if len(name) > 3:
                    kw["omega"] = float(name[3:])
            elif name[0] == "c":
                kw.update(smoother="cheb", nu1=1, nu2=1, cheb_deg=int(name[1]))
                if len(name) > 2:
                    kw["cheb_lo"] = float(name[2:])
            aff = kw.pop("aff", None); blocks = kw.pop("blocks", 1)
            M = MG(A, comp, ijk, (N, N, N), **kw)
            if aff:
                M = Affine(A, comp, ijk, M, aff, blocks, N)
            _, its = pcg(A, rhs, M, f32=kw.get("f32", False))
        print("%-8s N=%d: %d iterations (%.1f s)" % (v, N, its, time.time() - t), flush=True)


if __name__ == "__main__":
    main()
