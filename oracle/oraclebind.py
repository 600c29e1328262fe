"""ctypes binding for oracle/libfliporacle.so (the plain-C CPU restatement, oracle/flip_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .refbind import GRID_IDS, grid_shape  # noqa: F401  (same ids / shapes)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfliporacle.so")


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int), ("residual", C.c_double), ("status", C.c_int),
                ("rows", C.c_int), ("nnz", C.c_long)]

    def as_dict(self):
        return dict(iterations=self.iterations, residual=self.residual, status=self.status,
                    rows=self.rows, nnz=self.nnz)


def build(force=False):
    src = os.path.join(_HERE, "flip_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return LIB_PATH


_lib = None
fp = C.POINTER(C.c_float)
bp = C.POINTER(C.c_uint8)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        i3 = [C.c_int, C.c_int, C.c_int]
        L.oracle_fraction_inside2.restype = C.c_float
        L.oracle_fraction_inside2.argtypes = [C.c_float] * 2
        L.oracle_fraction_inside4.restype = C.c_float
        L.oracle_fraction_inside4.argtypes = [C.c_float] * 4
        L.oracle_volume_fraction8.restype = C.c_float
        L.oracle_volume_fraction8.argtypes = [fp]
        L.oracle_particle_sdf.argtypes = i3 + [C.c_float, fp, C.c_size_t, fp, fp]
        L.oracle_p2g_component.argtypes = i3 + [C.c_float, fp, C.c_size_t, C.c_int, fp, bp]
        L.oracle_p2g.argtypes = i3 + [C.c_float, fp, C.c_size_t, fp, fp, fp, fp, bp, bp, bp]
        L.oracle_extrapolate_grid.argtypes = i3 + [fp, bp, C.c_int]
        L.oracle_body_force.argtypes = i3 + [fp, fp, fp, fp] + [C.c_float] * 4
        L.oracle_cfl.restype = C.c_float
        L.oracle_cfl.argtypes = i3 + [C.c_float, fp, fp, fp, C.c_float]
        L.oracle_compute_weights.argtypes = i3 + [fp, fp, fp, fp]
        L.oracle_pressure_solve.argtypes = i3 + [C.c_float, C.c_float] + [fp] * 7 + [
            C.c_float, C.c_double, C.c_int, fp, C.POINTER(SolveInfo)]
        L.oracle_apply_pressure.argtypes = i3 + [C.c_float, C.c_float, fp, fp, fp, fp, fp, C.c_float,
                                                 fp, fp, fp, bp, bp, bp]
        L.oracle_constrain.argtypes = i3 + [fp] * 9
        L.oracle_viscosity_solve.argtypes = i3 + [C.c_float, C.c_float, fp, fp, fp, fp, fp, fp,
                                                  C.c_double, C.c_int, C.c_double, C.POINTER(SolveInfo)]
        L.oracle_viscosity_volumes.argtypes = i3 + [C.c_float] + [fp] * 8
        L.oracle_update_particle_velocities.argtypes = i3 + [C.c_float, fp, C.c_size_t] + [fp] * 6 + [C.c_float]
        L.oracle_advect_particles.argtypes = i3 + [C.c_float, C.c_float, fp, C.c_size_t] + [fp] * 7 + [C.c_float]
        L.oracle_sim_create.restype = C.c_void_p
        L.oracle_sim_create.argtypes = i3 + [C.c_float]
        L.oracle_sim_destroy.argtypes = [C.c_void_p]
        L.oracle_sim_set_solid.argtypes = [C.c_void_p, fp]
        L.oracle_sim_set_viscosity.argtypes = [C.c_void_p, fp]
        L.oracle_sim_set_gravity.argtypes = [C.c_void_p] + [C.c_float] * 3
        L.oracle_sim_set_particles.argtypes = [C.c_void_p, fp, C.c_size_t]
        L.oracle_sim_num_particles.restype = C.c_size_t
        L.oracle_sim_num_particles.argtypes = [C.c_void_p]
        L.oracle_sim_get_particles.argtypes = [C.c_void_p, fp]
        L.oracle_sim_set_solver_limits.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int]
        L.oracle_sim_get_grid.argtypes = [C.c_void_p, C.c_int, fp]
        L.oracle_sim_set_grid.argtypes = [C.c_void_p, C.c_int, fp]
        L.oracle_sim_substep.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_double),
                                         C.POINTER(SolveInfo), C.POINTER(SolveInfo)]
        L.oracle_sim_advance.restype = C.c_int
        L.oracle_sim_advance.argtypes = [C.c_void_p, C.c_float]
        _lib = L
    return _lib


def F(a):
    return a.ctypes.data_as(fp)


def B(a):
    return a.ctypes.data_as(bp)


def f32(a):
    return np.ascontiguousarray(a, np.float32)


# ---------- functional wrappers (numpy in, numpy out) ----------

def particle_sdf(I, J, K, dx, particles, solid):
    particles, solid = f32(particles), f32(solid)
    phi = np.empty(grid_shape("LIQUID_PHI", I, J, K), np.float32)
    lib().oracle_particle_sdf(I, J, K, dx, F(particles), len(particles), F(solid), F(phi))
    return phi


def p2g_component(I, J, K, dx, particles, d):
    particles = f32(particles)
    shp = grid_shape("UVW"[d], I, J, K)
    f = np.empty(shp, np.float32)
    s = np.empty(shp, np.uint8)
    lib().oracle_p2g_component(I, J, K, dx, F(particles), len(particles), d, F(f), B(s))
    return f, s


def p2g(I, J, K, dx, particles, phi):
    particles, phi = f32(particles), f32(phi)
    out = [np.empty(grid_shape(c, I, J, K), np.float32) for c in "UVW"]
    val = [np.empty(grid_shape(c, I, J, K), np.uint8) for c in "UVW"]
    lib().oracle_p2g(I, J, K, dx, F(particles), len(particles), F(phi), F(out[0]), F(out[1]), F(out[2]),
                     B(val[0]), B(val[1]), B(val[2]))
    return out, val


def extrapolate_grid(grid, valid, layers):
    g = f32(grid).copy()
    v = np.ascontiguousarray(valid, np.uint8)
    d, h, w = g.shape
    lib().oracle_extrapolate_grid(w, h, d, F(g), B(v), layers)
    return g


def body_force(I, J, K, phi, U, V, W, g, dt):
    U, V, W = f32(U).copy(), f32(V).copy(), f32(W).copy()
    phi = f32(phi)
    lib().oracle_body_force(I, J, K, F(phi), F(U), F(V), F(W), g[0], g[1], g[2], dt)
    return U, V, W


def cfl(I, J, K, dx, U, V, W, cfl_number=5.0):
    U, V, W = f32(U), f32(V), f32(W)
    return lib().oracle_cfl(I, J, K, dx, F(U), F(V), F(W), cfl_number)


def compute_weights(I, J, K, solid):
    solid = f32(solid)
    w = [np.empty(grid_shape(c, I, J, K), np.float32) for c in "UVW"]
    lib().oracle_compute_weights(I, J, K, F(solid), F(w[0]), F(w[1]), F(w[2]))
    return w


def pressure_solve(I, J, K, dx, dt, U, V, W, wU, wV, wW, phi, minfrac=0.01, tol=1e-9, maxiter=200):
    a = [f32(x) for x in (U, V, W, wU, wV, wW, phi)]
    p = np.empty(grid_shape("PRESSURE", I, J, K), np.float32)
    info = SolveInfo()
    lib().oracle_pressure_solve(I, J, K, dx, dt, *[F(x) for x in a], minfrac, tol, maxiter, F(p), C.byref(info))
    return p, info.as_dict()


def apply_pressure(I, J, K, dx, dt, p, phi, wU, wV, wW, U, V, W, minfrac=0.01):
    p, phi, wU, wV, wW = [f32(x) for x in (p, phi, wU, wV, wW)]
    U, V, W = f32(U).copy(), f32(V).copy(), f32(W).copy()
    val = [np.empty(grid_shape(c, I, J, K), np.uint8) for c in "UVW"]
    lib().oracle_apply_pressure(I, J, K, dx, dt, F(p), F(phi), F(wU), F(wV), F(wW), minfrac, F(U), F(V), F(W),
                                B(val[0]), B(val[1]), B(val[2]))
    return (U, V, W), val


def constrain(I, J, K, wU, wV, wW, U, V, W, sU, sV, sW):
    wU, wV, wW = [f32(x) for x in (wU, wV, wW)]
    o = [f32(x).copy() for x in (U, V, W, sU, sV, sW)]
    lib().oracle_constrain(I, J, K, F(wU), F(wV), F(wW), *[F(x) for x in o])
    return o


def viscosity_solve(I, J, K, dx, dt, U, V, W, phi, solid, visc, tol=1e-6, maxiter=700, accept=10.0):
    U, V, W = f32(U).copy(), f32(V).copy(), f32(W).copy()
    phi, solid, visc = f32(phi), f32(solid), f32(visc)
    info = SolveInfo()
    lib().oracle_viscosity_solve(I, J, K, dx, dt, F(U), F(V), F(W), F(phi), F(solid), F(visc), tol, maxiter,
                                 accept, C.byref(info))
    return (U, V, W), info.as_dict()


def viscosity_volumes(I, J, K, dx, phi):
    phi = f32(phi)
    shapes = [(K, J, I), (K, J, I + 1), (K, J + 1, I), (K + 1, J, I), (K + 1, J + 1, I), (K + 1, J, I + 1),
              (K, J + 1, I + 1)]
    out = [np.empty(s, np.float32) for s in shapes]
    lib().oracle_viscosity_volumes(I, J, K, dx, F(phi), *[F(o) for o in out])
    return dict(zip(["center", "U", "V", "W", "edgeU", "edgeV", "edgeW"], out))


def update_particle_velocities(I, J, K, dx, particles, U, V, W, sU, sV, sW, ratio=0.05):
    p = f32(particles).copy()
    g = [f32(x) for x in (U, V, W, sU, sV, sW)]
    lib().oracle_update_particle_velocities(I, J, K, dx, F(p), len(p), *[F(x) for x in g], ratio)
    return p


def advect_particles(I, J, K, dx, dt, particles, U, V, W, sU, sV, sW, solid, ratio=0.05):
    p = f32(particles).copy()
    g = [f32(x) for x in (U, V, W, sU, sV, sW, solid)]
    lib().oracle_advect_particles(I, J, K, dx, dt, F(p), len(p), *[F(x) for x in g], ratio)
    return p


class OracleSim:
    def __init__(self, I, J, K, dx):
        self.I, self.J, self.K, self.dx = I, J, K, float(np.float32(dx))
        self.h = lib().oracle_sim_create(I, J, K, dx)

    def close(self):
        if self.h:
            lib().oracle_sim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_solid(self, nodes):
        a = f32(nodes)
        assert a.shape == grid_shape("SOLID_PHI", self.I, self.J, self.K)
        lib().oracle_sim_set_solid(self.h, F(a))

    def set_viscosity(self, v):
        if np.isscalar(v):
            v = np.full(grid_shape("VISCOSITY", self.I, self.J, self.K), v, np.float32)
        a = f32(v)
        lib().oracle_sim_set_viscosity(self.h, F(a))

    def set_gravity(self, gx, gy, gz):
        lib().oracle_sim_set_gravity(self.h, gx, gy, gz)

    def set_solver_limits(self, ptol=0.0, pmaxiter=0, vtol=0.0, vmaxiter=0):
        lib().oracle_sim_set_solver_limits(self.h, ptol, pmaxiter, vtol, vmaxiter)

    @property
    def particles(self):
        n = lib().oracle_sim_num_particles(self.h)
        a = np.empty((n, 6), np.float32)
        if n:
            lib().oracle_sim_get_particles(self.h, F(a))
        return a

    @particles.setter
    def particles(self, a):
        a = f32(a).reshape(-1, 6)
        lib().oracle_sim_set_particles(self.h, F(a), len(a))

    def grid(self, name):
        a = np.empty(grid_shape(name, self.I, self.J, self.K), np.float32)
        assert lib().oracle_sim_get_grid(self.h, GRID_IDS[name], F(a)) == 0
        return a

    def set_grid(self, name, a):
        a = f32(a)
        assert a.shape == grid_shape(name, self.I, self.J, self.K)
        assert lib().oracle_sim_set_grid(self.h, GRID_IDS[name], F(a)) == 0

    def substep(self, dt):
        sec = (C.c_double * 7)()
        vi, pi = SolveInfo(), SolveInfo()
        lib().oracle_sim_substep(self.h, dt, sec, C.byref(vi), C.byref(pi))
        return list(sec), vi.as_dict(), pi.as_dict()

    def advance(self, dt):
        return lib().oracle_sim_advance(self.h, dt)
