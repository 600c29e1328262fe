"""ctypes binding for oracle/_ref/libflipref.so (the unmodified reference + oracle/ref_harness.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never from the product package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libflipref.so")

GRID_IDS = dict(U=0, V=1, W=2, SAVED_U=3, SAVED_V=4, SAVED_W=5, VALID_U=6, VALID_V=7, VALID_W=8,
                LIQUID_PHI=9, SOLID_PHI=10, WEIGHT_U=11, WEIGHT_V=12, WEIGHT_W=13, VISCOSITY=14,
                PRESSURE=15)


def grid_shape(which, I, J, K):
    """(depth, height, width) numpy shape of a grid in Array3d layout (x fastest)."""
    name = which if isinstance(which, str) else {v: k for k, v in GRID_IDS.items()}[which]
    if name.endswith("_U") or name == "U":
        return (K, J, I + 1)
    if name.endswith("_V") or name == "V":
        return (K, J + 1, I)
    if name.endswith("_W") or name == "W":
        return (K + 1, J, I)
    if name in ("SOLID_PHI", "VISCOSITY"):
        return (K + 1, J + 1, I + 1)
    return (K, J, I)


def available():
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        fp = C.POINTER(C.c_float)
        ip = C.POINTER(C.c_int)
        dp = C.POINTER(C.c_double)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_srand.argtypes = [C.c_uint]
        L.ref_add_boundary.argtypes = [C.c_void_p, fp, C.c_int, ip, C.c_int, C.c_int]
        L.ref_reset_boundary.argtypes = [C.c_void_p]
        L.ref_add_liquid.argtypes = [C.c_void_p, fp, C.c_int, ip, C.c_int]
        L.ref_mesh_sdf.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, fp, C.c_int, ip, C.c_int, C.c_int, fp, ip]
        L.ref_set_viscosity.argtypes = [C.c_void_p, C.c_float]
        L.ref_set_viscosity_grid.argtypes = [C.c_void_p, fp]
        L.ref_set_gravity.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.ref_set_viscosity_solver.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.ref_num_particles.restype = C.c_size_t
        L.ref_num_particles.argtypes = [C.c_void_p]
        L.ref_get_particles.argtypes = [C.c_void_p, fp]
        L.ref_set_particles.argtypes = [C.c_void_p, fp, C.c_size_t]
        L.ref_get_grid.argtypes = [C.c_void_p, C.c_int, fp]
        L.ref_set_grid.argtypes = [C.c_void_p, C.c_int, fp]
        L.ref_cfl.restype = C.c_float
        L.ref_cfl.argtypes = [C.c_void_p]
        for name in ("ref_update_liquid_sdf", "ref_advect_velocity_field", "ref_compute_weights",
                     "ref_extrapolate", "ref_constrain", "ref_update_particle_velocities"):
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("ref_add_body_force", "ref_apply_viscosity", "ref_solve_pressure",
                     "ref_apply_pressure", "ref_advect_particles"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_float]
        L.ref_p2g_component.argtypes = [C.c_void_p, C.c_int, fp, fp]
        L.ref_get_solver_stats.argtypes = [C.c_void_p, ip, dp, ip, dp]
        L.ref_substep.argtypes = [C.c_void_p, C.c_float, dp]
        L.ref_advance.restype = C.c_int
        L.ref_advance.argtypes = [C.c_void_p, C.c_float]
        L.ref_fraction_inside2.restype = C.c_float
        L.ref_fraction_inside2.argtypes = [C.c_float, C.c_float]
        L.ref_fraction_inside4.restype = C.c_float
        L.ref_fraction_inside4.argtypes = [C.c_float] * 4
        L.ref_volume_fraction8.restype = C.c_float
        L.ref_volume_fraction8.argtypes = [fp]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class RefSim:
    """The reference FluidSimulation, phase by phase."""

    def __init__(self, I, J, K, dx):
        self.I, self.J, self.K, self.dx = I, J, K, float(np.float32(dx))
        self.h = lib().ref_create(I, J, K, C.c_float(dx))

    def close(self):
        if self.h:
            lib().ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_boundary(self, verts, tris, inverted=False):
        v = np.ascontiguousarray(verts, np.float32)
        t = np.ascontiguousarray(tris, np.int32)
        lib().ref_add_boundary(self.h, _fp(v), len(v), _ip(t), len(t), int(inverted))

    def add_liquid(self, verts, tris):
        v = np.ascontiguousarray(verts, np.float32)
        t = np.ascontiguousarray(tris, np.int32)
        lib().ref_add_liquid(self.h, _fp(v), len(v), _ip(t), len(t))

    def set_viscosity(self, v):
        if np.isscalar(v):
            lib().ref_set_viscosity(self.h, float(v))
        else:
            a = np.ascontiguousarray(v, np.float32)
            lib().ref_set_viscosity_grid(self.h, _fp(a))

    def set_gravity(self, gx, gy, gz):
        lib().ref_set_gravity(self.h, gx, gy, gz)

    def set_viscosity_solver(self, maxiter=0, tol=0.0):
        lib().ref_set_viscosity_solver(self.h, maxiter, tol)

    @property
    def particles(self):
        n = lib().ref_num_particles(self.h)
        a = np.empty((n, 6), np.float32)
        if n:
            lib().ref_get_particles(self.h, _fp(a))
        return a

    @particles.setter
    def particles(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 6)
        lib().ref_set_particles(self.h, _fp(a), len(a))

    def grid(self, name):
        a = np.empty(grid_shape(name, self.I, self.J, self.K), np.float32)
        rc = lib().ref_get_grid(self.h, GRID_IDS[name], _fp(a))
        assert rc == 0
        return a

    def set_grid(self, name, a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.shape == grid_shape(name, self.I, self.J, self.K), (a.shape, name)
        rc = lib().ref_set_grid(self.h, GRID_IDS[name], _fp(a))
        assert rc == 0

    def cfl(self):
        return lib().ref_cfl(self.h)

    def update_liquid_sdf(self):
        lib().ref_update_liquid_sdf(self.h)

    def advect_velocity_field(self):
        lib().ref_advect_velocity_field(self.h)

    def p2g_component(self, d):
        shp = grid_shape("UVW"[d], self.I, self.J, self.K)
        f = np.empty(shp, np.float32)
        s = np.empty(shp, np.float32)
        lib().ref_p2g_component(self.h, d, _fp(f), _fp(s))
        return f, s

    def add_body_force(self, dt):
        lib().ref_add_body_force(self.h, dt)

    def apply_viscosity(self, dt):
        return lib().ref_apply_viscosity(self.h, dt)

    def compute_weights(self):
        lib().ref_compute_weights(self.h)

    def solve_pressure(self, dt):
        return lib().ref_solve_pressure(self.h, dt)

    def apply_pressure(self, dt):
        lib().ref_apply_pressure(self.h, dt)

    def extrapolate(self):
        lib().ref_extrapolate(self.h)

    def constrain(self):
        lib().ref_constrain(self.h)

    def update_particle_velocities(self):
        lib().ref_update_particle_velocities(self.h)

    def advect_particles(self, dt):
        lib().ref_advect_particles(self.h, dt)

    def solver_stats(self):
        vi, pi = C.c_int(), C.c_int()
        ve, pe = C.c_double(), C.c_double()
        lib().ref_get_solver_stats(self.h, C.byref(vi), C.byref(ve), C.byref(pi), C.byref(pe))
        return dict(visc_iters=vi.value, visc_err=ve.value, pres_iters=pi.value, pres_err=pe.value)

    def substep(self, dt):
        sec = (C.c_double * 7)()
        lib().ref_substep(self.h, dt, sec)
        return list(sec)

    def advance(self, dt):
        return lib().ref_advance(self.h, dt)


def mesh_sdf(I, J, K, dx, verts, tris, band=3):
    v = np.ascontiguousarray(verts, np.float32)
    t = np.ascontiguousarray(tris, np.int32)
    phi = np.empty((K + 1, J + 1, I + 1), np.float32)
    closest = np.empty((K + 1, J + 1, I + 1), np.int32)
    lib().ref_mesh_sdf(I, J, K, C.c_float(dx), _fp(v), len(v), _ip(t), len(t), band, _fp(phi), _ip(closest))
    return phi, closest
