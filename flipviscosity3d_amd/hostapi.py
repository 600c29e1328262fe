"""Python mirror of the host-side FluidSimulation class (flipviscosity3d_amd/host/fluidsimulation.h),
bound through the C wrapper in host/flipv_host.h.

Same call sequence as the reference's driver (reference main.cpp:42-90):

    sim = FluidSimulation(); sim.initialize(64, 64, 64, 1/64)
    sim.addBoundary(mesh, True); sim.addLiquid(mesh); sim.setViscosity(5.0); sim.setGravity(0, -9.81, 0)
    for frame in range(n): sim.advance(0.01); sim.particles ...

Setup runs on the host (C++), advance() on the GPU; nothing here computes anything in Python.
"""
import ctypes as C
import os

import numpy as np

from . import capi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libflipv_host.so")
SYMBOLS = ["flipvh_create", "flipvh_create_ex", "flipvh_save_state", "flipvh_load_state", "flipvh_get_dims", "flipvh_destroy", "flipvh_add_boundary", "flipvh_reset_boundary", "flipvh_set_seeding",
           "flipvh_add_liquid", "flipvh_set_viscosity", "flipvh_set_viscosity_grid", "flipvh_set_gravity",
           "flipvh_num_particles", "flipvh_get_particles", "flipvh_set_particles", "flipvh_get_solid_sdf",
           "flipvh_advance", "flipvh_context", "flipvh_mesh_sdf", "flipvh_load_ply"]

_lib = None
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def load():
    global _lib
    if _lib is not None:
        return _lib
    capi.load()  # libflipv.so first (rpath $ORIGIN also finds it)
    if not os.path.exists(LIB_PATH):
        raise OSError("libflipv_host.so not built: run `make -C flipviscosity3d_amd/host`")
    L = C.CDLL(LIB_PATH)
    h = C.c_void_p
    L.flipvh_create.restype = h
    L.flipvh_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    L.flipvh_create_ex.restype = h
    L.flipvh_create_ex.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    L.flipvh_destroy.argtypes = [h]
    L.flipvh_add_boundary.argtypes = [h, fp, C.c_int, ip, C.c_int, C.c_int]
    L.flipvh_reset_boundary.argtypes = [h]
    L.flipvh_set_seeding.argtypes = [h, C.c_int, C.c_ulonglong]
    L.flipvh_add_liquid.argtypes = [h, fp, C.c_int, ip, C.c_int]
    L.flipvh_set_viscosity.argtypes = [h, C.c_float]
    L.flipvh_set_viscosity_grid.argtypes = [h, fp]
    L.flipvh_set_gravity.argtypes = [h, C.c_float, C.c_float, C.c_float]
    L.flipvh_num_particles.restype = C.c_size_t
    L.flipvh_num_particles.argtypes = [h]
    L.flipvh_get_particles.argtypes = [h, fp]
    L.flipvh_set_particles.argtypes = [h, fp, C.c_size_t]
    L.flipvh_get_solid_sdf.argtypes = [h, fp]
    L.flipvh_advance.argtypes = [h, C.c_float, C.POINTER(capi.Stats)]
    L.flipvh_save_state.argtypes = [h, C.c_char_p]
    L.flipvh_load_state.argtypes = [h, C.c_char_p]
    L.flipvh_get_dims.argtypes = [h, C.POINTER(C.c_int)]
    L.flipvh_context.restype = h
    L.flipvh_context.argtypes = [h]
    L.flipvh_mesh_sdf.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, fp, C.c_int, ip, C.c_int, C.c_int, fp, ip]
    L.flipvh_load_ply.argtypes = [C.c_char_p, ip, ip, fp, ip]
    _lib = L
    return L


def _mesh_args(mesh):
    v = np.ascontiguousarray(mesh[0], np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(mesh[1], np.int32).reshape(-1, 3)
    return v, t, v.ctypes.data_as(fp), len(v), t.ctypes.data_as(ip), len(t)


def load_ply(path):
    """TriangleMesh::loadPLY of the host library -> (vertices (N,3) f32, triangles (M,3) i32)."""
    L = load()
    nv, nt = C.c_int(), C.c_int()
    if L.flipvh_load_ply(path.encode(), C.byref(nv), C.byref(nt), None, None) != 0:
        raise IOError("cannot load PLY %s" % path)
    v = np.empty((nv.value, 3), np.float32)
    t = np.empty((nt.value, 3), np.int32)
    L.flipvh_load_ply(path.encode(), C.byref(nv), C.byref(nt), v.ctypes.data_as(fp), t.ctypes.data_as(ip))
    return v, t


def mesh_sdf(I, J, K, dx, mesh, band=3):
    """MeshLevelSet::calculateSignedDistanceField -> (phi nodes, closest-triangle index nodes)."""
    L = load()
    v, t, vp, nv, tp, nt = _mesh_args(mesh)
    phi = np.empty((K + 1, J + 1, I + 1), np.float32)
    closest = np.empty((K + 1, J + 1, I + 1), np.int32)
    L.flipvh_mesh_sdf(I, J, K, C.c_float(dx), vp, nv, tp, nt, band, phi.ctypes.data_as(fp), closest.ctypes.data_as(ip))
    return phi, closest


class FluidSimulation:
    SEED_LIBC_RAND, SEED_COUNTER = 0, 1

    def __init__(self):
        self.L = load()
        self.h = None

    def initialize(self, i, j, k, dx, setup_on_device=False):
        """setup_on_device: mesh level sets and seeding run as HIP kernels (needs a GPU; counter-based seeding, seed 0
        unless setSeeding is called)."""
        self.close()
        self.I, self.J, self.K, self.dx = int(i), int(j), int(k), float(np.float32(dx))
        self.h = self.L.flipvh_create_ex(self.I, self.J, self.K, C.c_float(dx), int(bool(setup_on_device)))
        if not self.h:
            raise ValueError("initialize: bad grid dimensions")

    def close(self):
        if getattr(self, "h", None):
            self.L.flipvh_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def addBoundary(self, mesh, isInverted=False):
        v, t, vp, nv, tp, nt = _mesh_args(mesh)
        if self.L.flipvh_add_boundary(self.h, vp, nv, tp, nt, int(bool(isInverted))) != 0:
            raise ValueError("addBoundary: mesh bounding box must lie inside the domain")

    addSolid = addBoundary  # BASELINE.json north_star naming

    def resetBoundary(self):
        self.L.flipvh_reset_boundary(self.h)

    def setSeeding(self, mode, seed=0):
        self.L.flipvh_set_seeding(self.h, int(mode), int(seed))

    def addLiquid(self, mesh):
        v, t, vp, nv, tp, nt = _mesh_args(mesh)
        if self.L.flipvh_add_liquid(self.h, vp, nv, tp, nt) != 0:
            raise ValueError("addLiquid: mesh bounding box must lie inside the domain")

    def setViscosity(self, value):
        if np.isscalar(value):
            rc = self.L.flipvh_set_viscosity(self.h, float(value))
        else:
            a = np.ascontiguousarray(value, np.float32)
            if a.shape != (self.K + 1, self.J + 1, self.I + 1):
                raise ValueError("setViscosity: grid must be (K+1, J+1, I+1) nodes")
            rc = self.L.flipvh_set_viscosity_grid(self.h, a.ctypes.data_as(fp))
        if rc != 0:
            raise ValueError("setViscosity: values must be >= 0")

    def setGravity(self, gx, gy=None, gz=None):
        if gy is None:
            gx, gy, gz = gx
        self.L.flipvh_set_gravity(self.h, gx, gy, gz)

    @property
    def particles(self):
        n = self.L.flipvh_num_particles(self.h)
        a = np.empty((n, 6), np.float32)
        if n:
            self.L.flipvh_get_particles(self.h, a.ctypes.data_as(fp))
        return a

    @particles.setter
    def particles(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 6)
        self.L.flipvh_set_particles(self.h, a.ctypes.data_as(fp), len(a))

    def solid_sdf(self):
        a = np.empty((self.K + 1, self.J + 1, self.I + 1), np.float32)
        self.L.flipvh_get_solid_sdf(self.h, a.ctypes.data_as(fp))
        return a

    def saveState(self, path):
        if self.L.flipvh_save_state(self.h, str(path).encode()) != 0:
            raise IOError("saveState: cannot write %s" % path)

    def loadState(self, path):
        if self.L.flipvh_load_state(self.h, str(path).encode()) != 0:
            raise IOError("loadState: cannot read %s" % path)
        d = (C.c_int * 3)()
        self.L.flipvh_get_dims(self.h, d)
        self.I, self.J, self.K = d[0], d[1], d[2]

    def advance(self, dt):
        st = capi.Stats()
        self.L.flipvh_advance(self.h, dt, C.byref(st))
        return st.as_dict()

    update = advance  # BASELINE.json north_star naming
