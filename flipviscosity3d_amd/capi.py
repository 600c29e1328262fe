"""ctypes binding of the C-ABI in include/flipv.h (libflipv.so, HIP / gfx950).

This is plumbing only: every call goes straight to the hand-written HIP library.  There is no
Python or CPU implementation of any operator here; if the shared library is missing or no HIP
device is visible the calls fail loudly (FlipvError / OSError).
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLIPV_LIB") or os.path.join(_PKG, "libflipv.so")   # FLIPV_LIB: A/B builds (tools/ab_lib.sh)

GRID_IDS = dict(U=0, V=1, W=2, SAVED_U=3, SAVED_V=4, SAVED_W=5, VALID_U=6, VALID_V=7, VALID_W=8,
                LIQUID_PHI=9, SOLID_PHI=10, WEIGHT_U=11, WEIGHT_V=12, WEIGHT_W=13, VISCOSITY=14,
                PRESSURE=15)
PHASES = ["sdf", "p2g", "bodyforce", "viscosity", "project", "constrain", "advect"]
VOLUME_IDS = dict(center=0, U=1, V=2, W=3, edgeU=4, edgeV=5, edgeW=6)

# every symbol include/flipv.h declares (tests/test_abi.py checks the header against this and the .so)
SYMBOLS = [
    "flipv_create", "flipv_create_on_device", "flipv_create_slab", "flipv_slab_range", "flipv_create_block", "flipv_block_range", "flipv_create_setup",
    "flipv_destroy", "flipv_last_error", "flipv_device_name",
    "flipv_abi_version", "flipv_default_params", "flipv_set_params", "flipv_get_params", "flipv_default_debug_params", "flipv_set_debug_params",
    "flipv_get_debug_params", "flipv_set_gravity",
    "flipv_set_solid_sdf", "flipv_set_viscosity_uniform", "flipv_set_viscosity",
    "flipv_upload_particles", "flipv_download_particles", "flipv_num_particles",
    "flipv_grid_elements", "flipv_read_grid", "flipv_write_grid", "flipv_grid_box", "flipv_read_grid_box", "flipv_write_grid_box", "flipv_read_grid_region",
    "flipv_cfl", "flipv_particle_sdf", "flipv_p2g", "flipv_extrapolate", "flipv_save_velocity",
    "flipv_advect_velocity_field", "flipv_body_force", "flipv_viscosity_solve", "flipv_compute_weights",
    "flipv_pressure_solve", "flipv_apply_pressure", "flipv_constrain", "flipv_update_particle_velocities",
    "flipv_advect_particles", "flipv_read_viscosity_volume", "flipv_substep", "flipv_advance",
    "flipv_kernel_stats_reset", "flipv_kernel_stats_get", "flipv_synchronize", "flipv_bench_spmv",
    "flipv_bench_copy", "flipv_bench_stream",
    "flipv_mesh_level_set", "flipv_add_boundary_mesh", "flipv_reset_boundary", "flipv_add_liquid_mesh",
    "flipv_comm_unique_id_bytes", "flipv_comm_get_unique_id", "flipv_comm_init_rccl", "flipv_comm_init_local",
    "flipv_comm_init_rccl_grid", "flipv_comm_init_local_grid", "flipv_comm_init_host_grid", "flipv_comm_finalize",
]


class FlipvError(RuntimeError):
    pass


class Params(C.Structure):
    """flipv_params (include/flipv.h, FLIPV_VERSION 6)"""
    _fields_ = [("cfl_number", C.c_float), ("min_frac", C.c_float), ("pic_ratio", C.c_float),
                ("extrapolation_layers", C.c_int), ("pressure_tolerance", C.c_double),
                ("pressure_rel_tolerance", C.c_double), ("pressure_max_iterations", C.c_int),
                ("viscosity_tolerance", C.c_double), ("viscosity_max_iterations", C.c_int),
                ("viscosity_accept_tolerance", C.c_double), ("precision", C.c_int), ("check_every", C.c_int),
                ("pressure_preconditioner", C.c_int), ("viscosity_preconditioner", C.c_int),
                ("exact_viscosity_operator", C.c_int), ("viscosity_layout", C.c_int),
                ("multigrid_rank_local", C.c_int), ("multigrid_distributed_levels", C.c_int), ("verbose", C.c_int),
                ("viscosity_stage1_factor", C.c_float), ("viscosity_stage2_factor", C.c_float), ("viscosity_stage2_max_iterations", C.c_int),
                ("viscosity_stage2_rounds", C.c_int), ("viscosity_two_stage_max_stiffness", C.c_float),
                ("viscosity_defect_predictor", C.c_int), ("viscosity_velocity_tolerance", C.c_float), ("viscosity_velocity_window", C.c_int),
                ("viscosity_mass_scale", C.c_float), ("viscosity_mass_floor", C.c_float), ("viscosity_massless_polish", C.c_int), ("viscosity_velocity_stall_ratio", C.c_float), ("viscosity_pair_correction", C.c_int)]


class DebugParams(C.Structure):
    """flipv_debug_params: A/B, profiling and test switches"""
    _fields_ = [("kernel_timing", C.c_int), ("tile_rows", C.c_int), ("viscosity_mg_coarsest_sweeps", C.c_int), ("viscosity_mg_min_dim", C.c_int),
                ("pressure_mg_coarsest_sweeps", C.c_int), ("pressure_mg_omega", C.c_float), ("pressure_mg_overcorrection", C.c_float),
                ("viscosity_mg_omega_first", C.c_float), ("viscosity_mg_omega_second", C.c_float),
                ("no_liquid_box", C.c_int), ("no_comm_overlap", C.c_int), ("no_graph_replay", C.c_int), ("unbinned_scatter", C.c_int), ("grid_cap", C.c_int),
                ("viscosity_lane_width", C.c_int), ("viscosity_spmv_grid_cap", C.c_int), ("viscosity_update_grid_cap", C.c_int),
                ("beta_from_conjugacy", C.c_int), ("spmv_run_length", C.c_int), ("viscosity_mg_packed_rows", C.c_int), ("stall_guard_ratio", C.c_float), ("viscosity_pair_lambda_floor", C.c_float), ("velocity_patience", C.c_int)]


_PRODUCT_FIELDS = {f for f, _ in Params._fields_}
_DEBUG_FIELDS = {f for f, _ in DebugParams._fields_}


class AllParams:
    """both structs behind one attribute namespace (what Context.get_params returns)"""
    def __init__(self, p, d):
        self.product, self.debug = p, d

    def __getattr__(self, k):
        if k in _PRODUCT_FIELDS:
            return getattr(self.product, k)
        if k in _DEBUG_FIELDS:
            return getattr(self.debug, k)
        raise AttributeError(k)


LAYOUT_AUTO, LAYOUT_PLAIN, LAYOUT_SWIZZLED, LAYOUT_BRICK = 0, 1, 2, 3
PRECOND_AUTO, PRECOND_DIAGONAL, PRECOND_MULTIGRID = 0, 1, 2


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int), ("residual", C.c_double), ("rhs_norm", C.c_double),
                ("status", C.c_int), ("rows", C.c_int), ("active_tiles", C.c_int), ("total_tiles", C.c_int),
                ("preconditioner", C.c_int), ("layout", C.c_int), ("refinements", C.c_int), ("defect_residual", C.c_double),
                ("correction_iterations", C.c_int), ("comm_bytes_setup", C.c_double), ("comm_bytes_per_iteration", C.c_double), ("velocity_step", C.c_double),
                ("halo_exchanges_per_iteration", C.c_int), ("allreduces_per_iteration", C.c_int), ("correction_status", C.c_int), ("eliminated_rows", C.c_int), ("massless_cluster_edges", C.c_int)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [("phase_ms", C.c_double * 7), ("total_ms", C.c_double), ("dt", C.c_float),
                ("substeps", C.c_int), ("viscosity", SolveInfo), ("pressure", SolveInfo)]

    def as_dict(self):
        return dict(phase_ms=dict(zip(PHASES, list(self.phase_ms))), total_ms=self.total_ms, dt=self.dt,
                    substeps=self.substeps, viscosity=self.viscosity.as_dict(), pressure=self.pressure.as_dict())


_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t))
_REDUCE64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)
_REDUCE32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.c_size_t)
_BARRIER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class HostComm(C.Structure):
    """flipv_host_comm: the callbacks of the host-staged communicator"""
    _fields_ = [("user", C.c_void_p), ("exchange", _EXCHANGE_FN), ("allreduce_sum_f64", _REDUCE64_FN), ("allreduce_sum_f32", _REDUCE32_FN), ("barrier", _BARRIER_FN)]


def torch_distributed_callbacks(dist):
    """flipv_host_comm over an initialised torch.distributed process group of CPU tensors (gloo): plumbing for the multi-process rehearsal on one GPU
    (bench.py --comm host, tests/test_gpu_multiprocess.py).  Returns the struct; the caller keeps it alive as long as the communicator."""
    import sys
    import traceback
    import torch

    def view(ptr, nbytes):
        return torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(nbytes,)))

    def exchange(user, n, peer, sendbuf, sbytes, recvbuf, rbytes):
        try:
            seen, reqs = {}, []
            for m in range(n):   # my m-th operation towards a peer pairs with its m-th towards me: the tag is that count
                tag = seen.get(peer[m], 0)
                seen[peer[m]] = tag + 1
                if rbytes[m]:
                    reqs.append(dist.irecv(view(recvbuf[m], rbytes[m]), src=peer[m], tag=tag))
                if sbytes[m]:
                    reqs.append(dist.isend(view(sendbuf[m], sbytes[m]), dst=peer[m], tag=tag))
            for r in reqs:
                r.wait()
            return 0
        except Exception:   # noqa: BLE001  (an exception must not unwind through the C frames)
            traceback.print_exc(file=sys.stderr)
            return 1

    def reduce64(user, values, n):
        try:
            dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(values, shape=(n,))))
            return 0
        except Exception:   # noqa: BLE001
            traceback.print_exc(file=sys.stderr)
            return 1

    def reduce32(user, values, n):
        try:
            dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(values, shape=(n,))))
            return 0
        except Exception:   # noqa: BLE001
            traceback.print_exc(file=sys.stderr)
            return 1

    def barrier(user):
        try:
            dist.barrier()
            return 0
        except Exception:   # noqa: BLE001
            traceback.print_exc(file=sys.stderr)
            return 1
    return HostComm(None, _EXCHANGE_FN(exchange), _REDUCE64_FN(reduce64), _REDUCE32_FN(reduce32), _BARRIER_FN(barrier))


class KernelStats(C.Structure):
    _fields_ = [("pressure_spmv_ms", C.c_double), ("pressure_spmv_launches", C.c_long),
                ("pressure_spmv_cells", C.c_double), ("viscosity_spmv_ms", C.c_double),
                ("viscosity_spmv_launches", C.c_long), ("viscosity_spmv_cells", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None
fp = C.POINTER(C.c_float)


def load():
    """Load libflipv.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libflipv.so not built: run `make -C flipviscosity3d_amd/csrc` "
                      "(or __graft_entry__.build()); there is no fallback implementation")
    L = C.CDLL(LIB_PATH)
    ctx = C.c_void_p
    L.flipv_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(ctx)]
    L.flipv_create_on_device.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(ctx)]
    L.flipv_create_slab.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.POINTER(ctx)]
    L.flipv_slab_range.argtypes = [ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    i3 = C.POINTER(C.c_int)
    L.flipv_create_block.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, i3, i3, C.POINTER(ctx)]
    L.flipv_block_range.argtypes = [ctx, i3, i3]
    L.flipv_create_setup.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(ctx)]
    L.flipv_grid_box.argtypes = [ctx, C.c_int, C.c_int, i3, i3]
    L.flipv_read_grid_box.argtypes = [ctx, C.c_int, fp]
    L.flipv_read_grid_region.argtypes = [ctx, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), fp]
    L.flipv_write_grid_box.argtypes = [ctx, C.c_int, fp]
    L.flipv_comm_init_rccl_grid.argtypes = [ctx, C.c_void_p, C.c_int, i3]
    L.flipv_comm_init_local_grid.argtypes = [C.POINTER(ctx), i3]
    L.flipv_comm_get_unique_id.argtypes = [C.c_void_p]
    L.flipv_comm_init_rccl.argtypes = [ctx, C.c_void_p, C.c_int, C.c_int]
    L.flipv_comm_init_local.argtypes = [C.POINTER(ctx), C.c_int]
    L.flipv_comm_init_host_grid.argtypes = [ctx, C.POINTER(HostComm), C.c_int, i3]
    L.flipv_comm_finalize.argtypes = [ctx]
    L.flipv_destroy.argtypes = [ctx]
    L.flipv_last_error.restype = C.c_char_p
    L.flipv_last_error.argtypes = [ctx]
    L.flipv_device_name.argtypes = [ctx, C.c_char_p, C.c_size_t]
    L.flipv_default_params.argtypes = [C.POINTER(Params)]
    L.flipv_set_params.argtypes = [ctx, C.POINTER(Params)]
    L.flipv_get_params.argtypes = [ctx, C.POINTER(Params)]
    L.flipv_set_debug_params.argtypes = [ctx, C.POINTER(DebugParams)]
    L.flipv_get_debug_params.argtypes = [ctx, C.POINTER(DebugParams)]
    L.flipv_default_debug_params.argtypes = [C.POINTER(DebugParams)]
    L.flipv_abi_version.restype = C.c_int
    L.flipv_set_gravity.argtypes = [ctx, C.c_float, C.c_float, C.c_float]
    L.flipv_set_solid_sdf.argtypes = [ctx, fp]
    L.flipv_set_viscosity_uniform.argtypes = [ctx, C.c_float]
    L.flipv_set_viscosity.argtypes = [ctx, fp]
    ip = C.POINTER(C.c_int)
    L.flipv_mesh_level_set.argtypes = [ctx, fp, C.c_size_t, ip, C.c_size_t, C.c_int, fp, ip]
    L.flipv_add_boundary_mesh.argtypes = [ctx, fp, C.c_size_t, ip, C.c_size_t, C.c_int]
    L.flipv_reset_boundary.argtypes = [ctx]
    L.flipv_add_liquid_mesh.argtypes = [ctx, fp, C.c_size_t, ip, C.c_size_t, C.c_ulonglong, C.POINTER(C.c_size_t)]
    L.flipv_upload_particles.argtypes = [ctx, fp, C.c_size_t]
    L.flipv_download_particles.argtypes = [ctx, fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flipv_num_particles.restype = C.c_size_t
    L.flipv_num_particles.argtypes = [ctx]
    L.flipv_grid_elements.restype = C.c_size_t
    L.flipv_grid_elements.argtypes = [ctx, C.c_int]
    L.flipv_read_grid.argtypes = [ctx, C.c_int, fp]
    L.flipv_write_grid.argtypes = [ctx, C.c_int, fp]
    L.flipv_cfl.argtypes = [ctx, fp]
    for n in ("flipv_particle_sdf", "flipv_p2g", "flipv_extrapolate", "flipv_save_velocity",
              "flipv_advect_velocity_field", "flipv_compute_weights", "flipv_constrain",
              "flipv_update_particle_velocities", "flipv_kernel_stats_reset", "flipv_synchronize"):
        getattr(L, n).argtypes = [ctx]
    for n in ("flipv_body_force", "flipv_apply_pressure", "flipv_advect_particles"):
        getattr(L, n).argtypes = [ctx, C.c_float]
    L.flipv_viscosity_solve.argtypes = [ctx, C.c_float, C.POINTER(SolveInfo)]
    L.flipv_pressure_solve.argtypes = [ctx, C.c_float, C.POINTER(SolveInfo)]
    L.flipv_read_viscosity_volume.argtypes = [ctx, C.c_int, fp]
    L.flipv_substep.argtypes = [ctx, C.c_float, C.POINTER(Stats)]
    L.flipv_advance.argtypes = [ctx, C.c_float, C.POINTER(Stats)]
    L.flipv_kernel_stats_get.argtypes = [ctx, C.POINTER(KernelStats)]
    L.flipv_bench_spmv.argtypes = [ctx, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.flipv_bench_copy.argtypes = [ctx, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.flipv_bench_stream.argtypes = [ctx, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
    _lib = L
    return L


def grid_shape(name, I, J, K):
    """numpy shape (depth, height, width) of a grid in the reference's Array3d layout (x fastest)."""
    if name.endswith("_U") or name == "U":
        return (K, J, I + 1)
    if name.endswith("_V") or name == "V":
        return (K, J + 1, I)
    if name.endswith("_W") or name == "W":
        return (K + 1, J, I)
    if name in ("SOLID_PHI", "VISCOSITY"):
        return (K + 1, J + 1, I + 1)
    return (K, J, I)


def volume_shape(name, I, J, K):
    return dict(center=(K, J, I), U=(K, J, I + 1), V=(K, J + 1, I), W=(K + 1, J, I), edgeU=(K + 1, J + 1, I),
                edgeV=(K + 1, J, I + 1), edgeW=(K, J + 1, I + 1))[name]


def _F(a):
    return a.ctypes.data_as(fp)


class Context:
    """Owns one flipv_context (device state of one simulation / one rank)."""

    def __init__(self, I, J, K, dx, device=None, slab=None, block=None, setup_only=False):
        """slab = (k_begin, k_end): one rank of a slab decomposition along k (global grid I x J x K);
        block = ((i0, j0, k0), (i1, j1, k1)): one rank of a block decomposition, cells [lo, hi);
        setup_only: a light single-domain context for the scene-setup entry points (flipv_create_setup)."""
        self.L = load()
        self.I, self.J, self.K = int(I), int(J), int(K)
        self.dx = float(np.float32(dx))
        self.slab = slab
        self.block = block
        h = C.c_void_p()
        if setup_only:
            rc = self.L.flipv_create_setup(self.I, self.J, self.K, C.c_float(dx), int(device or 0), C.byref(h))
        elif block is not None:
            lo = (C.c_int * 3)(*[int(v) for v in block[0]])
            hi = (C.c_int * 3)(*[int(v) for v in block[1]])
            rc = self.L.flipv_create_block(self.I, self.J, self.K, C.c_float(dx), int(device or 0), lo, hi, C.byref(h))
        elif slab is not None:
            rc = self.L.flipv_create_slab(self.I, self.J, self.K, C.c_float(dx), int(device or 0), int(slab[0]), int(slab[1]),
                                          C.byref(h))
        elif device is None:
            rc = self.L.flipv_create(self.I, self.J, self.K, C.c_float(dx), C.byref(h))
        else:
            rc = self.L.flipv_create_on_device(self.I, self.J, self.K, C.c_float(dx), int(device), C.byref(h))
        if rc != 0:
            raise FlipvError("flipv_create failed (%d): %s" % (rc, self.L.flipv_last_error(None).decode()))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.flipv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise FlipvError("%s failed (%d): %s" % (what, rc, self.L.flipv_last_error(self.h).decode()))
        return rc

    # ---- scene setup on the device
    @staticmethod
    def _mesh(mesh):
        v = np.ascontiguousarray(mesh[0], np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(mesh[1], np.int32).reshape(-1, 3)
        return v, t, v.ctypes.data_as(C.POINTER(C.c_float)), len(v), t.ctypes.data_as(C.POINTER(C.c_int)), len(t)

    def mesh_level_set(self, mesh, band=3, want_closest=False):
        """MeshLevelSet::calculateSignedDistanceField on the device -> phi (K+1,J+1,I+1) [, closest]."""
        v, t, vp, nv, tp, nt = self._mesh(mesh)
        phi = np.empty((self.K + 1, self.J + 1, self.I + 1), np.float32)
        clo = np.empty((self.K + 1, self.J + 1, self.I + 1), np.int32) if want_closest else None
        self._chk(self.L.flipv_mesh_level_set(self.h, vp, nv, tp, nt, band, phi.ctypes.data_as(C.POINTER(C.c_float)),
                                              clo.ctypes.data_as(C.POINTER(C.c_int)) if want_closest else None),
                  "flipv_mesh_level_set")
        return (phi, clo) if want_closest else phi

    def add_boundary_mesh(self, mesh, inverted=False):
        v, t, vp, nv, tp, nt = self._mesh(mesh)
        self._chk(self.L.flipv_add_boundary_mesh(self.h, vp, nv, tp, nt, int(bool(inverted))), "flipv_add_boundary_mesh")

    def reset_boundary(self):
        self._chk(self.L.flipv_reset_boundary(self.h), "flipv_reset_boundary")

    def add_liquid_mesh(self, mesh, seed=0):
        v, t, vp, nv, tp, nt = self._mesh(mesh)
        added = C.c_size_t()
        self._chk(self.L.flipv_add_liquid_mesh(self.h, vp, nv, tp, nt, seed, C.byref(added)), "flipv_add_liquid_mesh")
        return added.value

    # ---- multi-GPU
    def comm_init_rccl(self, unique_id, rank, nranks, dims=None):
        """dims = (ranks along i, j, k) of a block decomposition (rank = x + dims[0] * (y + dims[1] * z)); None: slabs along k"""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        if dims is None:
            self._chk(self.L.flipv_comm_init_rccl(self.h, buf, rank, nranks), "flipv_comm_init_rccl")
        else:
            assert int(np.prod(dims)) == nranks
            self._chk(self.L.flipv_comm_init_rccl_grid(self.h, buf, rank, (C.c_int * 3)(*[int(d) for d in dims])), "flipv_comm_init_rccl_grid")

    def block_range(self):
        lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
        self._chk(self.L.flipv_block_range(self.h, lo, hi), "flipv_block_range")
        return tuple(lo), tuple(hi)

    def comm_init_host(self, callbacks, rank, dims):
        """the host-staged communicator (one process per rank, any number of ranks per device): callbacks = HostComm, e.g. torch_distributed_callbacks(dist)"""
        self._host_comm = callbacks     # the C side keeps the function pointers: keep the Python objects alive
        self._chk(self.L.flipv_comm_init_host_grid(self.h, C.byref(callbacks), int(rank), (C.c_int * 3)(*[int(d) for d in dims])), "flipv_comm_init_host_grid")

    def comm_finalize(self):
        self._chk(self.L.flipv_comm_finalize(self.h), "flipv_comm_finalize")

    # ---- configuration
    def device_name(self):
        buf = C.create_string_buffer(256)
        self._chk(self.L.flipv_device_name(self.h, buf, 256), "flipv_device_name")
        return buf.value.decode()

    def get_params(self):
        p, d = Params(), DebugParams()
        self._chk(self.L.flipv_get_params(self.h, C.byref(p)), "flipv_get_params")
        self._chk(self.L.flipv_get_debug_params(self.h, C.byref(d)), "flipv_get_debug_params")
        return AllParams(p, d)

    def set_params(self, **kw):
        """fields of flipv_params and of flipv_debug_params alike: each goes to its own struct"""
        a = self.get_params()
        touched = set()
        for k, v in kw.items():
            if k in _PRODUCT_FIELDS:
                setattr(a.product, k, v); touched.add("p")
            elif k in _DEBUG_FIELDS:
                setattr(a.debug, k, v); touched.add("d")
            else:
                raise AttributeError(k)
        if "p" in touched:
            self._chk(self.L.flipv_set_params(self.h, C.byref(a.product)), "flipv_set_params")
        if "d" in touched:
            self._chk(self.L.flipv_set_debug_params(self.h, C.byref(a.debug)), "flipv_set_debug_params")

    def set_gravity(self, gx, gy, gz):
        self._chk(self.L.flipv_set_gravity(self.h, gx, gy, gz), "flipv_set_gravity")

    def set_solid_sdf(self, nodes):
        a = np.ascontiguousarray(nodes, np.float32)
        assert a.shape == grid_shape("SOLID_PHI", self.I, self.J, self.K), a.shape
        self._chk(self.L.flipv_set_solid_sdf(self.h, _F(a)), "flipv_set_solid_sdf")

    def set_viscosity(self, v):
        if np.isscalar(v):
            self._chk(self.L.flipv_set_viscosity_uniform(self.h, float(v)), "flipv_set_viscosity_uniform")
        else:
            a = np.ascontiguousarray(v, np.float32)
            assert a.shape == grid_shape("VISCOSITY", self.I, self.J, self.K), a.shape
            self._chk(self.L.flipv_set_viscosity(self.h, _F(a)), "flipv_set_viscosity")

    # ---- particles
    @property
    def particles(self):
        n = self.L.flipv_num_particles(self.h)
        a = np.empty((n, 6), np.float32)
        got = C.c_size_t()
        self._chk(self.L.flipv_download_particles(self.h, _F(a), n, C.byref(got)), "flipv_download_particles")
        return a

    @particles.setter
    def particles(self, a):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 6)
        self._chk(self.L.flipv_upload_particles(self.h, _F(a), len(a)), "flipv_upload_particles")

    @property
    def num_particles(self):
        return self.L.flipv_num_particles(self.h)

    # ---- grids
    def grid(self, name, out=None):
        """Full-size grid.  On a block context only the entries the rank owns are written: pass the same `out` to every
        rank in turn to assemble the global grid (a fresh array is zero-filled first)."""
        shp = grid_shape(name, self.I, self.J, self.K)
        if out is None:
            a = np.zeros(shp, np.float32) if self.block is not None or self.slab is not None else np.empty(shp, np.float32)
        else:
            a = out
            assert a.shape == shp and a.dtype == np.float32 and a.flags.c_contiguous
        self._chk(self.L.flipv_read_grid(self.h, GRID_IDS[name], _F(a)), "flipv_read_grid")
        return a

    def grid_box(self, name, kind=0):
        """(lo, hi) of the part of grid `name` this rank owns (kind 0) / allocates (kind 1), global indices (i, j, k)"""
        lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
        self._chk(self.L.flipv_grid_box(self.h, GRID_IDS[name], kind, lo, hi), "flipv_grid_box")
        return tuple(lo), tuple(hi)

    def read_box(self, name):
        """the owned part of the grid, box-shaped (numpy (depth, height, width) of the box)"""
        lo, hi = self.grid_box(name, 0)
        a = np.empty((hi[2] - lo[2], hi[1] - lo[1], hi[0] - lo[0]), np.float32)
        self._chk(self.L.flipv_read_grid_box(self.h, GRID_IDS[name], _F(a)), "flipv_read_grid_box")
        return a

    def read_region(self, name, lo, hi):
        """any box [lo, hi) of global indices (i, j, k) inside what the context allocates of the grid, box-shaped"""
        a = np.empty((hi[2] - lo[2], hi[1] - lo[1], hi[0] - lo[0]), np.float32)
        self._chk(self.L.flipv_read_grid_region(self.h, GRID_IDS[name], (C.c_int * 3)(*lo), (C.c_int * 3)(*hi), _F(a)), "flipv_read_grid_region")
        return a

    def write_box(self, name, a):
        """the allocated part (owned + halo) of the grid, box-shaped"""
        lo, hi = self.grid_box(name, 1)
        a = np.ascontiguousarray(a, np.float32)
        assert a.shape == (hi[2] - lo[2], hi[1] - lo[1], hi[0] - lo[0]), (a.shape, lo, hi)
        self._chk(self.L.flipv_write_grid_box(self.h, GRID_IDS[name], _F(a)), "flipv_write_grid_box")

    def set_grid(self, name, a):
        a = np.ascontiguousarray(a, np.float32)
        assert a.shape == grid_shape(name, self.I, self.J, self.K), (name, a.shape)
        self._chk(self.L.flipv_write_grid(self.h, GRID_IDS[name], _F(a)), "flipv_write_grid")

    def viscosity_volume(self, name):
        a = np.empty(volume_shape(name, self.I, self.J, self.K), np.float32)
        self._chk(self.L.flipv_read_viscosity_volume(self.h, VOLUME_IDS[name], _F(a)), "flipv_read_viscosity_volume")
        return a

    # ---- operators
    def cfl(self):
        v = C.c_float()
        self._chk(self.L.flipv_cfl(self.h, C.byref(v)), "flipv_cfl")
        return v.value

    def particle_sdf(self):
        self._chk(self.L.flipv_particle_sdf(self.h), "flipv_particle_sdf")

    def p2g(self):
        self._chk(self.L.flipv_p2g(self.h), "flipv_p2g")

    def extrapolate(self):
        self._chk(self.L.flipv_extrapolate(self.h), "flipv_extrapolate")

    def save_velocity(self):
        self._chk(self.L.flipv_save_velocity(self.h), "flipv_save_velocity")

    def advect_velocity_field(self):
        self._chk(self.L.flipv_advect_velocity_field(self.h), "flipv_advect_velocity_field")

    def body_force(self, dt):
        self._chk(self.L.flipv_body_force(self.h, dt), "flipv_body_force")

    def viscosity_solve(self, dt):
        info = SolveInfo()
        rc = self._chk(self.L.flipv_viscosity_solve(self.h, dt, C.byref(info)), "flipv_viscosity_solve")
        d = info.as_dict()
        d["rc"] = rc
        return d

    def compute_weights(self):
        self._chk(self.L.flipv_compute_weights(self.h), "flipv_compute_weights")

    def pressure_solve(self, dt):
        info = SolveInfo()
        rc = self._chk(self.L.flipv_pressure_solve(self.h, dt, C.byref(info)), "flipv_pressure_solve")
        d = info.as_dict()
        d["rc"] = rc
        return d

    def apply_pressure(self, dt):
        self._chk(self.L.flipv_apply_pressure(self.h, dt), "flipv_apply_pressure")

    def constrain(self):
        self._chk(self.L.flipv_constrain(self.h), "flipv_constrain")

    def update_particle_velocities(self):
        self._chk(self.L.flipv_update_particle_velocities(self.h), "flipv_update_particle_velocities")

    def advect_particles(self, dt):
        self._chk(self.L.flipv_advect_particles(self.h, dt), "flipv_advect_particles")

    def substep(self, dt):
        st = Stats()
        rc = self._chk(self.L.flipv_substep(self.h, dt, C.byref(st)), "flipv_substep")
        d = st.as_dict()
        d["rc"] = rc
        return d

    def advance(self, dt):
        st = Stats()
        rc = self._chk(self.L.flipv_advance(self.h, dt, C.byref(st)), "flipv_advance")
        d = st.as_dict()
        d["rc"] = rc
        return d

    # ---- measurement
    def kernel_stats_reset(self):
        self._chk(self.L.flipv_kernel_stats_reset(self.h), "flipv_kernel_stats_reset")

    def kernel_stats(self):
        ks = KernelStats()
        self._chk(self.L.flipv_kernel_stats_get(self.h, C.byref(ks)), "flipv_kernel_stats_get")
        return ks.as_dict()

    def synchronize(self):
        self._chk(self.L.flipv_synchronize(self.h), "flipv_synchronize")

    def bench_spmv(self, which, reps=50):
        ms, cells = C.c_double(), C.c_double()
        self._chk(self.L.flipv_bench_spmv(self.h, {0: 0, "pressure": 0, 1: 1, "viscosity": 1, 2: 2, "viscosity_mg": 2}[which], reps, C.byref(ms),
                                          C.byref(cells)), "flipv_bench_spmv")
        return ms.value, cells.value

    def bench_stream(self, mode, nbytes=1 << 30, reps=10):
        """plain streaming kernel: mode 0 read-only, 1 copy, 2 write-only; 3 / 4 / 5 / 6 = read / copy / five reads : one write / ten reads : three writes in the tuned form (flipv.h) -> GB/s of bytes moved"""
        g = C.c_double()
        self._chk(self.L.flipv_bench_stream(self.h, C.c_size_t(nbytes), reps, mode, C.byref(g)), "flipv_bench_stream")
        return g.value

    def bench_copy(self, nbytes=1 << 30, reps=10):
        g = C.c_double()
        self._chk(self.L.flipv_bench_copy(self.h, nbytes, reps, C.byref(g)), "flipv_bench_copy")
        return g.value


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0 only); returns 128 bytes to broadcast."""
    L = load()
    buf = C.create_string_buffer(128)
    rc = L.flipv_comm_get_unique_id(buf)
    if rc != 0:
        raise FlipvError("flipv_comm_get_unique_id failed (%d)" % rc)
    return buf.raw


def comm_init_local(contexts, dims=None):
    """Attach the in-process verification communicator to the N contexts of this process: slabs along k, or with
    dims = (ranks along i, j, k) a block decomposition (contexts in rank order, x fastest)."""
    L = load()
    arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
    if dims is None:
        rc = L.flipv_comm_init_local(arr, len(contexts))
    else:
        assert int(np.prod(dims)) == len(contexts)
        rc = L.flipv_comm_init_local_grid(arr, (C.c_int * 3)(*[int(d) for d in dims]))
    if rc != 0:
        msg = "; ".join(L.flipv_last_error(c.h).decode() for c in contexts if L.flipv_last_error(c.h))
        raise FlipvError("flipv_comm_init_local failed (%d): %s" % (rc, msg))
