"""Size-independent properties at BASELINE.json's full size (256^3 bunny drop, the bench workload), where the oracle
would take minutes per substep: binned scatters = un-binned scatters, P2G reproduces a uniform field, the projected
velocity field is discretely divergence-free to the solver tolerance, the viscosity step obeys its cap/acceptance rule
and both precisions agree."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = 256


@pytest.fixture(scope="module")
def scene():
    from bench import build_scene
    return build_scene(N, 5.0)


def ctx(scene, **params):
    from flipviscosity3d_amd import capi
    dx, solid, P = scene
    c = capi.Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(5.0)
    if params:
        c.set_params(**params)
    c.particles = P
    return c


def test_fullsize_scatters(scene):
    dx, solid, P = scene
    Q = P.copy()
    Q[:, 3:] = np.array([0.25, -1.5, 0.75], np.float32)     # uniform particle velocity
    out = []
    for unbinned in (0, 1):
        c = ctx(scene)
        c.set_params(unbinned_scatter=unbinned)
        c.particles = Q
        c.particle_sdf()
        phi = c.grid("LIQUID_PHI")
        c.p2g()
        out.append((phi, [c.grid(n) for n in "UVW"], [c.grid("VALID_" + n) for n in "UVW"]))
        c.close()
    assert np.array_equal(out[0][0], out[1][0])                        # particle SDF: min is order-free
    assert (out[0][0] < 0).sum() > 500000
    for (a, b, va, vb, want) in zip(out[0][1], out[1][1], out[0][2], out[1][2], (0.25, -1.5, 0.75)):
        assert np.array_equal(va, vb)
        m = va != 0
        assert m.sum() > 500000
        # a weighted mean of equal values is that value (to rounding); both scatter kernels
        assert np.abs(a[m] - want).max() <= 2e-6 * abs(want) and np.abs(b[m] - want).max() <= 2e-6 * abs(want)
        assert not a[~m].any()


@pytest.mark.parametrize("precision", [0, 1])
def test_fullsize_projection_is_divergence_free(scene, precision):
    """After _project the weighted divergence of every pressure cell equals the PCG residual of the pressure solve
    (pressuresolver.cpp:227-243 defines b = -div; fluidsimulation.cpp:611-657 subtracts the gradient the matrix encodes)."""
    dx, solid, P = scene
    c = ctx(scene, precision=precision, pressure_rel_tolerance=1e-6 if precision == 0 else 0.0)
    st = c.substep(0.01)
    assert st["pressure"]["status"] == 0
    # the default viscosity solve converges inside the reference's cap: fp32 vectors -- the multigrid-preconditioned two-stage solve; precision = FP64 -- the same
    # loops refined until the FP64 residual on the reference's operator meets 1e-6 max|rhs| or a further stage stops paying (status 1 then; until round 4 fp64
    # vectors took the diagonal and stopped at the cap here)
    v = st["viscosity"]
    print("256^3, precision %d: %d viscosity iterations (%d in correction stages), status %d / %d, fp64 residual %.2e rhs" % (
        precision, v["iterations"], v["correction_iterations"], v["status"], v["correction_status"], v["defect_residual"] / v["rhs_norm"]))
    if precision == 0:
        assert v["status"] == 0 and v["preconditioner"] == 1 and v["iterations"] < 700, v
    else:
        assert v["preconditioner"] == 1 and v["iterations"] < 700 and v["status"] in (0, 1), v
        assert 0.0 < v["defect_residual"] <= 1e-5 * v["rhs_norm"], v
        assert v["status"] == (0 if v["defect_residual"] <= 1.0000001e-6 * v["rhs_norm"] else 1), v
    U, V, W = (c.grid(n).astype(np.float64) for n in "UVW")
    wU, wV, wW = (c.grid("WEIGHT_" + n).astype(np.float64) for n in "UVW")
    phi = c.grid("LIQUID_PHI")
    div = ((wU * U)[:, :, 1:] - (wU * U)[:, :, :-1] + (wV * V)[:, 1:, :] - (wV * V)[:, :-1, :] + (wW * W)[1:, :, :] - (wW * W)[:-1, :, :]) / dx
    cells = np.zeros_like(phi, bool)
    cells[1:-1, 1:-1, 1:-1] = phi[1:-1, 1:-1, 1:-1] < 0               # pressure cells (pressuresolver.cpp:196-225)
    open_ = (wU[:, :, 1:] + wU[:, :, :-1] + wV[:, 1:, :] + wV[:, :-1, :] + wW[1:, :, :] + wW[:-1, :, :]) > 0
    cells &= open_
    assert cells.sum() == st["pressure"]["rows"]
    tol = max(1e-9, (1e-6 if precision == 0 else 0.0) * st["pressure"]["rhs_norm"])
    assert st["pressure"]["residual"] <= tol
    # fp32 velocities: the update u -= dt grad p / (dx theta) rounds each face to 1 ulp of |u| ~ 0.1
    slack = 2e-8 / dx * 6 if precision == 0 else 0.0
    slack += 2e-8 / dx * 6   # the grids are stored in fp32 in both modes
    assert np.abs(div[cells]).max() <= st["pressure"]["residual"] * 1.01 + slack, (np.abs(div[cells]).max(), st["pressure"])
    c.close()


def test_fullsize_precisions_agree_on_pressure_only_step(scene):
    """viscosity off (fluidsimulation.cpp:171-184): everything converges, so fp32 and fp64 vectors must give the same
    velocities to the north_star tolerance at full size"""
    from flipviscosity3d_amd import capi
    dx, solid, P = scene
    res = []
    for precision in (0, 1):
        c = capi.Context(N, N, N, dx)
        c.set_solid_sdf(solid)
        c.set_viscosity(0.0)
        c.set_params(precision=precision, pressure_rel_tolerance=1e-7 if precision == 0 else 0.0)
        Q = P.copy()
        Q[:, 3] = 0.3 * np.sin(6.0 * Q[:, 1])                        # a shear so that the projection has work to do
        c.particles = Q
        # ONE substep: a second one would start from particle positions that differ by 1e-7 between the two runs, and a face
        # whose last particle leaves (a discrete decision) then differs by 1e-4 locally -- sensitivity of the method, not
        # of the kernels (tests/test_gpu_wide.py); seen once in ~10 runs
        st = c.substep(0.01)
        assert st["viscosity"]["status"] == 3 and st["pressure"]["status"] == 0
        res.append([c.grid(n) for n in "UVW"])
        c.close()
    scale = max(np.abs(g).max() for g in res[1])
    assert max(np.abs(a - b).max() for a, b in zip(*res)) <= 1e-4 * scale


def test_auto_preconditioner_over_the_drop_and_splash(scene):
    """70 substeps of the bench scene with the default (AUTO) viscosity preconditioner: the multigrid from the first solve on, the
    liquid moving all the while, i.e. every solve assembles its hierarchy next to the leftovers of one assembled several cells away.
    EVERY solve of a default run must converge (status 0), and in the number of iterations a clean hierarchy needs (<= 200 in the stiff
    start, 50-120 once the liquid has hit the wall).  (A Galerkin gather that read children outside the finer level's
    current box -- another solve's rows -- went from 63 to 326 iterations at substep 50 of `tools/soak.py 256 150 auto` and into the
    diagonal fallback a few substeps later; that failure depends on exactly when AUTO switches and does not reproduce on every
    trajectory, so this test is the guard for the whole mechanism rather than a reproducer of that one bug.)"""
    c = ctx(scene)
    mg = []
    for t in range(70):
        st = c.substep(min(c.cfl(), 0.01))
        v = st["viscosity"]
        assert st["rc"] in (0, 1), (t, st["rc"], v)
        assert v["status"] == 0, (t, v)                        # a default run never returns an iterate stopped at the cap
        if v["preconditioner"] == 1:
            mg.append(v["iterations"])
    Q = c.particles
    c.close()
    assert np.isfinite(Q).all()
    assert len(mg) >= 60, len(mg)                            # the multigrid runs (nearly) every solve
    # (round 5: from the impact on the solves carry the velocity criterion and the mass scale -- 50-120 iterations where round 4's rule took 33-64 and left 3e-4 ... 9e-4 of
    # max|u| on tens of thousands of faces, profiles/r5/eta_scan_256.log; the polluted hierarchy's 326 stays far outside)
    assert max(mg) <= 260 and max(mg[30:]) <= 170, mg


def test_config4_honey_buckling_at_512_properties():
    """BASELINE configs[3] at its REAL size on one GPU (512^3, rod.ply + sheet.ply added with two add-liquid calls, nu = 50; 45 GiB of the
    device's 288): the oracle would need hours, so size-independent properties over two default-parameter substeps --
    the seeded particle count, a viscosity solve that converges inside the reference's cap (the multigrid: nu dt/dx^2 = 131 072), a
    projection whose weighted divergence equals the pressure residual in every pressure cell, particles that stay finite and inside."""
    from bench import build_workload
    from flipviscosity3d_amd import capi
    M = 512
    I, J, K, dx, solid, P = build_workload("honey", M, on_device=True)
    assert (I, J, K) == (M, M, M)
    assert len(P) == 14528382                                   # 8 counter-jittered samples per cell inside rod + sheet (deterministic)
    c = capi.Context(I, J, K, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(50.0)
    c.particles = P
    del P
    for t in range(2):
        st = c.substep(min(c.cfl(), 0.01))
        v, p = st["viscosity"], st["pressure"]
        assert v["status"] == 0 and v["preconditioner"] == 1 and v["layout"] == 2 and v["iterations"] < 700, v
        assert v["residual"] <= 3e-3 * v["rhs_norm"] * 1.0001 and v["defect_residual"] > 0.0   # stage 1 of the two-stage solve stops at 3 000 x the tolerance; a stage 2 ran (include/flipv.h: exact_viscosity_operator)
        assert p["status"] == 0, p
    U, V, W = (c.grid(n).astype(np.float64) for n in "UVW")
    wU, wV, wW = (c.grid("WEIGHT_" + n).astype(np.float64) for n in "UVW")
    fx = wU * U; fy = wV * V; fz = wW * W
    del U, V, W
    div = (fx[:, :, 1:] - fx[:, :, :-1] + fy[:, 1:, :] - fy[:, :-1, :] + fz[1:, :, :] - fz[:-1, :, :]) / dx
    del fx, fy, fz
    phi = c.grid("LIQUID_PHI")
    cells = np.zeros_like(phi, bool)
    cells[1:-1, 1:-1, 1:-1] = phi[1:-1, 1:-1, 1:-1] < 0
    cells &= (wU[:, :, 1:] + wU[:, :, :-1] + wV[:, 1:, :] + wV[:, :-1, :] + wW[1:, :, :] + wW[:-1, :, :]) > 0
    assert cells.sum() == p["rows"]
    assert np.abs(div[cells]).max() <= p["residual"] * 1.01 + 4 * 2e-8 / dx * 6, (np.abs(div[cells]).max(), p)
    Q = c.particles
    c.close()
    assert len(Q) == 14528382 and np.isfinite(Q).all()
    assert Q[:, :3].min() > dx and Q[:, :3].max() < 1.0 - dx      # clamped into the domain inset (fluidsimulation.cpp:319-320)
