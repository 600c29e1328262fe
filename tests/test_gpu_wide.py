"""Parity on a domain wider than one solver tile (256 indices in i): the liquid spans the tile boundary, so the
wave-edge exchange of the SpMV kernels (lane 0 / lane 63 fetch their i-neighbour from memory) and the tile lists with
more than one tile per row are exercised against the oracle.  The cubic fixtures are all narrower than a tile."""
import numpy as np
import pytest

from helpers import rel_maxnorm3

pytestmark = pytest.mark.gpu


def box_mesh(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y0, z1], [x0, y0, z1], [x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3], [4, 7, 6], [4, 6, 5], [0, 3, 7], [0, 7, 4], [1, 5, 6], [1, 6, 2], [0, 4, 5], [0, 5, 1], [3, 2, 6], [3, 6, 7]], np.int32)
    return v, t


@pytest.mark.parametrize("precision", [0, 1])
def test_wide_domain_substeps_match_oracle(oracle, precision):
    from flipviscosity3d_amd import capi, hostapi as H
    I, J, K = 288, 20, 24
    dx = float(np.float32(1.0 / I))
    s = H.FluidSimulation()
    s.initialize(I, J, K, dx)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 11)
    s.addLiquid(box_mesh((0.2, 3.2 * dx, 4.3 * dx), (0.955, 13.1 * dx, 19.6 * dx)))   # i = 57 .. 275: across i = 256
    solid, P = s.solid_sdf(), s.particles
    s.close()
    x = P[:, 0]
    P[:, 3] = 0.4 * np.sin(4 * np.pi * x)
    P[:, 4] = 0.12 * np.cos(6 * np.pi * x)
    P[:, 5] = 0.08 * np.sin(2 * np.pi * x)
    assert len(P) > 100000 and x.min() < 0.25 and x.max() > 0.93
    nu, dt = 2.0, 0.005
    c = capi.Context(I, J, K, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(nu)
    c.set_params(precision=precision, viscosity_max_iterations=20000, viscosity_tolerance=1e-7,
                 pressure_rel_tolerance=1e-7 if precision == 0 else 0.0)
    c.particles = P
    o = oracle.OracleSim(I, J, K, dx)
    o.set_solid(solid)
    o.set_viscosity(nu)
    o.set_solver_limits(vmaxiter=20000)
    o.particles = P
    for t in range(3):
        # every substep starts from the oracle's particles: a 1e-6 difference in a position can flip a discrete decision
        # (a cell centre entering the liquid, a face losing its last particle) and the two runs then differ locally by
        # 1e-3 one substep later -- sensitivity of the method, not of the kernels (measured: chained 1.9e-3, this way 1e-5)
        c.particles = o.particles
        st = c.substep(dt)
        sec, vi, pi = o.substep(dt)
        assert st["viscosity"]["status"] == 0 and vi["status"] == 0
        assert st["viscosity"]["rows"] + st["viscosity"]["eliminated_rows"] == vi["rows"]
        # the tile lists really have more than one tile per row
        assert st["viscosity"]["total_tiles"] >= 2 * ((J + 1 + 15) // 16) * (K + 1)
        got = [c.grid(n) for n in "UVW"]
        ref = [o.grid(n) for n in "UVW"]
        assert np.abs(ref[0][:, :, 250:265]).max() > 0        # liquid on both sides of the tile boundary
        assert rel_maxnorm3(got, ref) <= 1e-4, (t, rel_maxnorm3(got, ref), st["viscosity"], vi)
        assert np.array_equal(c.grid("LIQUID_PHI"), o.grid("LIQUID_PHI"))
        assert np.abs(c.particles[:, :3] - o.particles[:, :3]).max() <= 2e-7
    c.close()
    o.close()


@pytest.mark.parametrize("I,J,filled_rows", [(248, 20, 64), (288, 20, 64), (288, 31, 16)])
def test_tile_geometry_follows_the_liquid(I, J, filled_rows):
    """the solver tiles switch between the two geometries (csrc/pcg_geo.inc) as the liquid changes: a small blob in the
    wide domain is solved on 64 x 16 tiles, the filled domain on 256 x 4 tiles again -- where those fit the lattice: 249 indices
    are one 256-wide row; 289 idle 44 % of two, and the filled domain stays on the narrow tiles where THEIR lanes are better used (32 rows: two 16-row tiles,
    0.90 against 0.56; with 21 rows the second narrow tile is mostly outside, 0.59 against 0.56, and fv_build_tiles asks for 10 %) --,
    and a context that went through the switch gives the velocities of a fresh one"""
    from flipviscosity3d_amd import capi, hostapi as H
    K = 24
    dx = float(np.float32(1.0 / I))
    s = H.FluidSimulation()
    s.initialize(I, J, K, dx)
    solid = s.solid_sdf()     # the reference's default boundary (a box of solid cells around the domain)
    s.close()
    rng = np.random.default_rng(11)
    n = (np.array([I - 2, J - 2, K - 2]) * 2).astype(int)
    gx, gy, gz = np.meshgrid(*[(np.arange(m) + 0.5) / 2 + 1.0 for m in n], indexing="ij")
    pos = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1) + rng.uniform(-0.2, 0.2, (gx.size, 3))
    x = pos[:, 0] / I
    full = np.zeros((len(pos), 6), np.float32)
    full[:, :3] = pos * dx
    full[:, 3] = 0.4 * np.sin(4 * np.pi * x)
    full[:, 4] = 0.12 * np.cos(6 * np.pi * x)
    blob = full[(np.abs(pos[:, 0] - 40.0) < 10.0) & (pos[:, 1] < 12.0)]
    if J > 20:
        full = full[pos[:, 1] < J - 3.0]   # (a free surface: this velocity field in a box filled to the lid has no divergence-free projection)
    assert len(blob) > 5000
    px, py = 8 * ((I + 1 + 7) // 8), 4 * ((J + 1 + 3) // 4)
    tiles = lambda rowl: -(-px // (4 * rowl)) * -(-py // (4 * (64 // rowl))) * (K + 1)

    def ctx():
        c = capi.Context(I, J, K, dx)
        c.set_solid_sdf(solid)
        c.set_viscosity(2.0)
        c.set_params(viscosity_max_iterations=20000, viscosity_tolerance=1e-7, pressure_rel_tolerance=1e-7, viscosity_layout=capi.LAYOUT_SWIZZLED)   # (the plane layouts: this test is about their tile geometry)
        return c

    a = ctx()
    a.particles = blob
    st = a.substep(0.005)
    assert st["pressure"]["total_tiles"] == tiles(16) and st["viscosity"]["total_tiles"] == tiles(16), st
    a.particles = full
    st = a.substep(0.005)
    assert st["viscosity"]["status"] == 0 and st["pressure"]["status"] == 0
    assert st["pressure"]["total_tiles"] == tiles(filled_rows) and st["viscosity"]["total_tiles"] == tiles(filled_rows), st
    b = ctx()
    b.particles = full
    st = b.substep(0.005)
    assert st["pressure"]["total_tiles"] == tiles(filled_rows) and st["viscosity"]["total_tiles"] == tiles(filled_rows), st
    va, vb = [a.grid(n_) for n_ in "UVW"], [b.grid(n_) for n_ in "UVW"]
    assert rel_maxnorm3(va, vb) <= 1e-4   # two runs differ by the order of the fp32 scatter atomics, amplified by the solves (measured 4e-5)
    a.close()
    b.close()


@pytest.mark.parametrize("dims", [(37, 19, 23), (70, 33, 9)])
@pytest.mark.parametrize("rowl", [16, 64, 0])
def test_odd_sized_domains_match_oracle(oracle, dims, rowl):
    """extents that are not multiples of anything the kernels like (lanes of 4, patches of 8 x 4, tiles of 64 x 16 or
    256 x 4, bricks of 8 x 4 x 2, 8^3 bins): the padded index space, partial tiles and the swizzled plane layout against the oracle, in
    both tile geometries (rowl 16 / 64: flipv_params.tile_rows, viscosity on the plane layouts) and in the brick layout (rowl 0: bricks forced)"""
    from flipviscosity3d_amd import capi, hostapi as H
    I, J, K = dims
    dx = float(np.float32(1.0 / max(dims)))
    s = H.FluidSimulation()
    s.initialize(I, J, K, dx)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 5)
    s.addLiquid(box_mesh((2.3 * dx, 2.2 * dx, 2.4 * dx), ((I - 2.6) * dx, (J - 5.3) * dx, (K - 2.2) * dx)))
    solid, P = s.solid_sdf(), s.particles
    s.close()
    x = P[:, 0] / (I * dx)
    P[:, 3] = 0.3 * np.sin(5 * np.pi * x)
    P[:, 4] = 0.1 * np.cos(3 * np.pi * x)
    P[:, 5] = 0.05 * np.sin(2 * np.pi * x)
    nu, dt = 3.0, 0.004
    c = capi.Context(I, J, K, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(nu)
    c.set_params(viscosity_max_iterations=20000, viscosity_tolerance=1e-7, pressure_rel_tolerance=1e-7, tile_rows=rowl,
                 viscosity_layout=capi.LAYOUT_SWIZZLED if rowl else capi.LAYOUT_BRICK)
    o = oracle.OracleSim(I, J, K, dx)
    o.set_solid(solid)
    o.set_viscosity(nu)
    o.set_solver_limits(vmaxiter=20000)
    o.particles = P
    for t in range(2):
        c.particles = o.particles
        st = c.substep(dt)
        sec, vi, pi = o.substep(dt)
        assert st["viscosity"]["status"] == 0 and vi["status"] == 0 and st["viscosity"]["rows"] + st["viscosity"]["eliminated_rows"] == vi["rows"]
        got, ref = [c.grid(n) for n in "UVW"], [o.grid(n) for n in "UVW"]
        assert rel_maxnorm3(got, ref) <= 1e-4, (t, rel_maxnorm3(got, ref))
        assert np.array_equal(c.grid("LIQUID_PHI"), o.grid("LIQUID_PHI"))
        assert np.abs(c.particles[:, :3] - o.particles[:, :3]).max() <= 2e-7
    c.close()
    o.close()


def test_liquid_box_restriction_over_a_long_run():
    """Inside a substep the sweeps whose result is trivial away from the liquid cover only the liquid's neighbourhood (this
    substep's and the previous one's).  Stale data would only show once the liquid has MOVED: 40 substeps of the bunny falling
    and splashing (64^3), every substep compared with a context that sweeps everything (flipv_params.no_liquid_box) started from the same
    particles: liquid SDF bit for bit, velocities to summation-order noise."""
    import os
    from flipviscosity3d_amd import capi, hostapi as H
    mesh = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes")
    N = 64
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    s.addBoundary(H.load_ply(os.path.join(mesh, "sphere_large.ply")), True)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 2)
    s.addLiquid(H.load_ply(os.path.join(mesh, "stanford_bunny.ply")))
    solid, P = s.solid_sdf(), s.particles
    s.close()
    P[:, 3:] = np.array([0.9, -2.5, 0.6], np.float32)          # moving fast: the box changes every substep
    a = capi.Context(N, N, N, dx)
    b = capi.Context(N, N, N, dx)
    b.set_params(no_liquid_box=1)
    for c in (a, b):   # (tight solver tolerances: the comparison is about which entries were swept, not about where an iteration stopped)
        c.set_solid_sdf(solid); c.set_viscosity(0.5)
        c.set_params(viscosity_max_iterations=5000, viscosity_tolerance=1e-7, pressure_rel_tolerance=1e-7,
                     viscosity_preconditioner=capi.PRECOND_DIAGONAL,   # (AUTO decides from iteration counts, which may differ by one between two runs)
                     exact_viscosity_operator=1)   # (the reference's float-rounded diagonal leaves the near-rigid modes of tiny liquid clusters ill-determined:
                                                   #  two runs of ONE configuration then differ by 4e-2 on the substep where the body touches the wall)
    a.particles = P
    worst = 0.0
    for t in range(40):
        b.particles = a.particles
        for n in "UVW":                                         # (the CFL step reads the field the last substep left)
            b.set_grid(n, a.grid(n))
        dt = min(a.cfl(), 0.01)
        assert dt == min(b.cfl(), 0.01)
        sa, sb = a.substep(dt), b.substep(dt)
        assert sa["viscosity"]["rows"] == sb["viscosity"]["rows"] and sa["pressure"]["rows"] == sb["pressure"]["rows"], t
        assert np.array_equal(a.grid("LIQUID_PHI"), b.grid("LIQUID_PHI")), t
        for n in "UVW":
            assert np.array_equal(a.grid("VALID_" + n), b.grid("VALID_" + n)), (t, n)
        # velocities: 5e-7 on most substeps; on a substep where the falling body first touches the wall two runs of the SAME
        # configuration differ by 1e-3 as well (the summation order of the P2G atomics decides how an almost enclosed pocket of
        # liquid is projected) -- entries that were not swept would show as differences of the order of the velocity itself
        worst = max(worst, rel_maxnorm3([a.grid(n) for n in "UVW"], [b.grid(n) for n in "UVW"]))
        assert worst <= 5e-3, (t, worst)
    # the liquid really travelled
    assert np.abs(a.particles[:, :3].mean(axis=0) - P[:, :3].mean(axis=0)).max() > 5 * dx
    a.close(); b.close()


def test_liquid_box_restriction_over_a_long_run_with_default_parameters(oracle):
    """The same question with NO solver parameter set: the reference's float-rounded operator, AUTO, the two-stage solve and its velocity
    criterion.  Three contexts per substep from the same particles: a (defaults), a2 (defaults again: what two runs of ONE configuration differ by -- the P2G
    atomics' summation order) and b (no_liquid_box: every sweep over the whole grid).  Stated bounds, relative max-norm over every face:
      * a against b: <= 1e-4 on every substep (an entry that was not swept shows as a difference of the order of the velocity itself; measured 1.5e-6 at most --
        with round 4's rule two runs of ONE configuration differed by 4e-2 where the body meets the wall, which is why the older test above pins the exact operator);
      * a against a2: printed; a against b must not exceed 20 x the run-to-run figure of the same substep or 1e-4, whichever is larger;
      * every 8th substep against the oracle run to 1e-13 from the same particles: <= 1e-4 (the parity bar)."""
    import os
    from flipviscosity3d_amd import capi, hostapi as H
    mesh = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes")
    N = 64
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    s.addBoundary(H.load_ply(os.path.join(mesh, "sphere_large.ply")), True)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 2)
    s.addLiquid(H.load_ply(os.path.join(mesh, "stanford_bunny.ply")))
    solid, P = s.solid_sdf(), s.particles
    s.close()
    P[:, 3:] = np.array([0.9, -2.5, 0.6], np.float32)
    ctx = [capi.Context(N, N, N, dx) for _ in range(3)]
    a, a2, b = ctx
    b.set_params(no_liquid_box=1)
    for c in ctx:
        c.set_solid_sdf(solid); c.set_viscosity(0.5)
    a.particles = P
    d_ab, d_aa, d_or = [], [], []
    short = 0
    for t in range(40):
        Pa = a.particles
        ua = [a.grid(n) for n in "UVW"]
        for c in (a2, b):
            c.particles = Pa
            for n, g in zip("UVW", ua):
                c.set_grid(n, g)
        dt = min(a.cfl(), 0.01)
        sa, sa2, sb = a.substep(dt), a2.substep(dt), b.substep(dt)
        assert sa["viscosity"]["status"] in (0, 1) and sa["pressure"]["status"] == 0, (t, sa["viscosity"], sa["pressure"])   # (1 = a stage ended short of its target: the result is applied and compared below like any other)
        short += sa["viscosity"]["status"] == 1
        assert sb["viscosity"]["status"] in (0, 1), (t, sb["viscosity"])   # (the every-entry sweeps are a debug configuration; what they must reproduce is the field, below)
        assert sa["viscosity"]["rows"] == sb["viscosity"]["rows"] and sa["pressure"]["rows"] == sb["pressure"]["rows"], t
        assert np.array_equal(a.grid("LIQUID_PHI"), b.grid("LIQUID_PHI")), t
        va, va2, vb = ([c.grid(n) for n in "UVW"] for c in (a, a2, b))
        d_ab.append(rel_maxnorm3(va, vb)); d_aa.append(rel_maxnorm3(va, va2))
        assert d_ab[-1] <= 1e-4, (t, d_ab[-1])
        assert d_ab[-1] <= max(20.0 * d_aa[-1], 1e-4), (t, d_ab[-1], d_aa[-1])
        if t % 8 == 7:
            o = oracle.OracleSim(N, N, N, dx)
            o.set_solid(solid); o.set_viscosity(0.5)
            o.set_solver_limits(vmaxiter=3000000, vtol=1e-13, ptol=1e-13)
            o.particles = Pa
            for n, g in zip("UVW", ua):
                o.set_grid(n, g)
            o.substep(dt)
            d_or.append(rel_maxnorm3(va, [o.grid(n) for n in "UVW"]))
            o.close()
    print("default parameters, 40 substeps: a vs no_liquid_box max %.2e median %.2e | two runs of one configuration max %.2e median %.2e | against the converged oracle %s"
          % (max(d_ab), float(np.median(d_ab)), max(d_aa), float(np.median(d_aa)), ["%.1e" % e for e in d_or]))
    assert max(d_or) <= 1e-4, d_or
    assert short <= 2, short
    assert np.abs(a.particles[:, :3].mean(axis=0) - P[:, :3].mean(axis=0)).max() > 5 * dx
    for c in ctx:
        c.close()
