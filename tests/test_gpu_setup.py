"""Scene setup on the device against the host C++ path, which tests/test_host_setup.py pins
bit-for-bit to the reference: mesh level set (exact band + sign identical, far field never larger), boundary union,
seeding (same particles), and a substep on the device-built scene."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MESH = os.path.join(os.path.dirname(__file__), "golden", "meshes")


def band_mask(mesh, I, J, K, dx, band=3):
    """nodes visited by the exact band of MeshLevelSet (meshlevelset.cpp:220-232): every triangle's bounding box in node
    units, widened by `band` below and band+1 above, clamped."""
    v, t = mesh
    inv = 1.0 / np.float64(dx)
    f = v.astype(np.float64)[t] * inv                      # (M, 3 corners, 3 axes)
    lo, hi = f.min(axis=1), f.max(axis=1)
    m = np.zeros((K + 1, J + 1, I + 1), bool)
    dims = (I + 1, J + 1, K + 1)
    a0 = [np.clip(lo[:, a].astype(np.int64) - band, 0, dims[a] - 1) for a in range(3)]
    a1 = [np.clip(hi[:, a].astype(np.int64) + band + 1, 0, dims[a] - 1) for a in range(3)]
    for n in range(len(t)):
        m[a0[2][n]:a1[2][n] + 1, a0[1][n]:a1[1][n] + 1, a0[0][n]:a1[0][n] + 1] = True
    return m


@pytest.mark.parametrize("name,N", [("stanford_bunny.ply", 32), ("sphere_large.ply", 40), ("cone.ply", 24)])
def test_mesh_level_set_matches_host(name, N):
    from flipviscosity3d_amd import capi, hostapi as H
    mesh = H.load_ply(os.path.join(MESH, name))
    dx = float(np.float32(1.0 / N))
    ref, ref_closest = H.mesh_sdf(N, N, N, dx, mesh)
    c = capi.Context(N, N, N, dx)
    got, closest = c.mesh_level_set(mesh, want_closest=True)
    c.close()
    band = band_mask(mesh, N, N, N, dx)
    assert band.any() and not band.all()
    assert np.array_equal(got[band], ref[band])                    # exact band: bit-identical distances ...
    assert np.array_equal(closest[band], ref_closest[band])        # ... and the same closest triangle (ties included)
    assert np.array_equal(np.signbit(got), np.signbit(ref))        # inside/outside identical everywhere
    far = ~band
    # far field: both are upper bounds of the true distance built from neighbours' closest triangles; the device fixed
    # point is mostly tighter (down to -35 % measured) and, where the reference's visiting order happened to carry a
    # better candidate, at most ~1 % looser
    assert (np.abs(got[far]) <= np.abs(ref[far]) * 1.02).all()
    assert (np.abs(got[far]) >= 3 * dx * 0.99).all()               # and still outside the band distance


def test_boundary_and_seeding_match_host():
    from flipviscosity3d_amd import capi, hostapi as H
    N = 32
    dx = float(np.float32(1.0 / N))
    sphere = H.load_ply(os.path.join(MESH, "sphere_large.ply"))
    bunny = H.load_ply(os.path.join(MESH, "stanford_bunny.ply"))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    box = s.solid_sdf()
    s.addBoundary(sphere, True)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 7)
    s.addLiquid(bunny)
    solid_ref, part_ref = s.solid_sdf(), s.particles
    s.close()

    c = capi.Context(N, N, N, dx)
    c.reset_boundary()
    got_box = c.grid("SOLID_PHI")
    near = np.abs(box) <= 2.5 * dx
    assert np.array_equal(got_box[near], box[near]) and np.array_equal(np.signbit(got_box), np.signbit(box))
    c.add_boundary_mesh(sphere, inverted=True)
    solid = c.grid("SOLID_PHI")
    near = np.abs(solid_ref) <= 2.5 * dx
    assert np.array_equal(solid[near], solid_ref[near]) and np.array_equal(np.signbit(solid), np.signbit(solid_ref))
    n = c.add_liquid_mesh(bunny, seed=7)
    assert n == len(part_ref) and c.num_particles == n
    assert np.array_equal(c.particles, part_ref)                    # same samples, same order
    # a second body is appended behind the first (fluidsimulation.cpp:89)
    cone = H.load_ply(os.path.join(MESH, "cone.ply"))
    n2 = c.add_liquid_mesh(cone, seed=7)
    assert c.num_particles == n + n2 and np.array_equal(c.particles[:n], part_ref)
    c.close()


def test_substep_on_device_built_scene_equals_host_built_scene():
    """Far from the surfaces the device level set differs from the reference's (smaller distances); nothing in a substep
    reads those values, so the velocities must agree bit for bit."""
    from flipviscosity3d_amd import capi, hostapi as H
    N = 32
    dx = float(np.float32(1.0 / N))
    sphere = H.load_ply(os.path.join(MESH, "sphere_large.ply"))
    bunny = H.load_ply(os.path.join(MESH, "stanford_bunny.ply"))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    s.addBoundary(sphere, True)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 0)
    s.addLiquid(bunny)
    solid_ref, part_ref = s.solid_sdf(), s.particles
    s.close()
    a = capi.Context(N, N, N, dx)
    a.set_solid_sdf(solid_ref)
    a.particles = part_ref
    b = capi.Context(N, N, N, dx)
    b.reset_boundary()
    b.add_boundary_mesh(sphere, inverted=True)
    b.add_liquid_mesh(bunny, seed=0)
    for c in (a, b):
        c.set_viscosity(5.0)
        c.set_params(viscosity_preconditioner=capi.PRECOND_DIAGONAL)   # (two runs compared bit for bit downstream: no per-run choice of preconditioner)
    for t in range(2):
        a.substep(0.01)
        b.substep(0.01)
        assert np.array_equal(a.grid("LIQUID_PHI"), b.grid("LIQUID_PHI"))
        ga, gb = [a.grid(n) for n in "UVW"], [b.grid(n) for n in "UVW"]
        scale = max(np.abs(g).max() for g in ga)
        # two runs of the same scene differ by this much as well: P2G and the PCG dot products sum in atomic order
        assert max(np.abs(x - y).max() for x, y in zip(ga, gb)) <= 2e-5 * scale
        assert np.abs(a.particles[:, :3] - b.particles[:, :3]).max() <= 1e-6
    a.close()
    b.close()


def test_host_class_with_device_setup():
    """FluidSimulation (host C++ mirror) with setSetupOnDevice: same particles and near-surface solid SDF as the host path."""
    from flipviscosity3d_amd import hostapi as H
    N = 32
    dx = float(np.float32(1.0 / N))
    sphere = H.load_ply(os.path.join(MESH, "sphere_large.ply"))
    bunny = H.load_ply(os.path.join(MESH, "stanford_bunny.ply"))
    out = []
    for dev in (False, True):
        s = H.FluidSimulation()
        s.initialize(N, N, N, dx, setup_on_device=dev)
        s.addBoundary(sphere, True)
        s.setSeeding(H.FluidSimulation.SEED_COUNTER, 3)
        s.addLiquid(bunny)
        out.append((s.solid_sdf(), s.particles))
        if dev:
            s.setViscosity(5.0)
            st = s.advance(0.01)
            assert st["substeps"] >= 1 and len(s.particles) == len(out[0][1])
        s.close()
    (sa, pa), (sb, pb) = out
    near = np.abs(sa) <= 2.5 * dx
    assert np.array_equal(sa[near], sb[near]) and np.array_equal(np.signbit(sa), np.signbit(sb))
    assert np.array_equal(pa, pb)


def test_resumed_run_continues_like_the_uninterrupted_one(tmp_path):
    from flipviscosity3d_amd import hostapi as H
    N = 24
    dx = float(np.float32(1.0 / N))
    def start():
        s = H.FluidSimulation()
        s.initialize(N, N, N, dx)
        s.addBoundary(H.load_ply(os.path.join(MESH, "sphere_large.ply")), True)
        s.setSeeding(H.FluidSimulation.SEED_COUNTER, 1)
        s.addLiquid(H.load_ply(os.path.join(MESH, "stanford_bunny.ply")))
        s.setViscosity(5.0)
        return s
    a = start()
    for f in range(4):
        a.advance(0.01)
    pa = a.particles
    a.close()
    b = start()
    for f in range(2):
        b.advance(0.01)
    path = str(tmp_path / "ck.flipv")
    b.saveState(path)
    b.close()
    r = H.FluidSimulation()
    r.initialize(4, 4, 4, 0.25)
    r.loadState(path)
    for f in range(2):
        r.advance(0.01)
    pr = r.particles
    r.close()
    assert pa.shape == pr.shape
    assert np.abs(pa[:, :3] - pr[:, :3]).max() <= 1e-5      # the same substeps; sums run in atomic order in both
    assert np.abs(pa[:, 3:] - pr[:, 3:]).max() <= 1e-4 * np.abs(pa[:, 3:]).max()


def test_resumed_run_takes_the_same_cfl_substeps(tmp_path):
    """The checkpoint carries the MAC field: _cfl() of the first frame after a resume reads the velocities the last
    substep left (reference fluidsimulation.cpp:139, 241-269).  Fast-moving liquid and a frame several CFL steps long: the
    resumed run must split its frame like the uninterrupted one (without the field the CFL step is +inf and the whole
    frame becomes one substep)."""
    from flipviscosity3d_amd import hostapi as H
    N = 24
    dx = float(np.float32(1.0 / N))
    def start():
        s = H.FluidSimulation()
        s.initialize(N, N, N, dx)
        s.addBoundary(H.load_ply(os.path.join(MESH, "sphere_large.ply")), True)
        s.setSeeding(H.FluidSimulation.SEED_COUNTER, 3)
        s.addLiquid(H.load_ply(os.path.join(MESH, "stanford_bunny.ply")))
        s.setViscosity(0.5)
        P = s.particles
        P[:, 4] = -6.0                      # 5 dx / 6 = 0.035: a frame of 0.1 needs three substeps
        s.particles = P
        return s
    a = start()
    a.advance(0.1)
    na = a.advance(0.1)["substeps"]
    pa = a.particles
    a.close()
    b = start()
    b.advance(0.1)
    path = str(tmp_path / "ck.flipv")
    b.saveState(path)
    b.close()
    r = H.FluidSimulation()
    r.initialize(4, 4, 4, 0.25)
    r.loadState(path)
    nr = r.advance(0.1)["substeps"]
    pr = r.particles
    r.close()
    assert na >= 2, na                       # the frame really is CFL-bound
    assert nr == na
    # a splash at 6 m/s on a 24^3 grid: atomic summation order alone moves particles by 1e-4 after these substeps
    assert np.abs(pa[:, :3] - pr[:, :3]).max() <= 1e-3


def test_setup_only_context_builds_the_same_scene():
    """flipv_create_setup: the light context a rank of a block decomposition builds its scene with -- same solid SDF and
    particles as a full context, none of the substep state, every substep entry point refused"""
    from flipviscosity3d_amd import capi, hostapi as H
    N = 40
    dx = float(np.float32(1.0 / N))
    sphere, bunny = H.load_ply(os.path.join(MESH, "sphere_large.ply")), H.load_ply(os.path.join(MESH, "stanford_bunny.ply"))
    out = []
    for setup_only in (False, True):
        c = capi.Context(N, N, N, dx, setup_only=setup_only)
        c.reset_boundary()
        c.add_boundary_mesh(sphere, inverted=True)
        n = c.add_liquid_mesh(bunny, seed=7)
        out.append((c.grid("SOLID_PHI"), c.particles, n))
        if setup_only:
            with pytest.raises(capi.FlipvError):
                c.substep(0.01)
            with pytest.raises(capi.FlipvError):
                c.particle_sdf()
        c.close()
    assert out[0][2] == out[1][2] > 1000
    sa, sb = out[0][0], out[1][0]
    near = np.abs(sa) <= 2.5 * dx                   # exact band + signs: bit-identical; the far field is a relaxation's fixed point
    assert np.array_equal(sa[near], sb[near]) and np.array_equal(np.signbit(sa), np.signbit(sb))
    assert np.abs(sa - sb).max() <= 0.02 * np.abs(sa).max()
    assert np.array_equal(out[0][1], out[1][1])


def test_setup_rejects_bad_input():
    from flipviscosity3d_amd import capi
    c = capi.Context(16, 16, 16, 1.0 / 16)
    tri = (np.array([[0.2, 0.2, 0.2], [0.5, 0.2, 0.2], [0.2, 0.5, 0.2]], np.float32), np.array([[0, 1, 2]], np.int32))
    with pytest.raises(capi.FlipvError):
        c.add_boundary_mesh((tri[0] + 2.0, tri[1]))            # outside the domain
    with pytest.raises(capi.FlipvError):
        c.mesh_level_set((tri[0], np.array([[0, 1, 5]], np.int32)))   # vertex index out of range
    c.close()
