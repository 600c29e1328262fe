"""CPU: host-side scene setup (mesh level set, boundary, seeding, PLY I/O) against the reference."""
import ctypes
import os

import numpy as np
import pytest

from helpers import GOLDEN, Golden

MESH = os.path.join(GOLDEN, "meshes")


def ref_or_skip():
    from oracle import refbind as R
    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return R


def test_ply_loader_reads_short_files_the_reference_cannot():
    from flipviscosity3d_amd import hostapi as H
    from flipviscosity3d_amd.plyio import load_ply
    for name, nv, nt in [("cube.ply", 8, 12), ("sheet.ply", 8, 12), ("cone.ply", 33, 62), ("rod.ply", 64, 124),
                         ("sphere_large.ply", 2562, 5120), ("stanford_bunny.ply", 7682, 15360)]:
        v, t = H.load_ply(os.path.join(MESH, name))
        assert v.shape == (nv, 3) and t.shape == (nt, 3)
        v2, t2 = load_ply(os.path.join(MESH, name))
        assert np.array_equal(v, v2) and np.array_equal(t, t2)
    with pytest.raises(IOError):
        H.load_ply(os.path.join(MESH, "does_not_exist.ply"))


def test_ply_roundtrip(tmp_path):
    from flipviscosity3d_amd.plyio import load_ply, save_ply
    v, t = load_ply(os.path.join(MESH, "cone.ply"))
    p = str(tmp_path / "x.ply")
    save_ply(p, v, t)
    v2, t2 = load_ply(p)
    assert np.array_equal(v, v2) and np.array_equal(t, t2)


def test_default_boundary_matches_golden():
    """_initializeBoundary (fluidsimulation.cpp:225-239): inset box, band 3, negated -- bit-exact vs the reference dump"""
    from flipviscosity3d_amd import hostapi as H
    g = Golden("cube24_inviscid")
    s = H.FluidSimulation()
    s.initialize(g.I, g.J, g.K, g.dx)
    assert np.array_equal(s.solid_sdf(), g["solid"])
    s.close()


def test_boundary_union_and_seeding_match_golden():
    """addBoundary(inverted sphere) + addLiquid(bunny) with the libc rand() stream == the reference's scene"""
    from flipviscosity3d_amd import hostapi as H
    g = Golden("bunny32_viscous")
    s = H.FluidSimulation()
    s.initialize(g.I, g.J, g.K, g.dx)
    s.addBoundary(H.load_ply(os.path.join(MESH, "sphere_large.ply")), True)
    assert np.array_equal(s.solid_sdf(), g["solid"])
    ctypes.CDLL(None).srand(1)  # glibc default seed = the reference's unseeded rand()
    s.addLiquid(H.load_ply(os.path.join(MESH, "stanford_bunny.ply")))
    assert np.array_equal(s.particles, g["particles0"])
    s.close()


def test_two_liquid_bodies_append():
    from flipviscosity3d_amd import hostapi as H
    g = Golden("twobody20_varvisc")
    s = H.FluidSimulation()
    s.initialize(g.I, g.J, g.K, g.dx)
    ctypes.CDLL(None).srand(1)
    s.addLiquid(H.load_ply(os.path.join(MESH, "sphere_small.ply")))
    n1 = len(s.particles)
    s.addLiquid(H.load_ply(os.path.join(MESH, "cone.ply")))
    assert len(s.particles) > n1
    assert np.array_equal(s.particles, g["particles0"])
    s.close()


def test_counter_seeding_is_deterministic_and_inside_the_mesh():
    from flipviscosity3d_amd import hostapi as H
    out = []
    for _ in range(2):
        s = H.FluidSimulation()
        s.initialize(24, 24, 24, 1.0 / 24)
        s.setSeeding(H.FluidSimulation.SEED_COUNTER, 7)
        s.addLiquid(H.load_ply(os.path.join(MESH, "cube.ply")))
        out.append(s.particles)
        s.close()
    assert np.array_equal(out[0], out[1])
    p = out[0][:, :3]
    assert len(p) == 12 ** 3 * 8          # cube [0.25,0.75]^3 = 12^3 cells, 8 samples each
    assert p.min() >= 0.25 and p.max() <= 0.75 and not out[0][:, 3:].any()


def test_out_of_domain_mesh_is_rejected():
    from flipviscosity3d_amd import hostapi as H
    s = H.FluidSimulation()
    s.initialize(16, 16, 16, 1.0 / 16)
    v, t = H.load_ply(os.path.join(MESH, "cube.ply"))
    with pytest.raises(ValueError):
        s.addLiquid((v + 0.6, t))         # FLUIDSIM_ASSERT(domain.isPointInside(...)) fluidsimulation.cpp:65-68
    with pytest.raises(ValueError):
        s.addBoundary((v - 0.5, t))
    with pytest.raises(ValueError):
        s.setViscosity(-1.0)
    s.close()


@pytest.mark.parametrize("mesh,N", [("bunny.ply", 40), ("rod.ply", 36), ("sheet.ply", 28)])
def test_mesh_level_set_bit_exact_vs_reference(mesh, N):
    R = ref_or_skip()
    from flipviscosity3d_amd import hostapi as H
    m = H.load_ply(os.path.join(MESH, mesh))
    dx = float(np.float32(1.0 / N))
    a, ac = H.mesh_sdf(N, N, N, dx, m, 3)
    b, bc = R.mesh_sdf(N, N, N, dx, m[0], m[1], 3)
    assert np.array_equal(a, b)
    assert np.array_equal(ac, bc)
    # non-cubic grid and a different band
    a, ac = H.mesh_sdf(N, N // 2 + 5, N - 7, dx, (m[0] * np.float32(0.45), m[1]), 2)
    b, bc = R.mesh_sdf(N, N // 2 + 5, N - 7, dx, m[0] * np.float32(0.45), m[1], 2)
    assert np.array_equal(a, b) and np.array_equal(ac, bc)


def test_checkpoint_round_trip(tmp_path):
    """saveState / loadState: the file restores grid size, solid SDF, viscosity and particles exactly"""
    from flipviscosity3d_amd import hostapi as H
    N = 20
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    s.addBoundary(H.load_ply(os.path.join(MESH, "sphere_large.ply")), True)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 5)
    s.addLiquid(H.load_ply(os.path.join(MESH, "stanford_bunny.ply")))
    P = s.particles
    P[:, 3:] = np.random.default_rng(0).uniform(-1, 1, (len(P), 3)).astype(np.float32)
    s.particles = P
    solid = s.solid_sdf()
    path = str(tmp_path / "state.flipv")
    s.saveState(path)
    s.close()
    r = H.FluidSimulation()
    r.initialize(8, 8, 8, 0.125)         # different size on purpose: loadState re-initialises
    r.loadState(path)
    assert (r.I, r.J, r.K) == (N, N, N)
    assert np.array_equal(r.solid_sdf(), solid) and np.array_equal(r.particles, P)
    with pytest.raises(IOError):
        r.loadState(str(tmp_path / "missing.flipv"))
    raw = open(path, "rb").read()
    # a truncated file, a header that promises more than the file holds (huge dims / particle count), a wrong magic:
    # all rejected with an error return, nothing allocated from the header's word alone, nothing thrown
    import struct
    bad = {"truncated": raw[:len(raw) // 2],
           "huge_dims": raw[:8] + struct.pack("<3i", 16000, 16000, 16000) + raw[20:],
           "overflow_dims": raw[:8] + struct.pack("<3i", 2**31 - 1, 2**31 - 1, 2**31 - 1) + raw[20:],
           "huge_np": raw[:44] + struct.pack("<Q", 2**62) + raw[52:],
           "trailing": raw + b"\0" * 24,
           "magic": b"XXXXXXXX" + raw[8:]}
    for name, blob in bad.items():
        q = str(tmp_path / (name + ".flipv"))
        open(q, "wb").write(blob)
        with pytest.raises(IOError):
            r.loadState(q)
    # the simulation is untouched by the failed loads, and a checkpoint written before any frame re-saves identically
    assert (r.I, r.J, r.K) == (N, N, N) and np.array_equal(r.particles, P)
    again = str(tmp_path / "again.flipv")
    r.saveState(again)
    assert open(again, "rb").read() == raw
    r.close()
