"""GPU (-m gpu): BASELINE config 4's STIFFNESS regime with the library's DEFAULT parameters.

Config 4 (512^3 honey buckling, nu = 50) runs at nu dt/dx^2 = 131 072; until round 4 every converged golden stopped at 8 192.  The two fixtures
here were cut from the reference with its cap lifted at 122 880 (config 1's scene, 64^3, nu = 3 000) and at 131 070 (config 4's own scene -- rod.ply +
sheet.ply through two addLiquid calls -- at 96^3, nu = 1 422.2); the oracle is pinned to both (tests/test_oracle_compact_golden.py).  Every substep
is UNCHAINED -- substep t starts from the reference's own particles, stored in the fixture -- because the chained state is ill-conditioned at this
stiffness (a 1e-5 difference in substep 0's velocities becomes 1e-3 in substep 1's whatever the solver does: profiles/r3/stiffness_scan_64.log).
NO field of flipv_params is set.  Bar: end-of-substep velocities <= 1e-4 relative max-norm at the fixture's 20 000 probe faces per component."""
import numpy as np
import pytest

from helpers import Golden
from test_oracle_compact_golden import STIFF, build_host_scene

pytestmark = pytest.mark.gpu


def probe_error(c, g, t):
    num = den = 0.0
    for n in "UVW":
        a = c.grid(n).reshape(-1)
        num = max(num, float(np.abs(a[g["s%d_probe_idx_%s" % (t, n)]].astype(np.float64) - g["s%d_probe_val_%s" % (t, n)]).max()))
        den = max(den, float(g["s%d_maxabs_%s" % (t, n)]))
    return num / den


@pytest.mark.parametrize("name,N,boundary,liquids", STIFF)
def test_default_params_at_config4_stiffness_unchained(name, N, boundary, liquids):
    from flipviscosity3d_amd.capi import Context
    g = Golden(name)
    dx, solid, P = build_host_scene(N, boundary, liquids)
    assert len(P) == int(g["nparticles"]) and np.array_equal(P.astype(np.float64).sum(axis=0), g["particles0_sum"])
    nu = float(g["nu"])
    assert 1.2e5 <= nu * g.dt / g.dx ** 2 <= 1.35e5
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(nu)
    for t in range(g.nsub):
        c.particles = P if t == 0 else g["s%d_particles" % (t - 1)]
        st = c.substep(g.dt)
        v = st["viscosity"]
        err = probe_error(c, g, t)
        print("%s substep %d: %d iterations (%d in %s correction stages; reference %d), status %d / %d, velocity error %.2e" % (
            name, t, v["iterations"], v["correction_iterations"], "the", int(g["s%d_visc_iters" % t]), v["status"], v["correction_status"], err))
        assert v["preconditioner"] == 1 and v["layout"] == 2 and v["iterations"] <= 700, v     # inside the stock cap, where the reference needs 2 200 - 4 100
        assert v["status"] == 0 and v["correction_status"] == 1 and v["defect_residual"] > 0.0, v   # every stage reached its target
        assert st["pressure"]["status"] in (0, 3)
        assert err <= 5e-5, (t, err)    # (the bar is 1e-4; measured 1e-5 ... 3e-5)
        d = np.abs(c.particles.astype(np.float64).sum(axis=0) - g["s%d_particles_sum" % t]) / len(P)
        assert d[:3].max() <= 1e-6 and d[3:].max() <= 1e-5, d
    c.close()


@pytest.mark.parametrize("name,N,boundary,liquids", STIFF[1:])
def test_round3_rule_misses_the_bar_there(name, N, boundary, liquids):
    """what round 3 shipped (one correction stage to 1 % of the defect in at most 48 iterations) measured against the same fixture: it does NOT meet 1e-4
    on the first substep -- the reason the stage's share is 0.1 % beyond nu dt/dx^2 = 2e4.  (If this starts passing the bar, the looser share can come back.)"""
    from flipviscosity3d_amd.capi import Context
    g = Golden(name)
    dx, solid, P = build_host_scene(N, boundary, liquids)
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(float(g["nu"]))
    c.set_params(viscosity_stage2_rounds=1, viscosity_stage2_factor=1e-2, viscosity_stage2_max_iterations=48)
    c.particles = P
    c.substep(g.dt)
    err = probe_error(c, g, 0)
    print("%s with round 3's rule: velocity error %.2e" % (name, err))
    assert 1e-4 < err < 1e-3
    c.close()


def dense_scene(N, top):
    """a liquid box that fills most of the default box boundary: x, z in [0.06, 0.94], y in [0.06, top] (counter-seeded by the host library)"""
    from flipviscosity3d_amd import hostapi as H
    from test_gpu_wide import box_mesh
    dx = float(np.float32(1.0 / N))
    s = H.FluidSimulation()
    s.initialize(N, N, N, dx)
    s.setSeeding(H.FluidSimulation.SEED_COUNTER, 4)
    s.addLiquid(box_mesh((0.06, 0.06, 0.06), (0.94, top, 0.94)))
    solid, P = s.solid_sdf(), s.particles
    s.close()
    return dx, solid, P


def test_dense_viscous_scene_defaults_against_converged_oracle(oracle):
    """Where the liquid fills the box the solver's arrays stay in the PLANE layout (bricks are for sparse liquids) -- and until round 4 that layout had
    no fp64 accumulator, so its default solve applied the EXACT operator and silently skipped the defect correction towards the reference's.  48^3, the
    liquid in 46 % of the domain, nu = 150 (nu dt/dx^2 = 3 456: the headline's stiffness), the particles given a shear so that the viscous solve has
    work to do; NO parameter set; two substeps, the second started from the oracle's particles; against the oracle with its cap lifted: <= 1e-4, and the
    solve must report the plane layout, the multigrid and a correction stage."""
    from flipviscosity3d_amd.capi import Context
    N, nu, dt = 48, 150.0, 0.01
    dx, solid, P = dense_scene(N, 0.66)
    P = P.copy()
    P[:, 3] = 0.8 * np.sin(7.0 * P[:, 1]) * np.cos(5.0 * P[:, 2]); P[:, 4] = -0.3 * np.cos(6.0 * P[:, 0]); P[:, 5] = 0.5 * np.sin(4.0 * P[:, 0] + 3.0 * P[:, 1])
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu)
    o = oracle.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=400000)
    o.particles = P
    for t in range(2):
        c.particles = o.particles
        st = c.substep(dt)
        sec, vi, pi = o.substep(dt)
        v = st["viscosity"]
        fill = v["rows"] / (3.0 * N ** 3)
        num = max(np.abs(c.grid(n).astype(np.float64) - o.grid(n)).max() for n in "UVW")
        den = max(np.abs(o.grid(n)).max() for n in "UVW")
        print("dense 48^3 substep %d: rows fill %.2f, layout %d, %d iterations (%d in corrections; oracle %d), status %d / %d, defect %.1e, velocity error %.2e" % (
            t, fill, v["layout"], v["iterations"], v["correction_iterations"], vi["iterations"], v["status"], v["correction_status"], v["defect_residual"] / v["rhs_norm"], num / den))
        assert vi["status"] == 0
        assert fill > 0.40 and v["layout"] in (0, 1) and v["preconditioner"] == 1, v
        assert v["status"] == 0 and v["defect_residual"] > 0.0 and v["correction_status"] == 1, v
        assert num / den <= 1e-4, (t, num / den)
    c.close()
    o.close()


@pytest.mark.parametrize("nu", [500.0, 1280.0, 2000.0])
def test_default_params_between_headline_and_config4_stiffness(oracle, nu):
    """nu dt/dx^2 = 20 480, 52 429 (the 1024 x 512 x 512 sheet's) and 81 920 on config 1's scene at 64^3, against the live oracle with its cap lifted (2 100 -
    2 700 iterations), NO parameter set, both substeps started from the oracle's particles: <= 1e-4 (measured 2e-6 ... 2.3e-5; round 3's rule: 1.4e-4 at 52 429)."""
    from flipviscosity3d_amd.capi import Context
    N = 64
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    o = oracle.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=400000)
    o.particles = P
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu)
    for t in range(2):
        c.particles = o.particles
        sec, vi, pi = o.substep(0.01)
        st = c.substep(0.01)
        v = st["viscosity"]
        num = max(np.abs(c.grid(n).astype(np.float64) - o.grid(n)).max() for n in "UVW")
        den = max(np.abs(o.grid(n)).max() for n in "UVW")
        print("64^3 nu %g (nu dt/dx^2 = %.0f) substep %d: %d iterations (%d in the correction stage; oracle %d), status %d / %d, velocity error %.2e" % (
            nu, nu * 0.01 / dx ** 2, t, v["iterations"], v["correction_iterations"], vi["iterations"], v["status"], v["correction_status"], num / den))
        assert vi["status"] == 0 and vi["iterations"] > 700
        assert v["status"] == 0 and v["correction_status"] == 1 and v["iterations"] <= 700, v
        assert num / den <= 1e-4, (t, num / den)
    c.close()
    o.close()


@pytest.mark.parametrize("name,N,boundary,liquids", STIFF[:1] + [("bunny128_nu5_converged", 128, ("sphere_large.ply", True), ["stanford_bunny.ply"])])
def test_defect_predictor_changes_the_iteration_count_not_the_velocities(name, N, boundary, liquids):
    """flipv_params.viscosity_defect_predictor: stage 1 of the two-stage solve on b - E u_old (the default) against stage 1 on b (-1).  Both deliver the reference's
    converged velocities to the same few 1e-5 (the correction stage is held to the same target either way); the predictor must not cost iterations
    (measured: 199 -> 162 on the stiff fixture's first substep, 52 -> 51 at 128^3)."""
    from flipviscosity3d_amd.capi import Context
    g = Golden(name)
    dx, solid, P = build_host_scene(N, boundary, liquids)
    its = {}
    for pred in (-1, 0):
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid)
        c.set_viscosity(float(g["nu"]))
        c.set_params(viscosity_defect_predictor=pred)
        c.particles = P
        st = c.substep(g.dt)
        v = st["viscosity"]
        err = probe_error(c, g, 0)
        print("%s, predictor %d: %d iterations (%d correction), velocity error %.2e" % (name, pred, v["iterations"], v["correction_iterations"], err))
        assert v["status"] == 0 and v["correction_status"] == 1 and err <= 5e-5, (pred, v, err)
        its[pred] = v["iterations"]
        c.close()
    assert its[0] <= its[-1] + 3, its


def test_strict_stage_1_converges_to_the_reference_tolerance():
    """flipv_params.viscosity_stage1_factor = 1 (bench.py: mode_b_strict): stage 1 of the two-stage solve runs to viscosity_tolerance x max|rhs| itself -- on the plain
    right-hand side (the defect predictor's b - E u_old is for a stage 1 that stops early: an fp32 loop run to 1e-6 stagnates on it) -- inside the stock cap, and the
    velocities are the default's."""
    from flipviscosity3d_amd.capi import Context
    name, N = "bunny128_nu5_converged", 128
    g = Golden(name)
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(float(g["nu"]))
    c.set_params(viscosity_stage1_factor=1.0)
    c.particles = P
    for t in range(g.nsub):
        st = c.substep(g.dt)
        v = st["viscosity"]
        err = probe_error(c, g, t)
        print("strict, substep %d: %d iterations (%d correction), loop residual %.2e of max|rhs|, velocity error %.2e" % (t, v["iterations"], v["correction_iterations"], v["residual"] / v["rhs_norm"], err))
        assert v["status"] == 0 and v["iterations"] <= 300 and v["residual"] <= 1.0e-6 * v["rhs_norm"], v
        assert err <= 3e-5, (t, err)
    c.close()


def test_sheared_start_where_the_defect_predictor_has_nothing_to_go_on(oracle):
    """The defect predictor assumes a substep changes the velocities little (stage 1 solves A x = b - E u_old).  Here it cannot: config 1's scene at nu = 200
    (nu dt/dx^2 = 8 192) with the particles' velocities overwritten by a strong shear, which this viscosity flattens within the substep -- the solution is far
    from the incoming field, the predicted defect is wrong by about as much as no prediction, and the correction stage has to do what it did before the
    predictor existed.  Default parameters against the oracle with its cap lifted, one substep: <= 1e-4."""
    from flipviscosity3d_amd.capi import Context
    N = 64
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    P = P.copy()
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    P[:, 3] = 1.5 * np.sin(14.0 * np.pi * y)
    P[:, 4] = -0.8 * np.cos(10.0 * np.pi * z) - 0.5
    P[:, 5] = 1.0 * np.sin(12.0 * np.pi * x)
    o = oracle.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(200.0); o.set_solver_limits(vmaxiter=400000)
    o.particles = P
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(200.0)
    c.particles = P
    st = c.substep(0.01)
    sec, vi, pi = o.substep(0.01)
    v = st["viscosity"]
    ref = [o.grid(n) for n in "UVW"]
    den = max(float(np.abs(r).max()) for r in ref)
    err = max(float(np.abs(c.grid(n).astype(np.float64) - r).max()) for n, r in zip("UVW", ref)) / den
    change = float(np.abs(o.particles[:, 3:] - P[:, 3:]).max()) / float(np.abs(P[:, 3:]).max())   # (the FLIP update hands the grid's change to the particles)
    print("sheared start: %d iterations (%d correction; oracle %d), velocity error %.2e; the substep changed the particles' velocities by %.2f of their maximum" % (
        v["iterations"], v["correction_iterations"], vi["iterations"], err, change))
    assert vi["status"] == 0 and v["status"] == 0 and v["correction_iterations"] > 0, (vi, v)
    assert change > 0.3            # the premise: the incoming velocities say little about the outgoing ones
    assert err <= 1e-4, err
    c.close()
    o.close()


def test_variable_viscosity_field_at_64_defaults_against_converged_oracle(oracle):
    """SURVEY 8 f4 at a size where the multigrid and the two-stage solve run: config 1's scene with a NODAL viscosity field (viscositysolver.cpp:394-427 reads it at
    cell centres and edges) -- 200 below y = 0.42, 5 above, a smooth ramp of four cells between, so the bunny's feet are 40 x as viscous as its ears
    (nu dt/dx^2 from 205 to 8 192 inside one system).  Default parameters against the oracle with its cap lifted, two chained substeps, <= 1e-4.
    With a variable field the reference's rows average the four viscosities around an edge each in its own order, so its matrix is one ulp off the symmetric
    one on the edge factors; the fp64 residual of the two-stage solve forms those rows as the reference does (k_viscosity_brick.hip: d_ref_row_factors):
    1.7e-5 / 4.2e-5 here (unchained 1.7e-5 / 1.4e-5), 4.5e-5 / 6.7e-5 with the stored symmetric factors."""
    from flipviscosity3d_amd.capi import Context
    N = 64
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    y = (np.arange(N + 1) * dx)[None, :, None]
    ramp = np.clip((0.42 + 2.0 * dx - y) / (4.0 * dx), 0.0, 1.0)
    nu = np.broadcast_to(5.0 + 195.0 * ramp, (N + 1, N + 1, N + 1)).astype(np.float32).copy()
    assert nu.min() == 5.0 and nu.max() == 200.0
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu)
    o = oracle.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=400000)
    c.particles = P
    o.particles = P
    for t in range(2):
        st = c.substep(0.01)
        sec, vi, pi = o.substep(0.01)
        v = st["viscosity"]
        ref = [o.grid(n) for n in "UVW"]
        err = max(float(np.abs(c.grid(n).astype(np.float64) - r).max()) for n, r in zip("UVW", ref)) / max(float(np.abs(r).max()) for r in ref)
        print("variable viscosity 64^3 substep %d: %d iterations (%d correction; oracle %d), status %d, velocity error %.2e" % (t, v["iterations"], v["correction_iterations"], vi["iterations"], v["status"], err))
        assert vi["status"] == 0 and v["status"] == 0 and v["preconditioner"] == 1 and v["iterations"] < 300, (vi, v)
        assert err <= 1e-4, (t, err)
    c.close()
    o.close()


@pytest.mark.parametrize("nu", [200.0, 3000.0])
def test_the_references_operator_is_reproduced_not_approximated(oracle, nu):
    """Config 1's scene from rest at nu dt/dx^2 = 8 192 and 122 880 against the oracle run to 1e-10 (4 869 / 5 527 iterations) -- i.e. against the SOLUTION of the
    reference's linear system, not against what its own 1e-6 stop leaves.  The default solve is 4e-6 / 1.6e-5 from it; pushed (two correction stages) 2e-7 / 2e-6;
    the EXACT operator converged to 1e-9 is 4e-5 / 7e-4 away: the float-rounded diagonal is part of the reference's answer, and the two-stage solve delivers it."""
    from flipviscosity3d_amd.capi import Context
    N = 64
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    o = oracle.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=2000000, vtol=1e-10, ptol=1e-13)
    o.particles = P
    sec, vi, pi = o.substep(0.01)
    assert vi["status"] == 0
    ref = [o.grid(n) for n in "UVW"]
    den = max(float(np.abs(r).max()) for r in ref)
    errs = {}
    for name, kw in (("default", {}), ("two correction stages", dict(viscosity_stage2_rounds=2)),
                     ("exact operator to 1e-9", dict(precision=1, viscosity_tolerance=1e-9, pressure_rel_tolerance=1e-9, viscosity_max_iterations=5000, exact_viscosity_operator=1))):
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu)
        if kw:
            c.set_params(**kw)
        c.particles = P
        st = c.substep(0.01)
        errs[name] = max(float(np.abs(c.grid(n).astype(np.float64) - r).max()) for n, r in zip("UVW", ref)) / den
        print("nu %g against the oracle at 1e-10 (%d iterations), %s: %d iterations, velocity error %.2e" % (nu, vi["iterations"], name, st["viscosity"]["iterations"], errs[name]))
        c.close()
    o.close()
    # (a second stage that ends short of its target and RAISES the fp64 residual is taken back -- the solve then delivers the first stage's result, status 1: at nu = 3 000
    # that is what happens on some boxes, with round 5's library as well (profiles/r6/ab_two_stages_nu3000.log); where the stage goes through it lands at 2e-6)
    assert errs["default"] <= 3e-5 and (errs["two correction stages"] <= 5e-6 or errs["two correction stages"] <= 1.001 * errs["default"]), errs
    assert errs["exact operator to 1e-9"] >= 5.0 * errs["default"], errs       # a different linear system


@pytest.mark.parametrize("scene", ["sparse (bricks)", "dense (planes)"])
def test_per_row_factor_residual_equals_the_stored_factor_residual_where_the_field_is_uniform(scene):
    """The fp64 residual that forms the reference's rows one by one (a variable viscosity field: k_bresidual with RefRowInputs on bricks, k_plane_residual_ref on
    the plane layouts) against the one that reads the stored edge factors: the viscosity field is uniform except for ONE node buried in the solid wall -- enough to
    switch the per-row path on, nowhere near a row -- so every factor is what the stored one is and the two runs must deliver the same velocities (the residual kernels
    differ, the arithmetic does not; since round 6 a viscosity FIELD also switches the pairs' weak modes and the wider stall guard on, so the two runs take
    different preconditioners to the same tolerance: <= 5e-6; both layouts)."""
    from flipviscosity3d_amd.capi import Context
    if scene.startswith("sparse"):
        N, nu0 = 64, 200.0
        dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
        layout = 2
    else:
        N, nu0 = 48, 150.0
        dx, solid, P = dense_scene(N, 0.66)
        P = P.copy()
        P[:, 3] = 0.8 * np.sin(7.0 * P[:, 1]) * np.cos(5.0 * P[:, 2]); P[:, 4] = -0.3 * np.cos(6.0 * P[:, 0]); P[:, 5] = 0.5 * np.sin(4.0 * P[:, 0] + 3.0 * P[:, 1])
        layout = 1
    out = []
    for bump in (0.0, 1.0):
        nu = np.full((N + 1, N + 1, N + 1), nu0, np.float32)
        nu[0, 0, 0] += bump                     # a corner node of the domain: solid, no row within reach
        c = Context(N, N, N, dx)
        c.set_solid_sdf(solid); c.set_viscosity(nu)
        c.particles = P
        st = c.substep(0.01)
        v = st["viscosity"]
        assert v["status"] == 0 and v["layout"] in ((2,) if layout == 2 else (0, 1)) and v["correction_iterations"] > 0, v
        out.append(([c.grid(n) for n in "UVW"], v["iterations"]))
        c.close()
    (a, ia), (b, ib) = out
    den = max(float(np.abs(x).max()) for x in a)
    err = max(float(np.abs(x.astype(np.float64) - y).max()) for x, y in zip(a, b)) / den
    print("%s: stored factors %d iterations, per-row factors %d, velocity difference %.2e" % (scene, ia, ib, err))
    assert abs(ia - ib) <= 3 and err <= 5e-6, (ia, ib, err)


def test_viscosity_field_that_is_zero_on_part_of_the_liquid(oracle):
    """nu = 0 below y = 0.42, 200 above (a sharp jump inside the bunny): the inviscid faces are pure mass rows and pin the viscous body along the interface.  One correction
    stage leaves 3.3e-4 there; the library takes two for such fields (k_viscosity.hip: vZeroRegion).  Default parameters against the oracle with its cap lifted: <= 1e-4
    (measured 4e-6)."""
    from flipviscosity3d_amd.capi import Context
    N = 64
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    y = (np.arange(N + 1) * dx)[None, :, None]
    nu = np.ascontiguousarray(np.broadcast_to(np.where(y < 0.42, 0.0, 200.0), (N + 1, N + 1, N + 1)), np.float32)
    o = oracle.OracleSim(N, N, N, dx)
    o.set_solid(solid); o.set_viscosity(nu); o.set_solver_limits(vmaxiter=3000000)
    o.particles = P
    sec, vi, pi = o.substep(0.01)
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu)
    c.particles = P
    st = c.substep(0.01)
    v = st["viscosity"]
    ref = [o.grid(n) for n in "UVW"]
    err = max(float(np.abs(c.grid(n).astype(np.float64) - r).max()) for n, r in zip("UVW", ref)) / max(float(np.abs(r).max()) for r in ref)
    print("nu = 0 | 200: %d iterations (%d correction; oracle %d), status %d, velocity error %.2e" % (v["iterations"], v["correction_iterations"], vi["iterations"], v["status"], err))
    assert vi["status"] == 0 and v["status"] == 0 and v["iterations"] < 700, (vi, v)
    assert err <= 1e-4, err
    c.close()
    o.close()
