"""GPU (-m gpu): LATE states -- liquid resting on / sliding along the wall, the fringe of a splash, long after the impact -- with DEFAULT parameters against the
solution of the reference's linear systems.

Every default-parameter parity test before round 5 started from rest or at most 25 substeps in.  A simulation spends its life elsewhere: the 64^3 bunny lies on the
container wall from substep ~20 on.  There the reference's stop test, max|r| <= 1e-6 max|rhs| (pcgsolver.h:259-272), does not bound the velocity error: light, weakly
attached parts of the liquid (films, specks: control volumes summing to a few per cent of a cell) have residual = mass x error, and the reference's OWN 1e-6 iterate is
1e-4 ... 3e-1 of max|u| away from the solution of its system on one substep in eight (profiles/r5/late_states.log).  So these tests compare with the oracle run to
1e-13 / 1e-13 -- at 1e-10 it has not converged in such states either (nu = 5 after 69 substeps: 1.5 max|u| between its 1e-10 and its 1e-13 answer on nine faces) -- and
print the distance of the reference at its defaults beside the GPU's.

The state of a FLIP substep is its particles: the oracle carries the scene there at ITS defaults (deterministic: a checksum of the state is asserted), then ONE substep
is taken on both sides from the oracle's particles, the GPU with NO field of flipv_params set.  What makes the default pass is the velocity criterion of the solve's
last loop, the mass scale of its tolerances and the massless clusters solved apart afterwards (flipv_params.viscosity_velocity_tolerance, viscosity_mass_scale,
viscosity_massless_polish): round 4's rule is asserted to miss the bar on the same states.
Bar: end-of-substep velocities <= 1e-4 relative max-norm (BASELINE.json north_star), every face."""
import numpy as np
import pytest

from helpers import rel_maxnorm3
from test_oracle_compact_golden import build_host_scene

pytestmark = pytest.mark.gpu

VEL_TOL = 1e-4
SCENES = {"bunny": (("sphere_large.ply", True), ["stanford_bunny.ply"]), "honey": (None, ["rod.ply", "sheet.ply"])}
_chains = {}


def late_state(oracle, scene, N, nu, nsub):
    """the oracle's particles after `nsub` substeps of dt = 0.01 at its defaults (chains are shared between the cases of one scene / viscosity)"""
    key = (scene, N, nu)
    if key not in _chains:
        dx, solid, P = build_host_scene(N, *SCENES[scene])
        o = oracle.OracleSim(N, N, N, dx)
        o.set_solid(solid); o.set_viscosity(nu)
        o.particles = P
        _chains[key] = dict(o=o, t=0, dx=dx, solid=solid, states={0: P.copy()})
    ch = _chains[key]
    assert nsub >= ch["t"] or nsub in ch["states"], "ask for the states of one chain in ascending order"
    while ch["t"] < nsub:
        ch["o"].substep(0.01)
        ch["t"] += 1
        ch["states"][ch["t"]] = ch["o"].particles.copy()
    return ch["dx"], ch["solid"], ch["states"][nsub]


def converged_and_default_reference(oracle, N, dx, solid, nu, P):
    ref = {}
    for name, lim in (("converged", dict(vmaxiter=3000000, vtol=1e-13, ptol=1e-13)), ("defaults", None)):
        o = oracle.OracleSim(N, N, N, dx)
        o.set_solid(solid); o.set_viscosity(nu)
        if lim:
            o.set_solver_limits(**lim)
        o.particles = P
        sec, vi, pi = o.substep(0.01)
        ref[name] = ([o.grid(n) for n in "UVW"], vi["iterations"])
        o.close()
    return ref


def gpu_substep(N, dx, solid, nu, P, **params):
    from flipviscosity3d_amd.capi import Context
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid); c.set_viscosity(nu)
    if params:
        c.set_params(**params)
    c.particles = P
    st = c.substep(0.01)
    uvw = [c.grid(n) for n in "UVW"]
    c.close()
    return uvw, st


# (scene, N, nu, substeps before, does round 4's rule miss the bar there?)   -- the first three rows are the states of profiles/r4/tight_oracle_scan.log
CASES = [("bunny", 64, 200.0, 40, True), ("bunny", 64, 200.0, 70, True),
         ("bunny", 64, 5.0, 18, True), ("bunny", 64, 5.0, 69, True), ("bunny", 64, 5.0, 86, True), ("bunny", 64, 5.0, 110, False),   # (18: the splash at 12 m/s -- a massless two-row cluster on a used face, 1.7e-4 until such clusters were solved apart)
         ("bunny", 64, 1e-3, 3, False), ("bunny", 64, 1e-3, 25, True), ("bunny", 64, 1e-3, 36, True),
         ("honey", 64, 50.0, 25, False)]
# float64 sum over the state's particle array as the build container's oracle produced it: the chain is deterministic, so the box must reproduce it bit for bit
STATE_SUM = {("bunny", 200.0, 40): 83295.08316674425, ("bunny", 200.0, 70): 83492.74480260964, ("bunny", 5.0, 18): -27345.86725991894, ("bunny", 5.0, 69): 81604.05777857917,
             ("bunny", 5.0, 86): 82051.0415431282, ("bunny", 5.0, 110): 82408.24963767857, ("bunny", 1e-3, 3): 82574.48443527038,
             ("bunny", 1e-3, 25): 25054.50262722154, ("bunny", 1e-3, 36): 118639.10160814502, ("honey", 50.0, 25): 35640.77097503832}


@pytest.mark.parametrize("scene,N,nu,nsub,old_rule_misses", CASES)
def test_late_state_default_parameters_against_the_converged_reference(oracle, scene, N, nu, nsub, old_rule_misses):
    dx, solid, P = late_state(oracle, scene, N, nu, nsub)
    assert float(P.astype(np.float64).sum()) == STATE_SUM[(scene, nu, nsub)]      # the state the cases were chosen on
    ref = converged_and_default_reference(oracle, N, dx, solid, nu, P)
    conv, its_conv = ref["converged"]
    dflt, its_dflt = ref["defaults"]
    assert max(np.abs(a).max() for a in conv) > 0.05          # a moving / settling liquid, not a trivial state
    uvw, st = gpu_substep(N, dx, solid, nu, P)                 # NO parameter set
    v = st["viscosity"]
    err = rel_maxnorm3(uvw, conv)
    err_ref = rel_maxnorm3(dflt, conv)
    old, st_old = gpu_substep(N, dx, solid, nu, P, viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1, viscosity_pair_correction=-1)   # round 4's rule: max|r| against max|rhs| alone, no cluster solve, no pair correction, repeated rows left in the system
    err_old = rel_maxnorm3(old, conv)
    print("%s %d^3 nu %g after %d substeps: GPU default %.2e from the converged reference in %d viscosity iterations (velocity step %.1e, status %d) | round 4's rule %.2e in %d | "
          "the reference at its defaults %.2e in %d (converged: %d)" % (scene, N, nu, nsub, err, v["iterations"], v["velocity_step"], v["status"], err_old,
                                                                      st_old["viscosity"]["iterations"], err_ref, its_dflt, its_conv))
    assert v["status"] == 0 and st["pressure"]["status"] == 0, st
    assert err <= VEL_TOL, err
    if old_rule_misses:
        assert err_old > VEL_TOL, err_old      # (what the criterion is for; if this starts passing the case no longer tests anything)


def test_velocity_criterion_costs_nothing_on_a_compact_falling_body(oracle):
    """first substeps of config 1's scene (64^3 bunny, nu = 5, falling from rest): the iteration has settled when the residual test passes, so the default solve
    takes at most a few iterations more than round 4's rule and lands in the same place"""
    N, nu = 64, 5.0
    dx, solid, P = late_state(oracle, "bunny", N, nu, 0)
    a, sa = gpu_substep(N, dx, solid, nu, P)
    b, sb = gpu_substep(N, dx, solid, nu, P, viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1, viscosity_pair_correction=-1)
    print("from rest: %d iterations with the velocity criterion, %d without; difference %.2e" % (sa["viscosity"]["iterations"], sb["viscosity"]["iterations"], rel_maxnorm3(a, b)))
    assert sa["viscosity"]["iterations"] <= sb["viscosity"]["iterations"] + 8
    assert rel_maxnorm3(a, b) <= 2e-5




def test_massless_clusters_are_what_the_splash_state_needs(oracle):
    """64^3 bunny at nu = 5 after 18 substeps (the splash at 12 m/s): with flipv_params.viscosity_massless_polish = -1 and everything else at its default the substep is
    1.7e-4 from the converged reference on 27 faces -- ONE face the substep uses sits in a two-row cluster without own volume whose split no fp32 iteration sees, and the
    projection and the extrapolation spread it --, with the cluster solve (the default) 1.5e-6."""
    N, nu = 64, 5.0
    dx, solid, P = late_state(oracle, "bunny", N, nu, 18)
    conv, _ = converged_and_default_reference(oracle, N, dx, solid, nu, P)["converged"]
    a, sa = gpu_substep(N, dx, solid, nu, P)
    b, sb = gpu_substep(N, dx, solid, nu, P, viscosity_massless_polish=-1, viscosity_pair_correction=-1)
    b2, sb2 = gpu_substep(N, dx, solid, nu, P, viscosity_massless_polish=-1)
    ea, eb = rel_maxnorm3(a, conv), rel_maxnorm3(b, conv)
    print("round 6: without the cluster solve but with the pairs' weak modes in the preconditioner: %.2e (%d iterations)" % (rel_maxnorm3(b2, conv), sb2["viscosity"]["iterations"]))
    print("default %.2e | without the cluster solve and the pair correction %.2e (%d / %d iterations)" % (ea, eb, sa["viscosity"]["iterations"], sb["viscosity"]["iterations"]))
    assert ea <= 2e-5, ea
    assert eb > VEL_TOL, eb      # (if this starts passing the case no longer shows anything)


def test_stall_exit_is_opt_in_and_shortens_a_plateau(oracle):
    """flipv_params.viscosity_velocity_stall_ratio (off by default): on the 64^3 bunny at nu = 200 after 40 substeps -- lying on the wall, the velocity criterion holding
    the correction stage on a plateau of the iteration's movement -- 0.5 ends the solve earlier and stays within the bar HERE (1.7e-5; the default 2e-6).  Why it is not the
    default: include/flipv.h."""
    N, nu = 64, 200.0
    dx, solid, P = late_state(oracle, "bunny", N, nu, 40)
    conv, _ = converged_and_default_reference(oracle, N, dx, solid, nu, P)["converged"]
    a, sa = gpu_substep(N, dx, solid, nu, P)
    b, sb = gpu_substep(N, dx, solid, nu, P, viscosity_velocity_stall_ratio=0.5)
    ea, eb = rel_maxnorm3(a, conv), rel_maxnorm3(b, conv)
    print("default %.2e in %d iterations | stall exit at 0.5: %.2e in %d" % (ea, sa["viscosity"]["iterations"], eb, sb["viscosity"]["iterations"]))
    assert sb["viscosity"]["iterations"] <= sa["viscosity"]["iterations"] and sb["viscosity"]["status"] == 0   # (round 6: with the pairs' weak modes in the preconditioner the plateau of this state is gone -- 63 iterations either way; round 5: 62 against 49)
    assert ea <= VEL_TOL and eb <= VEL_TOL


def test_late_state_128_against_the_reference_golden():
    """(d) of VERDICT r4's item 1: the same question at 128^3 against the REFERENCE itself (tests/golden/bunny128_nu200_late: the compiled reference's state after 45 of
    its own substeps at nu = 200, nu dt/dx^2 = 32 768, and its answer from there with the viscosity tolerance at 1e-13 -- 2 870 iterations; the oracle is pinned to the
    same fixture in tests/test_oracle_compact_golden.py).  105 000 probe faces per component, the 5 000 of largest |u| among them; per-octant particle checksums.
    GPU with NO parameter set: <= 1e-4; round 4's rule and the reference at its own defaults (4.1e-4) are printed beside it."""
    import os
    from helpers import GOLDEN
    path = os.path.join(GOLDEN, "bunny128_nu200_late.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not built")
    g = np.load(path)
    N = int(g["I"])
    dx, solid, P0 = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    assert np.float64(solid.astype(np.float64).sum()) == g["solid_sum"]
    den = float(g["maxabs"])

    def probe_error(uvw):
        return max(float(np.abs(a.reshape(-1)[g["probe_idx_" + c]].astype(np.float64) - g["probe_val_" + c]).max()) for c, a in zip("UVW", uvw)) / den
    uvw, st = gpu_substep(N, dx, solid, float(g["nu"]), g["state"])
    old, st_old = gpu_substep(N, dx, solid, float(g["nu"]), g["state"], viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0, viscosity_massless_polish=-1, viscosity_pair_correction=-1)
    err, err_old = probe_error(uvw), probe_error(old)
    v = st["viscosity"]
    print("128^3 nu 200 after 45 reference substeps: GPU default %.2e from the reference at 1e-13 in %d viscosity iterations (status %d, velocity step %.1e) | round 4's rule %.2e in %d | "
          "the reference at its defaults %.2e in %d (converged: %d)" % (err, v["iterations"], v["status"], v["velocity_step"], err_old, st_old["viscosity"]["iterations"],
                                                                      float(g["defaults_vs_converged"]), int(g["defaults_visc_iters"]), int(g["visc_iters"])))
    assert v["status"] == 0 and st["pressure"]["status"] == 0, st
    assert err <= VEL_TOL, err


@pytest.mark.parametrize("dims", [(1, 1, 2), (2, 2, 2)])
def test_splash_state_on_blocks(oracle, dims):
    """the 64^3 splash state (nu = 5 after 18 substeps) -- the case the massless-cluster solve exists for -- on 1 x 1 x 2 and 2 x 2 x 2 blocks, whose cut planes
    (k = 32; i, j, k = 32) pass through the splash.  Until round 6 a block context skipped every cluster within one entry of ANY face of its owned box, the domain's walls included
    (ADVICE r5); now only towards sides that have a neighbouring rank.  NO parameter set: <= 1e-4 against the oracle at 1e-13, every rank the same solve."""
    from test_gpu_multirank import assemble, run_ranks
    from test_gpu_multirank_default import assert_same_solve_on_every_rank, make_blocks
    N, nu = 64, 5.0
    dx, solid, P = late_state(oracle, "bunny", N, nu, 18)
    conv, _ = converged_and_default_reference(oracle, N, dx, solid, nu, P)["converged"]
    ctxs = make_blocks(N, dx, solid, P, nu, dims)
    sts = run_ranks(ctxs, lambda r, c: c.substep(0.01))
    assert_same_solve_on_every_rank(sts)
    err = rel_maxnorm3([assemble(ctxs, n) for n in "UVW"], conv)
    v = sts[0]["viscosity"]
    print("splash state on %s blocks: %.2e in %d viscosity iterations, status %d, %d rows eliminated on rank 0" % (dims, err, v["iterations"], v["status"], v["eliminated_rows"]))
    assert v["status"] == 0 and err <= VEL_TOL, (err, v)
    for c in ctxs:
        c.close()


def test_close_chains():
    for ch in _chains.values():
        ch["o"].close()
    _chains.clear()
