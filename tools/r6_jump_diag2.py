"""research (round 6): which part of the default viscosity solve stalls on holdout draw 9 (and 11)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from r6_jump_diag import run
from flipviscosity3d_amd import capi
for d in (9, 11):
    run(d, viscosity_mg_packed_rows=-1)
    run(d, viscosity_mg_packed_rows=-1, reps=2)
    run(d, exact_viscosity_operator=1)
    run(d, exact_viscosity_operator=1, viscosity_mg_packed_rows=-1)
    run(d, viscosity_preconditioner=capi.PRECOND_DIAGONAL, viscosity_max_iterations=20000)
    run(d, viscosity_stage1_factor=1.0)
    run(d, viscosity_stage1_factor=1.0, viscosity_mg_packed_rows=-1)
    run(d, viscosity_mg_min_dim=64)
    run(d, viscosity_velocity_tolerance=-1.0, viscosity_mass_scale=-1.0)
    run(d, viscosity_mg_coarsest_sweeps=64)
    run(d, viscosity_mg_packed_rows=-1, viscosity_mg_coarsest_sweeps=64, verbose=1)

# the slab test that failed with the elimination in (tests/test_gpu_multirank.py::test_slab_decomposition_matches_single_domain[bunny32_viscous-2])
import numpy as np
from helpers import Golden
from flipviscosity3d_amd import partition
from test_gpu_multirank import run_ranks
for polish in (0, -1):
    g = Golden("bunny32_viscous")
    I, J, K = g.dims()
    params = dict(viscosity_max_iterations=5000, viscosity_tolerance=1e-7, pressure_rel_tolerance=1e-7, viscosity_preconditioner=capi.PRECOND_DIAGONAL, viscosity_layout=capi.LAYOUT_SWIZZLED,
                  viscosity_massless_polish=polish, verbose=1)
    ref = capi.Context(I, J, K, g.dx)
    ref.set_solid_sdf(g["solid"]); ref.set_viscosity(g["viscosity"]); ref.set_gravity(*g.gravity); ref.set_params(**params)
    ref.particles = g["particles0"]
    ranges = partition.slab_ranges(K, 2)
    ctxs = [capi.Context(I, J, K, g.dx, device=0, slab=r) for r in ranges]
    capi.comm_init_local(ctxs)
    for c, p in zip(ctxs, partition.split_particles(g["particles0"], g.dx, ranges)):
        c.set_solid_sdf(g["solid"]); c.set_viscosity(g["viscosity"]); c.set_gravity(*g.gravity); c.set_params(**params)
        c.particles = p
    for t in range(g.nsub):
        ref.substep(g.dt)
        run_ranks(ctxs, lambda r, c: c.substep(g.dt))
        phi = partition.gather_owned([c.grid("LIQUID_PHI") for c in ctxs], ranges, K)
        pr = ref.grid("LIQUID_PHI")
        dv = max(np.abs(partition.gather_owned([c.grid(n) for c in ctxs], ranges, K) - ref.grid(n)).max() for n in "UVW")
        print("polish %d substep %d: phi differs on %d entries (max %.2e); velocity max abs diff %.3e" % (polish, t, (phi != pr).sum(), np.abs(phi - pr).max(), dv), flush=True)
