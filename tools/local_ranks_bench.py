"""The weak-scaling scene of bench.py (copies of the 256^3 bunny drop stacked along k, one slab per rank) run with the
in-process communicator: all ranks on ONE GPU, one host thread each.  Kernels of different ranks share the device, so
the wall time means little; the per-phase GPU times of one rank show whether any multi-rank-only code path (halo
packing, combine kernels, migration) costs more than it should.

    python tools/local_ranks_bench.py [ranks=2] [size=256]"""
import os
import sys
import threading
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_scene
from flipviscosity3d_amd import capi, partition

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dx, solid, P = build_scene(N, 5.0)
solid_g, parts = partition.stack_scene(solid, P, R, N, dx)
ranges = partition.slab_ranges(N * R, R)
ctxs = [capi.Context(N, N, N * R, dx, device=0, slab=r) for r in ranges]
capi.comm_init_local(ctxs)
for c, p in zip(ctxs, parts):
    c.set_solid_sdf(solid_g)
    c.set_viscosity(5.0)
    c.particles = p
out = [None] * R
def work(r):
    c = ctxs[r]
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
    out[r] = st
th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
for t in th: t.start()
for t in th: t.join()
for r, st in enumerate(out):
    print("rank", r, {k: round(v, 2) for k, v in st["phase_ms"].items()}, "total %.1f" % st["total_ms"], "its", st["viscosity"]["iterations"], st["pressure"]["iterations"],
          "particles", ctxs[r].num_particles)
