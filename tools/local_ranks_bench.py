"""bench.py's multi-rank scenes run with the in-process communicator: all ranks on ONE GPU, one host thread each.
Kernels of different ranks share the device, so the wall time means little; the per-phase GPU times of the ranks show
whether any multi-rank-only code path (halo packing, combine kernels, migration, the split launches) costs more than it
should, and the iteration counts show what the decomposition does to the solvers.

    python tools/local_ranks_bench.py weak   [ranks=2] [size=256]            copies of the bunny scene stacked along k, one slab each
    python tools/local_ranks_bench.py strong [px,py,pz=2,2,2] [size=256] [workload=bunny] [viscosity=5] [key=value ...  flipv_params fields]
                                                                            ONE scene split into blocks (bench.py --scaling strong)"""
import os
import sys
import threading
import time
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_scene, build_workload
from flipviscosity3d_amd import capi, partition

mode = sys.argv[1] if len(sys.argv) > 1 else "weak"
if mode == "weak":
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dx, solid, P = build_scene(N, 5.0)
    solid_g, parts = partition.stack_scene(solid, P, R, N, dx)
    ranges = partition.slab_ranges(N * R, R)
    ctxs = [capi.Context(N, N, N * R, dx, device=0, slab=r) for r in ranges]
    capi.comm_init_local(ctxs)
    nu = 5.0
else:
    dims = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,2,2").split(","))
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    wl = sys.argv[4] if len(sys.argv) > 4 else "bunny"
    nu = float(sys.argv[5]) if len(sys.argv) > 5 else 5.0
    I, J, K, dx, solid_g, P = build_workload(wl, N, on_device=True)
    boxes = partition.block_boxes(I, J, K, dims)
    R = len(boxes)
    ctxs = [capi.Context(I, J, K, dx, device=0, block=b) for b in boxes]
    capi.comm_init_local(ctxs, dims)
    parts = partition.split_particles_boxes(P, dx, boxes, dims)
    print("scene %dx%dx%d, %d particles, blocks %s" % (I, J, K, len(P), dims))
extra = {kv.split("=")[0]: (float(kv.split("=")[1]) if "." in kv.split("=")[1] or "e" in kv.split("=")[1] else int(kv.split("=")[1])) for kv in sys.argv[2:] if "=" in kv}
for c, p in zip(ctxs, parts):
    if extra:
        c.set_params(**extra)
    c.set_solid_sdf(solid_g)
    c.set_viscosity(nu)
    c.particles = p
out = [None] * R
hist = []
def work(r):
    c = ctxs[r]
    for t in range(3):
        st = c.substep(min(c.cfl(), 0.01))
        if r == 0:
            hist.append((st["viscosity"]["iterations"], st["pressure"]["iterations"]))
    out[r] = st
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
for t in th: t.start()
for t in th: t.join()
print("wall %.2f s for 3 substeps of %d ranks sharing one device" % (time.perf_counter() - t0, R), extra)
for r, st in enumerate(out):
    print("rank", r, {k: round(v, 2) for k, v in st["phase_ms"].items()}, "total %.1f" % st["total_ms"], "its", st["viscosity"]["iterations"], st["pressure"]["iterations"],
          "rows", st["viscosity"]["rows"], "particles", ctxs[r].num_particles)
print("iterations per substep (viscosity, pressure):", hist)
if mode != "weak":   # the single domain on the same three substeps
    del ctxs
    ref = capi.Context(I, J, K, dx, device=0)
    if extra:
        ref.set_params(**extra)
    ref.set_solid_sdf(solid_g); ref.set_viscosity(nu); ref.particles = P
    single = []
    for t in range(3):
        st = ref.substep(min(ref.cfl(), 0.01))
        single.append((st["viscosity"]["iterations"], st["pressure"]["iterations"]))
    print("single domain, the same three substeps:           ", single)
    ref.close()
v, p = out[0]["viscosity"], out[0]["pressure"]
print("viscosity solve: layout %d, preconditioner %d, %d iterations (%d in correction stages), status %d / %d; all-reduced by the global hierarchy: %.2f MB once per solve, %.3f MB per iteration" % (
    v["layout"], v["preconditioner"], v["iterations"], v["correction_iterations"], v["status"], v["correction_status"], v["comm_bytes_setup"] / 1e6, v["comm_bytes_per_iteration"] / 1e6))
print("pressure solve: %d iterations; all-reduced: %.2f MB once per solve, %.3f MB per iteration" % (p["iterations"], p["comm_bytes_setup"] / 1e6, p["comm_bytes_per_iteration"] / 1e6))
print("one iteration issues: viscosity %d neighbour exchanges + %d all-reduces, pressure %d + %d" % (v["halo_exchanges_per_iteration"], v["allreduces_per_iteration"], p["halo_exchanges_per_iteration"], p["allreduces_per_iteration"]))
