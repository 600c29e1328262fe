"""several substeps of a block-decomposed scene on in-process ranks, with the solver statistics of every substep:
    python tools/local_ranks_soak.py px,py,pz size workload viscosity substeps [key=value ...]"""
import os, sys, threading
sys.path.insert(0, os.getcwd())
import numpy as np
from bench import build_workload
from flipviscosity3d_amd import capi, partition
dims = tuple(int(v) for v in sys.argv[1].split(","))
N, wl, nu, nsub = int(sys.argv[2]), sys.argv[3], float(sys.argv[4]), int(sys.argv[5])
extra = {kv.split("=")[0]: (float(kv.split("=")[1]) if "." in kv.split("=")[1] or "e" in kv.split("=")[1] else int(kv.split("=")[1])) for kv in sys.argv[6:]}
I, J, K, dx, solid, P = build_workload(wl, N, on_device=True)
boxes = partition.block_boxes(I, J, K, dims)
R = len(boxes)
ctxs = [capi.Context(I, J, K, dx, device=0, block=b) for b in boxes]
capi.comm_init_local(ctxs, dims)
parts = partition.split_particles_boxes(P, dx, boxes, dims)
for c, p in zip(ctxs, parts):
    if extra: c.set_params(**extra)
    c.set_solid_sdf(solid); c.set_viscosity(nu); c.particles = p
log = [[] for _ in range(R)]
def work(r):
    c = ctxs[r]
    for t in range(nsub):
        st = c.substep(min(c.cfl(), 0.01))
        log[r].append((st["viscosity"]["iterations"], st["viscosity"]["status"], st["viscosity"]["preconditioner"], st["pressure"]["iterations"], st["pressure"]["status"], c.num_particles))
th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
for t in th: t.start()
for t in th: t.join()
same = all(tuple(x[:5] for x in log[r]) == tuple(x[:5] for x in log[0]) for r in range(R))
print("%s blocks of %s %dx%dx%d nu %g, %d substeps %s: every rank took the same solver path: %s" % (dims, wl, I, J, K, nu, nsub, extra, same))
import collections
print("viscosity status %s, preconditioner %s; pressure status %s" % (dict(collections.Counter(x[1] for x in log[0])), dict(collections.Counter(x[2] for x in log[0])), dict(collections.Counter(x[4] for x in log[0]))))
print("viscosity iterations", [x[0] for x in log[0]], "status", sorted(set(x[1] for x in log[0])), "preconditioner", sorted(set(x[2] for x in log[0])))
print("pressure iterations ", [x[3] for x in log[0]], "status", sorted(set(x[4] for x in log[0])))
print("particles per rank at the end", [l[-1][5] for l in log], "total", sum(l[-1][5] for l in log), "of", len(P))
