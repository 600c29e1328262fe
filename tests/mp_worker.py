"""One RANK of the multi-process rehearsal (tests/test_gpu_multiprocess.py; launched by torch.distributed.run, never imported by pytest): the path the 8-GPU run takes --
launcher -> rendezvous -> a block context per PROCESS -> halos, reductions, particle migration -> gathering the result -- with the host-staged communicator over gloo, so that
both ranks can share device 0 of a one-GPU box.  Rank 0 writes what the test compares: iteration counts per substep and the assembled velocities."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, N, nsub, dims = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), tuple(int(v) for v in sys.argv[4].split(","))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flipviscosity3d_amd import capi, partition
    from test_oracle_compact_golden import build_host_scene
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    boxes = partition.block_boxes(N, N, N, dims)
    c = capi.Context(N, N, N, dx, device=torch.cuda.current_device(), block=boxes[rank])
    c.comm_init_host(capi.torch_distributed_callbacks(dist), rank, dims)
    c.set_solid_sdf(solid)
    c.set_viscosity(5.0)
    c.particles = partition.split_particles_boxes(P, dx, boxes, dims)[rank]
    stats = []
    for t in range(nsub):
        dt = min(c.cfl(), 0.01)          # (a max all-reduce over the ranks)
        st = c.substep(dt)
        stats.append((dt, st["viscosity"]["iterations"], st["viscosity"]["status"], st["pressure"]["iterations"], st["pressure"]["status"], st["viscosity"]["residual"],
                      st["viscosity"]["halo_exchanges_per_iteration"], st["viscosity"]["allreduces_per_iteration"], c.num_particles))
    # gather (test plumbing): a read on a block context writes the entries the rank owns into a zero-filled full-size grid -- the ranks' grids add up to the domain's
    mine = dict(stats=stats, grids={n: c.grid(n) for n in "UVW"})
    every = [None] * world
    dist.all_gather_object(every, mine)
    if rank == 0:
        full = {n: sum(e["grids"][n].astype(np.float64) for e in every).astype(np.float32) for n in "UVW"}
        np.savez(out, U=full["U"], V=full["V"], W=full["W"], stats=np.array([e["stats"] for e in every], np.float64))
    dist.barrier()
    c.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
