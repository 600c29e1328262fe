"""CPU, world_size 2, gloo: the host-side logic of the N > 1 path -- slab ranges, particle ownership, the stacked
weak-scaling scene, the unique-id broadcast and the max-over-ranks timing reduction of bench.py.  (The device side of the
decomposition is verified on one GPU with the in-process communicator, tests/test_gpu_multirank.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from flipviscosity3d_amd import partition


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, dx = 16, 1.0 / 16
        rng = np.random.default_rng(0)                      # same scene on every rank
        particles = rng.uniform(0.1, 0.9, (5000, 6)).astype(np.float32)
        solid = rng.standard_normal((N + 1, N + 1, N + 1)).astype(np.float32)
        solid_g, parts = partition.stack_scene(solid, particles, world, N, dx)
        ranges = partition.slab_ranges(N * world, world)
        mine = parts[rank]
        # every particle of copy `rank` lies in slab `rank`
        own = partition.particle_owner(mine, dx, ranges)
        ok = bool((own == rank).all())
        # unique-id style broadcast (bench.py: rank 0 creates the 128 bytes, everybody receives them)
        uid = [bytes(range(128)) if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        # max-over-ranks timing
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # particle conservation across ranks
        n = torch.tensor([len(mine)], dtype=torch.int64)
        dist.all_reduce(n)
        q.put((rank, ok, uid[0] == bytes(range(128)), float(t.item()), int(n.item()), solid_g.shape, ranges[rank]))
    finally:
        dist.destroy_process_group()


def test_two_rank_host_logic():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, uid_ok, tmax, ntot, shape, rng in res:
        assert ok and uid_ok
        assert tmax == 2.0
        assert ntot == 10000
        assert shape == (16 * world + 1, 17, 17)
        assert rng == (16 * rank, 16 * (rank + 1))


def test_slab_ranges_and_ownership():
    assert partition.slab_ranges(10, 3, min_planes=1) == [(0, 4), (4, 7), (7, 10)]
    assert partition.slab_ranges(256, 8)[7] == (224, 256)
    with pytest.raises(ValueError):
        partition.slab_ranges(4, 5)
    with pytest.raises(ValueError):      # thinner than the widest halo (ceil(cfl) + 3 planes): the library would reject it too
        partition.slab_ranges(20, 3)
    with pytest.raises(ValueError):      # a communicator holds at most 32 ranks
        partition.slab_ranges(1024, 33)
    dx = 0.25
    p = np.zeros((5, 6), np.float32)
    p[:, 2] = [0.01, 0.26, 0.74, 0.99, -0.1]
    own = partition.particle_owner(p, dx, [(0, 2), (2, 4)])
    assert own.tolist() == [0, 0, 1, 1, 0]
    parts = partition.split_particles(p, dx, [(0, 2), (2, 4)])
    assert [len(x) for x in parts] == [3, 2]


def test_gather_owned():
    K = 6
    ranges = partition.slab_ranges(K, 2, min_planes=1)
    a = np.zeros((K + 1, 2, 2)); b = np.ones((K + 1, 2, 2))
    g = partition.gather_owned([a, b], ranges, K)
    assert g[:3].sum() == 0 and (g[3:] == 1).all()          # the last rank also owns the closing plane
