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


def test_block_boxes_and_ownership():
    """tensor-product block decomposition (BASELINE configs[3]: 2 x 2 x 2; configs[4]: 8 slabs along i)"""
    b = partition.block_boxes(512, 512, 512, (2, 2, 2))
    assert len(b) == 8 and b[0] == ((0, 0, 0), (256, 256, 256)) and b[7] == ((256, 256, 256), (512, 512, 512))
    assert b[1] == ((256, 0, 0), (512, 256, 256))                      # x is the fastest rank coordinate
    s = partition.block_boxes(1024, 512, 512, (8, 1, 1))
    assert [x[0][0] for x in s] == [128 * r for r in range(8)] and all(x[1][1:] == (512, 512) for x in s)
    # cuts along i are multiples of 8 whatever the size
    for I in (70, 100, 250):
        for lo, hi in partition.axis_cuts(I, 3, 8, 8):
            assert lo % 8 == 0 and (hi % 8 == 0 or hi == I)
    with pytest.raises(ValueError):
        partition.block_boxes(24, 24, 24, (4, 1, 1))                    # 6-cell blocks: thinner than the widest halo
    with pytest.raises(ValueError):
        partition.block_boxes(512, 512, 512, (4, 4, 4))                 # 64 ranks: more than a communicator takes
    # every particle has exactly one owner, and it is the block that holds its cell
    dx = 1.0 / 32
    rng = np.random.default_rng(3)
    p = rng.uniform(-0.05, 1.05, (20000, 6)).astype(np.float32)        # some outside the domain: they go to the end blocks
    dims = (2, 2, 2)
    boxes = partition.block_boxes(32, 32, 32, dims)
    own = partition.box_owner(p, dx, boxes, dims)
    cell = np.floor(p[:, :3].astype(np.float64) / float(np.float32(dx))).astype(int)
    for r, (lo, hi) in enumerate(boxes):
        m = own == r
        for a in range(3):
            inside_lo = (cell[m, a] >= lo[a]) | (lo[a] == 0)
            inside_hi = (cell[m, a] < hi[a]) | (hi[a] == 32)
            assert inside_lo.all() and inside_hi.all()
    parts = partition.split_particles_boxes(p, dx, boxes, dims)
    assert sum(len(x) for x in parts) == len(p)


def _block_worker(rank, world, port, q):
    """each process owns one block context of a 2-block decomposition and drives it through the library's RCCL backend --
    only where a GPU is visible; on a CPU-only host the workers check the host-side plumbing and report 'no device'"""
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    try:
        from flipviscosity3d_amd import capi
        N, dx = 32, 1.0 / 32
        dims = (1, 1, world)
        boxes = partition.block_boxes(N, N, N, dims)
        ndev = torch.cuda.device_count()                  # (counting devices does not initialise the GPU)
        have_gpu = ndev >= world                          # RCCL refuses two ranks on one device
        out = {"rank": rank, "box": boxes[rank], "gpu": have_gpu, "ndev": ndev}
        if ndev == 1:
            # one device for two processes: no communicator, but each process can still own its block on that device --
            # rank-local allocation, box I/O and the communication-free operators
            c = capi.Context(N, N, N, dx, device=0, block=boxes[rank])
            assert c.block_range() == boxes[rank]
            rng = np.random.default_rng(0)
            p = np.zeros((4000, 6), np.float32)
            p[:, :3] = rng.uniform(0.3, 0.7, (4000, 3))
            c.particles = partition.split_particles_boxes(p, dx, boxes, dims)[rank]
            c.particle_sdf()
            phi = c.read_box("LIQUID_PHI")
            lo, hi = c.grid_box("LIQUID_PHI", 0)
            out.update(shape=phi.shape, want=(hi[2] - lo[2], hi[1] - lo[1], hi[0] - lo[0]), liquid=int((phi < 0).sum()))
            c.close()
        elif have_gpu:
            c = capi.Context(N, N, N, dx, device=rank, block=boxes[rank])
            uid = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            c.comm_init_rccl(uid[0], rank, world, dims)
            rng = np.random.default_rng(0)
            p = np.zeros((4000, 6), np.float32)
            p[:, :3] = rng.uniform(0.3, 0.7, (4000, 3))
            p[:, 5] = 0.5 if rank == 0 else -0.5
            mine = partition.split_particles_boxes(p, dx, boxes, dims)[rank]
            c.set_viscosity(0.5)
            c.particles = mine
            for _ in range(2):
                st = c.substep(0.01)
            n = torch.tensor([c.num_particles], dtype=torch.int64)
            dist.all_reduce(n)
            own = partition.box_owner(c.particles, dx, boxes, dims)
            out.update(total=int(n.item()), owned_ok=bool((own == rank).all()), visc=st["viscosity"]["iterations"], pres=st["pressure"]["iterations"])
            c.comm_finalize()
            c.close()
        else:
            try:
                capi.Context(N, N, N, dx, device=0, block=boxes[rank])
                out["error"] = "a context was created without a device"
            except (capi.FlipvError, OSError) as e:      # no CPU path: the library refuses loudly
                out["refused"] = str(e)
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_two_processes_each_own_a_block():
    """world_size 2: two PROCESSES, one block context each.  With two GPUs the blocks exchange halos, all-reduce the PCG scalars
    and migrate particles through the library's RCCL backend; with one, each process owns its block on the shared device (RCCL
    refuses two ranks on one device); without any the library must refuse to create a context (it has no CPU path) and only
    the host-side decomposition is checked.  NOT part of the -m gpu suite: the workers are started with the `spawn` method
    (fork + exec), which must not happen from a process that has already initialised the GPU -- run this file on its own
    (python -m pytest tests/test_dist_gloo.py) on a GPU machine."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_block_worker, args=(r, world, port, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted((q.get(timeout=150) for _ in range(world)), key=lambda d: d["rank"])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:      # never leave a worker behind (a stuck rendezvous would otherwise outlive the test run)
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    assert res[0]["box"] == ((0, 0, 0), (32, 32, 16)) and res[1]["box"] == ((0, 0, 16), (32, 32, 32))
    if res[0]["gpu"]:
        for r in res:
            assert r["total"] == 4000 and r["owned_ok"]
        assert res[0]["visc"] == res[1]["visc"] and res[0]["pres"] == res[1]["pres"]   # every rank takes the same solver decisions
    elif res[0]["ndev"] == 1:
        for r in res:
            assert r["shape"] == r["want"] == (16, 32, 32) and r["liquid"] > 0
    else:
        for r in res:
            assert "refused" in r and "error" not in r


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment (the shape of the driver's command) starts two ranks as child
    processes and relays exactly rank 0's JSON line; a failing rank makes the exit code non-zero.  Plumbing only (gloo, no GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["sum"] == 3.0


def _callback_worker(rank, world, port, q):
    """the flipv_host_comm callbacks of capi.torch_distributed_callbacks, called the way libflipv's HostComm calls them (through the C function pointers), without a GPU"""
    import ctypes as C
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flipviscosity3d_amd import capi
        cb = capi.torch_distributed_callbacks(dist)
        peer = 1 - rank
        # one group of three operations towards the same peer (their order pairs them): sizes 40 / 0 / 8 bytes one way, 24 / 16 / 0 the other
        ssz = [40, 0, 8] if rank == 0 else [24, 16, 0]
        rsz = [24, 16, 0] if rank == 0 else [40, 0, 8]
        sbuf = [np.full(max(n, 1), 10 * rank + m, np.uint8) for m, n in enumerate(ssz)]
        rbuf = [np.zeros(max(n, 1), np.uint8) for n in rsz]
        n = 3
        rc = cb.exchange(None, n, (C.c_int * n)(peer, peer, peer), (C.c_void_p * n)(*[b.ctypes.data for b in sbuf]), (C.c_size_t * n)(*ssz),
                         (C.c_void_p * n)(*[b.ctypes.data for b in rbuf]), (C.c_size_t * n)(*rsz))
        ok = rc == 0 and all((rbuf[m][:rsz[m]] == 10 * peer + m).all() for m in range(n))
        d = np.arange(5, dtype=np.float64) + rank
        f = (np.arange(7, dtype=np.float32) + 1) * (rank + 1)
        rc2 = cb.allreduce_sum_f64(None, d.ctypes.data_as(C.POINTER(C.c_double)), 5)
        rc3 = cb.allreduce_sum_f32(None, f.ctypes.data_as(C.POINTER(C.c_float)), 7)
        rc4 = cb.barrier(None)
        q.put((rank, ok, rc2 == 0 and np.array_equal(d, 2 * np.arange(5.0) + 1), rc3 == 0 and np.array_equal(f, 3 * (np.arange(7, dtype=np.float32) + 1)), rc4 == 0))
    finally:
        dist.destroy_process_group()


def test_host_communicator_callbacks_over_gloo():
    """flipv_host_comm (include/flipv.h) as bench.py --comm host and tests/test_gpu_multiprocess.py supply it: a group of sends / receives whose m-th operation towards a peer pairs
    with that peer's m-th towards this rank (empty sides included), the two all-reduces, the barrier -- two processes over gloo, no GPU"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_callback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ok64, ok32, okb in got:
        assert ok and ok64 and ok32 and okb, (rank, ok, ok64, ok32, okb)
