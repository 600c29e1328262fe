"""GPU (-m gpu): the multi-PROCESS path rehearsed on one GPU.  Everything the 8-GPU run does except the transport: `torch.distributed.run` starts
one process per rank, the ranks rendezvous, each creates ITS block context (flipv_create_block), attaches a communicator, and the substeps exchange halos with the <= 26
neighbours, all-reduce the PCG scalars and the CFL maximum, and migrate particles.  The transport is the host-staged communicator (flipv_comm_init_host_grid: callbacks over
torch.distributed gloo), so both ranks can sit on device 0.  The ranks are started by a helper process that conftest.py created BEFORE anything in this session touched the GPU
(a process that has initialised the GPU must not exec another program on this pool).

Checked: every rank reports the same solve (iterations, status, all-reduced residual bits) on every substep; iteration counts within a few of the single domain's; the assembled
velocities within 5e-5 of the single domain's; no particle is lost; `bench.py --gpus 2 --comm host` prints a result line through the driver's own launch path."""
import json
import os
import sys

import numpy as np
import pytest

from helpers import ROOT, rel_maxnorm3
from test_oracle_compact_golden import build_host_scene

pytestmark = pytest.mark.gpu


def launch(cmd, timeout=600):
    from conftest import spawn_helper_run
    rc, out = spawn_helper_run(cmd, timeout)
    assert rc == 0, out[-4000:]
    return out


def torchrun(nproc, script_args):
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args


@pytest.mark.parametrize("dims", [(1, 1, 2), (2, 2, 1), (2, 2, 2)])      # (2 x 2 x 2: BASELINE configs[3]'s decomposition, eight processes on the one device)
def test_two_four_and_eight_processes_on_one_device_run_the_single_domains_solve(tmp_path, dims):
    from flipviscosity3d_amd.capi import Context
    N, nsub = 64, 3
    world = dims[0] * dims[1] * dims[2]
    out = str(tmp_path / "ranks.npz")
    launch(torchrun(world, [os.path.join(ROOT, "tests", "mp_worker.py"), out, str(N), str(nsub), "%d,%d,%d" % dims]))
    z = np.load(out)
    stats = z["stats"]          # (rank, substep, [dt, visc its, visc status, pressure its, pressure status, visc residual, exchanges, all-reduces, particles])
    dx, solid, P = build_host_scene(N, ("sphere_large.ply", True), ["stanford_bunny.ply"])
    ref = Context(N, N, N, dx)
    ref.set_solid_sdf(solid); ref.set_viscosity(5.0); ref.particles = P
    for t in range(nsub):
        dt = min(ref.cfl(), 0.01)
        st = ref.substep(dt)
        assert np.all(stats[:, t, 0] == stats[0, t, 0]) and abs(float(stats[0, t, 0]) - dt) <= 1e-6 * dt      # the CFL maximum over the ranks
        for col in (1, 2, 3, 4, 5, 6, 7):
            assert np.all(stats[:, t, col] == stats[0, t, col]), (t, col, stats[:, t, col])                     # every rank: the same solve, the same all-reduced bits
        print("%s processes, substep %d: viscosity %d iterations (single domain %d), pressure %d (%d); %d neighbour exchanges + %d all-reduces per viscosity iteration" % (
            dims, t, stats[0, t, 1], st["viscosity"]["iterations"], stats[0, t, 3], st["pressure"]["iterations"], stats[0, t, 6], stats[0, t, 7]))
        assert stats[0, t, 2] == 0 and stats[0, t, 4] == 0
        assert abs(stats[0, t, 1] - st["viscosity"]["iterations"]) <= 8 and abs(stats[0, t, 3] - st["pressure"]["iterations"]) <= 3
        assert stats[0, t, 6] >= 5 and stats[0, t, 7] == 3
        assert stats[:, t, 8].sum() == len(P)                                                                   # migration loses no particle
    err = rel_maxnorm3([z[n] for n in "UVW"], [ref.grid(n) for n in "UVW"])
    print("   velocity difference to the single domain after %d substeps: %.2e" % (nsub, err))
    assert err <= 5e-5, err
    ref.close()


def test_bench_two_ranks_through_the_drivers_launch_path():
    """`python -m torch.distributed.run ... bench.py --gpus 2 --comm host`: the launch line the driver uses for N > 1, with the host-staged transport -- one result line, two
    ranks, every timed solve complete"""
    out = launch(torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "host", "--size", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-dense", "--no-strict"]))
    lines = [ln for ln in out.splitlines() if ln.lstrip().startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    print({k: d[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, d["config"]["parallelism"])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "HOST-STAGED" in d["config"]["parallelism"]
