#!/usr/bin/env python3
"""bench.py -- MCells/s per FLIP substep on MI355X (BASELINE.json metric), one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--viscosity 5.0]

Workload (N=1): BASELINE.json configs[2] -- 256^3 grid, stanford_bunny.ply dropped inside the inverted
sphere_large.ply container, viscosity 5 at every node, gravity (0,-9.81,0), 8 jittered particles per cell
(counter-based RNG, seed 0), full substep: particle SDF + P2G + extrapolation + body force + variational
viscosity PCG (reference tolerance 1e-6 and cap 700; the library's default never returns an iterate stopped at the cap where a converged
one is affordable: the multigrid-preconditioned two-stage solve -- the exact operator's system to 3 000 x the tolerance, then the fp64 defect
correction towards the reference's float-rounded operator -- takes 60-70 iterations in the stiff start and 15-30 once the liquid moves, so the
timed substeps are "equal-accuracy" ones, mode B of SURVEY.md 8d: `mode_b` on the result line says whether every timed solve completed every
stage, `mode_b_strict` is the same window re-run with stage 1 taken to the reference's own 1e-6 -- flipv_params.viscosity_stage1_factor = 1) +
pressure PCG + extrapolation + constrain + G2P/RK2 advection.  A "step" is one substep of
min(CFL step, 0.01 s) exactly as FluidSimulation::advance takes them (fluidsimulation.cpp:138-167).
Inputs are resident in HBM before the timed region; value = grid cells / wall seconds per substep.
(The face weights and face states, functions of the solid SDF alone, are cached across substeps -- the reference recomputes
them every substep, fluidsimulation.cpp:585; ~0.2 ms of a 31-37 ms substep)

Extra objects on the JSON line:
  roofline     -- the dominant kernel of the substep, the matrix-free viscosity SpMV (on this sparse scene the brick-layout kernel
                  k_bvisc_spmv: the PCG's q = A p and, with other epilogues, the multigrid's four fine-level sweeps).  Unit = one MAC
                  cell's worth of unknowns (3 rows), 52 algorithmic bytes per unit; units per launch = rows of the
                  system / 3; duration measured with HIP events on the library's own stream around every 8th launch of the PCG's
                  own SpMV in an untimed second pass.
  roofline_dense -- the same two SpMV kernels on a completely filled 256^3 box (SURVEY.md 8d "pure kernel roofline
                  runs"), where a launch streams 0.4-0.9 GB and the HBM bound is the relevant one; measured live.
  cpu_baseline -- the reference itself (oracle/_ref, kind "reference") or our C restatement (kind "port"), one thread, timed here on the
                  metric's own scene AT ITS OWN SIZE: ONE substep of the 256^3 bunny drop from rest (~3 minutes of one host core; --cpu-size
                  picks another size).  cpu_baseline_128: the same at 128^3 (two substeps, ~12 s).  cpu_baseline_omp: the OpenMP build of the
                  C restatement at 128^3 on 8 / 16 / 32 / 64 threads (OMP_PLACES=cores), the best of them -- the reference's MIC(0) sweeps are
                  sequential, so this is not an "all cores" figure and is not called one.

The timed region runs the product configuration: no per-launch event timing, the PCG loops replayed as hipGraphs.
The SpMV launch durations for `roofline` come from a SECOND, untimed pass over the same number of substeps with
flipv_params.kernel_timing = 1 (HIP events on the library's stream around every 8th launch; that mode launches kernel by
kernel), plus a back-to-back launch figure of the same kernel on the last system (flipv_bench_spmv).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MESH = os.path.join(ROOT, "tests", "golden", "meshes")

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
VISC_SPMV_BYTES_PER_INDEX = 52   # 3 diag + 4 factor + 3 x reads, 3 y writes, fp32
PRES_SPMV_BYTES_PER_CELL = 24    # 4 coefficient + 1 s reads, 1 z write, fp32


def box_mesh(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y0, z1], [x0, y0, z1], [x0, y1, z0], [x1, y1, z0], [x1, y1, z1], [x0, y1, z1]], np.float32)
    t = np.array([[0, 1, 2], [0, 2, 3], [4, 7, 6], [4, 6, 5], [0, 3, 7], [0, 7, 4], [1, 5, 6], [1, 6, 2], [0, 4, 5], [0, 5, 1], [3, 2, 6], [3, 6, 7]], np.int32)
    return v, t


WORKLOADS = {
    # name: (grid as multiples of --size, boundary mesh (None = the default box), liquid meshes, description)
    "bunny": ((1, 1, 1), ("sphere_large.ply", True), ["stanford_bunny.ply"],
              "%d^3 bunny drop: stanford_bunny.ply liquid in inverted sphere_large.ply, viscosity %g, full variational viscosity + "
              "pressure substep (BASELINE.json configs[2])"),
    "honey": ((1, 1, 1), None, ["rod.ply", "sheet.ply"],
              "%d^3 honey buckling: rod.ply + sheet.ply liquid (two addLiquid calls) in the default box, viscosity %g (BASELINE.json configs[3])"),
    "sheet": ((1, 0.5, 0.5), None, [("box", (0.05, 0.30, 0.05), (0.95, 0.3323, 0.45))],
              "%dx%dx%d thin sheet: liquid box x 0.05..0.95, y 0.30..0.3323, z 0.05..0.45 in a 1 x 0.5 x 0.5 domain, viscosity %g "
              "(BASELINE.json configs[4]; SURVEY.md 8d-5)"),
}


def build_workload(name, N, on_device, device=None, keep_setup_context=False):
    """Scene setup (not timed) -> (I, J, K, dx, solid SDF nodes, particles).  Host C++ path (bit-identical level sets) or, with
    on_device, the library's HIP setup kernels on a single-domain context (seconds instead of minutes at 512^3)."""
    from flipviscosity3d_amd import hostapi as H
    mult, boundary, liquids, _ = WORKLOADS[name]
    I, J, K = [max(8, int(round(N * m))) for m in mult]
    dx = float(np.float32(1.0 / N))

    def mesh(m):
        return box_mesh(m[1], m[2]) if isinstance(m, tuple) else H.load_ply(os.path.join(MESH, m))
    if on_device:
        from flipviscosity3d_amd.capi import Context
        c = Context(I, J, K, dx, device=device, setup_only=True)    # 3 grids + the setup kernels' temporaries, not a full context
        c.reset_boundary()
        if boundary:
            c.add_boundary_mesh(H.load_ply(os.path.join(MESH, boundary[0])), inverted=boundary[1])
        for m in liquids:
            c.add_liquid_mesh(mesh(m), seed=0)
        if keep_setup_context:       # a rank of a block decomposition takes its box of the solid SDF straight from this context (read_region)
            return I, J, K, dx, c, c.particles
        solid, particles = c.grid("SOLID_PHI"), c.particles
        c.close()
        return I, J, K, dx, solid, particles
    sim = H.FluidSimulation()
    sim.initialize(I, J, K, dx)
    if boundary:
        sim.addBoundary(H.load_ply(os.path.join(MESH, boundary[0])), boundary[1])
    sim.setSeeding(H.FluidSimulation.SEED_COUNTER, 0)
    for m in liquids:
        sim.addLiquid(mesh(m))
    solid, particles = sim.solid_sdf(), sim.particles
    sim.close()
    return I, J, K, dx, solid, particles


def build_scene(N, viscosity, on_device=False):
    """the headline scene (workload "bunny") -> (dx, solid, particles); kept for the tools that import it"""
    I, J, K, dx, solid, particles = build_workload("bunny", N, on_device)
    return dx, solid, particles


def cpu_baseline(viscosity, budget_size):
    """Reference (or port) timed on the host cores of this box: same scene at a size that costs ~10-30 s."""
    from flipviscosity3d_amd import hostapi as H
    N = budget_size
    dx = float(np.float32(1.0 / N))
    sim = H.FluidSimulation()
    sim.initialize(N, N, N, dx)
    sim.addBoundary(H.load_ply(os.path.join(MESH, "sphere_large.ply")), True)
    sim.setSeeding(H.FluidSimulation.SEED_COUNTER, 0)
    sim.addLiquid(H.load_ply(os.path.join(MESH, "stanford_bunny.ply")))
    solid, particles = sim.solid_sdf(), sim.particles
    sim.close()
    nsub = 2 if N < 256 else 1   # (256^3: ~3 minutes per substep on one core)
    try:
        from oracle import refbind as R
        use_ref = R.available()
    except Exception:
        use_ref = False
    if use_ref:
        r = R.RefSim(N, N, N, dx)
        r.set_grid("SOLID_PHI", solid)
        r.set_viscosity(viscosity)
        r.particles = particles
        t0 = time.perf_counter()
        for _ in range(nsub):
            r.substep(0.01)
        sec = (time.perf_counter() - t0) / nsub
        st = r.solver_stats()
        r.close()
        kind = "reference"
        its = (st["visc_iters"], st["pres_iters"])
        resid = float(st["visc_err"])
    else:
        from oracle import oraclebind as O
        s = O.OracleSim(N, N, N, dx)
        s.set_solid(solid)
        s.set_viscosity(viscosity)
        s.particles = particles
        t0 = time.perf_counter()
        for _ in range(nsub):
            _, vi, pi = s.substep(0.01)
        sec = (time.perf_counter() - t0) / nsub
        s.close()
        kind = "port"
        its = (vi["iterations"], pi["iterations"])
        resid = float(vi["residual"])
    return {
        "value": (N ** 3) / 1e6 / sec, "unit": "MCells/s", "cores": 1, "kind": kind,
        "sample": "same scene (bunny in inverted sphere, viscosity %g) at %d^3 (%d particles), mean of %d substeps of "
                  "0.01 s from rest, single thread, %.2f s per substep, last viscosity/pressure iterations %d/%d "
                  "(the reference's own cap of 700 and tolerance 1e-6: at 256^3 its viscosity solve stops at the cap, unconverged)"
                  % (viscosity, N, len(particles), nsub, sec, its[0], its[1]),
        "seconds_per_substep": sec, "viscosity_iterations": int(its[0]), "pressure_iterations": int(its[1]), "viscosity_residual": resid,
        "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
    }


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def spawn_ranks(n, result_fd):
    """Run this script as `n` ranks under torch.distributed.run (children of this process, started before anything here has
    initialised the GPU), pass rank 0's JSON line -- the only thing the ranks write to their stdout -- through to the saved
    stdout descriptor, and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in p.stdout:
        txt = raw.decode(errors="replace")
        if txt.lstrip().startswith("{") and '"metric"' in txt:
            line = txt
        else:
            sys.stderr.write(txt)
    rc = p.wait()
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        rc = 1
    if line is not None and rc == 0:
        os.write(result_fd, line.encode())
    return rc


def cpu_baseline_omp(viscosity, budget_size, threads):
    """The multi-threaded figure SURVEY.md 8d asks for: the C restatement of the reference's algorithm built with OpenMP (oracle/omp_baseline.py,
    kind "port"), run as a child process on the same bounded sample with 8, 16, 32 and 64 threads (those the host has; `threads` > 0 adds one
    count), OMP_PLACES=cores: the best is reported, all are listed.  What the reference's algorithm makes sequential (MIC(0)'s triangular
    solves) stays sequential, so this is Amdahl-bound by construction -- not an all-cores figure."""
    import subprocess
    best, tried = None, {}
    ncpu = os.cpu_count() or 1
    counts = sorted({t for t in (8, 16, 32, 64, threads) if 0 < t <= max(ncpu, 8)})
    # (more threads are not faster here: the sequential sweeps pull every vector back into one core's cache each iteration; measured 64^3,
    # 8-CPU container: 0.80 s on 1 thread, 0.58 on 4, 1.4-1.8 on 8; round 3, the GPU box's host: 0.60 MCells/s on 8 threads, 0.26 on 32)
    for thr in counts:
        env = dict(os.environ)
        env["OMP_NUM_THREADS"] = str(thr)
        env["OMP_PLACES"] = "cores"
        env["OMP_PROC_BIND"] = "close"
        env["OMP_WAIT_POLICY"] = "active"
        try:
            p = subprocess.run([sys.executable, "-m", "oracle.omp_baseline", str(budget_size), "2" if budget_size < 256 else "1", repr(viscosity)], cwd=ROOT, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
            line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                tried[str(thr)] = {"error": p.stderr.decode()[-400:]}
                continue
            out = json.loads(line[-1])
            tried[str(thr)] = round(out["value"], 4)
            if best is None or out["value"] > best["value"]:
                best = out
        except Exception as e:   # (the baseline must never take the bench line down)
            tried[str(thr)] = {"error": repr(e)}
    if best is None:
        return {"error": tried}
    best["sample"] = "same scene at %d^3, mean of %d substeps of 0.01 s from rest, OpenMP build of the C restatement on %d threads, %.2f s per substep" % (
        best["size"], best["substeps"], best["cores"], best["seconds_per_substep"])
    best["MCells_per_s_by_threads"] = tried
    best["host_cpus"] = os.cpu_count()
    return best


def main():
    # Exactly one line may reach stdout: the JSON result.  RCCL prints a version banner on stdout when a communicator
    # is created (the library's own and torch.distributed's), HIP tools may print too: everything written to fd 1 during
    # the run goes to stderr, the result is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--viscosity", type=float, default=5.0)
    ap.add_argument("--cpu-size", type=int, default=256, help="grid size of the CPU baseline (256 = the metric's own size, ONE substep of the reference: ~3 minutes "
                    "of one core; 128: two substeps, ~12 s -- always reported as cpu_baseline_128)")
    ap.add_argument("--no-strict", action="store_true", help="skip mode_b_strict (the timed window re-run with viscosity_stage1_factor = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="a further thread count for cpu_baseline_omp (OpenMP build of the C restatement; 8 / 16 / 32 / 64 are always tried), "
                    "1 = skip that baseline")
    ap.add_argument("--no-dense", action="store_true", help="skip the filled-box SpMV roofline measurement")
    ap.add_argument("--gpu-setup", action="store_true", help="build the scene with the device setup kernels (for sizes where the host path takes minutes)")
    ap.add_argument("--force-comm", action="store_true",
                    help="N=1 only: attach a one-rank RCCL communicator, so that the multi-rank code path (split SpMV launches, "
                         "halo stream, per-iteration all-reduce) is what gets timed: a lower bound of its overhead")
    ap.add_argument("--viscosity-cap", type=int, default=700,
                    help="iteration cap of the viscosity PCG: 700 = the reference's (equal-work timing, SURVEY 8d mode A); "
                         "a large value runs the solve to its 1e-6 tolerance (equal-accuracy, mode B)")
    ap.add_argument("--viscosity-preconditioner", choices=["auto", "diagonal", "multigrid"], default="auto",
                    help="auto = the library default (flipv_params.viscosity_preconditioner = AUTO: per solve the diagonal or the Galerkin "
                         "multigrid V-cycle, whichever the previous solve's iteration count predicts to be cheaper; what the headline is "
                         "timed with); diagonal / multigrid pin one (multigrid: one GPU, fp32)")
    ap.add_argument("--precision", type=int, default=0, help="0 fp32 vectors (default), 1 fp64 vectors")
    ap.add_argument("--exact-operator", action="store_true", help="flipv_params.exact_viscosity_operator = 1: the exact viscosity operator instead of the "
                    "reference's float-rounded one (no defect-correction stage; 1.5e-4 from the reference's converged velocities at 256^3)")
    ap.add_argument("--dense-size", type=str, default="384,512", help="further sizes of the filled-box SpMV roofline, comma-separated (the first is min(--size, 256)); 0 = none.  "
                    "384^3 does not fit the 256-wide tiles (a quarter of the lanes idle), 512^3 does")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="bunny",
                    help="bunny = BASELINE configs[2] (the metric's scene); honey = configs[3] (rod + sheet, use --viscosity 50); "
                         "sheet = configs[4] (--size is the long axis: 1024 -> 1024x512x512)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = ONE scene of --size split over the ranks (slabs along k for 2 and 4 ranks, 2x2x2 blocks for 8; "
                         "slabs along i for the sheet); weak = N copies of the scene stacked along k, one slab per rank")
    ap.add_argument("--dims", type=str, default="", help="process grid 'px,py,pz' of the strong-scaling decomposition (default: see --scaling)")
    ap.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                    help="transport of a multi-rank run.  rccl (default): one rank per GPU over RCCL / xGMI.  host: the host-staged communicator over torch.distributed gloo -- "
                         "the ranks share the visible devices round-robin (on a one-GPU box: all on device 0): a REHEARSAL of the multi-process path, its throughput means nothing")
    ap.add_argument("--spawn-selftest", action="store_true",
                    help="plumbing check that needs no GPU (tests/test_dist_gloo.py): the ranks only rendezvous over gloo and rank 0 prints a result line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.spawn_selftest and "WORLD_SIZE" in os.environ:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        print("rank %d of %d up" % (rank, world))   # goes to stderr (fd 1 is redirected above): must not reach the relayed line
        if rank == 0:
            os.write(result_fd, (json.dumps({"metric": "spawn selftest", "n_gpus": world, "sum": float(t.item())}) + "\n").encode())
        dist.destroy_process_group()
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (one per GPU, torch.distributed.run on
        # 127.0.0.1) and relay rank 0's JSON line.  Nothing in this process has touched the GPU yet and nothing will: it only waits.
        sys.exit(spawn_ranks(args.gpus, result_fd))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.comm == "host":
            local_rank = local_rank % max(1, torch.cuda.device_count())      # several ranks per device
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)

    from flipviscosity3d_amd import capi, partition
    from flipviscosity3d_amd.capi import Context

    N = args.size
    box_handover = world > 1 and args.scaling == "strong" and args.gpu_setup   # no full-size host array of the solid SDF on any rank
    GI, GJ, GK, dx, solid, particles = build_workload(args.workload, N, on_device=args.gpu_setup, device=local_rank, keep_setup_context=box_handover)
    decomposition = "single GPU"
    if world == 1:
        c = Context(GI, GJ, GK, dx, device=local_rank, slab=(0, GK) if args.force_comm else None)
        if args.force_comm:
            c.comm_init_rccl(capi.comm_unique_id(), 0, 1)
            decomposition += ", one-rank RCCL communicator attached"
        c.set_solid_sdf(solid)
    elif args.scaling == "strong":
        # strong scaling: ONE scene, decomposed into blocks; every rank builds the (deterministic) scene and keeps its box
        if args.dims:
            dims = tuple(int(v) for v in args.dims.split(","))
        elif args.workload == "sheet":
            dims = (world, 1, 1)                                   # slabs along the long axis
        else:
            dims = {2: (1, 1, 2), 4: (1, 1, 4), 8: (2, 2, 2)}.get(world, (1, 1, world))
        assert dims[0] * dims[1] * dims[2] == world, "process grid does not match the number of ranks"
        boxes = partition.block_boxes(GI, GJ, GK, dims)
        c = Context(GI, GJ, GK, dx, device=local_rank, block=boxes[rank])
        if args.comm == "host":
            c.comm_init_host(capi.torch_distributed_callbacks(dist), rank, dims)
        else:
            uid = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            c.comm_init_rccl(uid[0], rank, world, dims)
        if box_handover:                                           # `solid` is the setup context: the rank's allocated box of its solid SDF, box-shaped
            lo, hi = c.grid_box("SOLID_PHI", 1)
            c.write_box("SOLID_PHI", solid.read_region("SOLID_PHI", lo, hi))
            solid.close()
        else:
            c.set_solid_sdf(solid)                                 # the library takes the entries of its box (owned + halo)
        particles = partition.split_particles_boxes(particles, dx, boxes, dims)[rank]
        transport = "HOST-STAGED (torch.distributed gloo; a rehearsal of the multi-process path, the ranks share devices)" if args.comm == "host" else "RCCL"
        decomposition = "%dx%dx%d blocks of one %dx%dx%d scene (rank-local allocation), %s halo exchange with the <= 26 neighbours + PCG scalar all-reduce + particle migration" % (
            dims[0], dims[1], dims[2], GI, GJ, GK, transport)
    else:
        # weak scaling: `world` copies of the closed 256^3 scene stacked along k form ONE domain of N x N x (N*world)
        # cells, decomposed into one slab per rank.  The copies do not interact physically, but they are solved as one
        # system: every PCG iteration exchanges the halo planes of the search direction and all-reduces its scalars,
        # every extrapolation layer / P2G / SDF exchanges halos, particles migrate between slabs.
        assert args.workload == "bunny", "the stacked weak-scaling scene is the closed bunny container"
        solid_g, parts = partition.stack_scene(solid, particles, world, N, dx)
        ranges = partition.slab_ranges(N * world, world)
        GK = N * world
        c = Context(N, N, N * world, dx, device=local_rank, slab=ranges[rank])
        decomposition = "%d slabs along k of a %dx%dx%d domain (%d stacked copies of the scene), RCCL halo exchange + PCG scalar " \
                        "all-reduce + particle migration" % (world, N, N, N * world, world)
        if args.comm == "host":
            c.comm_init_host(capi.torch_distributed_callbacks(dist), rank, (1, 1, world))
        else:
            uid = [capi.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            c.comm_init_rccl(uid[0], rank, world)
        c.set_solid_sdf(solid_g)
        particles = parts[rank]
        del solid_g, parts
    c.set_viscosity(args.viscosity)
    c.set_gravity(0.0, -9.81, 0.0)
    c.set_params(precision=args.precision, kernel_timing=0, viscosity_max_iterations=args.viscosity_cap, exact_viscosity_operator=1 if args.exact_operator else 0)
    if args.viscosity_preconditioner != "auto":
        c.set_params(viscosity_preconditioner=capi.PRECOND_MULTIGRID if args.viscosity_preconditioner == "multigrid" else capi.PRECOND_DIAGONAL)
    c.particles = particles
    dev_name = c.device_name()

    def step():
        dt = min(c.cfl(), 0.01)
        return c.substep(dt)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        c.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.comm == "host" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- second pass, NOT timed: a few more substeps with HIP events around every 8th launch of the PCG's own SpMV (kernel_timing
    # launches the loop kernel by kernel, so it reads a little above rocprofv3's kernel-only average)
    c.set_params(kernel_timing=1)
    c.kernel_stats_reset()
    for _ in range(max(1, min(args.steps, 5))):
        step()
    c.synchronize()
    ks = c.kernel_stats()
    c.set_params(kernel_timing=0)
    b2b = {}
    if world == 1:
        for which, name in ((1, "viscosity"), (0, "pressure")):
            try:
                ms, swept = c.bench_spmv(which, 200)
                b2b[name] = ms
            except Exception:   # no such system (viscosity off)
                pass

    # ---- mode_b_strict: the same window (a fresh context, the same warm-up and step counts) with stage 1 of every viscosity solve taken to the
    # reference's own tolerance, viscosity_stage1_factor = 1 (what a user sets who does not want the early stop); not part of `value`
    strict = None
    if world == 1 and not args.no_strict and not args.exact_operator and args.precision == 0 and args.viscosity > 0:
      try:   # (a second full context beside the first: an out-of-memory or solver error of this untimed re-run must not discard the measured headline)
          c2 = Context(GI, GJ, GK, dx, device=local_rank)
          c2.set_solid_sdf(solid)
          c2.set_viscosity(args.viscosity)
          c2.set_gravity(0.0, -9.81, 0.0)
          c2.set_params(viscosity_max_iterations=args.viscosity_cap, viscosity_stage1_factor=1.0)
          if args.viscosity_preconditioner != "auto":
              c2.set_params(viscosity_preconditioner=capi.PRECOND_MULTIGRID if args.viscosity_preconditioner == "multigrid" else capi.PRECOND_DIAGONAL)
          c2.particles = particles
          for _ in range(args.warmup):
              c2.substep(min(c2.cfl(), 0.01))
          c2.synchronize()
          ts = time.perf_counter()
          st2 = [c2.substep(min(c2.cfl(), 0.01)) for _ in range(args.steps)]
          c2.synchronize()
          el2 = time.perf_counter() - ts
          c2.close()
          strict = {"value": float(GI) * GJ * GK / 1e6 / (el2 / args.steps), "unit": "MCells/s", "ms_per_step": el2 * 1e3 / args.steps,
                    "viscosity_stage1_factor": 1.0,
                    "all_timed_solves_stage_complete": all(st["viscosity"]["status"] in (0, 3) for st in st2),
                    "mean_viscosity_iterations": float(np.mean([st["viscosity"]["iterations"] for st in st2])),
                    "loop_residual_rel_max": float(max((st["viscosity"]["residual"] / st["viscosity"]["rhs_norm"]) if st["viscosity"]["rhs_norm"] > 0 else 0.0 for st in st2)),
                    "note": "stage 1 of the two-stage solve to viscosity_tolerance (1e-6 max|rhs|) itself; same scene, warm-up and step counts as `value`"}
      except Exception as e:   # noqa: BLE001
        strict = {"error": "%s: %s" % (type(e).__name__, e)}

    def its(key):
        v = [st[key]["iterations"] for st in stats]
        return {"mean": float(np.mean(v)), "min": int(min(v)), "max": int(max(v)), "per_step": v if len(v) <= 32 else None}

    ms_per_step = elapsed * 1e3 / args.steps
    cells_total = float(GI) * GJ * GK   # the whole domain (weak mode: the stacked one)
    value = cells_total / 1e6 / (elapsed / args.steps)

    # ---- roofline of the dominant kernel.  Units processed per launch: unknowns of the system in cells' worth
    # (viscosity: rows / 3; pressure: pressure cells); the swept index positions of the tile lists are reported beside it.
    v_ms, v_n, v_cells = ks["viscosity_spmv_ms"], ks["viscosity_spmv_launches"], ks["viscosity_spmv_cells"]
    p_ms, p_n, p_cells = ks["pressure_spmv_ms"], ks["pressure_spmv_launches"], ks["pressure_spmv_cells"]
    last = stats[-1]
    roof = None
    if v_n > 0 and v_ms >= p_ms:
        avg_ms = v_ms / v_n
        units = last["viscosity"]["rows"] / 3.0   # this rank's rows (rank 0)
        gbs = VISC_SPMV_BYTES_PER_INDEX * units / (avg_ms * 1e-3) / 1e9
        kname = ("k_bvisc_spmv" if last["viscosity"]["layout"] == 2 else "k_visc_spmv") + ("<float>" if args.precision == 0 else "<double>")
        # the instantiation the PCG loop itself launches (the multigrid's fine-level sweeps are other instantiations of the same kernel)
        kvariant = None
        if last["viscosity"]["layout"] == 2:
            # (the multigrid-preconditioned loop launches EPI_SPMV_A = 4: q = A p and p.q alone; the diagonal loop EPI_SPMV = 0 with the fused (r/d, q) and (q, q/d))
            kvariant = "k_bvisc_spmv<%s, %s>" % ("float" if args.precision == 0 else "double", "false, 4" if last["viscosity"]["preconditioner"] == 1 else "true, 0")
        roof = {"kernel": kname, "bound": "hbm",
                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                "avg_launch_us": avg_ms * 1e3, "launches": v_n, "units_per_launch": units, "bytes_per_unit": VISC_SPMV_BYTES_PER_INDEX,
                "timing": "HIP events on the library's stream around every 8th launch of the PCG's own SpMV, untimed second pass",
                "layout": {0: "plain planes", 1: "plain planes, swizzled own-index arrays", 2: "bricks of 8 x 4 x 2 indices"}[last["viscosity"]["layout"]],
                "back_to_back_launch_us": b2b.get("viscosity", 0.0) * 1e3 or None,
                "unit_definition": "one cell's worth of unknowns = 3 rows of the viscosity system",
                "swept_indices_per_launch": v_cells / v_n}
    elif p_n > 0:
        avg_ms = p_ms / p_n
        units = float(last["pressure"]["rows"])
        gbs = PRES_SPMV_BYTES_PER_CELL * units / (avg_ms * 1e-3) / 1e9
        roof = {"kernel": "k_pressure_spmv<float>", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": avg_ms * 1e3, "launches": p_n,
                "units_per_launch": units, "bytes_per_unit": PRES_SPMV_BYTES_PER_CELL, "unit_definition": "pressure cell",
                "swept_indices_per_launch": p_cells / p_n}
    if roof is not None and roof.get("back_to_back_launch_us"):
        # `frac` is over the KERNEL's duration: the back-to-back figure, which is the one rocprofv3's kernel-only average agrees with (rocprof_avg_launch_us below).  The
        # bracket around single launches inside the solve also holds the dispatch gap of a dependent chain and the two event records; it is kept beside it.
        roof["in_solve_event_us"] = roof["avg_launch_us"]
        roof["frac_in_solve_events"] = roof["frac"]
        roof["avg_launch_us"] = roof["back_to_back_launch_us"]
        roof["achieved"] = roof["bytes_per_unit"] * roof["units_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["timing"] = ("HIP events on the library's stream around 200 back-to-back launches of the PCG's own SpMV on the system the last timed substep left (kernel duration; "
                          "rocprofv3's kernel-only average of the committed profile beside it); in_solve_event_us: events around every 8th launch inside the solves of an untimed second pass")
    if roof is not None:
        roof.update(committed_profile(kvariant if (v_n > 0 and v_ms >= p_ms and kvariant) else roof["kernel"].split("<")[0], N, args, world))
        if v_n > 0 and v_ms >= p_ms and kvariant:
            roof["kernel_instantiation"] = kvariant
        if roof.get("rocprof_avg_launch_us"):   # the same algorithmic bytes over rocprofv3's kernel-only duration (the event bracket above includes the dispatch gap)
            roof["frac_at_rocprof_duration"] = roof["bytes_per_unit"] * roof["units_per_launch"] / (roof["rocprof_avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
    extra = {}
    if p_n > 0:
        avg = p_ms / p_n
        extra["pressure_spmv"] = {"avg_launch_us": avg * 1e3, "launches": p_n, "cells_per_launch": last["pressure"]["rows"],
                                  "achieved_GBs": PRES_SPMV_BYTES_PER_CELL * last["pressure"]["rows"] / (avg * 1e-3) / 1e9}

    if rank == 0:
        out = {
            "metric": "MCells/s per substep (P2G+PCG+viscosity), %s grid" % ("%d^3" % N if GI == GJ == GK or args.scaling == "weak" else "%dx%dx%d" % (GI, GJ, GK)),
            "value": value, "unit": "MCells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
            "dtype": "f32" if args.precision == 0 else "f64", "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.workload][3] % ((N, args.viscosity) if args.workload != "sheet" else (GI, GJ, GK, args.viscosity)),
                "grid": [GI, GJ, GK], "particles_rank0": int(len(particles)), "dt": 0.01,
                "viscosity_cap": args.viscosity_cap, "viscosity_preconditioner": args.viscosity_preconditioner, "parallelism": decomposition,
            },
            "device": dev_name,
            "phase_ms": {k: float(np.mean([st["phase_ms"][k] for st in stats])) for k in last["phase_ms"]},
            "phase_ms_note": "mean over the timed substeps (GPU time per phase, HIP events)",
            "viscosity_iterations": its("viscosity"), "pressure_iterations": its("pressure"),
            "viscosity_preconditioner_per_step": [st["viscosity"]["preconditioner"] for st in stats],   # 0 diagonal, 1 multigrid (AUTO picks per solve)
            "viscosity_status_per_step": [st["viscosity"]["status"] for st in stats],                   # 0 converged = every stage of the solve reached its target, 1 cap reached / stalled / a correction stage ended short (iterate accepted), 3 trivial
            "viscosity_correction_iterations_per_step": [st["viscosity"].get("correction_iterations", 0) for st in stats],   # of `iterations`: spent in defect-correction stages
            "viscosity_correction_status_per_step": [st["viscosity"].get("correction_status", 0) for st in stats],           # flipv_solve_info.correction_status
            "viscosity_operator": "exact (vol u - div tau)" if args.exact_operator else
                                  "the reference's (float-rounded diagonal, viscositysolver.cpp:394-446): exact-operator multigrid-PCG + fp64 defect-correction stage(s)",
            # what every timed solve delivered, relative to max|rhs| (reference: 1e-6 on its operator): the Krylov loop's own residual on the exact
            # operator when it stopped (stage 1: viscosity_stage1_factor x 1e-6 = 3e-3 on this scene where a defect-correction stage follows, 1e-6 otherwise), and
            # max|b - A_ref x| recomputed in fp64 at the very end (0: no defect-correction stage ran -- diagonal preconditioner, exact operator or
            # trivial solve).  The latter is a max-norm that a few sliver rows dominate; what the solve delivers in the VELOCITIES is pinned by the
            # parity tests (<= 1e-4 of the reference run to convergence: 256^3 here, and nu dt/dx^2 up to 1.3e5)
            "viscosity_final_residual_rel": {
                "loop_on_exact_operator_max": float(max((st["viscosity"]["residual"] / st["viscosity"]["rhs_norm"]) if st["viscosity"]["rhs_norm"] > 0 else 0.0 for st in stats)),
                "recomputed_on_reference_operator_max": float(max((st["viscosity"].get("defect_residual", 0.0) / st["viscosity"]["rhs_norm"]) if st["viscosity"]["rhs_norm"] > 0 else 0.0 for st in stats)),
                "recomputed_on_reference_operator_per_step": [float((st["viscosity"].get("defect_residual", 0.0) / st["viscosity"]["rhs_norm"]) if st["viscosity"]["rhs_norm"] > 0 else 0.0) for st in stats]},
            # mode B of SURVEY.md 8d (equal accuracy): every viscosity solve run to convergence instead of to the cap.  A default run IS that when every
            # timed solve converged (status 0; what a converged default solve delivers against the reference's operator: viscosity_final_residual_rel, and
            # tests/test_gpu_baseline_sizes.py pins its velocities to 1e-4 of the reference run to convergence at 128^3 and 256^3); otherwise this object
            # says how many did not
            "mode_b": {"value": value if all(st["viscosity"]["status"] in (0, 3) for st in stats) else None,
                       "unit": "MCells/s", "all_timed_solves_stage_complete": all(st["viscosity"]["status"] in (0, 3) for st in stats),
                       "incomplete_solves": int(sum(st["viscosity"]["status"] not in (0, 3) for st in stats)),
                       "timed_solves_within_reference_residual": int(sum((st["viscosity"].get("defect_residual", 0.0) or st["viscosity"]["residual"]) <= 1.0000001e-6 * st["viscosity"]["rhs_norm"] for st in stats)),
                       "mean_viscosity_iterations": float(np.mean([st["viscosity"]["iterations"] for st in stats])),
                       "note": "same run as `value`.  'stage complete' = status 0 = every stage of the default viscosity solve (include/flipv.h: THE DEFAULT VISCOSITY SOLVE) reached its "
                               "target inside the reference's cap of 700 -- NOT 'max|b - A_ref x| <= 1e-6 max|rhs|', the reference's own stop test: the stages' targets are shares of "
                               "min(max|rhs|, 100 max|u|) and the delivering loop carries a velocity criterion, which is tighter than the reference's test once the liquid touches a wall and "
                               "looser on the sliver rows of a stiff start (timed_solves_within_reference_residual counts the solves whose fp64 residual on the reference's operator passes "
                               "that test too; viscosity_final_residual_rel has the values; mode_b_strict runs stage 1 to 1e-6 max|rhs|).  What the default delivers in the velocities is "
                               "pinned by the parity tests: <= 1e-4 of the reference run to convergence from rest AND in late states (tests/test_gpu_late_states.py)"},
            "mode_b_strict": strict,
            "viscosity": {k: last["viscosity"][k] for k in ("iterations", "residual", "rhs_norm", "status", "rows", "active_tiles", "total_tiles")},
            "pressure": {k: last["pressure"][k] for k in ("iterations", "residual", "rhs_norm", "status", "rows", "active_tiles", "total_tiles")},
            "roofline": roof,
        }
        out.update(extra)
        if world == 1 and not args.no_dense:
            c.close()
            c = None
            out["roofline_dense"] = dense_roofline(min(N, 256), args.precision)
            for ds in [int(v) for v in str(args.dense_size).split(",") if v.strip()]:
                if ds > 0 and ds != min(N, 256):
                    # (at 256^3 an fp32 array is 64 MiB against a 256 MiB Infinity Cache; 384^3 arrays are 216 MiB each, 512^3 arrays 512 MiB)
                    out["roofline_dense_%d" % ds] = dense_roofline(ds, args.precision, reps=20 if ds < 512 else 10)
        if world == 1 and not args.no_cpu_baseline and args.workload == "bunny":
            out["cpu_baseline"] = cpu_baseline(args.viscosity, args.cpu_size)
            if args.cpu_size != 128:
                out["cpu_baseline_128"] = cpu_baseline(args.viscosity, 128)
            if args.cpu_threads != 1:
                out["cpu_baseline_omp"] = cpu_baseline_omp(args.viscosity, 128, args.cpu_threads)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
    if (dist is not None or args.force_comm) and c is not None:
        c.comm_finalize()
    if c is not None:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


def dense_roofline(N, precision, reps=50, **extra):
    """The two SpMV kernels on a completely filled N^3 box (every interior cell liquid): active ~ swept cells, a launch
    streams hundreds of MB, so achieved GB/s against the HBM peak measures the kernels themselves.  flipv_bench_spmv
    launches the solver's own kernel `reps` times back to back between two HIP events on the library's stream."""
    from flipviscosity3d_amd import hostapi as H
    from flipviscosity3d_amd.capi import Context
    dx = float(np.float32(1.0 / N))
    sim = H.FluidSimulation()
    sim.initialize(N, N, N, dx)
    solid = sim.solid_sdf()   # the default box boundary (fluidsimulation.cpp:206-239)
    sim.close()
    c = Context(N, N, N, dx)
    c.set_solid_sdf(solid)
    c.set_viscosity(5.0)
    c.set_params(precision=precision, pressure_max_iterations=4, viscosity_max_iterations=4, check_every=4, **extra)
    rng = np.random.default_rng(0)
    c.set_grid("LIQUID_PHI", np.full((N, N, N), -0.5 * dx, np.float32))
    for n, shp in (("U", (N, N, N + 1)), ("V", (N, N + 1, N)), ("W", (N + 1, N, N))):
        c.set_grid(n, rng.uniform(-1, 1, shp).astype(np.float32))
    c.compute_weights()
    vi = c.viscosity_solve(0.01)
    pi = c.pressure_solve(0.01)
    out = {"workload": "filled %d^3 box, every interior cell liquid" % N, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "device_copy_GBs": c.bench_copy(1 << 30, 10),
           "attainable_GBs": {"read": c.bench_stream(0), "copy": c.bench_stream(1), "write": c.bench_stream(2),
                              "read_tuned": c.bench_stream(3), "copy_tuned": c.bench_stream(4), "mix_5to1": c.bench_stream(5), "mix_10to3": c.bench_stream(6, nbytes=1 << 29),
                              "note": "stencil-free kernels over 1 GiB; *_tuned, mix_5to1 (the pressure SpMV's own five reads : one write) and mix_10to3 (the viscosity SpMV's) move 16 B per lane with "
                                      "nontemporal loads/stores from a grid sized to the CUs (flipv_bench_stream modes 3-5)"}}
    for which, name, b, units in ((0, "pressure_spmv", PRES_SPMV_BYTES_PER_CELL, float(pi["rows"])),
                                  (1, "viscosity_spmv", VISC_SPMV_BYTES_PER_INDEX, vi["rows"] / 3.0)):
        ms, swept = c.bench_spmv(which, reps)
        gbs = b * units / (ms * 1e-3) / 1e9
        out[name] = {"avg_launch_us": ms * 1e3, "units_per_launch": units, "bytes_per_unit": b, "achieved": gbs,
                     "frac": gbs / HBM_PEAK_GBS, "swept_indices_per_launch": swept}
    out["viscosity_spmv"]["variant"] = "the diagonal-preconditioned loop's launch: reads the residual for the fused (r/d, q) as well (64 B per unit move, 52 are counted)"
    # the variant the multigrid-preconditioned loop launches (what AUTO runs on a stiff system): q = A p and p.q only, exactly the 52 B per unit
    ms, swept = c.bench_spmv(2, reps)
    gbs = VISC_SPMV_BYTES_PER_INDEX * (vi["rows"] / 3.0) / (ms * 1e-3) / 1e9
    out["viscosity_spmv_multigrid_loop"] = {"avg_launch_us": ms * 1e3, "units_per_launch": vi["rows"] / 3.0, "bytes_per_unit": VISC_SPMV_BYTES_PER_INDEX, "achieved": gbs,
                                            "frac": gbs / HBM_PEAK_GBS, "swept_indices_per_launch": swept,
                                            "variant": "flipv_bench_spmv(which = 2): the kernel the multigrid-preconditioned loop launches -- no residual read, p.q alone (EPI_SPMV_A)"}
    c.close()
    return out


def committed_profile(kernel, N, args, world):
    """HBM traffic per launch of `kernel` from the committed PMC summary of this very workload (FETCH_SIZE x2 + WRITE_SIZE,
    separate rocprofv3 --pmc passes, tools/profile_round.sh), and rocprofv3's kernel-only average duration next to the
    HIP-event figure measured in this run.  Counters cannot be collected from inside the process, so the numbers
    are the newest profiles/r*/bench<N>_* files; null when this run is a different workload."""
    import csv
    import glob
    import re
    out = {"traffic": None}
    if world != 1 or args.precision != 0 or abs(args.viscosity - 5.0) > 1e-12:
        return out
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    key = lambda p: [int(x) for x in re.findall(r"\d+", os.path.relpath(p, root))]
    pmc = sorted(glob.glob(os.path.join(root, "r*", "bench%d_*_pmc_traffic.json" % N)), key=key)
    if pmc:
        d = json.load(open(pmc[-1]))
        for name, v in d.items():
            name = re.sub(r"^g\d+::", "", name)   # tile-geometry namespace (csrc/pcg_geo.inc)
            if name.startswith(kernel if "<" in kernel else kernel + "<") or name == kernel:
                out["traffic"] = v["hbm_bytes_per_launch"]
                out["traffic_unit"] = "bytes per launch (HBM read + write, PMC)"
                out["traffic_source"] = os.path.relpath(pmc[-1], os.path.dirname(root))
                break
    st = sorted(glob.glob(os.path.join(root, "r*", "bench%d_*_kernel_stats.csv" % N)), key=key)
    if st:
        for r in csv.DictReader(open(st[-1])):
            nm = re.sub(r"^(void )?(g\d+::)?", "", r["Name"])
            if nm.startswith(kernel if "<" in kernel else kernel + "<") or nm.startswith(kernel + "("):
                out["rocprof_avg_launch_us"] = float(r["AverageNs"]) / 1e3
                out["rocprof_source"] = os.path.relpath(st[-1], os.path.dirname(root))
                break
    return out


if __name__ == "__main__":
    main()
