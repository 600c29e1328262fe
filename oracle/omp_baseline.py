"""All-cores CPU baseline (TEST INFRASTRUCTURE: bench.py's cpu_baseline leg runs this as a child process).

    python -m oracle.omp_baseline SIZE SUBSTEPS [VISCOSITY]   ->  one JSON line on stdout

The C restatement of the reference's algorithm (oracle/flip_oracle.c) built with -fopenmp (make -C oracle oracle_omp): its
data-parallel loops -- SpMV, dot products, axpys, max norms, the particle kernels -- run on every host core, the parts the reference's
algorithm makes sequential (MIC(0)'s triangular solves and factorisation, scatters in particle order, layered extrapolation) do not.
Scene: bench.py's (bunny in the inverted sphere), built by the host library; OMP_NUM_THREADS picks the thread count."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    N, nsub = int(sys.argv[1]), int(sys.argv[2])
    nu = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-s", "-C", here, "oracle_omp"])
    from oracle import oraclebind as O
    O.LIB_PATH = os.path.join(here, "libfliporacle_omp.so")   # before the first call loads the library
    O.build = lambda force=False: O.LIB_PATH
    import numpy as np
    from bench import build_workload
    I, J, K, dx, solid, P = build_workload("bunny", N, on_device=False)
    s = O.OracleSim(I, J, K, dx)
    s.set_solid(solid); s.set_viscosity(nu)
    s.particles = P
    threads = O.lib().oracle_omp_threads()
    t0 = time.perf_counter()
    its = (0, 0)
    for _ in range(nsub):
        _, vi, pi = s.substep(0.01)
        its = (vi["iterations"], pi["iterations"])
    sec = (time.perf_counter() - t0) / nsub
    s.close()
    print(json.dumps({"value": N ** 3 / 1e6 / sec, "unit": "MCells/s", "cores": int(threads), "kind": "port",
                      "seconds_per_substep": sec, "size": N, "substeps": nsub, "particles": int(len(P)),
                      "last_viscosity_iterations": int(its[0]), "last_pressure_iterations": int(its[1])}))


if __name__ == "__main__":
    main()
