#!/bin/bash
# round-4 measurement batch 1 (run on the GPU box through gpurun): multi-rank tests, bench lines of the four workloads, in-process 2x2x2 blocks
O=gpurun_out/r4c; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_multirank_default.py tests/test_gpu_baseline_sizes.py tests/test_gpu_stiff_regime.py tests/test_gpu_fullsize.py -q --timeout 900 > $O/tests_mr.log 2>&1; echo rc=$? >> $O/tests_mr.log)
tail -4 $O/tests_mr.log
B="--gpu-setup --no-cpu-baseline --no-dense"
timeout 300 python bench.py --steps 20 --warmup 5 $B > $O/bench256_default_20.json 2> $O/bench256_default_20.err
timeout 300 python bench.py --steps 10 --warmup 3 $B > $O/bench256_default_10.json 2> /dev/null
timeout 300 python bench.py --steps 10 --warmup 3 $B --force-comm --no-strict > $O/bench256_forcecomm_10.json 2> /dev/null
timeout 300 python bench.py --workload honey --viscosity 50 --steps 10 --warmup 3 $B > $O/bench_honey256.json 2> /dev/null
timeout 600 python bench.py --workload honey --viscosity 50 --size 512 --steps 5 --warmup 2 $B > $O/bench_honey512.json 2> /dev/null
timeout 600 python bench.py --size 512 --steps 10 --warmup 3 $B > $O/bench512.json 2> /dev/null
timeout 900 python bench.py --workload sheet --size 1024 --steps 3 --warmup 1 $B --no-strict > $O/bench_sheet1024.json 2> /dev/null
timeout 600 python tools/local_ranks_bench.py strong 2,2,2 256 > $O/local_ranks_222_256.log 2>&1
tail -12 $O/local_ranks_222_256.log
