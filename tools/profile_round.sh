#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate passes.
# usage (through gpurun): bash tools/profile_round.sh <tag> [extra bench.py arguments, e.g. --viscosity-cap 20000 --viscosity-preconditioner multigrid --gpu-setup]
tag=${1:-v3}
shift
extra="$@"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense $extra > $out/bench_under_rocprof.json 2> $out/stats.log
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense $extra > $out/pmc_f.json 2> $out/pmc_f.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense $extra > $out/pmc_w.json 2> $out/pmc_w.log
f=$(find $out/pmc_f -name '*counter_collection.csv' | head -1)
w=$(find $out/pmc_w -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $f $w $out/pmc_traffic.json
# the raw counter csvs are large; keep only the summary
rm -rf $out/pmc_f $out/pmc_w
find $out/stats -name '*kernel_trace.csv' -delete
ls -la $out $out/stats/*
