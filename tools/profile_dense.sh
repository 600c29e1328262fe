#!/bin/bash
# rocprofv3 evidence for the SpMV kernels on the filled box: kernel stats, then FETCH_SIZE and WRITE_SIZE in separate passes.
# usage (through gpurun): bash tools/profile_dense.sh <tag> <size> <runlen> [rowl]
tag=${1:-d1}; N=${2:-256}; rl=${3:-0}; rowl=${4:-}
export TMPDIR=/tmp
[ -n "$rowl" ] && rowl="--tile-rows $rowl"
out=gpurun_out/prof_dense_$tag
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 tools/ab_runlen.py dense $N $rowl -- $rl > $out/run_under_rocprof.log 2> $out/stats.log
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_f -o f -- python3 tools/ab_runlen.py dense $N $rowl -- $rl > $out/pmc_f.out 2> $out/pmc_f.log
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -o w -- python3 tools/ab_runlen.py dense $N $rowl -- $rl > $out/pmc_w.out 2> $out/pmc_w.log
f=$(find $out/pmc_f -name '*counter_collection.csv' | head -1)
w=$(find $out/pmc_w -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $f $w $out/pmc_traffic.json | grep -i spmv
rm -rf $out/pmc_f $out/pmc_w
find $out/stats -name '*kernel_trace.csv' -delete
cat $out/run_under_rocprof.log
grep -h spmv $(find $out/stats -name '*kernel_stats.csv') | head -6
