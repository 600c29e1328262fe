#!/bin/bash
# kernel statistics (rocprofv3 --kernel-trace --stats) of a default-parameter run of the bench scene: bash tools/prof_soak.sh <tag> <size> <substeps>
tag=${1:-s1}; N=${2:-256}; n=${3:-8}
export TMPDIR=/tmp
out=gpurun_out/prof_soak_$tag
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o $tag -- python3 tools/r3_soak.py $N $n 0 > $out/run.log 2> $out/stats.log
find $out/stats -name '*kernel_trace.csv' -delete
tail -3 $out/run.log
