#!/bin/bash
# A/B of compile-time variants of libflipv.so on the 256^3 bench scene.
#   here:     bash tools/ab_lib.sh build name1 "-DFLIPV_X=1" name2 "-DFLIPV_X=2" ...   (variants under csrc/build/variants/)
#   GPU box:  bash tools/ab_lib.sh run name1 name2 ...                                  (phase times of the third substep)
set -e
cs=flipviscosity3d_amd/csrc
mode=$1; shift
if [ "$mode" = build ]; then
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    mkdir -p $cs/build/variants/$name
    for f in $cs/*.hip; do
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function $flags -c $f -o $cs/build/variants/$name/$(basename $f .hip).o &
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $cs/build/variants/$name.so $cs/build/variants/$name/*.o -ldl -lpthread
    rm -rf $cs/build/variants/$name
    echo built $name: $flags
  done
else
  for name in "$@"; do
    echo "== $name"
    FLIPV_LIB=$PWD/$cs/build/variants/$name.so timeout 300 python3 tools/ab_switch.py no_graph_replay 0 2>&1 | grep no_graph_replay
  done
fi
