#!/bin/bash
# Round-4 evidence set (run through gpurun; outputs under gpurun_out/ev_r4/, copied into profiles/r4/ afterwards).  bash tools/evidence_r4.sh <tag>
tag=${1:-r4v1}
out=gpurun_out/ev_r4
mkdir -p $out
B="python3 bench.py --gpu-setup"
Q="--no-cpu-baseline --no-dense"
bash tools/profile_round.sh $tag --gpu-setup --no-strict > $out/prof_$tag.log 2>&1
bash tools/profile_dense.sh ${tag}_256 256 0 > $out/prof_dense_${tag}_256.log 2>&1
$B --steps 20 --warmup 5 > $out/bench256_driver_like_20steps.json 2> $out/bench256_driver_like_20steps.err
$B --steps 20 --warmup 5 $Q > $out/bench256_default_20steps.json 2>/dev/null
$B --steps 20 --warmup 5 --exact-operator $Q > $out/bench256_exact_operator_20steps.json 2>/dev/null
$B --steps 20 --warmup 5 --viscosity-preconditioner diagonal $Q --no-strict > $out/bench256_diagonal_20steps.json 2>/dev/null
$B --steps 10 --warmup 3 $Q > $out/bench256_default_10steps.json 2>/dev/null
$B --steps 10 --warmup 3 --gpus 1 --force-comm $Q --no-strict > $out/bench256_forcecomm_10steps.json 2>/dev/null
$B --size 512 --steps 10 --warmup 3 $Q > $out/bench512.json 2>/dev/null
$B --workload honey --size 256 --viscosity 50 --steps 10 --warmup 3 $Q > $out/bench_honey256.json 2>/dev/null
$B --workload honey --size 512 --viscosity 50 --steps 5 --warmup 2 $Q > $out/bench_honey512.json 2>/dev/null
$B --workload sheet --size 1024 --steps 3 --warmup 1 $Q --no-strict > $out/bench_sheet1024.json 2>/dev/null
python3 tools/local_ranks_bench.py strong 2,2,2 256 > $out/local_ranks_222_256.log 2>&1
python3 tools/local_ranks_bench.py strong 2,2,2 512 honey 50 > $out/local_ranks_222_512.log 2>&1
(python3 tools/r4_predict_scan.py 0 1 -1; python3 tools/r4_predict_scan.py 0 1 0) > $out/predictor_on_off.log 2>&1
python3 tools/r4_dense_geo_scan.py 256,320,384,448,512 > $out/dense_geometry_scan.log 2>&1
(python3 tools/r3_status.py bunny 256 5 2000; python3 tools/r3_status.py honey 256 50 200; python3 tools/r3_status.py sheet 512 5 200; python3 tools/r3_status.py bunny 128 5 500; python3 tools/r3_status.py honey 512 50 30) > $out/soak_status.log 2>&1
for f in $out/*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("mode_b_strict") or {}
    print("%-48s %8.1f %s  %.2f ms/step  its %s  statuses %s  strict %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d.get("mode_b", {}).get("mean_viscosity_iterations"), sorted(set(d.get("viscosity_status_per_step", []))), s.get("value")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
