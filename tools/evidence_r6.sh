#!/bin/bash
# Round-6 evidence set (run through gpurun; outputs under gpurun_out/ev_r6/, copied into profiles/r6/ afterwards).  bash tools/evidence_r6.sh <tag> [part]
tag=${1:-r6v1}
part=${2:-all}
out=gpurun_out/ev_r6
mkdir -p $out
B="python3 bench.py --gpu-setup"
Q="--no-cpu-baseline --no-dense"
if [ $part = all ] || [ $part = prof ]; then
bash tools/profile_round.sh $tag --gpu-setup --no-strict > $out/prof_$tag.log 2>&1
for n in 256 384 512; do bash tools/profile_dense.sh ${tag}_$n $n 0 > $out/prof_dense_${tag}_$n.log 2>&1; done
fi
if [ $part = all ] || [ $part = bench ]; then
$B --steps 20 --warmup 5 > $out/bench256_driver_like_20steps.json 2> $out/bench256_driver_like_20steps.err
$B --steps 10 --warmup 3 $Q > $out/bench256_default_10steps.json 2>/dev/null
$B --steps 10 --warmup 3 --gpus 1 --force-comm $Q --no-strict > $out/bench256_forcecomm_10steps.json 2>/dev/null
$B --size 512 --steps 10 --warmup 3 $Q > $out/bench512.json 2>/dev/null
$B --workload honey --size 256 --viscosity 50 --steps 10 --warmup 3 $Q > $out/bench_honey256.json 2>/dev/null
$B --workload honey --size 512 --viscosity 50 --steps 5 --warmup 2 $Q > $out/bench_honey512.json 2>/dev/null
$B --workload sheet --size 1024 --steps 3 --warmup 1 $Q --no-strict > $out/bench_sheet1024.json 2>/dev/null
python3 bench.py --gpus 2 --comm host --steps 10 --warmup 3 $Q --no-strict > $out/bench256_two_processes_host_comm.json 2> $out/bench256_two_processes_host_comm.err
python3 bench.py --gpus 4 --comm host --dims 2,2,1 --steps 10 --warmup 3 $Q --no-strict > $out/bench256_four_processes_host_comm.json 2> $out/bench256_four_processes_host_comm.err
for f in $out/*.json; do python3 - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    s = d.get("mode_b_strict") or {}
    print("%-52s %8.1f %s  %.2f ms/step  n_gpus %d  its %s  statuses %s  strict %s" % (sys.argv[1].split("/")[-1], d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d.get("mode_b", {}).get("mean_viscosity_iterations"), sorted(set(d.get("viscosity_status_per_step", []))), s.get("value")))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done > $out/bench_summary.log
fi
if [ $part = all ] || [ $part = soak ]; then
(python3 tools/r3_status.py bunny 256 5 2000; python3 tools/r3_status.py honey 256 50 200) > $out/soak_status.log 2>&1
fi
