set -x
mkdir -p gpurun_out/r5
V=default,old,eta1e-5,eta1e-4
python tools/r5_late_scan.py bunny 64 200 70 1,40,70 $V > gpurun_out/r5/late3_bunny64_nu200.log 2>&1
python tools/r5_late_scan.py bunny 64 5 110 18,69,86,110 $V > gpurun_out/r5/late3_bunny64_nu5.log 2>&1
python tools/r5_late_scan.py bunny 64 0.001 36 3,5,25,36 $V > gpurun_out/r5/late3_bunny64_nu0.001.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense > gpurun_out/r5/bench_velcrit2.json 2> gpurun_out/r5/bench_velcrit2.err
