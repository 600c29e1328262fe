set -x
mkdir -p gpurun_out/r5
python tools/r5_late_scan.py bunny 64 5 18 18 default,rounds2,rounds3,tight > gpurun_out/r5/late4_nu5_18.log 2>&1
python -m pytest tests/test_gpu_late_states.py -x -q -m gpu -s > gpurun_out/r5/test_late.log 2>&1
tail -n 30 gpurun_out/r5/test_late.log
python -m pytest tests -x -q -m gpu > gpurun_out/r5/test_all.log 2>&1
tail -n 15 gpurun_out/r5/test_all.log
