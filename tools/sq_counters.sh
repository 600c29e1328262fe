#!/bin/bash
# SQ wave-state counters of the PCG kernels of bench.py (one rocprofv3 --pmc pass, kernel trace only)
export TMPDIR=/tmp
out=gpurun_out/sq; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $out -o s -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/b.json 2> $out/log
python3 - <<'P'
import csv, collections, glob
f = glob.glob("gpurun_out/sq/*counter_collection.csv")[0]
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "spmv" in k or "pcg_update" in k:
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
P
rm -f $out/*counter_collection.csv $out/*trace.csv
