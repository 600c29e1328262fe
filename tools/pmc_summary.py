#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs with
--kernel-trace only, as MI355X_MICROARCH.md prescribes) into profiles/<round>/pmc_traffic.json.

Corrections (gfx950, this rocprofv3): FETCH_SIZE counts 64 B per 128-B fabric read request, i.e. reports half the
bytes of a coalesced streaming read -> x2; WRITE_SIZE calibrates 1:1.  Both were checked here on kernels with a known
byte count (k_pcg_update: 5 arrays read, 3 written), see DESIGN.md.
"""
import collections
import csv
import json
import sys


def agg(path, cname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == cname:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")   # (kernels in anonymous namespaces would otherwise all collapse to "")
            d[name.split("(")[0]].append(float(r["Counter_Value"]))
    return d


def main(fetch_csv, write_csv, out_json):
    f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
    out = {}
    for k in f:
        fm = sum(f[k]) / len(f[k])
        wm = sum(w[k]) / len(w[k]) if k in w and w[k] else 0.0
        out[k] = {"launches": len(f[k]), "FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm,
                  "hbm_read_bytes": 2.0 * fm * 1024.0, "hbm_write_bytes": wm * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * fm * 1024.0 + wm * 1024.0}
    json.dump(out, open(out_json, "w"), indent=1, sort_keys=True)
    for k in sorted(out, key=lambda k: -out[k]["hbm_bytes_per_launch"] * out[k]["launches"])[:8]:
        print("%-34s %6d launches  %8.1f MB per launch" % (k[:34], out[k]["launches"], out[k]["hbm_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
