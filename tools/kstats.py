"""print selected rows of a rocprofv3 kernel_stats.csv: python tools/kstats.py file.csv [substr ...]"""
import csv
import sys
keys = sys.argv[2:]
for r in csv.DictReader(open(sys.argv[1])):
    if not keys or any(k in r["Name"] for k in keys):
        print("%-60s %6s calls  avg %10.1f us  %5s %%" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
