#!/usr/bin/env python3
"""Per-kernel sum / per-launch mean of a rocprofv3 --pmc counter_collection.csv (values in KiB for FETCH_SIZE / WRITE_SIZE)."""
import collections
import csv
import json
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
name = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    agg[k][0] += 1
    agg[k][1] += float(r["Counter_Value"])
    name = r["Counter_Name"]
out = {k: {"launches": n, "sum_KiB": round(v, 1), "per_launch_KiB": round(v / n, 1)} for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])}
print(json.dumps({"counter": name, "kernels": out}, indent=1))
