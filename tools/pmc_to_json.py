#!/usr/bin/env python3
"""profiles/pmc_latest.json from the two rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --cpu-pairs 0`
(5 engine passes per process: timed, timing, stats, two PCIe-inclusive).  usage: pmc_to_json.py <fetch.csv> <write.csv> <pairs_per_gpu> <out.json>"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psvr::", "")] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
passes = 5.0
out = {"pairs_per_gpu": int(sys.argv[3]), "passes_per_process": passes, "unit": "KiB", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    out["kernels"][k] = {"fetch_KiB_per_step": round(fetch.get(k, 0.0) / passes, 1), "write_KiB_per_step": round(write.get(k, 0.0) / passes, 1)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("wrote", sys.argv[4], len(out["kernels"]), "kernels")
