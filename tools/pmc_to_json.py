#!/usr/bin/env python3
"""profiles/pmc_latest.json from the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs) of
`bench.py --one-pass --steps 1 --warmup 0 ...` (one engine pass per process: every counter value belongs to that one step).
usage: pmc_to_json.py <fetch.csv> <write.csv> <pairs_per_gpu> <out.json> [<engine passes per process> [<commit>]]"""
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
passes = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
out = {"pairs_per_gpu": int(sys.argv[3]), "passes_per_process": passes, "unit": "KiB", "commit": sys.argv[6] if len(sys.argv) > 6 else None,
       "command": "bench.py --one-pass --steps 1 --warmup 0", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    out["kernels"][k] = {"fetch_KiB_per_step": round(fetch.get(k, 0.0) / passes, 1), "write_KiB_per_step": round(write.get(k, 0.0) / passes, 1)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("wrote", sys.argv[4], len(out["kernels"]), "kernels")
