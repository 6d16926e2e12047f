#!/usr/bin/env python3
"""Idle time between consecutive kernels of one engine step, from a rocprofv3 --kernel-trace CSV.
usage: trace_gaps.py <kernel_trace.csv> [step_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psvr::", "")[:28]) for r in rows]
# a step starts with the k_fill_i64 triple that resets the offsets (run()); fall back to k_prep pairs
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_iota")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
a, b = starts[k], starts[k + 1]
busy = sum(e - s for s, e, _ in ev[a:b])
gaps = sorted(((ev[i][0] - ev[i - 1][1], ev[i - 1][2], ev[i][2]) for i in range(a + 1, b)), reverse=True)
print("step %d: span %.3f ms, kernels %.3f ms (%d launches), idle %.3f ms" % (k, (ev[b][0] - ev[a][0]) / 1e6, busy / 1e6, b - a, sum(g for g, _, _ in gaps) / 1e6))
for g, p, n in gaps[:15]:
    print("  %7.1f us  %s -> %s" % (g / 1e3, p, n))
