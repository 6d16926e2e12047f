#!/usr/bin/env python3
"""Per-kernel table of the SQ counter passes of tools/gpu_round_profile.sh (rocprofv3 --pmc, counter_collection.csv files).
usage: pmc_table.py <csv> [<csv> ...] > profiles/rNN_sq_counters.md
Derived columns: lane_util = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)  (share of the 64 lanes a vector instruction had active),
valu_per_wave / salu_per_wave = instructions per wavefront, wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wavefront parked on s_waitcnt / barrier),
lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS (conflict cycles per busy LDS cycle)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(int)
for path in sys.argv[1:]:
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("psvr::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if r["Counter_Name"] == "SQ_WAVES" and key not in seen:
            seen.add(key)
            launches[k] += 1


def g(d, k):
    return d.get(k, 0.0)


rows = []
for k, d in agg.items():
    waves = g(d, "SQ_WAVES") or 1.0
    rows.append((g(d, "SQ_BUSY_CYCLES"), k, launches[k], waves,
                 g(d, "SQ_INSTS_VALU") / waves, g(d, "SQ_INSTS_SALU") / waves, g(d, "SQ_INSTS_LDS") / waves, g(d, "SQ_INSTS_SMEM") / waves,
                 g(d, "SQ_THREAD_CYCLES_VALU") / (64.0 * g(d, "SQ_ACTIVE_INST_VALU")) if g(d, "SQ_ACTIVE_INST_VALU") else 0.0,
                 g(d, "SQ_WAIT_ANY") / g(d, "SQ_WAVE_CYCLES") if g(d, "SQ_WAVE_CYCLES") else 0.0,
                 g(d, "SQ_ACTIVE_INST_VALU") / g(d, "SQ_WAVE_CYCLES") if g(d, "SQ_WAVE_CYCLES") else 0.0,
                 g(d, "SQ_LDS_BANK_CONFLICT") / g(d, "SQ_ACTIVE_INST_LDS") if g(d, "SQ_ACTIVE_INST_LDS") else 0.0,
                 g(d, "SQ_ACTIVE_INST_LDS") / g(d, "SQ_WAVE_CYCLES") if g(d, "SQ_WAVE_CYCLES") else 0.0))
print("| kernel | launches | waves | VALU/wave | SALU/wave | LDS/wave | SMEM/wave | lane_util | wait_share | valu_busy_share | lds_conflict | lds_busy_share |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for row in sorted(rows, reverse=True):
    _, k, n, waves, valu, salu, lds, smem, lu, ws, vb, lc, lb = row
    print("| %s | %d | %.0f | %.0f | %.0f | %.1f | %.1f | %.2f | %.2f | %.3f | %.3f | %.4f |" % (k, n, waves, valu, salu, lds, smem, lu, ws, vb, lc, lb))
