#!/usr/bin/env python3
"""Generates tests/golden/ksw_kat.json.gz: inputs from tests/ksw_cases.fixed_cases() and the
outputs of the REFERENCE kswlib itself (oracle/_ref/libref_ksw.so, compiled from
/root/reference/src/kswlib by oracle/Makefile).  Run in the build container only."""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from ksw_cases import fixed_cases  # noqa: E402
from ksw_ref import ref_available, run_ref  # noqa: E402

assert ref_available(), "build oracle/_ref first: make -C oracle"
out = []
for c in fixed_cases():
    rec = {k: v for k, v in c.items() if k not in ("query", "target")}
    rec["query"] = "".join("ACGTN"[x] for x in c["query"])
    rec["target"] = "".join("ACGTN"[x] for x in c["target"])
    rec["extd2"] = run_ref(c, "extd2")
    c2 = dict(c)
    rec["extz2"] = run_ref(c2, "extz2")
    out.append(rec)
with gzip.open(os.path.join(HERE, "ksw_kat.json.gz"), "wt") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote", len(out), "cases")
