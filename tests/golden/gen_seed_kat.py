#!/usr/bin/env python3
"""Regenerates tests/golden/seed_kat.jsonl.gz: known answers for seam B3 from the REFERENCE's own deBGA_INDEX::search_kmer and
::UNITIG_MEM_search (oracle/_ref/ref_seed, built from /root/reference by oracle/Makefile).  BUILD CONTAINER ONLY; needs the
dense reference-built indexes gen_aln_golden.py leaves under $PSVR_GOLDEN_ROOT/<set>/idx."""
import gzip
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import aln_common as ac  # noqa: E402

SETS = [("fx1", "reads150", 40), ("fx2", "reads150", 60), ("fx3", "repeat", 20)]
root = os.environ.get("PSVR_GOLDEN_ROOT", "/tmp/gold")
out = []
for name, rname, n in SETS:
    w = ac.workdir(name)
    res = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_seed"), os.path.join(root, name, "idx"), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), str(n)],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    lines = [l for l in res.split("\n") if l.startswith("{")]
    out += ['{"set":"%s",%s' % (name, l[1:]) for l in lines]
    print(name, rname, len(lines), "lines")
with open(os.path.join(HERE, "seed_kat.jsonl.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
    f.write(("\n".join(out) + "\n").encode())
