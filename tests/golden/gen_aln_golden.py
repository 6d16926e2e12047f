#!/usr/bin/env python3
"""Regenerates the end-to-end golden fixtures of the `aln` path.  BUILD CONTAINER ONLY: it runs the
reference's own programs compiled under oracle/_ref/ (deBGA index builder + ref_aln, the reference
aligner objects behind oracle/ref_harness/ref_aln_main.cpp).

For every data set in DATASETS: synthesize anchors/reads (tests/synth.py, seeded), build the index
with the reference deBGA (-k 22), run ref_aln --trace, and commit
    tests/golden/<name>/idx/          compact index (tests/index_fixture.py)
    tests/golden/<name>/<reads>.jsonl.gz   one record per read pair, as printed by the reference objects
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import index_fixture  # noqa: E402
import datasets  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def main(names):
    for name in names:
        ds = datasets.DATASETS[name]
        work = os.environ.get("PSVR_GOLDEN_WORK", tempfile.mkdtemp(prefix="psvr_" + name))
        os.makedirs(work, exist_ok=True)
        datasets.materialize(name, work)
        idx = os.path.join(work, "idx")
        if not os.path.exists(os.path.join(idx, "unipath_g.hash")):
            os.makedirs(idx, exist_ok=True)
            subprocess.check_call([os.path.join(REF, "deBGA"), "index", "-k", "22", os.path.join(work, "anchors.fa"), idx + "/"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = os.path.join(HERE, name)
        os.makedirs(out, exist_ok=True)
        index_fixture.compact(idx, os.path.join(out, "idx"))
        for rname in ds["reads"]:
            res = subprocess.run([os.path.join(REF, "ref_aln"), idx, os.path.join(work, rname + ".fq"), os.path.join(work, "header.sam"), "--trace"],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
            lines = [l for l in res.stdout.decode().split("\n") if l.startswith("{") or l.startswith(" {")]
            with open(os.path.join(out, rname + ".jsonl.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:   # mtime=0: reproducible bytes
                f.write(("\n".join(l.strip() for l in lines) + "\n").encode())
            print(name, rname, len(lines), "pairs")


if __name__ == "__main__":
    main(sys.argv[1:] or list(datasets.DATASETS))
