#!/usr/bin/env python3
"""Regenerates tests/golden/ib1: a hand-made anchor set for the index builder test and the index the REFERENCE builder
(oracle/_ref/deBGA, compiled by oracle/Makefile) writes for it.  BUILD CONTAINER ONLY.  The dense first-level table is stored
in its (bucket, count) form (tests/index_fixture.py)."""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import index_fixture  # noqa: E402


def anchors():
    rng = np.random.RandomState(77)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.randint(0, 4, size=n))
    recs = []
    core = rnd(300)
    recs.append(("s1 first chromosome with description", rnd(200) + core + rnd(150)))
    recs.append(("s2", rnd(100) + "NNNNNNNNNN" + rnd(120) + "N" + rnd(60)))            # N runs
    recs.append(("s3", (rnd(80) + core[:150]).lower() + rnd(90)))                   # lower case, shares part of the core
    recs.append(("s4", rnd(50) + "RYKMSWBDHV" + rnd(70)))                            # IUPAC codes
    recs.append(("s5", rnd(15)))                                                 # shorter than k
    recs.append(("s6", "ACGT" * 40))                                               # tandem repeat (cycle in the graph)
    recs.append(("s7", recs[0][1]))                                              # exact duplicate of s1
    recs.append(("s8", rnd(22)))                                                 # exactly one k-mer
    recs.append(("s9", core[100:250] + rnd(30) + core[100:250]))                     # internal repeat
    return recs


def main():
    out = os.path.join(HERE, "ib1")
    os.makedirs(out, exist_ok=True)
    fa = os.path.join(out, "anchors.fa")
    with open(fa, "w") as f:
        for n, s in anchors():
            f.write(">%s\n" % n)
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60] + "\n")
    work = tempfile.mkdtemp(prefix="psvr_ib1")
    os.makedirs(os.path.join(work, "idx"))
    subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "deBGA"), "index", "-k", "22", fa, os.path.join(work, "idx") + "/"],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=work)
    shutil.rmtree(os.path.join(out, "idx"), ignore_errors=True)
    index_fixture.compact(os.path.join(work, "idx"), os.path.join(out, "idx"))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
