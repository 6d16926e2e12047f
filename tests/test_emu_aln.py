"""CPU check of the engine's host logic: the same stage functions (pansvr_amd/csrc/aln_device.h) and
batch orchestration (engine_core.h, incl. the speculative rand()-offset loop and arena growth) the
GPU engine uses, executed by the test-only host backend tests/emu against the reference's records.
(The oracle's DP stands in for the HIP DP kernel here; the GPU path itself is covered by -m gpu.)"""
import json
import os
import subprocess

import pytest

import aln_common as ac
import datasets

EMU = os.path.join(ac.HERE, "emu", "emu_aln")
CASES = [(n, r) for n in datasets.DATASETS for r in datasets.DATASETS[n]["reads"]
         if os.path.exists(os.path.join(ac.golden_dir(n), r + ".jsonl.gz"))]


def normalise(line):
    """Trace fields of a read that produced nothing are not results: for early-out reads the reference's
    vectors still hold the handler's previous read."""
    d = json.loads(line)
    for r in d["reads"]:
        if r["n"] == 0:
            r.pop("tr", None)
            r.pop("str", None)
    return d


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ac.HERE, "emu")])
    return EMU


@pytest.mark.parametrize("name,rname", CASES)
def test_emulated_engine_matches_reference_records(emu, name, rname):
    w = ac.workdir(name)
    out = subprocess.run([emu, os.path.join(ac.golden_dir(name), "idx"), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), "--trace"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    got = [l for l in out.split("\n") if l.strip()]
    want = ac.golden_lines(name, rname)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\nemu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])
