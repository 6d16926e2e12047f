"""-m gpu: the whole `aln` hot path on the MI355X through the drop-in CLI (which drives the C ABI:
psvr_index_load / psvr_engine_upload / run / download) against the records the REFERENCE's own
aligner objects produced for the same seeded inputs (tests/golden/<set>/*.jsonl.gz): candidate lists,
CIGARs, scores, mapq, pairing decisions and per-strand seed/chain hashes, bit for bit."""
import os
import subprocess
import tempfile

import pytest

import aln_common as ac
import datasets
from test_emu_aln import CASES, normalise

pytestmark = pytest.mark.gpu
CLI = os.path.join(ac.ROOT, "pansvr_amd", "bin", "panSVR")


@pytest.mark.parametrize("name,rname", CASES)
def test_gpu_engine_matches_reference_records(name, rname):
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_gpu_")
    rec = os.path.join(tmp, "records.jsonl")
    cmd = [CLI, "aln", "-S", "-o", os.path.join(tmp, "out.sam"), "-p", os.path.join(tmp, "ori.sam"), "--records", rec, "--trace",
           os.path.join(ac.golden_dir(name), "idx"), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in open(rec).read().split("\n") if l.strip()]
    want = ac.golden_lines(name, rname)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\ngpu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])
    sam = [l for l in open(os.path.join(tmp, "out.sam")) if not l.startswith("@")]
    n_gain = sum(1 for l in want if '"pe":[' in l and normalise(l)["pe"][3])
    assert len(sam) > 0 and len(sam) <= 2 * n_gain
