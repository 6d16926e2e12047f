"""The oracle restatement of the DP (oracle/ksw_oracle.c) is pinned against
(1) the committed known-answer vectors produced by the reference kswlib itself, and
(2) where oracle/_ref exists, the live reference objects on fresh random cases."""
import gzip
import json
import os

import pytest

from ksw_cases import random_cases
from ksw_ref import ref_available, run_oracle, run_ref

HERE = os.path.dirname(os.path.abspath(__file__))


def load_kat():
    with gzip.open(os.path.join(HERE, "golden", "ksw_kat.json.gz"), "rt") as f:
        recs = json.load(f)
    for r in recs:
        r["query"] = ["ACGTN".index(ch) for ch in r["query"]]
        r["target"] = ["ACGTN".index(ch) for ch in r["target"]]
    return recs


def diff(a, b):
    return {k: (a[k], b[k]) for k in a if a[k] != b[k]}


@pytest.mark.parametrize("kind", ["extd2", "extz2"])
def test_oracle_matches_reference_kat(kind):
    recs = load_kat()
    assert len(recs) >= 600
    bad = []
    for i, r in enumerate(recs):
        got = run_oracle(r, kind)
        if got != r[kind]:
            bad.append((i, r["flag"], len(r["query"]), len(r["target"]), diff(r[kind], got)))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:3])


@pytest.mark.skipif(not ref_available(), reason="oracle/_ref not built (no /root/reference here)")
@pytest.mark.parametrize("kind", ["extd2", "extz2"])
def test_oracle_matches_live_reference_random(kind):
    bad = []
    for i, c in enumerate(random_cases(4242, 1500, 220)):
        a, b = run_ref(c, kind), run_oracle(c, kind)
        if a != b:
            bad.append((i, len(c["query"]), len(c["target"]), diff(a, b)))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:3])
