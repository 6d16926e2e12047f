"""The full-path oracle (oracle/aln_oracle.cpp) is pinned against the records the REFERENCE's own
aligner objects printed for the same seeded inputs (tests/golden/<set>/*.jsonl.gz, produced by
tests/golden/gen_aln_golden.py with oracle/_ref/ref_aln): per-pair candidate lists, CIGARs, scores,
pairing decisions, and hashes of every read-strand's sorted seed list and chaining DP."""
import ctypes
import os

import pytest

import aln_common as ac
import datasets

CASES = [(n, r) for n in datasets.DATASETS for r in datasets.DATASETS[n]["reads"]
         if os.path.exists(os.path.join(ac.golden_dir(n), r + ".jsonl.gz"))]


@pytest.mark.parametrize("name,rname", CASES)
def test_oracle_matches_reference_records(name, rname):
    want = ac.golden_lines(name, rname)
    got = ac.run_oracle(name, rname, trace=True)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if a != b]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\norc: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])


SCORE_SETS = [(3, 9, 12, 2, 24, 1, 200), (1, 4, 6, 1, 20, 0, 50), (2, 30, 40, 3, 60, 2, 400)]


def score_tag(score):
    return "score_" + "_".join(str(x) for x in score)


@pytest.mark.parametrize("score", SCORE_SETS)
def test_oracle_matches_reference_records_with_scoring_options(score):
    """-M -m -O -E -P -F -z through the reference's own option parser and aligner objects (tests/golden/fx2/reads150.score_*.jsonl.gz)."""
    want = ac.golden_lines("fx2", "reads150." + score_tag(score))
    got = ac.run_oracle("fx2", "reads150", trace=True, score=score)
    assert len(got) == len(want) and got == want
    assert want != ac.golden_lines("fx2", "reads150")          # the options really changed the results


def test_private_rand_matches_libc():
    """orc::Rand3 must reproduce glibc rand() (seed 1) -- only meaningful on a glibc host."""
    libc = ctypes.CDLL(None)
    # the oracle consumes rand() only through its own generator; compare via the first golden pair whose
    # outcome depends on draws is covered above.  Here: the raw sequence, through a tiny helper in liboracle.
    lib = ctypes.CDLL(os.path.join(ac.ROOT, "oracle", "liboracle.so"))
    lib.orc_rand_selftest.restype = ctypes.c_int
    buf = (ctypes.c_int32 * 1000)()
    lib.orc_rand_selftest(1, buf, 1000)
    libc.srand(1)
    assert [int(x) for x in buf] == [libc.rand() for _ in range(1000)]
