"""`panSVR index` (pansvr_amd/csrc/index_build.h, host C++) against the reference's own index builder: for every golden
set the files it writes must equal, byte for byte, what oracle/_ref/deBGA wrote for the same anchors
(tests/golden/<set>/idx, committed by tests/golden/gen_aln_golden.py) -- unipath numbering, k-mer offsets, position lists,
the first-level table in its (bucket, count) form, the packed sequences and the chromosome table.  fx2 has duplicated and
short-tandem-repeat anchors, fx3 a 70 bp element shared by all 620 anchors (2002 unipaths, up to 620 positions each)."""
import filecmp
import os
import subprocess
import tempfile

import pytest

import aln_common as ac
import datasets

CLI = os.path.join(ac.ROOT, "pansvr_amd", "bin", "panSVR")
FILES = ["ref.seq", "unipath.seqb", "unipath.seqfb", "unipath.pos", "unipath.posp", "unipath_g.kmer", "unipath_g.offset", "unipath_g.hash.sparse", "unipath.chr"]


# ib1: a hand-made anchor set for the builder alone (tests/golden/ib1/anchors.fa + the reference builder's output): N runs,
# lower case, IUPAC codes, a sequence shorter than k (which still advances the coordinate by k), a tandem repeat (a cycle in
# the graph), an exact duplicate, a sequence of exactly one k-mer, an internal repeat, header descriptions
@pytest.mark.parametrize("name", [n for n in datasets.DATASETS if os.path.isdir(os.path.join(ac.golden_dir(n), "idx"))] + ["ib1"])
def test_index_builder_reproduces_the_reference_index(name):
    fasta = os.path.join(ac.golden_dir(name), "anchors.fa") if name == "ib1" else os.path.join(ac.workdir(name), "anchors.fa")
    out = tempfile.mkdtemp(prefix="psvr_idx_" + name)
    r = subprocess.run([CLI, "index", "-k", "22", "--sparse-hash", fasta, out + "/"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    fx = os.path.join(ac.golden_dir(name), "idx")
    for f in FILES:
        assert filecmp.cmp(os.path.join(out, f), os.path.join(fx, f), shallow=False), "%s differs from the reference builder's" % f


def test_index_builder_reproduces_the_reference_index_by_hash():
    """fx5 (0.5 Mbp of anchors, a unipath with 8100 positions) commits the SHA-256 of the reference builder's files instead of the files;
    ac.index_dir builds the index with `panSVR index` and compares every file's hash."""
    d = ac.index_dir("fx5")
    assert sorted(os.listdir(d)) == sorted(FILES)


def test_index_builder_rejects_other_k():
    r = subprocess.run([CLI, "index", "-k", "20", "a.fa", "d"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0
