"""A15 pinned: the product's record formatter (pansvr_amd/csrc/sam_emit.h + fastq_batch.h, the host code `panSVR aln` writes
its two files with) against the REFERENCE's own `fc_aln -t 1 -S` output, byte for byte.  The goldens
(tests/golden/<set>/<reads>.sam.gz / .ori.sam.gz) were written by the reference objects -- output_BAM / output_ori_bam ->
htslib sam_parse1 -> sam_format1, compiled from /root/reference by oracle/Makefile -- through tests/golden/gen_aln_golden.py.
Here the results come from the CPU emulation of the engine (tests/emu: same stage functions, oracle DP), so this runs
without a GPU; tests/test_aln_gpu.py makes the same comparison with the real engine behind the CLI."""
import gzip
import os
import subprocess
import tempfile

import pytest

import aln_common as ac
from test_emu_aln import CASES, EMU


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ac.HERE, "emu")])
    return EMU


def golden_text(name, rname, ext):
    with gzip.open(os.path.join(ac.golden_dir(name), rname + ext), "rb") as f:
        return f.read()


@pytest.mark.parametrize("name,rname", CASES)
def test_formatter_writes_the_reference_sam_files(emu, name, rname):
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_samg_")
    r = subprocess.run([emu, ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), "--no-records",
                        "--sam", os.path.join(tmp, "o.sam"), "--ori-sam", os.path.join(tmp, "p.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    for got_fn, ext in (("o.sam", ".sam.gz"), ("p.sam", ".ori.sam.gz")):
        got, want = open(os.path.join(tmp, got_fn), "rb").read(), golden_text(name, rname, ext)
        if got != want:
            gl, wl = got.split(b"\n"), want.split(b"\n")
            first = next((i for i, (a, b) in enumerate(zip(wl, gl)) if a != b), min(len(gl), len(wl)))
            raise AssertionError("%s differs from the reference's file at line %d (%d vs %d lines)\nref: %r\ngot: %r"
                                 % (got_fn, first, len(wl), len(gl), wl[first][:600] if first < len(wl) else None, gl[first][:600] if first < len(gl) else None))
    assert golden_text(name, rname, ".sam.gz").count(b"\n") > 40


NOT_ORI = [("fx1", "reads150"), ("fx2", "reads150")]


@pytest.mark.parametrize("name,rname", NOT_ORI)
def test_formatter_not_ori_option_matches_the_reference(emu, name, rname):
    """`-Q` / --not-ori (read_realignment.cpp:485: output_BAM returns before writing an ORIGINAL primary): both files of the reference's
    `fc_aln -t 1 -S -Q` run (tests/golden/<set>/<reads>.notori.*, written by ref_aln -Q) against the product formatter with not_ori set."""
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_samq_")
    r = subprocess.run([emu, ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), "--no-records", "-Q",
                        "--sam", os.path.join(tmp, "o.sam"), "--ori-sam", os.path.join(tmp, "p.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    for got_fn, ext in (("o.sam", ".notori.sam.gz"), ("p.sam", ".notori.ori.sam.gz")):
        got, want = open(os.path.join(tmp, got_fn), "rb").read(), golden_text(name, rname, ext)
        assert got == want, "%s differs from the reference's -Q file" % got_fn
    # the option really drops records: the -Q main file is a proper subset of the default one
    assert golden_text(name, rname, ".notori.sam.gz").count(b"\n") < golden_text(name, rname, ".sam.gz").count(b"\n")


@pytest.mark.parametrize("name,rname", [("fx2", "reads150"), ("fx3", "ragged"), ("fx3", "lower"), ("fx1", "reads150")])
def test_direct_bam_encoder_writes_the_bytes_of_the_text_path(emu, name, rname):
    """The main file's BAM records come straight from the engine's results (sam_emit.h, the direct encoder); BamWriter::encode on the
    SAM line's fields -- the path that was checked against the reference's SAM text through an independent BAM reader -- must give the
    same bytes, record for record (fx2: N bases, unmapped and full-score originals, secondary candidates; fx3: ragged lengths incl. even
    ones on the reverse strand, lower-case bases)."""
    w = ac.workdir(name)
    if not os.path.exists(os.path.join(w, rname + ".fq")):
        pytest.skip("no such read set")
    tmp = tempfile.mkdtemp(prefix="psvr_bamenc_")
    outs = []
    for tag, extra in (("direct", []), ("text", ["--bam-via-text"])):
        fn = os.path.join(tmp, tag + ".bamrec")
        r = subprocess.run([emu, ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), "--no-records", "--sam", os.path.join(tmp, tag + ".sam"),
                            "--ori-sam", os.path.join(tmp, tag + ".ori.sam"), "--bam-records", fn] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        outs.append(open(fn, "rb").read())
    assert len(outs[0]) > 10000 and outs[0] == outs[1]
