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
    # SAM text (A15, restated from output_BAM + sam_parse1/sam_format1; not reference-pinned): every record must be the
    # primary the decision record names, field for field
    sam = [l.rstrip("\n").split("\t") for l in open(os.path.join(tmp, "out.sam")) if not l.startswith("@")]
    by_key = {(f[0], int(f[1]) & 0x40): f for f in sam}
    assert len(by_key) == len(sam)
    n_expected = 0
    for line in want:
        d = normalise(line)
        if not d["pe"][3]:
            continue
        for k, r in enumerate(d["reads"]):
            prim = r.get("prim", -1)
            if prim == -1:
                continue
            rec = r["ori"] if prim == -2 else r["res"][prim]
            if rec[2] < 0 or rec[2] > 1 or rec[3] < 1:
                continue                                   # out-of-header chromosome / POS 0: the reference's record is dropped
            n_expected += 1
            f = by_key[("r%07d" % d["i"], 0x40 if k == 0 else 0)]
            assert int(f[3]) == rec[3] and f[5] == rec[7] and int(f[4]) == rec[6]
            assert ("AS:i:%d" % rec[0]) in f and (int(f[1]) & 0x10 != 0) == (rec[5] == 0)
            assert any(t.startswith("RC:Z:") for t in f) and any(t.startswith("OA:Z:") for t in f)
    assert n_expected == len(sam) and n_expected > 0
    ori = [l for l in open(os.path.join(tmp, "ori.sam")) if not l.startswith("@")]
    assert all("MS:i:" in l for l in ori)


def test_gpu_cli_bam_output_equals_sam_text():
    """Default output is BAM (like the reference's init_run): decode it with tests/bam_reader.py and compare every record,
    field for field and tag for tag, with the SAM text of a -S run of the same input; both files (-o and -p)."""
    import bam_reader
    name, rname = "fx2", "reads150"
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_bam_")
    base = [os.path.join(ac.golden_dir(name), "idx"), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    for mode, ext in (["-S"], "sam"), ([], "bam"):
        r = subprocess.run([CLI, "aln"] + mode + ["-o", os.path.join(tmp, "out." + ext), "-p", os.path.join(tmp, "ori." + ext)] + base,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
    header = open(os.path.join(w, "header.sam")).read()
    for stem in ("out", "ori"):
        assert bam_reader.check_bgzf(os.path.join(tmp, stem + ".bam")) >= 1
        text, refs, recs = bam_reader.read_bam(os.path.join(tmp, stem + ".bam"))
        sam_lines = open(os.path.join(tmp, stem + ".sam")).read().split("\n")
        sam_head = "".join(l + "\n" for l in sam_lines if l.startswith("@"))
        sam = [l.split("\t") for l in sam_lines if l and not l.startswith("@")]
        assert text == sam_head == "".join(l for l in header.splitlines(True) if l.startswith("@"))
        assert [n for n, _ in refs] == [l.split("SN:")[1].split("\t")[0].strip() for l in header.splitlines() if l.startswith("@SQ")]
        assert len(recs) == len(sam) and (stem == "ori" or len(sam) > 100)
        for a, b in zip(sam, recs):
            assert a == b, "\nsam: %s\nbam: %s" % ("\t".join(a), "\t".join(b))


@pytest.mark.parametrize("score", [(3, 9, 12, 2, 24, 1, 200), (1, 4, 6, 1, 20, 0, 50), (2, 30, 40, 3, 60, 2, 400)])
def test_gpu_cli_scoring_options_match_oracle(score):
    """-M -m -O -E -P -F -z: the engine against the CPU restatement run with the same options (the reference objects behind the
    golden records only ran the defaults).  The third set leaves the int8-safe regime, so its DP goes through the wavefront
    kernels instead of the team kernel."""
    name, rname = "fx2", "reads150"
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_score_")
    rec = os.path.join(tmp, "records.jsonl")
    M, m, O, E, P, F, z = score
    cmd = [CLI, "aln", "-S", "-M", str(M), "-m", str(m), "-O", str(O), "-E", str(E), "-P", str(P), "-F", str(F), "-z", str(z), "-o", os.path.join(tmp, "o.sam"), "-p",
           os.path.join(tmp, "p.sam"), "--records", rec, "--trace", os.path.join(ac.golden_dir(name), "idx"), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [normalise(l) for l in open(rec).read().split("\n") if l.strip()]
    want = [normalise(l) for l in ac.run_oracle(name, rname, trace=True, score=score)]
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if a != b]
    assert not bad, "%d/%d pairs differ; first %d:\norc: %s\ngpu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])
    assert got != [normalise(l) for l in ac.golden_lines(name, rname)]        # the options really changed the results


def test_gpu_cli_reads_name_sorted_bam_like_the_two_commands():
    """`panSVR aln <idx> x.bam hdr` (signal step in-process, through a pipe) gives byte for byte what `panSVR signal -N x.bam | panSVR aln <idx> - hdr`
    gives.  The BAM is rebuilt from the golden set's FASTQ comments (FLAG / CIGAR / MATE / TAG), so the reads do align to the index."""
    import re
    import test_signal as ts
    w = ac.workdir("fx1")
    tmp = tempfile.mkdtemp(prefix="psvr_bam_")
    lines = open(os.path.join(w, "reads150.fq")).read().split("\n")
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    recs = []
    for k in range(0, len(lines) - 3, 4):
        name, comment = lines[k][1:].split(" ", 1)
        m = re.search(r"FLAG_(\d+)_(\d+)_CIGAR_([^_]*)_MATE_(-?\d+)_(-?\d+)_(-?\d+)_TAG_(.*)$", comment)
        tok = comment.split("_")
        flag, mapq = int(m.group(1)), int(m.group(2))
        cigar = [(int(n), op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", m.group(3))]
        tags = []
        for t in m.group(7).split("_"):
            if t.count(":") >= 2:
                tg, ty, val = t.split(":", 2)
                tags.append((tg, "i" if ty == "i" else "Z", int(val) if ty == "i" else val))
        seq, qual = lines[k + 1], [ord(c) - 33 for c in lines[k + 3]]
        if flag & 16 and not flag & 4:
            seq, qual = "".join(comp[c] for c in reversed(seq)), qual[::-1]
        recs.append(ts.record(name, flag, int(tok[0]), int(tok[1]), mapq, cigar, int(m.group(4)), int(m.group(5)), int(m.group(6)), seq, qual, tags))
    bam = os.path.join(tmp, "x.bam")
    ts.write_bam(bam, recs, [("chr1", 250000000), ("chr2", 250000000)])
    # route 1: two commands
    fq = os.path.join(tmp, "x.fq")
    r = subprocess.run([CLI, "signal", "-N", "-D", "-H", os.path.join(tmp, "h1.sam"), "-S", os.path.join(tmp, "s1.txt"), bam], stdout=open(fq, "wb"), stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert os.path.getsize(fq) > 100000
    idx = os.path.join(ac.golden_dir("fx1"), "idx")
    outs = []
    for tag, reads, hdr in (("a", fq, os.path.join(tmp, "h1.sam")), ("b", bam, os.path.join(tmp, "h2.sam"))):
        o = os.path.join(tmp, tag)
        r = subprocess.run([CLI, "aln", "-S", "-D", "-o", o + ".sam", "-p", o + ".ori.sam", "--records", o + ".jsonl", idx, reads, hdr], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(o)
    for ext in (".sam", ".ori.sam", ".jsonl"):
        assert open(outs[0] + ext, "rb").read() == open(outs[1] + ext, "rb").read(), ext
    assert os.path.getsize(outs[0] + ".sam") > 10000
    assert open(os.path.join(tmp, "h1.sam"), "rb").read() == open(os.path.join(tmp, "h2.sam"), "rb").read()
