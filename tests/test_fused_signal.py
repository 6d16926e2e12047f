"""f2 fused (`panSVR aln x.bam`): the signal step hands its pairs to the batch reader without FASTQ text (PairFeed, fastq_batch.h).
A BAM is written here from a golden read set (the reads of tests/golden/fx1 with their original alignments as BAM fields); the
engine's records and both SAM files must be the same whether the BAM goes in directly or `panSVR signal` first writes the FASTQ
that is then read like any other -- i.e. the numbers the signal step hands over are what the comment parser would have read, the
batch limits fall where load_reads puts them, and names / comments / qualities reach the output records unchanged.
CPU: through the emulated engine (tests/emu); -m gpu: through the drop-in command."""
import os
import subprocess
import tempfile

import pytest

import aln_common as ac
import synth
import test_signal as ts
from test_emu_aln import EMU

CLI = os.path.join(ac.ROOT, "pansvr_amd", "bin", "panSVR")


def bam_of(name, rname, n_pairs, path):
    """The first n_pairs pairs of a golden FASTQ as a name-sorted BAM: fields from the comment's FLAG_/CIGAR_/MATE_ part."""
    lines = open(os.path.join(ac.workdir(name), rname + ".fq")).read().split("\n")
    recs = []
    for p in range(n_pairs):
        for k in range(2):
            h, seq, _, qual = lines[8 * p + 4 * k: 8 * p + 4 * k + 4]
            qname, cm = h[1:].split(" ", 1)
            tok = cm.split("_")
            tid, pos = int(tok[0]), int(tok[1])
            f = cm.split("FLAG_")[1].split("_")
            flag, mapq = int(f[0]), int(f[1])
            cig = cm.split("CIGAR_")[1].split("_")[0]
            mate = cm.split("MATE_")[1].split("_")
            cigar, num = [], ""
            for ch in cig:
                if ch.isdigit():
                    num += ch
                else:
                    cigar.append((int(num), ch))
                    num = ""
            if flag & 0x10:                                   # BAM stores the reference strand
                seq, qual = synth.revcomp(seq.encode()).decode(), qual[::-1]
            if tid > 24:
                tid = 29
            recs.append(ts.record(qname, flag, tid, pos, mapq, cigar, int(mate[0]), int(mate[1]), int(mate[2]), seq.replace("n", "N").upper(),
                                  [ord(c) - 33 for c in qual], [("NM", "i", 3)]))
    refs = [("chr%d" % i, 250000000) for i in range(1, 23)] + [("chrX", 250000000), ("chrY", 250000000), ("chrM", 250000000)] + [("decoy%d" % i, 250000000) for i in range(1, 8)]
    ts.write_bam(path, recs, refs)


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ac.HERE, "emu")])
    return EMU


@pytest.mark.parametrize("batch", [1 << 20, 37])
def test_bam_input_equals_signal_then_fastq_on_the_emulated_engine(emu, batch):
    tmp = tempfile.mkdtemp(prefix="psvr_fused_")
    bam = os.path.join(tmp, "in.bam")
    bam_of("fx1", "reads150", 500, bam)
    idx = ac.index_dir("fx1")
    # route B: `panSVR signal -N -D` -> FASTQ -> reader
    fq = os.path.join(tmp, "sig.fq")
    with open(fq, "wb") as f:
        r = subprocess.run([CLI, "signal", "-N", "-D", "-H", os.path.join(tmp, "hB.sam"), "-S", os.path.join(tmp, "sB.txt"), bam], stdout=f, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    outs = {}
    for tag, reads, hdr, extra in (("A", bam, os.path.join(tmp, "hA.sam"), ["-N", "-D"]), ("B", fq, os.path.join(tmp, "hB.sam"), [])):
        r = subprocess.run([emu, idx, reads, hdr, "--trace", "--batch", str(batch), "--sam", os.path.join(tmp, tag + ".sam"), "--ori-sam", os.path.join(tmp, tag + ".ori.sam")] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        outs[tag] = (r.stdout, open(os.path.join(tmp, tag + ".sam"), "rb").read(), open(os.path.join(tmp, tag + ".ori.sam"), "rb").read())
    assert open(os.path.join(tmp, "hA.sam"), "rb").read() == open(os.path.join(tmp, "hB.sam"), "rb").read()
    assert outs["A"][0].count(b"\n") == 500 and outs["A"] == outs["B"]
    assert len(outs["A"][1]) > 100000


@pytest.mark.gpu
def test_bam_input_equals_signal_then_fastq_on_the_gpu():
    tmp = tempfile.mkdtemp(prefix="psvr_fused_")
    bam = os.path.join(tmp, "in.bam")
    bam_of("fx1", "reads150", 2000, bam)
    idx = ac.index_dir("fx1")
    fq = os.path.join(tmp, "sig.fq")
    with open(fq, "wb") as f:
        r = subprocess.run([CLI, "signal", "-N", "-D", "-H", os.path.join(tmp, "hB.sam"), "-S", os.path.join(tmp, "sB.txt"), bam], stdout=f, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    outs = {}
    for tag, reads, hdr, extra in (("A", bam, os.path.join(tmp, "hA.sam"), ["-N", "-D"]), ("B", fq, os.path.join(tmp, "hB.sam"), [])):
        for sub in ("65536", "97"):
            r = subprocess.run([CLI, "aln", "-S", "--sub-batch", sub, "-o", os.path.join(tmp, tag + ".sam"), "-p", os.path.join(tmp, tag + ".ori.sam")] + extra + [idx, reads, hdr],
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            got = (open(os.path.join(tmp, tag + ".sam"), "rb").read(), open(os.path.join(tmp, tag + ".ori.sam"), "rb").read())
            assert outs.setdefault(tag, got) == got                  # the pieces a batch is cut into do not show
    assert outs["A"] == outs["B"] and len(outs["A"][0]) > 500000


GOLDEN = [("fx1", "reads150", 2000, ["-D"]), ("fx2", "reads150", 1500, ["-D"])]


def _golden(name, rname):
    import gzip
    out = []
    for ext in (".sam.gz", ".ori.sam.gz"):
        with gzip.open(os.path.join(ac.HERE, "golden", "fused", "%s_%s%s" % (name, rname, ext)), "rb") as f:
            out.append(f.read())
    return out


@pytest.mark.parametrize("name,rname,n_pairs,flags", GOLDEN)
def test_bam_input_writes_what_the_references_two_steps_write(emu, name, rname, n_pairs, flags):
    """tests/golden/fused/*: the reference's own `fc_signal` function (oracle/_ref/ref_signal) followed by the reference's own `fc_aln`
    objects (oracle/_ref/ref_aln -t 1 -S) on the same BAM (tests/golden/gen_fused_golden.py).  The fused route -- BAM in, the signal
    step in-process, pairs handed over without FASTQ text -- must write both files byte for byte; here on the emulated engine."""
    tmp = tempfile.mkdtemp(prefix="psvr_fused_")
    bam = os.path.join(tmp, "in.bam")
    bam_of(name, rname, n_pairs, bam)
    r = subprocess.run([emu, ac.index_dir(name), bam, os.path.join(tmp, "h.sam"), "--no-records", "--sam", os.path.join(tmp, "o.sam"), "--ori-sam", os.path.join(tmp, "p.sam"), "-N"] + flags,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    want = _golden(name, rname)
    assert open(os.path.join(tmp, "o.sam"), "rb").read() == want[0]
    assert open(os.path.join(tmp, "p.sam"), "rb").read() == want[1]


@pytest.mark.gpu
@pytest.mark.parametrize("name,rname,n_pairs,flags", GOLDEN)
def test_gpu_bam_input_writes_what_the_references_two_steps_write(name, rname, n_pairs, flags):
    """The same on the MI355X through the drop-in command: `panSVR aln -N -D -S ... in.bam header.sam`."""
    tmp = tempfile.mkdtemp(prefix="psvr_fused_")
    bam = os.path.join(tmp, "in.bam")
    bam_of(name, rname, n_pairs, bam)
    r = subprocess.run([CLI, "aln", "-S", "-N"] + flags + ["-o", os.path.join(tmp, "o.sam"), "-p", os.path.join(tmp, "p.sam"), ac.index_dir(name), bam, os.path.join(tmp, "h.sam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    want = _golden(name, rname)
    assert open(os.path.join(tmp, "o.sam"), "rb").read() == want[0]
    assert open(os.path.join(tmp, "p.sam"), "rb").read() == want[1]
