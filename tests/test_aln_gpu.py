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
           ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in open(rec).read().split("\n") if l.strip()]
    want = ac.golden_lines(name, rname)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\ngpu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])
    # A15: both files byte for byte the reference's own `fc_aln -t 1 -S` output (tests/golden/*/<reads>.sam.gz, .ori.sam.gz:
    # output_BAM / output_ori_bam -> sam_parse1 -> sam_format1 of the reference objects)
    import gzip
    for got_fn, ext in (("out.sam", ".sam.gz"), ("ori.sam", ".ori.sam.gz")):
        got = open(os.path.join(tmp, got_fn), "rb").read()
        with gzip.open(os.path.join(ac.golden_dir(name), rname + ext), "rb") as f:
            want = f.read()
        if got != want:
            gl, wl = got.split(b"\n"), want.split(b"\n")
            first = next((i for i, (a, b) in enumerate(zip(wl, gl)) if a != b), min(len(gl), len(wl)))
            raise AssertionError("%s: line %d differs (%d vs %d lines)\nref: %r\ngpu: %r" % (got_fn, first, len(wl), len(gl), wl[first][:500] if first < len(wl) else None,
                                                                                             gl[first][:500] if first < len(gl) else None))


@pytest.mark.parametrize("name,rname", [("fx1", "reads150"), ("fx2", "reads150")])
def test_gpu_cli_not_ori_option_matches_the_reference(name, rname):
    """`panSVR aln -S -Q` on the GPU == the reference's own `fc_aln -t 1 -S -Q` files (read_realignment.cpp:485), byte for byte."""
    import gzip
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_gpuq_")
    r = subprocess.run([CLI, "aln", "-S", "-Q", "-o", os.path.join(tmp, "out.sam"), "-p", os.path.join(tmp, "ori.sam"),
                        ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    for got_fn, ext in (("out.sam", ".notori.sam.gz"), ("ori.sam", ".notori.ori.sam.gz")):
        with gzip.open(os.path.join(ac.golden_dir(name), rname + ext), "rb") as f:
            want = f.read()
        assert open(os.path.join(tmp, got_fn), "rb").read() == want, "%s differs from the reference's -Q file" % got_fn


def test_gpu_cli_bam_output_equals_sam_text():
    """Default output is BAM (like the reference's init_run): decode it with tests/bam_reader.py and compare every record,
    field for field and tag for tag, with the SAM text of a -S run of the same input; both files (-o and -p)."""
    import bam_reader
    name, rname = "fx2", "reads150"
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_bam_")
    base = [ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    for mode, ext in (["-S"], "sam"), ([], "bam"):
        r = subprocess.run([CLI, "aln"] + mode + ["-o", os.path.join(tmp, "out." + ext), "-p", os.path.join(tmp, "ori." + ext)] + base,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
    # ... and with the BGZF blocks compressed on the device (--bgzf-device; the threshold that leaves small files to the host lowered)
    r = subprocess.run([CLI, "aln", "--bgzf-device", "-o", os.path.join(tmp, "outd.bam"), "-p", os.path.join(tmp, "orid.bam")] + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, PSVR_BGZF_DEVICE_MIN_BLOCKS="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert "BGZF on the device failed" not in r.stderr.decode()
    assert bam_reader.check_bgzf(os.path.join(tmp, "outd.bam")) >= 1
    assert bam_reader.read_bam(os.path.join(tmp, "outd.bam")) == bam_reader.read_bam(os.path.join(tmp, "out.bam"))
    assert bam_reader.read_bam(os.path.join(tmp, "orid.bam")) == bam_reader.read_bam(os.path.join(tmp, "ori.bam"))
    # ... and from the built-in encoder on the host threads (--bgzf-fast)
    r = subprocess.run([CLI, "aln", "--bgzf-fast", "-t", "4", "-o", os.path.join(tmp, "outf.bam"), "-p", os.path.join(tmp, "orif.bam")] + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert bam_reader.check_bgzf(os.path.join(tmp, "outf.bam")) >= 1
    assert bam_reader.read_bam(os.path.join(tmp, "outf.bam")) == bam_reader.read_bam(os.path.join(tmp, "out.bam"))
    assert bam_reader.read_bam(os.path.join(tmp, "orif.bam")) == bam_reader.read_bam(os.path.join(tmp, "ori.bam"))
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
def test_gpu_cli_scoring_options_match_reference(score):
    """-M -m -O -E -P -F -z: the engine against what the REFERENCE's objects decided with the same options
    (tests/golden/fx2/reads150.score_*.jsonl.gz, through the reference's own option parser).  The third set leaves the int8-safe
    regime, so its DP goes through the wavefront kernels instead of the team kernel."""
    name, rname = "fx2", "reads150"
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_score_")
    rec = os.path.join(tmp, "records.jsonl")
    M, m, O, E, P, F, z = score
    cmd = [CLI, "aln", "-S", "-M", str(M), "-m", str(m), "-O", str(O), "-E", str(E), "-P", str(P), "-F", str(F), "-z", str(z), "-o", os.path.join(tmp, "o.sam"), "-p",
           os.path.join(tmp, "p.sam"), "--records", rec, "--trace", ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [normalise(l) for l in open(rec).read().split("\n") if l.strip()]
    want = [normalise(l) for l in ac.golden_lines(name, rname + ".score_" + "_".join(str(x) for x in score))]
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if a != b]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\ngpu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])
    assert got != [normalise(l) for l in ac.golden_lines(name, rname)]        # the options really changed the results


def _run_cli(tmp, tag, name, rname, extra):
    w = ac.workdir(name)
    o = os.path.join(tmp, tag)
    r = subprocess.run([CLI, "aln", "-S", "-t", "4", "-o", o + ".sam", "-p", o + ".ori.sam", "--records", o + ".jsonl"] + extra +
                       [ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return o, r.stderr.decode()


@pytest.mark.parametrize("name,rname", [("fx2", "reads150"), ("fx3", "ragged")])
def test_gpu_cli_batch_boundaries_carry_the_draw_streams(name, rname):
    """A16: the three-stage pipeline over many batches (97 pairs each: raw text, rand() and random_r positions carried across every
    boundary) writes byte for byte the files of a single batch -- and those are the reference's (previous test)."""
    tmp = tempfile.mkdtemp(prefix="psvr_pipe_")
    one, _ = _run_cli(tmp, "one", name, rname, [])
    many, err = _run_cli(tmp, "many", name, rname, ["--batch", "97"])
    assert err.count("Processing ") >= 10
    for ext in (".sam", ".ori.sam", ".jsonl"):
        assert open(one + ext, "rb").read() == open(many + ext, "rb").read(), ext
    # the reference's own batching rule (a batch ends at 100 MB of bases, read_realignment.cpp:109,126), scaled down
    few, err = _run_cli(tmp, "few", name, rname, ["--batch-bases", "60000"])
    assert err.count("Processing ") >= 5
    assert open(one + ".sam", "rb").read() == open(few + ".sam", "rb").read()


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_gpu_cli_devices_split_every_batch_in_input_order(devices):
    """(e) in the product: `--devices` cuts every batch into contiguous blocks, one engine per entry (here all mapped to the box's one
    GPU), moves block d to where block d-1 stopped in the draw streams and gathers in order: the files must be those of one
    device -- for one big batch and for many small ones."""
    import json
    tmp = tempfile.mkdtemp(prefix="psvr_dev_")
    name, rname = "fx2", "reads150"
    one, _ = _run_cli(tmp, "one", name, rname, [])
    for tag, extra in (("split", []), ("split_batches", ["--batch", "211"])):
        got, err = _run_cli(tmp, tag, name, rname, ["--devices", devices] + extra)
        for ext in (".sam", ".ori.sam", ".jsonl"):
            assert open(one + ext, "rb").read() == open(got + ext, "rb").read(), (tag, ext)
        e2e = json.loads([l for l in err.split("\n") if "e2e_json" in l][-1].split("e2e_json ", 1)[1])
        assert e2e["devices"] == len(devices.split(",")) and e2e["rebase_iterations"] >= 1


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
    import synth
    ts.write_bam(bam, recs, [(l.split("SN:")[1].split("\t")[0], 250000000) for l in synth.header_text().split("\n") if l.startswith("@SQ")])
    # route 1: two commands
    fq = os.path.join(tmp, "x.fq")
    r = subprocess.run([CLI, "signal", "-N", "-D", "-H", os.path.join(tmp, "h1.sam"), "-S", os.path.join(tmp, "s1.txt"), bam], stdout=open(fq, "wb"), stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert os.path.getsize(fq) > 100000
    idx = os.path.join(ac.golden_dir("fx1"), "idx")
    outs = []
    for tag, reads, hdr in (("a", fq, os.path.join(tmp, "h1.sam")), ("b", bam, os.path.join(tmp, "h2.sam"))):
        o = os.path.join(tmp, tag)
        r = subprocess.run([CLI, "aln", "-S", "-D", "-N", "-o", o + ".sam", "-p", o + ".ori.sam", "--records", o + ".jsonl", idx, reads, hdr], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(o)
    for ext in (".sam", ".ori.sam", ".jsonl"):
        assert open(outs[0] + ext, "rb").read() == open(outs[1] + ext, "rb").read(), ext
    assert os.path.getsize(outs[0] + ".sam") > 10000
    assert open(os.path.join(tmp, "h1.sam"), "rb").read() == open(os.path.join(tmp, "h2.sam"), "rb").read()


@pytest.mark.parametrize("name,rname", [("fx2", "reads150"), ("fx2", "reads250"), ("fx3", "ragged"), ("fx3", "lower"), ("fx1", "anchor0"), ("fx5", "clamp0s"), ("fx5", "hicopy")])
def test_gpu_lane_per_pair_preparation_on_the_golden_sets(name, rname):
    """The engine prepares the reads of a round with one lane per pair (k_prep_pair) only when the round is large enough to fill the
    chip; PSVR_PREP_PAIR_MIN=1 sends these small sets down that path too: N draws in order, lower-case n (redone by the reference
    definition), ragged lengths, 250-base reads (nine packed words), and the mate-1 reads prepared again after mate 0's tie draws."""
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_gpu_")
    rec = os.path.join(tmp, "records.jsonl")
    cmd = [CLI, "aln", "-S", "-o", os.path.join(tmp, "out.sam"), "-p", os.path.join(tmp, "ori.sam"), "--records", rec, "--trace",
           ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PSVR_PREP_PAIR_MIN="1"))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = [l for l in open(rec).read().split("\n") if l.strip()]
    want = ac.golden_lines(name, rname)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\ngpu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])


@pytest.mark.parametrize("shrink", [4, 16])
def test_gpu_scratch_arena_growth_reruns_the_batch(shrink):
    """The overflow -> grow -> re-run path of every scratch arena on the GPU (PSVR_ARENA_SHRINK, see test_emu_aln): no stage of a
    round that overflowed may follow offsets of records that were never written, and the records are the reference's."""
    w = ac.workdir("fx2")
    tmp = tempfile.mkdtemp(prefix="psvr_gpu_")
    rec = os.path.join(tmp, "records.jsonl")
    cmd = [CLI, "aln", "-S", "-o", os.path.join(tmp, "out.sam"), "-p", os.path.join(tmp, "ori.sam"), "--records", rec, "--trace",
           os.path.join(ac.golden_dir("fx2"), "idx"), os.path.join(w, "reads150.fq"), os.path.join(w, "header.sam")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, PSVR_ARENA_SHRINK=str(shrink)))
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b"scratch arena overflow" in r.stderr
    got = [l for l in open(rec).read().split("\n") if l.strip()]
    want = ac.golden_lines("fx2", "reads150")
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d" % (len(bad), len(want), bad[0])


def test_gpu_aln_then_sort_against_the_reference_text():
    """f3, the two steps panSVR_run.sh runs after `aln` (`samtools sort` + `samtools index`, panSVR_run.sh:53-54; the reference
    holds no sorter of its own, so the ORDER is unpinned and checked against samtools' documented key: reference id, position,
    strand, ties in input order): `panSVR aln` (BAM, the default) then `panSVR sort` on a golden set.  Decoded with the
    independent reader of tests/bam_reader.py, the sorted file must hold exactly the records of the REFERENCE's own SAM text
    (tests/golden/fx1/reads150.sam.gz) in that order, and the .bai must answer region queries with what a scan finds."""
    import gzip
    import struct
    import bam_reader
    name, rname = "fx1", "reads150"
    w = ac.workdir(name)
    tmp = tempfile.mkdtemp(prefix="psvr_sort_")
    out, srt = os.path.join(tmp, "out.bam"), os.path.join(tmp, "sorted.bam")
    r = subprocess.run([CLI, "aln", "-o", out, "-p", os.path.join(tmp, "ori.bam"), ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    r = subprocess.run([CLI, "sort", "-t", "4", "-o", srt, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    with gzip.open(os.path.join(ac.golden_dir(name), rname + ".sam.gz"), "rt") as f:
        ref_lines = [l.split("\t") for l in f.read().split("\n") if l and not l.startswith("@")]
    text, refs, recs = bam_reader.read_bam(srt)
    assert "SO:coordinate" in text.split("\n")[0] and bam_reader.check_bgzf(srt) >= 1
    tid_of = {n: i for i, (n, _) in enumerate(refs)}

    def key(f):
        return (tid_of[f[2]] if f[2] != "*" else 1 << 40, int(f[3]) - 1, int(f[1]) & 16)
    want = sorted(ref_lines, key=key)                                    # stable: ties keep the reference's (= input) order
    assert len(recs) == len(want) > 3000
    for a, b in zip(want, recs):
        assert a == b, "\nref: %s\nbam: %s" % ("\t".join(a), "\t".join(b))
    # the index: every reference's records through the bins' chunks == a scan
    bai = open(srt + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01" and struct.unpack_from("<i", bai, 4)[0] == len(refs)
    n_mapped = sum(1 for f in recs if f[2] != "*")
    off, n_in_chunks = 8, 0
    for _ in range(len(refs)):
        n_bin = struct.unpack_from("<i", bai, off)[0]
        off += 4
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, off)
            off += 8 + 16 * n_chunk
            n_in_chunks += n_chunk if b != 37450 else 0
        n_intv = struct.unpack_from("<i", bai, off)[0]
        off += 4 + 8 * n_intv
    assert n_in_chunks >= 1 and n_mapped > 3000
