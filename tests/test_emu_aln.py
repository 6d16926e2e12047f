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
    out = subprocess.run([emu, ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), "--trace"],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    got = [l for l in out.split("\n") if l.strip()]
    want = ac.golden_lines(name, rname)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d:\nref: %s\nemu: %s" % (len(bad), len(want), bad[0], want[bad[0]], got[bad[0]])


@pytest.mark.parametrize("shrink", [4, 16])
def test_scratch_arena_growth_reruns_the_batch(emu, shrink):
    """PSVR_ARENA_SHRINK starts the engine with a fraction of its scratch arenas: every arena (MEMs, seeds, pieces, DP descriptors,
    candidates, CIGAR words) overflows, grows fourfold and the batch runs again -- several times for the smaller start -- and the
    records are the reference's all the same (engine_core.h grow_and_rerun; the stages of a round that overflowed must not walk
    records that were never written)."""
    w = ac.workdir("fx2")
    env = dict(os.environ, PSVR_ARENA_SHRINK=str(shrink))
    r = subprocess.run([emu, os.path.join(ac.golden_dir("fx2"), "idx"), os.path.join(w, "reads150.fq"), os.path.join(w, "header.sam"), "--trace"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True, env=env)
    assert b"scratch arena overflow" in r.stderr
    got = [l for l in r.stdout.decode().split("\n") if l.strip()]
    want = ac.golden_lines("fx2", "reads150")
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if normalise(a) != normalise(b)]
    assert not bad, "%d/%d pairs differ; first %d" % (len(bad), len(want), bad[0])


@pytest.mark.parametrize("variant", ["crlf", "no_final_newline", "tiny_batches", "truncated_pair", "gz", "pipe", "pipe_tiny_batches", "gz_pipe", "threads"])
def test_fastq_reader_edge_cases(emu, variant):
    """fastq_batch.h's batch reader (memory-mapped or streamed text, line index built on threads, nothing copied but the bases) on
    awkward inputs: CR LF line ends, a last line without a newline, batches of 7 pairs, a trailing incomplete pair (ignored, like
    the reference's read loop), a gzip file, a pipe (stream mode: text carried across batches), a gzip stream on a pipe, several index threads.  The
    records must be those of the plain file."""
    w = ac.workdir("fx1")
    text = open(os.path.join(w, "reads150.fq")).read()
    n_keep = 300
    lines = text.split("\n")[:8 * n_keep]
    extra = []
    if variant == "crlf":
        data = "\r\n".join(lines) + "\r\n"
    elif variant == "no_final_newline":
        data = "\n".join(lines)
    elif variant == "tiny_batches":
        data = "\n".join(lines) + "\n"
        extra = ["--batch", "7"]
    elif variant == "truncated_pair":
        data = "\n".join(lines + lines[:4]) + "\n"          # one more read without its mate
    else:
        data = "\n".join(lines) + "\n"
        if variant == "pipe_tiny_batches":
            extra = ["--batch", "11"]
        if variant == "threads":
            extra = ["--threads", "5", "--batch", "64"]
    path = os.path.join(w, "edge_%s.fq" % variant)
    stdin = None
    if variant in ("gz", "gz_pipe"):
        import gzip
        path += ".gz"
        with gzip.open(path, "wb") as f:
            f.write(data.encode())
    else:
        open(path, "w", newline="").write(data)
    if variant.startswith("pipe") or variant == "gz_pipe":                    # (a gzip stream on a pipe is decoded like a gzip file: the reference's xzopen)
        stdin, path = open(path, "rb"), "-"
        stdin = subprocess.Popen(["cat"], stdin=stdin, stdout=subprocess.PIPE).stdout      # a real pipe, not a seekable file
    out = subprocess.run([emu, os.path.join(ac.golden_dir("fx1"), "idx"), path, os.path.join(w, "header.sam"), "--trace"] + extra,
                         stdin=stdin, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    got = [normalise(l) for l in out.split("\n") if l.strip()]
    want = [normalise(l) for l in ac.golden_lines("fx1", "reads150")[:n_keep]]
    assert got == want


@pytest.mark.parametrize("name,rname", [c for c in CASES if c in (("fx1", "reads150"), ("fx3", "lower"), ("fx2", "reads150"))])
def test_rebase_equals_a_run_from_the_new_position(emu, name, rname):
    """A shard that ran from one place in the rand() / random_r streams and is moved to another (psvr_engine_rebase: what a rank of the
    multi-GPU path does when the ranks before it have reported their draws) must hold exactly the records -- and end exactly where -- a
    run from the new place does.  Several distances, so that pairs with N bases meet residues that select the variant slot they carry
    already as well as other ones (the adoption rules of engine_core.h), and tie draws fall on other values.  (The set that samples
    positions with random_r, fx5-hicopy, passes too: four minutes on the CPU, so it is not in this list.)"""
    w = ac.workdir(name)
    base = [emu, ac.index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam"), "--trace"]

    def run(extra):
        r = subprocess.run(base + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        end = [l for l in r.stderr.decode().split("\n") if "stream_end" in l][-1]
        return [normalise(l) for l in r.stdout.decode().split("\n") if l.strip()], end

    places = ((3, 0, 0), (2 + 1000, 0, 0), (2 + 12345, 7, 5), (2 + 400000, 0, 3))
    if name == "fx2":                                  # (the tie-heavy set takes its rounds: two places)
        places = places[1:3]
    for g, h0, h1 in places:
        pos = "%d,%d,%d" % (g, h0, h1)
        want, want_end = run(["--stream-pos", pos])
        got, got_end = run(["--stream-pos", pos, "--rebase-from", "2,0,0"])
        assert got_end == want_end, (pos, got_end, want_end)
        bad = [i for i, (a, b) in enumerate(zip(want, got)) if a != b]
        assert len(got) == len(want) and not bad, "%s: %d pairs differ after the rebase, first %d:\nrun:    %s\nrebase: %s" % (pos, len(bad), bad[0], want[bad[0]], got[bad[0]])
