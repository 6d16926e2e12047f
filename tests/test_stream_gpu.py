"""-m gpu: BASELINE configs[2] in shape -- a multi-million-pair signal set streamed through `panSVR aln` in batches, one MI355X.

What the reference does (read_realignment.cpp:24,109,121-152): a batch ends at 2 M pairs OR 100 MB of bases, whichever comes first;
for 150 bp reads the second limit binds, at 333 334 pairs.  Both are exercised on one 6 M-pair input:
  A. --batch 2000000 with the base limit lifted: 3 batches of 2 M pairs (the batch size SURVEY names);
  B. the reference's own rule (default options): 18 batches of 333 334 pairs.
Checked: A and B write identical files (the rand()/random_r positions, the carried text and the three pipeline slots are
batch-size invariant); the first 400 000 pairs of B -- across its first batch boundary -- are byte for byte what the
reference's objects write for that prefix (`ref_aln -t 1 -S -R 400000`); CIGAR / flag / tag invariants over a sample of all
records; the HBM footprint after the last batch is the footprint after the first."""
import hashlib
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

import aln_common as ac
import bench_data

pytestmark = pytest.mark.gpu
CLI = os.path.join(ac.ROOT, "pansvr_amd", "bin", "panSVR")
REF = os.path.join(ac.ROOT, "oracle", "_ref", "ref_aln")
PARTS, PART_PAIRS, PREFIX = 3, 2000000, 400000


def md5_of(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def e2e_of(err):
    return json.loads([l for l in err.split("\n") if "e2e_json" in l][-1].split("e2e_json ", 1)[1])


@pytest.mark.timeout(1500)
def test_six_million_pairs_streamed_in_batches():
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (40 << 30) else None
    if shm is None:
        pytest.skip("needs 40 GB of RAM-backed storage for the input and the output files")
    tmp = tempfile.mkdtemp(prefix="psvr_stream_", dir=shm)
    try:
        anc = bench_data.make_anchors(4000, seed=11)
        ix = bench_data.build_index_cli(anc, dense=os.path.exists(REF))
        dense = ix.pop("hash")
        bench_data.write_index_dir(ix, os.path.join(tmp, "idx"), dense_hash=dense)
        del dense, ix
        fq = os.path.join(tmp, "reads.fq")
        ncore = os.cpu_count() or 1
        for part in range(PARTS):                                            # 6 M pairs, 5 GB of text
            bases, base_off, ori, isize = bench_data.make_reads(anc, PART_PAIRS, seed=17 + part)
            bench_data.write_fastq(fq, bases, base_off, ori, isize, procs=min(16, ncore), name_base=part * PART_PAIRS, append=part > 0,
                                   stat=(150, 200, 400, 600) if part == 0 else None)
            del bases, base_off, ori, isize
        with open(os.path.join(tmp, "header.sam"), "w") as f:
            f.write("@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n")
        base = [os.path.join(tmp, "idx"), fq, os.path.join(tmp, "header.sam")]
        nt = str(min(16, ncore))

        def run(tag, extra):
            o = os.path.join(tmp, tag)
            r = subprocess.run([CLI, "aln", "-S", "-t", nt, "-o", o + ".sam", "-p", o + ".ori.sam"] + extra + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            return o, e2e_of(r.stderr.decode()), r.stderr.decode()

        a, ja, _ = run("a", ["--batch", str(PART_PAIRS), "--batch-bases", str(10 ** 12)])
        assert ja["pairs"] == PARTS * PART_PAIRS and ja["batches"] == PARTS
        assert ja["hbm_used_last"] <= ja["hbm_used_first"] * 1.02, ja           # steady footprint: the slots and arenas of batch 1 serve batch 3
        sums_a = (md5_of(a + ".sam"), md5_of(a + ".ori.sam"))
        size_a = os.path.getsize(a + ".sam")
        os.remove(a + ".sam"), os.remove(a + ".ori.sam")
        b, jb, err_b = run("b", [])
        assert jb["batches"] == (PARTS * PART_PAIRS * 300 + 10 ** 8 - 1) // 10 ** 8 == 18
        assert "Processing 333334 reads, at block ID 0" in err_b               # the reference's progress line, its batch size
        assert jb["hbm_used_last"] <= jb["hbm_used_first"] * 1.02, jb
        assert (md5_of(b + ".sam"), md5_of(b + ".ori.sam")) == sums_a and os.path.getsize(b + ".sam") == size_a
        print("stream: A %s  B %s" % (json.dumps(ja), json.dumps(jb)))
        # the prefix across B's first batch boundary (pair 333 334) against the reference's own objects (oracle/_ref travels to the GPU
        # box with the snapshot: a box without it has no checker, which is a failure of the test, not a reason to pass)
        assert os.path.exists(REF), "oracle/_ref/ref_aln is missing: build it with `make -C oracle` where /root/reference exists"
        if True:
            r = subprocess.run([REF, "-t", "1", "-S", "-R", str(PREFIX), "-o", os.path.join(tmp, "ref.sam"), "-p", os.path.join(tmp, "ref.ori.sam")] + base + ["--quiet"],
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()[-1000:]
            for ext in (".sam", ".ori.sam"):
                want = open(os.path.join(tmp, "ref" + ext), "rb").read()
                with open(b + ext, "rb") as f:
                    got = f.read(len(want))
                assert got == want, "prefix of %s differs from the reference's file" % ext
                assert len(want) > 10 ** 6
        # invariants over a sample of ALL records of the main file (every 40th line)
        consumes_q = set("MIS=X")
        n_seen = 0
        with open(b + ".sam", "rb") as f:
            for i, line in enumerate(f):
                if line[:1] == b"@" or i % 40:
                    continue
                fld = line.rstrip(b"\n").split(b"\t")
                assert len(fld) >= 14 and len(fld[9]) == len(fld[10]) == 150
                flag, pos, mapq, cigar = int(fld[1]), int(fld[3]), int(fld[4]), fld[5].decode()
                assert pos >= 1 and 0 <= mapq <= 40 and not flag & ~(0x40 | 0x10 | 0x8)
                num, q = 0, 0
                for ch in cigar:
                    if ch.isdigit():
                        num = num * 10 + int(ch)
                    else:
                        assert ch in "MIDNSHP=X" and num > 0
                        q += num if ch in consumes_q else 0
                        num = 0
                assert q == 150, cigar
                tags = [t[:5] for t in fld[11:]]
                assert tags[0] == b"AS:i:" and b"OS:i:" in tags and b"OA:Z:" in tags and tags[-1] == b"RC:Z:"
                n_seen += 1
        assert n_seen > 50000
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
