"""-m gpu: BASELINE configs[2] at its STATED size -- a 30x-WGS-scale signal set, 25 M pairs (50 M reads of 150 bp), against the
10 k-SV anchor reference, streamed through the drop-in command on one MI355X.

Nothing of that size is stored: tests/tools/gen_signal_fastq (C++, every pair a pure function of the seed and its number) writes
the 21 GB of FASTQ text into the command's stdin, and the command writes its two SAM files into FIFOs that tests/tools/sam_check
reads.  Checked:
  * EVERY record of the main file: field / tag layout of output_BAM (read_realignment.cpp:479-536), SEQ / QUAL length, a CIGAR that
    consumes the read, FLAG / POS / MAPQ ranges, input order kept across all 75 batches (the reference's 100 MB-of-bases rule);
  * prefix parity: the first 200 000 pairs -- generated again on their own -- through the reference's own objects
    (`oracle/_ref/ref_aln -t 1 -S`) give, byte for byte, the head of both files;
  * the batch count, the pair count and a steady HBM footprint (the buffers of batch 1 serve batch 75).
PSVR_CFG3_PAIRS overrides the size (the default is the configuration's)."""
import json
import os
import shutil
import subprocess
import tempfile
import time

import pytest

import aln_common as ac

pytestmark = pytest.mark.gpu
CLI = os.path.join(ac.ROOT, "pansvr_amd", "bin", "panSVR")
REF = os.path.join(ac.ROOT, "oracle", "_ref", "ref_aln")
TOOLS = os.path.join(ac.HERE, "tools")
N_PAIRS = int(os.environ.get("PSVR_CFG3_PAIRS", "25000000"))
N_ANCHORS, ANCHOR_SEED, READ_SEED, PREFIX = 10000, 11, 17, 200000


def build_tools():
    out = {}
    bindir = tempfile.mkdtemp(prefix="psvr_cfg3_bin_")                 # not under /dev/shm: it may be mounted noexec
    for name in ("gen_signal_fastq", "sam_check"):
        exe = os.path.join(bindir, name)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(TOOLS, name + ".cpp"), "-lpthread"])
        out[name] = exe
    return out


def body_of(path):
    """A SAM file without its header lines."""
    with open(path, "rb") as f:
        data = f.read()
    at = 0
    while data[at:at + 1] == b"@":
        at = data.index(b"\n", at) + 1
    return data[at:]


@pytest.mark.timeout(1100)
def test_configs2_at_its_stated_size_through_a_pipe():
    assert os.path.exists(REF), "oracle/_ref/ref_aln is missing: build it with `make -C oracle` where /root/reference exists"
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (8 << 30) else None
    tmp = tempfile.mkdtemp(prefix="psvr_cfg3_", dir=shm)
    procs = []
    try:
        t = build_tools()
        ncore = os.cpu_count() or 1
        fa, idx, hdr = os.path.join(tmp, "anchors.fa"), os.path.join(tmp, "idx"), os.path.join(tmp, "header.sam")
        with open(fa, "wb") as f:
            subprocess.check_call([t["gen_signal_fastq"], "anchors", str(N_ANCHORS), str(ANCHOR_SEED)], stdout=f)
        os.makedirs(idx)
        subprocess.check_call([CLI, "index", "-k", "22", fa, idx + "/"], stderr=subprocess.DEVNULL)     # dense first level: the reference's loader reads it
        with open(hdr, "w") as f:
            f.write("@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n")
        # ---- the prefix through the reference's objects
        pre = os.path.join(tmp, "prefix.fq")
        n_pre = min(PREFIX, N_PAIRS)
        with open(pre, "wb") as f:
            subprocess.check_call([t["gen_signal_fastq"], "reads", str(N_ANCHORS), str(ANCHOR_SEED), str(n_pre), str(READ_SEED), str(min(8, ncore))], stdout=f)
        r = subprocess.run([REF, "-t", "1", "-S", "-o", os.path.join(tmp, "ref.sam"), "-p", os.path.join(tmp, "ref.ori.sam"), idx, pre, hdr, "--quiet"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-1000:]
        want = {k: body_of(os.path.join(tmp, "ref" + k)) for k in (".sam", ".ori.sam")}
        assert len(want[".sam"]) > 10 ** 7
        os.remove(pre), os.remove(os.path.join(tmp, "ref.sam")), os.remove(os.path.join(tmp, "ref.ori.sam"))
        # ---- the whole set through the command: generator -> stdin, both outputs -> FIFOs -> checkers
        fifo = {k: os.path.join(tmp, "out" + k) for k in (".sam", ".ori.sam")}
        head = {k: os.path.join(tmp, "head" + k) for k in fifo}
        chk = {}
        for k in fifo:
            os.mkfifo(fifo[k])
            chk[k] = subprocess.Popen([t["sam_check"], fifo[k], "150", "--head-bytes", str(len(want[k])), "--head-file", head[k]], stdout=subprocess.PIPE)
            procs.append(chk[k])
        gen = subprocess.Popen([t["gen_signal_fastq"], "reads", str(N_ANCHORS), str(ANCHOR_SEED), str(N_PAIRS), str(READ_SEED), str(max(2, min(6, ncore // 3)))],
                               stdout=subprocess.PIPE)
        procs.append(gen)
        t0 = time.time()
        cli = subprocess.Popen([CLI, "aln", "-S", "-t", str(max(2, min(10, ncore - 6))), "-o", fifo[".sam"], "-p", fifo[".ori.sam"], idx, "-", hdr],
                               stdin=gen.stdout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(cli)
        gen.stdout.close()
        err = cli.communicate()[1].decode()
        wall = time.time() - t0
        assert cli.returncode == 0, err[-2000:]
        assert gen.wait() == 0
        rep = {}
        for k in fifo:
            out = chk[k].communicate()[0].decode()
            rep[k] = json.loads(out)
        j = json.loads([l for l in err.split("\n") if "e2e_json" in l][-1].split("e2e_json ", 1)[1])
        print("cfg3: %d pairs in %.1f s wall = %.2f M reads/s end to end (generator and checkers included); %s; main %s; ori %s"
              % (N_PAIRS, wall, 2 * N_PAIRS / wall / 1e6, json.dumps(j), json.dumps(rep[".sam"]), json.dumps(rep[".ori.sam"])))
        assert j["pairs"] == N_PAIRS and j["batches"] == (N_PAIRS * 300 + 10 ** 8 - 1) // 10 ** 8
        assert j["hbm_used_last"] <= j["hbm_used_first"] * 1.02, j
        assert j["dropped"] == 0
        # every record of the main file holds the invariants; both files start with what the reference's objects wrote
        assert rep[".sam"]["violation"] == "" and rep[".sam"]["records"] == rep[".sam"]["checked"] > N_PAIRS       # > half of the reads are written
        assert rep[".sam"]["last_pair"] == N_PAIRS - 1 or rep[".sam"]["last_pair"] > N_PAIRS - 100
        for k in fifo:
            with open(head[k], "rb") as f:
                got = f.read()
            assert got == want[k], "the head of out%s differs from the reference's file" % k
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(tmp, ignore_errors=True)
