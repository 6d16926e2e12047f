"""The one-lane-per-block DEFLATE encoder behind psvr_bgzf_compress (pansvr_amd/csrc/deflate_device.h), compiled for the host: every
block it writes must inflate, with zlib, to the bytes it was given -- synthetic buffers at the corners of the format (empty input, a
single byte, incompressible bytes -> stored blocks, length-258 matches, a match at distance 32768) and the BAM records of a golden read
set -- at three hash-table sizes."""
import os
import subprocess
import tempfile

import pytest

import aln_common as ac
from test_emu_aln import EMU


@pytest.fixture(scope="module")
def checker():
    exe = os.path.join(tempfile.mkdtemp(prefix="psvr_deflate_"), "deflate_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ac.HERE, "tools", "deflate_check.cpp"), "-lz"])
    return exe


@pytest.mark.parametrize("hbits", [8, 10, 15])
def test_every_block_inflates_to_its_input(checker, hbits):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ac.HERE, "emu")])
    w = ac.workdir("fx2")
    tmp = tempfile.mkdtemp(prefix="psvr_deflate_")
    rec = os.path.join(tmp, "records.bam")
    r = subprocess.run([EMU, ac.index_dir("fx2"), os.path.join(w, "reads150.fq"), os.path.join(w, "header.sam"), "--no-records", "--sam", os.path.join(tmp, "o.sam"),
                        "--ori-sam", os.path.join(tmp, "p.sam"), "--bam-records", rec], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1000:]
    r = subprocess.run([checker, str(hbits), rec, os.path.join(w, "reads150.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    ratio = float([l for l in r.stdout.decode().split("\n") if "records.bam" in l][0].split("ratio")[1].split()[0])
    assert ratio > 1.7, r.stdout.decode()
