"""The record formatter's fast paths (sam_emit.h: digits written in pairs, SEQ / QUAL sixteen bytes at a time, the second file's comment
sections scanned without sscanf / strstr) and the stages' worker pool (worker_pool.h) against plain restatements on generated and mutated
inputs: tests/tools/emit_check.cpp.  The golden record files (test_sam_golden.py) pin the same code on the reference's own output."""
import os
import subprocess
import tempfile

import aln_common as ac


def test_fast_paths_agree_with_their_restatements():
    exe = os.path.join(tempfile.mkdtemp(prefix="psvr_emit_"), "emit_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ac.HERE, "tools", "emit_check.cpp"), "-lz", "-lpthread"])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-2000:]
    assert " 0 differ" in r.stdout.decode()
