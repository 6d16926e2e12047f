#!/usr/bin/env python3
"""End-to-end wall of the drop-in CLI (FASTQ in -> BAM/SAM out) on a synthetic cfg-2 style input.
usage: python tools/cli_e2e.py [pairs] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench_data  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
threads = sys.argv[2] if len(sys.argv) > 2 else "8"
shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
tmp = tempfile.mkdtemp(prefix="psvr_cli_", dir=shm)
anc = bench_data.make_anchors(10000, seed=11)
ix = bench_data.build_index(anc, dense=True)
h = ix.pop("hash")
bench_data.write_index_dir(ix, os.path.join(tmp, "idx"), dense_hash=h)
bases, base_off, ori, isize = bench_data.make_reads(anc, pairs, seed=13)
bench_data.write_fastq(os.path.join(tmp, "r.fq"), bases, base_off, ori, isize)
open(os.path.join(tmp, "h.sam"), "w").write("@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n")
cli = os.path.join(ROOT, "pansvr_amd", "bin", "panSVR")
for mode, ext in (["-S"], "sam"), ([], "bam"):
    t = time.time()
    r = subprocess.run([cli, "aln", "-t", threads] + mode + ["-o", os.path.join(tmp, "o." + ext), "-p", os.path.join(tmp, "p." + ext), os.path.join(tmp, "idx"), os.path.join(tmp, "r.fq"),
                        os.path.join(tmp, "h.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    dt = time.time() - t
    err = r.stderr.decode()
    print(ext, "rc", r.returncode, "wall %.2f s" % dt, "-> %.0f reads/s incl. index load" % (2 * pairs / dt), "| out MB", os.path.getsize(os.path.join(tmp, "o." + ext)) >> 20)
    print("   ", " | ".join(l for l in err.split("\n") if "sec" in l or "wall:" in l)[:400])
import shutil
shutil.rmtree(tmp, ignore_errors=True)
