#!/usr/bin/env python3
"""Regenerates tests/golden/signal/*.fq.gz: the FASTQ the REFERENCE's own READ_SIGNAL_HANDLER::all_signal_records_read_pair
(getSignalRead.cpp, through oracle/_ref/ref_signal) writes for the seeded BAM records of tests/test_signal.make_pairs.
BUILD CONTAINER ONLY.  The insert-size / read-length numbers of the STAT_ field are the ones `panSVR signal` reports for the
same file (the reference takes them from sampling_analysis_stat, which needs htslib's file layer)."""
import gzip
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import bam_reader  # noqa: E402
import test_signal as ts  # noqa: E402

CASES = [([], 20240), (["-D"], 20241), (["-U"], 20241), (["-D", "-U", "-I", "22"], 20244)]


def case_name(flags, seed):
    return "pairs%d%s" % (seed, "".join(f.replace("-", "_") for f in flags))


def main():
    out_dir = os.path.join(HERE, "signal")
    os.makedirs(out_dir, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="psvr_sig_")
    for flags, seed in CASES:
        recs, refs = ts.make_pairs(seed, 600)
        bam = os.path.join(tmp, "in.bam")
        ts.write_bam(bam, recs, refs)
        text, _, sam = bam_reader.read_bam(bam, check_bin=False)
        open(os.path.join(tmp, "in.sam"), "w").write("\n".join("\t".join(f) for f in sam) + "\n")
        open(os.path.join(tmp, "hdr.sam"), "w").write(text)
        subprocess.run([ts.CLI, "signal", "-N"] + flags + ["-H", os.path.join(tmp, "h.sam"), "-S", os.path.join(tmp, "s.txt"), bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        st = open(os.path.join(tmp, "s.txt")).read().split("\n")[0].split("_")
        stat = "%d,%d,%d,%d" % (int(st[1]), int(st[4]), (int(st[2]) + int(st[3])) // 2, int(st[5]))
        res = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_signal"), os.path.join(tmp, "in.sam"), os.path.join(tmp, "hdr.sam"), "--stat", stat] + flags,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        with open(os.path.join(out_dir, case_name(flags, seed) + ".fq.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
            f.write(res)
        print(case_name(flags, seed), res.count(b"\n") // 8, "pairs, stat", stat)


if __name__ == "__main__":
    main()
