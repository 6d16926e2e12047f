#!/usr/bin/env python3
"""Regenerates tests/golden/fused/*.sam.gz: what the REFERENCE's two steps make of a BAM -- `fc_signal`'s per-pair function
(READ_SIGNAL_HANDLER::all_signal_records_read_pair through oracle/_ref/ref_signal) writes the FASTQ, the reference's `fc_aln`
objects (oracle/_ref/ref_aln -t 1 -S) align it -- for the BAM tests/test_fused_signal.bam_of builds from a golden read set.
`panSVR aln x.bam` (the fused route of this repo) must write these bytes.  BUILD CONTAINER ONLY.
The STAT_ numbers are the ones `panSVR signal` reports for the file (the reference takes them from htslib's file layer)."""
import gzip
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import aln_common as ac  # noqa: E402
import bam_reader  # noqa: E402
import index_fixture  # noqa: E402
import test_fused_signal as tf  # noqa: E402
import test_signal as ts  # noqa: E402

CASES = [("fx1", "reads150", 2000, ["-D"]), ("fx2", "reads150", 1500, ["-D"])]


def main():
    out_dir = os.path.join(HERE, "fused")
    os.makedirs(out_dir, exist_ok=True)
    for name, rname, n_pairs, flags in CASES:
        tmp = tempfile.mkdtemp(prefix="psvr_fusedg_")
        bam = os.path.join(tmp, "in.bam")
        tf.bam_of(name, rname, n_pairs, bam)
        text, _, sam = bam_reader.read_bam(bam, check_bin=False)
        open(os.path.join(tmp, "in.sam"), "w").write("\n".join("\t".join(f) for f in sam) + "\n")
        open(os.path.join(tmp, "hdr.sam"), "w").write(text)
        subprocess.run([ts.CLI, "signal", "-N"] + flags + ["-H", os.path.join(tmp, "h.sam"), "-S", os.path.join(tmp, "s.txt"), bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        st = open(os.path.join(tmp, "s.txt")).read().split("\n")[0].split("_")
        stat = "%d,%d,%d,%d" % (int(st[1]), int(st[4]), (int(st[2]) + int(st[3])) // 2, int(st[5]))
        fq = os.path.join(tmp, "sig.fq")
        with open(fq, "wb") as f:
            subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_signal"), os.path.join(tmp, "in.sam"), os.path.join(tmp, "hdr.sam"), "--stat", stat] + flags,
                           stdout=f, stderr=subprocess.DEVNULL, check=True)
        # the reference's loader wants the dense first-level table: expanded from the committed fixture
        idx = os.path.join(tmp, "idx")
        os.makedirs(idx)
        src = ac.index_dir(name)
        for fn in index_fixture.SMALL:
            os.symlink(os.path.join(src, fn), os.path.join(idx, fn))
        index_fixture.expand_hash(src).tofile(os.path.join(idx, "unipath_g.hash"))
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_aln"), "-t", "1", "-S", "-o", os.path.join(tmp, "o.sam"), "-p", os.path.join(tmp, "p.sam"), idx, fq, os.path.join(tmp, "h.sam"), "--quiet"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        os.remove(os.path.join(idx, "unipath_g.hash"))
        for src_fn, ext in (("o.sam", ".sam.gz"), ("p.sam", ".ori.sam.gz")):
            with open(os.path.join(out_dir, "%s_%s%s" % (name, rname, ext)), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
                f.write(open(os.path.join(tmp, src_fn), "rb").read())
        print(name, rname, n_pairs, "pairs, stat", stat, os.path.getsize(os.path.join(tmp, "o.sam")), "B sam")


if __name__ == "__main__":
    main()
