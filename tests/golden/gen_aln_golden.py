#!/usr/bin/env python3
"""Regenerates the end-to-end golden fixtures of the `aln` path.  BUILD CONTAINER ONLY: it runs the
reference's own programs compiled under oracle/_ref/ (deBGA index builder + ref_aln, the reference
aligner objects behind oracle/ref_harness/ref_aln_main.cpp).

For every data set in DATASETS: synthesize anchors/reads (tests/synth.py, seeded), build the index
with the reference deBGA (-k 22), run ref_aln --trace, and commit
    tests/golden/<name>/idx/          compact index (tests/index_fixture.py)
    tests/golden/<name>/<reads>.jsonl.gz   one record per read pair, as printed by the reference objects
    tests/golden/<name>/<reads>.sam.gz, <reads>.ori.sam.gz   the two output files of `fc_aln -t 1 -S` (the reference's own
                                           output_BAM / output_ori_bam -> sam_parse1 -> sam_format1 text)
"""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import index_fixture  # noqa: E402
import datasets  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
# non-default scoring options (-M -m -O -E -P -F -z) run through the reference objects as well: (data set, reads, options)
SCORE_SETS = [("fx2", "reads150", s) for s in ((3, 9, 12, 2, 24, 1, 200), (1, 4, 6, 1, 20, 0, 50), (2, 30, 40, 3, 60, 2, 400))]


# `-Q` / --not-ori (read_realignment.cpp:485: an ORIGINAL primary is not written to the main file): (data set, reads).  fx2 holds
# unmapped and full-score originals and repeats whose original alignment outscores the new one; fx1 is the plain set.
NOT_ORI_SETS = [("fx1", "reads150"), ("fx2", "reads150")]


def score_tag(s):
    return "score_" + "_".join(str(x) for x in s)


def main(names, only_not_ori=False):
    for name in names:
        ds = datasets.DATASETS[name]
        work = os.environ.get("PSVR_GOLDEN_WORK", tempfile.mkdtemp(prefix="psvr_" + name))
        os.makedirs(work, exist_ok=True)
        datasets.materialize(name, work)
        idx = os.path.join(work, "idx")
        if not os.path.exists(os.path.join(idx, "unipath_g.hash")):
            os.makedirs(idx, exist_ok=True)
            subprocess.check_call([os.path.join(REF, "deBGA"), "index", "-k", "22", os.path.join(work, "anchors.fa"), idx + "/"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = os.path.join(HERE, name)
        os.makedirs(out, exist_ok=True)
        if only_not_ori:
            pass
        elif ds.get("index") == "sha256":
            # too large to commit: the compact form's SHA-256 per file; tests rebuild the index with `panSVR index` and compare
            import hashlib
            cdir = os.path.join(work, "idx_compact")
            index_fixture.compact(idx, cdir)
            with open(os.path.join(out, "idx.sha256"), "w") as f:
                for fn in sorted(os.listdir(cdir)):
                    f.write("%s  %s\n" % (hashlib.sha256(open(os.path.join(cdir, fn), "rb").read()).hexdigest(), fn))
        else:
            index_fixture.compact(idx, os.path.join(out, "idx"))
        for rname in ([] if only_not_ori else ds["reads"]):
            sam, ori, rec = (os.path.join(work, rname + e) for e in (".ref.sam", ".ref.ori.sam", ".ref.jsonl"))
            subprocess.run([os.path.join(REF, "ref_aln"), "-t", "1", "-S", "-o", sam, "-p", ori, idx, os.path.join(work, rname + ".fq"), os.path.join(work, "header.sam"),
                            "--trace", "--records", rec], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            lines = [l for l in open(rec).read().split("\n") if l.startswith("{") or l.startswith(" {")]

            def put(path, data):
                with open(path, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:   # mtime=0: reproducible bytes
                    f.write(data)
            put(os.path.join(out, rname + ".jsonl.gz"), ("\n".join(l.strip() for l in lines) + "\n").encode())
            put(os.path.join(out, rname + ".sam.gz"), open(sam, "rb").read())
            put(os.path.join(out, rname + ".ori.sam.gz"), open(ori, "rb").read())
            print(name, rname, len(lines), "pairs", os.path.getsize(sam), "B sam", os.path.getsize(ori), "B ori sam")
        for sname, rname in NOT_ORI_SETS:
            if sname != name:
                continue
            sam, ori = (os.path.join(work, rname + e) for e in (".refQ.sam", ".refQ.ori.sam"))
            subprocess.run([os.path.join(REF, "ref_aln"), "-t", "1", "-S", "-Q", "-o", sam, "-p", ori, idx, os.path.join(work, rname + ".fq"), os.path.join(work, "header.sam")],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            for src, ext in ((sam, ".notori.sam.gz"), (ori, ".notori.ori.sam.gz")):
                with open(os.path.join(out, rname + ext), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
                    f.write(open(src, "rb").read())
            print(name, rname, "-Q", os.path.getsize(sam), "B sam", os.path.getsize(ori), "B ori sam")
        for sname, rname, sc in SCORE_SETS:
            if sname != name or only_not_ori:
                continue
            rec = os.path.join(work, rname + "." + score_tag(sc) + ".jsonl")
            M, m, O, E, P, F, z = sc
            subprocess.run([os.path.join(REF, "ref_aln"), "-t", "1", "-M", str(M), "-m", str(m), "-O", str(O), "-E", str(E), "-P", str(P), "-F", str(F), "-z", str(z),
                            idx, os.path.join(work, rname + ".fq"), os.path.join(work, "header.sam"), "--trace", "--records", rec],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            lines = [l for l in open(rec).read().split("\n") if l.startswith("{")]
            with open(os.path.join(out, rname + "." + score_tag(sc) + ".jsonl.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
                f.write(("\n".join(l.strip() for l in lines) + "\n").encode())
            print(name, rname, score_tag(sc), len(lines), "pairs")


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if a != "--only-not-ori"]      # --only-not-ori: just the -Q files (the other fixtures stay as committed)
    main(argv or list(datasets.DATASETS), only_not_ori="--only-not-ori" in sys.argv[1:])
