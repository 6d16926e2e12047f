"""Shared helpers for the `aln` parity tests: materialise seeded inputs, run the oracle executable,
load the committed reference records."""
import gzip
import json
import os
import subprocess
import tempfile

import numpy as np

import datasets

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
ORACLE_EXE = os.path.join(ROOT, "oracle", "aln_oracle")
_work = {}


def workdir(name):
    if name not in _work:
        d = tempfile.mkdtemp(prefix="psvr_" + name + "_")
        datasets.materialize(name, d)
        _work[name] = d
    return _work[name]


def golden_dir(name):
    return os.path.join(HERE, "golden", name)


_idx = {}


def index_dir(name):
    """The index directory of a golden set: tests/golden/<set>/idx when the reference builder's files are committed; for a set that
    commits only their SHA-256 (idx.sha256 -- fx5, 0.5 Mbp of anchors) the index is built here with `panSVR index` (host C++,
    byte-identical to the reference builder on every committed set, tests/test_index_build.py) and every file is checked against
    the hash of the reference builder's before it is used."""
    d = os.path.join(golden_dir(name), "idx")
    if os.path.isdir(d):
        return d
    if name not in _idx:
        import hashlib
        out = os.path.join(workdir(name), "idx_built")
        os.makedirs(out, exist_ok=True)
        cli = os.path.join(ROOT, "pansvr_amd", "bin", "panSVR")
        subprocess.run([cli, "index", "-k", "22", "--sparse-hash", os.path.join(workdir(name), "anchors.fa"), out + "/"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for line in open(os.path.join(golden_dir(name), "idx.sha256")):
            h, fn = line.split()
            got = hashlib.sha256(open(os.path.join(out, fn), "rb").read()).hexdigest()
            assert got == h, "%s/%s: `panSVR index` wrote a file that differs from the reference builder's" % (name, fn)
        _idx[name] = out
    return _idx[name]


def golden_lines(name, rname):
    with gzip.open(os.path.join(golden_dir(name), rname + ".jsonl.gz"), "rt") as f:
        return [l for l in f.read().split("\n") if l]


def run_oracle(name, rname, trace=True, limit=None, score=None):
    w = workdir(name)
    cmd = [ORACLE_EXE, index_dir(name), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    if trace:
        cmd.append("--trace")
    if limit:
        cmd += ["--limit", str(limit)]
    if score:
        cmd += ["--score", ",".join(str(x) for x in score)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, check=True).stdout.decode()
    return [l.strip() for l in out.split("\n") if l.strip()]


def strip_trace(line):
    d = json.loads(line)
    for r in d["reads"]:
        r.pop("tr", None)
        r.pop("str", None)
    return d


def engine_records(reads, pairs, cig, ori, lens, lo, hi):
    """The engine's downloaded arrays (pansvr_amd.aln.Engine.download) as the JSON records ref_aln / aln_oracle print."""
    out = []
    for p in range(lo, hi):
        rr = []
        for k in range(2):
            r = reads[2 * p + k]
            res = []
            for i in range(int(r["n_result"])):
                c = r["cand"][i]
                ops = cig[int(c["cigar_off"]):int(c["cigar_off"]) + int(c["n_cigar"])]
                cg = "".join("%d%s" % (int(np.int16(int(w) >> 4)), "MIDNSHP=XB"[int(w) & 0xf]) for w in ops)
                res.append([int(c["align_score"]), int(c["chain_score"]), int(c["chr_id"]), int(c["ref_bg"]), int(c["read_bg"]), int(c["direction"]), int(c["mapq"]), cg])
            o = ori[2 * p + k]
            oc = ("%dS" % o["read_bg"] if o["read_bg"] > 0 else "") + "%dM" % (lens[2 * p + k] - int(o["read_bg"]))
            d = {"n": int(r["n_result"]), "unmapped": int(r["unmapped"]), "res": res,
                 "ori": [int(o["align_score"]), 0, int(o["chr_id"]), int(o["ref_bg"]), int(o["read_bg"]), int(o["direction"]), int(o["mapq"]), oc]}
            if pairs[p]["gain"]:
                d["prim"], d["sec"] = int(r["primary"]), int(r["secondary"])
                if r["primary"] != -1:
                    d["mate"] = [int(r["has_mate"]), int(r["mate_chr_id"]) if r["has_mate"] else 0, int(r["mate_ref_bg"]) if r["has_mate"] else 0]
            rr.append(d)
        q = pairs[p]
        out.append({"i": p, "reads": rr, "pe": [int(q["max_score"]), int(q["cur_isize"]), int(q["proper"]), int(q["gain"]), int(q["max1"]), int(q["max2"])]})
    return out
