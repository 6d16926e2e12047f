"""Shared helpers for the `aln` parity tests: materialise seeded inputs, run the oracle executable,
load the committed reference records."""
import gzip
import json
import os
import subprocess
import tempfile

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


def golden_lines(name, rname):
    with gzip.open(os.path.join(golden_dir(name), rname + ".jsonl.gz"), "rt") as f:
        return [l for l in f.read().split("\n") if l]


def run_oracle(name, rname, trace=True, limit=None):
    w = workdir(name)
    cmd = [ORACLE_EXE, os.path.join(golden_dir(name), "idx"), os.path.join(w, rname + ".fq"), os.path.join(w, "header.sam")]
    if trace:
        cmd.append("--trace")
    if limit:
        cmd += ["--limit", str(limit)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, check=True).stdout.decode()
    return [l.strip() for l in out.split("\n") if l.strip()]


def strip_trace(line):
    d = json.loads(line)
    for r in d["reads"]:
        r.pop("tr", None)
        r.pop("str", None)
    return d
