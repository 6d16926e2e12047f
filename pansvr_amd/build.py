"""Builds libpsvr_engine.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpsvr_engine.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


CLI = os.path.join(HERE, "bin", "panSVR")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def needs_build():
    if not os.path.exists(OUT) or not os.path.exists(CLI):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(CLI))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "psvr_engine.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    bdir = os.path.join(ROOT, "build")
    os.makedirs(bdir, exist_ok=True)
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(bdir, os.path.splitext(s)[0] + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    # the drop-in CLI: plain host C++ above the C ABI (no HIP in this translation unit)
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", "-o", CLI, os.path.join(CSRC, "cli_main.cpp"),
           "-L" + HERE, "-lpsvr_engine", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
