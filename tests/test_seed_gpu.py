"""Seam B3: psvr_seed_search_kmer_batch / psvr_seed_mem_batch against known answers printed by the REFERENCE's own
deBGA_INDEX::search_kmer and ::UNITIG_MEM_search (tests/golden/seed_kat.jsonl.gz, made by tests/golden/gen_seed_kat.py with
oracle/_ref/ref_seed): every 20-mer offset of 120 reads of three index fixtures -- 15 720 probes, their hit ranges, and one
vertex_MEM (+ right_i) per hit."""
import ctypes as C
import gzip
import json
import os

import numpy as np
import pytest

import aln_common as ac
import index_fixture
import synth

VMEM_DTYPE = np.dtype([("uid", "<u8"), ("seed_id", "<u4"), ("read_pos", "<u4"), ("uni_pos_off", "<u4"), ("length", "<u4"), ("pos_n", "<u4"), ("right_i", "<u4")])


def load_kat():
    sets = {}
    with gzip.open(os.path.join(ac.HERE, "golden", "seed_kat.jsonl.gz"), "rt") as f:
        for l in f:
            d = json.loads(l)
            s = sets.setdefault(d["set"], {"reads": {}, "probes": []})
            if "words" in d:
                s["reads"][d["r"]] = d
            else:
                s["probes"].append(d)
    return sets


def test_known_answers_are_consistent_with_the_index_fixtures():
    """(no GPU) the ranges the reference printed, recomputed from the fixture arrays with numpy: the 22-mers of bucket kmer >> 12
    whose stored low bits >> 4 equal kmer & 0xfff."""
    for name, s in load_kat().items():
        d = ac.index_dir(name)
        sp = np.fromfile(os.path.join(d, "unipath_g.hash.sparse"), dtype=np.uint32).reshape(-1, 2)      # (bucket, count) of the non-empty buckets
        ids, start = sp[:, 0].astype(np.int64), np.concatenate([[0], np.cumsum(sp[:, 1].astype(np.int64))])
        kg = np.fromfile(os.path.join(d, "unipath_g.kmer"), dtype=np.uint32)
        for p in s["probes"][::7]:
            k = p["kmer"]
            j = int(np.searchsorted(ids, k >> 12))
            lo, hi = (int(start[j]), int(start[j + 1])) if j < len(ids) and ids[j] == k >> 12 else (0, 0)
            hits = [lo + i for i in range(hi - lo) if (int(kg[lo + i]) >> 4) == (k & 0xfff)]
            assert bool(hits) == bool(p["found"])
            if hits:
                assert [hits[0], hits[-1]] == p["range"] and hits == list(range(hits[0], hits[-1] + 1))
        assert sum(p["found"] for p in s["probes"]) > 500


@pytest.mark.gpu
def test_seed_entry_points_match_the_reference_functions():
    from pansvr_amd import aln
    from pansvr_amd._lib import check, lib
    L = lib()
    names = [l.split("SN:")[1].split("\t")[0] for l in synth.header_text().split("\n") if l.startswith("@SQ")]
    total_mems = 0
    for name, s in load_kat().items():
        index = aln.Index(index_fixture.load_arrays(ac.index_dir(name)), names, device=0)
        probes = s["probes"]
        n = len(probes)
        kmers = np.array([p["kmer"] for p in probes], dtype=np.uint64)
        rng, found = np.zeros(2 * n, dtype=np.int64), np.zeros(n, dtype=np.uint8)
        check(L.psvr_seed_search_kmer_batch(index.h, C.c_int64(n), kmers.ctypes.data_as(C.c_void_p), rng.ctypes.data_as(C.c_void_p), found.ctypes.data_as(C.c_void_p)))
        for i, p in enumerate(probes):
            assert int(found[i]) == p["found"], (name, i)
            if p["found"]:
                assert [int(rng[2 * i]), int(rng[2 * i + 1])] == p["range"], (name, i)
        # every hit the reference extended
        words, woff = [], {}
        for r, d in sorted(s["reads"].items()):
            woff[r] = len(words)
            words += d["words"]
        items = [(m, p) for p in probes for m in p["mems"]]
        m_n = len(items)
        kidx = np.array([m[0] for m, _ in items], dtype=np.uint64)
        wo = np.array([woff[p["r"]] for _, p in items], dtype=np.int64)
        ro = np.array([p["off"] for _, p in items], dtype=np.uint32)
        rl = np.array([s["reads"][p["r"]]["len"] for _, p in items], dtype=np.uint32)
        rb = np.array(words, dtype=np.uint64)
        out = np.zeros(m_n, dtype=VMEM_DTYPE)
        check(L.psvr_seed_mem_batch(index.h, C.c_int64(m_n), kidx.ctypes.data_as(C.c_void_p), rb.ctypes.data_as(C.c_void_p), C.c_int64(len(rb)), wo.ctypes.data_as(C.c_void_p),
                                    ro.ctypes.data_as(C.c_void_p), rl.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        for i, (m, p) in enumerate(items):
            got = [int(out[i][f]) for f in ("uid", "read_pos", "uni_pos_off", "length", "pos_n", "right_i")]
            assert got == m[1:], (name, i, got, m)
        total_mems += m_n
        # arguments are checked on the host before anything is launched
        bad = kidx.copy()
        bad[0] = 1 << 40
        assert L.psvr_seed_mem_batch(index.h, C.c_int64(m_n), bad.ctypes.data_as(C.c_void_p), rb.ctypes.data_as(C.c_void_p), C.c_int64(len(rb)), wo.ctypes.data_as(C.c_void_p),
                                     ro.ctypes.data_as(C.c_void_p), rl.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 1
        index.close()
    assert total_mems > 3000
