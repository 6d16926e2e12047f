"""The numpy index builder used for the bench-scale anchor set writes deBGA's format: on the fx1
anchors it must reproduce the reference-built fixture (tests/golden/fx1/idx) array for array."""
import os

import numpy as np

import aln_common as ac
import bench_data
import datasets


def test_numpy_index_builder_matches_reference_deBGA_index():
    anchors = datasets.anchors_of("fx1")
    tab = np.zeros(256, dtype=np.uint8)
    for i, ch in enumerate(b"ACGT"):
        tab[ch] = i
    codes = np.concatenate([tab[np.frombuffer(s, dtype=np.uint8)] for _, s in anchors])
    lens = np.array([len(s) for _, s in anchors])
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    anc = dict(codes=codes, starts=starts, lens=lens, names=[n for n, _ in anchors], st_pos=None)
    mine = bench_data.build_index(anc, dense=False)
    fx = os.path.join(ac.golden_dir("fx1"), "idx")
    ref = {k: np.fromfile(os.path.join(fx, f), dtype=dt) for k, f, dt in
           (("ref_seq", "ref.seq", np.uint64), ("seqf", "unipath.seqfb", np.uint64), ("pos", "unipath.pos", np.uint64), ("posp", "unipath.posp", np.uint64),
            ("kmer", "unipath_g.kmer", np.uint32), ("off", "unipath_g.offset", np.uint64))}
    ref["chr"] = open(os.path.join(fx, "unipath.chr")).read()
    n = len(mine["ref_seq"])
    assert np.array_equal(mine["ref_seq"], ref["ref_seq"][:n])
    # first-level table compared in its sparse (bucket, count) form: the dense table is 2 GiB
    assert np.array_equal(mine["hash_sparse"], np.fromfile(os.path.join(fx, "unipath_g.hash.sparse"), dtype=np.uint32).reshape(-1, 2))
    assert np.array_equal(mine["kmer"], ref["kmer"])
    assert mine["chr"].split() == ref["chr"].split()
    # unipath order may differ between builders: compare what each k-mer resolves to (reference position)
    def resolve(ix):
        uid = np.searchsorted(ix["seqf"], ix["off"], side="right") - 1
        return ix["pos"][ix["posp"][uid].astype(np.int64)] + (ix["off"] - ix["seqf"][uid])
    assert np.array_equal(resolve(mine), resolve(ref))
    assert len(mine["seqf"]) == len(ref["seqf"])
