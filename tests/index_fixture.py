"""Compact on-disk form of a deBGA index directory for fixtures: every file verbatim except the
2 GiB prefix-sum table unipath_g.hash, stored as (bucket, count) uint32 pairs of its non-empty
first-level buckets (unipath_g.hash.sparse).  expand_hash() rebuilds the dense uint64 table."""
import os
import shutil

import numpy as np

SMALL = ["ref.seq", "unipath.chr", "unipath.pos", "unipath.posp", "unipath.seqb", "unipath.seqfb", "unipath_g.kmer", "unipath_g.offset"]
NBUCKET = 1 << 28


def compact(src_dir, dst_dir):
    os.makedirs(dst_dir, exist_ok=True)
    for f in SMALL:
        shutil.copy(os.path.join(src_dir, f), os.path.join(dst_dir, f))
    h = np.memmap(os.path.join(src_dir, "unipath_g.hash"), dtype=np.uint64, mode="r")
    assert h.shape[0] == NBUCKET + 1
    ids, cnts = [], []
    step = 1 << 24
    for s in range(0, NBUCKET, step):
        blk = np.asarray(h[s:s + step + 1])
        d = np.diff(blk)
        nz = np.nonzero(d)[0]
        ids.append((nz + s).astype(np.uint32))
        cnts.append(d[nz].astype(np.uint32))
    ids, cnts = np.concatenate(ids), np.concatenate(cnts)
    np.stack([ids, cnts], axis=1).astype(np.uint32).tofile(os.path.join(dst_dir, "unipath_g.hash.sparse"))
    return len(ids)


def expand_hash(fix_dir):
    sp = np.fromfile(os.path.join(fix_dir, "unipath_g.hash.sparse"), dtype=np.uint32).reshape(-1, 2)
    out = np.zeros(NBUCKET + 1, dtype=np.uint64)
    # hash[b+1] - hash[b] = count(b)  =>  hash = exclusive prefix sum
    out[sp[:, 0].astype(np.int64) + 1] = sp[:, 1]
    np.cumsum(out, out=out)
    return out


def load_arrays(fix_dir):
    """All nine index arrays as numpy (dense hash included), as the C ABI's psvr_index_view wants them."""
    a = {}
    a["ref_seq"] = np.fromfile(os.path.join(fix_dir, "ref.seq"), dtype=np.uint64)
    a["seq"] = np.fromfile(os.path.join(fix_dir, "unipath.seqb"), dtype=np.uint64)
    a["seqf"] = np.fromfile(os.path.join(fix_dir, "unipath.seqfb"), dtype=np.uint64)
    a["pos"] = np.fromfile(os.path.join(fix_dir, "unipath.pos"), dtype=np.uint64)
    a["posp"] = np.fromfile(os.path.join(fix_dir, "unipath.posp"), dtype=np.uint64)
    a["kmer"] = np.fromfile(os.path.join(fix_dir, "unipath_g.kmer"), dtype=np.uint32)
    a["off"] = np.fromfile(os.path.join(fix_dir, "unipath_g.offset"), dtype=np.uint64)
    a["hash"] = expand_hash(fix_dir)
    a["chr"] = open(os.path.join(fix_dir, "unipath.chr")).read()
    return a
