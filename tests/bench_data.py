"""Bench-scale synthetic workload (SURVEY 8(d) cfg 2), vectorised with numpy so that 10 k anchors and
1 M read pairs are generated in seconds:

  * build_index(): a deBGA-FORMAT index for random (repeat-free) anchors, one unipath per anchor.
    The nine arrays have exactly the layout deBGA writes (tests/test_bench_data.py checks them against the
    reference-built fixture of tests/golden/fx1); unlike deBGA it does not merge/split unipaths at repeated
    22-mers, which random anchors do not have.
  * make_reads(): 150 bp pairs drawn from the anchors with substitutions / deletions / insertions, a share
    of random non-anchor pairs, N bases and the constant original-alignment fields of the tests' generator.
"""
import numpy as np

from pansvr_amd.aln import ORI_DTYPE

NB = 1 << 28


def make_anchors(n_anchors, seed=11, edge=500, allele=(60, 300)):
    rng = np.random.RandomState(seed)
    alen = rng.randint(allele[0], allele[1] + 1, size=n_anchors)
    lens = alen + 2 * edge
    starts = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    codes = rng.randint(0, 4, size=int(starts[-1])).astype(np.uint8)
    names = ["%d_chr1_%d_%d_INS_%d_%d_%d_sv.INS.%d" % (i, 10000 * (i + 1), lens[i], 10000 * (i + 1) + edge, 10000 * (i + 1) + edge, 10000 * (i + 1) + 2 * edge, i)
             for i in range(n_anchors)]
    return dict(codes=codes, starts=starts, lens=lens, names=names, st_pos=10000 * (np.arange(n_anchors) + 1))


def pack2bit(codes):
    n = len(codes)
    pad = (-n) % 32
    c = np.concatenate([codes.astype(np.uint64), np.zeros(pad, np.uint64)]).reshape(-1, 32)
    sh = ((31 - np.arange(32)) * 2).astype(np.uint64)
    return (c << sh).sum(axis=1, dtype=np.uint64)


def build_index(anc, dense=True):
    codes, starts, lens = anc["codes"], anc["starts"], anc["lens"]
    N = len(codes)
    words = pack2bit(codes)
    # every 22-mer start that lies inside one anchor
    c64 = codes.astype(np.uint64)
    v = np.zeros(N - 21, dtype=np.uint64)
    for j in range(22):
        v = (v << np.uint64(2)) | c64[j:N - 21 + j]
    valid = np.ones(N - 21, dtype=bool)
    for e in starts[1:-1]:
        valid[max(0, e - 21):e] = False
    offs = np.nonzero(valid)[0].astype(np.uint64)
    v = v[valid]
    order = np.argsort(v, kind="stable")     # (first 14 bases, last 8 bases), ties by offset
    v, offs = v[order], offs[order]
    bucket = (v >> np.uint64(16)).astype(np.int64)
    ub, uc = np.unique(bucket, return_counts=True)              # non-empty first-level buckets
    sparse = np.stack([ub.astype(np.uint32), uc.astype(np.uint32)], axis=1)
    h = None
    if dense:                                                    # the 2 GiB prefix-sum table deBGA writes
        h = np.zeros(NB + 1, dtype=np.uint64)
        h[ub + 1] = uc
        np.cumsum(h, out=h)
    chr_text = "".join("%s\n%d\n" % (nm, starts[i + 1] + 1) for i, nm in enumerate(anc["names"]))
    U = len(lens)
    return dict(ref_seq=words, seq=words.copy(), seqf=starts.astype(np.uint64), pos=(starts[:-1] + 1).astype(np.uint64),
                posp=np.arange(U + 1, dtype=np.uint64), hash=h, hash_sparse=sparse, kmer=(v & np.uint64(0xffff)).astype(np.uint32), off=offs, chr=chr_text)


def build_index_cli(anc, dense=True):
    """The same dict as build_index, but through `panSVR index` (pansvr_amd/csrc/index_build.h), the builder that reproduces the
    reference's index files byte for byte -- also for the handful of 22-mers a random 12 Mbp anchor set repeats."""
    import os
    import shutil
    import subprocess
    import tempfile
    import index_fixture
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tmp = tempfile.mkdtemp(prefix="psvr_idxb_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        codes, starts = anc["codes"], anc["starts"]
        with open(os.path.join(tmp, "anchors.fa"), "wb") as f:
            for i, nm in enumerate(anc["names"]):
                f.write(b">" + nm.encode() + b"\n")
                f.write(lut[codes[starts[i]:starts[i + 1]]].tobytes() + b"\n")
        subprocess.check_call([os.path.join(root, "pansvr_amd", "bin", "panSVR"), "index", "-k", "22", "--sparse-hash", os.path.join(tmp, "anchors.fa"), tmp],
                              stderr=subprocess.DEVNULL)
        a = index_fixture.load_arrays(tmp) if dense else None
        if a is None:
            a = {}
            for k, fn, dt in (("ref_seq", "ref.seq", np.uint64), ("seq", "unipath.seqb", np.uint64), ("seqf", "unipath.seqfb", np.uint64), ("pos", "unipath.pos", np.uint64),
                              ("posp", "unipath.posp", np.uint64), ("kmer", "unipath_g.kmer", np.uint32), ("off", "unipath_g.offset", np.uint64)):
                a[k] = np.fromfile(os.path.join(tmp, fn), dtype=dt)
            a["hash"] = None
            a["chr"] = open(os.path.join(tmp, "unipath.chr")).read()
        a["hash_sparse"] = np.fromfile(os.path.join(tmp, "unipath_g.hash.sparse"), dtype=np.uint32).reshape(-1, 2)
        return a
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def write_index_dir(ix, path, dense_hash=None):
    """On-disk form for the oracle executable (sparse first level, see tests/index_fixture.py).  dense_hash: also write the
    2 GiB prefix-sum table unipath_g.hash, which the reference's own loader (oracle/_ref/ref_aln) reads."""
    import os
    os.makedirs(path, exist_ok=True)
    if dense_hash is not None:
        dense_hash.tofile(os.path.join(path, "unipath_g.hash"))
    for f, k in (("ref.seq", "ref_seq"), ("unipath.seqb", "seq"), ("unipath.seqfb", "seqf"), ("unipath.pos", "pos"), ("unipath.posp", "posp"),
                 ("unipath_g.kmer", "kmer"), ("unipath_g.offset", "off")):
        ix[k].tofile(os.path.join(path, f))
    ix["hash_sparse"].tofile(os.path.join(path, "unipath_g.hash.sparse"))
    with open(os.path.join(path, "unipath.chr"), "w") as f:
        f.write(ix["chr"])


def make_reads(anc, n_pairs, seed=13, L=150, frag=(300, 500), maxindel=8, miss_frac=0.2, n_frac=0.01):
    """Returns (bases uint8 ASCII [2P*L], base_off int64 [2P+1], ori ORI_DTYPE [2P], isize[P])."""
    rng = np.random.RandomState(seed)
    codes, starts, lens = anc["codes"], anc["starts"], anc["lens"]
    P = n_pairs
    a = rng.randint(1, len(lens), size=P)                       # anchor 0 is never sampled (see tests/datasets.py)
    flen = rng.randint(frag[0], frag[1] + 1, size=P)
    room = lens[a] - flen - 2 * maxindel - 2
    off = (rng.random_sample(P) * np.maximum(room, 1)).astype(np.int64)
    g0 = starts[a] + off                                        # global start of the fragment
    W = L + maxindel
    ar = np.arange(W)
    end1 = codes[g0[:, None] + ar[None, :]]                     # forward end
    g1 = g0 + flen - 1
    end2 = 3 - codes[g1[:, None] - ar[None, :]]                 # reverse-complement end
    src = np.stack([end1, end2], axis=1).reshape(2 * P, W)      # record 2p = end1, 2p+1 = end2
    R = 2 * P
    miss = np.repeat(rng.random_sample(P) < miss_frac, 2)
    src[miss] = rng.randint(0, 4, size=(int(miss.sum()), W))
    kind = rng.choice(4, size=R, p=[0.3, 0.2, 0.2, 0.3])
    idx = np.tile(np.arange(L), (R, 1))
    # deletions: skip d source bases at p
    d = rng.randint(1, maxindel + 1, size=R)
    p = rng.randint(10, L - 10 - maxindel, size=R)
    dele = kind == 1
    idx[dele] += (np.arange(L)[None, :] >= p[dele, None]) * d[dele, None]
    # insertions: d random bases at p, the rest shifts right
    ins = kind == 2
    j = np.arange(L)[None, :]
    shift = np.clip(j - p[:, None], 0, d[:, None])
    idx[ins] = (j - shift)[ins]
    reads = np.take_along_axis(src, idx, axis=1)
    inside = ins[:, None] & (j >= p[:, None]) & (j < (p + d)[:, None])
    reads[inside] = rng.randint(0, 4, size=int(inside.sum()))
    # substitutions: 1-4 per read of kind 0
    sub = np.nonzero(kind == 0)[0]
    for _ in range(4):
        sel = sub[rng.random_sample(len(sub)) < 0.625]
        pos = rng.randint(0, L, size=len(sel))
        reads[sel, pos] = (reads[sel, pos] + rng.randint(1, 4, size=len(sel))) % 4
    asc = np.frombuffer(b"ACGT", dtype=np.uint8)[reads]
    nn = np.nonzero(rng.random_sample(R) < n_frac)[0]
    asc[nn, rng.randint(0, L, size=len(nn))] = ord("N")
    # which mate is first in pair
    swap = rng.random_sample(P) < 0.5
    order = np.arange(R).reshape(P, 2)
    order[swap] = order[swap][:, ::-1]
    order = order.reshape(-1)
    asc = asc[order]
    fwd = (order % 2 == 0)
    ori = np.zeros(R, dtype=ORI_DTYPE)
    ori["chr_id"] = 0
    pos1 = anc["st_pos"][a] + off
    pos2 = pos1 + flen - L
    ori["ref_bg"] = np.where(fwd, np.repeat(pos1, 2), np.repeat(pos2, 2))
    ori["read_bg"], ori["align_score"], ori["mapq"] = 40, 140, 20
    ori["direction"] = fwd.astype(np.uint8)
    base_off = (np.arange(R + 1, dtype=np.int64) * L)
    return asc.reshape(-1), base_off, ori, flen


def _fastq_chunk(args):
    path, bases, base_off, ori, isize, stat, p0, p1, name_base = args
    out = []
    for p in range(p0, p1):
        for k in range(2):
            r = 2 * p + k
            o = ori[r]
            fw = bool(o["direction"])
            mate = ori[2 * p + 1 - k]
            flag = (0x40 if k == 0 else 0x80) | 0x1 | (0 if fw else 0x10) | (0x20 if fw else 0)
            c = "%d_%d_%d_%d_20_20_0_0_%d_%sNNY_%sNNY_" % (o["chr_id"], o["ref_bg"], o["read_bg"], o["align_score"], isize[p], "F" if fw else "R", "R" if fw else "F")
            if p == 0 and k == 0 and stat is not None:
                c += "STAT_%d_%d_%d_%d_" % stat
            c += "FLAG_%d_20_CIGAR_40S110M_MATE_0_%d_%d_TAG_NM:i:3_" % (flag, mate["ref_bg"], isize[p] if fw else -isize[p])
            s = bases[base_off[r]:base_off[r + 1]].tobytes().decode()
            out.append("@r%07d %s\n%s\n+\n%s\n" % (name_base + p, c, s, "I" * len(s)))
    with open(path, "w") as f:
        f.write("".join(out))
    return path


def write_fastq(path, bases, base_off, ori, isize, stat=(150, 200, 400, 600), n_pairs=None, procs=1, name_base=0, append=False):
    """FASTQ with the fc_signal comment (tests/synth.py format) for the first n_pairs pairs.  procs > 1: chunks are formatted by forked
    worker processes (call this BEFORE the process touches the GPU); append: add to an existing file (a multi-part input)."""
    import os
    import shutil
    P = (len(base_off) - 1) // 2 if n_pairs is None else n_pairs
    nchunk = max(1, min(procs, P // 20000 + 1))
    bounds = [P * i // nchunk for i in range(nchunk + 1)]
    jobs = [(path + ".part%d" % i, bases, base_off, ori, isize, stat, bounds[i], bounds[i + 1], name_base) for i in range(nchunk)]
    if nchunk > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(nchunk) as pool:
            parts = pool.map(_fastq_chunk, jobs)
    else:
        parts = [_fastq_chunk(jobs[0])]
    with open(path, "ab" if append else "wb") as f:
        for q in parts:
            with open(q, "rb") as g:
                shutil.copyfileobj(g, f, 1 << 24)
            os.remove(q)
