"""-m gpu, BASELINE.json configs[1] scale (1 M synthetic 150 bp pairs vs the 10 k-anchor reference, the bench
workload).  At this size the oracle cannot replay everything in test time, so the engine is checked through
properties that do not depend on size, plus exact parity on the part the oracle can replay:
  * prefix parity: read pairs are processed in input order and nothing ever looks ahead, so the first 20 000 pairs'
    records must equal what oracle/aln_oracle prints for them (same index, same FASTQ prefix);
  * idempotence: a second run of the resident batch returns identical bytes;
  * batch-split invariance: the same pairs as two consecutive batches (rand()/random_r state carried across) give the
    records of the single batch;
  * every candidate CIGAR consumes exactly the read length, scores/mapq are in range, and a checksum of all records is
    reported for the log."""
import json
import os
import subprocess
import tempfile
import zlib

import numpy as np
import pytest

import aln_common as ac
import bench_data
from test_emu_aln import normalise

pytestmark = pytest.mark.gpu
N_PAIRS = int(os.environ.get("PSVR_FULLSIZE_PAIRS", "1000000"))
N_ORACLE = 20000


records = ac.engine_records


def canon(reads, cig):
    """Arena offsets depend on the order in which device threads allocate: compare records with the offsets cleared and the
    CIGAR words gathered in (read, candidate) order."""
    r = reads.copy()
    k = np.arange(12)[None, :] < r["n_result"][:, None]
    off = r["cand"]["cigar_off"][k].astype(np.int64)
    n = r["cand"]["n_cigar"][k].astype(np.int64)
    tot = int(n.sum())
    starts = np.repeat(off - np.concatenate([[0], np.cumsum(n)[:-1]]), n)
    words = cig[starts + np.arange(tot)] if tot else np.zeros(0, np.uint32)
    r["cand"]["cigar_off"] = 0
    r["cand"][~k] = np.zeros((), dtype=r["cand"].dtype)        # slots past n_result hold leftovers of cut candidates
    return r.tobytes(), words.tobytes()


def test_fullsize_properties_and_prefix_parity():
    """BASELINE configs[1]: 1 M pairs of 150 bp against 10 k anchors."""
    run_config(dict(n_anchors=10000, seed=11), dict(seed=13), (150, 200, 400, 600), N_PAIRS, N_ORACLE, True)


def test_cfg5_long_reads_properties_and_prefix_parity():
    """BASELINE configs[4] shape (250 bp reads, 2 000 anchors with 2 kbp edges, indels up to 40: DP problems of several hundred
    anti-diagonals, direction bytes in the HBM slab, K = 3..5 kernels) at the configuration's full 1 M pairs, oracle parity on the
    first 5 k (the oracle's behaviour on this shape is pinned against the reference objects by tests/golden/fx4)."""
    run_config(dict(n_anchors=2000, seed=17, edge=2000, allele=(60, 2000)), dict(seed=19, L=250, frag=(400, 700), maxindel=40), (250, 400, 550, 700),
               int(os.environ.get("PSVR_CFG5_PAIRS", "1000000")), 5000, False)


def run_config(anc_kw, reads_kw, stat, N_PAIRS, N_ORACLE, split):
    from pansvr_amd import aln
    anc = bench_data.make_anchors(**anc_kw)
    ix = bench_data.build_index(anc, dense=True)
    index = aln.Index(ix, ["chr1", "chr2"], device=0)
    bases, base_off, ori, isize = bench_data.make_reads(anc, N_PAIRS, **reads_kw)
    lens = np.diff(base_off)
    params = aln.default_params(stat)
    eng = aln.Engine(index, params)
    eng.upload(bases, base_off, ori)
    eng.run()
    reads, pairs, cig = eng.download()
    st = eng.stats()
    # the offset iteration took its short cuts: pairs with N bases adopted their variant slot's records, and the batch ended with the
    # re-run round that changed no draw count (the oracle parity below is what says they are right)
    assert st["rounds"] >= 2 and st["adopted_pairs"] > 0 and st["pair_runs"] < N_PAIRS * 1.03
    # --- idempotence
    eng.run()
    reads2, pairs2, cig2 = eng.download()
    assert canon(reads, cig) == canon(reads2, cig2) and pairs.tobytes() == pairs2.tobytes()
    # --- the timed mode runs every kernel on the one stream, one after the other: same records as the overlapped launches
    eng.run(timing=True)
    reads2, pairs2, cig2 = eng.download()
    assert canon(reads, cig) == canon(reads2, cig2) and pairs.tobytes() == pairs2.tobytes()
    del reads2, pairs2, cig2
    # --- CIGAR / range invariants over ALL candidates
    n_res = reads["n_result"]
    assert n_res.min() >= 0 and n_res.max() <= 12
    consumes = np.array([1, 1, 0, 1, 1, 0, 0, 1, 1, 0], dtype=np.int64)      # read bases per op: M I D N S H P = X B  (reverseGIGAR counts M,I,N,S)
    consumes[7] = consumes[8] = 0
    tot_cand = 0
    for k in range(12):
        sel = np.nonzero(n_res > k)[0]
        if len(sel) == 0:
            break
        c = reads["cand"][sel, k]
        tot_cand += len(sel)
        assert (c["align_score"] >= 40).all() and (c["align_score"] <= 2 * lens[sel]).all() and (c["mapq"] <= 40).all()
        off, n = c["cigar_off"].astype(np.int64), c["n_cigar"].astype(np.int64)
        flat = np.concatenate([cig[o:o + m] for o, m in zip(off[:20000], n[:20000])]) if len(sel) else np.zeros(0, np.uint32)
        seg = np.repeat(np.arange(min(len(sel), 20000)), n[:20000])
        rl = np.bincount(seg, weights=((flat >> 4).astype(np.int64) * consumes[flat & 0xf]), minlength=min(len(sel), 20000))
        assert (rl == lens[sel][:20000]).all(), "a CIGAR does not consume the read length"
    checksum = zlib.crc32(reads.tobytes()) ^ zlib.crc32(pairs.tobytes())
    print("fullsize: %d pairs, %d candidates, gain %d, checksum %08x" % (N_PAIRS, tot_cand, int(pairs["gain"].sum()), checksum))
    # --- prefix parity against the oracle
    # (the reference's own objects, oracle/_ref/ref_aln, when they are there and RAM-backed storage can hold the dense first-level table
    # their loader reads; the restatement oracle/aln_oracle -- pinned against them by the fx1..fx5 goldens -- otherwise)
    import shutil
    ref_exe = os.path.join(ac.ROOT, "oracle", "_ref", "ref_aln")
    use_ref = os.path.exists(ref_exe) and os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (6 << 30)
    tmp = tempfile.mkdtemp(prefix="psvr_full_", dir="/dev/shm" if use_ref else None)
    small = {k: v for k, v in ix.items() if k != "hash"}
    bench_data.write_index_dir(small, os.path.join(tmp, "idx"), dense_hash=ix["hash"] if use_ref else None)
    bench_data.write_fastq(os.path.join(tmp, "sample.fq"), bases, base_off, ori, isize, stat=stat, n_pairs=N_ORACLE)
    with open(os.path.join(tmp, "header.sam"), "w") as f:
        f.write("@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n")
    base = [os.path.join(tmp, "idx"), os.path.join(tmp, "sample.fq"), os.path.join(tmp, "header.sam")]
    try:
        if use_ref:
            out = subprocess.run([ref_exe, "-t", "1", "-R", str(N_ORACLE)] + base + ["--quiet"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
        else:
            out = subprocess.run([ac.ORACLE_EXE] + base, stdout=subprocess.PIPE, check=True).stdout.decode()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("prefix parity against", "oracle/_ref/ref_aln (reference objects)" if use_ref else "oracle/aln_oracle")
    want = [json.loads(l) for l in out.split("\n") if l.lstrip().startswith("{")][:N_ORACLE]
    got = records(reads, pairs, cig, ori, lens, 0, N_ORACLE)
    assert len(want) == N_ORACLE
    bad = [i for i in range(N_ORACLE) if want[i] != got[i]]
    assert not bad, "%d/%d prefix pairs differ, first %d:\noracle %s\nengine %s" % (len(bad), N_ORACLE, bad[0], json.dumps(want[bad[0]]), json.dumps(got[bad[0]]))
    if not split:
        eng.close(), index.close()
        return
    # --- batch-split invariance (first 200 k pairs as 2 x 100 k with the stream state carried across)
    n2 = min(200000, N_PAIRS)
    h = n2 // 2
    eng2 = aln.Engine(index, params)
    parts = []
    for lo, hi in ((0, h), (h, n2)):
        eng2.upload(bases[base_off[2 * lo]:base_off[2 * hi]], base_off[2 * lo:2 * hi + 1] - base_off[2 * lo], ori[2 * lo:2 * hi])
        eng2.run()
        parts.append(eng2.download())
    for (lo, hi), (r2, p2, c2) in zip(((0, h), (h, n2)), parts):
        assert p2.tobytes() == pairs[lo:hi].tobytes()
        a, b = reads[2 * lo:2 * hi].copy(), r2.copy()
        for x in (a, b):
            for fld in ("seed_hash", "chain_hash", "n_seed"):
                x[fld] = 0
        assert canon(a, cig) == canon(b, c2)
    eng.close()
    eng2.close()
    index.close()
