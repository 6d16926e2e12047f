"""-m gpu: the N > 1 path on the real engine.  Three ranks (processes) share the one GPU of the test box, each owns a
contiguous shard of the pairs, runs it with psvr_engine_run, exchanges draw counts over gloo (pansvr_amd/dist.py, what
bench.py does over RCCL) and moves to its true stream position with psvr_engine_rebase.  The concatenated results must
equal one engine's results over the whole input: records, CIGARs, pairing decisions."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

import aln_common as ac
import bench_data
from test_fullsize_gpu import canon

pytestmark = pytest.mark.gpu
N_PAIRS, WORLD = 60000, 3


def _setup():
    anc = bench_data.make_anchors(1500, seed=23)
    ix = bench_data.build_index(anc, dense=True)
    bases, base_off, ori, isize = bench_data.make_reads(anc, N_PAIRS, seed=29, n_frac=0.03)
    return ix, bases, base_off, ori


def _rank_main(rank, world, port, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ac.ROOT)
    from pansvr_amd import aln
    from pansvr_amd import dist as pd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ix, bases, base_off, ori = _setup()
    lo, hi = pd.shard_bounds(N_PAIRS, rank, world)
    index = aln.Index(ix, ["chr1", "chr2"], device=0)
    eng = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
    eng.upload(bases[base_off[2 * lo]:base_off[2 * hi]], base_off[2 * lo:2 * hi + 1] - base_off[2 * lo], ori[2 * lo:2 * hi])
    calls = []

    def run_at(pos):
        calls.append(("run", list(pos)))
        eng.set_stream_pos(pos)
        eng.run()
        return eng.stream_end()

    def rebase_to(pos):
        calls.append(("rebase", list(pos)))
        eng.rebase(pos)
        return eng.stream_end()

    start, end, iters = pd.resolve_stream_order([2, 0, 0], run_at, rebase_to)
    reads, pairs, cig = eng.download()
    np.save(os.path.join(outdir, "reads%d.npy" % rank), reads)
    np.save(os.path.join(outdir, "pairs%d.npy" % rank), pairs)
    np.save(os.path.join(outdir, "cig%d.npy" % rank), cig)
    np.save(os.path.join(outdir, "meta%d.npy" % rank), np.array([lo, hi, iters, len(calls)] + list(start) + list(end), dtype=np.int64))
    dist.barrier()
    eng.close(), index.close()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_three_rank_sharding_on_the_engine_equals_one_engine():
    from pansvr_amd import aln
    outdir = tempfile.mkdtemp(prefix="psvr_shard_")
    mp.spawn(_rank_main, args=(WORLD, 29541, outdir), nprocs=WORLD, join=True)
    ix, bases, base_off, ori = _setup()
    index = aln.Index(ix, ["chr1", "chr2"], device=0)
    eng = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
    eng.upload(bases, base_off, ori)
    eng.run()
    reads, pairs, cig = eng.download()
    end_all = eng.stream_end()
    metas = [np.load(os.path.join(outdir, "meta%d.npy" % r)) for r in range(WORLD)]
    assert metas[0][0] == 0 and metas[-1][1] == N_PAIRS
    for r in range(1, WORLD):
        assert list(metas[r][4:7]) == list(metas[r - 1][7:10]), "rank %d must start where rank %d ended" % (r, r - 1)
        assert list(metas[r][4:7]) != [2, 0, 0] and metas[r][3] >= 2          # the exchange really moved it
    assert list(metas[-1][7:10]) == list(end_all)
    for r in range(WORLD):
        lo, hi = int(metas[r][0]), int(metas[r][1])
        rr, pp, cc = (np.load(os.path.join(outdir, "%s%d.npy" % (k, r))) for k in ("reads", "pairs", "cig"))
        assert pp.tobytes() == pairs[lo:hi].tobytes(), "pairing records of rank %d differ" % r
        a, b = reads[2 * lo:2 * hi].copy(), rr.copy()
        for x in (a, b):
            for fld in ("seed_hash", "chain_hash", "n_seed"):
                x[fld] = 0
        ca, cb = canon(a, cig), canon(b, cc)
        if ca != cb:                                  # say which read and which fields before failing
            ra, rb = np.frombuffer(ca[0], dtype=a.dtype), np.frombuffer(cb[0], dtype=b.dtype)
            bad = [i for i in range(len(ra)) if ra[i].tobytes() != rb[i].tobytes()]
            detail = "cigar words differ only" if not bad else "first of %d reads: %d (pair %d) one=%r shard=%r" % (len(bad), bad[0], lo + bad[0] // 2, ra[bad[0]], rb[bad[0]])
            raise AssertionError("read records of rank %d differ: %s" % (r, detail))
    eng.close(), index.close()


def test_rebase_equals_a_run_from_the_new_position():
    """One engine, no exchange: a batch that ran from [2, 0, 0] and is moved (psvr_engine_rebase) must hold what a run from the new
    position holds, and end where it ends -- at several distances, so that the N pairs (3 % of the reads here) meet residues that select
    the variant slot they carry already as well as other ones, and with no pair evaluated in full again by the rebase
    (`pair_runs`: the adoption / re-selection / pairing-only paths of engine_core.h carry it)."""
    from pansvr_amd import aln
    ix, bases, base_off, ori = _setup()
    index = aln.Index(ix, ["chr1", "chr2"], device=0)
    eng = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
    eng.upload(bases, base_off, ori)

    def records():
        reads, pairs, cig = eng.download()
        r = reads.copy()
        for fld in ("seed_hash", "chain_hash", "n_seed"):
            r[fld] = 0
        return canon(r, cig), pairs.tobytes(), eng.stream_end()

    for pos in ([3, 0, 0], [2 + 1000, 0, 0], [2 + 123457, 5, 3], [2 + 40000, 0, 0]):
        eng.set_stream_pos(pos)
        eng.run()
        want = records()
        eng.set_stream_pos([2, 0, 0])
        eng.run()
        runs = eng.stats()["pair_runs"]
        eng.rebase(pos)
        got = records()
        assert got[2] == want[2], (pos, got[2], want[2])
        assert got[1] == want[1], "pairing records differ after the rebase to %r" % (pos,)
        assert got[0] == want[0], "read records differ after the rebase to %r" % (pos,)
        if pos[1] == 0 and pos[2] == 0:                      # (a moved random_r stream sends the reads that sampled positions through in full)
            assert eng.stats()["pair_runs"] == runs, (pos, eng.stats()["pair_runs"], runs)
    eng.close(), index.close()
