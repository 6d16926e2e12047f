"""N > 1 path on CPU: two gloo ranks shard fx1's read pairs, run their shard through the engine's stage logic
(the test-only host backend tests/emu) and exchange draw counts with pansvr_amd/dist.py exactly as bench.py does
over RCCL.  Concatenating the ranks' records must reproduce the reference's single-stream records bit for bit."""
import json
import os
import re
import subprocess
import sys
import tempfile

import pytest
import torch.multiprocessing as mp

import aln_common as ac
from test_emu_aln import normalise

EMU = os.path.join(ac.HERE, "emu", "emu_aln")


def free_port():
    """A port nobody listens on right now (two runs of the suite on one box, or a leftover rendezvous, must not collide)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, workdir, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ac.ROOT)
    from pansvr_amd import dist as pd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs = open(os.path.join(workdir, "reads150.fq")).read().split("\n")
    n_pairs = len(recs) // 8
    lo, hi = pd.shard_bounds(n_pairs, rank, world)
    shard = os.path.join(outdir, "shard%d.fq" % rank)
    lines = recs[8 * lo:8 * hi]
    if rank > 0:   # STAT_ travels with the very first read of the whole input only: keep the parameters identical
        first = recs[0].split(" ", 1)[1]
        stat = re.search(r"STAT_\d+_\d+_\d+_\d+_", first).group(0)
        lines[0] = lines[0].replace("FLAG_", stat + "FLAG_", 1)
    open(shard, "w").write("\n".join(lines) + "\n")
    out = os.path.join(outdir, "out%d.jsonl" % rank)
    calls = []

    def run(pos, extra=()):
        cmd = [EMU, os.path.join(ac.golden_dir("fx1"), "idx"), shard, os.path.join(workdir, "header.sam"), "--trace",
               "--stream-pos", "%d,%d,%d" % tuple(pos)] + list(extra)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        open(out, "wb").write(r.stdout)
        calls.append(list(pos))
        m = re.search(r"stream_end (\d+) (\d+) (\d+)", r.stderr.decode())
        return [int(m.group(i)) for i in (1, 2, 3)]

    prev = {}

    def run_at(pos):
        prev["pos"] = list(pos)
        return run(pos)

    def rebase_to(pos):   # the emulator process is stateless: replay the first run, then rebase inside it
        e = run(pos, ("--rebase-from", "%d,%d,%d" % tuple(prev["pos"])))
        prev["pos"] = list(pos)
        return e

    start, end, iters = pd.resolve_stream_order([2, 0, 0], run_at, rebase_to)
    json.dump({"start": start, "end": end, "iters": iters, "calls": calls, "lo": lo, "hi": hi}, open(os.path.join(outdir, "meta%d.json" % rank), "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharding_reproduces_the_single_stream_records():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ac.HERE, "emu")])
    w = ac.workdir("fx1")
    outdir = tempfile.mkdtemp(prefix="psvr_gloo_")
    world = 2
    mp.spawn(_rank_main, args=(world, free_port(), w, outdir), nprocs=world, join=True)
    metas = [json.load(open(os.path.join(outdir, "meta%d.json" % r))) for r in range(world)]
    assert metas[0]["lo"] == 0 and metas[0]["hi"] == metas[1]["lo"] and metas[1]["hi"] == 2000
    assert metas[1]["start"] == metas[0]["end"], "rank 1 must start where rank 0 ended"
    assert metas[1]["start"] != [2, 0, 0] and len(metas[1]["calls"]) >= 2      # the exchange really moved rank 1
    got = []
    for r in range(world):
        lines = [l for l in open(os.path.join(outdir, "out%d.jsonl" % r)).read().split("\n") if l.strip()]
        for l in lines:
            d = normalise(l)
            d["i"] += metas[r]["lo"]
            got.append(d)
    want = [normalise(l) for l in ac.golden_lines("fx1", "reads150")]
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(want, got)) if a != b]
    assert not bad, "%d pairs differ, first %d:\n%s\n%s" % (len(bad), bad[0], want[bad[0]], got[bad[0]])


def test_bench_launch_contract_selects_the_local_rank_device():
    """The driver's launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) must bring rank r to HIP
    device LOCAL_RANK before anything touches a GPU: `bench.py --dry-run` goes through the rendezvous (127.0.0.1), the device
    choice and one barrier on CPU and prints what it chose."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                        os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    import re
    got = sorted((json.loads(m) for m in re.findall(r"\{[^{}]*\"dry_run\"[^{}]*\}", r.stdout.decode())), key=lambda d: d["rank"])
    assert [d["rank"] for d in got] == [0, 1] and all(d["device"] == d["local_rank"] == d["rank"] and d["world"] == 2 for d in got)


def _gather_main(rank, world, port, outdir):
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ac.ROOT)
    from pansvr_amd import dist as pd
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group, device, how = pd.data_plane(False)
    assert group is None and device is None
    g = pd.BlockGather(device, group)
    res = None
    for step in range(3):                               # the buffers are kept across steps; block sizes change from step to step
        n = 1000 + 777 * rank + 4096 * step * (rank + 1)
        block = ((np.arange(n, dtype=np.int64) * (rank + 3) + step) % 251).astype(np.uint8)

        def pack(buf):
            buf[:n] = torch.from_numpy(block)
            return n, [rank, step, 7]
        res = g.gather(n, pack)
        if rank == 0:
            assert len(res) == world
            for r, (buf, nb, meta) in enumerate(res):
                want = ((np.arange(nb, dtype=np.int64) * (r + 3) + step) % 251).astype(np.uint8)
                assert nb == 1000 + 777 * r + 4096 * step * (r + 1) and meta == [r, step, 7], (r, nb, meta)
                assert np.array_equal(buf[:nb].numpy(), want), "block %d of step %d arrived damaged or out of order" % (r, step)
        else:
            assert res is None
    if rank == 0:
        open(os.path.join(outdir, "gather_ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_block_gather_keeps_block_order_and_bytes():
    """The ordered gather of bench.py's N > 1 step (pansvr_amd/dist.py::BlockGather): three gloo ranks, blocks of different and changing
    sizes -- rank 0 must end up with every rank's bytes, in rank order (= input order, the reference's output_results contract)."""
    outdir = tempfile.mkdtemp(prefix="psvr_gather_")
    mp.spawn(_gather_main, args=(3, free_port(), outdir), nprocs=3, join=True)
    assert os.path.exists(os.path.join(outdir, "gather_ok"))
