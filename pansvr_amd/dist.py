"""Sharding one batch of read pairs over ranks (one process per GPU, torch.distributed).

Pairs are independent given the replicated index, so rank r simply takes the r-th contiguous block (concatenating the
ranks' outputs restores input order).  The only coupling is the reference's single rand()/random_r draw sequence:
shard r starts where shard r-1 ended.  `resolve_stream_order` finds those positions with one tiny all-gather (six integers) per
iteration: every rank runs (or rebases) its shard at its current start position, all ranks exchange how many draws their
shards consumed, and each recomputes its start as first + sum of the draws of the ranks before it -- until no start
moves (normally two iterations: draw counts almost never depend on the start position)."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, order-preserving split: pair i belongs to rank floor(i * world / n)."""
    lo = (n * rank + world - 1) // world
    hi = (n * (rank + 1) + world - 1) // world
    return lo, hi


def all_gather_i64(vec, device=None, group=None):
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.tensor([int(x) for x in vec], dtype=torch.int64, device=device)
    if world == 1:
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).tolist()          # one readback for all ranks' values (a .tolist() per rank is a device synchronisation each)


def data_plane(want_rccl, timeout_s=180):
    """The process group the bulk data (index broadcast, record gather) and the draw-count exchange travel on.

    The default group is gloo (the control plane: it comes up wherever torch.distributed does).  When `want_rccl`, an RCCL group
    (backend "nccl") over the same ranks is created beside it and tried with one all-reduce of a device tensor; the ranks then AGREE over
    gloo that every one of them got through -- a rank whose communicator or first collective failed sends everybody to the gloo plane
    (host tensors), so the collective is never entered by some ranks only.  Returns (group, device, how): group None = the default (gloo)
    group, device "cuda" or None (host tensors)."""
    import datetime
    import os
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None, None, "single process"
    if not want_rccl:
        return None, None, "gloo (host tensors)"
    ok, pg, why = 1, None, ""
    try:
        # a collective a peer never joins must end in an exception here, not in the watchdog taking the process down
        os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
        pg = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=timeout_s))
        t = torch.ones(1, dtype=torch.int64, device="cuda")
        dist.all_reduce(t, group=pg)
        torch.cuda.synchronize()
        if int(t.item()) != dist.get_world_size():
            ok, why = 0, "trial all-reduce returned %d" % int(t.item())
    except Exception as ex:                   # noqa: BLE001 -- whatever went wrong, the fallback is the same
        ok, why = 0, repr(ex)[:200]
    flag = torch.tensor([ok])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)         # default group: gloo
    if int(flag.item()) == 1:
        return pg, "cuda", "RCCL (backend nccl) beside a gloo control plane"
    return None, None, "gloo (host tensors): the RCCL plane did not come up on every rank" + (" -- this rank: " + why if why else "")


def resolve_stream_order(first_pos, run_at, rebase_to, device=None, max_iter=16, group=None, times=None):
    """first_pos: [g, h0, h1] where rank 0's shard starts.  run_at(pos) -> end runs this rank's shard from `pos`;
    rebase_to(pos) -> end moves the finished run to `pos`.  Returns (start, end, iterations) of this rank.
    times: optional dict, the seconds spent in run_at / rebase_to / the exchange are added to its "run", "rebase", "exchange" entries."""
    import time
    rank = dist.get_rank() if dist.is_initialized() else 0
    start = [int(x) for x in first_pos]

    def clocked(key, f, *a):
        t = time.time()
        r = f(*a)
        if times is not None:
            times[key] = times.get(key, 0.0) + (time.time() - t)
        return r
    end = clocked("run", run_at, start)
    it = 1
    while True:
        # one collective per iteration: every rank's draw counts AND the start it ran from, so that each rank can tell for itself whether
        # any rank has to move (the ranks must agree on when to stop)
        every = clocked("exchange", all_gather_i64, [e - s for e, s in zip(end, start)] + start, device, group)
        starts = [[int(first_pos[k]) + sum(every[p][k] for p in range(q)) for k in range(3)] for q in range(len(every))]
        if all(starts[q] == every[q][3:6] for q in range(len(every))):
            return start, end, it
        if it >= max_iter:
            raise RuntimeError("stream-order resolution did not converge")
        if starts[rank] != start:
            start = starts[rank]
            end = clocked("rebase", rebase_to, start)
        it += 1


class BlockGather:
    """Ordered gather of the ranks' result blocks on rank 0 (the reference's output_results walks a batch in input order,
    read_realignment.cpp:165-176; rank r owns the r-th contiguous block, so block order IS input order).

    Every rank packs its block into one byte buffer (pack(buf) -> bytes used; buf is a uint8 tensor on `device`, or in host memory for the
    gloo plane); the sizes travel in one all-gather, the blocks point to point: rank 0 posts a receive per peer into that peer's own buffer
    and all of them progress together (batch_isend_irecv: one ncclGroup of sends / receives -- xGMI is point to point, the seven peers
    of a node arrive over seven different links).  Rank 0 ends up with blocks[0 .. world) in input order; nothing is concatenated: a record
    writer walks the blocks in turn (block-relative offsets stay valid)."""

    def __init__(self, device=None, group=None):
        self.device, self.group = device, group
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.mine = None
        self.blocks = [None] * self.world          # rank 0: the peers' blocks (its own is self.mine)
        self.sizes = [0] * self.world

    def _buf(self, old, nbytes):
        if old is not None and old.numel() >= nbytes:
            return old
        cap = nbytes + nbytes // 8 + 4096
        if self.device:
            return torch.empty(cap, dtype=torch.uint8, device=self.device)
        t = torch.empty(cap, dtype=torch.uint8)
        try:
            t = t.pin_memory()
        except Exception:                         # noqa: BLE001 -- no device: plain host memory
            pass
        return t

    def gather(self, need_bytes, pack):
        """need_bytes: upper bound of this rank's packed block; pack(buf) fills buf[0:n) and returns (n, meta) -- meta: a few integers that
        describe the block's layout (the same number on every rank); they travel with the sizes.  Returns the list of
        (tensor, nbytes, meta) in block order on rank 0, None elsewhere."""
        self.mine = self._buf(self.mine, need_bytes)
        n, meta = pack(self.mine)
        n, meta = int(n), [int(x) for x in meta]
        every = all_gather_i64([n] + meta, self.device, self.group)
        sizes = [int(v[0]) for v in every]
        self.sizes = sizes
        if self.world > 1:
            ops = []
            if self.rank == 0:
                for r in range(1, self.world):
                    self.blocks[r] = self._buf(self.blocks[r], sizes[r])
                    if sizes[r]:
                        ops.append(dist.P2POp(dist.irecv, self.blocks[r][:sizes[r]], r, self.group))
            elif n:
                ops.append(dist.P2POp(dist.isend, self.mine[:n], 0, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if self.device:
                torch.cuda.synchronize()
        if self.rank != 0:
            return None
        return [(self.mine, n, meta)] + [(self.blocks[r], sizes[r], [int(x) for x in every[r][1:]]) for r in range(1, self.world)]


class EngineGroup:
    """Several engines of ONE process that share a block of pairs (on one device: the stages of the path are latency / issue bound, so
    two or three engines running beside each other on their own HIP queues fill the chip better than one; or one per device).  Engine j
    owns the j-th contiguous sub-block; run_at / rebase_to give the group the interface resolve_stream_order expects of one engine:
    inside the group the same draw-order rule holds -- engine j starts where engine j-1 ended."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.starts = [[2, 0, 0] for _ in self.engines]
        self.rebases = 0

    def _parallel(self, fns):
        import threading
        if len(fns) == 1:
            fns[0]()
            return
        err = []

        def wrap(f):
            try:
                f()
            except Exception as ex:      # surfaced on the calling thread
                err.append(ex)
        th = [threading.Thread(target=wrap, args=(f,)) for f in fns[1:]]
        for t in th:
            t.start()
        wrap(fns[0])
        for t in th:
            t.join()
        if err:
            raise err[0]

    def _resolve(self, pos):
        pos = [int(x) for x in pos]
        for it in range(64):
            ends = [e.stream_end() for e in self.engines]
            acc, moved = list(pos), []
            for j, e in enumerate(self.engines):
                used = [ends[j][k] - self.starts[j][k] for k in range(3)]
                if self.starts[j] != acc:
                    self.starts[j] = list(acc)
                    moved.append(j)
                acc = [acc[k] + used[k] for k in range(3)]
            if not moved:
                return ends[-1]
            self.rebases += len(moved)
            self._parallel([(lambda j=j: self.engines[j].rebase(self.starts[j])) for j in moved])
        raise RuntimeError("draw-order resolution inside the engine group did not converge")

    def run_at(self, pos):
        pos = [int(x) for x in pos]

        def one(e):
            e.set_stream_pos(pos)
            e.run()
        self.starts = [list(pos) for _ in self.engines]
        self._parallel([(lambda e=e: one(e)) for e in self.engines])
        return self._resolve(pos)

    def rebase_to(self, pos):
        return self._resolve(pos)
