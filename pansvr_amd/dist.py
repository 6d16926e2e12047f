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


def all_gather_i64(vec, device=None):
    world = dist.get_world_size() if dist.is_initialized() else 1
    t = torch.tensor([int(x) for x in vec], dtype=torch.int64, device=device)
    if world == 1:
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.stack(out).tolist()          # one readback for all ranks' values (a .tolist() per rank is a device synchronisation each)


def resolve_stream_order(first_pos, run_at, rebase_to, device=None, max_iter=16):
    """first_pos: [g, h0, h1] where rank 0's shard starts.  run_at(pos) -> end runs this rank's shard from `pos`;
    rebase_to(pos) -> end moves the finished run to `pos`.  Returns (start, end, iterations) of this rank."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    start = [int(x) for x in first_pos]
    end = run_at(start)
    it = 1
    while True:
        # one collective per iteration: every rank's draw counts AND the start it ran from, so that each rank can tell for itself whether
        # any rank has to move (the ranks must agree on when to stop)
        every = all_gather_i64([e - s for e, s in zip(end, start)] + start, device)
        starts = [[int(first_pos[k]) + sum(every[p][k] for p in range(q)) for k in range(3)] for q in range(len(every))]
        if all(starts[q] == every[q][3:6] for q in range(len(every))):
            return start, end, it
        if it >= max_iter:
            raise RuntimeError("stream-order resolution did not converge")
        if starts[rank] != start:
            start = starts[rank]
            end = rebase_to(start)
        it += 1


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
