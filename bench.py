#!/usr/bin/env python3
"""bench.py -- signal reads realigned / sec on MI355X (BASELINE.json metric).

One "step" = one pass of the whole `aln` hot path (prep -> seeding -> chaining -> candidate
selection -> extension DP -> assembly -> pairing, incl. the speculative rand()-offset rounds) over one
batch of synthetic 150 bp read pairs that is already resident in HBM (configs[1]: 1 M pairs vs a
10 k-anchor SV reference).  With --gpus N every rank owns one GPU and one contiguous block of ONE input
(weak scaling: 1 M pairs per GPU; block r of the input is generated from seed 13 + r, the input is their
concatenation); read pairs are independent given the replicated index, so there is no data-path collective -- except
that the reference draws from one rand()/random_r sequence in input order: after its run every rank all-gathers three
integers (draws its block consumed) over RCCL and rebases to start where the previous rank ended (pansvr_amd/dist.py), so N
GPUs produce exactly the records of one `-t 1` pass over the whole input.  Time is max over ranks between barriers.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel, timed live with HIP events on its launch stream
  "cpu_baseline": the reference's own objects (oracle/_ref/ref_aln, kind "reference") or the oracle restatement
                  (kind "port") timed on this box's host cores on a bounded sample: -t 1 and -t <all cores>
  "parity_check": every rank's block (first pairs of it) against the reference objects started at that block's position
  "e2e":          wall of the drop-in command `panSVR aln` on the same workload (FASTQ in -> SAM / BAM out), phase split
  "pcie_inclusive": upload + run + compact download of one batch through the C ABI
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HEADER = "@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n"
# The team DP kernel's cell update as compiled (pansvr_amd/csrc/ksw_row_step.inc; profiles/r03e_sq_counters.md: 99 323 vector instructions per
# wavefront for 2 112 cells per lane-step mix = 47 per cell), priced with profiles/r01j_valu_op_rates.txt at ~3.0 SIMD-cycles per instruction
DP_OWN_CYCLES = 47 * 3.0


def align_seconds(err, key="ALIGN_SECONDS"):
    return max(float([l for l in err.split("\n") if l.startswith(key)][-1].split()[1]), 1e-9)


def pipelined_abi(aln, index, params, pin_in, n_batches=8, slots=3):
    """The rate a caller of the public C ABI sustains when it overlaps the three steps of a batch the way the reference's kt_pipeline
    overlaps load / align / write (src/clib/kthread.c:157-197): `slots` job slots, each an engine of its own on its own HIP queue,
    driven by its own host thread -- upload(N+1) | run(N) | download_compact(N-1).  The runs themselves are a chain (batch N+1 starts in
    the rand() / random_r streams where batch N ended), everything else overlaps.  All buffers are page-locked (psvr_host_alloc) and
    allocated, touched and sized before the clock starts.  Returns the timing and, per batch, a checksum of the downloaded records,
    which the caller compares with the serial path's."""
    import threading
    import zlib
    pb, po, pr = pin_in
    engs = [aln.Engine(index, params) for _ in range(slots)]
    outs = [aln.HostBuffers() for _ in range(n_batches)]
    state = {"turn": 0, "pos": [2, 0, 0], "err": None}
    cv = threading.Condition()
    res = [None] * n_batches
    done = [0.0] * n_batches              # when a batch's records were in the caller's buffers
    phase = [(0.0, 0.0, 0.0, 0.0)] * n_batches
    sub = [(0.0, 0.0, 0.0)] * n_batches
    # warm-up: every slot's engine allocates its buffers, every output buffer set is allocated and touched by one transfer
    for e in engs:
        e.upload(pb, po, pr)
        e.set_stream_pos([2, 0, 0])
        e.run()
    for k in range(n_batches):
        engs[k % slots].download_compact(outs[k])

    def worker(s):
        try:
            for k in range(s, n_batches, slots):
                e = engs[s]
                ta = time.time()
                e.upload(pb, po, pr)
                tb = time.time()
                with cv:
                    while state["turn"] != k and state["err"] is None:
                        cv.wait()
                    if state["err"] is not None:
                        return
                    pos = list(state["pos"])
                tc = time.time()
                e.set_stream_pos(pos)
                tc1 = time.time()
                e.run()
                tc2 = time.time()
                end = e.stream_end()
                td = time.time()
                sub[k] = (tc1 - tc, tc2 - tc1, td - tc2)
                with cv:
                    state["pos"], state["turn"] = end, k + 1
                    cv.notify_all()
                res[k] = e.download_compact(outs[k])
                done[k] = time.time()
                phase[k] = (tb - ta, tc - tb, td - tc, done[k] - td)     # upload, wait for the turn, run, download
        except Exception as ex:          # noqa: BLE001 -- surfaced below
            with cv:
                state["err"] = ex
                cv.notify_all()
    th = [threading.Thread(target=worker, args=(s,)) for s in range(slots)]
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.time() - t0
    if state["err"] is not None:
        for e in engs:
            e.close()
        raise state["err"]
    sums = [zlib.crc32(b"".join(a.tobytes() for a in r)) for r in res]
    # the serial path through the same entry points: one engine, batch after batch
    e = engs[0]
    pos, serial = [2, 0, 0], []
    hb = aln.HostBuffers()
    ts = time.time()
    for k in range(n_batches):
        e.upload(pb, po, pr)
        e.set_stream_pos(pos)
        e.run()
        pos = e.stream_end()
        serial.append(zlib.crc32(b"".join(a.tobytes() for a in e.download_compact(hb))))
    ts = time.time() - ts
    for x in engs:
        x.close()
    for o in outs:
        o.close()
    hb.close()
    phase = [p + q for p, q in zip(phase, sub)]
    return dt, sums, serial, ts, done, phase


def pipeline_probe(aln, index, params, pin_in, reps=12):
    """What a run costs beside another job slot's transfers (PSVR_BENCH_PIPE_PROBE=1): engine A's run() timed alone, beside a thread that
    uploads to engine B in a loop, beside one that downloads B's records in a loop, and beside both."""
    import threading
    pb, po, pr = pin_in
    a, b, c2 = aln.Engine(index, params), aln.Engine(index, params), aln.Engine(index, params)
    hb = aln.HostBuffers()
    for e in (a, b, c2):
        e.upload(pb, po, pr)
        e.run()
    b.download_compact(hb)
    out = {}
    for name, jobs in (("alone", ()), ("beside_uploads", ("up",)), ("beside_downloads", ("down",)), ("beside_both", ("up", "down"))):
        stop = threading.Event()

        def loop_up():
            while not stop.is_set():
                c2.upload(pb, po, pr)

        def loop_down():
            while not stop.is_set():
                b.download_compact(hb)
        th = [threading.Thread(target=loop_up if j == "up" else loop_down) for j in jobs]
        for t in th:
            t.start()
        time.sleep(0.05)
        t0 = time.time()
        for _ in range(reps):
            a.run()
        out[name] = round((time.time() - t0) / reps * 1e3, 2)
        stop.set()
        for t in th:
            t.join()
    # the first run after an upload against a repeat of it (same batch, same position): wall, and the kernels that differ most
    fr = {}
    for tag in ("first", "repeat"):
        if tag == "first":
            a.upload(pb, po, pr)
        t0 = time.time()
        a.run()
        fr[tag + "_ms"] = round((time.time() - t0) * 1e3, 2)
    a.upload(pb, po, pr)
    a.run(timing=True)
    k1 = {k: v["ms"] for k, v in a.stats()["kernels"].items()}
    a.run(timing=True)
    k2 = {k: v["ms"] for k, v in a.stats()["kernels"].items()}
    fr["kernels_first_minus_repeat_ms"] = {k: round(k1[k] - k2.get(k, 0.0), 3) for k in sorted(k1, key=lambda k: -(k1[k] - k2.get(k, 0.0)))[:6]}
    out["first_run_after_upload"] = fr
    for e in (a, b, c2):
        e.close()
    hb.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=1000000, help="read pairs per GPU (configs[1]: 1 M)")
    ap.add_argument("--anchors", type=int, default=10000)
    ap.add_argument("--cpu-pairs", type=int, default=300000, help="pairs of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--check-pairs", type=int, default=50000, help="pairs per rank whose engine records are compared with the reference objects")
    ap.add_argument("--no-ref-cpu", action="store_true", help="time only the oracle port even if oracle/_ref/ref_aln is present")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end `panSVR aln` leg")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the overlapped-ABI leg (pcie_inclusive.pipelined)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the configs[4] (250 bp / edge-2000) leg")
    ap.add_argument("--engines", type=int, default=int(os.environ.get("PSVR_BENCH_ENGINES", "1")),
                    help="engines per GPU: the rank's block is cut into that many contiguous sub-blocks, each run by its own engine on its own HIP queue.  Measured on "
                         "configs[1]: 1 engine 12.5 ms/step, 2 engines 16.5, 4 engines 27.6 -- the stages fill the chip by themselves, a second queue only adds the rebase rounds")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the multi-threaded CPU-baseline leg (0 = all cores, capped at the reference's 48)")
    ap.add_argument("--one-pass", action="store_true", help="profiling runs (rocprofv3 --pmc): the warm-up and timed steps only, none of the extra engine passes (per-kernel timing, "
                                                            "work counters, PCIe legs, e2e, cfg5): every kernel's counters then belong to exactly warmup + steps passes")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous, rank -> device selection and one barrier only (no GPU is touched): the launch contract, testable on CPU")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearse = os.environ.get("PSVR_BENCH_REHEARSE") == "1"      # all ranks share GPU 0 and exchange over gloo (single-GPU rehearsal of the N > 1 path)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the default group is gloo: the control plane (barriers, agreement on fallbacks) comes up wherever torch.distributed does; the RCCL
        # plane for the bulk data is created beside it and tried before anything depends on it (pansvr_amd/dist.py::data_plane)
        dist.init_process_group("gloo")
        if rehearse:
            local_rank = 0
    if args.dry_run:
        # what the driver's launch line must lead to: rank r of the node drives HIP device LOCAL_RANK, chosen before anything touches a GPU
        if world > 1:
            dist.barrier()
        print(json.dumps({"dry_run": True, "rank": rank, "world": world, "local_rank": local_rank, "device": local_rank, "master_addr": os.environ.get("MASTER_ADDR")}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # the engine library, the CLI (its `index` sub-command builds the bench index) and the checkers: built here if the tree is a
    # fresh checkout (no-ops otherwise); one rank builds, the others wait
    if rank == 0:
        from pansvr_amd import build as _build
        _build.build(force=False, verbose=False)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.barrier()

    import bench_data
    from pansvr_amd import aln

    # ---- workload (host side, before this process touches the GPU: the FASTQ writer forks)
    t_setup = time.time()
    shm_ok = os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (8 << 30)
    tmp = tempfile.mkdtemp(prefix="psvr_bench_r%d_" % rank, dir="/dev/shm" if shm_ok else None)
    anc = bench_data.make_anchors(args.anchors, seed=11)
    ix_arrays = bench_data.build_index_cli(anc, dense=True)      # `panSVR index`: byte-identical to the reference builder's files
    bases, base_off, ori, isize = bench_data.make_reads(anc, args.pairs, seed=13 + rank)     # block `rank` of the input
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "ref_aln")    # the reference's own aligner objects (oracle/Makefile)
    have_ref = os.path.exists(ref_exe) and not args.no_ref_cpu and shm_ok     # its loader reads the dense 2 GiB table: RAM-backed tmp only
    n_cpu = min(args.cpu_pairs, args.pairs) if rank == 0 else 0
    n_chk = min(args.check_pairs, args.pairs)
    n_e2e = args.pairs if (rank == 0 and world == 1 and not args.no_e2e) else 0
    n_fq = max(n_cpu, n_chk, n_e2e)
    fq = os.path.join(tmp, "block.fq")
    ncore = os.cpu_count() or 1
    if n_fq:
        # (forked formatting workers only in the single-process case, where nothing has touched the GPU yet)
        bench_data.write_fastq(fq, bases, base_off, ori, isize, n_pairs=n_fq, procs=min(16, ncore) if world == 1 else 1)
    with open(os.path.join(tmp, "header.sam"), "w") as f:
        f.write(HEADER)
    ix_small = {k: v for k, v in ix_arrays.items() if k != "hash"}
    # one index directory per node in RAM-backed storage: the sparse form for `panSVR aln` / aln_oracle, the dense table for ref_aln
    idx_dir = os.path.join("/dev/shm" if shm_ok else tempfile.gettempdir(), "psvr_bench_idx_%d" % os.getppid()) if world > 1 else os.path.join(tmp, "idx")
    if local_rank == 0 or world == 1:
        bench_data.write_index_dir(ix_small, idx_dir, dense_hash=ix_arrays["hash"] if have_ref else None)
    t_host = time.time() - t_setup

    torch.cuda.set_device(local_rank)
    from pansvr_amd import dist as pdist
    # (the one-GPU rehearsal shares a device between the ranks, which RCCL refuses: PSVR_BENCH_REHEARSE_TRY_RCCL=1 tries it all the same --
    # that is the fallback path of data_plane(), run on purpose)
    data_pg, xdev, plane_how = pdist.data_plane(world > 1 and (not rehearse or os.environ.get("PSVR_BENCH_REHEARSE_TRY_RCCL") == "1"))
    index_bytes = int(sum(v.nbytes for k, v in ix_arrays.items() if hasattr(v, "nbytes") and k != "hash_sparse"))
    index_bcast, index_how = None, "every rank uploads its own copy from host memory"
    index = None
    t_up = time.time()
    if world > 1:
        # rank 0 uploads the index once, the others receive the eight arrays through a broadcast (RCCL over xGMI; in the
        # one-GPU rehearsal gloo moves host copies instead) and build their index from device memory (psvr_index_create_from_device).
        # All ranks agree first that every rank is ready for it; any failure before the collective falls back to per-rank uploads.
        keys = ("ref_seq", "seq", "seqf", "pos", "posp", "hash", "kmer", "off")
        ok = 1
        try:
            tens = {}
            for k in keys:
                a = ix_arrays[k]
                host = torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a.view(np.int32))
                if rank == 0:
                    tens[k] = host.to("cuda") if xdev else host
                else:
                    tens[k] = torch.empty(host.shape, dtype=host.dtype, device="cuda" if xdev else "cpu")
            torch.cuda.synchronize()
        except Exception as ex:
            ok, tens = 0, {}
            print("[bench] rank %d: index broadcast not prepared: %r" % (rank, ex), file=sys.stderr)
        t_index_upload = time.time() - t_up                      # rank 0: the one host upload; others: allocation only
        flag = torch.tensor([ok])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)              # (gloo control plane)
        if int(flag.item()) == 1:
            dist.barrier()
            if xdev:
                torch.cuda.synchronize()
            tb = time.time()
            try:
                for k in keys:
                    dist.broadcast(tens[k], src=0, group=data_pg)
                if xdev:
                    torch.cuda.synchronize()
                index_bcast = round((time.time() - tb) * 1e3, 2)
                if not xdev:
                    tens = {k: v.to("cuda") for k, v in tens.items()}
                    torch.cuda.synchronize()
                index = aln.Index.from_device_tensors(tens, ix_arrays["chr"], ["chr1", "chr2"], device=local_rank)
            except Exception as ex:
                ok, index = 0, None
                print("[bench] rank %d: index broadcast failed: %r" % (rank, ex), file=sys.stderr)
            # every rank has its index, or every rank uploads its own (a rank left without one must not be the only one to fall back)
            flag = torch.tensor([1 if index is not None else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                index_how = "rank 0 uploads, %s broadcast of the eight arrays, every rank builds its index device to device" % ("RCCL" if xdev else "gloo (host copies)")
            else:
                if index is not None:
                    index.close()
                index, index_bcast = None, None
            del tens
    if index is None:
        t_up = time.time()
        index = aln.Index(ix_arrays, ["chr1", "chr2"], device=local_rank)
        t_index_upload = time.time() - t_up
    del ix_arrays
    K = max(1, args.engines)
    cuts = [args.pairs * j // K for j in range(K + 1)]
    engs = []
    for j in range(K):
        e = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
        lo, hi = cuts[j], cuts[j + 1]
        e.upload(bases[base_off[2 * lo]:base_off[2 * hi]], base_off[2 * lo:2 * hi + 1] - base_off[2 * lo], ori[2 * lo:2 * hi])
        engs.append(e)
    group = pdist.EngineGroup(engs)
    index_device_bytes = index.device_bytes
    t_setup = time.time() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_iters = []
    my_start = [2, 0, 0]
    phase = {}                                             # seconds per phase of the N > 1 step, summed over the timed steps
    gather = pdist.BlockGather(xdev, data_pg) if world > 1 else None
    gathered = None

    def pack(buf):
        return engs[0].compact_pack(buf.data_ptr(), buf.numel())

    def step():
        nonlocal my_start, gathered
        if world == 1 and K == 1:
            engs[0].run()                                  # (a run does not advance the engine's stream position: every step starts at the same place)
            return
        if world == 1:
            group.run_at([2, 0, 0])
            return
        my_start, _, it = pdist.resolve_stream_order([2, 0, 0], group.run_at, group.rebase_to, device=xdev, group=data_pg, times=phase)
        exchange_iters.append(it)
        # the ordered gather (the reference's output_results): every rank's block of compact records -> rank 0, in block order
        if K == 1:
            tg = time.time()
            nc, nw = engs[0].compact_sizes()
            gathered = gather.gather(aln.Engine.compact_layout(args.pairs, nc, nw)[4], pack)
            phase["gather"] = phase.get("gather", 0.0) + (time.time() - tg)

    for _ in range(args.warmup):
        step()
    barrier()
    phase.clear()
    del exchange_iters[:]
    t0 = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)            # (gloo control plane)
        dt = float(t.item())
    # what rank 0 holds after the last step's gather must be the ranks' own blocks, in rank order: every rank sums its block as 64-bit
    # words, rank 0 sums what it received (checker, outside the timed region)
    gather_check = None
    if world > 1 and K == 1:
        def words_sum(t, n):
            return int(t[:n - n % 8].view(torch.int64).sum().item())
        own = pdist.all_gather_i64([words_sum(gather.mine, gather.sizes[rank]), gather.sizes[rank]], xdev, data_pg)
        if rank == 0:
            bad = [r for r, (buf, n, meta) in enumerate(gathered) if n != own[r][1] or words_sum(buf, n) != own[r][0] or meta[0] != args.pairs]
            gather_check = {"blocks": world, "bytes": int(sum(n for _, n, _ in gathered)), "blocks_differing": len(bad)}
    reads_per_step = 2 * args.pairs * world
    value = reads_per_step * args.steps / dt

    if args.one_pass:
        if rank == 0:
            print(json.dumps({"one_pass": True, "value": round(value, 1), "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "engine_passes": args.steps + args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3)}), flush=True)
        for e in engs:
            e.close()
        index.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        shutil.rmtree(tmp, ignore_errors=True)
        if world > 1 and local_rank == 0:
            shutil.rmtree(idx_dir, ignore_errors=True)
        return

    # ---- every rank's block against the reference objects, started where the block starts in the draw streams (checker only)
    parity_local = None
    if n_chk > 0 and (have_ref or os.path.exists(os.path.join(ROOT, "oracle", "aln_oracle"))):
        import aln_common as ac
        lens = np.diff(base_off)
        got = []
        for j, e in enumerate(engs):                      # the sub-blocks in order = the block
            lo, hi = cuts[j], cuts[j + 1]
            if lo >= n_chk:
                break
            reads_o, pairs_o, cig_o = e.download()
            part = ac.engine_records(reads_o, pairs_o, cig_o, ori[2 * lo:2 * hi], lens[2 * lo:2 * hi], 0, min(hi, n_chk) - lo)
            for q, rec in enumerate(part):
                rec["i"] = lo + q
            got += part
            del reads_o, pairs_o, cig_o
        base = [idx_dir, fq, os.path.join(tmp, "header.sam")]
        if have_ref:
            cmd = [ref_exe, "-t", "1", "-R", str(n_chk)] + base + ["--quiet", "--stream-pos", ",".join(str(x) for x in my_start)]
            against = "oracle/_ref/ref_aln (reference objects)"
        else:
            cmd = [os.path.join(ROOT, "oracle", "aln_oracle")] + base + ["--limit", str(n_chk)]
            against = "oracle/aln_oracle"
            if world > 1 and rank > 0:
                cmd = None           # the port has no --stream-pos: only rank 0's block can be replayed
        if cmd:
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout.decode()
            want = [json.loads(l) for l in out.split("\n") if l.lstrip().startswith("{")][:n_chk]
            for i, w in enumerate(want):
                w["i"] = i
            parity_local = [n_chk, sum(1 for i in range(n_chk) if got[i] != want[i]), against]
    parity = None
    if world > 1:
        cnt = pdist.all_gather_i64([parity_local[0] if parity_local else 0, parity_local[1] if parity_local else 0], xdev, data_pg)
        if rank == 0:
            parity = {"pairs_checked": sum(c[0] for c in cnt), "pairs_differing": sum(c[1] for c in cnt), "ranks_checked": sum(1 for c in cnt if c[0]),
                      "against": parity_local[2] if parity_local else None, "per_rank": cnt}
    elif parity_local:
        parity = {"pairs_checked": parity_local[0], "pairs_differing": parity_local[1], "ranks_checked": 1, "against": parity_local[2]}

    # two extra passes (not timed above) on ONE engine over the whole block: per-kernel HIP-event durations on the launch stream, then
    # the work counters (their atomics would distort the timings)
    group_rebases = group.rebases
    if K > 1:
        for e in engs:
            e.close()
        eng = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
        eng.upload(bases, base_off, ori)
        eng.run()
    else:
        eng = engs[0]
    eng.run(timing=True)
    kern = eng.stats()["kernels"]
    eng.run(stats=True)
    st = eng.stats()
    dom = max(kern, key=lambda k: kern[k]["ms"])
    # the boundary hands over host buffers: one extra, separately reported pass incl. H2D of the batch and D2H of the results
    # (compact form: headers + the candidates / CIGAR words that exist), first from pageable memory ...
    torch.cuda.synchronize()
    tp = time.time()
    eng.upload(bases, base_off, ori)
    eng.set_stream_pos([2, 0, 0])      # a new upload continues the reference's rand() streams; this is the same first batch again
    eng.run()
    out_host = eng.download_compact()
    tp = time.time() - tp
    pcie = {"reads_per_s": round(2 * args.pairs / tp, 1), "ms": round(tp * 1e3, 2),
            "h2d_bytes": int(bases.nbytes + base_off.nbytes + ori.nbytes), "d2h_bytes": int(sum(a.nbytes for a in out_host)),
            "note": "upload + run + compact download (psvr_engine_download_compact) of one batch through the C ABI from pageable host memory; not `value`"}
    del out_host
    # ... then with buffers a pipeline slot keeps page-locked across batches (psvr_host_alloc; allocated before the clock starts), inputs included
    hb = aln.HostBuffers()
    eng.download_compact(hb)
    pin_in = aln.HostBuffers()
    pb, po, pr = pin_in.input_views(bases, base_off, ori)
    torch.cuda.synchronize()
    tp = time.time()
    eng.upload(pb, po, pr)
    eng.set_stream_pos([2, 0, 0])
    eng.run()
    eng.download_compact(hb)
    tp = time.time() - tp
    pcie["page_locked"] = {"reads_per_s": round(2 * args.pairs / tp, 1), "ms": round(tp * 1e3, 2)}
    hb.close()
    if rank == 0 and world == 1 and not args.no_pipeline:
        # ... and the overlapped rate: three job slots, upload(N+1) | run(N) | download_compact(N-1) through the same entry points
        try:
            nb = int(os.environ.get("PSVR_BENCH_PIPE_BATCHES", "12"))
            slots = int(os.environ.get("PSVR_BENCH_PIPE_SLOTS", "3"))
            pdt, psums, ssums, sdt, done, phase = pipelined_abi(aln, index, aln.default_params((150, 200, 400, 600)), (pb, po, pr), n_batches=nb, slots=slots)
            done = sorted(done)
            steady = (done[-1] - done[1]) / (nb - 2) if nb > 3 else pdt / nb       # between the hand-overs of the second and the last batch: no fill of the pipeline
            pcie["pipelined"] = {"reads_per_s": round(2 * args.pairs * nb / pdt, 1), "ms_per_batch": round(pdt / nb * 1e3, 2), "batches": nb, "slots": slots,
                                 "sustained_reads_per_s": round(2 * args.pairs / steady, 1), "sustained_ms_per_batch": round(steady * 1e3, 2),
                                 "equal_to_serial": psums == ssums, "serial_ms_per_batch_incl_checksum": round(sdt / nb * 1e3, 2),
                                 "phase_ms_mean": dict(zip(("upload", "wait_turn", "run", "download", "run:set_stream_pos", "run:run", "run:stream_end"), [round(1e3 * sum(p[i] for p in phase[2:]) / max(len(phase) - 2, 1), 2) for i in range(7)])),
                                 "note": "%d batches of the bench batch back to back (the draw streams continue from batch to batch), %d engines on their own HIP queues driven by %d host threads: "
                                         "upload(N+1) | run(N) | download_compact(N-1); page-locked buffers allocated and touched before the clock starts; the records of every batch == the serial path's "
                                         "(crc32).  reads_per_s: all batches over the whole wall, the first upload and the last download included; sustained_*: the interval between hand-overs once "
                                         "the pipeline is full" % (nb, slots, slots)}
            if os.environ.get("PSVR_BENCH_PIPE_PROBE"):
                pcie["pipelined"]["run_ms_probe"] = pipeline_probe(aln, index, aln.default_params((150, 200, 400, 600)), (pb, po, pr))
        except Exception as ex:          # noqa: BLE001
            pcie["pipelined"] = {"error": repr(ex)[:300]}
    pin_in.close()

    roofline, cpu, e2e = None, None, None
    if rank == 0:
        bytes_per_read, own_bytes = None, None
        base = [idx_dir, fq, os.path.join(tmp, "header.sam")]
        if n_cpu > 0:
            # algorithmic bytes per read, counted by the oracle on a part of the CPU sample; the executables report the
            # wall of their per-pair loop on stderr (index load excluded)
            n_port = min(n_cpu, 60000)
            exe = os.path.join(ROOT, "oracle", "aln_oracle")
            r = subprocess.run([exe] + base + ["--stats", "--limit", str(n_port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
            cs = json.loads([l for l in r.stdout.decode().split("\n") if l.startswith("{")][-1])
            bytes_per_read = cs["bytes"]["total"] / (2.0 * n_port)
            # each kernel's OWN share of the algorithmic bytes (SURVEY 8(d) terms, per read): bases in -> prep; hash probes + MEM hits -> seeding;
            # unipath positions -> chaining; reference windows -> walk + fetch; ksw_extz_t + CIGAR out -> the DP kernels; candidate records -> assembly + tails
            b = {k: v / (2.0 * n_port) for k, v in cs["bytes"].items()}
            own_bytes = {"k_prep": b["read"], "k_seed": b["probe"] + b["hit"], "k_chain": b["pos"], "k_walk+k_dp_fetch": b["ref"], "extd2_*": b["dp_out"], "k_assemble+k_finalize_pair": b["cand"]}
            port_rate = round(2 * n_port / align_seconds(r.stderr.decode()), 1)
            cpu = {"value": port_rate, "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": "first %d pairs of the same workload, oracle/aln_oracle (scalar C++ restatement, -t 1 equivalent), index load excluded" % n_port}
            if have_ref:
                nt = args.cpu_threads or min(48, ncore)
                r1 = subprocess.run([ref_exe, "-t", "1", "-S", "-o", os.path.join(tmp, "r1.sam"), "-p", os.path.join(tmp, "r1o.sam"), "-R", str(n_cpu)] + base + ["--quiet"],
                                    stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True).stderr.decode()
                rn = subprocess.run([ref_exe, "-t", str(nt), "-S", "-o", os.path.join(tmp, "rn.sam"), "-p", os.path.join(tmp, "rno.sam"), "-R", str(n_cpu)] + base + ["--quiet"],
                                    stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True).stderr.decode()
                cpu = {"value": round(2 * n_cpu / align_seconds(rn), 1), "unit": "reads/s", "cores": nt, "kind": "reference",
                       "value_1_thread": round(2 * n_cpu / align_seconds(r1), 1),
                       "whole_command": {"reads_per_s_%d_threads" % nt: round(2 * n_cpu / align_seconds(rn, "TOTAL_SECONDS"), 1), "reads_per_s_1_thread": round(2 * n_cpu / align_seconds(r1, "TOTAL_SECONDS"), 1),
                                         "note": "load_reads (kseq) + kt_for + sam_format1 text of both files, index load excluded"},
                       "port_value": port_rate, "host_cores": ncore,
                       "sample": "first %d pairs of the same workload through the reference's own fc_aln objects (oracle/_ref/ref_aln: read_realignment / deBGA_index / graph / "
                                 "ksw2_extd2_sse / htslib sam_parse1, compiled from the reference tree): `value` = the reference's kt_for over align_read_pair (incl. output_BAM) at -t %d, "
                                 "value_1_thread the same at -t 1; port_value = oracle/aln_oracle on %d pairs" % (n_cpu, nt, n_port)}
        # which term of the algorithmic bytes each timed kernel owns
        groups = {"k_prep": ["k_prep", "k_prep_pair", "k_prep_mate1"], "k_seed": ["k_seed"], "k_chain": ["k_chain", "k_chain_select", "k_chain_small"], "k_walk+k_dp_fetch": ["k_walk", "k_dp_fetch"],
                  "k_assemble+k_finalize_pair": ["k_assemble", "k_finalize_pair"], "extd2_*": [k for k in kern if k.startswith("extd2_")]}
        group_of = {k: g for g, names in groups.items() for k in names}
        launches = max(1, kern[dom]["launches"])
        avg_ms = kern[dom]["ms"] / launches
        # HBM-side traffic per kernel from the committed rocprofv3 --pmc passes of this same command on the tree named in the file
        # (profiles/pmc_latest.json, written by tools/pmc_to_json.py; FETCH_SIZE / WRITE_SIZE are KiB per dispatch)
        pmc, pmc_src = {}, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pj.get("pairs_per_gpu") == args.pairs:
                for k, v in pj["kernels"].items():
                    b = k.split("<")[0]
                    pmc[b] = pmc.get(b, 0) + int((v["fetch_KiB_per_step"] + v["write_KiB_per_step"]) * 1024)
                pmc_src = "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `%s`, tree %s)" % (pj.get("command", "bench.py --one-pass"), pj.get("commit", "?"))
        except Exception:
            pmc = {}

        def traffic_of(names):
            t = [pmc[b] for b in set(n.split("<")[0] for n in names) if b in pmc]
            return int(sum(t)) if t else None
        if own_bytes is not None:
            # The dominant kernel's roofline: ITS OWN algorithmic bytes (the term of SURVEY 8(d)'s sum it is responsible for) x the reads one
            # launch covers / its average launch duration (HIP events on the launch stream, this run).  Reads per launch: a per-mate
            # kernel (seeding, chaining) covers the P reads of one mate per launch, the others all 2P; the few reads of the re-run rounds
            # ride in launches that are counted (they lower the average a little, never raise it).
            g = group_of.get(dom)
            share = own_bytes.get(g) if g else None
            if share is not None:
                alg_bytes = share * 2 * args.pairs                  # per step, all launches of the kernel together
                achieved = alg_bytes / (kern[dom]["ms"] * 1e-3) / 1e9
                step_ms = dt / args.steps * 1e3
                whole = bytes_per_read * 2 * args.pairs / (step_ms * 1e-3) / 1e9
                tr = traffic_of([dom])
                roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6),
                            "traffic": tr, "traffic_over_algorithmic": round(tr / alg_bytes, 2) if tr else None,
                            "traffic_note": "raw FETCH_SIZE+WRITE_SIZE (KiB*1024) of this kernel per step; gfx950 FETCH_SIZE under-reports coalesced reads by up to 2x", "traffic_source": pmc_src,
                            "alg_bytes_per_read": round(share, 1), "alg_bytes_per_launch": round(alg_bytes / launches, 1), "alg_bytes_per_read_whole_path": round(bytes_per_read, 1),
                            "kernel_ms_per_step": round(kern[dom]["ms"], 4), "kernel_launches_per_step": launches, "avg_launch_ms": round(avg_ms, 4),
                            "whole_step": {"achieved": round(whole, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(whole / 8000.0, 6), "ms": round(step_ms, 3),
                                           "traffic": int(sum(pmc.values())) if pmc else None,
                                           "traffic_over_algorithmic": round(sum(pmc.values()) / (bytes_per_read * 2 * args.pairs), 2) if pmc else None,
                                           "note": "the whole path's algorithmic bytes over the whole step (per GPU): the step is latency / issue bound"}}
        if roofline is not None and own_bytes is not None:
            # per-kernel table, measured live (HIP events of the timed pass): own algorithmic bytes, time, GB/s, fraction of the HBM peak, counter
            # traffic / algorithmic; for the DP kernels also cells/s against two vector-issue bounds (profiles/r01j_valu_op_rates.txt prices a
            # wavefront instruction at ~2.3 SIMD-cycles (add / sub / logic / right shift / mov) or ~4.2 (max / min / compare / left shift /
            # three-operand / DPP / v_pk_*)):
            #   own stream: the cell update of ksw_row_step.inc as compiled (see DP_INSTR below) -- how close the sweep runs to the price of its own instructions
            #   recurrence: the dual-affine cell as few instructions as the recurrence allows (2 x max3 for z + direction, 4 state updates of
            #               add + max, 2 differences, 1 score lookup, 1 H add, 1 direction store share: ~16 instructions, half of the dearer kind)
            DP_INSTR = {"own_stream_cycles_per_64_cells": DP_OWN_CYCLES, "recurrence_cycles_per_64_cells": 16 * 3.25}
            issue_bound = 1024 * 2.4e9 * 64 / DP_INSTR["own_stream_cycles_per_64_cells"]
            issue_bound_min = 1024 * 2.4e9 * 64 / DP_INSTR["recurrence_cycles_per_64_cells"]
            rows = []
            for g, names in groups.items():
                ms = sum(kern[k]["ms"] for k in names if k in kern)
                if ms <= 0:
                    continue
                gbs = own_bytes[g] * 2 * args.pairs / (ms * 1e-3) / 1e9
                tr = traffic_of(names)       # (every kernel of the group the counters saw: the engine times the preparation kernels under one name, the counters name three)
                row = {"kernels": g, "own_alg_bytes_per_read": round(own_bytes[g], 1), "ms_per_step": round(ms, 4), "achieved_GBps": round(gbs, 2), "frac_of_hbm_peak": round(gbs / 8000.0, 6),
                       "traffic": tr, "traffic_over_algorithmic": round(tr / (own_bytes[g] * 2 * args.pairs), 2) if tr and own_bytes[g] > 0 else None}
                if g == "extd2_*":
                    row["cells_per_s"] = round(st["dp_cells"] / (ms * 1e-3), 1)
                    row["valu_issue_bound_cells_per_s"] = round(issue_bound, 1)
                    row["frac_of_issue_bound"] = round(st["dp_cells"] / (ms * 1e-3) / issue_bound, 4)
                    row["frac_of_recurrence_bound"] = round(st["dp_cells"] / (ms * 1e-3) / issue_bound_min, 4)
                    row["issue_model"] = DP_INSTR
                    row["note"] = "sum over all DP launches of a step as if they ran one after the other (timed pass); integer DP is vector-issue bound, the byte fraction says little"
                rows.append(row)
            rows.append({"kernels": "others (" + ", ".join(sorted(k for k in kern if not any(k in v for v in groups.values()))) + ")",
                         "ms_per_step": round(sum(v["ms"] for k, v in kern.items() if not any(k in n for n in groups.values())), 4), "own_alg_bytes_per_read": 0.0})
            roofline["kernels"] = rows
        # ---- the drop-in command end to end: FASTQ file in, both record files out (index load reported separately)
        if n_e2e:
            cli = os.path.join(ROOT, "pansvr_amd", "bin", "panSVR")
            nt = min(48, ncore)                      # the reference's own thread limit (read_realignment.hpp:121)
            e2e = {"pairs": n_e2e, "threads": nt, "input": "FASTQ of the bench batch in RAM-backed storage (%.2f GB)" % (os.path.getsize(fq) / 1e9)}
            # SAM text; BAM at zlib's default level (what htslib's "wb" -- the reference's output -- uses); BAM at level 1 (--compress-level 1); BAM with the BGZF blocks
            # from the built-in encoder on the host threads (--bgzf-fast); ... compressed on the GPU (--bgzf-device)
            for key, mode, ext in (("sam", ["-S"], "sam"), ("bam", [], "bam"), ("bam_level1", ["--compress-level", "1"], "bam"), ("bam_fast", ["--bgzf-fast"], "bam"), ("bam_device", ["--bgzf-device"], "bam")):
                for fn in ("o." + ext, "p." + ext):      # (a 0.9 GB file opened for writing again is first emptied: 0.1 s of the next run's wall)
                    if os.path.exists(os.path.join(tmp, fn)):
                        os.remove(os.path.join(tmp, fn))
                r = subprocess.run([cli, "aln", "-t", str(nt)] + mode + ["-o", os.path.join(tmp, "o." + ext), "-p", os.path.join(tmp, "p." + ext)] + base, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                if r.returncode != 0:
                    e2e[key] = {"error": r.stderr.decode()[-300:]}
                    continue
                j = json.loads([l for l in r.stderr.decode().split("\n") if "e2e_json" in l][-1].split("e2e_json ", 1)[1])
                e2e[key] = {"reads_per_s": round(2 * n_e2e / j["wall_s"], 1), "wall_s": j["wall_s"], "index_load_s": j["index_s"], "read_parse_s": j["read_parse_s"], "engine_s": j["engine_s"],
                            "format_s": j["format_s"], "write_s": j["write_s"], "teardown_s": j.get("teardown_s"), "batches": j["batches"], "d2h_bytes": j["d2h_bytes"], "out_bytes": os.path.getsize(os.path.join(tmp, "o." + ext))}
            e2e["note"] = "wall_s = first FASTQ byte to both files closed (four overlapped stages: read+parse | engine | format | write); index_load_s (files -> HBM) before it and teardown_s (HBM given back) after it are outside it"
    # ---- configs[4] beside it (not `value`): 250 bp reads against edge-2000 anchors, the shape whose DP problems are several hundred
    # anti-diagonals wide, so the wavefront-per-alignment kernels get a timing and a roofline of their own
    cfg5 = None
    if rank == 0 and world == 1 and not args.no_cfg5:
        eng.close(), index.close()
        eng = index = None
        try:
            anc5 = bench_data.make_anchors(2000, seed=17, edge=2000, allele=(60, 2000))
            ix5 = bench_data.build_index_cli(anc5, dense=True)
            b5, o5, r5, _ = bench_data.make_reads(anc5, args.pairs, seed=19, L=250, frag=(400, 700), maxindel=40)
            index5 = aln.Index(ix5, ["chr1", "chr2"], device=local_rank)
            del ix5
            eng5 = aln.Engine(index5, aln.default_params((250, 400, 550, 700)))
            eng5.upload(b5, o5, r5)
            eng5.run()
            torch.cuda.synchronize()
            t5 = time.time()
            for _ in range(3):
                eng5.run()
            torch.cuda.synchronize()
            t5 = (time.time() - t5) / 3
            eng5.run(timing=True)
            k5 = eng5.stats()["kernels"]
            eng5.run(stats=True)
            s5 = eng5.stats()
            dom5 = max(k5, key=lambda k: k5[k]["ms"])
            wide = {k: round(v["ms"], 4) for k, v in k5.items() if k.startswith("extd2_")}
            # algorithmic bytes of a DP kernel: query + target bytes in, ksw_extz_t + CIGAR words out; per problem ~ qlen + tlen + 40 + 4 * n_cigar.
            # Counted from the run: dp_cells / dp_problems gives the mean shape; direction bytes are scratch and not counted.
            cfg5 = {"workload": "configs[4]: %d synthetic 250 bp PE pairs vs 2000 anchors with 2 kbp edges (%.1f Mbp), indels up to 40" % (args.pairs, len(anc5["codes"]) / 1e6),
                    "reads_per_s": round(2 * args.pairs / t5, 1), "ms_per_step": round(t5 * 1e3, 3), "dominant_kernel": dom5,
                    "rounds": s5.get("rounds"), "pair_runs": s5.get("pair_runs"), "pair_only_runs": s5.get("pair_only_runs"),
                    "dp_problems": s5["dp_problems"], "dp_cells": s5["dp_cells"], "dp_kernels_ms": wide,
                    "kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in sorted(k5.items(), key=lambda kv: -kv[1]["ms"])}}
            dp_ms = sum(wide.values())
            if dp_ms > 0 and s5["dp_problems"]:
                # counted by the engine: the query + target bytes its DP launches read (dp_seq_bytes); out: one ksw_extz_t (56 B) per problem and
                # its CIGAR words (<= the candidates' merged CIGAR words + one per problem, counted from the download)
                _, _, cig5 = eng5.download()
                dp_bytes = s5.get("dp_seq_bytes", 0) + s5["dp_problems"] * (56 + 4) + 4 * int(len(cig5))
                del cig5
                cfg5["dp_roofline"] = {"bound": "hbm", "kernels": "all extd2_* launches of a step", "achieved": round(dp_bytes / (dp_ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                                       "frac": round(dp_bytes / (dp_ms * 1e-3) / 1e9 / 8000.0, 6), "alg_bytes": int(dp_bytes), "cells_per_s": round(s5["dp_cells"] / (dp_ms * 1e-3), 1),
                                       "frac_of_issue_bound": round(s5["dp_cells"] / (dp_ms * 1e-3) / (1024 * 2.4e9 * 64 / DP_OWN_CYCLES), 4),
                                       "frac_of_recurrence_bound": round(s5["dp_cells"] / (dp_ms * 1e-3) / (1024 * 2.4e9 * 64 / (16 * 3.25)), 4),
                                       "note": "bytes counted by the engine (sequences in, ksw_extz_t + CIGAR out); integer DP is vector-issue bound: cells/s against 1024 SIMDs x 2.4 GHz x 64 lanes / the cycles of a cell update (the kernel's own instruction stream: %.0f; the recurrence's minimum: %.0f) is the figure to watch" % (DP_OWN_CYCLES, 16 * 3.25)}
            eng5.close(), index5.close()
        except Exception as ex:
            cfg5 = {"error": repr(ex)}
    multi = None
    if world > 1:
        # per-step phases of the N > 1 step, the slowest rank's clock for each (rank 0 never rebases; the step itself is max over ranks between barriers)
        keys = ("run", "rebase", "exchange", "gather")
        allp = pdist.all_gather_i64([int(phase.get(k, 0.0) * 1e9) for k in keys], xdev, data_pg)
        per = {k: round(max(r[i] for r in allp) / 1e9 / args.steps * 1e3, 3) for i, k in enumerate(keys)}
        multi = {"data_plane": plane_how, "run_ms": per.get("run"), "rebase_ms": per.get("rebase", 0.0), "exchange_ms": per.get("exchange"), "gather_ms": per.get("gather"),
                 "exchange_iters": max(exchange_iters) if exchange_iters else 0, "gather_check": gather_check,
                 "gather": "every rank's compact records (psvr_engine_download_compact into device memory) -> rank 0, point to point, in block order = input order; inside the timed step" if K == 1 else "not timed with --engines > 1"}
    failed = bool(parity and parity["pairs_differing"]) or bool(gather_check and gather_check["blocks_differing"])
    if rank == 0:
        line = {"metric": "signal reads realigned/sec (150 bp PE); bit-exact CIGAR vs CPU ref", "value": round(value, 1), "unit": "reads/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 (2-bit k-mers, 8-bit DP deltas, 32-bit scores)",
                "data": "synthetic", "config": {"workload": "configs[1]: %d synthetic 150 bp PE signal read pairs per GPU vs %d-anchor SV reference (%.1f Mbp), `panSVR aln` hot path"
                                                 % (args.pairs, args.anchors, len(anc["codes"]) / 1e6),
                                                 "pairs_per_gpu": args.pairs, "reads_per_step": reads_per_step,
                                                 "parallelism": "shard%d x %d engine(s) per GPU (index replicated, one input cut into contiguous blocks, draw-order exchange: %s)" % (world, K, "none" if world == 1 else "all-gather of 6 int64 per rank over %s, %d per step" % ("RCCL" if xdev else "gloo", max(exchange_iters) if exchange_iters else 0)),
                                                 "engines_per_gpu": K, "in_process_rebases": group_rebases,
                                                 "index_hbm_bytes": index_device_bytes, "index_upload_s": round(t_index_upload, 2), "index_broadcast_ms": index_bcast, "index_distribution": index_how,
                                                 "host_setup_s": round(t_host, 1), "setup_s": round(t_setup, 1)},
                "roofline": roofline, "cpu_baseline": cpu, "parity_check": parity, "multi_gpu": multi, "e2e": e2e, "pcie_inclusive": pcie, "cfg5": cfg5,
                "engine": {k: st[k] for k in ("rounds", "pair_runs", "pair_only_runs", "shadow_runs", "sensitive_pairs", "window_misses", "adopted_pairs", "dp_problems", "dp_seq_bytes", "stale_open", "candidates", "walk_pairs", "walk_us", "n_special", "special_const", "special_nomove", "probes", "hits", "seeds", "dp_cells", "hbm_used_bytes") if k in st},
                "kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])}}
        if failed:
            # a fast result that differs from the reference's is not a result: the line says so and the run fails
            line["error"] = "parity check failed: %s" % json.dumps({"parity_check": parity, "gather_check": gather_check})
        print(json.dumps(line), flush=True)
    if eng is not None:
        eng.close(), index.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    shutil.rmtree(tmp, ignore_errors=True)
    if world > 1 and local_rank == 0:
        shutil.rmtree(idx_dir, ignore_errors=True)
    if rank == 0 and failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
