#!/usr/bin/env python3
"""bench.py -- signal reads realigned / sec on MI355X (BASELINE.json metric).

One "step" = one pass of the whole `aln` hot path (prep -> seeding -> chaining -> candidate
selection -> extension DP -> assembly -> pairing, incl. the speculative rand()-offset rounds) over one
batch of synthetic 150 bp read pairs that is already resident in HBM (configs[1]: 1 M pairs vs a
10 k-anchor SV reference).  With --gpus N every rank owns one GPU and an independent shard of the
same size (weak scaling, no data-path collective: read pairs are independent given the replicated
index); the shards form ONE input stream: after its run every rank all-gathers three integers (draws its shard consumed
from the reference's rand()/random_r sequences) and rebases to start where the previous rank ended
(pansvr_amd/dist.py), so N GPUs produce exactly the records of one `-t 1` pass over the concatenated shards.  Time is
max over ranks between barriers.

Prints ONE JSON line (rank 0) with the contract fields plus
  "roofline":     HBM roofline of the dominant kernel, timed live with HIP events on its launch stream
  "cpu_baseline": the oracle restatement (kind "port") timed on this box's host cores on a bounded sample.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1000000, help="read pairs per GPU (configs[1]: 1 M)")
    ap.add_argument("--anchors", type=int, default=10000)
    ap.add_argument("--cpu-pairs", type=int, default=60000, help="pairs of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--check-pairs", type=int, default=50000, help="pairs of the sample whose engine records are compared with the CPU run")
    ap.add_argument("--no-ref-cpu", action="store_true", help="time only the oracle port even if oracle/_ref/ref_aln is present")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PSVR_BENCH_REHEARSE=1: all ranks share GPU 0 and exchange over gloo (single-GPU rehearsal of the N>1 code path)
        rehearse = os.environ.get("PSVR_BENCH_REHEARSE") == "1"
        dist.init_process_group("gloo" if rehearse or not torch.cuda.is_available() else "nccl")
        if rehearse:
            local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    # the engine library, the CLI (its `index` sub-command builds the bench index) and the checkers: built here if the tree is a
    # fresh checkout (no-ops otherwise); one rank builds, the others wait
    if rank == 0:
        from pansvr_amd import build as _build
        _build.build(force=False, verbose=False)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    if world > 1:
        dist.barrier()
    xdev = "cuda" if (world > 1 and dist.get_backend() == "nccl") else None

    import bench_data
    from pansvr_amd import aln

    t_setup = time.time()
    anc = bench_data.make_anchors(args.anchors, seed=11)
    ix_arrays = bench_data.build_index_cli(anc, dense=True)      # `panSVR index`: byte-identical to the reference builder's files
    index = aln.Index(ix_arrays, ["chr1", "chr2"], device=local_rank)
    ix_sparse, ix_small = ix_arrays["hash_sparse"], {k: v for k, v in ix_arrays.items() if k != "hash"}
    ix_hash = ix_arrays["hash"] if rank == 0 and args.cpu_pairs > 0 else None
    del ix_arrays
    bases, base_off, ori, isize = bench_data.make_reads(anc, args.pairs, seed=13 + rank)
    eng = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
    eng.upload(bases, base_off, ori)
    t_setup = time.time() - t_setup

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    from pansvr_amd import dist as pdist

    exchange_iters = []

    def step():
        if world == 1:
            eng.run()
            return

        def run_at(pos):
            eng.set_stream_pos(pos)
            eng.run()
            return eng.stream_end()

        def rebase_to(pos):
            eng.rebase(pos)
            return eng.stream_end()

        _, _, it = pdist.resolve_stream_order([2, 0, 0], run_at, rebase_to, device=xdev)
        exchange_iters.append(it)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=xdev or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    reads_per_step = 2 * args.pairs * world
    value = reads_per_step * args.steps / dt

    # two extra passes (not timed above): per-kernel HIP-event durations on the launch stream, then the work
    # counters (their atomics would distort the timings)
    eng.run(timing=True)
    kern = eng.stats()["kernels"]
    eng.run(stats=True)
    st = eng.stats()
    dom = max(kern, key=lambda k: kern[k]["ms"])
    # the boundary hands over host buffers: one extra, separately reported pass incl. H2D of the batch and D2H of the results
    torch.cuda.synchronize()
    tp = time.time()
    eng.upload(bases, base_off, ori)
    eng.set_stream_pos([2, 0, 0])      # a new upload continues the reference's rand() streams; this is the same first batch again
    eng.run()
    out_host = eng.download()
    tp = time.time() - tp
    pcie = {"reads_per_s": round(2 * args.pairs / tp, 1), "ms": round(tp * 1e3, 2),
            "h2d_bytes": int(bases.nbytes + base_off.nbytes + ori.nbytes), "d2h_bytes": int(sum(a.nbytes for a in out_host)),
            "note": "upload (pageable host memory) + run + download of one batch through psvr_engine_upload/run/download; not `value`"}
    del out_host
    # the same with result buffers a pipeline slot keeps page-locked across batches (psvr_host_alloc; allocated before the clock starts)
    hb = aln.HostBuffers()
    eng.download_into(hb)
    torch.cuda.synchronize()
    tp = time.time()
    eng.upload(bases, base_off, ori)
    eng.set_stream_pos([2, 0, 0])
    eng.run()
    eng.download_into(hb)
    tp = time.time() - tp
    pcie["page_locked_results"] = {"reads_per_s": round(2 * args.pairs / tp, 1), "ms": round(tp * 1e3, 2)}
    hb.close()
    roofline, cpu, parity = None, None, None
    if rank == 0:
        # algorithmic bytes per read, counted by the oracle on the CPU sample below
        shm_ok = os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (6 << 30)
        tmp = tempfile.mkdtemp(prefix="psvr_bench_", dir="/dev/shm" if shm_ok else None)
        n_cpu = min(args.cpu_pairs, args.pairs)
        bytes_per_read, seed_bytes_per_read = None, None
        if n_cpu > 0:
            ref_exe = os.path.join(ROOT, "oracle", "_ref", "ref_aln")      # the reference's own aligner objects (oracle/Makefile)
            have_ref = os.path.exists(ref_exe) and not args.no_ref_cpu and shm_ok     # its loader reads the dense 2 GiB table: RAM-backed tmp only
            ix_small["hash_sparse"] = ix_sparse
            bench_data.write_index_dir(ix_small, os.path.join(tmp, "idx"), dense_hash=ix_hash if have_ref else None)
            bench_data.write_fastq(os.path.join(tmp, "sample.fq"), bases, base_off, ori, isize, n_pairs=n_cpu)
            with open(os.path.join(tmp, "header.sam"), "w") as f:
                f.write("@SQ\tSN:chr1\tLN:250000000\n@SQ\tSN:chr2\tLN:250000000\n")
            base = [os.path.join(tmp, "idx"), os.path.join(tmp, "sample.fq"), os.path.join(tmp, "header.sam")]

            def timed(cmd, out_path):
                # the executables report the wall of their per-pair loop on stderr (index load excluded)
                with open(out_path, "w") as fo:
                    err = subprocess.run(cmd, stdout=fo, stderr=subprocess.PIPE, check=True).stderr.decode()
                return max(float([l for l in err.split("\n") if l.startswith("ALIGN_SECONDS")][-1].split()[1]), 1e-9)

            exe = os.path.join(ROOT, "oracle", "aln_oracle")
            tc = timed([exe] + base + ["--stats", "--print"], os.path.join(tmp, "port.jsonl"))
            port_lines = open(os.path.join(tmp, "port.jsonl")).read().strip().split("\n")
            cs = json.loads(port_lines[-1])
            bytes_per_read = cs["bytes"]["total"] / (2.0 * n_cpu)
            seed_bytes_per_read = (cs["bytes"]["probe"] + cs["bytes"]["hit"] + cs["bytes"]["read"]) / (2.0 * n_cpu)
            port_rate = round(2 * n_cpu / tc, 1)
            cpu = {"value": port_rate, "unit": "reads/s", "cores": 1, "kind": "port",
                   "sample": "first %d pairs of the same workload, oracle/aln_oracle (scalar C++ restatement, -t 1 equivalent), index load excluded" % n_cpu}
            want = [json.loads(l) for l in port_lines[:-1] if l.startswith("{")][:n_cpu]
            against = "oracle/aln_oracle"
            if have_ref:
                tr = timed([ref_exe] + base, os.path.join(tmp, "ref.jsonl"))
                cpu = {"value": round(2 * n_cpu / tr, 1), "unit": "reads/s", "cores": 1, "kind": "reference", "port_value": port_rate,
                       "sample": "first %d pairs of the same workload through the reference's own aligner objects (oracle/_ref/ref_aln: read_realignment/deBGA_index/"
                                 "graph/ksw2_extd2_sse compiled from the reference tree, -t 1 code path, no BAM encode), index load excluded; port_value = oracle/aln_oracle on the same sample" % n_cpu}
                want = [json.loads(l) for l in open(os.path.join(tmp, "ref.jsonl")) if l.lstrip().startswith("{")][:n_cpu]
                against = "oracle/_ref/ref_aln (reference objects)"
            # the batch just timed, checked against the CPU run on that sample (checker only; nothing here is timed)
            import aln_common as ac
            n_chk = min(n_cpu, args.check_pairs)
            reads_o, pairs_o, cig_o = eng.download()
            lens = np.diff(base_off)
            got = ac.engine_records(reads_o, pairs_o, cig_o, ori, lens, 0, n_chk)
            differ = sum(1 for i in range(n_chk) if got[i] != want[i])
            parity = {"pairs_checked": n_chk, "pairs_differing": differ, "against": against}
            shutil.rmtree(tmp, ignore_errors=True)
        launches = max(1, kern[dom]["launches"])
        avg_ms = kern[dom]["ms"] / launches
        share = {"k_seed": seed_bytes_per_read}.get(dom, bytes_per_read)
        # HBM-side traffic of the dominant kernel from the committed rocprofv3 --pmc passes of this same command
        # (profiles/pmc_latest.json, written by tools/pmc_to_json.py; FETCH_SIZE/WRITE_SIZE are KiB per dispatch)
        traffic = None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            want = dom.replace(",hbm>", ", true>").replace(",lds>", ", false>")
            kd = pj["kernels"].get(want) or next((v for k, v in pj["kernels"].items() if k.split("<")[0] == want), None)
            if kd and pj.get("pairs_per_gpu") == args.pairs:
                traffic = int((kd["fetch_KiB_per_step"] + kd["write_KiB_per_step"]) * 1024)
        except Exception:
            traffic = None
        if share is not None:
            # first launch of the dominant kernel covers all 2P reads of the batch; later (speculative) launches cover few
            alg_bytes = share * 2 * args.pairs
            achieved = alg_bytes / (kern[dom]["ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6),
                        "traffic": traffic, "traffic_note": "raw FETCH_SIZE+WRITE_SIZE (KiB*1024) per step from profiles/pmc_latest.json; gfx950 FETCH_SIZE under-reports coalesced reads by up to 2x", "alg_bytes_per_read": round(share, 1), "alg_bytes_per_read_whole_path": round(bytes_per_read, 1),
                        "kernel_ms_per_step": round(kern[dom]["ms"], 4), "kernel_launches_per_step": launches, "avg_launch_ms": round(avg_ms, 4)}
    if rank == 0:
        line = {"metric": "signal reads realigned/sec (150 bp PE); bit-exact CIGAR vs CPU ref", "value": round(value, 1), "unit": "reads/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8/int32 (2-bit k-mers, 8-bit DP deltas, 32-bit scores)",
                "data": "synthetic", "config": {"workload": "configs[1]: %d synthetic 150 bp PE signal read pairs per GPU vs %d-anchor SV reference (%.1f Mbp), `panSVR aln` hot path"
                                                 % (args.pairs, args.anchors, len(anc["codes"]) / 1e6),
                                                 "pairs_per_gpu": args.pairs, "reads_per_step": reads_per_step, "parallelism": "shard%d (index replicated, draw-order exchange: %s)" % (world, "none" if world == 1 else "all-gather of 3 int64 per rank, %d iteration(s)/step" % (max(exchange_iters) if exchange_iters else 0)),
                                                 "index_hbm_bytes": index.device_bytes, "setup_s": round(t_setup, 1)},
                "roofline": roofline, "cpu_baseline": cpu, "parity_check": parity, "pcie_inclusive": pcie,
                "engine": {k: st[k] for k in ("rounds", "pair_runs", "pair_only_runs", "shadow_runs", "sensitive_pairs", "window_misses", "adopted_pairs", "dp_problems", "candidates", "probes", "hits", "seeds", "dp_cells")},
                "kernels_ms_per_step": {k: round(v["ms"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])}}
        print(json.dumps(line), flush=True)
    eng.close()
    index.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
