import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench_data
from pansvr_amd import aln
anc = bench_data.make_anchors(10000, seed=11)
ix = bench_data.build_index_cli(anc, dense=True)
bases, base_off, ori, isize = bench_data.make_reads(anc, 1000000, seed=13)
index = aln.Index(ix, ["chr1", "chr2"], device=0)
eng = aln.Engine(index, aln.default_params((150, 200, 400, 600)))
eng.upload(bases, base_off, ori)
eng.run(); eng.run()
torch.cuda.synchronize()
# (a) what a rank of the multi-GPU bench pays: the rand() stream's start moves by what the ranks before it drew (the same every step, so the
#     host's copy of the stream is long enough after the first time); this workload draws nothing from the random_r streams
# (b) all three streams move, each time further: the pairs that sampled positions (random_r) run in full, and the host extends its streams
ts, ts_all = [], []
end = eng.stream_end()
for i in range(9):
    eng.set_stream_pos([2, 0, 0]); eng.run()
    torch.cuda.synchronize(); t0 = time.time()
    eng.rebase([2 + 3 * (end[0] - 2), 0, 0])
    torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
    if i == 8:
        st = eng.stats()
        print("after run + rebase:", {k: st.get(k) for k in ("rounds", "pair_runs", "pair_only_runs", "shadow_runs", "sensitive_pairs", "adopted_pairs", "window_misses")})
for i in range(8):
    eng.set_stream_pos([2, 0, 0]); eng.run()
    torch.cuda.synchronize(); t0 = time.time()
    eng.rebase([2 + 50000 * (i + 1), 40 * (i + 1), 30 * (i + 1)])
    torch.cuda.synchronize(); ts_all.append((time.time() - t0) * 1e3)
t0 = time.time()
for i in range(5): eng.set_stream_pos([2, 0, 0]); eng.run()
torch.cuda.synchronize(); run_ms = (time.time() - t0) / 5 * 1e3
print("rebase to a fixed rand() position, ms (first: the host extends its stream):", [round(t, 2) for t in ts])
print("rebase of all three streams, further each time, ms:", [round(t, 2) for t in ts_all])
print("run ms:", round(run_ms, 2), "rounds", eng.stats().get("rounds"), "draws of the batch", [int(e - s) for e, s in zip(end, [2, 0, 0])])
