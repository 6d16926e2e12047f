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
ts = []
for i in range(8):
    eng.set_stream_pos([2, 0, 0]); eng.run()
    torch.cuda.synchronize(); t0 = time.time()
    eng.rebase([2 + 50000 * (i + 1), 40 * (i + 1), 30 * (i + 1)])
    torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
t0 = time.time()
for i in range(5): eng.set_stream_pos([2, 0, 0]); eng.run()
torch.cuda.synchronize(); run_ms = (time.time() - t0) / 5 * 1e3
print("rebase ms:", [round(t, 2) for t in ts], "run ms:", round(run_ms, 2), eng.stats().get("rounds"))
