#!/usr/bin/env python3
"""The DP kernels at `fc_sv`'s contig re-alignment shapes (SignalAssembly.hpp:418-420,463: 2/-10, 24+2k | 32+1k, w = zdrop = 132, query /
target 300-3100 bases) through the device-pointer C ABI (psvr_dp_plan_*): the ring kernel against the general (LDS) kernel
(PSVR_DP_NO_RING=1 in the environment sends the same plan through the latter).  Usage: python tools/dp_bench_wide.py [n] [lmin] [lmax]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ksw_cases import mat5  # noqa: E402
from pansvr_amd import ksw  # noqa: E402
from pansvr_amd._lib import check, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lmin = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lmax = int(sys.argv[3]) if len(sys.argv) > 3 else 3100
rng = np.random.RandomState(7)
qlen = rng.randint(lmin, lmax + 1, size=n).astype(np.int32)
tlen = np.clip(qlen + rng.randint(-60, 61, size=n), lmin, lmax).astype(np.int32)
q_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int64)
t_off = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
t = rng.randint(0, 4, size=int(t_off[-1])).astype(np.uint8)
q = rng.randint(0, 4, size=int(q_off[-1])).astype(np.uint8)
for i in range(n):  # query = the target with ~3 % substitutions, laid over its common prefix (length differences end up as an end gap)
    m = min(qlen[i], tlen[i])
    s = t[t_off[i]:t_off[i] + m].copy()
    k = rng.random_sample(m) < 0.03
    s[k] = (s[k] + 1 + rng.randint(3, size=int(k.sum()))) % 4
    q[q_off[i]:q_off[i] + m] = s
L = lib()
dev = torch.device("cuda:0")
dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
dqo, dto = torch.from_numpy(q_off[:-1].copy()).to(dev), torch.from_numpy(t_off[:-1].copy()).to(dev)
cig_off = (q_off[:-1] + t_off[:-1] + 2 * np.arange(n)).astype(np.int64)
ez_host = np.zeros(n, dtype=np.dtype([("f", "<i4", 12), ("cigar_off", "<i8")]))
ez_host["cigar_off"] = cig_off
dcig = torch.zeros(int(q_off[-1] + t_off[-1] + 2 * n + 16), dtype=torch.int32, device=dev)
band = 2 * 132 + 1
cells = float(np.minimum(qlen.astype(np.int64) * tlen, np.maximum(qlen, tlen).astype(np.int64) * band).sum())   # cells inside the band
res = {}
for ring in (1, 0):
    if ring:
        os.environ.pop("PSVR_DP_NO_RING", None)
    else:
        os.environ["PSVR_DP_NO_RING"] = "1"
    dez = torch.from_numpy(ez_host.view(np.uint8).reshape(-1).copy()).to(dev)
    p = ksw.make_params(5, mat5(2, 10), 24, 2, 32, 1, 132, 132, -1, 0)
    plan = C.c_void_p()
    check(L.psvr_dp_plan_create(0, C.c_int64(n), qlen.ctypes.data_as(C.c_void_p), tlen.ctypes.data_as(C.c_void_p), C.byref(p), 0, C.byref(plan)))
    L.psvr_dp_plan_workspace_bytes.restype = C.c_int64
    ws = torch.zeros(int(L.psvr_dp_plan_workspace_bytes(plan)) + 256, dtype=torch.uint8, device=dev)
    buf = C.create_string_buffer(4096)
    L.psvr_dp_plan_describe(plan, buf, 4096)

    def run():
        check(L.psvr_dp_plan_launch(plan, C.c_void_p(dq.data_ptr()), C.c_void_p(dqo.data_ptr()), C.c_void_p(dt.data_ptr()), C.c_void_p(dto.data_ptr()),
                                    C.c_void_p(dez.data_ptr()), C.c_void_p(dcig.data_ptr()), C.c_void_p(ws.data_ptr()), None))
    run()
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt_ = (time.time() - t0) / reps
    res[ring] = (dt_, dez.cpu().numpy().tobytes(), dcig.cpu().numpy().tobytes())
    print("%-14s %9.3f ms  %8.1f kproblems/s  %7.1f G band cells/s  workspace %.1f MB  [%s]" % ("ring kernel" if ring else "general kernel", dt_ * 1e3, n / dt_ / 1e3, cells / dt_ / 1e9,
                                                                                            ws.numel() / 1e6, buf.value.decode().strip()[:200]), flush=True)
    L.psvr_dp_plan_destroy(plan)
print("speed-up %.2fx; results identical: %s" % (res[0][0] / res[1][0], res[0][1] == res[1][1] and res[0][2] == res[1][2]))
