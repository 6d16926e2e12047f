#!/usr/bin/env python3
"""Micro-benchmark of the DP kernels through the device-pointer C ABI (psvr_dp_plan_*): typical extension problems
(q ~ U[qmin,qmax], t = q + 30, ~5 % divergence), with and without CIGAR (KSW_EZ_SCORE_ONLY skips the direction-byte
stores and the traceback).  Usage: python tools/dp_bench.py [n] [qmin] [qmax]"""
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
from pansvr_amd._lib import Extz, check, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
qmin = int(sys.argv[2]) if len(sys.argv) > 2 else 35
qmax = int(sys.argv[3]) if len(sys.argv) > 3 else 94
ext = int(sys.argv[4]) if len(sys.argv) > 4 else 30
rng = np.random.RandomState(5)
qlen = rng.randint(qmin, qmax + 1, size=n).astype(np.int32)
tlen = (qlen + ext).astype(np.int32)
q_off = np.concatenate([[0], np.cumsum(qlen)]).astype(np.int64)
t_off = np.concatenate([[0], np.cumsum(tlen)]).astype(np.int64)
t = rng.randint(0, 4, size=int(t_off[-1])).astype(np.uint8)
q = np.zeros(int(q_off[-1]), dtype=np.uint8)
for i in range(n):  # query = target prefix with ~5 % substitutions (and an occasional gap)
    s = t[t_off[i]:t_off[i] + qlen[i]].copy()
    m = rng.random_sample(qlen[i]) < 0.05
    s[m] = (s[m] + 1 + rng.randint(3, size=int(m.sum()))) % 4
    q[q_off[i]:q_off[i] + qlen[i]] = s
L = lib()
dev = torch.device("cuda:0")
dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
dqo, dto = torch.from_numpy(q_off[:-1].copy()).to(dev), torch.from_numpy(t_off[:-1].copy()).to(dev)
cig_off = (q_off[:-1] + t_off[:-1] + 2 * np.arange(n)).astype(np.int64)
ez_host = np.zeros(n, dtype=np.dtype([("f", "<i4", 12), ("cigar_off", "<i8")]))
ez_host["cigar_off"] = cig_off
dez = torch.from_numpy(ez_host.view(np.uint8).reshape(-1)).to(dev)
dcig = torch.zeros(int(q_off[-1] + t_off[-1] + 2 * n + 16), dtype=torch.int32, device=dev)
cells = float((qlen.astype(np.int64) * tlen).sum())
for flag, name in ((0, "with CIGAR"), (1, "score only")):
    p = ksw.make_params(5, mat5(2, 12), 16, 1, 32, 0, 200, 400, -1, flag)
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
    reps = 5
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt_ = (time.time() - t0) / reps
    print("%-11s %8.3f ms  %7.2f Mproblems/s  %7.1f GCUPS  workspace %.1f MB  [%s]" % (name, dt_ * 1e3, n / dt_ / 1e6, cells / dt_ / 1e9, ws.numel() / 1e6, buf.value.decode().strip()))
    L.psvr_dp_plan_destroy(plan)
