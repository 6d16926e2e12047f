"""Batched DP through the C ABI (seam B2): mirrors the argument meaning of ksw_extd2_sse /
ksw_extz2_sse (reference src/kswlib/ksw2.h:57-64) for a list of problems."""
import ctypes as C

import numpy as np

from ._lib import Extz, KswParams, check, lib

EZ_FIELDS = ["max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar", "reach_end"]


def make_params(m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag):
    p = KswParams()
    p.m = m
    for i, x in enumerate(mat):
        p.mat[i] = x
    p.q, p.e, p.q2, p.e2 = q, e, q2, e2
    p.w, p.zdrop, p.end_bonus, p.flag = w, zdrop, end_bonus, flag
    return p


def ext_batch(queries, targets, params, variant="extd2", device=0):
    """queries/targets: lists of uint8 sequences (codes 0..m-1).  Returns a list of dicts with
    every ksw_extz_t field plus the CIGAR (list of uint32 len<<4|op)."""
    n = len(queries)
    qlen = np.array([len(x) for x in queries], dtype=np.int32)
    tlen = np.array([len(x) for x in targets], dtype=np.int32)
    q_off = np.zeros(n, dtype=np.int64)
    t_off = np.zeros(n, dtype=np.int64)
    if n:
        q_off[1:] = np.cumsum(qlen[:-1])
        t_off[1:] = np.cumsum(tlen[:-1])
    qcat = np.concatenate([np.asarray(x, dtype=np.uint8) for x in queries] + [np.zeros(1, np.uint8)])
    tcat = np.concatenate([np.asarray(x, dtype=np.uint8) for x in targets] + [np.zeros(1, np.uint8)])
    ez = (Extz * max(n, 1))()
    cap = int(qlen.sum() + tlen.sum() + 2 * n + 16)
    cig = np.zeros(cap, dtype=np.uint32)
    fn = lib().psvr_extd2_batch if variant == "extd2" else lib().psvr_extz2_batch
    check(fn(device, C.c_int64(n), qcat.ctypes.data_as(C.c_void_p), q_off.ctypes.data_as(C.c_void_p), qlen.ctypes.data_as(C.c_void_p),
             tcat.ctypes.data_as(C.c_void_p), t_off.ctypes.data_as(C.c_void_p), tlen.ctypes.data_as(C.c_void_p),
             C.byref(params), ez, cig.ctypes.data_as(C.c_void_p), C.c_int64(cap)))
    out = []
    for i in range(n):
        d = {f: int(getattr(ez[i], f)) for f in EZ_FIELDS}
        o = int(ez[i].cigar_off)
        d["cigar"] = [int(x) for x in cig[o:o + d["n_cigar"]]]
        out.append(d)
    return out
