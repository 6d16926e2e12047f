"""ctypes bindings of the index + engine entry points (seam B1) of include/psvr_engine.h."""
import ctypes as C
import json

import numpy as np

from ._lib import EngineError, check, lib

ORI_DTYPE = np.dtype([("chr_id", "<i4"), ("ref_bg", "<u4"), ("read_bg", "<u4"), ("align_score", "<u4"),
                      ("mapq", "u1"), ("direction", "u1"), ("unmapped", "u1"), ("reserved", "u1")])
CAND_DTYPE = np.dtype([("align_score", "<u4"), ("chain_score", "<u4"), ("ref_bg", "<u4"), ("read_bg", "<u4"), ("chr_id", "<i4"), ("sv_id", "<i4"),
                       ("max_index", "<u4"), ("n_cigar", "<u4"), ("cigar_off", "<i8"), ("direction", "u1"), ("mapq", "u1"), ("reserved", "u1", 6)], align=True)
READ_DTYPE = np.dtype([("n_result", "<i4"), ("unmapped", "u1"), ("early_out", "u1"), ("is_str", "u1"), ("reserved", "u1"),
                       ("primary", "<i4"), ("secondary", "<i4"), ("has_mate", "<i4"), ("mate_chr_id", "<i4"), ("mate_ref_bg", "<u4"),
                       ("prim_sv_id", "<i4"), ("mate_sv_id", "<i4"), ("n_seed", "<u4", 2), ("reserved1", "<u4"), ("seed_hash", "<u8", 2), ("chain_hash", "<u8", 2),
                       ("cand", CAND_DTYPE, 12)], align=True)
HDR_DTYPE = np.dtype([("n_result", "<i4"), ("unmapped", "u1"), ("early_out", "u1"), ("is_str", "u1"), ("reserved", "u1"),
                      ("primary", "<i4"), ("secondary", "<i4"), ("has_mate", "<i4"), ("mate_chr_id", "<i4"), ("mate_ref_bg", "<u4"),
                      ("prim_sv_id", "<i4"), ("mate_sv_id", "<i4"), ("reserved2", "<i4"), ("cand_off", "<i8")], align=True)
PAIR_DTYPE = np.dtype([("max_score", "<i4"), ("cur_isize", "<i4"), ("proper", "<i4"), ("gain", "<i4"), ("max1", "<i4"), ("max2", "<i4")])


class AlnParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("match", "mismatch", "gap_open", "gap_ex", "gap_open2", "gap_ex2", "zdrop",
                                          "normal_read_length", "isize_min", "isize_max", "min_filter_score")]


class IndexView(C.Structure):
    _fields_ = [("ref_seq", C.c_void_p), ("n_ref_seq", C.c_uint64), ("seq", C.c_void_p), ("n_seq", C.c_uint64),
                ("seqf", C.c_void_p), ("n_seqf", C.c_uint64), ("pos", C.c_void_p), ("n_pos", C.c_uint64),
                ("posp", C.c_void_p), ("n_posp", C.c_uint64), ("hash", C.c_void_p), ("n_hash", C.c_uint64),
                ("kmer", C.c_void_p), ("n_kmer", C.c_uint64), ("off", C.c_void_p), ("n_off", C.c_uint64),
                ("chr_text", C.c_char_p), ("header_names", C.POINTER(C.c_char_p)), ("n_header", C.c_int32)]


def default_params(stat=None):
    p = AlnParams()
    lib().psvr_aln_params_default(C.byref(p))
    if stat is not None:  # STAT_len_min_mid_max of the first read (load_reads, read_realignment.cpp:134-148)
        p.normal_read_length, p.isize_min, p.isize_max = stat[0], stat[1], stat[3]
        p.min_filter_score = max(50, stat[0] * p.match * 2 - 80)
    return p


class Index:
    def __init__(self, arrays, header_names, device=0):
        """arrays: dict with ref_seq, seq, seqf, pos, posp, hash, off (uint64), kmer (uint32), chr (str)."""
        L = lib()
        L.psvr_index_device_bytes.restype = C.c_int64
        self._keep = {k: np.ascontiguousarray(v) for k, v in arrays.items() if k != "chr"}
        v = IndexView()
        for f, k in (("ref_seq", "ref_seq"), ("seq", "seq"), ("seqf", "seqf"), ("pos", "pos"), ("posp", "posp"), ("hash", "hash"), ("kmer", "kmer"), ("off", "off")):
            a = self._keep[k]
            setattr(v, f, a.ctypes.data)
            setattr(v, "n_" + f, a.shape[0])
        v.chr_text = arrays["chr"].encode()
        names = (C.c_char_p * len(header_names))(*[n.encode() for n in header_names])
        v.header_names, v.n_header = names, len(header_names)
        self.h = C.c_void_p()
        check(L.psvr_index_create(C.byref(v), device, C.byref(self.h)))
        self.device_bytes = int(L.psvr_index_device_bytes(self.h))
        self._keep = None  # host copies are no longer needed: the index lives in HBM

    @classmethod
    def from_device_tensors(cls, tensors, chr_text, header_names, device=0):
        """tensors: dict of torch tensors ON `device` (ref_seq, seq, seqf, pos, posp, hash, off: int64 views of the uint64 arrays; kmer:
        int32) -- e.g. what an RCCL broadcast delivered.  The index is built from them device to device."""
        L = lib()
        L.psvr_index_device_bytes.restype = C.c_int64
        self = cls.__new__(cls)
        v = IndexView()
        for f in ("ref_seq", "seq", "seqf", "pos", "posp", "hash", "kmer", "off"):
            t = tensors[f]
            setattr(v, f, t.data_ptr())
            setattr(v, "n_" + f, t.numel())
        v.chr_text = chr_text.encode()
        names = (C.c_char_p * len(header_names))(*[n.encode() for n in header_names])
        v.header_names, v.n_header = names, len(header_names)
        self.h = C.c_void_p()
        check(L.psvr_index_create_from_device(C.byref(v), device, C.byref(self.h)))
        self.device_bytes = int(L.psvr_index_device_bytes(self.h))
        self._keep = None
        return self

    def close(self):
        if self.h:
            lib().psvr_index_destroy(self.h)
            self.h = None


class Engine:
    def __init__(self, index, params=None):
        self.index = index
        self.h = C.c_void_p()
        self.params = params or default_params()
        check(lib().psvr_engine_create(index.h, C.byref(self.params), C.byref(self.h)))
        self.n_pairs = 0

    def upload(self, bases, base_off, ori):
        """bases: uint8 ASCII array; base_off: int64[2P+1]; ori: ORI_DTYPE[2P]."""
        self._b = np.ascontiguousarray(bases, dtype=np.uint8)
        self._o = np.ascontiguousarray(base_off, dtype=np.int64)
        self._r = np.ascontiguousarray(ori, dtype=ORI_DTYPE)
        self.n_pairs = (len(self._o) - 1) // 2
        check(lib().psvr_engine_upload(self.h, C.c_int64(self.n_pairs), self._b.ctypes.data_as(C.c_char_p), self._o.ctypes.data_as(C.c_void_p),
                                       self._r.ctypes.data_as(C.c_void_p)))

    def run(self, trace=False, stats=False, timing=False, stream=None):
        check(lib().psvr_engine_run(self.h, (1 if trace else 0) | (2 if stats else 0) | (4 if timing else 0), C.c_void_p(stream)))

    def set_stream_pos(self, pos):
        a = (C.c_int64 * 3)(*[int(x) for x in pos])
        check(lib().psvr_engine_set_stream_pos(self.h, a))

    def stream_end(self):
        a = (C.c_int64 * 3)()
        check(lib().psvr_engine_stream_end(self.h, a))
        return [int(x) for x in a]

    def rebase(self, pos, stream=None):
        a = (C.c_int64 * 3)(*[int(x) for x in pos])
        check(lib().psvr_engine_rebase(self.h, a, C.c_void_p(stream)))

    def stats(self):
        buf = C.create_string_buffer(8192)
        check(lib().psvr_engine_stats(self.h, buf, 8192))
        return json.loads(buf.value.decode())

    def download(self):
        P = self.n_pairs
        used = C.c_int64(0)
        rc = lib().psvr_engine_download(self.h, None, None, None, C.c_int64(0), C.byref(used))
        if rc not in (0, 6):
            check(rc)
        reads = np.zeros(2 * P, dtype=READ_DTYPE)
        pairs = np.zeros(P, dtype=PAIR_DTYPE)
        cig = np.zeros(int(used.value) + 1, dtype=np.uint32)
        check(lib().psvr_engine_download(self.h, reads.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p),
                                         C.c_int64(len(cig)), C.byref(used)))
        return reads, pairs, cig

    def download_compact(self, bufs=None):
        """psvr_engine_download_compact: (hdr[2P], pairs[P], cands[sum n_result], cigar words) -- the form a record writer needs.
        bufs: optional HostBuffers (page-locked)."""
        P = self.n_pairs
        L = lib()
        nc, nw = C.c_int64(0), C.c_int64(0)
        rc = L.psvr_engine_download_compact(self.h, None, None, None, C.c_int64(0), C.byref(nc), None, C.c_int64(0), C.byref(nw))
        if rc not in (0, 6):
            check(rc)
        if bufs is None:
            hdr, pairs = np.zeros(2 * P, dtype=HDR_DTYPE), np.zeros(P, dtype=PAIR_DTYPE)
            cands, cig = np.zeros(int(nc.value) + 1, dtype=CAND_DTYPE), np.zeros(int(nw.value) + 1, dtype=np.uint32)
        else:
            hdr, pairs, cands, cig = bufs.compact_views(P, int(nc.value) + 1, int(nw.value) + 1)
        check(L.psvr_engine_download_compact(self.h, hdr.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), cands.ctypes.data_as(C.c_void_p),
                                             C.c_int64(len(cands)), C.byref(nc), cig.ctypes.data_as(C.c_void_p), C.c_int64(len(cig)), C.byref(nw)))
        return hdr, pairs, cands[:int(nc.value)], cig[:int(nw.value)]

    def compact_sizes(self):
        """(candidates, CIGAR words) of the compact form (builds it on the device if it is not there yet)."""
        nc, nw = C.c_int64(0), C.c_int64(0)
        rc = lib().psvr_engine_download_compact(self.h, None, None, None, C.c_int64(0), C.byref(nc), None, C.c_int64(0), C.byref(nw))
        if rc not in (0, 6):
            check(rc)
        return int(nc.value), int(nw.value)

    @staticmethod
    def compact_layout(n_pairs, nc, nw):
        """Byte offsets of hdr | pairs | cands | cigar in a packed block (each section on a 64-byte boundary) and the block's size."""
        def up(x):
            return (x + 63) & ~63
        o_hdr = 0
        o_pairs = up(o_hdr + 2 * n_pairs * HDR_DTYPE.itemsize)
        o_cands = up(o_pairs + n_pairs * PAIR_DTYPE.itemsize)
        o_cig = up(o_cands + nc * CAND_DTYPE.itemsize)
        return o_hdr, o_pairs, o_cands, o_cig, up(o_cig + nw * 4)

    def compact_pack(self, base_ptr, cap):
        """The compact form as ONE block at base_ptr (host memory, or memory of the engine's device: psvr_engine_download_compact takes
        either) -- what the ordered gather of a one-process-per-GPU host sends to rank 0.  Returns (bytes used, [n_pairs, nc, nw])."""
        P = self.n_pairs
        nc, nw = self.compact_sizes()
        o_hdr, o_pairs, o_cands, o_cig, total = self.compact_layout(P, nc, nw)
        if total > cap:
            raise EngineError("compact_pack: block of %d bytes does not fit %d" % (total, cap))
        a, b = C.c_int64(0), C.c_int64(0)
        check(lib().psvr_engine_download_compact(self.h, C.c_void_p(base_ptr + o_hdr), C.c_void_p(base_ptr + o_pairs), C.c_void_p(base_ptr + o_cands), C.c_int64(nc + 1), C.byref(a),
                                                 C.c_void_p(base_ptr + o_cig), C.c_int64(nw + 1), C.byref(b)))
        return total, [P, nc, nw]

    def download_into(self, bufs):
        """Like download(), into the caller's page-locked buffers (HostBuffers): the copies run at the link's rate."""
        P = self.n_pairs
        used = C.c_int64(0)
        rc = lib().psvr_engine_download(self.h, None, None, None, C.c_int64(0), C.byref(used))
        if rc not in (0, 6):
            check(rc)
        reads, pairs, cig = bufs.views(P, int(used.value) + 1)
        check(lib().psvr_engine_download(self.h, reads.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), cig.ctypes.data_as(C.c_void_p),
                                         C.c_int64(len(cig)), C.byref(used)))
        return reads, pairs, cig

    def close(self):
        if self.h:
            lib().psvr_engine_destroy(self.h)
            self.h = None


class HostBuffers:
    """Page-locked result buffers from psvr_host_alloc, kept across batches (what a pipeline slot of the reference would own)."""

    def __init__(self):
        self._p = [None] * 7
        self._cap = [0] * 7

    def _get(self, i, nbytes):
        if nbytes > self._cap[i]:
            L = lib()
            L.psvr_host_alloc.restype = C.c_void_p
            L.psvr_host_alloc.argtypes = [C.c_size_t]
            L.psvr_host_free.argtypes = [C.c_void_p]
            if self._p[i]:
                L.psvr_host_free(self._p[i])
            want = nbytes + nbytes // 8
            self._p[i] = L.psvr_host_alloc(want)
            if not self._p[i]:
                raise EngineError("psvr_host_alloc(%d) failed" % want)
            self._cap[i] = want
        return (C.c_char * nbytes).from_address(self._p[i])

    def views(self, n_pairs, n_cig):
        reads = np.frombuffer(self._get(0, 2 * n_pairs * READ_DTYPE.itemsize), dtype=READ_DTYPE)
        pairs = np.frombuffer(self._get(1, max(1, n_pairs) * PAIR_DTYPE.itemsize), dtype=PAIR_DTYPE)[:n_pairs]
        cig = np.frombuffer(self._get(2, n_cig * 4), dtype=np.uint32)
        return reads, pairs, cig

    def compact_views(self, n_pairs, n_cand, n_cig):
        hdr = np.frombuffer(self._get(0, 2 * n_pairs * HDR_DTYPE.itemsize), dtype=HDR_DTYPE)
        pairs = np.frombuffer(self._get(1, max(1, n_pairs) * PAIR_DTYPE.itemsize), dtype=PAIR_DTYPE)[:n_pairs]
        cands = np.frombuffer(self._get(3, n_cand * CAND_DTYPE.itemsize), dtype=CAND_DTYPE)
        cig = np.frombuffer(self._get(2, n_cig * 4), dtype=np.uint32)
        return hdr, pairs, cands, cig

    def input_views(self, bases, base_off, ori):
        """Page-locked copies of a batch's input arrays (what a pipeline slot would parse its reads into)."""
        b = np.frombuffer(self._get(4, max(1, bases.nbytes)), dtype=np.uint8)[:len(bases)]
        o = np.frombuffer(self._get(5, base_off.nbytes), dtype=np.int64)
        r = np.frombuffer(self._get(6, max(1, ori.nbytes)), dtype=ORI_DTYPE)[:len(ori)]
        b[:], o[:], r[:] = bases, base_off, ori
        return b, o, r

    def close(self):
        for i in range(7):
            if self._p[i]:
                lib().psvr_host_free.argtypes = [C.c_void_p]
                lib().psvr_host_free(self._p[i])
                self._p[i], self._cap[i] = None, 0
