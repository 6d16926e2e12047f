"""-m gpu: entry points of include/psvr_engine.h called directly through ctypes (no CLI in between):
psvr_engine_align_batch (the call INTEGRATION.md tells a maintainer to make) against upload / run / download,
psvr_engine_download_compact against the fixed-size records, psvr_index_clone against the index it was copied from."""
import ctypes as C
import os

import numpy as np
import pytest

import aln_common as ac
import index_fixture
import synth
from test_emu_aln import normalise

pytestmark = pytest.mark.gpu


def _inputs(name="fx2", rname="reads150", limit=600):
    """bases / base_off / ori of the first `limit` pairs of a golden read set, parsed as the CLI's reader parses them."""
    w = ac.workdir(name)
    lines = open(os.path.join(w, rname + ".fq")).read().split("\n")
    from pansvr_amd import aln
    seqs, oris = [], []
    for k in range(0, 8 * limit, 4):
        cm = lines[k].split(" ", 1)[1]
        tok = [t for t in cm.split("_") if t][:10]
        seqs.append(lines[k + 1])
        oris.append((int(tok[0]), int(tok[1]) & 0xffffffff, int(tok[2]), int(tok[3]), int(tok[4]) & 0xff, 1 if tok[9][0] == "F" else 0, 1 if tok[9][1] == "Y" else 0, 0))
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    base_off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.int64)
    ori = np.array(oris, dtype=aln.ORI_DTYPE)
    stat = (150, 200, 400, 600)
    return bases, base_off, ori, stat


def _index(name="fx2", device=0):
    from pansvr_amd import aln
    names = [l.split("SN:")[1].split("\t")[0] for l in synth.header_text().split("\n") if l.startswith("@SQ")]
    return aln.Index(index_fixture.load_arrays(ac.index_dir(name)), names, device=device)


def test_align_batch_equals_upload_run_download_and_the_reference():
    from pansvr_amd import aln
    from pansvr_amd._lib import check, lib
    bases, base_off, ori, stat = _inputs()
    P = (len(base_off) - 1) // 2
    index = _index()
    # (1) upload / run / download
    e1 = aln.Engine(index, aln.default_params(stat))
    e1.upload(bases, base_off, ori)
    e1.run()
    r1, p1, c1 = e1.download()
    # (2) the one-call form, into caller buffers
    e2 = aln.Engine(index, aln.default_params(stat))
    r2, p2 = np.zeros(2 * P, dtype=aln.READ_DTYPE), np.zeros(P, dtype=aln.PAIR_DTYPE)
    c2 = np.zeros(len(c1) + 16, dtype=np.uint32)
    check(lib().psvr_engine_align_batch(e2.h, C.c_int64(P), bases.ctypes.data_as(C.c_char_p), base_off.ctypes.data_as(C.c_void_p), ori.ctypes.data_as(C.c_void_p),
                                        r2.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p), c2.ctypes.data_as(C.c_void_p), C.c_int64(len(c2)), 0))
    lens = np.diff(base_off)
    got = ac.engine_records(r1, p1, c1, ori, lens, 0, P)
    # (cigar_off values are positions in each engine's own arena: compare what they point at, not the offsets)
    assert p1.tobytes() == p2.tobytes() and got == ac.engine_records(r2, p2, c2, ori, lens, 0, P)
    # a too small CIGAR arena is an error, not a silent truncation
    rc = lib().psvr_engine_align_batch(e2.h, C.c_int64(P), bases.ctypes.data_as(C.c_char_p), base_off.ctypes.data_as(C.c_void_p), ori.ctypes.data_as(C.c_void_p),
                                       r2.ctypes.data_as(C.c_void_p), p2.ctypes.data_as(C.c_void_p), c2.ctypes.data_as(C.c_void_p), C.c_int64(8), 0)
    assert rc == 6
    # (3) and they are the reference's records
    want = [ac.strip_trace(l) for l in ac.golden_lines("fx2", "reads150")[:P]]
    assert got == want
    e1.close(), e2.close(), index.close()


def test_compact_download_holds_the_same_results_in_a_sixth_of_the_bytes():
    from pansvr_amd import aln
    bases, base_off, ori, stat = _inputs(limit=1500)
    P = (len(base_off) - 1) // 2
    index = _index()
    eng = aln.Engine(index, aln.default_params(stat))
    eng.upload(bases, base_off, ori)
    eng.run()
    reads, pairs, cig = eng.download()
    hdr, pairs2, cands, cig2 = eng.download_compact()
    assert pairs.tobytes() == pairs2.tobytes()
    assert int(hdr["n_result"].sum()) == len(cands) == int(reads["n_result"].sum())
    for f in ("n_result", "unmapped", "early_out", "is_str", "primary", "secondary", "has_mate", "mate_chr_id", "mate_ref_bg", "prim_sv_id", "mate_sv_id"):
        assert (hdr[f] == reads[f]).all(), f
    assert (hdr["cand_off"][hdr["n_result"] > 0] == (np.cumsum(hdr["n_result"]) - hdr["n_result"])[hdr["n_result"] > 0]).all()      # dense, in read order
    k = 0
    for r in range(2 * P):
        for i in range(int(hdr["n_result"][r])):
            a, b = reads["cand"][r][i], cands[k]
            for f in ("align_score", "chain_score", "ref_bg", "read_bg", "chr_id", "sv_id", "max_index", "n_cigar", "direction", "mapq"):
                assert a[f] == b[f], (r, i, f)
            n = int(a["n_cigar"])
            assert (cig[int(a["cigar_off"]):int(a["cigar_off"]) + n] == cig2[int(b["cigar_off"]):int(b["cigar_off"]) + n]).all()
            k += 1
    assert int(cands["n_cigar"].sum()) == len(cig2)                               # only the CIGAR words that exist
    full_bytes = reads.nbytes + pairs.nbytes + cig.nbytes
    compact_bytes = hdr.nbytes + pairs2.nbytes + cands.nbytes + cig2.nbytes
    assert compact_bytes * 4 < full_bytes, (compact_bytes, full_bytes)
    # page-locked buffers, second call on the same run (cached on the device)
    hb = aln.HostBuffers()
    h2, p2, c2, g2 = eng.download_compact(hb)
    assert h2.tobytes() == hdr.tobytes() and c2.tobytes() == cands.tobytes() and g2.tobytes() == cig2.tobytes()
    hb.close()
    eng.close(), index.close()


def test_index_clone_is_an_index():
    from pansvr_amd import aln
    from pansvr_amd._lib import check, lib
    bases, base_off, ori, stat = _inputs(limit=400)
    index = _index()
    h = C.c_void_p()
    check(lib().psvr_index_clone(index.h, 0, C.byref(h)))
    lib().psvr_index_device_bytes.restype = C.c_int64
    assert lib().psvr_index_device_bytes(h) == index.device_bytes
    clone = aln.Index.__new__(aln.Index)
    clone.h, clone.device_bytes = h, index.device_bytes
    outs = []
    for ix in (index, clone):
        eng = aln.Engine(ix, aln.default_params(stat))
        eng.upload(bases, base_off, ori)
        eng.run()
        outs.append(eng.download())
        eng.close()
    lens = np.diff(base_off)
    P = (len(base_off) - 1) // 2
    a, b = (ac.engine_records(r, p, c, ori, lens, 0, P) for r, p, c in outs)      # (arena positions differ from run to run: compare what they point at)
    assert a == b and outs[0][1].tobytes() == outs[1][1].tobytes()
    clone.close(), index.close()


def test_index_from_device_arrays_is_an_index():
    """psvr_index_create_from_device: the eight arrays already in device memory (as an RCCL broadcast leaves them on the ranks that did
    not upload: bench.py --gpus N) become an index that aligns like the one created from the host arrays.  In a process of its own:
    torch brings its own HIP runtime and has to initialise the device before the engine library does (the order bench.py keeps)."""
    import subprocess
    import sys
    code = """
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, %r); sys.path.insert(0, %r)
import aln_common as ac, index_fixture, synth
import test_abi_gpu as t
from pansvr_amd import aln
bases, base_off, ori, stat = t._inputs(limit=400)
index = t._index()
a = index_fixture.load_arrays(ac.index_dir("fx2"))
dev = torch.device("cuda:0")
tens = {k: torch.from_numpy(a[k].view(np.int64) if a[k].dtype == np.uint64 else a[k].view(np.int32)).to(dev) for k in ("ref_seq", "seq", "seqf", "pos", "posp", "hash", "kmer", "off")}
torch.cuda.synchronize()
names = [l.split("SN:")[1].split("\\t")[0] for l in synth.header_text().split("\\n") if l.startswith("@SQ")]
idx2 = aln.Index.from_device_tensors(tens, a["chr"], names, device=0)
assert idx2.device_bytes == index.device_bytes
del tens
outs = []
for ix in (index, idx2):
    eng = aln.Engine(ix, aln.default_params(stat))
    eng.upload(bases, base_off, ori)
    eng.run()
    outs.append(eng.download())
    eng.close()
lens = np.diff(base_off)
P = (len(base_off) - 1) // 2
x, y = (ac.engine_records(r, p, c, ori, lens, 0, P) for r, p, c in outs)
assert x == y and outs[0][1].tobytes() == outs[1][1].tobytes()
print("FROM_DEVICE_OK", P)
""" % (ac.ROOT, ac.HERE)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"FROM_DEVICE_OK 400" in r.stdout, r.stderr.decode()[-2000:]


def test_index_built_in_hbm_from_the_anchor_fasta_aligns_like_the_reference_built_one():
    """f1: psvr_index_build (host builder + first-level table expanded on the device, nothing written to disk) against the index
    the reference's deBGA built for the same anchors (tests/golden/fx2/idx): same HBM footprint, same records; and the CLI given
    the FASTA in place of <IndexDir> writes the reference's SAM files."""
    import gzip
    import subprocess
    import tempfile
    from pansvr_amd import aln
    from pansvr_amd._lib import check, lib
    w = ac.workdir("fx2")
    bases, base_off, ori, stat = _inputs(limit=800)
    index = _index()
    h = C.c_void_p()
    check(lib().psvr_index_build(os.path.join(w, "anchors.fa").encode(), os.path.join(w, "header.sam").encode(), 0, C.byref(h)))
    lib().psvr_index_device_bytes.restype = C.c_int64
    assert lib().psvr_index_device_bytes(h) == index.device_bytes
    built = aln.Index.__new__(aln.Index)
    built.h, built.device_bytes = h, index.device_bytes
    outs = []
    for ix in (index, built):
        eng = aln.Engine(ix, aln.default_params(stat))
        eng.upload(bases, base_off, ori)
        eng.run()
        outs.append(eng.download())
        eng.close()
    lens = np.diff(base_off)
    P = (len(base_off) - 1) // 2
    a, b = (ac.engine_records(r, p, c, ori, lens, 0, P) for r, p, c in outs)
    assert a == b
    built.close(), index.close()
    tmp = tempfile.mkdtemp(prefix="psvr_fa_")
    cli = os.path.join(ac.ROOT, "pansvr_amd", "bin", "panSVR")
    r = subprocess.run([cli, "aln", "-S", "-o", os.path.join(tmp, "o.sam"), "-p", os.path.join(tmp, "p.sam"), os.path.join(w, "anchors.fa"), os.path.join(w, "reads150.fq"),
                        os.path.join(w, "header.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-1500:]
    with gzip.open(os.path.join(ac.golden_dir("fx2"), "reads150.sam.gz"), "rb") as f:
        assert open(os.path.join(tmp, "o.sam"), "rb").read() == f.read()


def test_job_slots_of_one_process_continue_each_other():
    """Three engines of one process take turns, batch after batch, each starting in the draw streams where the one before stopped -- the
    pattern of a caller that overlaps upload / run / download over job slots (bench.py's pipelined leg, the reference's kt_pipeline).  The
    engines share the process-wide rand() stream (engine_core.h SharedRand) and take their batches as windows of ONE array (offsets that
    do not start at 0: what the device's pass over an uploaded batch, k_scan_batch, must cope with, unaligned read starts included).
    Their records, batch by batch, are the reference's for the whole input: the golden run is one stream over all pairs."""
    from pansvr_amd import aln
    bases, base_off, ori, stat = _inputs("fx2", "reads150", 900)      # fx2: N bases, unmapped / full-score originals, tied chains
    P = (len(base_off) - 1) // 2
    index = _index()
    engs = [aln.Engine(index, aln.default_params(stat)) for _ in range(3)]
    cuts = [0, 97, 98, 350, 351, 600, 899, 900]                        # ragged batches, one of a single pair
    lens = np.diff(base_off)
    got, pos = [], None
    for k in range(len(cuts) - 1):
        lo, hi = cuts[k], cuts[k + 1]
        e = engs[k % 3]
        e.upload(bases, base_off[2 * lo:2 * hi + 1], ori[2 * lo:2 * hi])
        if pos is not None:
            e.set_stream_pos(pos)
        e.run()
        pos = e.stream_end()
        r, p, c = e.download()
        recs = ac.engine_records(r, p, c, ori[2 * lo:2 * hi], lens[2 * lo:2 * hi], 0, hi - lo)
        for x in recs:
            x["i"] += lo
        got += recs
    want = [ac.strip_trace(l) for l in ac.golden_lines("fx2", "reads150")[:P]]
    assert got == want
    for e in engs:
        e.close()
    index.close()


def test_bgzf_members_from_the_device_decode_to_their_input():
    """psvr_bgzf_compress: every block (16 KB; PSVR_BGZF_BLOCK) of the input becomes one BGZF member on the device (a lane per block); the members,
    decoded one by one with zlib -- header fields, BSIZE, CRC32 and ISIZE checked by hand --, give back the input.  Input: BAM-like
    records, incompressible bytes (stored blocks), long runs, a last block of 1 byte."""
    import struct
    import zlib
    from pansvr_amd._lib import check, lib
    rng = np.random.RandomState(5)
    parts = [bytes(rng.randint(0, 256, size=200000, dtype=np.uint8)), b"\0" * 300000,
             b"".join(b"read%07d\tAS:i:%d\tXA:Z:chr%d,%d;\n" % (i, rng.randint(300), rng.randint(24), rng.randint(1 << 30)) for i in range(60000)),
             bytes(rng.randint(0, 4, size=400000, dtype=np.uint8) + 65), b"x"]
    data = b"".join(parts)
    blk = int(os.environ.get("PSVR_BGZF_BLOCK", "16384"))
    data += b"y" * ((-len(data)) % blk + 1)                         # the last block holds a single byte
    L = lib()
    L.psvr_bgzf_bound.restype = C.c_int64
    L.psvr_bgzf_bound.argtypes = [C.c_int64]
    cap = L.psvr_bgzf_bound(len(data))
    out = C.create_string_buffer(cap)
    got = C.c_int64(0)
    check(L.psvr_bgzf_compress(0, data, C.c_int64(len(data)), out, C.c_int64(cap), C.byref(got)))
    raw, pos, back, stored = out.raw[:got.value], 0, [], 0
    while pos < len(raw):
        assert raw[pos:pos + 4] == b"\x1f\x8b\x08\x04" and raw[pos + 10:pos + 16] == b"\x06\x00BC\x02\x00"
        bsize = struct.unpack("<H", raw[pos + 16:pos + 18])[0] + 1
        body = raw[pos + 18:pos + bsize - 8]
        crc, isize = struct.unpack("<II", raw[pos + bsize - 8:pos + bsize])
        d = zlib.decompressobj(-15)
        block = d.decompress(body) + d.flush()
        assert d.eof and not d.unused_data and len(block) == isize and zlib.crc32(block) == crc
        stored += (body[0] & 6) == 0
        back.append(block)
        pos += bsize
    assert b"".join(back) == data and all(len(b) == blk for b in back[:-1]) and len(back[-1]) == 1
    assert stored >= 3 and got.value < 0.75 * len(data)              # the random bytes are stored, the rest is compressed
