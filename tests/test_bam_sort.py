"""f3: `panSVR sort` (pansvr_amd/csrc/bam_sort.h) in place of `samtools sort` + `samtools index` (panSVR_run.sh:53-54).
Checked with the independent reader of tests/bam_reader.py: the output holds the input's records in samtools' coordinate
order ((reference id as unsigned, position, strand), ties in input order) or name order; the .bai answers region queries --
bins from reg2bins, chunks read back through their virtual offsets, linear index as the lower bound -- with exactly the records
a scan of the file finds."""
import os
import struct
import subprocess
import zlib

import numpy as np

import bam_reader
import test_signal as ts

CLI = ts.CLI


def make_input(tmp_path, n=900):
    recs, refs = ts.make_pairs(4242, n)
    rng = np.random.RandomState(5)
    order = rng.permutation(len(recs))
    bam = str(tmp_path / "in.bam")
    ts.write_bam(bam, [recs[i] for i in order], refs)
    return bam


def sam_key(f):
    tid = 1 << 40 if f[2] == "*" else int(f[2][3:]) - 1
    return (tid, int(f[3]) - 1, int(f[1]) & 16)


def test_coordinate_sort_and_bai_index(tmp_path):
    bam = make_input(tmp_path)
    out = str(tmp_path / "sorted.bam")
    r = subprocess.run([CLI, "sort", "-t", "3", "-o", out, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert bam_reader.check_bgzf(out) >= 1
    _, refs0, recs0 = bam_reader.read_bam(bam, check_bin=False)
    text, refs, recs = bam_reader.read_bam(out)
    assert refs == refs0 and "SO:coordinate" in text.split("\n")[0]
    assert sorted(map(tuple, recs)) == sorted(map(tuple, recs0))
    keys = [sam_key(f) for f in recs]
    assert keys == sorted(keys)
    want = sorted(range(len(recs0)), key=lambda i: sam_key(recs0[i]))            # python's sort is stable: ties keep the input order
    assert [tuple(recs0[i]) for i in want] == [tuple(f) for f in recs]
    # ---- the index
    bai = open(out + ".bai", "rb").read()
    assert bai[:4] == b"BAI\x01"
    n_ref = struct.unpack_from("<i", bai, 4)[0]
    assert n_ref == len(refs)
    off, index = 8, []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", bai, off)[0]
        off += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", bai, off)
            off += 8
            bins[b] = [struct.unpack_from("<QQ", bai, off + 16 * k) for k in range(n_chunk)]
            off += 16 * n_chunk
        n_intv = struct.unpack_from("<i", bai, off)[0]
        lin = list(struct.unpack_from("<%dQ" % n_intv, bai, off + 4))
        off += 4 + 8 * n_intv
        index.append((bins, lin))
    n_no_coor = struct.unpack_from("<Q", bai, off)[0]
    assert off + 8 == len(bai)
    assert n_no_coor == sum(1 for f in recs if f[2] == "*")
    data = open(out, "rb").read()

    def region_query(tid, beg, end):
        bins, lin = index[tid]
        want_bins = [0]
        for shift, base in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
            want_bins += list(range(base + (beg >> shift), base + ((end - 1) >> shift) + 1))
        lo = lin[beg >> 14] if (beg >> 14) < len(lin) else (lin[-1] if lin else 0)
        found = []
        for b in want_bins:
            for vb, ve in bins.get(b, []):
                if ve <= lo:
                    continue
                # decode the chunk: inflate from the block of vb until ve is reached
                cpos, upos = vb >> 16, vb & 0xffff
                stream, cur_c = b"", cpos
                base_of = {}
                while cur_c <= (ve >> 16) and cur_c < len(data) - 28:
                    bsize = struct.unpack_from("<H", data, cur_c + 16)[0] + 1
                    base_of[cur_c] = len(stream)
                    stream += zlib.decompress(data[cur_c + 18:cur_c + bsize - 8], -15)
                    cur_c += bsize
                    if len(stream) - upos > 4 and cur_c > (ve >> 16):
                        # a record may spill into the next block: keep inflating while its end is missing
                        pass
                end_u = base_of.get(ve >> 16, len(stream)) + (ve & 0xffff) if (ve >> 16) in base_of else len(stream)
                p = upos
                while p < end_u:
                    while p + 4 > len(stream) or p + 4 + struct.unpack_from("<i", stream, p)[0] > len(stream):
                        bsize = struct.unpack_from("<H", data, cur_c + 16)[0] + 1
                        stream += zlib.decompress(data[cur_c + 18:cur_c + bsize - 8], -15)
                        cur_c += bsize
                    bs = struct.unpack_from("<i", stream, p)[0]
                    rtid, rpos, l_rn, _, _, n_cig = struct.unpack_from("<iiBBHH", stream, p + 4)
                    cig = struct.unpack_from("<%dI" % n_cig, stream, p + 36 + l_rn)
                    rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8)) or 1
                    name = stream[p + 36:p + 36 + l_rn - 1].decode()
                    flag = struct.unpack_from("<H", stream, p + 18)[0]
                    if rtid == tid and rpos < end and rpos + rlen > beg:
                        found.append((name, flag, rpos))
                    p += 4 + bs
        return sorted(found)

    def brute(tid, beg, end):
        out_ = []
        for f in recs:
            if f[2] == "*" or int(f[2][3:]) - 1 != tid:
                continue
            pos = int(f[3]) - 1
            rlen, num = 0, 0
            for ch in f[5]:
                if ch.isdigit():
                    num = num * 10 + int(ch)
                else:
                    rlen += num if ch in "MDN=X" else 0
                    num = 0
            rlen = rlen or 1
            if pos < end and pos + rlen > beg:
                out_.append((f[0], int(f[1]), pos))
        return sorted(out_)

    rng = np.random.RandomState(9)
    n_hits = 0
    for _ in range(60):
        tid = int(rng.choice([0, 0, 0, 1, 23, 24, 29]))
        beg = int(rng.randint(0, 1000000))
        end = beg + int(rng.choice([1, 200, 5000, 100000, 1000000]))
        got, want_ = region_query(tid, beg, end), brute(tid, beg, end)
        assert got == want_, (tid, beg, end, len(got), len(want_))
        n_hits += len(want_)
    assert n_hits > 500


def test_name_sort(tmp_path):
    bam = make_input(tmp_path, 400)
    out = str(tmp_path / "n.bam")
    r = subprocess.run([CLI, "sort", "-n", "-o", out, bam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    _, _, recs0 = bam_reader.read_bam(bam, check_bin=False)
    text, _, recs = bam_reader.read_bam(out)
    assert "SO:queryname" in text.split("\n")[0] and not os.path.exists(out + ".bai")
    assert sorted(map(tuple, recs)) == sorted(map(tuple, recs0))
    names = [f[0] for f in recs]
    assert names == sorted(names)
    for a, b in zip(recs, recs[1:]):
        if a[0] == b[0]:
            assert (int(a[1]) & 0xc0) <= (int(b[1]) & 0xc0)
    # and the signal step reads it: name order is its -N input
    r = subprocess.run([CLI, "signal", "-N", "-D", "-H", str(tmp_path / "h.sam"), "-S", str(tmp_path / "s.txt"), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout.count(b"\n") >= 8 * 300
