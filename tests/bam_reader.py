"""Minimal independent BAM reader for the tests (SAM/BAM specification v1, sections 4.1-4.2): BGZF is a series of
gzip members, so the standard gzip module inflates it; records are decoded back to SAM text lines."""
import gzip
import struct

NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"


def check_bgzf(path):
    """Every member carries the BC extra field with its own size, and the file ends with the 28-byte EOF block."""
    data = open(path, "rb").read()
    off, n = 0, 0
    while off < len(data):
        assert data[off:off + 4] == b"\x1f\x8b\x08\x04", "not a BGZF member at %d" % off
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        assert data[off + 12:off + 16] == b"BC\x02\x00" and xlen == 6
        bsize = struct.unpack_from("<H", data, off + 16)[0] + 1
        isize = struct.unpack_from("<I", data, off + bsize - 4)[0]
        assert isize <= 0x10000
        off += bsize
        n += 1
    assert off == len(data)
    assert data[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), "missing BGZF EOF marker"
    return n


def read_bam(path, check_bin=True):
    """Returns (header_text, [(name, length)], [sam_line_fields]).  check_bin: the stored bin must be reg2bin of the record's interval."""
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].decode()
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, off)[0]
    off += 4
    refs = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, off)[0]
        name = raw[off + 4:off + 4 + l_name - 1].decode()
        l_ref = struct.unpack_from("<i", raw, off + 4 + l_name)[0]
        refs.append((name, l_ref))
        off += 8 + l_name
    recs = []
    while off < len(raw):
        bs = struct.unpack_from("<i", raw, off)[0]
        b = raw[off + 4:off + 4 + bs]
        off += 4 + bs
        tid, pos, l_rn, mapq, bin_, n_cig, flag, l_seq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
        p = 32
        qname = b[p:p + l_rn - 1].decode()
        p += l_rn
        cig = struct.unpack_from("<%dI" % n_cig, b, p)
        p += 4 * n_cig
        seq = "".join(NT16[(b[p + (i >> 1)] >> (4 if i % 2 == 0 else 0)) & 15] for i in range(l_seq))
        p += (l_seq + 1) // 2
        q = b[p:p + l_seq]
        p += l_seq
        qual = "*" if l_seq == 0 or q[0] == 0xff else "".join(chr(x + 33) for x in q)
        tags = []
        while p < len(b):
            tag, ty = b[p:p + 2].decode(), chr(b[p + 2])
            p += 3
            if ty in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
                v = struct.unpack_from(fmt, b, p)[0]
                p += struct.calcsize(fmt)
                tags.append("%s:i:%d" % (tag, v))
            elif ty == "A":
                tags.append("%s:A:%s" % (tag, chr(b[p])))
                p += 1
            elif ty == "f":
                tags.append("%s:f:%g" % (tag, struct.unpack_from("<f", b, p)[0]))
                p += 4
            elif ty in "ZH":
                e = b.index(b"\0", p)
                tags.append("%s:%s:%s" % (tag, ty, b[p:e].decode()))
                p = e + 1
            else:
                raise ValueError("tag type %r" % ty)
        rname = refs[tid][0] if tid >= 0 else "*"
        rnext = "*" if mtid < 0 else ("=" if mtid == tid else refs[mtid][0])
        cigar = "".join("%d%s" % (c >> 4, CIGAR_OPS[c & 15]) for c in cig) or "*"
        # reg2bin of the record's own interval (section 5.3)
        rlen = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8)) or 1
        assert not check_bin or bin_ == reg2bin(max(pos, 0), max(pos, 0) + rlen), (bin_, pos, rlen)
        recs.append([qname, str(flag), rname, str(pos + 1), str(mapq), cigar, rnext, str(mpos + 1), str(tlen), seq or "*", qual] + tags)
    return text, refs, recs


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0
