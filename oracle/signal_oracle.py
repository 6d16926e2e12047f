#!/usr/bin/env python3
"""TEST-ONLY restatement of the reference's `fc_signal` step in its name-sorted mode (BAM -> interleaved FASTQ with the
original alignment in the comment; PanSVgenerateVCF/getSignalRead.cpp:15-256,491-519 and getSignalRead.hpp:76-190,
helpers clib/bam_file.c:330-350,427-468,614-680,1031-1069).

PARITY UNPINNED: the reference's own build of this step needs htslib (unbuildable in this image: cram_io.c wants
<lzma.h>), and the reference ships no fixture for it.  This file and pansvr_amd/csrc/signal_step.h are two independent
restatements of the same source text; tests/test_signal.py compares them.  Only tests may import or run this.

usage: signal_oracle.py [-D] [-U] [-I max_tid] in.bam status_out header_out  > reads.fq
"""
import gzip
import struct
import sys

MATCH, MISMATCH, GO, GE, GO2, GE2 = 2, 12, 16, 1, 32, 0


def read_bam(path):
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    text = raw[8:8 + l_text].rstrip(b"\0")
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, off)[0]
    off += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", raw, off)[0]
        off += 4 + l_name + 4
    recs = []
    while off < len(raw):
        bs = struct.unpack_from("<i", raw, off)[0]
        tid, pos, l_qname, mapq, _bin, n_cig, flag, l_seq, mtid, mpos, isize = struct.unpack_from("<iiBBHHHiiii", raw, off + 4)
        p = off + 36
        name = raw[p:p + l_qname - 1].decode()
        p += l_qname
        cigar = list(struct.unpack_from("<%dI" % n_cig, raw, p))
        p += 4 * n_cig
        seq4 = raw[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        qual = raw[p:p + l_seq]
        p += l_seq
        aux = raw[p:off + 4 + bs]
        recs.append(dict(name=name, tid=tid, pos=pos, mapq=mapq, flag=flag, cigar=cigar, l_seq=l_seq, seq4=seq4, qual=qual, mtid=mtid, mpos=mpos, isize=isize,
                         tags=parse_aux(aux)))
        off += 4 + bs
    return text, recs


def parse_aux(aux):
    tags, p = {}, 0
    sizes = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "d": 8}
    fmts = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<i"}     # bam_aux2i of 'I' through an int32_t
    while p + 3 <= len(aux):
        tag, t = aux[p:p + 2].decode(), chr(aux[p + 2])
        p += 3
        if t in ("Z", "H"):
            e = aux.index(b"\0", p)
            val = aux[p:e].decode()
            p = e + 1
        elif t == "B":
            st, cnt = chr(aux[p]), struct.unpack_from("<I", aux, p + 1)[0]
            p += 5 + cnt * {"c": 1, "C": 1, "s": 2, "S": 2}.get(st, 4)
            val = None
        else:
            val = struct.unpack_from(fmts[t], aux, p)[0] if t in fmts else None
            p += sizes[t]
        if tag not in tags:                                                        # bam_aux_get returns the first occurrence
            tags[tag] = (t, val)
    return tags


def num_tag(r, tag):
    t = r["tags"].get(tag)
    return (True, t[1]) if t and t[0] in "cCsSiI" else (False, 0)


def str_tag(r, tag):
    t = r["tags"].get(tag)
    return t[1] if t and t[0] == "Z" else None


def primary(r):
    return not (r["flag"] & 0x100) and not (r["flag"] & 0x800)


class Stat:
    def __init__(self):
        self.reset()
        self.read_len, self.min_l2, self.max_l2 = -1, 0, 0

    def reset(self):
        self.isz, self.lens, self.total = [0] * 100000, [0] * 1000, 0

    def collect(self, r):
        i = abs(r["isize"])
        if 0 < i < 100000:
            self.isz[i] += 1
        if r["l_seq"] < 1000:
            self.lens[r["l_seq"]] += 1

    def global_stat(self):
        self.read_len, tot_len = -1, 0.0
        for i in range(1000):
            tot_len += i * self.lens[i]
            if self.lens[i] > 0.6 * self.total:
                self.read_len = i
                break
        if self.read_len == -1:
            self.read_len = int(tot_len / self.total)
        lim = int(struct.unpack("<f", struct.pack("<f", struct.unpack("<f", struct.pack("<f", 0.01))[0] * self.total))[0])   # float arithmetic
        self.min_l2 = self.max_l2 = 0
        s = 0
        for i in range(100000):
            s += self.isz[i]
            if s > lim:
                self.min_l2 = i
                break
        s = 0
        for i in range(99999, 0, -1):
            s += self.isz[i]
            if s > lim:
                self.max_l2 = i
                break


def score_by_cigar(r):
    score = gap = 0
    for c in r["cigar"]:
        op, ln = c & 0xf, c >> 4
        if op in (0, 7):
            score += ln * MATCH
        elif op in (1, 2, 4, 5):
            if op in (1, 2):
                gap += ln
            score -= min(GO + ln * GE, GO2 + ln * GE2)
    nm = num_tag(r, "NM")[1]
    score -= (MISMATCH + MATCH) * (nm - gap)
    return max(0, score)


def xa_number(r):
    if r["mapq"] > 0:
        return 0
    xa = str_tag(r, "XA")
    return 6 if xa is None else xa.count(";")


def fastq(r, comment):
    seq = []
    for i in range(r["l_seq"]):
        c = (r["seq4"][i >> 1] >> (0 if i & 1 else 4)) & 0xf
        if c in (1, 2, 4, 8, 15):
            seq.append({1: "A", 2: "C", 4: "G", 8: "T", 15: "N"}[c])
    qual = [(q + 33) & 0xff for q in r["qual"]]
    if not (r["flag"] & 4) and (r["flag"] & 16):
        n = r["l_seq"]
        if len(seq) == n:
            comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
            seq = [comp.get(ch, "N") for ch in reversed(seq)]
        for i in range(n // 2 + 1):                                  # the len/2 + 1 bound: the middle pair of an even length is swapped back
            if n == 0:
                break
            ri = n - 1 - i
            qual[i], qual[ri] = qual[ri], qual[i]
    return "@%s %s\n%s\n+\n%s\n" % (r["name"], comment, "".join(seq), bytes(qual).decode("latin-1"))


def main():
    a = sys.argv[1:]
    not_filter = discard = False
    max_tid = 24
    while a and a[0].startswith("-"):
        if a[0] == "-D":
            not_filter = True
        elif a[0] == "-U":
            discard = True
        elif a[0] == "-I":
            max_tid = int(a[1])
            a = a[1:]
        a = a[1:]
    path, status_fn, header_fn = a
    text, recs = read_bam(path)
    prim = [r for r in recs if primary(r)]
    st = Stat()
    for r in prim:
        st.total += 1
        if st.total == 100000:
            break
        st.collect(r)
    st.global_stat()
    mn, mx = st.min_l2, st.max_l2
    mid = (mn + mx) // 2
    with open(status_fn, "w") as f:
        f.write("%f_%d_%d_%d_%d_%d\n" % (0.0, st.read_len, mn, mx, mn, mx))
        for i in range(mn, mx):
            v = struct.unpack("<f", struct.pack("<f", st.isz[i] / struct.unpack("<f", struct.pack("<f", st.total + 1))[0]))[0]
            f.write("%f\n" % v)
    open(header_fn, "wb").write(text)
    read_len = st.read_len
    isize_max, isize_min = mx + 150, max(1, mn - 150)
    out = sys.stdout
    stat_written = False
    for k in range(0, len(prim) - 1, 2):
        b = [prim[k], prim[k + 1]]
        assert b[0]["name"] == b[1]["name"] and b[0]["flag"] & 0x40 and b[1]["flag"] & 0x80
        unm = [bool(x["flag"] & 4) for x in b]
        direction = [not (x["flag"] & 16) for x in b]
        lowq = [sum(1 for q in x["qual"] if q < ord("/")) for x in b]
        sl, sr = [0, 0], [0, 0]
        for i, x in enumerate(b):
            if x["cigar"]:
                f, l = x["cigar"][0], x["cigar"][-1]
                if f & 0xf in (4, 5):
                    sl[i] = f >> 4
                if l & 0xf in (4, 5):
                    sr[i] = l >> 4
        clip = [sl[i] + sr[i] for i in range(2)]
        indel_nm = [sum(c >> 4 for c in x["cigar"] if c & 0xf in (1, 2)) + num_tag(x, "NM")[1] for x in b]
        score = [score_by_cigar(x) for x in b]
        xa = [xa_number(x) for x in b]
        tid = [x["tid"] for x in b]
        isz = [x["isize"] for x in b]
        isize = abs(isz[0])
        if discard:
            min_score = (b[0]["l_seq"] + b[1]["l_seq"]) * MATCH - 4 * (MATCH + MISMATCH)
            if score[0] + score[1] >= min_score and isize != 0 and isize_min < isize < isize_max and tid[0] == tid[1] and tid[0] <= max_tid and tid[1] <= max_tid:
                continue
        if b[0]["pos"] > b[1]["pos"]:
            direction.reverse()
        if isize == b[0]["l_seq"] and isize == b[1]["l_seq"] and direction == [False, True]:
            direction.reverse()
        reason = ["%d_%d_%d_%d_%d_%d_%d_%d_%d_" % (tid[i], b[i]["pos"], sl[i], score[i], b[i]["mapq"], b[1 - i]["mapq"], xa[i], xa[1 - i], isize) for i in range(2)]
        fl = ["%s%s%s%s" % ("R" if b[i]["flag"] & 16 else "F", "Y" if unm[i] else "N", "Y" if indel_nm[i] > 8 else "N", "Y" if clip[i] > 10 else "N") for i in range(2)]
        for i in range(2):
            reason[i] += "%s_%s_" % (fl[i], fl[1 - i])
        for i in range(2):
            clip[i] -= lowq[i]
            if clip[i] < 0:
                lowq[i], clip[i] = -clip[i], 0
            lowq[i] >>= 1
            indel_nm[i] -= lowq[i]
            if indel_nm[i] < 0:
                lowq[i], indel_nm[i] = -indel_nm[i], 0
        ok = True
        if b[0]["mapq"] < 10 and b[1]["mapq"] < 10:
            ok = False
        if unm[0] or unm[1]:
            ok = False
        if isize > 1000:
            ok = False
        if direction != [True, False]:
            ok = False
        if indel_nm[0] + indel_nm[1] > 15:
            ok = False
        if clip[0] + clip[1] > 10:
            ok = False
        if tid[0] != tid[1] or tid[0] > max_tid or tid[1] > max_tid:
            ok = False
        if ok and not not_filter:
            continue
        if not stat_written:
            reason[0] += "STAT_%d_%d_%d_%d_" % (read_len, mn, mid, mx)
            stat_written = True
        for i, x in enumerate(b):
            reason[i] += "FLAG_%d_%d_CIGAR_" % (x["flag"], x["mapq"])
            reason[i] += "".join("%d%s" % (c >> 4, "MIDNSHP=XB"[c & 0xf]) for c in x["cigar"]) + "_"
            reason[i] += "MATE_%d_%d_%d_TAG_" % (x["mtid"], x["mpos"], x["isize"])
            for t in ("XA", "MC", "SA"):
                v = str_tag(x, t)
                if v is not None:
                    reason[i] += "%s:Z:%s_" % (t, v)
            has, nm = num_tag(x, "NM")
            if has:
                reason[i] += "NM:i:%d_" % nm
        out.write(fastq(b[0], reason[0]))
        out.write(fastq(b[1], reason[1]))


if __name__ == "__main__":
    main()
