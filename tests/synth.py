"""Seeded synthetic inputs for the `aln` path: SV anchor FASTA, header.sam and an
interleaved FASTQ whose comment field follows fc_signal's wire format
(reference src/PanSVgenerateVCF/getSignalRead.cpp:158-247).  No reference data ships with panSVR,
so every fixture is generated here."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def revcomp(s):
    return _COMP[np.frombuffer(s, dtype=np.uint8)][::-1].tobytes()


def rand_dna(rng, n):
    return ACGT[rng.randint(0, 4, size=n)].tobytes()


def make_anchors(n_anchors, seed=7, edge=500, allele=(60, 300), str_frac=0.0, dup_frac=0.0, repeat_len=0, repeat_copies=1, repeat_spacer=24, repeat_anchors=None):
    """INS anchors named ID_chr_st_len_TYPE_bp1_bp2_end_vcfid (get_anchor_ref.hpp:322-323).
    str_frac: fraction of anchors whose allele is a short tandem repeat; dup_frac: fraction of anchors that
    re-use the previous anchor's left flank (identical flanks => tied chains); repeat_len: one element shared by every allele;
    repeat_copies > 1: every allele holds that many copies of it, `repeat_spacer` unique bases apart (n_anchors x repeat_copies
    positions of one unipath: beyond POS_N_MAX_LEVEL2 = 8000 expand_seed gives up, deBGA_index.cpp:224); repeat_anchors: only the
    first so many anchors hold the element."""
    rng = np.random.RandomState(seed)
    out = []
    prev_left = None
    repeat = rand_dna(rng, repeat_len) if repeat_len else b""    # the same element inside every allele: a high-copy unipath
    for i in range(n_anchors):
        st = 10000 * (i + 1)
        alen = rng.randint(allele[0], allele[1] + 1)
        left = rand_dna(rng, edge)
        if prev_left is not None and rng.random_sample() < dup_frac:
            left = prev_left
        if rng.random_sample() < str_frac:
            unit = rand_dna(rng, rng.randint(2, 7))
            al = (unit * (alen // len(unit) + 1))[:alen]
        else:
            al = rand_dna(rng, alen)
        if repeat_anchors is not None and i >= repeat_anchors:
            pass
        elif repeat_len and repeat_copies > 1:
            al = al[:alen // 2] + b"".join(repeat + rand_dna(rng, repeat_spacer) for _ in range(repeat_copies)) + al[alen // 2:]
        elif repeat_len:
            al = al[:alen // 2] + repeat + al[alen // 2:]
        right = rand_dna(rng, edge)
        seq = left + al + right
        prev_left = left
        name = "%d_chr1_%d_%d_INS_%d_%d_%d_sv.INS.%d" % (i, st, len(seq), st + edge, st + edge, st + 2 * edge, i)
        out.append((name, seq))
    return out


def header_text():
    """The ORIGINAL genome's header: 32 contigs, so that every tid the read generator emits (0, 1 and the decoy-like 30 of the
    `tid > 24 => unmapped` case) names a contig -- the reference indexes target_name[] with it unchecked."""
    names = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY", "chrM"] + ["decoy%d" % i for i in range(1, 8)]
    return "@HD\tVN:1.6\tSO:unsorted\n" + "".join("@SQ\tSN:%s\tLN:250000000\n" % n for n in names)


def _mutate(rng, s, kind, maxindel=8):
    b = bytearray(s)
    if kind == 0:      # 1-4 substitutions
        for _ in range(rng.randint(1, 5)):
            p = rng.randint(len(b))
            b[p] = b"ACGT"[(b"ACGT".index(b[p]) + 1 + rng.randint(3)) % 4] if b[p] in b"ACGT" else b[p]
    elif kind == 1:    # deletion 1..maxindel
        k = rng.randint(1, maxindel + 1)
        p = rng.randint(10, len(b) - 10 - k)
        del b[p:p + k]
    elif kind == 2:    # insertion 1..maxindel
        k = rng.randint(1, maxindel + 1)
        p = rng.randint(10, len(b) - 10)
        b[p:p] = rand_dna(rng, k)
    return bytes(b)


def make_reads(anchors, n_pairs, seed=13, L=150, frag=(300, 500), maxindel=8, miss_frac=0.2, n_frac=0.01,
               str_frac=0.02, stat=(150, 200, 400, 600), unmapped_frac=0.02, fullscore_frac=0.02, lengths=None, heavy_n_frac=0.0,
               center_frac=0.0, lower_frac=0.0):
    """lengths: optional list of read lengths drawn per read (ragged batch); heavy_n_frac: reads that get 4-8 N bases;
    center_frac: fragments centred on the middle of the anchor (where make_anchors puts the shared repeat);
    lower_frac: reads with a few lower-case bases and lower-case n (charToDna5n maps n to code 4, which spills a bit into the
    neighbouring base of the packed read words)."""
    Lmax = max(lengths) if lengths else L
    return _make_reads(anchors, n_pairs, seed, Lmax, frag, maxindel, miss_frac, n_frac, str_frac, stat, unmapped_frac, fullscore_frac, lengths, heavy_n_frac,
                       center_frac, lower_frac)


def _make_reads(anchors, n_pairs, seed, L, frag, maxindel, miss_frac, n_frac, str_frac, stat, unmapped_frac, fullscore_frac, lengths, heavy_n_frac, center_frac, lower_frac=0.0):
    """Returns a list of (name, comment, seq, qual) FASTQ records, two per pair (interleaved)."""
    rng = np.random.RandomState(seed)
    recs = []
    for p in range(n_pairs):
        miss = rng.random_sample() < miss_frac
        if miss:
            flen = rng.randint(frag[0], frag[1] + 1)
            fragment = rand_dna(rng, flen + 2 * maxindel + 2)
            st_pos, off = 10000 * (1 + rng.randint(len(anchors))), rng.randint(0, 500)
        else:
            a = rng.randint(len(anchors))
            name, seq = anchors[a]
            flen = min(rng.randint(frag[0], frag[1] + 1), len(seq) - 2 * maxindel - 2)
            off = rng.randint(0, len(seq) - flen - 2 * maxindel - 1)
            if center_frac and rng.random_sample() < center_frac:
                off = max(0, min(len(seq) - flen - 2 * maxindel - 2, len(seq) // 2 - rng.randint(40, L)))
            fragment = seq[off:off + flen + 2 * maxindel + 2]
            st_pos = int(name.split("_")[2])
        ends = []
        for e in range(2):
            kind = rng.choice(4, p=[0.3, 0.2, 0.2, 0.3])
            if e == 0:
                src = fragment[:L + maxindel]
            else:
                src = revcomp(fragment[flen - L - maxindel:flen])
            Lr = int(rng.choice(lengths)) if lengths else L
            r = _mutate(rng, src, kind, maxindel)[:Lr]
            if heavy_n_frac and rng.random_sample() < heavy_n_frac:
                b = bytearray(r)
                for _ in range(rng.randint(4, 9)):
                    b[rng.randint(len(b))] = ord("N")
                r = bytes(b)
            if lower_frac and rng.random_sample() < lower_frac:
                b = bytearray(r)
                for _ in range(rng.randint(1, 6)):
                    k = rng.randint(len(b))
                    b[k] = ord(chr(b[k]).lower())
                for _ in range(rng.randint(0, 3)):
                    b[rng.randint(len(b))] = ord("n")
                r = bytes(b)
            if rng.random_sample() < str_frac:
                unit = rand_dna(rng, rng.randint(2, 6))
                k = rng.randint(30, min(90, len(r) - 5))
                s0 = rng.randint(0, len(r) - k)
                r = r[:s0] + (unit * (k // len(unit) + 1))[:k] + r[s0 + k:]
            if rng.random_sample() < n_frac:
                b = bytearray(r)
                for _ in range(rng.randint(1, 3)):
                    b[rng.randint(len(b))] = ord("N")
                r = bytes(b)
            ends.append(r)
        swap = rng.random_sample() < 0.5         # which mate is "first in pair"
        isize = flen
        pos1 = st_pos + off                      # 0-based, as core.pos
        pos2 = st_pos + off + flen - L
        names = "r%07d" % p
        for k in range(2):
            e = k ^ int(swap)                     # fragment end this record holds
            fwd = e == 0
            unm = rng.random_sample() < unmapped_frac
            full = (not unm) and rng.random_sample() < fullscore_frac
            tid = 30 if unm else 0                # tid > 24 => treated as unmapped (read_realignment.cpp:413)
            Lk = len(ends[e])
            softl = 0 if full else 40
            score = 2 * Lk if full else 140
            pos = pos1 if fwd else pos2
            mpos = pos2 if fwd else pos1
            flag = (0x40 if k == 0 else 0x80) | 0x1 | (0 if fwd else 0x10) | (0x20 if fwd else 0)
            cigar = "%dM" % Lk if full else "40S%dM" % (Lk - 40)
            c = "%d_%d_%d_%d_20_20_0_0_%d_%sN%sY_%sNNY_" % (tid, pos, softl, score, isize, "F" if fwd else "R",
                                                           "N", "R" if fwd else "F")
            if p == 0 and k == 0 and stat is not None:
                c += "STAT_%d_%d_%d_%d_" % stat
            c += "FLAG_%d_20_CIGAR_%s_MATE_0_%d_%d_TAG_NM:i:3_" % (flag, cigar, mpos, isize if fwd else -isize)
            qual = bytes(33 + rng.randint(20, 41, size=Lk).astype(np.uint8))
            recs.append((names, c, ends[e].decode(), qual.decode()))
    return recs


def _sub(rng, b, p):
    b[p] = b"ACGT"[(b"ACGT".index(b[p]) + 1 + rng.randint(3)) % 4]


def _pair_records(rng, p, ends, st_pos, off, flen, stat, prefix="s"):
    """Two FASTQ records of one pair in fc_signal's wire format (original alignment: 40S<L-40>M, score 140)."""
    recs = []
    swap = rng.random_sample() < 0.5
    pos1, pos2 = st_pos + off, st_pos + off + max(0, flen - len(ends[1]))
    for k in range(2):
        e = k ^ int(swap)
        fwd = e == 0
        Lk = len(ends[e])
        pos, mpos = (pos1, pos2) if fwd else (pos2, pos1)
        flag = (0x40 if k == 0 else 0x80) | 0x1 | (0 if fwd else 0x10) | (0x20 if fwd else 0)
        c = "0_%d_40_140_20_20_0_0_%d_%sNNY_%sNNY_" % (pos, flen, "F" if fwd else "R", "R" if fwd else "F")
        if p == 0 and k == 0 and stat is not None:
            c += "STAT_%d_%d_%d_%d_" % stat
        c += "FLAG_%d_20_CIGAR_40S%dM_MATE_0_%d_%d_TAG_NM:i:3_" % (flag, Lk - 40, mpos, flen if fwd else -flen)
        qual = bytes(33 + rng.randint(20, 41, size=Lk).astype(np.uint8))
        recs.append(("%s%07d" % (prefix, p), c, ends[e].decode(), qual.decode()))
    return recs


def make_sparse_long_reads(anchors, n_pairs, seed, L=(1100, 1500), keep=(60, 160), period=(9, 16), stat=(1400, 1500, 2200, 3000)):
    """Long reads of which only one end seeds: the other L - keep bases carry a substitution every `period` bases (no 20-mer
    survives), so the candidate is one seed cluster plus ONE extension of ~1000+ bases with far more than 6 mismatches -- the DP
    problem of more than 10^6 cells that align_non_splice answers with a made-up <qlen>I<tlen>N CIGAR (read_realignment.cpp:874-887).
    A third of the reads keep both ends (two extensions short enough for the DP, a 10^6-cell end-to-end gap cannot arise: chained
    seeds are at most 50 / 400 read bases apart), a few are exact."""
    rng = np.random.RandomState(seed)
    recs = []
    for p in range(n_pairs):
        name, seq = anchors[rng.randint(len(anchors))]
        st_pos = int(name.split("_")[2])
        ends = []
        L0 = [int(rng.randint(L[0], L[1] + 1)) for _ in range(2)]
        flen = min(len(seq) - 2, max(L0) + rng.randint(50, 400))
        off = rng.randint(0, len(seq) - flen - 1)
        fragment = seq[off:off + flen]
        for e in range(2):
            Lr = min(L0[e], flen)
            src = fragment[:Lr] if e == 0 else revcomp(fragment[flen - Lr:flen])
            b = bytearray(src)
            mode = rng.randint(0, 7)              # 0-2 keep the head, 3-5 keep the tail, 6 exact
            k = int(rng.randint(keep[0], keep[1] + 1))
            if mode < 6:
                lo, hi = (k, Lr) if mode < 3 else (0, Lr - k)
                if mode % 3 == 2:                 # both ends seed: mutate only the middle
                    lo, hi = k, Lr - k
                q = lo + int(rng.randint(0, period[0]))
                while q < hi:
                    _sub(rng, b, q)
                    q += int(rng.randint(period[0], period[1] + 1))
            ends.append(bytes(b))
        recs += _pair_records(rng, p, ends, st_pos, off, flen, stat, "l")
    return recs


def make_clamp0_reads(anchors, n_pairs, seed, L=150, stat=(150, 200, 400, 600), small=False):
    """Reads over the START of the first anchor, i.e. of the whole concatenated reference: a left extension there is clamped at
    reference position 0 (read_realignment.cpp:323 `MAX(aln_ref_begin, 0)`), and when the read reaches further left than the
    reference (an overhang of unrelated bases, an insertion near the start) the window is SHORTER than the read piece: the reference
    then compares the piece's remaining bases with whatever its 1600-byte scratch buffer holds from earlier calls
    (KSW_ALN_handler::alignment, read_realignment.cpp:939).  Every read here starts at anchor offset 0..12 with 0..40 foreign
    bases in front and one or two edits within the first 30 anchor bases, so that the first seed starts a little inside; mates
    come from 150-400 bases further in.  Both orientations.  `small`: overhangs of 1-4 bases behind a single substitution -- the
    pieces whose outcome (scored as a plain match run, or handed to the DP) hangs on what the stale bytes happen to be."""
    rng = np.random.RandomState(seed)
    name, seq = anchors[0]
    st_pos = int(name.split("_")[2])
    recs = []
    for p in range(n_pairs):
        over = int(rng.randint(1, 5)) if small else int(rng.randint(0, 41))
        a0 = int(rng.randint(0, 3)) if small else int(rng.randint(0, 13))
        body = bytearray(seq[a0:a0 + L - over])
        for _ in range(1 if small else int(rng.randint(1, 3))):
            q = int(rng.randint(3, 40)) if small else int(rng.randint(1, 30))
            kind = 0 if small else rng.randint(0, 4)
            if kind < 2:
                _sub(rng, body, q)
            elif kind == 2:
                del body[q:q + int(rng.randint(1, 4))]
            else:
                body[q:q] = rand_dna(rng, int(rng.randint(1, 4)))
        r0 = (rand_dna(rng, over) + bytes(body))[:L]
        flen = int(rng.randint(300, 551))
        m = bytearray(seq[a0 + flen - L:a0 + flen])
        if rng.random_sample() < 0.5:
            _sub(rng, m, int(rng.randint(L)))
        ends = [r0, revcomp(bytes(m))]
        if rng.random_sample() < 0.3:             # both reads turned over: the overhanging read then aligns through its reverse strand
            ends = [revcomp(r0), bytes(m)]
        recs += _pair_records(rng, p, ends, st_pos, a0, flen, stat, "c")
    return recs


def write_fasta(path, anchors):
    with open(path, "w") as f:
        for name, seq in anchors:
            f.write(">%s\n" % name)
            s = seq.decode()
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")


def write_fastq(path, recs):
    with open(path, "w") as f:
        for name, c, s, q in recs:
            f.write("@%s %s\n%s\n+\n%s\n" % (name, c, s, q))
