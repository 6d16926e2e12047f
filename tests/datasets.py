"""Seeded data sets used by the golden fixtures and the parity tests (inputs are regenerated from
the seeds; only the reference's outputs and the compact index are committed)."""
import os

import synth

DATASETS = {
    # cfg-1 style smoke set: 20 INS anchors (edge 500), 2000 pairs of 150 bp
    "fx1": dict(anchors=dict(n_anchors=20, seed=7), reads={"reads150": dict(n_pairs=2000, seed=13),
                                                            # reads of the first three anchors, the FIRST included: the one the reference attributes to
                                                            # its neighbour (calloc'ed chr_file_n) -- reproduced, and pinned here
                                                            "anchor0": dict(n_pairs=400, seed=77, anchors=(0, 3))}),
    # tie-breaks / STR / N bases / 250 bp: duplicated flanks and tandem-repeat alleles
    "fx2": dict(anchors=dict(n_anchors=30, seed=21, edge=600, allele=(60, 400), str_frac=0.2, dup_frac=0.3),
                reads={"reads150": dict(n_pairs=1500, seed=23, str_frac=0.05, n_frac=0.05),
                       "reads250": dict(n_pairs=600, seed=29, L=250, frag=(500, 700), maxindel=40, str_frac=0.05, n_frac=0.03,
                                        stat=(250, 300, 600, 900))}),
    # edge cases: a 70 bp element shared by all 620 alleles (one unipath with 620 > POS_N_MAX positions: expand_seed's random_r
    # sampling), ragged read lengths incl. even ones (the reverse-strand middle-swap quirk), reads with 4-8 N bases
    "fx3": dict(anchors=dict(n_anchors=620, seed=31, edge=300, allele=(120, 200), repeat_len=70),
                reads={"ragged": dict(n_pairs=1200, seed=37, lengths=[100, 101, 126, 150, 151, 200, 250], frag=(520, 560), maxindel=8, n_frac=0.03,
                                      heavy_n_frac=0.03, stat=(150, 300, 500, 800)),
                       "repeat": dict(n_pairs=300, seed=41, L=150, frag=(300, 420), center_frac=0.9, miss_frac=0.05),
                       "lower": dict(n_pairs=300, seed=43, L=150, frag=(520, 560), lower_frac=0.5, stat=(150, 300, 500, 800))}),
    # configs[4] shape: edge-2000 anchors with alleles up to 2 kbp, 250 bp reads, indels up to 40 -- the wide DP problems
    # (extensions of q <= 250 against t = q + 30, 529 anti-diagonals) pinned against the reference objects
    "fx4": dict(anchors=dict(n_anchors=60, seed=51, edge=2000, allele=(60, 2000)),
                reads={"reads250": dict(n_pairs=2000, seed=53, L=250, frag=(400, 700), maxindel=40, n_frac=0.01, stat=(250, 300, 550, 800)),
                       # reads beyond 288 bases take the engine's wavefront-per-read preparation (the lane-per-pair one keeps a read in registers)
                       "long400": dict(n_pairs=300, seed=61, L=400, frag=(700, 900), maxindel=20, n_frac=0.03, str_frac=0.05, stat=(400, 600, 800, 1000))}),
    # the three reference branches no other set reaches, on one anchor set: 34 anchors, 27 of them with alleles that hold 300 copies each
    # of one 30 bp element (a unipath with 8100 > POS_N_MAX_LEVEL2 positions: expand_seed returns and drops the seeds behind it,
    # deBGA_index.cpp:224), long unique flanks for reads of 1100-1500 bases whose single extension exceeds 10^6 DP cells (the made-up
    # CIGAR of align_non_splice, read_realignment.cpp:874-887), and reads over the very start of the reference (left extension clamped
    # at position 0 with a window shorter than the read piece: the stale-scratch compare of read_realignment.cpp:939).
    # The index is not committed (0.5 Mbp): tests build it with `panSVR index` and check the files against the SHA-256 of the
    # reference builder's (tests/golden/fx5/idx.sha256).
    "fx5": dict(index="sha256", anchors=dict(n_anchors=34, seed=71, edge=900, allele=(800, 1600), repeat_len=30, repeat_copies=300, repeat_spacer=24, repeat_anchors=27),
                reads={"hicopy": dict(n_pairs=400, seed=73, L=150, frag=(300, 420), center_frac=0.8, miss_frac=0.05, anchors=(1, 27)),
                       "long": dict(kind="sparse_long", n_pairs=150, seed=79, anchors=(27, 34)),
                       "clamp0": dict(kind="clamp0", n_pairs=300, seed=83),
                       "clamp0s": dict(kind="clamp0", n_pairs=400, seed=89, small=True)}),
}


def anchors_of(name):
    return synth.make_anchors(**DATASETS[name]["anchors"])


def reads_of(name, rname):
    a = anchors_of(name)
    kw = dict(DATASETS[name]["reads"][rname])
    # by default anchor 0 is not sampled: the reference mis-assigns it (calloc'ed chr_file_n) and a left extension that
    # clamps at reference position 0 reads stale scratch bytes there; the sets with an `anchors` range include it on purpose
    lo, hi = kw.pop("anchors", (1, len(a)))
    kind = kw.pop("kind", None)
    if kind == "sparse_long":
        return synth.make_sparse_long_reads(a[lo:hi], **kw)
    if kind == "clamp0":
        return synth.make_clamp0_reads(a, **kw)
    return synth.make_reads(a[lo:hi], **kw)


def materialize(name, out_dir):
    os.makedirs(out_dir, exist_ok=True)
    synth.write_fasta(os.path.join(out_dir, "anchors.fa"), anchors_of(name))
    with open(os.path.join(out_dir, "header.sam"), "w") as f:
        f.write(synth.header_text())
    for rname in DATASETS[name]["reads"]:
        synth.write_fastq(os.path.join(out_dir, rname + ".fq"), reads_of(name, rname))
    return out_dir
