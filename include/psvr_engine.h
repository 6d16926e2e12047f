/*
 * psvr_engine.h -- C ABI of the MI355X (gfx950) re-alignment engine.
 *
 * This is the drop-in boundary for panSVR's `aln` / `fc_aln` hot path.  Every entry point
 * names the reference interface it replaces (paths relative to the panSVR tree).
 * Plain C types only; no exceptions cross the boundary; every call returns an int status
 * (0 = PSVR_OK) and psvr_last_error() describes the last failure on the calling thread.
 *
 * Seam B2 (kernel):  psvr_extd2_batch*  replaces  ksw_extd2_sse   (src/kswlib/ksw2.h:63-64)
 *                    psvr_extz2_batch   replaces  ksw_extz2_sse   (src/kswlib/ksw2.h:57-58)
 * Seam B3 (seeding): psvr_seed_search_kmer_batch, psvr_seed_mem_batch
 *                                       replace   deBGA_INDEX::search_kmer / UNITIG_MEM_search
 *                                                 (src/deBGA_index.hpp:205-208; built copy
 *                                                  src/PanSVgenerateVCF/deBGA_index.hpp:198-201)
 * Seam B1 (batch):   psvr_engine_*      replaces  kt_for(worker_for -> align_read_pair)
 *                                                 (src/jlra_aln.cpp:115,140-147;
 *                                                  src/PanSVgenerateVCF/read_realignment.cpp:114,154-161,745-803)
 */
#ifndef PSVR_ENGINE_H_
#define PSVR_ENGINE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSVR_OK                0
#define PSVR_ERR_ARG           1   /* bad argument (null pointer, negative size, ...) */
#define PSVR_ERR_UNSUPPORTED   2   /* shape/flag outside what the device kernels implement */
#define PSVR_ERR_DEVICE        3   /* HIP runtime error (message in psvr_last_error) */
#define PSVR_ERR_NOMEM         4
#define PSVR_ERR_IO            5   /* index files missing / malformed */
#define PSVR_ERR_OVERFLOW      6   /* a caller-provided output arena was too small */

const char *psvr_last_error(void);
/* number of visible HIP devices; <0 on error.  Never falls back to a CPU path. */
int psvr_device_count(void);
/* Page-locked host memory for the buffers handed to psvr_engine_upload / psvr_engine_download / psvr_engine_align_batch
 * (optional: any host memory works, page-locked memory moves at the link's rate instead of a third of it).  The reference
 * side keeps its read / record buffers alive across batches (Classify_buff_pool, read_realignment.hpp:324-345): allocate
 * them once with these.  NULL on failure (psvr_last_error says why). */
void *psvr_host_alloc(size_t bytes);
void psvr_host_free(void *p);
/* Optional: sets up the device's queues on a thread of its own and returns at once.  The first streams of a process cost 3 - 14 ms each
 * and an engine uses four; a caller that has something else to do first (the command: loading the index, src/jlra_aln.cpp:29-57 in the
 * reference's order of work) calls this before it and finds them ready when it creates its engines.  Nothing depends on it. */
int psvr_device_warmup(int device, int n_streams);

/* ------------------------------------------------------------------------------------------
 * Seam B2: batched banded DP.  Field-for-field ksw_extz_t (src/kswlib/ksw2.h:26-35); the
 * CIGAR is written to a caller arena at [cigar_off, cigar_off + n_cigar) instead of a
 * callee-realloc'ed pointer (the reference's kalloc/malloc mix-up is not reproduced).
 * ------------------------------------------------------------------------------------------ */
#define PSVR_KSW_NEG_INF (-0x40000000)
#define PSVR_EZ_SCORE_ONLY  0x01
#define PSVR_EZ_RIGHT       0x02
#define PSVR_EZ_GENERIC_SC  0x04
#define PSVR_EZ_APPROX_MAX  0x08
#define PSVR_EZ_APPROX_DROP 0x10
#define PSVR_EZ_EXTZ_ONLY   0x40
#define PSVR_EZ_REV_CIGAR   0x80

typedef struct psvr_extz {
	int32_t max, zdropped;
	int32_t max_q, max_t;
	int32_t mqe, mqe_t;
	int32_t mte, mte_q;
	int32_t score;
	int32_t n_cigar;
	int32_t reach_end;
	int32_t reserved;
	int64_t cigar_off;      /* first op of this alignment in the cigar arena (uint32 units) */
} psvr_extz_t;

/* the scalar arguments of ksw_extd2_sse, in its order (m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag) */
typedef struct psvr_ksw_params {
	int8_t  m;              /* alphabet size (5) */
	int8_t  mat[25];        /* m*m scoring matrix (KSW_ALN_handler::ksw_gen_mat_D, read_realignment.cpp:829-843) */
	int8_t  q, e, q2, e2;   /* gap open/extend pairs; extz2 uses q,e only */
	int32_t w;              /* band width (<0 disables) */
	int32_t zdrop;
	int32_t end_bonus;
	int32_t flag;           /* PSVR_EZ_* */
} psvr_ksw_params_t;

/* Upper bound of CIGAR ops alignment (qlen,tlen) can emit; the arena needs the sum over the batch. */
static inline int64_t psvr_cigar_bound(int32_t qlen, int32_t tlen) { return (int64_t)qlen + tlen + 2; }

/*
 * Host-buffer form: n independent problems; problem i has query  qseq[q_off[i] .. +qlen[i])  and
 * target tseq[t_off[i] .. +tlen[i]) with codes 0..m-1.  Results in ez[i]; CIGAR ops (BAM encoding
 * len<<4|op) in cigar_arena (capacity cigar_cap uint32).  Runs on HIP device `device`.
 */
int psvr_extd2_batch(int device, int64_t n,
                     const uint8_t *qseq, const int64_t *q_off, const int32_t *qlen,
                     const uint8_t *tseq, const int64_t *t_off, const int32_t *tlen,
                     const psvr_ksw_params_t *par,
                     psvr_extz_t *ez, uint32_t *cigar_arena, int64_t cigar_cap);

int psvr_extz2_batch(int device, int64_t n,
                     const uint8_t *qseq, const int64_t *q_off, const int32_t *qlen,
                     const uint8_t *tseq, const int64_t *t_off, const int32_t *tlen,
                     const psvr_ksw_params_t *par,
                     psvr_extz_t *ez, uint32_t *cigar_arena, int64_t cigar_cap);

/*
 * Device-pointer form (all pointers are HIP device memory, `stream` is a hipStream_t or NULL):
 * nothing is copied and nothing synchronises; `work` is a device workspace of at least
 * psvr_dp_plan_workspace_bytes(plan) bytes.  ez[i].cigar_off must be pre-set by the caller
 * (e.g. an exclusive scan of psvr_cigar_bound).  tools/dp_bench.py times this form; bench.py times
 * psvr_engine_run (the whole path).
 */
typedef struct psvr_dp_plan psvr_dp_plan_t;   /* opaque: size classes + index lists for one batch shape */
int psvr_dp_plan_create(int device, int64_t n, const int32_t *qlen_host, const int32_t *tlen_host,
                        const psvr_ksw_params_t *par, int variant /*0 extd2, 1 extz2*/, psvr_dp_plan_t **plan);
int64_t psvr_dp_plan_workspace_bytes(const psvr_dp_plan_t *plan);
int psvr_dp_plan_launch(psvr_dp_plan_t *plan,
                        const uint8_t *d_qseq, const int64_t *d_q_off,
                        const uint8_t *d_tseq, const int64_t *d_t_off,
                        psvr_extz_t *d_ez, uint32_t *d_cigar_arena, void *d_work, void *stream);
/* fills n_kernels/name/launch geometry of the plan for profiling reports */
int psvr_dp_plan_describe(const psvr_dp_plan_t *plan, char *buf, size_t buflen);
void psvr_dp_plan_destroy(psvr_dp_plan_t *plan);


/* ------------------------------------------------------------------------------------------
 * The deBGA unipath k-mer index, resident in HBM.
 * Replaces deBGA_INDEX::load_index_file + building_chr_index + building_bam_header
 * (src/deBGA_index.cpp:40-86,363-439 / src/PanSVgenerateVCF/deBGA_index.cpp:33-80,354-431).
 * The nine files are uploaded once; the 2 GiB first-level table `unipath_g.hash` stays a dense
 * uint64[4^14+1] prefix-sum array so that one 16-byte gather returns hash[h], hash[h+1].
 * ------------------------------------------------------------------------------------------ */
typedef struct psvr_index psvr_index_t;

typedef struct psvr_index_view {           /* host pointers, element counts (not bytes) */
	const uint64_t *ref_seq;  uint64_t n_ref_seq;   /* ref.seq          2-bit MSB-first, 32 bases/word */
	const uint64_t *seq;      uint64_t n_seq;       /* unipath.seqb     same packing */
	const uint64_t *seqf;     uint64_t n_seqf;      /* unipath.seqfb    unipath start offsets (U+1) */
	const uint64_t *pos;      uint64_t n_pos;       /* unipath.pos      1-based reference positions */
	const uint64_t *posp;     uint64_t n_posp;      /* unipath.posp     (U+1) */
	const uint64_t *hash;     uint64_t n_hash;      /* unipath_g.hash   4^14+1 prefix sums */
	const uint32_t *kmer;     uint64_t n_kmer;      /* unipath_g.kmer   low 16 bits of each 22-mer */
	const uint64_t *off;      uint64_t n_off;       /* unipath_g.offset */
	const char *chr_text;                           /* contents of unipath.chr (name, cumulative end+1 alternating) */
	const char *const *header_names; int32_t n_header; /* @SQ names of the ORIGINAL genome header (bam_name2id lookups) */
} psvr_index_view_t;

int psvr_index_create(const psvr_index_view_t *view, int device, psvr_index_t **out);
/* The same with the eight arrays of `view` already in the memory of `device` (device pointers; chr_text / header_names stay host
 * pointers): one process per GPU receives them through a collective -- rank 0 uploads, an RCCL broadcast over xGMI brings them to
 * the other ranks (SURVEY 8(e) "broadcast index"; bench.py --gpus N) -- and builds its index from them, device to device. */
int psvr_index_create_from_device(const psvr_index_view_t *view, int device, psvr_index_t **out);
/* reads the nine files from `index_dir` and the @SQ lines of `header_sam` */
int psvr_index_load(const char *index_dir, const char *header_sam, int device, psvr_index_t **out);
/* Multi-GPU: a second copy of an index that is already resident on another device, moved device to device (xGMI peer
 * copies instead of N host uploads; SURVEY 8(e) "broadcast index").  `src` stays valid. */
int psvr_index_clone(const psvr_index_t *src, int device, psvr_index_t **out);
/* The index straight from the anchor FASTA (what `deBGA index -k 22` + load_index_file do through nine files: the builder of
 * `panSVR index` runs on the host, the 2 GiB first-level table is expanded in HBM and never exists on the host or on disk) */
int psvr_index_build(const char *anchors_fa, const char *header_sam, int device, psvr_index_t **out);
void psvr_index_destroy(psvr_index_t *idx);
int64_t psvr_index_device_bytes(const psvr_index_t *idx);
int32_t psvr_index_n_anchor(const psvr_index_t *idx);
/* SV_chr_info::vcf_print_string / vcf_id of anchor `sv_id` (deBGA_index.hpp:116-119); NULL if out of range */
const char *psvr_index_sv_print_string(const psvr_index_t *idx, int32_t sv_id);
const char *psvr_index_sv_vcf_id(const psvr_index_t *idx, int32_t sv_id);

/* ------------------------------------------------------------------------------------------
 * Seam B3: the two index look-ups of the seed loop, batched (src/PanSVgenerateVCF/deBGA_index.hpp:198-201; north-star copy
 * src/deBGA_index.hpp:205-208).  The engine runs the same device functions inside its seeding kernel; these entry points
 * expose them on their own (host buffers in, host buffers out) so the seeding stage can be bound or tested separately.
 * ------------------------------------------------------------------------------------------ */
/* bool deBGA_INDEX::search_kmer(20, kmer, range, 2) for n 20-mers (40 significant bits each): found[i], and when found the
 * inclusive index range range[2i] .. range[2i+1] of the 22-mers of unipath_g.kmer / unipath_g.offset that start with it */
int psvr_seed_search_kmer_batch(const psvr_index_t *idx, int64_t n, const uint64_t *kmers, int64_t *range, uint8_t *found);
typedef struct psvr_vertex_mem {            /* vertex_MEM, deBGA_index.hpp:24-58 (+ the right_i the reference returns through max_right_i) */
	uint64_t uid;
	uint32_t seed_id;                       /* position in the caller's vector in the reference: always 0 here */
	uint32_t read_pos, uni_pos_off, length, pos_n;
	uint32_t right_i;
} psvr_vertex_mem_t;
/* int deBGA_INDEX::UNITIG_MEM_search(kmer_index, ..., read_bit, read_off, read_length, 20, max_right_i) for n items: item i extends
 * index entry kmer_index[i] inside its unipath against the 2-bit packed read (32 bases per word, MSB first) that starts at
 * read_bits[word_off[i]]; every read needs ceil(len / 32) + 1 words (the reference's read_bit arrays are padded the same way) */
int psvr_seed_mem_batch(const psvr_index_t *idx, int64_t n, const uint64_t *kmer_index, const uint64_t *read_bits, int64_t n_words,
                        const int64_t *word_off, const uint32_t *read_off, const uint32_t *read_len, psvr_vertex_mem_t *out);

/* ------------------------------------------------------------------------------------------
 * Seam B1: one batch of read pairs through seeding -> chaining -> extension DP -> pairing.
 * Replaces kt_for(worker_for -> align_read_pair) minus the SAM text formatting
 * (src/PanSVgenerateVCF/read_realignment.cpp:114,154-161,745-775; legacy src/jlra_aln.cpp:115,140-147).
 * Results are those of the reference at `-t 1`: the engine consumes the same rand()/random_r
 * draw sequence in input order (see DESIGN.md "rand() order").
 * ------------------------------------------------------------------------------------------ */
typedef struct psvr_engine psvr_engine_t;

typedef struct psvr_aln_params {           /* MAP_PARA, read_realignment.hpp:46-129 */
	int32_t match, mismatch, gap_open, gap_ex, gap_open2, gap_ex2, zdrop;
	int32_t normal_read_length, isize_min, isize_max;   /* STAT_ of the first read or 150/100/900 */
	int32_t min_filter_score;
} psvr_aln_params_t;
void psvr_aln_params_default(psvr_aln_params_t *p);

/* the original alignment parsed from the FASTQ comment (single_end_handler::parse_ori_mapping_rst,
 * read_realignment.hpp:392-429): tokens 0-4 and the signal-flag token */
typedef struct psvr_ori {
	int32_t  chr_id;
	uint32_t ref_bg, read_bg, align_score;
	uint8_t  mapq, direction /* 1 = FORWARD */, unmapped, reserved;
} psvr_ori_t;

#define PSVR_MAX_RESULT 12                  /* MAX_OUTPUT_NUMBER * 2, read_realignment.hpp:323,328 */
typedef struct psvr_cand {                  /* MAX_IDX_OUTPUT, read_realignment.hpp:243-319 */
	uint32_t align_score, chain_score, ref_bg, read_bg;
	int32_t  chr_id, sv_id;
	uint32_t max_index;
	uint32_t n_cigar;
	int64_t  cigar_off;                     /* into the engine's cigar arena (uint32 len<<4|op) */
	uint8_t  direction, mapq, reserved[6];
} psvr_cand_t;

typedef struct psvr_read_result {
	int32_t n_result;
	uint8_t unmapped, early_out, is_str, reserved;
	int32_t primary, secondary;             /* -1 none, -2 the original alignment, k>=0 = cand[k] */
	int32_t has_mate, mate_chr_id;
	uint32_t mate_ref_bg;
	int32_t prim_sv_id, mate_sv_id;         /* SV:Z / MV:Z anchors of the primary record */
	uint32_t n_seed[2];                     /* trace: UNI_SEEDs per strand */
	uint32_t reserved1;                     /* 0 (the alignment hole in front of the 64-bit fields, named so that every byte of a record is written) */
	uint64_t seed_hash[2], chain_hash[2];   /* trace: FNV-1a of the sorted seeds / chaining DP per strand */
	psvr_cand_t cand[PSVR_MAX_RESULT];
} psvr_read_result_t;

/* The same results in the form the engine keeps them in HBM: one 48-byte header per read and a dense list of only the
 * candidates that exist (the fixed 12-slot psvr_read_result_t is materialised from these on request).  A pipeline that
 * formats SAM records needs nothing else; a 1 M-pair batch comes back as ~0.2 GB instead of 1.35 GB. */
typedef struct psvr_read_hdr {
	int32_t n_result;
	uint8_t unmapped, early_out, is_str, reserved;
	int32_t primary, secondary;             /* -1 none, -2 the original alignment, k>=0 = cands[cand_off + k] */
	int32_t has_mate, mate_chr_id;
	uint32_t mate_ref_bg;
	int32_t prim_sv_id, mate_sv_id;
	int32_t reserved2;
	int64_t cand_off;                       /* first of this read's n_result candidates in the candidate list */
} psvr_read_hdr_t;

typedef struct psvr_pair_result {           /* PE_score, read_realignment.hpp:434-628 */
	int32_t max_score, cur_isize;
	int32_t proper, gain;
	int32_t max1, max2;                     /* -1 none, -2 original, k */
} psvr_pair_result_t;

int psvr_engine_create(const psvr_index_t *idx, const psvr_aln_params_t *par, psvr_engine_t **out);
void psvr_engine_destroy(psvr_engine_t *eng);
/*
 * n_pairs read pairs; read r = 2*pair + mate has bases[base_off[r] .. base_off[r+1]) (ASCII) and ori[r].
 * reads[2*n_pairs] / pairs[n_pairs] receive the results; CIGARs of all candidates are appended to
 * cigar[] (capacity cigar_cap uint32; PSVR_ERR_OVERFLOW if too small).  `trace` != 0 also fills the
 * per-strand trace hashes.  Host-buffer form: copies in, runs, copies out.
 */
int psvr_engine_align_batch(psvr_engine_t *eng, int64_t n_pairs, const char *bases, const int64_t *base_off,
                            const psvr_ori_t *ori, psvr_read_result_t *reads, psvr_pair_result_t *pairs,
                            uint32_t *cigar, int64_t cigar_cap, int trace);
/* Device-resident form used by bench.py: upload once, run the hot path any number of times with the
 * rand() state rewound, download once.  No host<->device traffic inside psvr_engine_run. */
int psvr_engine_upload(psvr_engine_t *eng, int64_t n_pairs, const char *bases, const int64_t *base_off, const psvr_ori_t *ori);
int psvr_engine_run(psvr_engine_t *eng, int trace, void *stream);
int psvr_engine_download(psvr_engine_t *eng, psvr_read_result_t *reads, psvr_pair_result_t *pairs,
                         uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_used);
/* Compact form of the same results: hdr[2*n_pairs], pairs[n_pairs], then only the candidates and CIGAR words that exist,
 * densely packed in read order (cands[k].cigar_off indexes `cigar`).  Call with cands == NULL (or cigar == NULL) to learn
 * cand_used / cigar_used first; PSVR_ERR_OVERFLOW if a capacity is too small (the counts are still returned).
 * The four destinations may be host memory (page-locked: psvr_host_alloc) or memory of the engine's device: the ordered
 * gather of a one-process-per-GPU host (reference: output_results, read_realignment.cpp:165-176, which walks the batch in
 * input order) sends a block on to its peer straight from HBM. */
int psvr_engine_download_compact(psvr_engine_t *eng, psvr_read_hdr_t *hdr, psvr_pair_result_t *pairs,
                                 psvr_cand_t *cands, int64_t cand_cap, int64_t *cand_used,
                                 uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_used);
/*
 * Multi-GPU: read pairs are independent given the index, except that the reference consumes ONE rand()/random_r draw
 * sequence in input order.  When consecutive shards of a batch run on different GPUs, shard r starts at the stream
 * position where shard r-1 ended.  pos/end = {rand() draws, handler-0 random_r draws, handler-1 random_r draws}.
 *   psvr_engine_set_stream_pos : where the NEXT run starts (default: where the previous batch ended)
 *   psvr_engine_stream_end     : where the last run ended
 *   psvr_engine_rebase         : the last run should have started at `pos`: move it there, re-running only the pairs whose
 *                                draws moved (results afterwards == a run started at `pos`)
 * The exchange of the three integers between ranks is the caller's (one all-gather; see pansvr_amd/dist.py).
 */
int psvr_engine_set_stream_pos(psvr_engine_t *eng, const int64_t pos[3]);
int psvr_engine_stream_end(psvr_engine_t *eng, int64_t end[3]);
int psvr_engine_rebase(psvr_engine_t *eng, const int64_t pos[3], void *stream);
/* work counters of the last run (probes, hits, dp problems, cells, speculative re-runs ...) as JSON */
int psvr_engine_stats(const psvr_engine_t *eng, char *buf, size_t buflen);

/* ---- BGZF members on the device (the BAM output's compression) --------------------------------------------------------------
 * Replaces, for the drop-in command's BAM output, htslib's bgzf_compress (htslib bgzf.c: zlib deflate of 0xff00-byte blocks on the host,
 * reached from the reference's sam_write1 calls, read_realignment.cpp:166-176 -> bam_file.c).  `in` (host memory, n_bytes) is cut into
 * blocks of 16 KB (BGZF allows any size up to 64 KB; htslib uses 0xff00); every block becomes one BGZF member (gzip header with the BC field, raw DEFLATE, CRC32, ISIZE), the members are
 * written next to each other into `out` (host memory, out_cap bytes; psvr_bgzf_bound(n_bytes) always suffices) and *out_bytes is their
 * total size.  The EOF marker block is the caller's.  Any BGZF / gzip reader decodes the result; the bytes differ from zlib's. */
int64_t psvr_bgzf_bound(int64_t n_bytes);
int psvr_bgzf_compress(int device, const void *in, int64_t n_bytes, void *out, int64_t out_cap, int64_t *out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PSVR_ENGINE_H_ */
