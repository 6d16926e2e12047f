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
 * Seam B3 (seeding): psvr_seed_*        replaces  deBGA_INDEX::search_kmer / UNITIG_MEM_search
 *                                                 (src/deBGA_index.hpp:198-201)
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
 * psvr_extd2_workspace_bytes(...) bytes.  ez[i].cigar_off must be pre-set by the caller
 * (e.g. an exclusive scan of psvr_cigar_bound).  This is what bench.py times.
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

#ifdef __cplusplus
}
#endif
#endif /* PSVR_ENGINE_H_ */
