/*
 * oracle/ksw_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's banded anti-diagonal DP
 * (src/kswlib/ksw2_extd2_sse.c, ksw2_extz2_sse.c, ksw2.h).  Nothing in the
 * product path (pansvr_amd/, include/) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 */
#ifndef PSVR_KSW_ORACLE_H_
#define PSVR_KSW_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NEG_INF (-0x40000000)

#define ORC_EZ_SCORE_ONLY  0x01
#define ORC_EZ_RIGHT       0x02
#define ORC_EZ_GENERIC_SC  0x04
#define ORC_EZ_APPROX_MAX  0x08
#define ORC_EZ_APPROX_DROP 0x10
#define ORC_EZ_EXTZ_ONLY   0x40
#define ORC_EZ_REV_CIGAR   0x80

/* field-for-field image of ksw_extz_t (ksw2.h:26-35); cigar is caller storage */
typedef struct {
	int32_t max, zdropped;
	int32_t max_q, max_t;
	int32_t mqe, mqe_t;
	int32_t mte, mte_q;
	int32_t score;
	int32_t n_cigar;
	int32_t reach_end;
	int32_t cigar_overflow; /* 1 if n_cigar > cigar_cap (cigar truncated) */
} orc_extz_t;

/* ksw_extd2_sse (ksw2_extd2_sse.c:26-396).  cigar[] receives up to cigar_cap ops */
void orc_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
               orc_extz_t *ez, uint32_t *cigar, int cigar_cap);

/* ksw_extz2_sse (ksw2_extz2_sse.c:23-305), SSE2 code path */
void orc_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag,
               orc_extz_t *ez, uint32_t *cigar, int cigar_cap);

#ifdef __cplusplus
}
#endif
#endif
