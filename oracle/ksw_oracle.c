/*
 * oracle/ksw_oracle.c -- TEST INFRASTRUCTURE ONLY (see ksw_oracle.h).
 *
 * Scalar C restatement of the reference's SSE2 anti-diagonal DP.  It keeps the
 * reference's exact observable arithmetic:
 *   - 8-bit wrap-around adds/subs and signed 8-bit compares
 *     (ksw2_extd2_sse.c:30-58, 238-265),
 *   - the st/en rounding to 16-lane blocks, so "garbage" lanes outside the
 *     band are computed and later read back (ksw2_extd2_sse.c:140-156),
 *   - the flat u|v|x|y|x2|y2|s|sf|qr memory image, so unaligned 16-byte score
 *     loads/stores that run past an array land where the reference's do
 *     (ksw2_extd2_sse.c:100-103, 159-173),
 *   - the 4-lane max/arg-max tie-break of the exact-max loop (:316-351),
 *   - ksw_apply_zdrop / ksw_backtrack_D / ksw_push_cigar (ksw2.h:106-151, 245-261).
 * Parity of this file against the compiled reference objects (oracle/_ref) and
 * against tests/golden/ksw_kat.json is checked by tests/test_oracle_ksw.py.
 */
#include <stdlib.h>
#include <string.h>
#include "ksw_oracle.h"

static inline int8_t add8(int8_t a, int8_t b) { return (int8_t)(uint8_t)((uint8_t)a + (uint8_t)b); }
static inline int8_t sub8(int8_t a, int8_t b) { return (int8_t)(uint8_t)((uint8_t)a - (uint8_t)b); }
static inline uint8_t maxu8(uint8_t a, uint8_t b) { return a > b ? a : b; }
static inline uint8_t minu8(uint8_t a, uint8_t b) { return a < b ? a : b; }

/* ksw_reset_extz, ksw2.h:238-243 */
static void orc_reset(orc_extz_t *ez)
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = ORC_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0, ez->cigar_overflow = 0;
}

/* ksw_apply_zdrop with is_rot=1, ksw2.h:245-261 */
static int orc_apply_zdrop(orc_extz_t *ez, int32_t H, int r, int t, int zdrop, int8_t e)
{
	if (H > ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez->max - H > zdrop + l * e) {
			ez->zdropped = 1;
			return 1;
		}
	}
	return 0;
}

typedef struct { uint32_t *c; int n, cap, overflow; } cig_t;

/* ksw_push_cigar, ksw2.h:106-116 (caller-owned storage instead of krealloc) */
static void push_cigar(cig_t *cg, uint32_t op, int len)
{
	if (cg->overflow) return;
	if (cg->n == 0 || op != (cg->c[cg->n - 1] & 0xf)) {
		if (cg->n == cg->cap) { cg->overflow = 1; return; }
		cg->c[cg->n++] = (uint32_t)len << 4 | op;
	} else cg->c[cg->n - 1] += (uint32_t)len << 4;
}

/* ksw_backtrack_D with is_rot=1, min_intron_len=0, ksw2.h:119-151 */
static void orc_backtrack(int is_rev, const uint8_t *p, const int *off, const int *off_end, int n_col, int i0, int j0, cig_t *cg)
{
	int i = i0, j = j0, r, state = 0;
	uint32_t tmp;
	cg->n = 0;
	while (i >= 0 && j >= 0) {
		int force_state = -1;
		r = i + j;
		if (i < off[r]) force_state = 2;
		if (off_end && i > off_end[r]) force_state = 1;
		tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) push_cigar(cg, 0, 1), --i, --j;
		else if (state == 1 || state == 3) push_cigar(cg, 2, 1), --i;
		else push_cigar(cg, 1, 1), --j;
	}
	if (i >= 0) push_cigar(cg, 2, i + 1);
	if (j >= 0) push_cigar(cg, 1, j + 1);
	if (!is_rev && !cg->overflow)
		for (i = 0; i < cg->n >> 1; ++i)
			tmp = cg->c[i], cg->c[i] = cg->c[cg->n - 1 - i], cg->c[cg->n - 1 - i] = tmp;
}

void orc_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
               orc_extz_t *ez, uint32_t *cigar, int cigar_cap)
{
	int r, t, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, wl, wr, max_sc, min_sc, long_thres, long_diff;
	int with_cigar = !(flag & ORC_EZ_SCORE_ONLY), approx_max = !!(flag & ORC_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int8_t sc_mch, sc_mis, sc_N, m1, qe8, qe28;
	uint8_t *mem, *p = 0;
	int8_t *u8, *v8, *x8, *y8, *x28, *y28, *s8;
	uint8_t *sf, *qr;
	size_t T;
	cig_t cg;

	orc_reset(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;

	if (q2 + e2 < q + e) t = q, q = q2, q2 = t, t = e, e = e2, e2 = t; /* :70 */
	/* NB: `qe` keeps its PRE-swap value (declared at :60, never refreshed); it only feeds H[0] at r==0 (:351,372) */
	qe8 = (int8_t)(q + e), qe28 = (int8_t)(q2 + e2);
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (int8_t)(-e2) : mat[m * m - 1];
	m1 = m - 1;

	if (w < 0) w = tlen > qlen ? tlen : qlen;
	wl = wr = w;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;

	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e)
		++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	T = (size_t)tlen_ * 16;
	/* flat image: u v x y x2 y2 s sf qr  (:100-103); +64 zero bytes of slack */
	mem = (uint8_t*)calloc(((size_t)tlen_ * 8 + qlen_ + 1) * 16 + 64, 1);
	u8 = (int8_t*)mem, v8 = u8 + T, x8 = v8 + T, y8 = x8 + T, x28 = y8 + T, y28 = x28 + T, s8 = y28 + T;
	sf = (uint8_t*)(s8 + T), qr = sf + T;
	memset(u8,  -q  - e,  T);
	memset(v8,  -q  - e,  T);
	memset(x8,  -q  - e,  T);
	memset(y8,  -q  - e,  T);
	memset(x28, -q2 - e2, T);
	memset(y28, -q2 - e2, T);
	if (!approx_max) {
		H = (int32_t*)malloc(T * 4);
		for (t = 0; t < (int)T; ++t) H[t] = ORC_NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int*)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}

	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		uint8_t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) {
			ez->zdropped = 1;
			break;
		}
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) {
				x1 = x8[st - 1], x21 = x28[st - 1], v1 = v8[st - 1];
			} else {
				x1 = -q - e, x21 = -q2 - e2;
				v1 = -q - e;
			}
		} else {
			x1 = -q - e, x21 = -q2 - e2;
			v1 = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		}
		if (en >= r) {
			y8[r] = -q - e, y28[r] = -q2 - e2;
			u8[r] = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		}
		/* scores (:158-177): unaligned 16-byte groups starting at st0 */
		if (!(flag & ORC_EZ_GENERIC_SC)) {
			for (t = st0; t <= en0; t += 16) {
				int k;
				int8_t tmp16[16];
				for (k = 0; k < 16; ++k) {
					uint8_t sq = sf[t + k], sq2 = qrr[t + k];
					int mask = (sq == (uint8_t)m1) || (sq2 == (uint8_t)m1);
					int8_t sc = sq == sq2 ? sc_mch : sc_mis;
					tmp16[k] = mask ? sc_N : sc;
				}
				memcpy(s8 + t, tmp16, 16); /* loads precede the store, as in the SSE code */
			}
		} else {
			for (t = st0; t <= en0; ++t)
				((uint8_t*)s8)[t] = mat[sf[t] * m + qrr[t]];
		}
		/* core loop over the 16-rounded lane range [st,en] (:178-315) */
		{
			int8_t cx = x1, cx2 = x21, cv = v1;
			uint8_t *pr = 0;
			if (with_cigar) {
				pr = p + ((size_t)r * n_col_) * 16 - st;
				off[r] = st, off_end[r] = en;
			}
			for (t = st; t <= en; ++t) {
				int8_t z = s8[t], xt1 = cx, vt1 = cv, x2t1 = cx2, ut, a, b, a2, b2, tmp;
				uint8_t d = 0;
				cx = x8[t], cv = v8[t], cx2 = x28[t];
				a = add8(xt1, vt1);
				ut = u8[t];
				b = add8(y8[t], ut);
				a2 = add8(x2t1, vt1);
				b2 = add8(y28[t], ut);
				if (!with_cigar || !(flag & ORC_EZ_RIGHT)) { /* left-alignment / score-only */
					if (a > z)  d = 1, z = a;
					if (b > z)  d = 2, z = b;
					if (a2 > z) d = 3, z = a2;
					if (b2 > z) d = 4, z = b2;
					if (sc_mch < z) z = sc_mch;
				} else { /* right-alignment (:268-299) */
					d = z > a ? 0 : 1;   z = z > a ? z : a;
					d = z > b ? d : 2;   z = z > b ? z : b;
					d = z > a2 ? d : 3;  z = z > a2 ? z : a2;
					d = z > b2 ? d : 4;  z = z > b2 ? z : b2;
					if (sc_mch < z) z = sc_mch;
				}
				u8[t] = sub8(z, vt1);
				v8[t] = sub8(z, ut);
				tmp = sub8(z, q);
				a = sub8(a, tmp);
				b = sub8(b, tmp);
				tmp = sub8(z, q2);
				a2 = sub8(a2, tmp);
				b2 = sub8(b2, tmp);
				if (!with_cigar || !(flag & ORC_EZ_RIGHT)) {
					x8[t]  = sub8(a  > 0 ? a  : 0, qe8);  if (a  > 0) d |= 0x08;
					y8[t]  = sub8(b  > 0 ? b  : 0, qe8);  if (b  > 0) d |= 0x10;
					x28[t] = sub8(a2 > 0 ? a2 : 0, qe28); if (a2 > 0) d |= 0x20;
					y28[t] = sub8(b2 > 0 ? b2 : 0, qe28); if (b2 > 0) d |= 0x40;
				} else {
					x8[t]  = sub8(0 > a  ? 0 : a,  qe8);  if (!(0 > a))  d |= 0x08;
					y8[t]  = sub8(0 > b  ? 0 : b,  qe8);  if (!(0 > b))  d |= 0x10;
					x28[t] = sub8(0 > a2 ? 0 : a2, qe28); if (!(0 > a2)) d |= 0x20;
					y28[t] = sub8(0 > b2 ? 0 : b2, qe28); if (!(0 > b2)) d |= 0x40;
				}
				if (with_cigar) pr[t] = d;
			}
		}
		if (!approx_max) { /* :316-359 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u8[en0] : H[en0] + v8[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4) {
					for (i = 0; i < 4; ++i) {
						H[t + i] += v8[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v8[t];
					if (H[t] > max_H)
						max_H = H[t], max_t = t;
				}
			} else H[0] = v8[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte)
				ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe)
				ez->mqe = H[st0], ez->mqe_t = st0;
			if (orc_apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1)
				ez->score = H[tlen - 1];
		} else { /* :360-376 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v8[last_H0_t];
					int32_t d1 = u8[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v8[last_H0_t];
				} else {
					++last_H0_t, H0 += u8[last_H0_t];
				}
			} else H0 = v8[0] - qe, last_H0_t = 0;
			if ((flag & ORC_EZ_APPROX_DROP) && orc_apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1)
				ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) { /* :382-395 */
		int rev_cigar = !!(flag & ORC_EZ_REV_CIGAR);
		cg.c = cigar, cg.n = 0, cg.cap = cigar_cap, cg.overflow = 0;
		if (!ez->zdropped && !(flag & ORC_EZ_EXTZ_ONLY)) {
			orc_backtrack(rev_cigar, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &cg);
		} else if (!ez->zdropped && (flag & ORC_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			orc_backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->mqe_t, qlen - 1, &cg);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) {
			orc_backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &cg);
		}
		ez->n_cigar = cg.n, ez->cigar_overflow = cg.overflow;
		free(p);
		free(off);
	}
}

void orc_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag,
               orc_extz_t *ez, uint32_t *cigar, int cigar_cap)
{
	int r, t, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, wl, wr, max_sc, min_sc;
	int with_cigar = !(flag & ORC_EZ_SCORE_ONLY), approx_max = !!(flag & ORC_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	uint8_t *mem, *p = 0, *u8, *v8, *x8, *y8, *s8, *sf, *qr;
	uint8_t qe2b, sc_mch, sc_mis, sc_N, m1, max_scb, qb;
	size_t T;
	cig_t cg;

	orc_reset(ez);
	if (m <= 0 || qlen <= 0 || tlen <= 0) return;

	qb = (uint8_t)q;
	qe2b = (uint8_t)((q + e) * 2);
	sc_mch = (uint8_t)mat[0], sc_mis = (uint8_t)mat[1];
	sc_N = mat[m * m - 1] == 0 ? (uint8_t)(-e) : (uint8_t)mat[m * m - 1];
	m1 = (uint8_t)(m - 1);
	max_scb = (uint8_t)(mat[0] + (q + e) * 2);

	if (w < 0) w = tlen > qlen ? tlen : qlen;
	wl = wr = w;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;

	T = (size_t)tlen_ * 16;
	mem = (uint8_t*)calloc(((size_t)tlen_ * 6 + qlen_ + 1) * 16 + 64, 1); /* u v x y s sf qr (:85-87) */
	u8 = mem, v8 = u8 + T, x8 = v8 + T, y8 = x8 + T, s8 = y8 + T, sf = s8 + T, qr = sf + T;
	if (!approx_max) {
		H = (int32_t*)malloc(T * 4);
		for (t = 0; t < (int)T; ++t) H[t] = ORC_NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int*)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, v1;
		uint8_t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) {
			ez->zdropped = 1;
			break;
		}
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en)
				x1 = (int8_t)x8[st - 1], v1 = (int8_t)v8[st - 1];
			else x1 = v1 = 0;
		} else x1 = 0, v1 = r ? q : 0;
		if (en >= r) y8[r] = 0, u8[r] = r ? qb : 0;
		if (!(flag & ORC_EZ_GENERIC_SC)) {
			for (t = st0; t <= en0; t += 16) {
				int k;
				uint8_t tmp16[16];
				for (k = 0; k < 16; ++k) {
					uint8_t sq = sf[t + k], sq2 = qrr[t + k];
					int mask = (sq == m1) || (sq2 == m1);
					uint8_t sc = sq == sq2 ? sc_mch : sc_mis;
					tmp16[k] = mask ? sc_N : sc;
				}
				memcpy(s8 + t, tmp16, 16);
			}
		} else {
			for (t = st0; t <= en0; ++t)
				s8[t] = (uint8_t)mat[sf[t] * m + qrr[t]];
		}
		{
			/* _mm_cvtsi32_si128(int8_t) sign-extends into lanes 1..3 of the first block (:147-148) */
			uint32_t x1w = (uint32_t)(int32_t)x1, v1w = (uint32_t)(int32_t)v1;
			uint8_t cx = 0, cv = 0;
			uint8_t *pr = 0;
			if (with_cigar) {
				pr = p + ((size_t)r * n_col_) * 16 - st;
				off[r] = st, off_end[r] = en;
			}
			for (t = st; t <= en; ++t) {
				uint8_t z, xt1, vt1, ut, a, b, d = 0;
				int k = t - st;
				xt1 = k == 0 ? 0 : cx;
				vt1 = k == 0 ? 0 : cv;
				if (k < 4) xt1 |= (uint8_t)(x1w >> (8 * k)), vt1 |= (uint8_t)(v1w >> (8 * k));
				cx = x8[t], cv = v8[t];
				z = (uint8_t)(s8[t] + qe2b);
				a = (uint8_t)(xt1 + vt1);
				ut = u8[t];
				b = (uint8_t)(y8[t] + ut);
				if (!with_cigar) {
					z = (int8_t)z > 0 ? z : 0;
					z = maxu8(z, a);
				} else if (!(flag & ORC_EZ_RIGHT)) {
					d = (int8_t)a > (int8_t)z ? 1 : 0;
					z = (int8_t)z > 0 ? z : 0;
					z = maxu8(z, a);
					if ((int8_t)b > (int8_t)z) d = 2;
				} else {
					d = (int8_t)z > (int8_t)a ? 0 : 1;
					z = (int8_t)z > 0 ? z : 0;
					z = maxu8(z, a);
					if (!((int8_t)z > (int8_t)b)) d = 2;
				}
				z = maxu8(z, b);
				z = minu8(z, max_scb);
				u8[t] = (uint8_t)(z - vt1);
				v8[t] = (uint8_t)(z - ut);
				z = (uint8_t)(z - qb);
				a = (uint8_t)(a - z);
				b = (uint8_t)(b - z);
				if (!with_cigar || !(flag & ORC_EZ_RIGHT)) {
					x8[t] = (int8_t)a > 0 ? a : 0; if ((int8_t)a > 0) d |= 0x08;
					y8[t] = (int8_t)b > 0 ? b : 0; if ((int8_t)b > 0) d |= 0x10;
				} else {
					x8[t] = 0 > (int8_t)a ? 0 : a; if (!(0 > (int8_t)a)) d |= 0x08;
					y8[t] = 0 > (int8_t)b ? 0 : b; if (!(0 > (int8_t)b)) d |= 0x10;
				}
				if (with_cigar) pr[t] = d;
			}
		}
		if (!approx_max) {
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u8[en0] - qe : H[en0] + v8[en0] - qe;
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4) {
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v8[t + i] - qe;
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v8[t] - qe;
					if (H[t] > max_H)
						max_H = H[t], max_t = t;
				}
			} else H[0] = v8[0] - qe - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte)
				ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe)
				ez->mqe = H[st0], ez->mqe_t = st0;
			if (orc_apply_zdrop(ez, max_H, r, max_t, zdrop, e)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1)
				ez->score = H[tlen - 1];
		} else {
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v8[last_H0_t] - qe;
					int32_t d1 = u8[last_H0_t + 1] - qe;
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v8[last_H0_t] - qe;
				} else {
					++last_H0_t, H0 += u8[last_H0_t] - qe;
				}
				if ((flag & ORC_EZ_APPROX_DROP) && orc_apply_zdrop(ez, H0, r, last_H0_t, zdrop, e)) break;
			} else H0 = v8[0] - qe - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1)
				ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) {
		int rev_cigar = !!(flag & ORC_EZ_REV_CIGAR);
		cg.c = cigar, cg.n = 0, cg.cap = cigar_cap, cg.overflow = 0;
		if (!ez->zdropped && !(flag & ORC_EZ_EXTZ_ONLY)) {
			orc_backtrack(rev_cigar, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1, &cg);
		} else if (!ez->zdropped && (flag & ORC_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			orc_backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->mqe_t, qlen - 1, &cg);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) {
			orc_backtrack(rev_cigar, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q, &cg);
		}
		ez->n_cigar = cg.n, ez->cigar_overflow = cg.overflow;
		free(p);
		free(off);
	}
}
