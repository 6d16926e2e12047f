// ksw_kernels.hip -- K4 `extd2_dp`: the banded dual-affine anti-diagonal DP on gfx950.
// See ksw_device.h for the design.  Two kernels:
//   extd2_reg_kernel<K>  : fast path. tlen rounded to 16 fits 64*K columns (K<=5, i.e. tlen<=320),
//                          DP state in VGPRs, direction bytes in LDS, traceback from LDS.
//   extd2_lds_kernel     : general path (any shape the C ABI accepts, all KSW_EZ flags, and the
//                          single-affine extz2 variant): DP state in LDS, direction bytes in a
//                          global scratch slab.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ksw_device.h"

namespace psvr {

__device__ __forceinline__ void write_ez(psvr_extz_t *o, const EzAcc &a, int n_cigar)
{
	o->max = a.max, o->zdropped = a.zdropped;
	o->max_q = a.max_q, o->max_t = a.max_t;
	o->mqe = a.mqe, o->mqe_t = a.mqe_t;
	o->mte = a.mte, o->mte_q = a.mte_q;
	o->score = a.score, o->n_cigar = n_cigar, o->reach_end = a.reach_end, o->reserved = 0;
}

// ------------------------------------------------------------------------------------------
// fast path
// ------------------------------------------------------------------------------------------
// the anti-diagonal sweep.  WRAP = emulate the 8-bit wrap-around of every SSE add/sub (needed whenever an in-band cell can
// read a lane outside the band); without it the arithmetic stays in 32 bits, which is value-identical when wrap cannot be
// observed (see extd2_reg_kernel).
template <int K, bool WRAP>
__device__ __forceinline__ void dp_main_loop(const DpParams &P, const uint8_t *target, int lane, int qlen, int tlen, int w, int rowb, int n_rows,
                                             const uint8_t *QR, uint8_t *Pm, EzAcc &ez)
{
#define W8(e) (WRAP ? s8(e) : (int)(e))
	const int neg_qe = s8(-P.q - P.e), neg_qe2 = s8(-P.q2 - P.e2);
	const int qe8 = s8(P.q + P.e), qe28 = s8(P.q2 + P.e2);
	int u[K], v[K], x[K], y[K], x2[K], y2[K], s[K], H[K], tb[K];
#pragma unroll
	for (int c = 0; c < K; ++c) {
		int t = c * 64 + lane;
		u[c] = v[c] = x[c] = y[c] = neg_qe;
		x2[c] = y2[c] = neg_qe2;
		s[c] = 0;
		H[c] = PSVR_KSW_NEG_INF;
		tb[c] = t < tlen ? target[t] : 0;
	}
	__builtin_amdgcn_wave_barrier();      // the query image was written by this wave's own lanes (LDS is in-order per wave)

	int last_st = -1;
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	for (int r = 0; r < n_rows; ++r) {
		int st0, en0, st, en;
		if (!band_limits(r, qlen, tlen, w, st0, en0, st, en)) { ez.zdropped = 1; break; }
		const bool adv = st > 0 && st > last_st;   // (r-1,st-1) was computed last round (:143)
		const int ur = r == 0 ? neg_qe : r < P.long_thres ? s8(-P.e) : r == P.long_thres ? s8(P.long_diff) : s8(-P.e2);
		const int fresh_end = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;  // score groups of 16 from st0 (:159)
		const int qbase = qlen - 1 - r;
		const int c_first = st >> 6, c_last = en >> 6;
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		int h_prev = 0;                                  // H[en0-1] of the previous diagonal (:322)
		if (en0 > 0) {
#pragma unroll
			for (int c = 0; c < K; ++c)
				if (((en0 - 1) >> 6) == c) h_prev = __builtin_amdgcn_readlane(H[c], (en0 - 1) & 63);
		}
		int bh = (int)0x80000000; unsigned bk = 0xffffffffu;
		uint8_t *prow = Pm + r * rowb - st;
#pragma unroll
		for (int c = K - 1; c >= 0; --c) {
			if (c < c_first || c > c_last) continue;
			const int t = c * 64 + lane;
			// branch-free: every lane computes, the state of lanes outside [st,en] is kept by selects (v_cndmask)
			const bool act = (t >= st) & (t <= en);
			const bool ovr = (en >= r) & (t == r);                                  // (:153-156); lane r is always inside [st,en]
			const int yy = ovr ? neg_qe : y[c], yy2 = ovr ? neg_qe2 : y2[c], ut = ovr ? ur : u[c];
			int cx = 0, cv = 0, cx2 = 0;
			if (c > 0) {
				cx = __builtin_amdgcn_readlane(x[c - 1], 63);
				cv = __builtin_amdgcn_readlane(v[c - 1], 63);
				cx2 = __builtin_amdgcn_readlane(x2[c - 1], 63);
			}
			int xt1 = dpp_wave_shr1(x[c], cx), vt1 = dpp_wave_shr1(v[c], cv), x2t1 = dpp_wave_shr1(x2[c], cx2);
			const bool bnd = (t == st) & !adv;                                       // (:142-152)
			xt1 = bnd ? neg_qe : xt1, x2t1 = bnd ? neg_qe2 : x2t1, vt1 = bnd ? (st > 0 ? neg_qe : ur) : vt1;
			const bool fresh = (t >= st0) & (t <= fresh_end);                          // fresh score (:158-173)
			const int qb = QR[fresh ? qbase + t : 0];
			int sc = tb[c] == qb ? P.sc_mch : P.sc_mis;
			sc = ((tb[c] == P.m1) | (qb == P.m1)) ? P.sc_N : sc;
			const int sv = fresh ? sc : s[c];
			s[c] = sv;
			int z = sv;
			int a = W8(xt1 + vt1), b = W8(yy + ut), a2 = W8(x2t1 + vt1), b2 = W8(yy2 + ut);
			int d = a > z ? 1 : 0;   z = max(z, a);
			d = b > z ? 2 : d;       z = max(z, b);
			d = a2 > z ? 3 : d;      z = max(z, a2);
			d = b2 > z ? 4 : d;      z = max(z, b2);
			z = min(z, P.sc_mch);
			const int un = W8(z - vt1), vn = W8(z - ut);
			int tmp = W8(z - P.q);
			a = W8(a - tmp), b = W8(b - tmp);
			tmp = W8(z - P.q2);
			a2 = W8(a2 - tmp), b2 = W8(b2 - tmp);
			d |= (a > 0 ? 0x08 : 0) | (b > 0 ? 0x10 : 0) | (a2 > 0 ? 0x20 : 0) | (b2 > 0 ? 0x40 : 0);
			u[c] = act ? un : u[c], v[c] = act ? vn : v[c];
			x[c] = act ? W8(max(a, 0) - qe8) : x[c];
			y[c] = act ? W8(max(b, 0) - qe8) : y[c];
			x2[c] = act ? W8(max(a2, 0) - qe28) : x2[c];
			y2[c] = act ? W8(max(b2, 0) - qe28) : y2[c];
			if (act & (with_cigar != 0)) prow[t] = (uint8_t)d;
			// exact H tracking (:316-351)
			const int hold = H[c];
			int hn = (t == en0) ? (en0 > 0 ? h_prev + un : hold + vn) : (((t >= st0) & (t < en0)) ? hold + vn : hold);
			hn = r == 0 ? (t == 0 ? vn - P.qe_pre : hold) : hn;
			hn = act ? hn : hold;
			H[c] = hn;
			const bool valid = (t >= st0) & (t <= en0);
			const unsigned rank = t == en0 ? 0u : (t < en1 ? 1u + (unsigned)((t - st0) & 3) * 4096u + (unsigned)(t - st0)
			                                                : 1u + 4u * 4096u + (unsigned)(t - st0));
			const bool better = valid & ((hn > bh) | ((hn == bh) & (rank < bk)));
			bh = better ? hn : bh, bk = better ? rank : bk;
		}
		const int max_H = wave_max_i32(bh);
		// arg-max with the reference's tie order: usually one lane holds the maximum, then its rank is a single v_readlane
		const unsigned long long top = __ballot(bh == max_H);
		unsigned rk;
		if (__popcll(top) == 1) rk = (unsigned)__builtin_amdgcn_readlane((int)bk, __ffsll((unsigned long long)top) - 1);
		else rk = wave_min_u32(bh == max_H ? bk : 0xffffffffu);
		const int max_t = rk == 0 ? en0 : st0 + (int)((rk - 1u) & 4095u);
		int H_en0 = 0, H_st0 = 0;
#pragma unroll
		for (int c = 0; c < K; ++c) {
			if ((en0 >> 6) == c) H_en0 = __builtin_amdgcn_readlane(H[c], en0 & 63);
			if ((st0 >> 6) == c) H_st0 = __builtin_amdgcn_readlane(H[c], st0 & 63);
		}
		if (en0 == tlen - 1 && H_en0 > ez.mte) ez.mte = H_en0, ez.mte_q = r - en;
		if (r - st0 == qlen - 1 && H_st0 > ez.mqe) ez.mqe = H_st0, ez.mqe_t = st0;
		if (ez.apply_zdrop(max_H, r, max_t, P.zdrop, P.e2)) break;
		if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H_en0;
		last_st = st;
	}
#undef W8
}

// The same sweep for matrices wider than 64 K columns whose BAND fits 64 K columns (`fc_sv`'s contig re-alignment: q/t up to 3100 at
// w = 132, SignalAssembly.hpp:418-420,463; reads beyond 320 bases on the `aln` path at its fixed w = 200): the lanes hold a RING of
// R = 64 K columns -- column t lives in slot t mod R (chunk (t mod R) / 64, lane t mod 64) -- that slides along the matrix with the band.
// The reference's per-column arrays (u, v, x, y, x2, y2, s, H) are only ever touched inside [st, max(en, fresh_end)] of the current
// anti-diagonal (ksw2_extd2_sse.c:125-140,158-173), that interval moves right monotonically and is at most w + 32 columns wide, so a
// column that has fallen out of it on the left is dead and its slot is given to column t + R -- with the arrays' initial values
// (:100-121) -- when that one comes into reach on the right.  The neighbour (r-1, t-1) is the lane to the left as before, lane 0 of
// chunk c receiving lane 63 of chunk c-1 and chunk 0 that of chunk K-1 (read before any chunk is updated).  Everything else -- the
// 16-lane block rounding, the stale-score lanes, the 8-bit wrap of every add/sub -- is dp_main_loop's, cell for cell.
template <int K>
__device__ __forceinline__ void dp_ring_loop(const DpParams &P, const uint8_t *target, int lane, int qlen, int tlen, int w, int rowb, int n_rows,
                                             const uint8_t *QR, uint8_t *Pm, EzAcc &ez)
{
	constexpr int R = 64 * K;
	const int neg_qe = s8(-P.q - P.e), neg_qe2 = s8(-P.q2 - P.e2);
	const int qe8 = s8(P.q + P.e), qe28 = s8(P.q2 + P.e2);
	int u[K], v[K], x[K], y[K], x2[K], y2[K], s[K], H[K], tb[K], tc[K];
#pragma unroll
	for (int c = 0; c < K; ++c) {
		const int t = c * 64 + lane;
		tc[c] = t;
		u[c] = v[c] = x[c] = y[c] = neg_qe;
		x2[c] = y2[c] = neg_qe2;
		s[c] = 0;
		H[c] = PSVR_KSW_NEG_INF;
		tb[c] = t < tlen ? target[t] : 0;
	}
	__builtin_amdgcn_wave_barrier();      // the query / target images were written by this wave's own lanes (LDS is in-order per wave)

	int last_st = -1;
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	for (int r = 0; r < n_rows; ++r) {
		int st0, en0, st, en;
		if (!band_limits(r, qlen, tlen, w, st0, en0, st, en)) { ez.zdropped = 1; break; }
		const bool adv = st > 0 && st > last_st;   // (r-1,st-1) was computed last round (:143)
		const int ur = r == 0 ? neg_qe : r < P.long_thres ? s8(-P.e) : r == P.long_thres ? s8(P.long_diff) : s8(-P.e2);
		const int fresh_end = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;  // score groups of 16 from st0 (:159)
		const int qbase = qlen - 1 - r;
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		const int reach = en > fresh_end ? en : fresh_end;
		// a slot whose next column has come into reach starts over with the arrays' initial values (:100-121) -- before its neighbour to the
		// right, which may enter the band on this very diagonal (en moves 16 columns at a time), looks at it.  Its old column lies left of
		// st - 1 (the interval [st - 1, reach] is at most w + 33 <= R columns wide), so nobody reads that one again.
#pragma unroll
		for (int c = 0; c < K; ++c) {
			if (tc[c] + R <= reach) {
				tc[c] += R;
				u[c] = v[c] = x[c] = y[c] = neg_qe;
				x2[c] = y2[c] = neg_qe2;
				s[c] = 0;
				H[c] = PSVR_KSW_NEG_INF;
				tb[c] = tc[c] < tlen ? target[tc[c]] : 0;
			}
		}
		// the neighbours' values of the previous diagonal that cross a chunk boundary, and H[en0-1] (:322), before anything is updated
		int cx[K], cv[K], cx2[K];
#pragma unroll
		for (int c = 0; c < K; ++c) {
			const int src = (c + K - 1) % K;
			cx[c] = __builtin_amdgcn_readlane(x[src], 63);
			cv[c] = __builtin_amdgcn_readlane(v[src], 63);
			cx2[c] = __builtin_amdgcn_readlane(x2[src], 63);
		}
		int h_prev = 0;
		if (en0 > 0) {
			const int j = (en0 - 1) % R;
#pragma unroll
			for (int c = 0; c < K; ++c)
				if ((j >> 6) == c) h_prev = __builtin_amdgcn_readlane(H[c], j & 63);
		}
		int bh = (int)0x80000000; unsigned bk = 0xffffffffu;
		uint8_t *prow = Pm + (size_t)r * rowb - st;
#pragma unroll
		for (int c = 0; c < K; ++c) {
			const int t = tc[c];
			const bool act = (t >= st) & (t <= en);
			const bool ovr = (en >= r) & (t == r);                                  // (:153-156); lane r is always inside [st,en]
			const int yy = ovr ? neg_qe : y[c], yy2 = ovr ? neg_qe2 : y2[c], ut = ovr ? ur : u[c];
			int xt1 = dpp_wave_shr1(x[c], cx[c]), vt1 = dpp_wave_shr1(v[c], cv[c]), x2t1 = dpp_wave_shr1(x2[c], cx2[c]);   // (r-1,t-1)
			const bool bnd = (t == st) & !adv;                                       // (:142-152)
			xt1 = bnd ? neg_qe : xt1, x2t1 = bnd ? neg_qe2 : x2t1, vt1 = bnd ? (st > 0 ? neg_qe : ur) : vt1;
			const bool fresh = (t >= st0) & (t <= fresh_end);                          // fresh score (:158-173)
			const int qb = QR[fresh ? qbase + t : 0];
			int sc = tb[c] == qb ? P.sc_mch : P.sc_mis;
			sc = ((tb[c] == P.m1) | (qb == P.m1)) ? P.sc_N : sc;
			const int sv = fresh ? sc : s[c];
			s[c] = sv;
			int z = sv;
			int a = s8(xt1 + vt1), b = s8(yy + ut), a2 = s8(x2t1 + vt1), b2 = s8(yy2 + ut);
			int d = a > z ? 1 : 0;   z = max(z, a);
			d = b > z ? 2 : d;       z = max(z, b);
			d = a2 > z ? 3 : d;      z = max(z, a2);
			d = b2 > z ? 4 : d;      z = max(z, b2);
			z = min(z, P.sc_mch);
			const int un = s8(z - vt1), vn = s8(z - ut);
			int tmp = s8(z - P.q);
			a = s8(a - tmp), b = s8(b - tmp);
			tmp = s8(z - P.q2);
			a2 = s8(a2 - tmp), b2 = s8(b2 - tmp);
			d |= (a > 0 ? 0x08 : 0) | (b > 0 ? 0x10 : 0) | (a2 > 0 ? 0x20 : 0) | (b2 > 0 ? 0x40 : 0);
			u[c] = act ? un : u[c], v[c] = act ? vn : v[c];
			x[c] = act ? s8(max(a, 0) - qe8) : x[c];
			y[c] = act ? s8(max(b, 0) - qe8) : y[c];
			x2[c] = act ? s8(max(a2, 0) - qe28) : x2[c];
			y2[c] = act ? s8(max(b2, 0) - qe28) : y2[c];
			if (act & (with_cigar != 0)) prow[t] = (uint8_t)d;
			// exact H tracking (:316-351)
			const int hold = H[c];
			int hn = (t == en0) ? (en0 > 0 ? h_prev + un : hold + vn) : (((t >= st0) & (t < en0)) ? hold + vn : hold);
			hn = r == 0 ? (t == 0 ? vn - P.qe_pre : hold) : hn;
			hn = act ? hn : hold;
			H[c] = hn;
			const bool valid = (t >= st0) & (t <= en0);
			const unsigned rank = t == en0 ? 0u : (t < en1 ? 1u + (unsigned)((t - st0) & 3) * 4096u + (unsigned)(t - st0)
			                                                : 1u + 4u * 4096u + (unsigned)(t - st0));
			const bool better = valid & ((hn > bh) | ((hn == bh) & (rank < bk)));
			bh = better ? hn : bh, bk = better ? rank : bk;
		}
		const int max_H = wave_max_i32(bh);
		const unsigned long long top = __ballot(bh == max_H);
		unsigned rk;
		if (__popcll(top) == 1) rk = (unsigned)__builtin_amdgcn_readlane((int)bk, __ffsll((unsigned long long)top) - 1);
		else rk = wave_min_u32(bh == max_H ? bk : 0xffffffffu);
		const int max_t = rk == 0 ? en0 : st0 + (int)((rk - 1u) & 4095u);
		int H_en0 = 0, H_st0 = 0;
		{
			const int je = en0 % R, js = st0 % R;
#pragma unroll
			for (int c = 0; c < K; ++c) {
				if ((je >> 6) == c) H_en0 = __builtin_amdgcn_readlane(H[c], je & 63);
				if ((js >> 6) == c) H_st0 = __builtin_amdgcn_readlane(H[c], js & 63);
			}
		}
		if (en0 == tlen - 1 && H_en0 > ez.mte) ez.mte = H_en0, ez.mte_q = r - en;
		if (r - st0 == qlen - 1 && H_st0 > ez.mqe) ez.mqe = H_st0, ez.mqe_t = st0;
		if (ez.apply_zdrop(max_H, r, max_t, P.zdrop, P.e2)) break;
		if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H_en0;
		last_st = st;
	}
}

// The sweep for problems whose band never clips the matrix and whose in-band values fit int8 (P.nowrap_ok &&
// dp_band_never_binds): the common case on the `aln` path (qlen <= 200, tlen <= 201).  Same results as dp_main_loop, far
// fewer instructions per anti-diagonal:
//   * an in-band cell only ever reads in-band cells of the previous diagonal or one of the explicit boundary values
//     (:142-156): the 16-lane block rounding, the stale-score lanes and the 8-bit wrap are not modelled at all, and lanes
//     below the band may hold anything;
//   * the left boundary of the band is simply the carry shifted into lane 0 of chunk 0; the first-row boundary is the
//     initial state of a lane, which it keeps until its column enters the band (updates run under the band's exec mask);
//   * all band / z-drop / end-score bookkeeping is wave-uniform (SGPRs, scalar branches);
//   * the arg-max lane is found with a ballot; the reference's 4-lane tie order is only evaluated when two lanes tie.
template <int K>
__device__ __forceinline__ void dp_lean_loop(const DpParams &P, const uint8_t *target, int lane, int qlen, int tlen, int rowb, int n_rows,
                                             const uint8_t *QR, uint8_t *Pm, EzAcc &ez_out)
{
	const int neg_qe = s8(-P.q - P.e), neg_qe2 = s8(-P.q2 - P.e2);
	const int qe8 = s8(P.q + P.e), qe28 = s8(P.q2 + P.e2);
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	int u[K], v[K], x[K], y[K], x2[K], y2[K], H[K], tb[K], mis[K];
#pragma unroll
	for (int c = 0; c < K; ++c) {
		const int t = c * 64 + lane;
		v[c] = x[c] = y[c] = neg_qe;
		x2[c] = y2[c] = neg_qe2;
		// lanes keep these values until their column enters the band at r == t: u/y/y2 of the first cell of column t (:153-156)
		u[c] = t == 0 ? neg_qe : t < P.long_thres ? s8(-P.e) : t == P.long_thres ? s8(P.long_diff) : s8(-P.e2);
		H[c] = -P.qe_pre;                 // H[0] = v - qe on the first diagonal (:351); every other lane is set before it is read
		const int tc = t < tlen ? target[t] : 0;
		tb[c] = tc == P.m1 ? 0x100 : tc;  // an N never equals a query code
		mis[c] = tc == P.m1 ? P.sc_N : P.sc_mis;
	}
	__builtin_amdgcn_wave_barrier();
	int e_max = 0, e_max_t = -1, e_max_q = -1, e_mqe = PSVR_KSW_NEG_INF, e_mqe_t = -1, e_mte = PSVR_KSW_NEG_INF, e_mte_q = -1;
	int e_score = PSVR_KSW_NEG_INF, e_zd = 0, last_H = PSVR_KSW_NEG_INF;
	for (int r = 0; r < n_rows; ++r) {
		const int st0 = max(0, r - qlen + 1), en0 = min(tlen - 1, r);
		int ur = s8(-P.e2);                // v of column -1 (:142-152); only the first long_thres + 1 diagonals differ
		if (r <= P.long_thres) ur = r == 0 ? neg_qe : r < P.long_thres ? s8(-P.e) : s8(P.long_diff);
		const int c_first = st0 >> 6, c_last = en0 >> 6;
		const int qbase = qlen - 1 - r;
		const unsigned prow = (unsigned)(r * rowb - (st0 & ~15));
		int h_prev = 0;
#pragma unroll
		for (int c = 0; c < K; ++c) {
			if (en0 > 0 && ((en0 - 1) >> 6) == c) h_prev = __builtin_amdgcn_readlane(H[c], (en0 - 1) & 63);
		}
		const int sel_t = en0 > 0 ? en0 : -1;
		int bh = (int)0x80000000;
		unsigned long long amask[K];
#pragma unroll
		for (int c = K - 1; c >= 0; --c) {
			amask[c] = 0;
			if (c < c_first || c > c_last) continue;
			const int t = c * 64 + lane;
			// (r-1,t-1): lane 0 of chunk 0 receives the boundary values of column -1 (:142-152), other chunks the old lane 63
			int cx = neg_qe, cv = ur, cx2 = neg_qe2;
			if (c > 0) {
				cx = __builtin_amdgcn_readlane(x[c - 1], 63);
				cv = __builtin_amdgcn_readlane(v[c - 1], 63);
				cx2 = __builtin_amdgcn_readlane(x2[c - 1], 63);
			}
			const int xt1 = dpp_wave_shr1(x[c], cx), vt1 = dpp_wave_shr1(v[c], cv), x2t1 = dpp_wave_shr1(x2[c], cx2);
			const int qb = QR[(unsigned)(qbase + t)];
			int sc = tb[c] == qb ? P.sc_mch : mis[c];
			sc = qb == P.m1 ? P.sc_N : sc;
			const int ut = u[c];
			int a = xt1 + vt1, b = y[c] + ut, a2 = x2t1 + vt1, b2 = y2[c] + ut;
			int z = max(max(sc, a), b);
			z = max(max(z, a2), b2);
			int d = 4;                                     // first of {sc,a,b,a2,b2} that reaches the maximum (:176-213)
			d = a2 == z ? 3 : d;
			d = b == z ? 2 : d;
			d = a == z ? 1 : d;
			d = sc == z ? 0 : d;
			z = min(z, P.sc_mch);
			const int un = z - vt1, vn = z - ut;
			const int zq = z - P.q, zq2 = z - P.q2;
			a -= zq, b -= zq, a2 -= zq2, b2 -= zq2;
			d |= (a > 0 ? 0x08 : 0) | (b > 0 ? 0x10 : 0) | (a2 > 0 ? 0x20 : 0) | (b2 > 0 ? 0x40 : 0);
			// exact H (:316-351): the last in-band lane continues from its left neighbour, the others from themselves
			const int hn = t == sel_t ? h_prev + un : H[c] + vn;
			const bool act = (t >= st0) & (t <= en0);
			amask[c] = __ballot(act);
			if (act) {   // lanes above the band must keep their initial u/y/y2; what lanes below it hold no longer matters
				u[c] = un, v[c] = vn;
				x[c] = max(a, 0) - qe8, y[c] = max(b, 0) - qe8;
				x2[c] = max(a2, 0) - qe28, y2[c] = max(b2, 0) - qe28;
				H[c] = hn;
				if (with_cigar) Pm[prow + (unsigned)t] = (uint8_t)d;
				bh = max(bh, hn);
			}
		}
		const int max_H = wave_max_i32(bh);
		// arg-max: usually a single lane holds the maximum
		int n_top = 0, max_t = 0;
		unsigned long long top[K];
#pragma unroll
		for (int c = 0; c < K; ++c) {
			top[c] = 0;
			if (c < c_first || c > c_last) continue;
			top[c] = __ballot(H[c] == max_H) & amask[c];
			n_top += __popcll(top[c]);
			if (top[c]) max_t = c * 64 + __ffsll((unsigned long long)top[c]) - 1;
		}
		if (n_top != 1) {                                  // the reference's order among equal lanes (:322-349)
			const int en1 = st0 + (en0 - st0) / 4 * 4;
			unsigned bk = 0xffffffffu;
#pragma unroll
			for (int c = 0; c < K; ++c) {
				const int t = c * 64 + lane;
				const unsigned rank = t == en0 ? 0u : (t < en1 ? 1u + (unsigned)((t - st0) & 3) * 4096u + (unsigned)(t - st0)
				                                                : 1u + 4u * 4096u + (unsigned)(t - st0));
				if ((top[c] >> lane) & 1) bk = min(bk, rank);
			}
			const unsigned rk = wave_min_u32(bk);
			max_t = rk == 0 ? en0 : st0 + (int)((rk - 1u) & 4095u);
		}
		if (en0 == tlen - 1) {
			int H_en0 = 0;
#pragma unroll
			for (int c = 0; c < K; ++c)
				if ((en0 >> 6) == c) H_en0 = __builtin_amdgcn_readlane(H[c], en0 & 63);
			if (H_en0 > e_mte) e_mte = H_en0, e_mte_q = r - (((en0 + 16) & ~15) - 1);
			if (r == n_rows - 1) last_H = H_en0;
		}
		if (r - st0 == qlen - 1) {
			int H_st0 = 0;
#pragma unroll
			for (int c = 0; c < K; ++c)
				if ((st0 >> 6) == c) H_st0 = __builtin_amdgcn_readlane(H[c], st0 & 63);
			if (H_st0 > e_mqe) e_mqe = H_st0, e_mqe_t = st0;
		}
		// ksw_apply_zdrop, is_rot = 1 (ksw2.h:245-261)
		if (max_H > e_max) {
			e_max = max_H, e_max_t = max_t, e_max_q = r - max_t;
		} else if (max_t >= e_max_t && r - max_t >= e_max_q) {
			const int tl = max_t - e_max_t, ql = (r - max_t) - e_max_q;
			const int l = tl > ql ? tl - ql : ql - tl;
			if (P.zdrop >= 0 && e_max - max_H > P.zdrop + l * P.e2) { e_zd = 1; break; }
		}
		if (r == n_rows - 1) e_score = last_H;             // only when the last diagonal did not z-drop (:350-351 come after the check)
	}
	ez_out.max = e_max, ez_out.max_t = e_max_t, ez_out.max_q = e_max_q;
	ez_out.mqe = e_mqe, ez_out.mqe_t = e_mqe_t, ez_out.mte = e_mte, ez_out.mte_q = e_mte_q;
	ez_out.score = e_score, ez_out.zdropped = e_zd;
}

// kDpWaves independent alignments per workgroup (one per wavefront, no inter-wave communication): single-wave workgroups
// run into the workgroups-per-CU limit long before the wave slots are full
// RING: the matrix is wider than 64 K columns but its band is not (dp_ring_loop); the target is staged in LDS behind the query image
// (a slot that is handed on fetches its new column's base from there), the direction bytes are in the HBM slab (PG)
template <int K, bool PG, bool RING>
__device__ __forceinline__ void extd2_wave_body(const DpBatch &B, const DpParams &P)
{
	extern __shared__ __align__(16) uint8_t lds_all[];
	// everything per-alignment is wave-uniform; threadIdx.x >> 6 is not provably so for the compiler, and without the
	// readfirstlane all band / z-drop bookkeeping of the sweep is done per lane in VALU with exec-mask control flow
	const int wave = uni(threadIdx.x >> 6);
	const long long slot = (long long)blockIdx.x * kDpWaves + wave;
	if (slot >= B.n) return;
	uint8_t *lds = lds_all + (size_t)wave * B.lds_per_wave;
	const int pid = uni(B.idx[slot]);
	const int lane = threadIdx.x & 63;
	const int qlen = uni(B.qlen[pid]), tlen = uni(B.tlen[pid]);
	psvr_extz_t *out = B.ez + pid;
	EzAcc ez;
	ez.reset();
	if (P.skip || qlen <= 0 || tlen <= 0) {
		if (lane == 0) write_ez(out, ez, 0);
		return;
	}
	const uint8_t *query = B.qseq + uni64(B.q_off[pid]), *target = B.tseq + uni64(B.t_off[pid]);
	const int w = P.w < 0 ? (tlen > qlen ? tlen : qlen) : P.w;
	int n_col = qlen < tlen ? qlen : tlen;
	n_col = ((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1;
	const int rowb = n_col * 16;
	const int n_rows = qlen + tlen - 1;
	const int qimg = (qlen + 16 + 15) & ~15;
	uint8_t *QR = lds;                 // reversed query + >=16 zero bytes (the calloc'ed tail of `qr`, :100,121)
	// direction bytes, row pitch rowb (:115): behind the query image in LDS, or (PG) in this problem's slice of the HBM slab
	uint8_t *Pm = PG ? B.pslab + (uni64(B.p_off[pid]) << B.p_unit_shift) : lds + qimg;
	const int p_end = n_rows * rowb + 16;
	for (int i = lane; i < qimg; i += 64) QR[i] = i < qlen ? query[qlen - 1 - i] : 0;
	if (RING) {
		uint8_t *TG = lds + qimg;
		for (int i = lane; i < tlen; i += 64) TG[i] = target[i];
		dp_ring_loop<K>(P, TG, lane, qlen, tlen, w, rowb, n_rows, QR, Pm, ez);
	}
	// 8-bit wrap-around only has to be emulated when it can be observed: if the band never clips the matrix, in-band cells
	// never read a lane outside the band (dp_band_never_binds) and all in-band values fit int8 for these scoring parameters
	// (P.nowrap_ok, make_dp_params), so the sign-extension after every add/sub is dropped
	else if (P.nowrap_ok && dp_band_never_binds(qlen, tlen, w)) dp_lean_loop<K>(P, target, lane, qlen, tlen, rowb, n_rows, QR, Pm, ez);
	else dp_main_loop<K, true>(P, target, lane, qlen, tlen, w, rowb, n_rows, QR, Pm, ez);
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	int n_cigar = 0;
	if (with_cigar) {
		if (PG) __threadfence_block();
		__builtin_amdgcn_wave_barrier();
		int i0 = -1, j0 = -1;
		if (!ez.zdropped && !(P.flag & PSVR_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
		else if (!ez.zdropped && (P.flag & PSVR_EZ_EXTZ_ONLY) && ez.mqe + P.end_bonus > ez.max) ez.reach_end = 1, i0 = ez.mqe_t, j0 = qlen - 1;
		else if (ez.max_t >= 0 && ez.max_q >= 0) i0 = ez.max_t, j0 = ez.max_q;
		if (i0 >= 0 && j0 >= 0) {
			// ops are staged in the already-consumed tail of the direction-byte area (rows > r are dead)
			uint32_t *stage_end = (uint32_t*)(Pm + p_end);
			n_cigar = traceback(i0, j0, qlen, tlen, w,
				[&](int r, int k) { return PG ? (int)__builtin_nontemporal_load(Pm + (size_t)r * rowb + k) : (int)Pm[r * rowb + k]; },
				[&](int k, uint32_t word) { if (lane == 0) stage_end[-1 - k] = word; });
			if (PG) __threadfence_block();
			__builtin_amdgcn_wave_barrier();
			uint32_t *dst = B.cigar + uni64(out->cigar_off);
			const bool rev = (P.flag & PSVR_EZ_REV_CIGAR) != 0;
			for (int m = lane; m < n_cigar; m += 64) {
				const uint32_t *src = rev ? stage_end - 1 - m : stage_end - n_cigar + m;
				dst[m] = PG ? __builtin_nontemporal_load(src) : *src;
			}
		}
	}
	if (lane == 0) write_ez(out, ez, n_cigar);
}
template <int K, bool PG>
__global__ __launch_bounds__(64 * kDpWaves) void extd2_reg_kernel(DpBatch B, DpParams P) { extd2_wave_body<K, PG, false>(B, P); }
template <int K>
__global__ __launch_bounds__(64 * kDpWaves) void extd2_ring_kernel(DpBatch B, DpParams P) { extd2_wave_body<K, true, true>(B, P); }
template __global__ void extd2_ring_kernel<3>(DpBatch, DpParams);
template __global__ void extd2_ring_kernel<4>(DpBatch, DpParams);

template __global__ void extd2_reg_kernel<1, false>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<2, false>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<3, false>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<4, false>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<5, false>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<1, true>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<2, true>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<3, true>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<4, true>(DpBatch, DpParams);
template __global__ void extd2_reg_kernel<5, true>(DpBatch, DpParams);


// ------------------------------------------------------------------------------------------
// tiny problems: one THREAD per alignment
// ------------------------------------------------------------------------------------------
// Most DP calls of the `aln` path are end-to-end gap fills of a few bases (median 6 x 6): a wavefront per problem spends its
// time in per-diagonal bookkeeping with a handful of lanes active.  Here 64 problems with qlen, tlen <= 16 share a wavefront,
// each thread sweeping its own matrix anti-diagonal by anti-diagonal in the lean regime of dp_lean_loop (same recurrences,
// same boundary values, same tie order, same z-drop rule; nothing of the SSE layout can be observed at this size).
// Per-thread state lives in LDS in [slot][thread] order (bank-conflict free): per column t one word u|v|x|y (int8 each) and
// one word x2|y2|H (int8, int8, int16), plus the direction bytes of rows x 16 cells; the CIGAR is staged in the dead state words.
__global__ __launch_bounds__(64) void extd2_tiny_kernel(DpBatch B, DpParams P, int max_rows)
{
	extern __shared__ __align__(16) uint32_t tiny_lds[];
	const int lane = threadIdx.x;
	const long long slot = (long long)blockIdx.x * 64 + lane;
	if (slot >= B.n) return;
	uint32_t *W0 = tiny_lds + lane, *W1 = tiny_lds + 16 * 64 + lane;       // element t at [t * 64]
	uint8_t *PD = (uint8_t *)(tiny_lds + 2 * 16 * 64) + lane;              // cell (r, t) at [(r * 16 + t) * 64]
	const int pid = B.idx[slot];
	const int qlen = B.qlen[pid], tlen = B.tlen[pid];
	psvr_extz_t *out = B.ez + pid;
	EzAcc ez;
	ez.reset();
	if (P.skip || qlen <= 0 || tlen <= 0 || qlen > PSVR_DP_TINY_MAX || tlen > PSVR_DP_TINY_MAX || qlen + tlen - 1 > max_rows) { write_ez(out, ez, 0); return; }
	const uint8_t *query = B.qseq + B.q_off[pid], *target = B.tseq + B.t_off[pid];
	unsigned long long qp = 0, tp = 0;                                      // 4 bits per base
	for (int i = 0; i < qlen; ++i) qp |= (unsigned long long)(query[i] & 15) << (4 * i);
	for (int i = 0; i < tlen; ++i) tp |= (unsigned long long)(target[i] & 15) << (4 * i);
	const int neg_qe = s8(-P.q - P.e), neg_qe2 = s8(-P.q2 - P.e2);
	const int qe8 = s8(P.q + P.e), qe28 = s8(P.q2 + P.e2);
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	auto pack0 = [](int u, int v, int x, int y) { return (uint32_t)(u & 0xff) | (uint32_t)(v & 0xff) << 8 | (uint32_t)(x & 0xff) << 16 | (uint32_t)(y & 0xff) << 24; };
	auto pack1 = [](int x2, int y2, int h) { return (uint32_t)(x2 & 0xff) | (uint32_t)(y2 & 0xff) << 8 | (uint32_t)(h & 0xffff) << 16; };
	auto ur_of = [&](int r) { return r == 0 ? neg_qe : r < P.long_thres ? s8(-P.e) : r == P.long_thres ? s8(P.long_diff) : s8(-P.e2); };
	for (int t = 0; t < tlen; ++t) {
		W0[t * 64] = pack0(ur_of(t), neg_qe, neg_qe, neg_qe);                // u/y/y2 of the first cell of column t (:153-156)
		W1[t * 64] = pack1(neg_qe2, neg_qe2, -P.qe_pre);
	}
	const int n_rows = qlen + tlen - 1;
	for (int r = 0; r < n_rows; ++r) {
		const int st0 = max(0, r - qlen + 1), en0 = min(tlen - 1, r);
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		// (r-1, st0-1): the boundary column -1 (:142-152) or the old state of the slot left of the band
		int x1 = neg_qe, v1 = ur_of(r), x21 = neg_qe2, h1 = 0;
		if (st0 > 0) {
			const uint32_t a = W0[(st0 - 1) * 64], b = W1[(st0 - 1) * 64];
			v1 = s8(a >> 8), x1 = s8(a >> 16), x21 = s8(b), h1 = (int)(int16_t)(b >> 16);
		}
		int bh = (int)0x80000000, H_en0 = 0, H_st0 = 0;
		unsigned bk = 0xffffffffu;
		for (int t = st0; t <= en0; ++t) {
			const uint32_t a = W0[t * 64], b = W1[t * 64];
			const int ut = s8(a), vo = s8(a >> 8), xo = s8(a >> 16), yo = s8(a >> 24), x2o = s8(b), y2o = s8(b >> 8), ho = (int)(int16_t)(b >> 16);
			const int xt1 = x1, vt1 = v1, x2t1 = x21, hl = h1;
			x1 = xo, v1 = vo, x21 = x2o, h1 = ho;                               // old values of this slot feed the next column
			const int tc = (int)(tp >> (4 * t)) & 15, qc = (int)(qp >> (4 * (r - t))) & 15;
			int sc = tc == qc ? P.sc_mch : P.sc_mis;
			sc = (tc == P.m1 || qc == P.m1) ? P.sc_N : sc;
			int za = xt1 + vt1, zb = yo + ut, za2 = x2t1 + vt1, zb2 = y2o + ut;
			int z = max(max(sc, za), zb);
			z = max(max(z, za2), zb2);
			int d = 4;
			d = za2 == z ? 3 : d;
			d = zb == z ? 2 : d;
			d = za == z ? 1 : d;
			d = sc == z ? 0 : d;
			z = min(z, P.sc_mch);
			const int un = z - vt1, vn = z - ut;
			const int zq = z - P.q, zq2 = z - P.q2;
			za -= zq, zb -= zq, za2 -= zq2, zb2 -= zq2;
			d |= (za > 0 ? 0x08 : 0) | (zb > 0 ? 0x10 : 0) | (za2 > 0 ? 0x20 : 0) | (zb2 > 0 ? 0x40 : 0);
			const int hn = (t == en0 && en0 > 0) ? hl + un : ho + vn;             // exact H (:316-351)
			W0[t * 64] = pack0(un, vn, max(za, 0) - qe8, max(zb, 0) - qe8);
			W1[t * 64] = pack1(max(za2, 0) - qe28, max(zb2, 0) - qe28, hn);
			if (with_cigar) PD[(r * 16 + t) * 64] = (uint8_t)d;
			if (t == en0) H_en0 = hn;
			if (t == st0) H_st0 = hn;
			const unsigned rank = t == en0 ? 0u : (t < en1 ? 1u + (unsigned)((t - st0) & 3) * 4096u + (unsigned)(t - st0)
			                                                : 1u + 4u * 4096u + (unsigned)(t - st0));
			if (hn > bh || (hn == bh && rank < bk)) bh = hn, bk = rank;          // the reference's order among equal cells (:322-349)
		}
		const int max_t = bk == 0 ? en0 : st0 + (int)((bk - 1u) & 4095u);
		if (en0 == tlen - 1 && H_en0 > ez.mte) ez.mte = H_en0, ez.mte_q = r - (((en0 + 16) & ~15) - 1);
		if (r - st0 == qlen - 1 && H_st0 > ez.mqe) ez.mqe = H_st0, ez.mqe_t = st0;
		if (ez.apply_zdrop(bh, r, max_t, P.zdrop, P.e2)) break;
		if (r == n_rows - 1 && en0 == tlen - 1) ez.score = H_en0;
	}
	int n_cigar = 0;
	if (with_cigar) {
		int i0 = -1, j0 = -1;
		if (!ez.zdropped && !(P.flag & PSVR_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
		else if (!ez.zdropped && (P.flag & PSVR_EZ_EXTZ_ONLY) && ez.mqe + P.end_bonus > ez.max) ez.reach_end = 1, i0 = ez.mqe_t, j0 = qlen - 1;
		else if (ez.max_t >= 0 && ez.max_q >= 0) i0 = ez.max_t, j0 = ez.max_q;
		if (i0 >= 0 && j0 >= 0) {
			const int w = P.w < 0 ? (tlen > qlen ? tlen : qlen) : P.w;
			// at most qlen + tlen <= 32 ops: staged in the 32 state words, which are dead now
			n_cigar = traceback(i0, j0, qlen, tlen, w,
				[&](int r, int k) { return (int)PD[(r * 16 + k) * 64]; },
				[&](int k, uint32_t word) { (k < 16 ? W0[k * 64] : W1[(k - 16) * 64]) = word; });
			uint32_t *dst = B.cigar + out->cigar_off;
			const bool rev = (P.flag & PSVR_EZ_REV_CIGAR) != 0;
			for (int m = 0; m < n_cigar; ++m) {
				const int k = rev ? m : n_cigar - 1 - m;
				dst[m] = k < 16 ? W0[k * 64] : W1[(k - 16) * 64];
			}
		}
	}
	write_ez(out, ez, n_cigar);
}

// ------------------------------------------------------------------------------------------
// lean regime, any size: a TEAM of lanes per alignment, a group of target columns per lane in registers
// ------------------------------------------------------------------------------------------
// A wavefront per alignment keeps ~40 of 64 lanes busy and pays ~200 instructions of per-diagonal bookkeeping.  Here a
// wavefront carries 64 / LANES alignments (32); each is swept in strips of LANES x CPL (16) target columns, every lane of the
// team holding the state of CPL (8) columns (u,v,x,y,x2,y2,H) in registers with static indices.  A step of the sweep is ONE
// ROW of the lane's columns, left to right: a cell needs the cell to its left in the same row (just computed, or the
// neighbour lane's last column through one DPP row_shr:1 per value: that lane is one row ahead) and the cell above (the
// column's own registers) -- the same data flow as the reference's anti-diagonal sweep, cell for cell, in another order.  A
// strip of a q-row query takes q + LANES - 1 steps; sweeping anti-diagonals inside the strip took q + 15 (2.44 -> 1.89 ms,
// profiles/r03e).  What crosses a strip boundary -- v,x,x2 of the strip's last column in every row -- goes through a
// per-alignment array E[row] in scratch memory, ping-ponged between strips and read one step ahead.  What the z-drop /
// end-score rules need per ANTI-DIAGONAL -- the maximum with the reference's tie order, H at the band ends -- is collected
// on the way: a diagonal passes a lane's columns from right to left, one column per row, so its running maximum rides in
// a register that moves one column left per step (M[], with the diagonal's tail threshold in K[]), hops to the left
// neighbour through DPP and is folded into D[r] by the team's first lane; the int16 halves of D23[r] (H in the top row / last column /
// last row) are stored by the lane that owns the cell.  extd2_team_finish_kernel evaluates the rules in anti-diagonal order
// afterwards; diagonals past a z-drop are computed but never looked at.  Direction bytes: CPL per lane and step.  Scratch
// is laid out [index][lane or team] so a wavefront touches consecutive addresses; a wavefront's slice is sized by the
// longest query of its class (the planners bin on the strip count and order by query length).  The kernel is bound by
// vector issue: see the price list in profiles/r01j_valu_op_rates.txt (add/sub/logic/right shift ~2 cycles,
// max/min/cmp/three-operand/DPP ~4) -- which is why the differences are kept x 8 with the candidate's priority in the low
// bits (below): ~47 vector instructions per cell.
#ifndef PSVR_TEAM_WAVES
#define PSVR_TEAM_WAVES 3          /* wavefronts per SIMD the register allocation aims at: 2 lanes x 8 columns needs 168 VGPRs (3 per SIMD); 1 x 16 at two per
                                      SIMD runs as fast (1.84 vs 1.88 ms), 2 x 8 at two 2.29 ms, 4 x 4 at five 2.17 ms, 4 x 8 at three 2.54 ms: profiles/r03e */
#endif
// LEAN = 1: no per-anti-diagonal maximum (M[], K[], G[], the D[r] records).  That maximum feeds ksw_apply_zdrop and ez.max / max_q / max_t
// only.  With e2 == 0 a gap of ANY length costs at most q2, so every anti-diagonal behind the cell that holds the running maximum has a cell
// reachable from it by one insertion and one deletion: its maximum is at least max - 2 q2, and with zdrop >= 2 q2 (the reference's defaults:
// 32 / 0, zdrop 400) the rule `max - H > zdrop + l * e2` can never hold -- dp_zdrop_inert().  The engine's own launches (seam B1 reads
// score, mqe and the CIGAR of its pieces) then run this variant; psvr_extd2_batch, whose ksw_extz_t carries max / max_q / max_t, never does.
template <int LANES, int CPL, int LEAN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PSVR_TEAM_WAVES, 8))) void extd2_team_kernel(DpBatch B, DpParams P, TeamPlan T)
{
	static_assert(CPL == 4 || CPL == 8 || CPL == 16, "columns per lane: direction bytes go out as dwords");
	constexpr int SW = CPL * LANES, PB = 64 / LANES;                 // strip width, alignments per wavefront
	constexpr int kNone = (int)0x80000000;                           // "no cell yet" in the per-diagonal maximum
	const int lane = threadIdx.x, team = lane / LANES, ql = lane % LANES;
	// substitution scores of every (target, query) code pair, scaled and tagged like the other candidates (see below), written by
	// all 64 lanes before any of them leaves
	__shared__ int sc_lut[64];
	{
		const int tcode = lane >> 3, qcode = lane & 7;
		sc_lut[lane] = 8 * ((tcode == P.m1 || qcode == P.m1) ? P.sc_N : (tcode == qcode ? P.sc_mch : P.sc_mis)) + 4;
	}
	__builtin_amdgcn_wave_barrier();
	int cls = 0;
	while (cls + 1 < T.n_classes && (int)blockIdx.x >= T.first_block[cls + 1]) ++cls;
	const int n_strips16 = T.n_strips16[cls];
	const long long slot = (long long)((int)blockIdx.x - T.first_block[cls]) * PB + team;
	const bool live = slot < T.count[cls];
	int pid = 0, qlen = 0, tlen = 0;
	if (live) pid = B.idx[T.first_slot[cls] + slot], qlen = B.qlen[pid], tlen = B.tlen[pid];
	const int n_strips = (n_strips16 * 16 + SW - 1) / SW;
	const int qmax = wave_max_i32(qlen > 0 ? qlen : 0);
	const int ksteps = uni(qmax + LANES - 1);                       // steps of a strip: lane ql does row k - ql in step k
	const int k_last = uni(-wave_max_i32(live && qlen > 0 ? -qlen : (int)0x80000000)) - 1;   // the shortest live query's last row, in lane 0's count
	const int R = ksteps, NR = qmax + SW * n_strips + 1;
	const unsigned long long need = dp_team_ws_bytes(qmax, n_strips16, LANES, CPL);
	const unsigned long long base = T.ws_base[cls] + (unsigned long long)((int)blockIdx.x - T.first_block[cls]) * T.ws_need[cls];   // need <= ws_need: the class's longest query bounds this wavefront's
	const bool bad_shape = qlen <= 0 || tlen <= 0 || (tlen + SW - 1) / SW > n_strips;
	if (!live) return;
	if (base + need > B.ws_cap && B.err) *B.err = 20;
	if (P.skip || bad_shape || base + need > B.ws_cap) return;      // (extd2_team_finish_kernel writes the empty record)
	uint8_t *w0 = B.ws + base;
	const size_t offE = (size_t)(64 * CPL) * n_strips * R;
	// Every scratch address is a wave-uniform base (scalar registers) plus a constant 32-bit lane offset, so a step spends no
	// vector instruction on addresses: direction dwords of strip s, step k at [(s * R + k) * 64 + lane]; strip-boundary values
	// (8 bytes) and the per-diagonal records D, D2, D3 (4 bytes) of diagonal r at [r * PB + team].
	// (a boundary record is one dword: v, x, x2 are multiples of 8 plus a constant tag and fit int8 once divided, see the step)
	uint8_t *const uE0 = w0 + offE, *const uE1 = w0 + offE + (size_t)4 * PB * NR;
	// per anti-diagonal r: the maximum's key D[r] (a dword the first lane reads, folds and writes back) and, in one dword of a second array,
	// H at the band's end (low half) and in the last row (high half) as int16, each stored once by the lane that owns the cell
	uint8_t *const uD = w0 + offE + (size_t)8 * PB * NR, *const uD23 = uD + (size_t)4 * PB * NR;
	const unsigned lc = (unsigned)CPL * (unsigned)lane, t4 = 4u * (unsigned)team;
	auto at4 = [=](uint8_t *ub, int r) -> int & { return *(int *)(ub + (size_t)r * (4 * PB) + t4); };
	auto atD2 = [=](int r) -> short & { return *(short *)(uD23 + (size_t)r * (4 * PB) + t4); };
	auto atD3 = [=](int r) -> short & { return *(short *)(uD23 + (size_t)r * (4 * PB) + t4 + 2); };
	const uint8_t *query = B.qseq + B.q_off[pid], *target = B.tseq + B.t_off[pid];
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	// All difference values are kept times 8, and the five candidates of a cell carry their priority in the low three bits
	// (sc 4, a 3, b 2, a2 1, b2 0: the first of {sc,a,b,a2,b2} that reaches the maximum wins, :176-213), so one max3 pair
	// yields both z and the direction; x / y / x2 / y2 live with their tag added so that a candidate is a single add.
	const int neg_qe = s8(-P.q - P.e), neg_qe2 = s8(-P.q2 - P.e2);
	const int x_init = 8 * neg_qe + 3, y_init = 8 * neg_qe + 2, x2_init = 8 * neg_qe2 + 1, y2_init = 8 * neg_qe2;
	const int z_cap = 8 * P.sc_mch + 7;                               // (the substitution scores, 8 * sc + 4, sit in sc_lut)
	const int q_m8 = 8 * P.q - 8, q2_m8 = 8 * P.q2 - 8;               // a - (z - q) > 0  <=>  tagged difference - 8 >= 0
	const int cx = 8 - 8 * s8(P.q + P.e), cx2 = 8 - 8 * s8(P.q2 + P.e2);
	const int long_thres = P.long_thres, ur_short = s8(-P.e), ur_at = s8(P.long_diff), ur_long = s8(-P.e2);
	auto ur_of = [=](int r) { return r == 0 ? neg_qe : r < long_thres ? ur_short : r == long_thres ? ur_at : ur_long; };
	// H is tracked by vertical steps only, H(i,t) = H(i-1,t) + v(i,t).  The reference takes a horizontal step for the last
	// in-band cell of a diagonal (:322), which is the same number: both deltas come from one z.  For the top row that needs the
	// value "above" it: -qe for column 0 (the reference's H[0] = v - qe at r == 0, :351), then the boundary u of every column
	// added up, since H(0,t) = H(0,t-1) + u(0,t) and v(0,t) - u(0,t) + v(0,t-1) = ur(t).
	auto h_above = [&](int t) {
		int h = -P.qe_pre;
		const int lt = P.long_thres;
		if (t >= 1) {
			if (lt >= 1) h += t < lt ? t * s8(-P.e) : (lt - 1) * s8(-P.e) + s8(P.long_diff) + (t - lt) * s8(-P.e2);
			else h += t * s8(-P.e2);
		}
		return h;
	};
	auto frank = [](int p) { return (((p << 12) + p) & 0x3fff) ^ 0x3fff; };   // (place & 3) << 12 | place, inverted: larger = earlier in the reference's order
	const int ns = (tlen + SW - 1) / SW;
	for (int sv = 0; sv < ns; ++sv) {
		const int s = uni(sv);                                          // the wavefront's teams sweep their strips in step: scalar registers for what depends on s and k only
		const int c0 = SW * s;
		const int ncols = tlen - c0 < SW ? tlen - c0 : SW;
		const int jb = CPL * ql, cb = c0 + jb;                          // this lane's first column: in the strip, in the target
		const int jx = tlen - 1 - cb;                                   // the last target column, if this lane holds it
		const bool own = (unsigned)jx < (unsigned)CPL;
		int U[CPL], V[CPL], X[CPL], Y[CPL], X2[CPL], Y2[CPL], H[CPL], TC[CPL];
		int M[CPL], K[CPL], G[CPL];                                   // per anti-diagonal passing the column: maximum so far, tail threshold; per column: rank bits of place = t
		bool inr[CPL];                                                // column inside the target
#pragma unroll
		for (int jj = 0; jj < CPL; ++jj) {
			U[jj] = 8 * ur_of(cb + jj);                                 // u/y/y2 of the first cell of a column (:153-156)
			V[jj] = 8 * neg_qe, X[jj] = x_init, Y[jj] = y_init, X2[jj] = x2_init, Y2[jj] = y2_init;
			H[jj] = 8 * h_above(cb + jj);
			inr[jj] = jb + jj < ncols;
			TC[jj] = inr[jj] ? (target[cb + jj] & 7) * 32 : 0;          // row of the score table (codes are 0..4)
			M[jj] = kNone, G[jj] = frank(cb + jj);
			const int d = cb + jj - ql;                                 // the anti-diagonal of column jj in this lane's row of step 0
			const int st = max(0, d - qlen + 1), en = min(tlen - 1, d);
			K[jj] = st + ((en - st) & ~3) - cb;
		}
		(void)M, (void)K, (void)G;
		uint8_t *const Ein = (s & 1) ? uE0 : uE1, *const Eout = (s & 1) ? uE1 : uE0;
		// what a step reads from memory (query base, the previous strip's boundary values, the diagonal's running maximum) is
		// loaded one step ahead, so the loads have a whole step to arrive
		unsigned e_prev = 0;
		unsigned dirw[CPL / 4];                                       // the direction bytes of the step before
		int d_cur = kNone, out_prev = kNone;
		if (s > 0) {
			e_prev = (unsigned)at4(Ein, 0);
			if (!LEAN && 0 <= ksteps - LANES) d_cur = at4(uD, c0);
		}
		unsigned q_cur = ql == 0 ? query[0] : 0u;                       // query[k - ql] for k = 0 (raw byte: masking it here would wait for the load)
		for (int kv = 0; kv < LANES - 1; ++kv) {
			const int k = uni(kv);
#define TEAM_MASKED 1
#define TEAM_TOPROW 1
#include "ksw_row_step.inc"
#undef TEAM_MASKED
#undef TEAM_TOPROW
		}
		{
			const int k = LANES - 1;                                    // the last lane's row 0
#define TEAM_MASKED 0
#define TEAM_TOPROW 1
#include "ksw_row_step.inc"
#undef TEAM_MASKED
#undef TEAM_TOPROW
		}
		for (int kv = LANES; kv < ksteps; ++kv) {
			const int k = uni(kv);
#define TEAM_MASKED 0
#define TEAM_TOPROW 0
#include "ksw_row_step.inc"
#undef TEAM_MASKED
#undef TEAM_TOPROW
		}
		if (with_cigar) {
			uint32_t *const dst = (uint32_t *)(w0 + (size_t)(s * R + ksteps - 1) * (64 * CPL) + lc);
#pragma unroll
			for (int g = 0; g < CPL / 4; ++g) dst[g] = dirw[g];
		}
		// what is still on its way through the lanes: anti-diagonals c0 + ksteps and later, none of which an earlier strip wrote
		if (LEAN) { __threadfence_block(); continue; }
		if (LANES > 1) {
			const int inc = __builtin_amdgcn_update_dpp(kNone, out_prev, 0x101, 0xf, 0xf, false);
			const int m0 = __builtin_amdgcn_update_dpp(kNone, M[0], 0x101, 0xf, 0xf, false);   // the neighbour's first is this lane's last
			M[CPL - 2] = max(M[CPL - 2], ql == LANES - 1 ? kNone : inc);
			M[CPL - 1] = max(M[CPL - 1], ql == LANES - 1 ? kNone : m0);
		}
#pragma unroll
		for (int jj = 0; jj < CPL; ++jj)
			if (jj > 0 || ql == 0) at4(uD, ksteps - ql + cb + jj) = M[jj];
		__threadfence_block();                                          // the next strip's first lane reads what any lane stored here
	}
}

// The per-diagonal rules and the traceback of the alignments extd2_team_kernel swept: one THREAD per alignment.  Both are chains of
// dependent loads (a record per anti-diagonal, a direction byte per CIGAR step); inside the sweep kernel, one lane per team at three
// wavefronts per SIMD, they took 0.39 of 2.21 ms on the DP micro-benchmark (profiles/r03e).  Thread t of block b serves team t % PB of the sweep's
// block (b * 64 + t) / PB and finds that wavefront's scratch the way it did.
#ifndef PSVR_FINISH_WAVES
#define PSVR_FINISH_WAVES 8          // most wavefronts of the finish launch per SIMD (experiments)
#endif
template <int LANES, int CPL, int LEAN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, PSVR_FINISH_WAVES))) void extd2_team_finish_kernel(DpBatch B, DpParams P, TeamPlan T)
{
	constexpr int SW = CPL * LANES, PB = 64 / LANES, kWinRows = 8;
	__shared__ uint32_t win[kWinRows * (CPL / 4) * 64];              // the traceback's window of direction bytes: [row][dword][thread]
	const int lane = threadIdx.x, team = lane % PB;
	const int tb = (int)((blockIdx.x * 64u + (unsigned)lane) / (unsigned)PB);   // the sweep's block
	const bool in_grid = tb < T.first_block[T.n_classes];
	int cls = 0;
	while (cls + 1 < T.n_classes && tb >= T.first_block[cls + 1]) ++cls;
	const int n_strips16 = T.n_strips16[cls];
	const long long slot = (long long)(tb - T.first_block[cls]) * PB + team;
	const bool live = in_grid && slot < T.count[cls];
	int pid = 0, qlen = 0, tlen = 0;
	if (live) pid = B.idx[T.first_slot[cls] + slot], qlen = B.qlen[pid], tlen = B.tlen[pid];
	const int n_strips = (n_strips16 * 16 + SW - 1) / SW;
	int qmax = qlen > 0 ? qlen : 0;                                  // the longest query of the sweep's wavefront: over the PB threads that serve it
#pragma unroll
	for (int d = 1; d < PB; d <<= 1) qmax = max(qmax, __shfl_xor(qmax, d, 64));
	const int ksteps = qmax + LANES - 1;
	const int R = ksteps, NR = qmax + SW * n_strips + 1;
	const unsigned long long need = dp_team_ws_bytes(qmax, n_strips16, LANES, CPL);
	const unsigned long long base = T.ws_base[cls] + (unsigned long long)(tb - T.first_block[cls]) * T.ws_need[cls];
	if (!live) return;
	psvr_extz_t *out = B.ez + pid;
	EzAcc ez;
	ez.reset();
	const bool bad_shape = qlen <= 0 || tlen <= 0 || (tlen + SW - 1) / SW > n_strips;
	if (P.skip || bad_shape || base + need > B.ws_cap) { write_ez(out, ez, 0); return; }
	uint8_t *w0 = B.ws + base;
	const size_t offE = (size_t)(64 * CPL) * n_strips * R;
	uint8_t *const uD = w0 + offE + (size_t)8 * PB * NR, *const uD23 = uD + (size_t)4 * PB * NR;   // key; H at the band's end | in the last row << 16
	const unsigned t4 = 4u * (unsigned)team;
	auto at4 = [=](uint8_t *ub, int r) -> int & { return *(int *)(ub + (size_t)r * (4 * PB) + t4); };
	const int with_cigar = !(P.flag & PSVR_EZ_SCORE_ONLY);
	const int n_rows = qlen + tlen - 1;
	// the per-diagonal rules, in anti-diagonal order (ksw2_extd2_sse.c:316-351, ksw_apply_zdrop)
	// (the records are fetched eight diagonals at a time: one lane per alignment walks them, and a load per iteration would cost a
	// memory round trip each)
	bool stop = false;
	for (int r0 = 0; r0 < n_rows && !stop; r0 += 8) {
		int kd[8], h2[8], h3[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int r = min(r0 + u, n_rows - 1);
			const int w23 = at4(uD23, r);
			kd[u] = LEAN ? 0 : at4(uD, r), h2[u] = (int)(short)(w23 & 0xffff), h3[u] = w23 >> 16;   // the high half holds a value only where the last query row meets the diagonal
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			const int r = r0 + u;
			if (r >= n_rows || stop) continue;
			const int st0 = max(0, r - qlen + 1), en0 = min(tlen - 1, r);
			const int key = kd[u];
			const int max_H = key >> 16;
			const int H_en0 = h2[u];
			const int max_t = H_en0 == max_H ? en0 : st0 + ((0x7fff - (key & 0xffff)) & 4095);
			if (en0 == tlen - 1 && H_en0 > ez.mte) ez.mte = H_en0, ez.mte_q = r - (((en0 + 16) & ~15) - 1);
			if (r - st0 == qlen - 1 && h3[u] > ez.mqe) ez.mqe = h3[u], ez.mqe_t = st0;
			if (!LEAN && ez.apply_zdrop(max_H, r, max_t, P.zdrop, P.e2)) stop = true;   // (LEAN: the rule cannot hold, see the sweep; ez.max / max_q / max_t stay as reset)
			else if (r == n_rows - 1) ez.score = H_en0;                      // en0 == tlen - 1 on the last diagonal
		}
	}
	int n_cigar = 0;
	if (with_cigar) {
		int i0 = -1, j0 = -1;
		if (!ez.zdropped && !(P.flag & PSVR_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
		else if (!ez.zdropped && (P.flag & PSVR_EZ_EXTZ_ONLY) && ez.mqe + P.end_bonus > ez.max) ez.reach_end = 1, i0 = ez.mqe_t, j0 = qlen - 1;
		else if (ez.max_t >= 0 && ez.max_q >= 0) i0 = ez.max_t, j0 = ez.max_q;
		if (i0 >= 0 && j0 >= 0) {
			const int w = P.w < 0 ? (tlen > qlen ? tlen : qlen) : P.w;
			const uint8_t *pb = w0 + team * SW;                      // the team's SW direction bytes of a step are contiguous
			uint32_t *stage = (uint32_t *)(w0 + offE) + team;         // <= qlen + tlen ops; the strip-boundary arrays are dead now
			// A byte per CIGAR step, each address known only when the byte before it has arrived, is a chain of memory round trips.  The
			// path moves up and to the left by at most one cell a step, so a window of kWinRows rows of the column group it is in (one lane's
			// CPL bytes of kWinRows consecutive steps: independent loads, in flight together) serves about eight steps; it is kept in LDS.
			int win_row = -1, win_grp = -1;                          // the window's bottom row and column group (t / CPL)
			n_cigar = traceback(i0, j0, qlen, tlen, w,
				[&](int r, int k) {                                     // k counts from the 16-rounded band start of row r, as in the reference
					const int t = k + (max(0, r - qlen + 1) & ~15), row = r - t, grp = t / CPL;
					if (grp != win_grp || row > win_row || row <= win_row - kWinRows) {
						const int s = grp / LANES, gl = grp % LANES;          // strip, lane of the team: it did row `row` in step row + gl
						const uint8_t *src = pb + (size_t)(s * R + row + gl) * (64 * CPL) + gl * CPL;
						uint32_t tmp[kWinRows][CPL / 4];
#pragma unroll
						for (int d = 0; d < kWinRows; ++d)
#pragma unroll
							for (int u = 0; u < CPL / 4; ++u) tmp[d][u] = d <= row ? ((const uint32_t *)(src - (size_t)d * (64 * CPL)))[u] : 0u;
#pragma unroll
						for (int d = 0; d < kWinRows; ++d)
#pragma unroll
							for (int u = 0; u < CPL / 4; ++u) win[(d * (CPL / 4) + u) * 64 + lane] = tmp[d][u];
						win_row = row, win_grp = grp;
					}
					const int x = t % CPL;
					const unsigned b = (win[((win_row - row) * (CPL / 4) + (x >> 2)) * 64 + lane] >> (8 * (x & 3))) & 0xffu, n = ~b;
					// back to the reference's byte: direction in bits 0-2, "gap extended" for a, b, a2, b2 in bits 3-6
					return (int)((4u - ((b >> 4) & 7u)) | ((n >> 3) & 1u) << 3 | ((n >> 2) & 1u) << 4 | ((n >> 1) & 1u) << 5 | (n & 1u) << 6);
				},
				[&](int k, uint32_t word) { stage[(size_t)k * PB] = word; });
			uint32_t *dst = B.cigar + out->cigar_off;
			const bool rev = (P.flag & PSVR_EZ_REV_CIGAR) != 0;
			for (int m = 0; m < n_cigar; ++m) dst[m] = stage[(size_t)(rev ? m : n_cigar - 1 - m) * PB];
		}
	}
	write_ez(out, ez, n_cigar);
}
template __global__ void extd2_team_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 0>(DpBatch, DpParams, TeamPlan);
template __global__ void extd2_team_finish_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 0>(DpBatch, DpParams, TeamPlan);
template __global__ void extd2_team_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 1>(DpBatch, DpParams, TeamPlan);
template __global__ void extd2_team_finish_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 1>(DpBatch, DpParams, TeamPlan);

// ------------------------------------------------------------------------------------------
// general path: DP state in LDS laid out exactly like the reference's flat image
//   extd2: u|v|x|y|x2|y2|s|sf|qr   (ksw2_extd2_sse.c:100-103)
//   extz2: u|v|x|y|s|sf|qr         (ksw2_extz2_sse.c:85-87)
// so that the unaligned 16-byte score loads/stores that run past an array behave identically.
// Direction bytes go to a global slab (they can exceed LDS for long sequences).
// VAR = 0: dual affine (extd2);  VAR = 1: single affine (extz2, SSE2 code path).
// ------------------------------------------------------------------------------------------
template <int VAR>
__global__ __launch_bounds__(64) void extd2_lds_kernel(DpBatch B, DpParams P)
{
	extern __shared__ __align__(16) uint8_t lds[];
	const int pid = B.idx[blockIdx.x];
	const int lane = threadIdx.x;
	const int qlen = B.qlen[pid], tlen = B.tlen[pid];
	psvr_extz_t *out = B.ez + pid;
	EzAcc ez;
	ez.reset();
	if (P.skip || qlen <= 0 || tlen <= 0) {
		if (lane == 0) write_ez(out, ez, 0);
		return;
	}
	const uint8_t *query = B.qseq + B.q_off[pid], *target = B.tseq + B.t_off[pid];
	const int w = P.w < 0 ? (tlen > qlen ? tlen : qlen) : P.w;
	int n_col = qlen < tlen ? qlen : tlen;
	n_col = ((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1;
	const int rowb = n_col * 16;
	const int n_rows = qlen + tlen - 1;
	const int T = ((tlen + 15) / 16) * 16, QL = ((qlen + 15) / 16) * 16;
	constexpr int NARR = VAR == 0 ? 7 : 5;
	int8_t *u8 = (int8_t*)lds, *v8 = u8 + T, *x8 = v8 + T, *y8 = x8 + T;
	int8_t *x28 = VAR == 0 ? y8 + T : nullptr, *y28 = VAR == 0 ? x28 + T : nullptr;
	int8_t *sa = VAR == 0 ? y28 + T : y8 + T;
	uint8_t *sf = (uint8_t*)(sa + T), *qr = sf + T;
	const int img = NARR * T + T + QL + 16;            // bytes of the byte image (+16 calloc tail)
	int32_t *H = (int32_t*)(lds + ((img + 15) & ~15));
	uint8_t *Pm = B.pslab + (B.p_off[pid] << B.p_unit_shift);
	const int flag = P.flag;
	const int with_cigar = !(flag & PSVR_EZ_SCORE_ONLY), approx_max = !!(flag & PSVR_EZ_APPROX_MAX);
	const int right = with_cigar && (flag & PSVR_EZ_RIGHT);

	const int neg_qe = s8(-P.q - P.e), neg_qe2 = s8(-P.q2 - P.e2);
	const int qe8 = s8(P.q + P.e), qe28 = s8(P.q2 + P.e2);
	const int qe = P.q + P.e;                       // extz2: no swap, qe used throughout
	for (int i = lane; i < img; i += 64) {
		int a = i / T;                               // which array
		uint8_t val = 0;
		if (VAR == 0) { if (a < 4) val = (uint8_t)neg_qe; else if (a < 6) val = (uint8_t)neg_qe2; }
		if (a == NARR) { int t = i - NARR * T; val = t < tlen ? target[t] : 0; }
		if (a > NARR)  { int k = i - (NARR + 1) * T; val = k < qlen ? query[qlen - 1 - k] : 0; }
		lds[i] = val;
	}
	for (int i = lane; i < T; i += 64) H[i] = PSVR_KSW_NEG_INF;
	__syncthreads();

	int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
	const uint8_t qe2b = (uint8_t)((P.q + P.e) * 2), max_scb = (uint8_t)(P.mat[0] + (P.q + P.e) * 2), qb8 = (uint8_t)P.q;
	for (int r = 0; r < n_rows; ++r) {
		int st0, en0, st, en;
		if (!band_limits(r, qlen, tlen, w, st0, en0, st, en)) { ez.zdropped = 1; break; }
		int x1, x21 = 0, v1;
		if (VAR == 0) {
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) x1 = x8[st - 1], x21 = x28[st - 1], v1 = v8[st - 1];
				else x1 = neg_qe, x21 = neg_qe2, v1 = neg_qe;
			} else {
				x1 = neg_qe, x21 = neg_qe2;
				v1 = r == 0 ? neg_qe : r < P.long_thres ? s8(-P.e) : r == P.long_thres ? s8(P.long_diff) : s8(-P.e2);
			}
		} else {
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) x1 = x8[st - 1], v1 = v8[st - 1];   // int8_t in the reference (:104,121)
				else x1 = v1 = 0;
			} else x1 = 0, v1 = r ? s8(P.q) : 0;
		}
		__syncthreads();
		if (en >= r && lane == 0) {
			if (VAR == 0) {
				y8[r] = neg_qe, y28[r] = neg_qe2;
				u8[r] = r == 0 ? neg_qe : r < P.long_thres ? s8(-P.e) : r == P.long_thres ? s8(P.long_diff) : s8(-P.e2);
			} else y8[r] = 0, u8[r] = r ? (int8_t)qb8 : 0;
		}
		// scores
		const uint8_t *qrr = qr + (qlen - 1 - r);
		if (!(flag & PSVR_EZ_GENERIC_SC)) {
			const int fresh_end = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;
			for (int t0 = st0; t0 <= fresh_end; t0 += 64) {
				int t = t0 + lane;
				if (t <= fresh_end) {
					uint8_t sq = sf[t], sq2 = qrr[t];
					int sc = sq == sq2 ? P.sc_mch : P.sc_mis;
					if (sq == (uint8_t)P.m1 || sq2 == (uint8_t)P.m1) sc = P.sc_N;
					sa[t] = (int8_t)sc;
				}
			}
		} else {
			for (int t = st0 + lane; t <= en0; t += 64) sa[t] = P.mat[sf[t] * P.m + qrr[t]];
		}
		__syncthreads();
		// core: 64-lane groups from high t to low t
		const int ngrp = (en - st) / 64 + 1;
		for (int g = ngrp - 1; g >= 0; --g) {
			const int t = st + g * 64 + lane;
			if (t <= en) {
				int d = 0;
				if (VAR == 0) {
					int xt1 = t == st ? x1 : (int)x8[t - 1], vt1 = t == st ? v1 : (int)v8[t - 1], x2t1 = t == st ? x21 : (int)x28[t - 1];
					int z = sa[t], ut = u8[t];
					int a = s8(xt1 + vt1), b = s8(y8[t] + ut), a2 = s8(x2t1 + vt1), b2 = s8(y28[t] + ut);
					if (!right) {
						if (a > z)  d = 1, z = a;
						if (b > z)  d = 2, z = b;
						if (a2 > z) d = 3, z = a2;
						if (b2 > z) d = 4, z = b2;
					} else {
						d = z > a ? 0 : 1;  z = z > a ? z : a;
						d = z > b ? d : 2;  z = z > b ? z : b;
						d = z > a2 ? d : 3; z = z > a2 ? z : a2;
						d = z > b2 ? d : 4; z = z > b2 ? z : b2;
					}
					z = min(z, P.sc_mch);
					u8[t] = (int8_t)(z - vt1), v8[t] = (int8_t)(z - ut);
					int tmp = s8(z - P.q);
					a = s8(a - tmp), b = s8(b - tmp);
					tmp = s8(z - P.q2);
					a2 = s8(a2 - tmp), b2 = s8(b2 - tmp);
					if (!right) {
						x8[t]  = (int8_t)(max(a, 0) - qe8);   d |= a  > 0 ? 0x08 : 0;
						y8[t]  = (int8_t)(max(b, 0) - qe8);   d |= b  > 0 ? 0x10 : 0;
						x28[t] = (int8_t)(max(a2, 0) - qe28); d |= a2 > 0 ? 0x20 : 0;
						y28[t] = (int8_t)(max(b2, 0) - qe28); d |= b2 > 0 ? 0x40 : 0;
					} else {
						x8[t]  = (int8_t)(max(a, 0) - qe8);   d |= a  >= 0 ? 0x08 : 0;
						y8[t]  = (int8_t)(max(b, 0) - qe8);   d |= b  >= 0 ? 0x10 : 0;
						x28[t] = (int8_t)(max(a2, 0) - qe28); d |= a2 >= 0 ? 0x20 : 0;
						y28[t] = (int8_t)(max(b2, 0) - qe28); d |= b2 >= 0 ? 0x40 : 0;
					}
				} else {
					const int k = t - st;
					unsigned x1w = (unsigned)x1, v1w = (unsigned)v1;   // sign-extended into lanes 1..3 (:147-148)
					uint8_t xt1 = k == 0 ? 0 : (uint8_t)x8[t - 1], vt1 = k == 0 ? 0 : (uint8_t)v8[t - 1];
					if (k < 4) xt1 |= (uint8_t)(x1w >> (8 * k)), vt1 |= (uint8_t)(v1w >> (8 * k));
					uint8_t z = (uint8_t)((uint8_t)sa[t] + qe2b), a = (uint8_t)(xt1 + vt1), ut = (uint8_t)u8[t], b = (uint8_t)((uint8_t)y8[t] + ut);
					if (!with_cigar) {
						z = (int8_t)z > 0 ? z : 0; z = z > a ? z : a;
					} else if (!right) {
						d = (int8_t)a > (int8_t)z ? 1 : 0;
						z = (int8_t)z > 0 ? z : 0; z = z > a ? z : a;
						if ((int8_t)b > (int8_t)z) d = 2;
					} else {
						d = (int8_t)z > (int8_t)a ? 0 : 1;
						z = (int8_t)z > 0 ? z : 0; z = z > a ? z : a;
						if (!((int8_t)z > (int8_t)b)) d = 2;
					}
					z = z > b ? z : b;
					z = z < max_scb ? z : max_scb;
					u8[t] = (int8_t)(uint8_t)(z - vt1), v8[t] = (int8_t)(uint8_t)(z - ut);
					z = (uint8_t)(z - qb8);
					a = (uint8_t)(a - z), b = (uint8_t)(b - z);
					if (!right) {
						x8[t] = (int8_t)a > 0 ? (int8_t)a : 0; d |= (int8_t)a > 0 ? 0x08 : 0;
						y8[t] = (int8_t)b > 0 ? (int8_t)b : 0; d |= (int8_t)b > 0 ? 0x10 : 0;
					} else {
						x8[t] = (int8_t)a < 0 ? 0 : (int8_t)a; d |= (int8_t)a < 0 ? 0 : 0x08;
						y8[t] = (int8_t)b < 0 ? 0 : (int8_t)b; d |= (int8_t)b < 0 ? 0 : 0x10;
					}
				}
				if (with_cigar) Pm[(size_t)r * rowb + (t - st)] = (uint8_t)d;
			}
			__syncthreads();
		}
		const int e_drop = VAR == 0 ? P.e2 : P.e;
		if (!approx_max) {
			int max_H, max_t;
			if (r > 0) {
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				const int uen = VAR == 0 ? (int)u8[en0] : (int)(uint8_t)u8[en0] - qe;
				const int ven = VAR == 0 ? (int)v8[en0] : (int)(uint8_t)v8[en0] - qe;
				const int h_en0 = en0 > 0 ? H[en0 - 1] + uen : H[en0] + ven;
				__syncthreads();
				int bh = (int)0x80000000; unsigned bk = 0xffffffffu;
				for (int t0 = st0; t0 <= en0; t0 += 64) {
					int t = t0 + lane;
					if (t <= en0) {
						int h;
						if (t == en0) h = h_en0;
						else h = H[t] + (VAR == 0 ? (int)v8[t] : (int)(uint8_t)v8[t] - qe);
						H[t] = h;
						unsigned rank = t == en0 ? 0u : t < en1 ? 1u + (unsigned)((t - st0) & 3) * 16384u + (unsigned)(t - st0)
						                                        : 1u + 4u * 16384u + (unsigned)(t - st0);
						if (h > bh || (h == bh && rank < bk)) bh = h, bk = rank;
					}
				}
				max_H = wave_max_i32(bh);
				unsigned rk = wave_min_u32(bh == max_H ? bk : 0xffffffffu);
				max_t = rk == 0 ? en0 : st0 + (int)((rk - 1u) & 16383u);
				__syncthreads();
			} else {
				int h0 = VAR == 0 ? (int)v8[0] - P.qe_pre : (int)(uint8_t)v8[0] - qe - qe;
				__syncthreads();
				if (lane == 0) H[0] = h0;
				max_H = h0, max_t = 0;
				__syncthreads();
			}
			const int H_en0 = H[en0], H_st0 = H[st0];
			if (en0 == tlen - 1 && H_en0 > ez.mte) ez.mte = H_en0, ez.mte_q = r - en;
			if (r - st0 == qlen - 1 && H_st0 > ez.mqe) ez.mqe = H_st0, ez.mqe_t = st0;
			if (ez.apply_zdrop(max_H, r, max_t, P.zdrop, e_drop)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H[tlen - 1];
		} else {
			const int bias = VAR == 0 ? 0 : qe;
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int d0 = (VAR == 0 ? (int)v8[last_H0_t] : (int)(uint8_t)v8[last_H0_t]) - bias;
					int d1 = (VAR == 0 ? (int)u8[last_H0_t + 1] : (int)(uint8_t)u8[last_H0_t + 1]) - bias;
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += (VAR == 0 ? (int)v8[last_H0_t] : (int)(uint8_t)v8[last_H0_t]) - bias;
				} else {
					++last_H0_t, H0 += (VAR == 0 ? (int)u8[last_H0_t] : (int)(uint8_t)u8[last_H0_t]) - bias;
				}
				if (VAR == 1 && (flag & PSVR_EZ_APPROX_DROP) && ez.apply_zdrop(H0, r, last_H0_t, P.zdrop, e_drop)) break;
			} else H0 = VAR == 0 ? (int)v8[0] - P.qe_pre : (int)(uint8_t)v8[0] - qe - qe, last_H0_t = 0;
			// extd2 tests the approximate drop on every diagonal including r==0 (ksw2_extd2_sse.c:373);
			// extz2 only for r>0 (ksw2_extz2_sse.c:283)
			if (VAR == 0 && (flag & PSVR_EZ_APPROX_DROP) && ez.apply_zdrop(H0, r, last_H0_t, P.zdrop, e_drop)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
		}
		last_st = st, last_en = en;
	}
	int n_cigar = 0;
	if (with_cigar) {
		__threadfence_block();
		__syncthreads();
		int i0 = -1, j0 = -1;
		if (!ez.zdropped && !(flag & PSVR_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
		else if (!ez.zdropped && (flag & PSVR_EZ_EXTZ_ONLY) && ez.mqe + P.end_bonus > ez.max) ez.reach_end = 1, i0 = ez.mqe_t, j0 = qlen - 1;
		else if (ez.max_t >= 0 && ez.max_q >= 0) i0 = ez.max_t, j0 = ez.max_q;
		if (i0 >= 0 && j0 >= 0) {
			uint32_t *dst = B.cigar + uni64(out->cigar_off);
			n_cigar = traceback(i0, j0, qlen, tlen, w,
				[&](int r, int k) { return (int)__builtin_nontemporal_load(Pm + (size_t)r * rowb + k); },
				[&](int k, uint32_t word) { if (lane == 0) dst[k] = word; });
			if (!(flag & PSVR_EZ_REV_CIGAR)) {
				__threadfence_block();
				__syncthreads();
				for (int m = lane; m < (n_cigar >> 1); m += 64) {
					uint32_t a = __builtin_nontemporal_load(dst + m), b = __builtin_nontemporal_load(dst + n_cigar - 1 - m);
					dst[m] = b, dst[n_cigar - 1 - m] = a;
				}
			}
		}
	}
	if (lane == 0) write_ez(out, ez, n_cigar);
}

template __global__ void extd2_lds_kernel<0>(DpBatch, DpParams);
template __global__ void extd2_lds_kernel<1>(DpBatch, DpParams);

} // namespace psvr
