// ksw_device.h -- device-side pieces of the banded dual-affine anti-diagonal DP for gfx950.
//
// Replaces ksw_extd2_sse / ksw_extz2_sse (reference: src/kswlib/ksw2_extd2_sse.c:26-396,
// ksw2_extz2_sse.c:23-305) and ksw_backtrack_D / ksw_apply_zdrop (src/kswlib/ksw2.h:119-151,245-261).
//
// Design (MI355X-first, not a port of the SSE code).  Four kernels share the recurrences; the planners route a problem by shape:
//   * extd2_team_kernel<LANES, CPL> + extd2_team_finish_kernel -- the `aln` path's kernel: band never clips the matrix, values fit
//     int8 (dp_band_never_binds && nowrap_ok).  2 lanes per alignment, 32 alignments per wavefront, the matrix swept in strips of
//     16 target columns with the state of 8 columns per lane in registers, a ROW of them per step; see the kernel for the strip
//     boundary / per-diagonal bookkeeping.  The z-drop / end rules and the traceback are the second launch, a thread per alignment.
//   * extd2_tiny_kernel     -- same regime, qlen, tlen <= 16: one thread per alignment, state in LDS.
//   * extd2_reg_kernel<K,PG> -- one 64-lane wavefront per alignment, lane L of chunk c owns target column t = 64c + L, the
//     per-column state (u,v,x,y,x2,y2,s,H) in VGPRs, (r-1,t-1) neighbours by DPP wave_shr:1 with the inter-chunk carry
//     through v_readlane, direction bytes in LDS or an HBM slab, exact max / arg-max by DPP reductions.  Its dp_main_loop
//     keeps the reference's 16-lane block rounding of [st,en] bit for bit (lanes outside the band but inside the rounded
//     block are computed and read back exactly as the SSE code does, 8-bit wrap included): needed when the band clips.
//   * extd2_lds_kernel<VAR>  -- any shape and flag, and the single-affine extz2 variant: state in LDS in the reference's layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/psvr_engine.h"

namespace psvr {

struct DpParams {
	int32_t m;
	int32_t q, e, q2, e2;       // after the q+e <= q2+e2 swap (ksw2_extd2_sse.c:70)
	int32_t qe_pre;             // q+e BEFORE the swap: only feeds H[0] at r==0 (:60,351)
	int32_t sc_mch, sc_mis, sc_N, m1;
	int32_t w, zdrop, end_bonus, flag;
	int32_t long_thres, long_diff;
	int32_t skip;               // 1: parameter set makes the reference return right after reset (:68,93)
	int32_t nowrap_ok;          // in-band values provably fit int8 for these scoring parameters
	int8_t  mat[25];
};

struct DpBatch { // device pointers of one batch
	const int32_t *idx;        // problem ids handled by this launch (one per workgroup)
	const uint8_t *qseq; const int64_t *q_off; const int32_t *qlen;
	const uint8_t *tseq; const int64_t *t_off; const int32_t *tlen;
	psvr_extz_t *ez; uint32_t *cigar;
	uint8_t *pslab; const int64_t *p_off;   // lds kernel only: direction-byte slab, offsets in (1 << p_unit_shift)-byte units
	int32_t p_unit_shift;
	int32_t lds_per_wave;      // reg kernels: dynamic LDS bytes of one wavefront's problem
	long long n;               // problems in this launch
	uint8_t *ws; unsigned long long ws_cap;   // team kernel: per-wavefront scratch, a slice per wavefront (TeamPlan::ws_base)
	int *err;                  // set to 20 if that scratch runs out (cannot happen with the planners' bounds; never silent)
};

#define PSVR_DP_NUM_LDS_CLASSES 13
#define PSVR_DP_KIND_TINY 11
#ifndef PSVR_DP_USE_TINY
#define PSVR_DP_USE_TINY 1     /* 0 sends them to the team kernel instead: same total time on the bench workload */
#endif
#define PSVR_DP_KIND_STRIP 12
#define PSVR_DP_STRIP 16               // extd2_team_kernel: size classes count 16-column strips
#define PSVR_DP_TINY_MAX 16            // extd2_tiny_kernel: qlen, tlen <= 16, one thread per alignment
static const int kDpWaves = 4;   // alignments (wavefronts) per workgroup of the register-resident kernels
template <int K, bool PG> __global__ void extd2_reg_kernel(DpBatch B, DpParams P);   // ksw_kernels.hip
template <int K> __global__ void extd2_ring_kernel(DpBatch B, DpParams P);             // ksw_kernels.hip: the same sweep on a ring of 64 K columns that slides with the band
template <int VAR> __global__ void extd2_lds_kernel(DpBatch B, DpParams P); // ksw_kernels.hip
__global__ void extd2_tiny_kernel(DpBatch B, DpParams P, int max_rows);        // ksw_kernels.hip
// all size classes of the team kernel go out in ONE launch (a class alone rarely fills the chip): block b serves class c with
// first_block[c] <= b < first_block[c + 1]; its alignments are idx[first_slot[c] + ...], count[c] of them
struct TeamPlan {
	int32_t n_classes;
	int32_t first_block[PSVR_DP_NUM_LDS_CLASSES + 1];
	int32_t n_strips16[PSVR_DP_NUM_LDS_CLASSES];
	long long first_slot[PSVR_DP_NUM_LDS_CLASSES], count[PSVR_DP_NUM_LDS_CLASSES];
	// scratch of class c's wavefront w at ws_base[c] + w * ws_need[c] (sized by the class's longest query)
	unsigned long long ws_base[PSVR_DP_NUM_LDS_CLASSES], ws_need[PSVR_DP_NUM_LDS_CLASSES];
};
template <int LANES, int CPL, int LEAN> __global__ void extd2_team_kernel(DpBatch B, DpParams P, TeamPlan T);          // ksw_kernels.hip: the sweep, a row per lane and step
template <int LANES, int CPL, int LEAN> __global__ void extd2_team_finish_kernel(DpBatch B, DpParams P, TeamPlan T);   // z-drop / end rules and traceback, a thread per alignment
// the z-drop rule cannot trigger whatever the sequences are: a gap of any length costs at most q2 (e2 == 0), so an anti-diagonal's maximum
// is never more than 2 q2 below the running maximum (one insertion + one deletion from the cell that holds it); the team kernel's LEAN
// variant (no per-diagonal maximum) is exact then, for a caller that reads neither ez.max nor max_q / max_t
inline bool dp_zdrop_inert(const DpParams &P) { return P.e2 == 0 && (P.zdrop < 0 || P.zdrop >= 2 * P.q2) && !(P.flag & PSVR_EZ_EXTZ_ONLY); }
// the tiny / team kernels need the lean regime (values fit int8, band never clips) and only the flags they implement
__host__ __device__ inline bool dp_tiny_ok(const DpParams &P, bool fast_ok) { return fast_ok && P.nowrap_ok && !P.skip && (P.w < 0 || P.w >= PSVR_DP_TINY_MAX); }
// lanes per alignment of the team kernel for the class of problems with n_strips16 16-column strips
// A team = PSVR_DP_TEAM_LANES lanes, each with PSVR_DP_TEAM_CPL target columns of a strip in registers (strip width = their product).
// 4 x 4 was the first shape; 2 x 8 keeps the 16-column strips but spends a step's fixed cost -- neighbour exchange, boundary records,
// per-diagonal maximum -- on eight cells instead of four, and puts 32 alignments in a wavefront.  Other shapes of the row sweep on the
// bench batch (profiles/r03e_team_kernel_row_sweep.txt): 1 x 16 at two wavefronts per SIMD as fast, 4 x 4 and 4 x 8 slower.
#ifndef PSVR_DP_TEAM_LANES
#define PSVR_DP_TEAM_LANES 2
#endif
#ifndef PSVR_DP_TEAM_CPL
#define PSVR_DP_TEAM_CPL 8
#endif
__host__ __device__ inline int dp_team_lanes(int n_strips16) { return PSVR_DP_TEAM_LANES; }
// scratch bytes one wavefront of the team kernel needs for alignments with at most qmax query bases in that class
__host__ __device__ inline unsigned long long dp_team_ws_bytes(int qmax, int n_strips16, int lanes, int cpl = PSVR_DP_TEAM_CPL)
{
	const int sw = cpl * lanes, pb = 64 / lanes, n_strips = (n_strips16 * 16 + sw - 1) / sw;
	// direction bytes (one per cell, 64 x cpl per step), then per row / diagonal and alignment: two boundary dwords (ping-pong), the key D and the
	// dword with the two band-end values (16 bytes; sized for 20: a fifth dword per diagonal is head-room, not used)
	return (unsigned long long)(64 * cpl) * n_strips * (qmax + sw - 1) + (unsigned long long)pb * 20 * (qmax + sw * n_strips + 1);
}

// true when the band [(r-w+1)>>1, (r+w)>>1] never clips the DP matrix: then st0/en0 follow the matrix edges only, every
// in-band cell's (r-1,t-1)/(r-1,t) neighbours are in-band or one of the explicit boundary values (ksw2_extd2_sse.c:142-156),
// and the lanes of the 16-rounded blocks outside the band are never read back
__host__ __device__ inline bool dp_band_never_binds(int qlen, int tlen, int w) { return qlen <= w && tlen <= w + 1; }

// ---- size classes shared by the host planner (ksw_host.hip) and the device-side planner (engine.hip)
#define PSVR_DP_MAX_LDS (160 * 1024)
__host__ __device__ inline int dp_lds_class_bytes(int cls)
{
	const int t[PSVR_DP_NUM_LDS_CLASSES] = {2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072, PSVR_DP_MAX_LDS};
	return t[cls];
}
__host__ __device__ inline int dp_n_col(int qlen, int tlen, int w_in)
{
	int w = w_in < 0 ? (qlen > tlen ? qlen : tlen) : w_in;
	int n_col = qlen < tlen ? qlen : tlen;
	n_col = ((n_col < w + 1 ? n_col : w + 1) + 15) / 16 + 1;
	return n_col;
}
__host__ __device__ inline long long dp_reg_lds_need(int qlen, int tlen, int w_in)
{
	return (long long)((qlen + 16 + 15) & ~15) + (long long)(qlen + tlen - 1) * dp_n_col(qlen, tlen, w_in) * 16 + 16;
}
__host__ __device__ inline long long dp_p_bytes(int qlen, int tlen, int w_in)
{
	return ((long long)(qlen + tlen - 1) * dp_n_col(qlen, tlen, w_in) + 1) * 16;
}
__host__ __device__ inline int dp_lds_kernel_need(int qlen, int tlen, int variant)
{
	int T = (tlen + 15) / 16 * 16, QL = (qlen + 15) / 16 * 16;
	int narr = variant == 0 ? 7 : 5;
	int img = narr * T + T + QL + 16;
	return ((img + 15) & ~15) + 4 * T;
}
// Direction bytes stay in LDS only while the whole footprint is at most this (keeps >= 32 waves per CU resident);
// larger problems stream them to an HBM slab and trace back through L2
#define PSVR_DP_PG_THRESHOLD 4096
#define PSVR_DP_NUM_KINDS 15
#define PSVR_DP_KIND_RING3 13          // extd2_ring_kernel<3>: any tlen, band (+ its 16-lane rounding) within 192 columns
#define PSVR_DP_KIND_RING4 14          // extd2_ring_kernel<4>: ... within 256 columns
// kinds whose direction bytes live in the HBM slab (DpBatch::pslab)
__host__ __device__ inline bool dp_kind_uses_slab(int kind) { return kind == 0 || (kind > 5 && kind < PSVR_DP_KIND_TINY) || kind >= PSVR_DP_KIND_RING3; }
// kind: 1..5 = extd2_reg_kernel<kind,false> (direction bytes in LDS), 6..10 = extd2_reg_kernel<kind-5,true> (in HBM), 13 / 14 = extd2_ring_kernel<3 / 4>,
// 11 = extd2_tiny_kernel (one thread per alignment; *need = 512 x anti-diagonals, which bins the problems by size),
// 0 = general kernel, -1 = unsupported; *need = dynamic LDS bytes
__host__ __device__ inline int dp_classify(int qlen, int tlen, int w, bool fast_ok, int variant, bool skip, int *need, bool tiny_ok = false, bool team_ok = true, bool ring_ok = true)
{
	if (qlen <= 0 || tlen <= 0 || skip) { *need = 0; return 1; }
	if (PSVR_DP_USE_TINY && tiny_ok && qlen <= PSVR_DP_TINY_MAX && tlen <= PSVR_DP_TINY_MAX) { *need = (qlen + tlen - 1) * 512; return PSVR_DP_KIND_TINY; }
	// one thread per alignment, 16-column strips in registers: whenever the band never clips the matrix (the lean regime).
	// The size class is the number of strips (1..13), expressed through `need` as that class's byte threshold.
	// (team_ok = false: a batch too small to fill the chip with 16 alignments per wavefront goes to the wavefront-per-alignment kernels,
	// whose sweep is qlen + tlen steps instead of strips x (qlen + 15))
	if (tiny_ok && team_ok && dp_band_never_binds(qlen, tlen, w < 0 ? (qlen > tlen ? qlen : tlen) : w) && tlen <= PSVR_DP_STRIP * PSVR_DP_NUM_LDS_CLASSES) {
		*need = dp_lds_class_bytes((tlen + PSVR_DP_STRIP - 1) / PSVR_DP_STRIP - 1);
		return PSVR_DP_KIND_STRIP;
	}
	int T = (tlen + 15) / 16 * 16;
	long long n = dp_reg_lds_need(qlen, tlen, w);
	if (fast_ok && T <= 320) {
		if (n <= PSVR_DP_PG_THRESHOLD) { *need = (int)n; return (T + 63) / 64; }
		*need = ((qlen + 16 + 15) & ~15) + 16;
		return 5 + (T + 63) / 64;
	}
	if (fast_ok && ring_ok) {
		// wider than the register-resident kernels' 320 columns: the ring kernels, when the columns an anti-diagonal can touch -- the band,
		// w + 1 wide at most (and never wider than the shorter sequence), plus the 16-lane rounding at both ends, the stale-score block and
		// the left neighbour of its first column -- fit their ring.  LDS: the query image and the target.
		const int wf = w < 0 ? (qlen > tlen ? qlen : tlen) : w, sh = qlen < tlen ? qlen : tlen;
		const int span = (wf < sh - 1 ? wf : sh - 1) + 33;
		if (span <= 256) {
			*need = ((qlen + 16 + 15) & ~15) + ((tlen + 15) & ~15) + 16;
			return span <= 192 ? PSVR_DP_KIND_RING3 : PSVR_DP_KIND_RING4;
		}
	}
	int g = dp_lds_kernel_need(qlen, tlen, variant);
	*need = g;
	return g <= PSVR_DP_MAX_LDS ? 0 : -1;
}

__device__ __forceinline__ int s8(int v) { return (int)(int8_t)v; }

// mark a value the code knows to be identical in all lanes of the wavefront as uniform (-> SGPR, scalar control flow)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v)
{
	const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
	return (long long)((unsigned long long)hi << 32 | lo);
}

// DPP controls (GFX9): row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143
__device__ __forceinline__ int dpp_wave_shr1(int v, int carry_in)
{
	return __builtin_amdgcn_update_dpp(carry_in, v, 0x138, 0xf, 0xf, false);
}

// wave-wide max of a signed 32-bit value; result valid in every lane (broadcast from lane 63)
__device__ __forceinline__ int wave_max_i32(int v)
{
	const int idn = (int)0x80000000;
	int t;
	t = __builtin_amdgcn_update_dpp(idn, v, 0x111, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(idn, v, 0x112, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(idn, v, 0x114, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(idn, v, 0x118, 0xf, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(idn, v, 0x142, 0xa, 0xf, false); v = max(v, t);
	t = __builtin_amdgcn_update_dpp(idn, v, 0x143, 0xc, 0xf, false); v = max(v, t);
	return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
	const int idn = (int)0xffffffff;
	unsigned t;
	t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x111, 0xf, 0xf, false); v = min(v, t);
	t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x112, 0xf, 0xf, false); v = min(v, t);
	t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x114, 0xf, 0xf, false); v = min(v, t);
	t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x118, 0xf, 0xf, false); v = min(v, t);
	t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x142, 0xa, 0xf, false); v = min(v, t);
	t = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)v, 0x143, 0xc, 0xf, false); v = min(v, t);
	return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// band limits of anti-diagonal r (ksw2_extd2_sse.c:125-140); returns false when st > en
__device__ __forceinline__ bool band_limits(int r, int qlen, int tlen, int w, int &st0, int &en0, int &st, int &en)
{
	st0 = 0, en0 = tlen - 1;
	if (st0 < r - qlen + 1) st0 = r - qlen + 1;
	if (en0 > r) en0 = r;
	if (st0 < ((r - w + 1) >> 1)) st0 = (r - w + 1) >> 1;
	if (en0 > ((r + w) >> 1)) en0 = (r + w) >> 1;
	st = st0 & ~15;
	en = ((en0 + 16) & ~15) - 1;
	return st0 <= en0;
}

struct EzAcc { // running ksw_extz_t (uniform per wave)
	int max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end;
	__device__ __forceinline__ void reset()
	{
		max_q = max_t = mqe_t = mte_q = -1;
		max = 0, score = mqe = mte = PSVR_KSW_NEG_INF;
		zdropped = 0, reach_end = 0;
	}
	// ksw_apply_zdrop with is_rot=1 (ksw2.h:245-261)
	__device__ __forceinline__ bool apply_zdrop(int H, int r, int t, int zdrop, int e)
	{
		if (H > max) {
			max = H, max_t = t, max_q = r - t;
		} else if (t >= max_t && r - t >= max_q) {
			int tl = t - max_t, ql = (r - t) - max_q;
			int l = tl > ql ? tl - ql : ql - tl;
			if (zdrop >= 0 && max - H > zdrop + l * e) {
				zdropped = 1;
				return true;
			}
		}
		return false;
	}
};

// Traceback over direction bytes (ksw_backtrack_D with is_rot=1, min_intron_len=0; ksw2.h:119-151).
// `P` reads one byte of row r at column offset k; ops are staged through `emit(k, word)`.
// Every lane runs the same (uniform) walk.  Returns the number of CIGAR ops.
template <class ReadP, class Emit>
__device__ __forceinline__ int traceback(int i0, int j0, int qlen, int tlen, int w, ReadP readp, Emit emit)
{
	int i = i0, j = j0, state = 0, n = 0;
	int cur_op = -1, cur_len = 0;
	auto push = [&](int op, int len) {
		if (op == cur_op) cur_len += len;
		else {
			if (cur_op >= 0) emit(n++, (uint32_t)cur_len << 4 | (uint32_t)cur_op);
			cur_op = op, cur_len = len;
		}
	};
	while (i >= 0 && j >= 0) {
		int r = i + j, st0, en0, off, off_end, force_state = -1;
		band_limits(r, qlen, tlen, w, st0, en0, off, off_end);
		if (i < off) force_state = 2;
		if (i > off_end) force_state = 1;
		int tmp = force_state < 0 ? readp(r, i - off) : 0;
		if (state == 0) state = tmp & 7;
		else if (!((tmp >> (state + 2)) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) push(0, 1), --i, --j;
		else if (state == 1 || state == 3) push(2, 1), --i;
		else push(1, 1), --j;
	}
	if (i >= 0) push(2, i + 1);
	if (j >= 0) push(1, j + 1);
	if (cur_op >= 0) emit(n++, (uint32_t)cur_len << 4 | (uint32_t)cur_op);
	return n;
}

} // namespace psvr
