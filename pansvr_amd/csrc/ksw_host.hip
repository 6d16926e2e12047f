// ksw_host.hip -- host side of seam B2 (psvr_extd2_batch / psvr_dp_plan_*): parameter
// preparation (ksw2_extd2_sse.c:60-98), size-class planning and kernel launches.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "common.h"
#include "ksw_device.h"
#include "ksw_launch.h"

namespace psvr {

std::string &last_error_ref()
{
	static thread_local std::string s;
	return s;
}

static const int kMaxLen = 8000;          // per-sequence limit of the general kernel
static const int kMaxLds = 160 * 1024;    // gfx950: 160 KiB LDS per CU / workgroup
static const int kLdsClasses[] = {2048, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072, kMaxLds};
static const int kNumLdsClasses = sizeof(kLdsClasses) / sizeof(int);

int make_dp_params(const psvr_ksw_params_t *par, int variant, DpParams *P)
{
	memset(P, 0, sizeof *P);
	int m = par->m, q = par->q, e = par->e, q2 = par->q2, e2 = par->e2;
	if (m > 5 || m < 0) return set_error(PSVR_ERR_UNSUPPORTED, "alphabet size m=%d (supported: 0..5)", m);
	P->m = m;
	P->qe_pre = q + e;
	memcpy(P->mat, par->mat, 25);
	P->w = par->w, P->zdrop = par->zdrop, P->end_bonus = par->end_bonus, P->flag = par->flag;
	if (variant == 0) {
		if (m <= 1) { P->skip = 1; return PSVR_OK; }
		if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t, t = e, e = e2, e2 = t; }   // :70
	} else {
		if (m <= 0) { P->skip = 1; return PSVR_OK; }
		q2 = q, e2 = e;
	}
	P->q = q, P->e = e, P->q2 = q2, P->e2 = e2;
	P->sc_mch = par->mat[0], P->sc_mis = par->mat[1];
	P->sc_N = par->mat[m * m - 1] == 0 ? (int8_t)(-e2) : par->mat[m * m - 1];
	if (variant == 1) P->sc_N = par->mat[m * m - 1] == 0 ? (int8_t)(-e) : par->mat[m * m - 1];
	P->m1 = m - 1;
	int min_sc = par->mat[1];
	for (int t = 1; t < m * m; ++t) min_sc = std::min<int>(min_sc, par->mat[t]);
	if (-min_sc > 2 * (q + e)) { P->skip = 1; return PSVR_OK; }                 // :93
	{
		// in-band deltas of the difference recurrences stay within [-(q2+e2) - max_sc, max_sc + 2(q2+e2)] and the sums the
		// kernel forms within twice that: far inside int8 for the usual scoring (2/-12, 16+1, 32+0 -> |v| <= 100)
		int max_sc = par->mat[0];
		for (int t = 1; t < m * m; ++t) max_sc = std::max<int>(max_sc, par->mat[t]);
		int g = std::max(q + e, q2 + e2);
		P->nowrap_ok = (max_sc + 3 * g + std::max(-min_sc, 0) <= 127) && q >= 0 && e >= 0 && q2 >= 0 && e2 >= 0;
	}
	if (variant == 0) {
		int lt = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;                             // :95-98
		if (q2 + e2 + lt * e2 > q + e + lt * e) ++lt;
		P->long_thres = lt;
		P->long_diff = lt * (e - e2) - (q2 - q) - e2;
	}
	return PSVR_OK;
}

struct Launch {
	int kind;           // 1..5: extd2_reg_kernel<kind>; 0: extd2_lds_kernel
	int lds_bytes;
	int64_t first, count;   // slice of the index list
	int qmax;               // team kernel: the longest query of the class
};

} // namespace psvr

using namespace psvr;

struct psvr_dp_plan {
	int device = 0, variant = 0;
	int64_t n = 0;
	DpParams P;
	std::vector<Launch> launches;
	DevBuf d_idx, d_poff, d_qlen, d_tlen, d_wstop;
	int64_t pslab_bytes = 0, ws_bytes = 0;    // direction-byte slab, then (256-aligned) the strip kernel's scratch
	std::string desc;
};




extern "C" const char *psvr_last_error(void) { return last_error_ref().c_str(); }
extern "C" void *psvr_host_alloc(size_t bytes)
{
	void *p = nullptr;
	hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocPortable);   // page-locked for every device of the process
	if (e != hipSuccess) { set_error(PSVR_ERR_NOMEM, "psvr_host_alloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
	return p;
}
extern "C" void psvr_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int psvr_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) { set_error(PSVR_ERR_DEVICE, "hipGetDeviceCount failed"); return -1; }
	return n;
}

extern "C" int psvr_dp_plan_create(int device, int64_t n, const int32_t *qlen, const int32_t *tlen,
                                   const psvr_ksw_params_t *par, int variant, psvr_dp_plan_t **out)
{
	if (!out || !par || n < 0 || (n > 0 && (!qlen || !tlen)) || (variant != 0 && variant != 1))
		return set_error(PSVR_ERR_ARG, "psvr_dp_plan_create: bad argument");
	*out = nullptr;
	psvr_dp_plan *pl = new psvr_dp_plan;
	pl->device = device, pl->variant = variant, pl->n = n;
	int rc = make_dp_params(par, variant, &pl->P);
	if (rc) { delete pl; return rc; }
	{
		hipError_t he = hipSetDevice(device);
		if (he != hipSuccess) { delete pl; return set_error(PSVR_ERR_DEVICE, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(he)); }
	}
	const int fast_flags = PSVR_EZ_EXTZ_ONLY | PSVR_EZ_REV_CIGAR | PSVR_EZ_SCORE_ONLY;
	const bool fast_ok = variant == 0 && (par->flag & ~fast_flags) == 0;
	const bool ring_ok = getenv("PSVR_DP_NO_RING") == nullptr;      // (A/B runs: wide shapes through the general kernel)
	// bucket = kind * classes + lds class
	std::vector<std::vector<int32_t>> bucket(PSVR_DP_NUM_KINDS * kNumLdsClasses);
	std::vector<int64_t> poff(n, 0);
	int64_t pslab = 0;
	for (int64_t i = 0; i < n; ++i) {
		int ql = qlen[i], tl = tlen[i];
		if (ql > kMaxLen || tl > kMaxLen) {
			delete pl;
			return set_error(PSVR_ERR_UNSUPPORTED, "problem %lld: qlen=%d tlen=%d exceeds %d", (long long)i, ql, tl, kMaxLen);
		}
		int need = 0;
		int kind = dp_classify(ql, tl, par->w, fast_ok, variant, pl->P.skip != 0, &need, dp_tiny_ok(pl->P, fast_ok), true, ring_ok);
		if (kind < 0) { delete pl; return set_error(PSVR_ERR_UNSUPPORTED, "problem %lld needs %d B of LDS", (long long)i, need); }
		if (dp_kind_uses_slab(kind) && ql > 0 && tl > 0) {
			poff[i] = pslab;
			pslab += (dp_p_bytes(ql, tl, par->w) + 255) & ~(int64_t)255;
		}
		int cls = 0;
		while (kLdsClasses[cls] < need) ++cls;
		bucket[kind * kNumLdsClasses + cls].push_back((int32_t)i);
	}
	std::vector<int32_t> idx;
	idx.reserve(n);
	// general kernel first, then the HBM-direction-byte kernels, then the LDS ones; large LDS classes first
	const int kind_order[PSVR_DP_NUM_KINDS] = {0, 14, 13, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 12, 11};
	for (int ko = 0; ko < PSVR_DP_NUM_KINDS; ++ko)
		for (int cls = kNumLdsClasses - 1; cls >= 0; --cls) {
			auto &b = bucket[kind_order[ko] * kNumLdsClasses + cls];
			if (b.empty()) continue;
			// team kernel: alignments of similar query length share a wavefront (their strips take similar numbers of steps)
			if (kind_order[ko] == PSVR_DP_KIND_STRIP) std::stable_sort(b.begin(), b.end(), [&](int32_t x, int32_t y) { return qlen[x] > qlen[y]; });
			Launch L{kind_order[ko], kLdsClasses[cls], (int64_t)idx.size(), (int64_t)b.size(), 0};
			if (L.kind == PSVR_DP_KIND_STRIP) {      // a wavefront's scratch slice is sized by the class's longest query
				for (int32_t i : b) L.qmax = std::max(L.qmax, qlen[i]);
				const int lanes = dp_team_lanes(cls + 1);
				pl->ws_bytes += (int64_t)(((uint64_t)b.size() * lanes + 63) / 64 * dp_team_ws_bytes(L.qmax > 0 ? L.qmax : 1, cls + 1, lanes));
			}
			pl->launches.push_back(L);
			idx.insert(idx.end(), b.begin(), b.end());
			char buf[160];
			snprintf(buf, sizeof buf, "%s[lds=%d] x%lld; ", dp_kind_name(L.kind, variant), L.lds_bytes, (long long)L.count);
			pl->desc += buf;
		}
	pl->pslab_bytes = pslab;
	PSVR_HIP(pl->d_idx.alloc(idx.size() * 4));
	PSVR_HIP(pl->d_poff.alloc(n * 8));
	PSVR_HIP(pl->d_qlen.alloc(n * 4));
	PSVR_HIP(pl->d_tlen.alloc(n * 4));
	PSVR_HIP(pl->d_wstop.alloc(16));      // the error flag (4 B, at offset 8)
	if (n) {
		PSVR_HIP(hipMemcpy(pl->d_idx.p, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
		PSVR_HIP(hipMemcpy(pl->d_poff.p, poff.data(), n * 8, hipMemcpyHostToDevice));
		PSVR_HIP(hipMemcpy(pl->d_qlen.p, qlen, n * 4, hipMemcpyHostToDevice));
		PSVR_HIP(hipMemcpy(pl->d_tlen.p, tlen, n * 4, hipMemcpyHostToDevice));
	}
	PSVR_HIP(dp_allow_big_lds());
	*out = pl;
	return PSVR_OK;
}

extern "C" int64_t psvr_dp_plan_workspace_bytes(const psvr_dp_plan_t *pl) { return pl ? ((pl->pslab_bytes + 255) & ~(int64_t)255) + pl->ws_bytes + 256 : 0; }

extern "C" int psvr_dp_plan_describe(const psvr_dp_plan_t *pl, char *buf, size_t buflen)
{
	if (!pl || !buf || !buflen) return set_error(PSVR_ERR_ARG, "psvr_dp_plan_describe: bad argument");
	snprintf(buf, buflen, "%s", pl->desc.c_str());
	return PSVR_OK;
}

extern "C" void psvr_dp_plan_destroy(psvr_dp_plan_t *pl) { delete pl; }

extern "C" int psvr_dp_plan_launch(psvr_dp_plan_t *pl, const uint8_t *d_qseq, const int64_t *d_q_off,
                                   const uint8_t *d_tseq, const int64_t *d_t_off,
                                   psvr_extz_t *d_ez, uint32_t *d_cigar, void *d_work, void *stream_)
{
	if (!pl) return set_error(PSVR_ERR_ARG, "psvr_dp_plan_launch: null plan");
	if (pl->n == 0) return PSVR_OK;
	if (!d_qseq || !d_q_off || !d_tseq || !d_t_off || !d_ez || !d_cigar || ((pl->pslab_bytes || pl->ws_bytes) && !d_work))
		return set_error(PSVR_ERR_ARG, "psvr_dp_plan_launch: null device pointer");
	hipStream_t stream = (hipStream_t)stream_;
	DpBatch B;
	B.qseq = d_qseq, B.q_off = d_q_off, B.qlen = pl->d_qlen.as<int32_t>();
	B.tseq = d_tseq, B.t_off = d_t_off, B.tlen = pl->d_tlen.as<int32_t>();
	B.ez = d_ez, B.cigar = d_cigar;
	B.pslab = (uint8_t *)d_work, B.p_off = pl->d_poff.as<int64_t>(), B.p_unit_shift = 0;
	B.ws = (uint8_t *)d_work + ((pl->pslab_bytes + 255) & ~(int64_t)255), B.ws_cap = (unsigned long long)pl->ws_bytes;
	B.err = (int *)(pl->d_wstop.as<unsigned long long>() + 1);
	PSVR_HIP(hipMemsetAsync(pl->d_wstop.p, 0, 16, stream));
	TeamLaunch team;
	for (const Launch &L : pl->launches) {
		if (L.kind == PSVR_DP_KIND_STRIP) { team.add(dp_class_of(L.lds_bytes) + 1, L.first, L.count, L.qmax); continue; }
		B.idx = pl->d_idx.as<int32_t>() + L.first;
		dp_launch_kind(L.kind, pl->variant, (unsigned)L.count, L.lds_bytes, stream, B, pl->P);
		PSVR_HIP(hipGetLastError());
	}
	B.idx = pl->d_idx.as<int32_t>();
	// (tests: PSVR_DP_FORCE_LEAN=1 sends a plan whose parameters make the z-drop rule inert through the variant the engine's own launches
	// use; ez.max / max_q / max_t of its problems are then not filled in)
	team.launch(stream, B, pl->P, getenv("PSVR_DP_FORCE_LEAN") != nullptr && dp_zdrop_inert(pl->P));
	PSVR_HIP(hipGetLastError());
	return PSVR_OK;
}

static int dp_batch_host(int variant, int device, int64_t n,
                         const uint8_t *qseq, const int64_t *q_off, const int32_t *qlen,
                         const uint8_t *tseq, const int64_t *t_off, const int32_t *tlen,
                         const psvr_ksw_params_t *par, psvr_extz_t *ez, uint32_t *cigar, int64_t cigar_cap)
{
	if (n < 0 || !par || (n > 0 && (!qseq || !q_off || !qlen || !tseq || !t_off || !tlen || !ez)))
		return set_error(PSVR_ERR_ARG, "psvr_ext*_batch: bad argument");
	if (n == 0) return PSVR_OK;
	int ndev = psvr_device_count();
	if (ndev <= 0) return set_error(PSVR_ERR_DEVICE, "no HIP device visible: the engine has no CPU path");
	if (device < 0 || device >= ndev) return set_error(PSVR_ERR_ARG, "device %d out of range (have %d)", device, ndev);
	int64_t qbytes = 0, tbytes = 0, cig = 0;
	for (int64_t i = 0; i < n; ++i) {
		qbytes = std::max<int64_t>(qbytes, q_off[i] + std::max(qlen[i], 0));
		tbytes = std::max<int64_t>(tbytes, t_off[i] + std::max(tlen[i], 0));
		ez[i].cigar_off = cig;
		cig += psvr_cigar_bound(std::max(qlen[i], 0), std::max(tlen[i], 0));
	}
	const bool want_cigar = !(par->flag & PSVR_EZ_SCORE_ONLY);
	if (want_cigar && (cig > cigar_cap || !cigar))
		return set_error(PSVR_ERR_OVERFLOW, "cigar arena too small: need %lld uint32, have %lld", (long long)cig, (long long)cigar_cap);
	psvr_dp_plan_t *pl = nullptr;
	int rc = psvr_dp_plan_create(device, n, qlen, tlen, par, variant, &pl);
	if (rc) return rc;
	struct Guard { psvr_dp_plan_t *p; ~Guard() { psvr_dp_plan_destroy(p); } } guard{pl};
	DevBuf dq, dt, dqo, dto, dez, dcig, dwork;
	PSVR_HIP(dq.alloc(qbytes + 16)); PSVR_HIP(dt.alloc(tbytes + 16));
	PSVR_HIP(dqo.alloc(n * 8)); PSVR_HIP(dto.alloc(n * 8));
	PSVR_HIP(dez.alloc(n * sizeof(psvr_extz_t)));
	PSVR_HIP(dcig.alloc((want_cigar ? cig : 1) * 4));
	PSVR_HIP(dwork.alloc(psvr_dp_plan_workspace_bytes(pl)));
	PSVR_HIP(hipMemcpy(dq.p, qseq, qbytes, hipMemcpyHostToDevice));
	PSVR_HIP(hipMemcpy(dt.p, tseq, tbytes, hipMemcpyHostToDevice));
	PSVR_HIP(hipMemcpy(dqo.p, q_off, n * 8, hipMemcpyHostToDevice));
	PSVR_HIP(hipMemcpy(dto.p, t_off, n * 8, hipMemcpyHostToDevice));
	PSVR_HIP(hipMemcpy(dez.p, ez, n * sizeof(psvr_extz_t), hipMemcpyHostToDevice));
	rc = psvr_dp_plan_launch(pl, dq.as<uint8_t>(), dqo.as<int64_t>(), dt.as<uint8_t>(), dto.as<int64_t>(),
	                         dez.as<psvr_extz_t>(), dcig.as<uint32_t>(), dwork.p, nullptr);
	if (rc) return rc;
	PSVR_HIP(hipDeviceSynchronize());
	{
		int kerr = 0;
		PSVR_HIP(hipMemcpy(&kerr, (char *)pl->d_wstop.p + 8, 4, hipMemcpyDeviceToHost));
		if (kerr) return set_error(PSVR_ERR_OVERFLOW, "DP kernel scratch exhausted (internal error %d)", kerr);
	}
	PSVR_HIP(hipMemcpy(ez, dez.p, n * sizeof(psvr_extz_t), hipMemcpyDeviceToHost));
	if (want_cigar) PSVR_HIP(hipMemcpy(cigar, dcig.p, cig * 4, hipMemcpyDeviceToHost));
	return PSVR_OK;
}

extern "C" int psvr_extd2_batch(int device, int64_t n, const uint8_t *qseq, const int64_t *q_off, const int32_t *qlen,
                                const uint8_t *tseq, const int64_t *t_off, const int32_t *tlen,
                                const psvr_ksw_params_t *par, psvr_extz_t *ez, uint32_t *cigar, int64_t cigar_cap)
{
	return dp_batch_host(0, device, n, qseq, q_off, qlen, tseq, t_off, tlen, par, ez, cigar, cigar_cap);
}

extern "C" int psvr_extz2_batch(int device, int64_t n, const uint8_t *qseq, const int64_t *q_off, const int32_t *qlen,
                                const uint8_t *tseq, const int64_t *t_off, const int32_t *tlen,
                                const psvr_ksw_params_t *par, psvr_extz_t *ez, uint32_t *cigar, int64_t cigar_cap)
{
	return dp_batch_host(1, device, n, qseq, q_off, qlen, tseq, t_off, tlen, par, ez, cigar, cigar_cap);
}
