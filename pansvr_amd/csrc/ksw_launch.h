// ksw_launch.h -- one place that maps a DP size class to its kernel instantiation
#pragma once
#include <hip/hip_runtime.h>
#include "ksw_device.h"

namespace psvr {

inline const char *dp_kind_name(int kind, int variant)
{
	static const char *n[PSVR_DP_NUM_KINDS] = {"extd2_lds_kernel", "extd2_reg_kernel<1,lds>", "extd2_reg_kernel<2,lds>", "extd2_reg_kernel<3,lds>", "extd2_reg_kernel<4,lds>",
	                                           "extd2_reg_kernel<5,lds>", "extd2_reg_kernel<1,hbm>", "extd2_reg_kernel<2,hbm>", "extd2_reg_kernel<3,hbm>", "extd2_reg_kernel<4,hbm>",
	                                           "extd2_reg_kernel<5,hbm>", "extd2_tiny_kernel", "extd2_team_kernel", "extd2_ring_kernel<3>", "extd2_ring_kernel<4>"};
	if (kind == 0 && variant == 1) return "extz2_lds_kernel";
	return n[kind];
}

inline void dp_launch_kind(int kind, int variant, unsigned count, int lds, hipStream_t stream, const DpBatch &B0, const DpParams &P)
{
	DpBatch B = B0;
	B.n = count, B.lds_per_wave = lds;
	if (kind == PSVR_DP_KIND_TINY) {        // `lds` is the size bin: 512 bytes per anti-diagonal
		const int max_rows = lds / 512;
		hipLaunchKernelGGL(extd2_tiny_kernel, dim3((count + 63) / 64), dim3(64), (size_t)8192 + (size_t)max_rows * 1024, stream, B, P, max_rows);
		return;
	}
	const bool reg = kind >= 1;
	dim3 g(reg ? (count + kDpWaves - 1) / kDpWaves : count), b(reg ? 64 * kDpWaves : 64);
	if (reg) lds *= kDpWaves;
	switch (kind) {
	case 1: hipLaunchKernelGGL((extd2_reg_kernel<1, false>), g, b, lds, stream, B, P); break;
	case 2: hipLaunchKernelGGL((extd2_reg_kernel<2, false>), g, b, lds, stream, B, P); break;
	case 3: hipLaunchKernelGGL((extd2_reg_kernel<3, false>), g, b, lds, stream, B, P); break;
	case 4: hipLaunchKernelGGL((extd2_reg_kernel<4, false>), g, b, lds, stream, B, P); break;
	case 5: hipLaunchKernelGGL((extd2_reg_kernel<5, false>), g, b, lds, stream, B, P); break;
	case 6: hipLaunchKernelGGL((extd2_reg_kernel<1, true>), g, b, lds, stream, B, P); break;
	case 7: hipLaunchKernelGGL((extd2_reg_kernel<2, true>), g, b, lds, stream, B, P); break;
	case 8: hipLaunchKernelGGL((extd2_reg_kernel<3, true>), g, b, lds, stream, B, P); break;
	case 9: hipLaunchKernelGGL((extd2_reg_kernel<4, true>), g, b, lds, stream, B, P); break;
	case 10: hipLaunchKernelGGL((extd2_reg_kernel<5, true>), g, b, lds, stream, B, P); break;
	case PSVR_DP_KIND_RING3: hipLaunchKernelGGL(extd2_ring_kernel<3>, g, b, lds, stream, B, P); break;
	case PSVR_DP_KIND_RING4: hipLaunchKernelGGL(extd2_ring_kernel<4>, g, b, lds, stream, B, P); break;
	default:
		if (variant == 0) hipLaunchKernelGGL(extd2_lds_kernel<0>, g, b, lds, stream, B, P);
		else hipLaunchKernelGGL(extd2_lds_kernel<1>, g, b, lds, stream, B, P);
	}
}

// the team kernel: every class in one launch (largest classes first, their wavefronts run longest), then the launch that turns what
// the sweep left in scratch into ksw_extz_t records and CIGARs
struct TeamLaunch {
	TeamPlan T;
	TeamLaunch() { T.n_classes = 0; T.first_block[0] = 0; }
	unsigned long long ws_next = 0;
	// qmax: the longest query of the class (a wavefront's scratch slice is sized by it)
	void add(int n_strips16, long long first_slot, long long count, int qmax)
	{
		const int c = T.n_classes++;
		T.n_strips16[c] = n_strips16, T.first_slot[c] = first_slot, T.count[c] = count;
		const int lanes = dp_team_lanes(n_strips16), pb = 64 / lanes;
		const int blocks = (int)((count + pb - 1) / pb);
		T.first_block[c + 1] = T.first_block[c] + blocks;
		T.ws_need[c] = dp_team_ws_bytes(qmax > 0 ? qmax : 1, n_strips16, lanes);
		T.ws_base[c] = ws_next;
		ws_next += T.ws_need[c] * (unsigned long long)blocks;
	}
	// lean: the variant without the per-diagonal maximum -- only for a caller that has checked dp_zdrop_inert(P) and reads neither ez.max nor max_q / max_t
	void launch_sweep(hipStream_t stream, const DpBatch &B, const DpParams &P, bool lean = false) const
	{
		if (!T.n_classes) return;
		if (lean) hipLaunchKernelGGL((extd2_team_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 1>), dim3((unsigned)T.first_block[T.n_classes]), dim3(64), 0, stream, B, P, T);
		else hipLaunchKernelGGL((extd2_team_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 0>), dim3((unsigned)T.first_block[T.n_classes]), dim3(64), 0, stream, B, P, T);
	}
	void launch_finish(hipStream_t stream, const DpBatch &B, const DpParams &P, bool lean = false) const
	{
		if (!T.n_classes) return;
		const unsigned blocks = (unsigned)T.first_block[T.n_classes], pb = 64u / PSVR_DP_TEAM_LANES;
		if (lean) hipLaunchKernelGGL((extd2_team_finish_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 1>), dim3((blocks * pb + 63u) / 64u), dim3(64), 0, stream, B, P, T);
		else hipLaunchKernelGGL((extd2_team_finish_kernel<PSVR_DP_TEAM_LANES, PSVR_DP_TEAM_CPL, 0>), dim3((blocks * pb + 63u) / 64u), dim3(64), 0, stream, B, P, T);
	}
	void launch(hipStream_t stream, const DpBatch &B, const DpParams &P, bool lean = false) const { launch_sweep(stream, B, P, lean), launch_finish(stream, B, P, lean); }
};
// the class index a launch's `lds` value names (team kernel: index + 1 = number of 16-column strips)
inline int dp_class_of(int lds)
{
	int cls = 0;
	while (cls < PSVR_DP_NUM_LDS_CLASSES - 1 && dp_lds_class_bytes(cls) < lds) ++cls;
	return cls;
}

inline hipError_t dp_allow_big_lds()
{
	hipError_t e = hipSuccess;
#define PSVR_ATTR(k) do { hipError_t x = hipFuncSetAttribute((const void *)(k), hipFuncAttributeMaxDynamicSharedMemorySize, PSVR_DP_MAX_LDS); if (x != hipSuccess) e = x; } while (0)
	PSVR_ATTR((extd2_reg_kernel<1, false>)); PSVR_ATTR((extd2_reg_kernel<2, false>)); PSVR_ATTR((extd2_reg_kernel<3, false>));
	PSVR_ATTR((extd2_reg_kernel<4, false>)); PSVR_ATTR((extd2_reg_kernel<5, false>));
	PSVR_ATTR(extd2_lds_kernel<0>); PSVR_ATTR(extd2_lds_kernel<1>); PSVR_ATTR(extd2_ring_kernel<3>); PSVR_ATTR(extd2_ring_kernel<4>);
#undef PSVR_ATTR
	return e;
}

} // namespace psvr
