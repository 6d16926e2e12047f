// engine.hip -- the MI355X backend of seam B1: one named HIP kernel per stage of aln_device.h,
// device-side planning of the DP launches, the HBM-resident index and the psvr_index_* /
// psvr_engine_* entry points of include/psvr_engine.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.h"
#include "engine_core.h"
#include "host_io.h"
#include "index_build.h"
#include "ksw_device.h"
#include "ksw_launch.h"

namespace psvr {

int make_dp_params(const psvr_ksw_params_t *par, int variant, DpParams *P);   // ksw_host.hip

static const int kBlock = 256;
static inline unsigned grid_for(long long n, int block = kBlock) { return (unsigned)((n + block - 1) / block); }

__device__ __forceinline__ long long pair_of(const int32_t *work, long long i) { return work ? (long long)work[i] : i; }

// ---- stage kernels: one thread per item; the read-level stages run per mate ------------------------
// prep (prep_read in aln_device.h), one wavefront per read: coalesced base loads, N draws ordered by a ballot
// prefix, codes of both strands staged in LDS, then one lane per packed 32-base word (no per-base read-modify-write).
// `n_dev`: optional, the number of work items where only the device knows it (the slots k_prep_mate1 lists); wavefronts take items in turn
__global__ __launch_bounds__(kBlock) void k_prep(Ctx c, const int32_t *work, long long n, int mate, int tsize, int per_wave, const unsigned int *n_dev)
{
	extern __shared__ __align__(16) uint8_t prep_lds[];
	// charToDna5n as a 128-entry table at the start of the workgroup's LDS (one ds_read instead of a compare chain per base)
	if (threadIdx.x < 128) {
		const int ch = threadIdx.x;
		prep_lds[ch] = (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : (ch == 'n') ? 4 : 0;
	}
	__syncthreads();
	const uint8_t *lut = prep_lds;
	const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;   // per-read values are wave-uniform: keep them in SGPRs
	if (n_dev) n = (long long)*n_dev;
	const int wpb = (int)(blockDim.x >> 6);                           // wavefronts per workgroup: fewer for long reads (their LDS share grows with the read length)
	for (long long wi = blockIdx.x * (long long)wpb + wave; wi < n; wi += (long long)gridDim.x * wpb) {
	const long long slot = pair_of(work, wi), read = slot * 2 + mate;
	uint8_t *fw = prep_lds + 128 + (size_t)wave * per_wave, *rv = fw + c.lmax;   // codes of both strands, zero-padded to lmax
	uint64_t *pw = (uint64_t *)(rv + c.lmax);                            // forward strand's packed words (for the STR screen)
	unsigned int *bits = (unsigned int *)(pw + c.wmax);                  // tsize words: hashed 20-mer set
	// A wavefront's life here is a chain of dependent memory round trips (slot -> source read -> offsets -> bases), not arithmetic: keep the
	// chain short.  A real pair is its own source (only variant / shadow slots look theirs up), and everything that hangs on `sr` is
	// requested before anything is used.
	const long long sr = (c.src && slot >= c.n_pairs) ? (long long)c.src[slot] * 2 + mate : read;
	const uint32_t *ow = (const uint32_t *)(c.ori + sr);                 // psvr_ori_t as words: chr_id, ref_bg, read_bg, align_score, {mapq, direction, unmapped, -}
	const uint32_t o_chr = ow[0], o_score = ow[3], o_unm = (ow[4] >> 16) & 0xffu;
	const long long bo0 = c.base_off[sr], bo1 = c.base_off[sr + 1];
	const long long p_off = c.poff[slot];
	const int r_m0 = mate ? c.rcnt[slot * 3] : 0;
	const int L = (int)(bo1 - bo0);
	const long long item = slot * 3 + mate;
	bool unm = o_unm != 0 || o_chr > 24u;
	bool act = !(L > kMaxReadLen || L < kLenKmer) && !(!unm && o_score == (uint32_t)(L * c.par.match));
	if (lane == 0) {
		c.read_l[read] = L, c.unmapped[read] = unm, c.is_str[read] = 0, c.has_mem[read] = 0, c.hcnt[read] = 0, c.n_ccand[read] = 0, c.active[read] = act;
		if (L > kMaxReadLen) *c.err = 1;
	}
	if (lane < 2) { Strand &st = c.strand[read * 2 + lane]; st.mem_n = st.us_n = 0; st.mem_off = st.us_off = 0; st.seed_hash = st.chain_hash = 1469598103934665603ULL; }
	if (!act) { if (lane == 0) c.rcnt[item] = 0; continue; }
	const char *s = c.bases + bo0;
	uint8_t *b0 = c.bin + (read * 2) * (long long)c.lmax, *b1 = b0 + c.lmax;
	uint64_t *w0 = c.rb + (read * 2) * (long long)c.wmax, *w1 = w0 + c.wmax;
	const long long ro = p_off + r_m0;
	int draws = 0;
	bool any4 = false;
	// the first 256 bases are requested at once (a load per 64-base round would be a round trip each)
	char pre[4];
#pragma unroll
	for (int u = 0; u < 4; ++u) pre[u] = 64 * u + lane < L ? s[64 * u + lane] : 'A';
	auto round = [&](int i0, char ch) {
		const int i = i0 + lane;
		const bool isn = i < L && ch == 'N';
		const unsigned long long m = __ballot(isn);
		if (isn) {
			const int d = draws + __popcll(m & ((1ull << lane) - 1));
			long long k = ro + d - c.grand_base;
			int32_t r;
			if (c.force && d < (int)c.force[4 * read]) r = c.force[4 * read + 1 + d];
			else r = (k >= 0 && k < c.grand_n) ? c.grand[k] : (*c.err = 2, 0);
			ch = "ACGT"[r % 4];
		}
		draws += __popcll(m);
		const uint8_t code = lut[ch & 0x7f];                                   // input is 7-bit ASCII
		if (i < L) fw[i] = code, rv[L - 1 - i] = code ^ 3;
		else if (i < c.lmax) fw[i] = 0, rv[i] = 0;
		any4 |= __ballot(i < L && code > 3) != 0;
	};
#pragma unroll
	for (int u = 0; u < 4; ++u) if (64 * u < c.lmax) round(64 * u, pre[u]);
	for (int i0 = 256; i0 < c.lmax; i0 += 64) round(i0, i0 + lane < L ? s[i0 + lane] : 'A');
	if (lane == 0) c.has_n4[read] = any4;
	__builtin_amdgcn_wave_barrier();
	// the per-base byte form of both strands is only needed where the 2-bit words cannot hold the read: a lower-case 'n' (code 4).
	// Every other consumer (mismatch counts, DP query bytes) reads the packed words, so 2 x lmax bytes per read stay unwritten.
	if (any4) for (int i = lane; i < L; i += 64) b0[i] = fw[i], b1[i] = rv[i];
	// packed words of both strands (binary_read_64_bit, rr.cpp:295-300): word = OR of code << 2*(31 - (i & 31)).  One lane per
	// word: four codes (one per byte of a 32-bit LDS word) are gathered MSB-first into eight bits by one multiply.
	for (int j = lane; j < 2 * c.wmax; j += 64) {
		const int s2 = j >= c.wmax, w = j - s2 * c.wmax;
		uint64_t word = 0;
		if (w * 32 < c.lmax) {
			const uint8_t *src = (s2 ? rv : fw) + w * 32;
			if (!any4) {
				const uint4 lo = *(const uint4 *)src, hi = *(const uint4 *)(src + 16);
				const uint32_t x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
				for (int g = 0; g < 8; ++g) word |= (uint64_t)((x[g] * 0x40100401u) >> 24) << (56 - 8 * g);
			} else {
				// a code of 4 (lower-case n) spills its third bit into the neighbouring base exactly as the reference's shift does
				for (int k = 0; k < 32; ++k) word |= ((uint64_t)src[k]) << ((31 - k) << 1);
			}
		}
		(s2 ? w1 : w0)[w] = word;
		if (!s2) pw[w] = word;
	}
	// STR screen (rr.cpp:549-598 decides STR when fewer than kn - 15 of the kn 20-mers are distinct, i.e. at least 16 repeats):
	// hash every 20-mer into a 32*tsize-bit set; a k-mer that finds its bit taken is a repeat or a collision, so fewer than 16
	// such events prove the read is not STR.  The few reads left (is_str = 2) get the exact count in k_str_detect.
	const int kn = L - kLenKmer + 1;
	int verdict = 2;
	if (kn >= 15) {
		const unsigned bmask = (unsigned)tsize * 32u - 1u;
		for (int i = lane; i < tsize; i += 64) bits[i] = 0;
		__builtin_amdgcn_wave_barrier();
		int taken = 0;
		for (int i0 = 0; i0 < kn; i0 += 64) {
			const int i = i0 + lane;
			bool hit = false;
			if (i < kn) {
				const unsigned h = (unsigned)((get_kmer((uint32_t)i, pw) * 0x9E3779B97F4A7C15ull) >> 40) & bmask;
				hit = (atomicOr(&bits[h >> 5], 1u << (h & 31)) >> (h & 31)) & 1u;
			}
			taken += __popcll(__ballot(hit));
		}
		if (taken < 16) verdict = 0;
	}
	if (lane == 0) {
		c.is_str[read] = (uint8_t)verdict; c.rcnt[item] = draws; if (c.stats) stat_add(c, ST_READS, 1);
		if (verdict == 2) c.str_list[atomicAdd(c.str_cnt, 1u)] = (int32_t)read;      // a few percent of the reads: k_str_detect runs on these only
	}
	__builtin_amdgcn_wave_barrier();                                         // the wavefront's LDS is reused by its next read
	}
}

// prep of short reads (lmax <= 32 W), one LANE per pair, mate 0 then mate 1.  The wavefront-per-read kernel above spends ~280 vector
// instructions per read and is bound by exactly that (1.5 ms per 2 M reads however its memory round trips are arranged); here the
// lane keeps the read in registers: a dword of four bases becomes four 2-bit codes by three bit operations, one v_perm_b32 maps the
// codes back to letters to prove the four were A/C/G/T (anything else -- an 'N' that draws, an 'n', another letter, the read's end --
// takes the per-base branch), the reverse strand is the packed forward strand bit-reversed in 2-bit groups, complemented and
// shifted, and the STR screen rolls the 20-mer through the packed words into a private bit set in LDS (word k of lane l at
// [k * 64 + l]: no bank conflicts).  A read with a lower-case 'n' (code 4, which spills into its neighbour's bits and needs the
// per-base byte arrays) is redone by prep_read() of aln_device.h, the definition every path is tested against.
struct __attribute__((packed, aligned(4))) PrepU4 { uint32_t x, y, z, w; };
// Mate 1's N draws follow ALL of mate 0's draws, and mate 0's chain selection may add tie draws after this kernel has run.  So a mate 1
// that drew leaves its count marked, with the mate-0 count it started from: k_prep_mate1 (launched when mate 1's turn comes) clears the
// mark and redoes the read if mate 0's count has moved.
static const int32_t kPrepMark = 1 << 30;

__device__ __forceinline__ uint64_t rev_groups2(uint64_t x)          // the 32 2-bit groups of x in reverse order
{
	const uint64_t t = ((uint64_t)__builtin_bitreverse32((uint32_t)x) << 32) | __builtin_bitreverse32((uint32_t)(x >> 32));
	return ((t >> 1) & 0x5555555555555555ull) | ((t & 0x5555555555555555ull) << 1);
}

// Appends up to two entries per thread to a list with ONE atomic per workgroup (every thread calls it at the same place; see
// block_arena_alloc below for why: an atomic per wavefront instruction on the one counter line costs ~12 ns, and appends out of
// divergent code are an instruction per handful of lanes).
__device__ __forceinline__ void block_list_append2(int32_t *list, unsigned int *cnt, int32_t e0, int32_t e1)
{
	__shared__ unsigned int wsum2[kBlock / 64];
	__shared__ unsigned int bbase2;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
	const unsigned int n = (e0 >= 0) + (e1 >= 0);
	unsigned int x = n;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned int y = __shfl_up(x, o);
		if (lane >= o) x += y;
	}
	if (lane == 63) wsum2[wave] = x;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned int tot = 0;
		for (int w = 0; w < nw; ++w) { const unsigned int t = wsum2[w]; wsum2[w] = tot; tot += t; }
		bbase2 = tot ? atomicAdd(cnt, tot) : 0u;
	}
	__syncthreads();
	unsigned int at = bbase2 + wsum2[wave] + x - n;
	if (e0 >= 0) list[at++] = e0;
	if (e1 >= 0) list[at] = e1;
}

#ifndef PSVR_PREP_BITS
#define PSVR_PREP_BITS 11            // log2 of the bits of a lane's STR-screen set, reads of up to 160 bases
#endif
template <int W, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_prep_pair(Ctx c, const int32_t *work, long long n, int bits_log2)
{
	extern __shared__ __align__(16) uint32_t prep_bits[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int nb = 1 << (bits_log2 - 5);                                  // words of a lane's bit set
	uint32_t *bits = prep_bits + (size_t)wave * nb * 64;
	const long long wi = blockIdx.x * (long long)BLOCK + threadIdx.x;
	const bool have = wi < n;
	const long long slot = have ? pair_of(work, wi) : 0;
	const bool shadow = c.src && slot >= c.n_pairs;                       // a real pair is its own source
	const long long sp = shadow ? (long long)c.src[slot] : slot;
	const long long p_off = have ? c.poff[slot] : 0;
	int draws0 = 0;
	unsigned n_act = 0;
	int32_t listed[2] = {-1, -1};                                         // reads for the exact STR count, appended by the workgroup at the end
	// The four Strand records of a pair are 192 contiguous bytes that every read gets reset to one pattern.  A wavefront whose 64 lanes have
	// 64 consecutive pairs (the first round: no work list) writes its 12 KB of them as twelve coalesced stores instead of twelve stores of
	// a 16-byte piece per lane 192 bytes apart (64 lines each: the kernel is bound by such requests)
	bool strands_done = false;
#if !defined(PSVR_PREP_STRAND_FILL) || PSVR_PREP_STRAND_FILL
	if (work == nullptr && wi - lane + 63 < n) {
		const uint32_t hl = (uint32_t)1469598103934665603ULL, hh = (uint32_t)(1469598103934665603ULL >> 32);
		uint4 *st = (uint4 *)(c.strand + (wi - lane) * 4);
#pragma unroll
		for (int k = 0; k < 12; ++k) {
			const int j = k * 64 + lane;
			st[j] = j % 3 == 2 ? make_uint4(hl, hh, hl, hh) : make_uint4(0, 0, 0, 0);
		}
		strands_done = true;
	}
#endif
	// what both mates' turns start from, asked for once: the pair's three base offsets and the words of its two psvr_ori_t records (chr_id,
	// ref_bg, read_bg, align_score, {mapq, direction, unmapped, -}) -- mate 1's turn then starts with its bases' loads instead of a round trip
	// for their address -- and mate 1's first dword of bases, on its way while mate 0 is worked on
	long long bo_0 = 0, bo_1 = 0, bo_2 = 0;
	uint32_t oc0 = 0, os0 = 0, ou0 = 0, oc1 = 0, os1 = 0, ou1 = 0, first1 = 0;
	if (have) {
		const long long *bo = c.base_off + sp * 2;
		bo_0 = bo[0], bo_1 = bo[1], bo_2 = bo[2];
		const uint32_t *ow = (const uint32_t *)(c.ori + sp * 2);
		static_assert(sizeof(psvr_ori_t) == 20, "psvr_ori_t as five words");
		oc0 = ow[0], os0 = ow[3], ou0 = ow[4], oc1 = ow[5], os1 = ow[8], ou1 = ow[9];
		first1 = *(const uint32_t *)((uintptr_t)(c.bases + bo_1) & ~(uintptr_t)3);
	}
#if !defined(PSVR_PREP_WORDS_STAGE) || PSVR_PREP_WORDS_STAGE
	const bool coalesced = work == nullptr && wi - lane + 63 < n && 64 * 16 * c.wmax <= nb * 64 * 4;
#else
	const bool coalesced = false;
#endif
#pragma unroll 1
	for (int mate = 0; mate < 2; ++mate) {
		const long long read = slot * 2 + mate, item = slot * 3 + mate;
		uint64_t F[W], Rv[W];                                              // the packed words of both strands (when `produce`)
		int L = 0;
		// ---- the lane's own part: false = no packed words from this lane (no read, an inactive one, or one prep_read() has redone)
		const bool produce = [&]() -> bool {
		if (!have) return false;
		const uint32_t o_chr = mate ? oc1 : oc0, o_score = mate ? os1 : os0, o_unm = ((mate ? ou1 : ou0) >> 16) & 0xffu;
		const long long bo0 = mate ? bo_1 : bo_0, bo1 = mate ? bo_2 : bo_1;
		L = (int)(bo1 - bo0);
		const bool unm = o_unm != 0 || o_chr > 24u;
		const bool act = !(L > 32 * W || L < kLenKmer) && !(!unm && o_score == (uint32_t)(L * c.par.match));
		c.read_l[read] = L, c.unmapped[read] = unm, c.has_mem[read] = 0, c.hcnt[read] = 0, c.n_ccand[read] = 0, c.active[read] = act;
		if (!strands_done) {
			uint4 *st = (uint4 *)(c.strand + read * 2);                       // two Strand records: counts and offsets 0, both hashes the FNV basis
			const uint32_t hl = (uint32_t)1469598103934665603ULL, hh = (uint32_t)(1469598103934665603ULL >> 32);
			st[0] = make_uint4(0, 0, 0, 0), st[1] = make_uint4(0, 0, 0, 0), st[2] = make_uint4(hl, hh, hl, hh);
			st[3] = make_uint4(0, 0, 0, 0), st[4] = make_uint4(0, 0, 0, 0), st[5] = make_uint4(hl, hh, hl, hh);
		}
		if (!act) { c.is_str[read] = 0, c.rcnt[item] = 0; return false; }
		++n_act;
		const long long ro = p_off + (mate ? draws0 : 0);
		int draws = 0, err = 0;
		bool any4 = false;
		// ---- forward strand: 32 bases (eight dwords, at the read's own byte alignment) per packed word.  Groups are converted as if
		// the read went on (what lies behind base L-1 is cleared per word below); a group with anything but A/C/G/T is left out and noted.
		const uintptr_t a = (uintptr_t)(c.bases + bo0);
		const uint32_t *q = (const uint32_t *)(a & ~(uintptr_t)3);
		const uint32_t sh = (uint32_t)(a & 3);
		uint32_t other[W];
		uint32_t prev = mate ? first1 : q[0];
#pragma unroll
		for (int w = 0; w < W; ++w) {
			uint64_t word = 0;
			uint32_t ot = 0;
			if (32 * w < L) {
				const PrepU4 d0 = *(const PrepU4 *)(q + 8 * w + 1), d1 = *(const PrepU4 *)(q + 8 * w + 5);
				const uint32_t d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
				uint32_t hi = 0, lo = 0;
#pragma unroll
				for (int g = 0; g < 8; ++g) {
					const uint32_t x = __builtin_amdgcn_alignbyte(d[g], prev, sh);
					prev = d[g];
					const uint32_t code = ((x >> 1) ^ (x >> 2)) & 0x03030303u;      // A/a 0, C/c 1, G/g 2, T/t 3
					const uint32_t canon = __builtin_amdgcn_perm(0u, 0x54474341u, code);   // code -> 'A' 'C' 'G' 'T'
					const bool ok = (x & 0xDFDFDFDFu) == canon;
					const uint32_t g8 = ok ? ((code << 6) | (code >> 4) | (code >> 14) | (code >> 24)) & 0xffu : 0u;   // first base in the top bits
					ot |= ok ? 0u : 1u << g;
					if (g < 4) hi |= g8 << (24 - 8 * g);
					else lo |= g8 << (56 - 8 * g);
				}
				const int keep = L - 32 * w;                                      // > 0 here
				word = ((uint64_t)hi << 32) | lo;
				if (keep < 32) word &= ~0ull << (64 - 2 * keep);
			}
			F[w] = word, other[w] = ot;
		}
		// the noted groups, base by base, in read order (the draws of the 'N's are taken in that order): charToDna5n
#pragma unroll
		for (int w = 0; w < W; ++w) {
			uint32_t ot = other[w];
			while (ot) {
				const int g = __builtin_ctz(ot);
				ot &= ot - 1;
				const uint32_t x = __builtin_amdgcn_alignbyte(q[8 * w + g + 1], q[8 * w + g], sh);
				for (int b = 0; b < 4; ++b) {
					const int i = 32 * w + 4 * g + b;
					if (i >= L) break;
					char ch = (char)((x >> (8 * b)) & 0xffu);
					if (ch == 'N') {
						const long long k = ro + draws - c.grand_base;
						int32_t r;
						if (c.force && draws < (int)c.force[4 * read]) r = c.force[4 * read + 1 + draws];
						else r = (k >= 0 && k < c.grand_n) ? c.grand[k] : (err = 2, 0);
						ch = "ACGT"[r % 4];
						++draws;
					}
					const uint64_t code = (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : (ch == 'n') ? 4 : 0;
					any4 |= code > 3;
					F[w] |= code << ((31 - (i & 31)) << 1);
				}
			}
		}
		if (err) *c.err = err;
		if (mate == 0) draws0 = draws;
		if (any4) {
			// code 4 does not fit the scheme above: the reference definition redoes the read (same draws), the exact count decides STR
			if (mate) c.rcnt[slot * 3] = draws0;
			prep_read(c, read);
			if (mate && draws) c.rcnt[item] = kPrepMark | draws | (draws0 << 10);
			c.is_str[read] = 2;
			listed[mate] = (int32_t)read;
			--n_act;
			return false;
		}
		c.has_n4[read] = 0;
		c.rcnt[item] = (mate && draws) ? kPrepMark | draws | (draws0 << 10) : draws;
		// ---- reverse strand: base j is 3 - base (L-1-j).  Shift the forward string right until it ends at the array's end,
		// reverse the 2-bit groups of the whole array, complement, clear what lies behind base L-1.
		{
			uint64_t G[W];
#pragma unroll
			for (int w = 0; w < W; ++w) G[w] = F[w];
			const int s = 2 * (32 * W - L), qw = s >> 6, r = s & 63;
#pragma unroll
			for (int step = 1; step < W; step <<= 1) {
				if (qw & step) {
#pragma unroll
					for (int w = W - 1; w >= 0; --w) G[w] = w - step >= 0 ? G[w - step >= 0 ? w - step : 0] : 0;
				}
			}
			if (r) {
#pragma unroll
				for (int w = W - 1; w >= 0; --w) G[w] = (G[w] >> r) | (w ? G[w ? w - 1 : 0] << (64 - r) : 0);
			}
#pragma unroll
			for (int w = 0; w < W; ++w) {
				const int keep = L - 32 * w;                                    // bases of this word that exist
				uint64_t v = ~rev_groups2(G[W - 1 - w]);
				v = keep <= 0 ? 0 : keep >= 32 ? v : v & (~0ull << (64 - 2 * keep));
				Rv[w] = v;
			}
		}
		return true;
		}();
		// ---- the words go out.  A wavefront with 64 consecutive pairs (see the Strand records above) stages them in its LDS -- the bit sets'
		// space, not yet in use -- and writes the 16-byte pieces of its reads side by side: wmax stores a mate that touch 16 lines each
		// instead of 4 wmax stores of eight bytes per lane 32 wmax bytes apart (64 lines each)
		if (coalesced) {
			const int wm = c.wmax;
			uint64_t *stg = (uint64_t *)bits + (size_t)lane * 2 * wm;
			if (produce) {
#pragma unroll
				for (int w = 0; w < W + 2; ++w) if (w < wm) stg[w] = w < W ? F[w < W ? w : 0] : 0, stg[wm + w] = w < W ? Rv[w < W ? w : 0] : 0;
			}
			const unsigned long long pm = __ballot(produce);
			__builtin_amdgcn_wave_barrier();
			char *g0 = (char *)c.rb + ((wi - lane) * 2 + mate) * (long long)(16 * wm);
			for (int q = lane; q < 64 * wm; q += 64) {
				const int pl = q / wm, r = q - pl * wm;
				if ((pm >> pl) & 1) *(uint4 *)(g0 + (long long)pl * (32 * wm) + r * 16) = ((const uint4 *)bits)[q];
			}
			__builtin_amdgcn_wave_barrier();
		} else if (produce) {
			uint64_t *w0 = c.rb + (read * 2) * (long long)c.wmax, *w1 = w0 + c.wmax;
#pragma unroll
			for (int w = 0; w < W + 2; ++w) if (w < c.wmax) w0[w] = w < W ? F[w < W ? w : 0] : 0, w1[w] = w < W ? Rv[w < W ? w : 0] : 0;
		}
		// every lane of the wavefront, with or without a read, clears its share of the wavefront's bit sets
		for (int k = lane; k < nb * 16; k += 64) ((uint4 *)bits)[k] = make_uint4(0, 0, 0, 0);
		__builtin_amdgcn_wave_barrier();
		if (!produce) continue;
		// ---- STR screen (rr.cpp:549-598 decides STR when fewer than kn - 15 of the kn 20-mers are distinct, i.e. at least 16 repeats):
		// every 20-mer sets a hashed bit of the lane's set; one that finds its bit taken is a repeat or a collision, so fewer than
		// 16 such events prove the read is not STR.  The few reads left (is_str = 2) get the exact count in k_str_detect.
		const int kn = L - kLenKmer + 1;
		int verdict = 2;
#if defined(PSVR_DIAG_PREP) && PSVR_DIAG_PREP == 1     /* timing experiment: no STR screen (results are wrong) */
		if (kn >= 15) verdict = 0;
		if (false) {
#else
		if (kn >= 15) {
#endif
			uint32_t klo = 0, khi = 0;
			int taken = 0;
#pragma unroll
			for (int w = 0; w < W; ++w) {
				if (32 * w >= L) break;
				uint64_t cur = F[w];
				// four 20-mers per turn: their bit-set updates are in flight together (each returns the word it found)
				for (int j4 = 0; j4 < 32; j4 += 4) {
					uint32_t old[4], bit[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const int i = 32 * w + j4 + u;
						const uint32_t code = (uint32_t)(cur >> 62);
						cur <<= 2;
						khi = __builtin_amdgcn_alignbit(khi, klo, 30);
						klo = (klo << 2) | code;
						const uint32_t h = ((klo + __umul24(khi & 0xffu, 0x00C2B2AFu)) * 0x9E3779B1u) >> (32 - bits_log2);   // any function of the 40 bits will do
						bit[u] = (i >= kLenKmer - 1 && i < L) ? 1u << (h & 31) : 0u;
						old[u] = atomicOr(&bits[(h >> 5) * 64 + lane], bit[u]);
					}
#pragma unroll
					for (int u = 0; u < 4; ++u) taken += (old[u] & bit[u]) != 0;
				}
			}
			if (taken < 16) verdict = 0;
		}
		c.is_str[read] = (uint8_t)verdict;
		if (verdict == 2) listed[mate] = (int32_t)read;                                // a few percent of the reads: k_str_detect runs on these only
	}
	block_list_append2(c.str_list, c.str_cnt, listed[0], listed[1]);
	if (c.stats) {
		// one atomic per wavefront (prep_read counted the reads it redid)
		unsigned tot = n_act;
		for (int o = 32; o; o >>= 1) tot += __shfl_xor(tot, o);
		if (lane == 0 && tot) stat_add(c, ST_READS, tot);
	}
}

__global__ __launch_bounds__(kBlock) void k_prep_mate1(Ctx c, const int32_t *work, long long n, int32_t *redo, unsigned int *redo_cnt)
{
	const long long wi = blockIdx.x * (long long)kBlock + threadIdx.x;
	if (wi >= n) return;
	const long long slot = pair_of(work, wi);
	const int32_t v = c.rcnt[slot * 3 + 1];
	if (!(v & kPrepMark)) return;
	c.rcnt[slot * 3 + 1] = v & 0x3ff;
	if (c.rcnt[slot * 3] == ((v >> 10) & 0x3ff)) return;
	if (c.stats) atomicAdd(c.stats + ST_READS, ~0ull);                       // the read is counted again when it is prepared again
	redo[atomicAdd(redo_cnt, 1u)] = (int32_t)slot;                           // (a handful per million pairs) -> k_prep, one wavefront per read
}

// STR detection (rr.cpp:549-598), one wavefront per read: the read's 20-mers are counted in an open-addressing hash
// table in LDS (64-bit compare-and-swap inserts, linear probing) -- same counts as the reference's per-read std::map /
// str_detect() in aln_device.h, independent of insertion order.
__global__ __launch_bounds__(kBlock) void k_str_detect(Ctx c, const int32_t *work, long long n, int mate, int tsize, int per_wave)
{
	extern __shared__ __align__(16) uint8_t str_lds[];
	const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63;   // per-read values are wave-uniform: keep them in SGPRs
	// the reads k_prep's screen could not clear (a few percent), from its list; the grid is a fixed number of wavefronts
	const unsigned n_list = *c.str_cnt;
	const unsigned wpb = blockDim.x >> 6;
	for (unsigned e = blockIdx.x * wpb + (unsigned)wave; e < n_list; e += gridDim.x * wpb) {
	const long long read = c.str_list[e];
	unsigned long long *key = (unsigned long long *)(str_lds + (size_t)wave * (size_t)per_wave);   // keys | counts | seed_list staging
	unsigned int *cnt = (unsigned int *)(key + tsize);
	const unsigned long long EMPTY = ~0ull;               // a 20-mer has 40 significant bits
	const unsigned mask = (unsigned)tsize - 1;
	const int L = c.read_l[read];
	const uint64_t *rb = c.rb + (read * 2) * (long long)c.wmax;
	const int kn = L - kLenKmer + 1;
	for (int i = lane; i < tsize; i += 64) key[i] = EMPTY, cnt[i] = 0;
	int distinct = 0;
	for (int i0 = 0; i0 < kn; i0 += 64) {
		const int i = i0 + lane;
		bool fresh = false;
		if (i < kn) {
			const unsigned long long km = get_kmer((uint32_t)i, rb);
			unsigned h = (unsigned)((km * 0x9E3779B97F4A7C15ull) >> 40) & mask;
			for (;;) {
				unsigned long long old = atomicCAS(&key[h], EMPTY, km);
				if (old == EMPTY) { fresh = true; atomicAdd(&cnt[h], 1u); break; }
				if (old == km) { atomicAdd(&cnt[h], 1u); break; }
				h = (h + 1) & mask;
			}
		}
		distinct += __popcll(__ballot(fresh));
	}
	if (!((uint32_t)distinct < (uint32_t)kn - 15u)) { if (lane == 0) c.is_str[read] = 0; continue; }
	// an STR read (rare): per-offset mask, then the begin/end rules, staged in the table's tail
	uint8_t *sl = (uint8_t *)(cnt + tsize);
	for (int i = lane; i < kn; i += 64) {
		const unsigned long long km = get_kmer((uint32_t)i, rb);
		unsigned h = (unsigned)((km * 0x9E3779B97F4A7C15ull) >> 40) & mask;
		while (key[h] != km) h = (h + 1) & mask;
		sl[i] = cnt[h] >= 4 ? 0 : 1;
	}
	if (lane == 0) {
		c.is_str[read] = 1;
		int bg = 0, ed = 0;
		for (int o = 0; o < kSeedStep; ++o) {
			bg += sl[o] == 0, ed += sl[L - kLenKmer - o] == 0;
			sl[o] += 2, sl[L - kLenKmer - o] += 4;
		}
		if (bg < kSeedStep && ed < kSeedStep) {
			int tot = 0;
			for (int o = 0; tot < kSeedStep && o < kn; ++o) {
				if (sl[o] > 0) continue;
				sl[o] += 8, tot++;
			}
		}
	}
	uint8_t *out = c.seed_list + read * (long long)c.lmax;
	for (int i = lane; i < kn; i += 64) out[i] = sl[i];
	}
}
// K1 seed_probe + K2 mem_extend: hash gather, bucket search, unipath lookup, MEM extension
// Each thread first copies its strand's packed words (wmax x 8 B) into LDS -- row pitch an odd number of 8-byte words, so the
// lanes of a wavefront spread over the banks -- because every probe and every MEM extension re-reads them; lds_pitch == 0
// (reads too long for the LDS budget) keeps them in global memory.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_seed(Ctx c, const int32_t *work, long long n, int mate, int lds_pitch)
{
	extern __shared__ __align__(16) uint64_t seed_lds[];
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= 2 * n) return;
	const long long rs = (pair_of(work, i >> 1) * 2 + mate) * 2 + (i & 1);
	if (lds_pitch) {
		uint64_t *mine = seed_lds + (size_t)threadIdx.x * lds_pitch;
		if (c.active[rs >> 1]) {
			const uint64_t *src = c.rb + rs * (long long)c.wmax;
			for (int k = 0; k < c.wmax; ++k) mine[k] = src[k];
		}
		seed_strand_t<true>(c, rs, mine);
	} else seed_strand_t<false>(c, rs, nullptr);
}
// The device's pass over an uploaded batch (EngineCore::upload): a lane per pair counts the 'N' bases of its two reads -- aligned dwords,
// the bytes outside the read masked off, zero bytes of x ^ 'NNNN' counted exactly -- applies the early-out rule (a read the reference
// returns on before it draws, rr.cpp:413-414, or one shorter than a k-mer, draws nothing), appends (pair, n0 | n1 << 8) for the pairs that
// will draw, and folds the longest read into one maximum per wavefront.
__global__ __launch_bounds__(kBlock) void k_scan_batch(const char *bases, const long long *off, const psvr_ori_t *ori, long long P, int match, int32_t *list, unsigned int *cnt_lmax)
{
	const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	int lmax = 0, nn[2] = {0, 0};
	if (p < P) {
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const long long r = 2 * p + k, o = off[r], L = off[r + 1] - o;
			lmax = max(lmax, (int)(L < 0x7fffffff ? L : 0x7fffffff));
			const psvr_ori_t q = ori[r];
			const bool unm = q.unmapped || (uint32_t)q.chr_id > 24u;
			if ((!unm && q.align_score == (uint32_t)(L * match)) || L < kLenKmer || L > kMaxReadLen) continue;
			const uintptr_t a0 = (uintptr_t)(bases + o), a1 = a0 + (uintptr_t)L;
			int n = 0;
			for (uintptr_t a = a0 & ~(uintptr_t)3; a < a1; a += 4) {
				uint32_t x = *(const uint32_t *)a ^ 0x4E4E4E4Eu;                    // 'N' bytes become zero
				if (a < a0) x |= 0xffffffffu >> (8 * (4 - (unsigned)(a0 - a)));       // bytes in front of the read
				if (a + 4 > a1) x |= 0xffffffffu << (8 * (unsigned)(a1 - a));        // bytes behind it
				const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in every zero byte
				n += __popc(z);
			}
			nn[k] = n < 255 ? n : 255;
		}
		if (nn[0] + nn[1] >= 1) {
			const unsigned at = atomicAdd(cnt_lmax, 1u);
			list[2 * (size_t)at] = (int32_t)p, list[2 * (size_t)at + 1] = nn[0] | (nn[1] << 8);
		}
	}
	lmax = wave_max_i32(lmax);
	if ((threadIdx.x & 63) == 0 && lmax > 0) atomicMax(cnt_lmax + 1, (unsigned int)lmax);
}
// K3 chain: merge, expand, sort, sparse chaining DP
// Compaction of the items a predicate keeps into list[0 .. *cnt): kListItems items per thread (item = block base + k * blockDim + thread,
// so a wavefront's loads stay coalesced), ranks inside a wavefront from ballots, one LDS atomic per wavefront and ONE global atomic per
// workgroup -- same-address global atomics cost ~10 ns each, a workgroup per 256 items made them the kernel's time.
static const int kListItems = 4;
template <class Pred> __device__ __forceinline__ void compact_list(long long n_items, int32_t *list, unsigned int *cnt, Pred pred)
{
	__shared__ unsigned int n_blk, b_blk;
	if (threadIdx.x == 0) n_blk = 0;
	__syncthreads();
	const long long base = blockIdx.x * (long long)(kBlock * kListItems);
	int32_t val[kListItems];
	unsigned int rank[kListItems];
	bool has[kListItems];
#pragma unroll
	for (int k = 0; k < kListItems; ++k) {
		const long long i = base + (long long)k * kBlock + threadIdx.x;
		has[k] = i < n_items && pred(i, val[k]);
		const unsigned long long m = __ballot(has[k]);
		unsigned int w0 = 0;
		if ((threadIdx.x & 63) == 0 && m) w0 = atomicAdd(&n_blk, (unsigned int)__popcll(m));
		rank[k] = (unsigned int)__builtin_amdgcn_readfirstlane((int)w0) + (unsigned int)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1));
	}
	__syncthreads();
	if (threadIdx.x == 0 && n_blk) b_blk = atomicAdd(cnt, n_blk);
	__syncthreads();
#pragma unroll
	for (int k = 0; k < kListItems; ++k) if (has[k]) list[b_blk + rank[k]] = val[k];
}
// the mate's reads that have MEMs: chaining and selection leave the others as k_prep initialised them
__global__ __launch_bounds__(kBlock) void k_mem_list(Ctx c, const int32_t *work, long long n, int mate, int32_t *list, unsigned int *cnt)
{
	compact_list(n, list, cnt, [&](long long i, int32_t &read) {
		const long long r = pair_of(work, i) * 2 + mate;
		read = (int32_t)r;
		return c.active[r] && c.has_mem[r];
	});
}
__global__ __launch_bounds__(kBlock) void k_chain(Ctx c, const int32_t *list, const unsigned int *cnt)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < (long long)*cnt) chain_read(c, list[i]);
}
// chaining and chain selection of a read by the same thread, one launch: the selection walks what the chaining has just written
__global__ __launch_bounds__(kBlock) void k_chain_select(Ctx c, const int32_t *list, const unsigned int *cnt)
{
	const long long n = (long long)*cnt;          // (what k_chain_small left over: a fixed grid walks the list)
	const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
	if (n <= 2 * nwaves) {
		// a handful of reads: one per wavefront at a time.  The lanes of a wavefront that walk different reads' loops take turns, so 64 of
		// these reads in one wavefront last as long as 64 reads one after the other (~80 us for the ~800 reads a 1 M-pair round leaves over)
		if (threadIdx.x & 63) return;
		for (long long i = blockIdx.x * (long long)(blockDim.x >> 6) + (threadIdx.x >> 6); i < n; i += nwaves) { const long long r = list[i]; chain_read(c, r); select_read(c, r); }
		return;
	}
	for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) { const long long r = list[i]; chain_read(c, r); select_read(c, r); }
}
// chain + select of the listed reads in registers (chain_select_small, aln_device.h); the few reads it declines -- MEMs on both strands,
// more than two seeds, a unipath with several reference positions -- are listed for k_chain_select (one atomic per such read: a handful per
// thousand)
// `left` == nullptr: the thread takes the read through the generic pair of stages itself.  A handful of reads per thousand: their
// wavefronts last ~50 us longer, beside 17 k others -- a launch of its own for them was ~50 us of a nearly empty chip per mate.
__global__ __launch_bounds__(kBlock) void k_chain_small(Ctx c, const int32_t *list, const unsigned int *cnt, int32_t *left, unsigned int *left_cnt)
{
	const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= (long long)*cnt) return;
	const int32_t r = list[i];
	if (chain_select_small(c, r)) return;
	if (left) left[atomicAdd(left_cnt, 1u)] = r;
	else chain_read(c, r), select_read(c, r);
}
// the pairing stage over a list whose length only the device knows yet (right behind k_dirty / k_reselect, before the host has read the counts)
__global__ __launch_bounds__(kBlock) void k_pair_dev(Ctx c, const int32_t *list, const unsigned long long *cnt)
{
	const unsigned long long n = *cnt;
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) pair_reads(c, list[i]);
}
__global__ __launch_bounds__(kBlock) void k_select(Ctx c, const int32_t *list, const unsigned int *cnt)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < (long long)*cnt) select_read(c, list[i]);
}
// the reads that have candidates (a third of the reads have none: their lanes would idle through the walk of the others)
__global__ __launch_bounds__(kBlock) void k_walk_list(Ctx c, const int32_t *work, long long n, int32_t *list, unsigned int *cnt)
{
	compact_list(2 * n, list, cnt, [&](long long i, int32_t &read) {
		const long long r = pair_of(work, i >> 1) * 2 + (i & 1);
		read = (int32_t)r;
		return c.active[r] && c.n_ccand[r] > 0;
	});
}
// One reservation per WORKGROUP on a single-counter arena: every thread of the workgroup calls this at the same place with its amount
// (0 = nothing); an inclusive scan inside each wavefront, the wavefronts' sums through LDS, one atomic by thread 0.  (Atomics on one
// line are served one wavefront instruction after the other at ~12 ns: a reservation per read / candidate by every wavefront was
// 0.3-0.4 ms per step and counter.)  Returns the caller's offset, or -1 for everybody when the arena is full.
template <class T> __device__ __forceinline__ long long block_arena_alloc(const Arena<T> &a, unsigned long long n)
{
	__shared__ unsigned long long wsum[kBlock / 64];
	__shared__ long long bbase;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
	unsigned long long x = n;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned long long y = __shfl_up(x, o);
		if (lane >= o) x += y;
	}
	if (lane == 63) wsum[wave] = x;
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long tot = 0;
		for (int w = 0; w < nw; ++w) { const unsigned long long t = wsum[w]; wsum[w] = tot; tot += t; }
		long long b = 0;
		if (tot) {
			const unsigned long long o = atomicAdd(a.top, tot);
			b = (long long)o;
			if (o + tot > a.cap) { *a.overflow = 1; b = -1; }
		}
		bbase = b;
	}
	__syncthreads();
	const long long b = bbase;
	const unsigned long long mine = wsum[wave] + x - n;
	__syncthreads();                                                       // wsum / bbase may be written again by the next call
	return b < 0 ? -1 : b + (long long)mine;
}
// (four wavefronts per SIMD: the rare scratch-buffer replay of stale_compare must not cost the common path a wavefront)
#ifndef PSVR_WALK_WAVES
#define PSVR_WALK_WAVES 4
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PSVR_WALK_WAVES, 8))) void k_walk(Ctx c, const int32_t *list, const unsigned int *cnt)
{
	const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;   // one thread per read, looping over its (few) candidates
	const long long read = i < (long long)*cnt ? (long long)list[i] : -1;  // no early exit: the reservations are made by the whole workgroup
	WalkRead wr;
	walk_sizes(c, read, wr);
	const long long cw0 = block_arena_alloc(c.cw, (unsigned long long)wr.nc);
	const long long so = wr.nc > 0 ? arena_alloc(c.seg, (unsigned long long)wr.total) : 0;   // (16 counters: per lane)
	walk_body(c, read, wr, cw0, so);
	const long long id0 = block_arena_alloc(c.dp, (unsigned long long)(wr.nc > 0 ? wr.n_dp : 0));
	walk_number_dp(c, read, wr, id0);
}
__global__ __launch_bounds__(kBlock) void k_assemble(Ctx c, long long begin, long long end)
{
	const long long i = begin + blockIdx.x * (long long)blockDim.x + threadIdx.x;
	const bool have = i < end;
	AsmResult ar;
	ar.n = ar.first = 0;
	if (have) assemble_compute(c, i, ar);
	const int m = ar.n - ar.first;
	const long long co = block_arena_alloc(c.cig, (unsigned long long)(m > 0 ? m : 0));
	if (have) assemble_store(c, i, ar, co);
}
__global__ __launch_bounds__(kBlock) void k_finalize(Ctx c, const int32_t *work, long long n)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < 2 * n) finalize_read(c, pair_of(work, i >> 1) * 2 + (i & 1));
}
__global__ __launch_bounds__(kBlock) void k_pair(Ctx c, const int32_t *work, long long n)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) pair_reads(c, pair_of(work, i));
}
// tail of both reads (finalize_read) and the pairing in one pass: the thread pairs the records it has just written
#ifndef PSVR_FIN_WAVES
#define PSVR_FIN_WAVES 4
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(PSVR_FIN_WAVES, 8))) void k_finalize_pair(Ctx c, const int32_t *work, long long n)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= n) return;
	const long long p = pair_of(work, i);
	// both headers are built in registers and stored once, as six 16-byte words (field by field through c.rh they were some 30 narrow
	// stores per read, each lane to its own line, and the pairing read them back)
	psvr_read_hdr_t h[2];
	h[0].cand_off = c.rh[2 * p].cand_off, h[1].cand_off = c.rh[2 * p + 1].cand_off;
	PeItem it0[3], it1[3];                                                 // the pairing's view of the records, handed on in registers
	finalize_read(c, 2 * p, h[0], it0), finalize_read(c, 2 * p + 1, h[1], it1);
	pair_reads(c, p, h[0], h[1], it0, it1);
	static_assert(sizeof(psvr_read_hdr_t) == 48, "header size");
	uint4 w[6];
	__builtin_memcpy(w, h, sizeof w);                                      // (a copy between locals: registers; a pointer cast put the headers in scratch memory)
	uint4 *dst = (uint4 *)(c.rh + 2 * p);
#pragma unroll
	for (int k = 0; k < 6; ++k) dst[k] = w[k];
}
// compaction of the dirty pairs into the two work lists: kDirtyItems pairs per thread (pair = workgroup base + k * blockDim + thread, the
// loads stay coalesced), ranks inside a wavefront from ballots, one LDS atomic per list, wavefront and k, and ONE global atomic per list
// and workgroup -- the two counters sit in different cache lines, and a workgroup per 256 pairs bumped both: 3906 x 2 x 12 ns was the
// kernel's 92 us whatever the data (tools/atomic_rate_bench.hip)
static const int kDirtyItems = 8;
__global__ __launch_bounds__(kBlock) void k_dirty(Ctx c, const long long *noff, const long long *nhoff, int32_t *out, unsigned long long *cnt,
                                                  int32_t *outp, unsigned long long *cntp, const uint8_t *has_n, int32_t *out3, unsigned long long *cnt3, unsigned long long cap3)
{
	__shared__ unsigned int n_full, n_pair;
	__shared__ unsigned long long b_full, b_pair;
	if (threadIdx.x == 0) n_full = n_pair = 0;
	__syncthreads();
	const long long base = blockIdx.x * (long long)(kBlock * kDirtyItems);
	const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1;
	int d[kDirtyItems];
	unsigned int me[kDirtyItems];
#pragma unroll
	for (int k = 0; k < kDirtyItems; ++k) {
		const long long p = base + (long long)k * kBlock + threadIdx.x;
		d[k] = p < c.n_pairs ? mark_dirty(c, p, noff, nhoff, has_n) : 0;
		if (d[k] == 3) {                                                          // (a few hundred per million pairs: tied chains) -> k_reselect
			const unsigned long long at = atomicAdd(cnt3, 1ull);
			if (at < cap3) out3[at] = (int32_t)p, d[k] = 0; else d[k] = 2;         // no room to keep its lists for the comparison: the pair runs in full
		}
		const unsigned long long m2 = __ballot(d[k] == 2), m1 = __ballot(d[k] == 1);
		unsigned int w2 = 0, w1 = 0;
		if ((threadIdx.x & 63) == 0) {
			if (m2) w2 = atomicAdd(&n_full, (unsigned int)__popcll(m2));
			if (m1) w1 = atomicAdd(&n_pair, (unsigned int)__popcll(m1));
		}
		w2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)w2), w1 = (unsigned int)__builtin_amdgcn_readfirstlane((int)w1);
		me[k] = d[k] == 2 ? w2 + (unsigned int)__popcll(m2 & below) : w1 + (unsigned int)__popcll(m1 & below);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		b_full = n_full ? atomicAdd(cnt, (unsigned long long)n_full) : 0;
		b_pair = n_pair ? atomicAdd(cntp, (unsigned long long)n_pair) : 0;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < kDirtyItems; ++k) {
		const long long p = base + (long long)k * kBlock + threadIdx.x;
		if (d[k] == 2) out[b_full + me[k]] = (int32_t)p;
		else if (d[k] == 1) outp[b_pair + me[k]] = (int32_t)p;
	}
}
// the pairs of the third list (reselect_pair in aln_device.h): one thread each, over a list whose length only the device knows;
// a pair whose candidate lists stand joins the pairing-only list, another the list of full re-runs
__global__ __launch_bounds__(64) void k_reselect(Ctx c, const int32_t *list, const unsigned long long *n_list, unsigned long long cap3, int32_t *out4,
                                                 unsigned long long *cnt4, int32_t *outp, unsigned long long *cntp)
{
	const unsigned long long n = *n_list < cap3 ? *n_list : cap3;
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
		const long long p = list[i];
		if (reselect_pair(c, p) != 2) outp[atomicAdd(cntp, 1ull)] = (int32_t)p;   // lists stand, or are made of chains evaluated before: the pairing follows
		else out4[atomicAdd(cnt4, 1ull)] = (int32_t)p;                       // a chain nobody has walked yet: the pair goes on from the walk
	}
}
__global__ void k_copy_i32(int32_t *dst, long long at, const int32_t *src, long long n)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) dst[at + i] = src[i];
}
// one wavefront per adopted pair (adopt_variant in aln_device.h)
__global__ __launch_bounds__(kBlock) void k_adopt(Ctx c, const int32_t *pairs, const int32_t *slots, long long n, const long long *noff)
{
	const long long i = blockIdx.x * (long long)(kBlock / 64) + (threadIdx.x >> 6);
	if (i < n) adopt_variant(c, pairs[i], slots[i], noff, threadIdx.x & 63, 64);
}
// which special pairs draw alike under every residue assignment (class 1: unmasked, resolved on the device from here on)
__global__ void k_special_class(Ctx c, const SpecialPair *sp, long long n, uint8_t *mask, uint8_t *cls)
{
	const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int k = special_is_const(c, sp[i]);
	cls[i] = (uint8_t)k;
	if (k) mask[sp[i].pair] = 0;
}
// the special pairs of class 1 that are still predictable (not count-sensitive): adopt_auto in aln_device.h
#ifndef PSVR_ADOPT_LANES
#define PSVR_ADOPT_LANES 4
#endif
static const int kAdoptLanes = PSVR_ADOPT_LANES;
__global__ __launch_bounds__(kBlock) void k_adopt_auto(Ctx c, const SpecialPair *sp, long long n, const uint8_t *cls, const uint8_t *mask, const long long *noff,
                                                       int32_t *adopted, long long *adopted_at, unsigned long long *count, const int32_t *host_pairs, const int32_t *host_slots, long long n_host)
{
	// kAdoptLanes lanes per adoption (what lanes share is the copy of 24 header words; the rest is one lane's chain of loads and stores): a
	// wavefront per adoption kept 20 k wavefronts alive for one lane's chain each, 78 us; sixteen adoptions per wavefront are one round, 
	const long long i = blockIdx.x * (long long)(kBlock / kAdoptLanes) + (threadIdx.x / kAdoptLanes);
	const int part = (int)(threadIdx.x % kAdoptLanes);
	if (i >= n) {                                                     // behind the special pairs: the adoptions the host's walk decided (pair, slot)
		if (i - n < n_host) adopt_variant(c, host_pairs[i - n], host_slots[i - n], noff, part, kAdoptLanes);
		return;
	}
	if (!cls[i] || mask[sp[i].pair]) return;
	const int did = adopt_auto(c, sp[i], noff, adopted + i, adopted_at + i, part, kAdoptLanes);
	if (count && did && part == 0) atomicAdd(count, 1ull);           // (statistics only: 17 k wavefronts on one counter are 0.2 ms, tools/atomic_rate_bench.hip)
}
// ---- result hand-over -------------------------------------------------------------------------
// the fixed-size ABI records (psvr_engine_download): one thread per read
__global__ __launch_bounds__(kBlock) void k_materialize(Ctx c, long long R, psvr_read_result_t *out)
{
	const long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (r < R) materialize_read(c, r, out + r);
}
// compact form (psvr_engine_download_compact): per read the number of candidates and of CIGAR words that exist ...
__global__ __launch_bounds__(kBlock) void k_compact_count(Ctx c, long long R, int32_t *cnt_c, int32_t *cnt_w)
{
	const long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (r > R) return;
	int nc = 0, nw = 0;
	if (r < R) {
		const psvr_read_hdr_t &h = c.rh[r];
		nc = h.n_result;
		for (int i = 0; i < nc; ++i) nw += (int)c.cand[h.cand_off + i].n_cigar;
	}
	cnt_c[r] = nc, cnt_w[r] = nw;                     // entry R is the scans' scratch element
}
// ... and, after the two scans, the dense copies: 16 lanes per read (header words, candidate words, CIGAR words)
__global__ __launch_bounds__(kBlock) void k_compact_copy(Ctx c, long long R, const long long *off_c, const long long *off_w, psvr_read_hdr_t *hdr, psvr_cand_t *cands, uint32_t *cig)
{
	const long long r = blockIdx.x * (long long)(kBlock / 16) + (threadIdx.x >> 4);
	const int lane = threadIdx.x & 15;
	if (r >= R) return;
	const psvr_read_hdr_t &h = c.rh[r];
	const long long oc = off_c[r];
	long long ow = off_w[r];
	const int hw = (int)(sizeof(psvr_read_hdr_t) / 4), cwn = (int)(sizeof(psvr_cand_t) / 4);
	const int ho = (int)(offsetof(psvr_read_hdr_t, cand_off) / 4);
	if (lane < hw && lane != ho && lane != ho + 1) ((uint32_t *)(hdr + r))[lane] = ((const uint32_t *)&h)[lane];
	if (lane == 0) hdr[r].cand_off = oc;
	const int n = h.n_result;
	const uint32_t *src = (const uint32_t *)(c.cand + h.cand_off);
	uint32_t *dst = (uint32_t *)(cands + oc);
	const int co = (int)(offsetof(psvr_cand_t, cigar_off) / 4);          // the two words of cigar_off are written by lane 0 below, not copied
	for (int i = lane; i < n * cwn; i += 16) { const int w = i % cwn; if (w != co && w != co + 1) dst[i] = src[i]; }
	for (int k = 0; k < n; ++k) {
		const psvr_cand_t &cd = c.cand[h.cand_off + k];
		const int m = (int)cd.n_cigar;
		for (int i = lane; i < m; i += 16) cig[ow + i] = c.cig.base[cd.cigar_off + i];
		if (lane == 0) cands[oc + k].cigar_off = ow;
		ow += m;
	}
}
__global__ __launch_bounds__(kBlock) void k_run_init(RunInit r)
{
	run_init_slot(r, blockIdx.x * (long long)kBlock + threadIdx.x);
}
__global__ void k_fill_i64(long long *p, long long n, int stride, int off, long long v)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) p[off + i * stride] = v;
}

__global__ void k_iota(int32_t *w, long long at, long long start, long long n)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) w[at + i] = (int32_t)(start + i);
}
// the three per-pair values the host's offset walk needs, for the listed pairs, in one pass: out = [n x i64 | n x i64 | n x i32]
__global__ void k_gather_listed(const long long *a, const long long *b, const int32_t *c, const int32_t *idx, long long n, long long *out)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int32_t j = idx[i];
	out[i] = a[j], out[n + i] = b[j], ((int32_t *)(out + 2 * n))[i] = c[j];
}
__global__ void k_scatter_i32(int32_t *a, const int32_t *idx, const int32_t *val, long long n)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) a[idx[i]] = val[i];
}
__global__ void k_hoff_shadows(Ctx c, long long P, long long n)
{
	long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (j >= n) return;
	for (int k = 0; k < 2; ++k) c.hoff[2 * (P + j) + k] = c.hoff[2 * (long long)c.src[P + j] + k];
}
// total draws of the evaluated slots; a real pair whose total changed since its previous evaluation is count-sensitive
// cnt[1] is set when a slot's rand() or random_r draw counts differ from its previous evaluation's
__device__ __forceinline__ void totals_of(const Ctx &c, long long s, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt, int detect)
{
	int32_t t = c.rcnt[3 * s] + c.rcnt[3 * s + 1] + c.rcnt[3 * s + 2];
	const int32_t h0 = c.hcnt[2 * s], h1 = c.hcnt[2 * s + 1];
	if (t != ctot[s] || h0 != hprev[2 * s] || h1 != hprev[2 * s + 1]) cnt[1] = 1;
	if (detect && s < c.n_pairs && t != ctot[s] && !sens[s]) { sens[s] = 1; slist[atomicAdd(cnt, 1ull)] = (int32_t)s; }
	ctot[s] = t, hprev[2 * s] = h0, hprev[2 * s + 1] = h1;
}
// the same over a list whose length only the device knows yet
__global__ void k_totals_dev(Ctx c, const int32_t *list, const unsigned long long *n_dev, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt)
{
	const unsigned long long n = *n_dev;
	for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) totals_of(c, list[i], ctot, hprev, sens, slist, cnt, 1);
}
__global__ void k_totals(Ctx c, const int32_t *work, long long n, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt, int detect)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= n) return;
	long long s = pair_of(work, i);
	int32_t t = c.rcnt[3 * s] + c.rcnt[3 * s + 1] + c.rcnt[3 * s + 2];
	const int32_t h0 = c.hcnt[2 * s], h1 = c.hcnt[2 * s + 1];
	if (t != ctot[s] || h0 != hprev[2 * s] || h1 != hprev[2 * s + 1]) cnt[1] = 1;
	if (detect && s < c.n_pairs && t != ctot[s] && !sens[s]) { sens[s] = 1; slist[atomicAdd(cnt, 1ull)] = (int32_t)s; }
	ctot[s] = t, hprev[2 * s] = h0, hprev[2 * s + 1] = h1;
}

// exclusive scan of int32 counts into int64 offsets, three small launches: per-tile sums, scan of the tile sums
// (one workgroup), per-tile exclusive scan + tile base.  Up to three arrays of one length go through the same three launches
// (blockIdx.y picks the array): the DP planning scans three, and a launch is ~5 us whatever it does.
static const int kScanTile = 2048;      // elements per 256-thread workgroup
struct ScanSet {
	const int32_t *cnt[3];
	long long *out[3];
	int stride[3], off[3];
	long long base[3];
};
__global__ __launch_bounds__(256) void k_scan_sums(ScanSet S, long long n, long long *tile_sum)
{
	__shared__ long long red[256];
	const int y = blockIdx.y;
	const int32_t *cnt = S.cnt[y];
	const int stride = S.stride[y], off = S.off[y];
	const long long base = blockIdx.x * (long long)kScanTile;
	long long s = 0;
	for (int k = 0; k < kScanTile / 256; ++k) {
		long long i = base + k * 256 + threadIdx.x;
		if (i < n) s += cnt[off + i * stride];
	}
	red[threadIdx.x] = s;
	__syncthreads();
	for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
	if (threadIdx.x == 0) tile_sum[(long long)y * gridDim.x + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void k_scan_tiles(long long *tile_sum_all, long long ntile, ScanSet S)
{
	__shared__ long long part[1024];
	__shared__ long long carry;
	const int tid = threadIdx.x;
	long long *tile_sum = tile_sum_all + (long long)blockIdx.x * ntile;
	if (tid == 0) carry = S.base[blockIdx.x];
	__syncthreads();
	for (long long t0 = 0; t0 < ntile; t0 += 1024) {
		long long i = t0 + tid;
		long long v = i < ntile ? tile_sum[i] : 0;
		part[tid] = v;
		__syncthreads();
		for (int d = 1; d < 1024; d <<= 1) {
			long long t = tid >= d ? part[tid - d] : 0;
			__syncthreads();
			part[tid] += t;
			__syncthreads();
		}
		if (i < ntile) tile_sum[i] = carry + part[tid] - v;
		__syncthreads();
		if (tid == 1023) carry += part[1023];
		__syncthreads();
	}
}
__global__ __launch_bounds__(256) void k_scan_apply(ScanSet S, long long n, const long long *tile_base)
{
	__shared__ long long part[256];
	const int y = blockIdx.y;
	const int32_t *cnt = S.cnt[y];
	long long *out = S.out[y];
	const int stride = S.stride[y], off = S.off[y];
	const long long base = blockIdx.x * (long long)kScanTile;
	const int per = kScanTile / 256;
	long long v[per], s = 0;
	for (int k = 0; k < per; ++k) {
		long long i = base + (long long)threadIdx.x * per + k;
		v[k] = i < n ? cnt[off + i * stride] : 0;
		s += v[k];
	}
	part[threadIdx.x] = s;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) {
		long long t = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
		__syncthreads();
		part[threadIdx.x] += t;
		__syncthreads();
	}
	long long run = tile_base[(long long)y * gridDim.x + blockIdx.x] + part[threadIdx.x] - s;
	for (int k = 0; k < per; ++k) {
		long long i = base + (long long)threadIdx.x * per + k;
		if (i < n) out[off + i * stride] = run;
		run += v[k];
	}
}
__global__ void k_mask_totals(const int32_t *ctot, const uint8_t *mask, long long n, int32_t *out)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) out[i] = mask[i] ? 0 : ctot[i];
}
__global__ void k_scatter_u8(uint8_t *a, const int32_t *idx, long long n, uint8_t v)
{
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) a[idx[i]] = v;
}

// ---- DP planning on the device ---------------------------------------------------------------
struct DpPlanDev {
	const DpDesc *desc; long long n;
	int32_t *qlen, *tlen; long long *q_off, *t_off, *p_off;
	int32_t *qpad, *tpad;          // the lengths rounded up to 16: every sequence starts on a 16-byte boundary of its buffer (k_dp_fetch stores 16 bases at a time)
	int32_t *plen;                 // padded direction-byte bytes for general-kernel problems (0 otherwise), as int32 units of 256 B
	int32_t *bucket;               // bucket id per problem
	unsigned long long *hist;      // [512] counts, [512] cursors, then [16] the longest query per team-kernel class + [1] the scratch top
	int32_t *idx;
	psvr_extz_t *ez;
};
// bucket ids: kind * 13 + class for the wavefront / tiny kernels (< 256); the team kernel's problems are additionally binned by
// query length inside their class (256 + class * 16 + (qlen - 1) / 16), so that the 16 alignments of a wavefront have similar
// numbers of steps per strip
__device__ __forceinline__ int dp_bucket_of(int kind, int cls, int qlen)
{
	if (kind == PSVR_DP_KIND_STRIP) { int qb = (qlen - 1) >> 4; return 256 + cls * 16 + (qb > 15 ? 15 : qb); }
	return (kind < 0 ? 0 : kind) * PSVR_DP_NUM_LDS_CLASSES + cls;
}
__global__ __launch_bounds__(kBlock) void k_dp_lens(DpPlanDev d, int w, int tiny_ok, int team_ok)
{
	__shared__ unsigned int lh[512];
	__shared__ unsigned int lq[PSVR_DP_NUM_LDS_CLASSES];       // longest query per team-kernel class, aggregated per block
	lh[threadIdx.x] = 0, lh[threadIdx.x + 256] = 0;
	if (threadIdx.x < PSVR_DP_NUM_LDS_CLASSES) lq[threadIdx.x] = 0;
	__syncthreads();
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	unsigned int seq_bytes = 0;                                  // query + target bytes of this thread's problem (summed per block into hist[1041])
	if (i == 0) d.qlen[d.n] = d.tlen[d.n] = d.plen[d.n] = d.qpad[d.n] = d.tpad[d.n] = 0;      // the scans run over n + 1 entries
	if (i < d.n) {
		const DpDesc &x = d.desc[i];
		d.qlen[i] = x.qlen, d.tlen[i] = x.tlen;
		d.qpad[i] = (x.qlen + 15) & ~15, d.tpad[i] = (x.tlen + 15) & ~15;
		seq_bytes = (unsigned int)(x.qlen + x.tlen);
		int need;
		int kind = dp_classify(x.qlen, x.tlen, w, true, 0, false, &need, tiny_ok != 0, team_ok != 0);
		int cls = 0;
		while (cls < PSVR_DP_NUM_LDS_CLASSES - 1 && dp_lds_class_bytes(cls) < need) ++cls;
		d.plen[i] = dp_kind_uses_slab(kind) ? (int32_t)((dp_p_bytes(x.qlen, x.tlen, w) + 255) >> 8) : 0;
		const int b = dp_bucket_of(kind, cls, x.qlen);
		d.bucket[i] = b;
		if (kind == PSVR_DP_KIND_STRIP) atomicMax(&lq[cls], (unsigned int)x.qlen);
		atomicAdd(&lh[b], 1u);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < 512; t += 256) if (lh[t]) atomicAdd(d.hist + t, (unsigned long long)lh[t]);
	if (threadIdx.x < PSVR_DP_NUM_LDS_CLASSES && lq[threadIdx.x]) atomicMax(d.hist + 1024 + threadIdx.x, (unsigned long long)lq[threadIdx.x]);
	// (one atomic per workgroup, through LDS: 11 k wavefronts on one address were 0.1 ms of this kernel)
	for (int o = 32; o; o >>= 1) seq_bytes += __shfl_xor(seq_bytes, o);
	__syncthreads();                                              // lh is free again
	if (threadIdx.x == 0) lh[0] = 0;
	__syncthreads();
	if ((threadIdx.x & 63) == 0 && seq_bytes) atomicAdd(&lh[0], seq_bytes);
	__syncthreads();
	if (threadIdx.x == 0 && lh[0]) atomicAdd(d.hist + 1041, (unsigned long long)lh[0]);
}
__global__ __launch_bounds__(kBlock) void k_dp_scatter(DpPlanDev d, const long long *bucket_start)
{
	__shared__ unsigned int lh[512];
	__shared__ unsigned long long lb[512];
	lh[threadIdx.x] = 0, lh[threadIdx.x + 256] = 0;
	__syncthreads();
	long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	int b = 0;
	unsigned int me = 0;
	if (i < d.n) { b = d.bucket[i]; me = atomicAdd(&lh[b], 1u); }
	__syncthreads();
	for (int t = threadIdx.x; t < 512; t += 256) if (lh[t]) lb[t] = atomicAdd(d.hist + 512 + t, (unsigned long long)lh[t]);
	__syncthreads();
	if (i < d.n) {
		d.idx[bucket_start[b] + lb[b] + me] = (int32_t)i;
		d.ez[i].cigar_off = d.q_off[i] + d.t_off[i] + 2 * i;
	}
}
// K5 ref_fetch: unpack the 2-bit reference window / slice the read for every queued DP problem (get_refseq + the reversal of left
// extensions, rr.cpp:920-928).  16 lanes per problem, four problems per wavefront (most are a few bases long: the end-to-end gap fills);
// a lane turns 16 bases -- one 32-bit window of the packed source -- into 16 bytes and stores them in one piece (the sequences start on
// 16-byte boundaries of their buffers): a base at a time was a load, a dozen instructions and a byte store per base.
__device__ __forceinline__ uint32_t expand4(uint32_t x)             // 8 bits = four bases, first in the top bits -> four bytes, first in the lowest
{
	return ((x >> 6) & 3u) | ((x << 4) & 0x300u) | ((x << 14) & 0x30000u) | ((x << 24) & 0x3000000u);
}
__device__ __forceinline__ uint4 expand16(uint32_t w)               // 32 bits = 16 bases, first in the top bits
{
	return make_uint4(expand4(w >> 24), expand4((w >> 16) & 0xffu), expand4((w >> 8) & 0xffu), expand4(w & 0xffu));
}
__device__ __forceinline__ uint32_t rev_bases16(uint32_t w)         // the 16 2-bit groups of w in reverse order
{
	const uint32_t t = __builtin_bitreverse32(w);
	return ((t >> 1) & 0x55555555u) | ((t & 0x55555555u) << 1);
}
__global__ __launch_bounds__(256) void k_dp_fetch(Ctx c, long long begin, long long np, const long long *q_off, const long long *t_off, uint8_t *qbuf, uint8_t *tbuf)
{
	const long long p = blockIdx.x * 16ll + (threadIdx.x >> 4);
	if (p >= np) return;
	const DpDesc &x = c.dp.base[begin + p];
	uint8_t *q = qbuf + q_off[p], *t = tbuf + t_off[p];
	const int lane = threadIdx.x & 15;
	const bool rev = x.type == 0;
	const bool bytes = c.has_n4[x.read] != 0;                       // a read with a lower-case 'n' (code 4): its bases come from the per-base bytes
	const uint64_t *rw = c.rb + ((long long)x.read * 2 + x.strand) * c.wmax;
	const int nq = x.qlen >> 4, nt = x.tlen >> 4;                    // whole 16-base pieces
	for (int k = lane; k < nq + nt; k += 16) {
		const bool is_t = k >= nq;
		const int i0 = 16 * (is_t ? k - nq : k), len = is_t ? x.tlen : x.qlen;
		if (!is_t && bytes) {                                        // query bytes only: the target's chunks belong to other lanes
			const uint8_t *rs = c.bin + ((long long)x.read * 2 + x.strand) * c.lmax;
			for (int i = i0; i < i0 + 16; ++i) q[i] = rs[x.q_st + (rev ? x.qlen - 1 - i : i)];
			continue;
		}
		const uint64_t *src = is_t ? c.idx.ref_seq : rw;
		const uint64_t st = is_t ? (uint64_t)x.ref_st : (uint64_t)x.q_st;
		const uint32_t w = (uint32_t)(window32(src, st + (uint64_t)(rev ? len - 16 - i0 : i0)) >> 32);
		*(uint4 *)((is_t ? t : q) + i0) = expand16(rev ? rev_bases16(w) : w);
	}
	// the tails (< 16 bases each), a base per lane
	{
		const int i = 16 * nq + lane;
		if (i < x.qlen) {
			const int qi = x.q_st + (rev ? x.qlen - 1 - i : i);
			q[i] = bytes ? c.bin[((long long)x.read * 2 + x.strand) * c.lmax + qi] : (uint8_t)base_at(rw, (uint64_t)qi);
		}
		const int j = 16 * nt + lane;
		if (j < x.tlen) t[j] = (uint8_t)base_at(c.idx.ref_seq, (uint64_t)x.ref_st + (rev ? x.tlen - 1 - j : j));
	}
}

// Streams kept per device for the engines of this process.  The first streams of a process each set up a hardware queue (3 - 14 ms a
// stream, and the first copy on one another 8 ms): 25 - 40 ms of an engine's first batch.  psvr_device_warmup() does that on a thread of
// its own while the caller is busy with something else (the command: loading the index); an engine takes its streams from here and
// gives them back when it is destroyed, so the next engine of the process finds them too.
struct StreamPool {
	std::mutex mu;
	std::vector<hipStream_t> idle[64];
	hipStream_t get(int dev)
	{
		{
			std::lock_guard<std::mutex> lk(mu);
			if (dev >= 0 && dev < 64 && !idle[dev].empty()) { hipStream_t s = idle[dev].back(); idle[dev].pop_back(); return s; }
		}
		hipStream_t s = nullptr;
		if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
		return s;
	}
	void put(int dev, hipStream_t s)
	{
		if (!s) return;
		(void)hipStreamSynchronize(s);
		if (dev < 0 || dev >= 64) { (void)hipStreamDestroy(s); return; }
		std::lock_guard<std::mutex> lk(mu);
		idle[dev].push_back(s);
	}
};
static StreamPool &stream_pool() { static StreamPool *p = new StreamPool; return *p; }        // (never destroyed: no HIP calls from a static destructor)

struct GpuBE {
	static constexpr unsigned int kArenaShards = kArenaMaxShards;   // storage arenas: one counter (cache line) per workgroup residue
	hipStream_t stream = nullptr;
	hipError_t last = hipSuccess;
	std::vector<std::pair<std::string, long long>> launches;   // for psvr_engine_stats
	DevBuf plan_bucket, plan_hist, plan_idx, plan_poff, plan_plen, plan_bstart, plan_qpad, plan_tpad, pslab, strip_ws;
	DpParams dpP;
	bool dp_ready = false, dp_lean = false;
	static constexpr long long kTeamMinProblems = 32768;       // below this a round's DP problems go to the wavefront-per-alignment kernels
	static constexpr int kSide = 3;                            // side streams: the DP kernels of a round are independent of each other
	int device = -1;                                           // (for the stream pool)
	hipStream_t side[kSide] = {};
	hipEvent_t ev_fork = nullptr, ev_join[kSide] = {};
	bool side_ok = false;
	bool side_streams()
	{
		if (side_ok) return true;
		if (ev_fork) return false;                                // tried before and failed: stay on the one stream
		if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return false;
		for (int i = 0; i < kSide; ++i)
			if (!(side[i] = stream_pool().get(device)) || hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming) != hipSuccess) return false;
		return side_ok = true;
	}
	// engine_core.h runs the pairing-only repeats on the first side stream: `stream` is switched, so the st_* launches follow
	hipStream_t main_stream = nullptr;
	bool side_begin()
	{
		if (timing || !side_streams()) return false;
		note(hipEventRecord(ev_fork, stream));
		note(hipStreamWaitEvent(side[0], ev_fork, 0));
		main_stream = stream, stream = side[0];
		return true;
	}
	void side_end() { note(hipEventRecord(ev_join[0], side[0])); stream = main_stream; }
	void side_wait() { note(hipStreamWaitEvent(stream, ev_join[0], 0)); }

	// live per-kernel timing with HIP events on the launch stream (bench.py's roofline figure)
	bool timing = false;
	struct Ev { const char *name; hipEvent_t a, b; };
	std::vector<Ev> evs;
	std::vector<std::pair<std::string, std::pair<double, long long>>> timed;   // name -> (ms, launches)
	void t0(const char *name)
	{
		if (!timing) return;
		Ev e{name, nullptr, nullptr};
		note(hipEventCreate(&e.a)), note(hipEventCreate(&e.b));
		note(hipEventRecord(e.a, stream));
		evs.push_back(e);
	}
	void t1() { if (timing && !evs.empty()) note(hipEventRecord(evs.back().b, stream)); }
	void collect_timing()
	{
		for (Ev &e : evs) {
			float ms = 0;
			note(hipEventSynchronize(e.b));
			note(hipEventElapsedTime(&ms, e.a, e.b));
			bool found = false;
			for (auto &t : timed) if (t.first == e.name) { t.second.first += ms, t.second.second++; found = true; break; }
			if (!found) timed.push_back({e.name, {ms, 1}});
			(void)hipEventDestroy(e.a), (void)hipEventDestroy(e.b);
		}
		evs.clear();
	}

	void note(hipError_t e) { if (e != hipSuccess && last == hipSuccess) last = e; }
	void *dalloc(size_t n) { void *p = nullptr; hipError_t e = hipMalloc(&p, n ? n : 16); note(e); return e == hipSuccess ? p : nullptr; }
	void dfree(void *p) { if (p) (void)hipFree(p); }
	void dzero(void *p, size_t n) { note(hipMemsetAsync(p, 0, n, stream)); }
	void dfill(void *p, int byte, size_t n) { note(hipMemsetAsync(p, byte, n, stream)); }
	void d2d(void *dst, const void *src, size_t n) { if (n) note(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, stream)); }
	void scatter_u8_dev(uint8_t *a, const int32_t *d_idx, long long n, uint8_t v)
	{
		if (n > 0) hipLaunchKernelGGL(k_scatter_u8, dim3(grid_for(n)), dim3(kBlock), 0, stream, a, d_idx, n, v);
		note(hipGetLastError());
	}
	// small transfers (counters, lists of a few thousand pairs) go through a pinned staging buffer: a copy to or from pageable
	// memory costs several times the latency
	void *pin = nullptr;
	// layout of the 4 MB buffer: [0, 1.5 MB) staging of the small readbacks; [1.5 MB, 4 MB - 4 KB) the long readback of d2h_early_late, which stays
	// valid (the host reads the variant table in place) until the next one; the last 4 KB st_dp's slot for an upload nobody waits for
	static constexpr size_t kPin = (size_t)4 << 20, kLateAt = (size_t)3 << 19, kPinUse = kLateAt;
	void *pinned() { if (!pin && hipHostMalloc(&pin, kPin, hipHostMallocDefault) != hipSuccess) pin = nullptr; return pin; }
	// Small uploads do not wait: they are staged in a ring of page-locked memory and ride the stream in order.  A slot is reused only
	// after kUp bytes of later uploads, and every round of the engine synchronises the stream several times in between (its readbacks),
	// so the copy out of a slot has long finished when the ring comes round; an upload that would not fit the ring's free span waits.
	void *up_ring = nullptr;
	static constexpr size_t kUp = (size_t)8 << 20;
	size_t up_pos = 0, up_since_sync = 0;
	hipStream_t up_stream = nullptr;                         // the queue the outstanding slots' copies were put on
	// `stream` has just been synchronised: every copy out of the ring that was queued on it has left its slot.  The ring then never has to
	// touch a stream it is not currently running on -- a caller's stream (psvr_engine_run / _rebase) may be gone by the next upload
	void synced() { if (up_since_sync && up_stream == stream) up_since_sync = 0; }
	void h2d(void *d, const void *h, size_t n)
	{
		if (n && n <= kUp / 4 && (up_ring || hipHostMalloc(&up_ring, kUp, hipHostMallocDefault) == hipSuccess)) {
			const size_t need = (n + 255) & ~(size_t)255;
			// the engine's queue and a caller's stream take turns (psvr_engine_run / _rebase): copies still queued on the other one must
			// have left their slots before the ring counts on the current stream's synchronisations
			if (up_since_sync && up_stream != stream) { note(hipStreamSynchronize(up_stream)); up_since_sync = 0; }
			up_stream = stream;
			if (up_since_sync + need > kUp / 2) { note(hipStreamSynchronize(stream)); up_since_sync = 0; }
			if (up_pos + need > kUp) up_pos = 0;
			memcpy((char *)up_ring + up_pos, h, n);
			note(hipMemcpyAsync(d, (char *)up_ring + up_pos, n, hipMemcpyHostToDevice, stream));
			up_pos += need, up_since_sync += need;
			return;
		}
		note(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream)); note(hipStreamSynchronize(stream)); synced();
	}
	// a large upload that the caller waits for later (EngineCore::upload: the host's passes over the batch run beside it)
	void h2d_start(void *d, const void *h, size_t n) { if (n) note(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, stream)); }
	void h2d_wait() { note(hipStreamSynchronize(stream)); synced(); }
	// EngineCore::upload's pass over the batch (k_scan_batch); also the wait for the batch's copies, which are ahead of it on the stream
	DevBuf scan_out;
	bool scan_batch(const char *bases, const long long *off, const psvr_ori_t *ori, long long P, int match, int32_t *list, int *lmax, std::vector<int32_t> &out)
	{
		if (scan_out.ensure(64) != hipSuccess) return false;
		note(hipMemsetAsync(scan_out.p, 0, 8, stream));
		hipLaunchKernelGGL(k_scan_batch, dim3(grid_for(P)), dim3(kBlock), 0, stream, bases, off, ori, P, match, list, (unsigned int *)scan_out.p);
		note(hipGetLastError());
		unsigned int h[2] = {0, 0};
		d2h(h, scan_out.p, 8);
		*lmax = (int)h[1];
		out.resize((size_t)2 * h[0]);
		if (h[0]) d2h(out.data(), list, (size_t)h[0] * 8);
		return last == hipSuccess;
	}
	void d2h(void *h, const void *d, size_t n)
	{
		if (n && n <= kPinUse && pinned()) { note(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, stream)); note(hipStreamSynchronize(stream)); synced(); memcpy(h, pin, n); return; }
		note(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, stream)); note(hipStreamSynchronize(stream)); synced();
	}
	// two small readbacks with one synchronisation
	void d2h2(void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2)
	{
		if (n1 + n2 <= kPinUse && pinned()) {
			note(hipMemcpyAsync(pin, d1, n1, hipMemcpyDeviceToHost, stream));
			note(hipMemcpyAsync((char *)pin + n1, d2, n2, hipMemcpyDeviceToHost, stream));
			note(hipStreamSynchronize(stream)); synced();
			memcpy(h1, pin, n1), memcpy(h2, (char *)pin + n1, n2);
			return;
		}
		d2h(h1, d1, n1), d2h(h2, d2, n2);
	}
	// Two small readbacks the host waits for (an event behind them), and a long one queued behind the event that it does NOT wait for: the
	// caller goes on with what the small ones brought, and the long copy is there after the stream's next synchronisation (d2h_late_done
	// makes sure).  Returns where the long one lands -- a region of the page-locked buffer that nothing else uses, valid until the next call
	// -- or nullptr when it does not fit (nothing was queued for it then: the caller fetches it some other way).
	hipEvent_t ev_early = nullptr;
	bool late_pending = false;
	int32_t *d2h_early_late(void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2, const void *dl, size_t nl)
	{
		if (!h2) n2 = 0;
		if (!pinned() || n1 + n2 > kLateAt || (!ev_early && hipEventCreateWithFlags(&ev_early, hipEventDisableTiming) != hipSuccess)) {
			if (n2) d2h2(h1, d1, n1, h2, d2, n2); else d2h(h1, d1, n1);
			return nullptr;
		}
		char *p = (char *)pin;
		note(hipMemcpyAsync(p, d1, n1, hipMemcpyDeviceToHost, stream));
		if (n2) note(hipMemcpyAsync(p + n1, d2, n2, hipMemcpyDeviceToHost, stream));
		note(hipEventRecord(ev_early, stream));
		const bool fits = nl <= kPin - 4096 - kLateAt;
		if (fits) { note(hipMemcpyAsync(p + kLateAt, dl, nl, hipMemcpyDeviceToHost, stream)); late_pending = true; }
		note(hipEventSynchronize(ev_early));
		memcpy(h1, p, n1);
		if (n2) memcpy(h2, p + n1, n2);
		return fits ? (int32_t *)(p + kLateAt) : nullptr;
	}
	void d2h_late_done() { if (late_pending) { note(hipStreamSynchronize(stream)); synced(); late_pending = false; } }
	// four small readbacks with one synchronisation
	void d2h4(void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2, void *h3, const void *d3, size_t n3, void *h4, const void *d4, size_t n4)
	{
		if (n1 + n2 + n3 + n4 <= kPinUse && pinned()) {
			char *p = (char *)pin;
			note(hipMemcpyAsync(p, d1, n1, hipMemcpyDeviceToHost, stream));
			note(hipMemcpyAsync(p + n1, d2, n2, hipMemcpyDeviceToHost, stream));
			note(hipMemcpyAsync(p + n1 + n2, d3, n3, hipMemcpyDeviceToHost, stream));
			note(hipMemcpyAsync(p + n1 + n2 + n3, d4, n4, hipMemcpyDeviceToHost, stream));
			note(hipStreamSynchronize(stream)); synced();
			memcpy(h1, p, n1), memcpy(h2, p + n1, n2), memcpy(h3, p + n1 + n2, n3), memcpy(h4, p + n1 + n2 + n3, n4);
			return;
		}
		d2h2(h1, d1, n1, h2, d2, n2), d2h2(h3, d3, n3, h4, d4, n4);
	}
	~GpuBE()
	{
		if (pin) (void)hipHostFree(pin);
		if (up_ring) (void)hipHostFree(up_ring);
		if (ev_early) (void)hipEventDestroy(ev_early);
		for (int i = 0; i < kSide; ++i) {
			stream_pool().put(device, side[i]);
			if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
		}
		if (ev_fork) (void)hipEventDestroy(ev_fork);
	}
	void fill_i64(long long *p, long long n, int stride, int off, long long v)
	{
		if (n) hipLaunchKernelGGL(k_fill_i64, dim3(grid_for(n)), dim3(kBlock), 0, stream, p, n, stride, off, v);
	}
	void st_seed(const Ctx &c, const int32_t *w, long long n, int mate)
	{
		if (n > 0) {
			int pitch = c.wmax | 1;
			if ((size_t)pitch * 8 * kBlock > 64 * 1024) pitch = 0;
			t0("k_seed");
			hipLaunchKernelGGL(k_seed, dim3(grid_for(2 * n)), dim3(kBlock), (size_t)pitch * 8 * kBlock, stream, c, w, n, mate, pitch);
			t1();
			// the list k_chain / k_select of this mate run on
			note(mem_list.ensure((size_t)(n + 4) * 4));
			note(hipMemsetAsync(mem_list.p, 0, 4, stream));
			hipLaunchKernelGGL(k_mem_list, dim3(grid_for(n, kBlock * kListItems)), dim3(kBlock), 0, stream, c, w, n, mate, mem_list.as<int32_t>() + 4, (unsigned int *)mem_list.p);
		}
		note(hipGetLastError());
	}
	DevBuf mem_list, left_list;
	static bool chain_small_on() { static const bool v = getenv("PSVR_NO_CHAIN_SMALL") == nullptr; return v; }   // (A/B runs and tests: the generic kernel for every read)
	void st_chain(const Ctx &c, const int32_t *w, long long n, int mate)
	{
		(void)w, (void)mate;
		if (n <= 0) return;
		const int32_t *list = mem_list.as<int32_t>() + 4;
		const unsigned int *cnt = (const unsigned int *)mem_list.p;
		if (chain_small_on() && left_list.ensure((size_t)(n + 4) * 4) == hipSuccess) {
			// the register-resident small case for (nearly) every read, then the generic pair of stages over what it left
			// (PSVR_CHAIN_LEFT_LIST=1: the declined reads go to a list and a launch of their own, as until round 4 -- A/B runs)
			static const bool left_launch = getenv("PSVR_CHAIN_LEFT_LIST") != nullptr;
			if (!left_launch) {
				t0("k_chain_small");
				hipLaunchKernelGGL(k_chain_small, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, list, cnt, (int32_t *)nullptr, (unsigned int *)nullptr);
				t1();
				note(hipGetLastError());
				return;
			}
			note(hipMemsetAsync(left_list.p, 0, 4, stream));
			t0("k_chain_small");
			hipLaunchKernelGGL(k_chain_small, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, list, cnt, left_list.as<int32_t>() + 4, (unsigned int *)left_list.p);
			t1();
			list = left_list.as<int32_t>() + 4, cnt = (const unsigned int *)left_list.p;
			const unsigned g = grid_for(n);
			t0("k_chain_select");
			hipLaunchKernelGGL(k_chain_select, dim3(g < 512u ? g : 512u), dim3(kBlock), 0, stream, c, list, cnt);
			t1();
		} else {
			t0("k_chain_select");
			hipLaunchKernelGGL(k_chain_select, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, list, cnt);
			t1();
		}
		note(hipGetLastError());
	}
	void st_select(const Ctx &, const int32_t *, long long, int) {}      // (done by k_chain_select)
	void st_pair_dev(const Ctx &c, const int32_t *list, const unsigned long long *cnt)
	{
		const long long blocks = c.n_pairs / kBlock + 1;
		hipLaunchKernelGGL(k_pair_dev, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0, stream, c, list, cnt);
		note(hipGetLastError());
	}
	static int str_tsize(const Ctx &c)
	{
		int tsize = 256;                                          // >= 2 x (L - 19) k-mers, power of two
		while (tsize < 2 * c.lmax) tsize <<= 1;
		return tsize;
	}
	DevBuf redo;                                                 // [0] count, [4..] slots whose mate 1 is prepared again
	// below this many slots a round is latency, not throughput: the wavefront-per-read kernel is through sooner (PSVR_PREP_PAIR_MIN: tests)
	static long long prep_pair_min() { static const long long v = [] { const char *e = getenv("PSVR_PREP_PAIR_MIN"); return e ? atoll(e) : 65536ll; }(); return v; }
	// wavefronts per workgroup (4, 2 or 1) whose LDS shares fit the 64 KB a launch may ask for without further ado: reads of up to
	// MAX_READ_LEN = 1600 bases need ~20 KB (prep) / ~50 KB (exact STR count) per wavefront
	static int waves_for_lds(size_t per_wave, size_t fixed)
	{
		int w = kBlock / 64;
		while (w > 1 && fixed + (size_t)w * per_wave > (size_t)64 * 1024) w >>= 1;
		return w;
	}
	void launch_prep_wave(const Ctx &c, const int32_t *w, long long n, int mate, const unsigned int *n_dev, unsigned grid)
	{
		// the screen's hashed set: 32 bits per word; for reads up to ~270 bp 8 x tsize bits keep the expected number of chance
		// collisions at kn / 32 (far below the 16 that would send a read to the exact count); longer reads get the full 32 x tsize
		const int tsize = c.lmax <= 288 ? str_tsize(c) / 4 : str_tsize(c);
		const size_t per_wave = ((size_t)2 * c.lmax + (size_t)c.wmax * 8 + (size_t)tsize * 4 + 15) & ~(size_t)15;
		const int wpb = waves_for_lds(per_wave, 128);
		hipLaunchKernelGGL(k_prep, dim3(grid * (unsigned)(kBlock / 64 / wpb)), dim3(64 * wpb), (size_t)128 + (size_t)wpb * per_wave, stream, c, w, n, mate, tsize, (int)per_wave, n_dev);
	}
	void st_prep(const Ctx &c, const int32_t *w, long long n, int mate)
	{
		if (n <= 0) return;
		dzero(c.str_cnt, 4);
		// reads of up to 288 bases, a round that fills the chip: one lane per pair does both mates (mate 1's N draws follow mate 0's, nothing
		// else connects them), launched with mate 0; the list of reads for the exact STR count then holds both mates'.  When mate 1's turn
		// comes, k_prep_mate1 lists the few mate-1 reads whose draws have moved behind mate 0's tie draws since, and the wavefront-per-read
		// kernel prepares those again.
		if (c.lmax <= 288 && n >= prep_pair_min() && redo.ensure((size_t)(n + 4) * 4) == hipSuccess) {
			t0("k_prep");
			if (mate) {
				note(hipMemsetAsync(redo.p, 0, 4, stream));
				hipLaunchKernelGGL(k_prep_mate1, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, w, n, redo.as<int32_t>() + 4, (unsigned int *)redo.p);
				launch_prep_wave(c, (const int32_t *)(redo.as<int32_t>() + 4), 0, 1, (const unsigned int *)redo.p, 64);
			} else if (c.lmax <= 160) hipLaunchKernelGGL((k_prep_pair<5, 256>), dim3(grid_for(n, 256)), dim3(256), (size_t)4 * 64 * (1 << (PSVR_PREP_BITS - 3)), stream, c, w, n, PSVR_PREP_BITS);
			else hipLaunchKernelGGL((k_prep_pair<9, 128>), dim3(grid_for(n, 128)), dim3(128), (size_t)2 * 64 * 512, stream, c, w, n, 12);
			t1();
			note(hipGetLastError());
			return;
		}
		t0("k_prep");
		launch_prep_wave(c, w, n, mate, nullptr, grid_for(n, kBlock / 64));
		t1();
		note(hipGetLastError());
	}
	void st_str(const Ctx &c, const int32_t *w, long long n, int mate)
	{
		if (n <= 0) return;
		const int tsize = str_tsize(c);
		// per wave: tsize keys (8 B) + tsize counts (4 B); the seed_list staging (<= lmax bytes) reuses the space behind them
		const size_t per_wave = (size_t)tsize * 12 + (((size_t)c.lmax + 15) & ~(size_t)15);
		const int wpb = waves_for_lds(per_wave, 0);
		const size_t lds = (size_t)wpb * per_wave;
		t0("k_str_detect");
		const long long waves = n < 8192 ? n : 8192;                      // each takes list entries in turn
		hipLaunchKernelGGL(k_str_detect, dim3(grid_for(waves, wpb)), dim3(64 * wpb), lds, stream, c, w, n, mate, tsize, (int)per_wave);
		t1();
		note(hipGetLastError());
	}
#define PSVR_STAGE(name, kern, mult, blk)                                                                          \
	void name(const Ctx &c, const int32_t *w, long long n)                                                         \
	{                                                                                                              \
		if (n > 0) { t0(#kern); hipLaunchKernelGGL(kern, dim3(grid_for((mult) * n, blk)), dim3(blk), 0, stream, c, w, n); t1(); } \
		note(hipGetLastError());                                                                                   \
	}
	DevBuf walk_list;
	void st_walk(const Ctx &c, const int32_t *w, long long n)
	{
		if (n <= 0) return;
		note(walk_list.ensure((size_t)(2 * n + 4) * 4));
		int32_t *list = walk_list.as<int32_t>() + 4;
		unsigned int *cnt = (unsigned int *)walk_list.p;
		note(hipMemsetAsync(cnt, 0, 4, stream));
		hipLaunchKernelGGL(k_walk_list, dim3(grid_for(2 * n, kBlock * kListItems)), dim3(kBlock), 0, stream, c, w, n, list, cnt);
		t0("k_walk");
		hipLaunchKernelGGL(k_walk, dim3(grid_for(2 * n)), dim3(kBlock), 0, stream, c, (const int32_t *)list, (const unsigned int *)cnt);
		t1();
		note(hipGetLastError());
	}
	PSVR_STAGE(st_finalize, k_finalize, 2, kBlock)
	PSVR_STAGE(st_pair, k_pair, 1, kBlock)
	PSVR_STAGE(st_finalize_pair, k_finalize_pair, 1, kBlock)
#undef PSVR_STAGE
	DevBuf tmp_idx, tmp_val, tmp_out;
	void fill_iota(int32_t *p, long long n) { if (n) hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kBlock), 0, stream, p, 0ll, 0ll, n); }
	void run_init(const RunInit &r)
	{
		long long n = r.S;
		for (long long k : {r.nsp, (long long)r.n_tops, (long long)r.n_atops, 16ll}) if (k > n) n = k;
		hipLaunchKernelGGL(k_run_init, dim3(grid_for(n)), dim3(kBlock), 0, stream, r);
		note(hipGetLastError());
	}
	void append_iota(int32_t *w, long long at, long long start, long long n) { if (n) hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kBlock), 0, stream, w, at, start, n); }
	void append_list(int32_t *w, long long at, const int32_t *src, long long n) { if (n) hipLaunchKernelGGL(k_copy_i32, dim3(grid_for(n)), dim3(kBlock), 0, stream, w, at, src, n); }
	// one index upload, one kernel, one synchronisation; the indices stay on the device for scatter_listed_i32
	// oa / ob / oc = a / b / cc at the listed indices; x1..x3: up to three more small readbacks (or null) that ride on the same synchronisation
	void gather_listed(const long long *a, const long long *b, const int32_t *cc, const int32_t *idx, long long n, long long *oa, long long *ob, int32_t *oc,
	                   void *x1h, const void *x1d, size_t x1n, void *x2h, const void *x2d, size_t x2n, void *x3h, const void *x3d, size_t x3n)
	{
		void *xh[3] = {x1h, x2h, x3h};
		const void *xd[3] = {x1d, x2d, x3d};
		size_t xn[3] = {x1h ? x1n : 0, x2h ? x2n : 0, x3h ? x3n : 0};
		const size_t xtot = xn[0] + xn[1] + xn[2];
		if (n) {
			note(tmp_idx.ensure(n * 4)), note(tmp_out.ensure(n * 20 + 16));
			h2d(tmp_idx.p, idx, n * 4);
			hipLaunchKernelGGL(k_gather_listed, dim3(grid_for(n)), dim3(kBlock), 0, stream, a, b, cc, (const int32_t *)tmp_idx.p, n, (long long *)tmp_out.p);
		}
		if (!n && !xtot) return;
		if ((size_t)n * 20 + xtot <= kPinUse && pinned()) {
			char *p = (char *)pin;
			if (n) note(hipMemcpyAsync(p, tmp_out.p, n * 20, hipMemcpyDeviceToHost, stream));
			size_t at = (size_t)n * 20;
			for (int k = 0; k < 3; ++k) if (xn[k]) { note(hipMemcpyAsync(p + at, xd[k], xn[k], hipMemcpyDeviceToHost, stream)); at += xn[k]; }
			note(hipStreamSynchronize(stream)); synced();
			if (n) memcpy(oa, p, n * 8), memcpy(ob, p + n * 8, n * 8), memcpy(oc, p + n * 16, n * 4);
			at = (size_t)n * 20;
			for (int k = 0; k < 3; ++k) if (xn[k]) { memcpy(xh[k], p + at, xn[k]); at += xn[k]; }
			return;
		}
		if (n) {
			note(hipMemcpyAsync(oa, tmp_out.p, n * 8, hipMemcpyDeviceToHost, stream));
			note(hipMemcpyAsync(ob, (char *)tmp_out.p + n * 8, n * 8, hipMemcpyDeviceToHost, stream));
			note(hipMemcpyAsync(oc, (char *)tmp_out.p + n * 16, n * 4, hipMemcpyDeviceToHost, stream));
		}
		for (int k = 0; k < 3; ++k) if (xn[k]) note(hipMemcpyAsync(xh[k], xd[k], xn[k], hipMemcpyDeviceToHost, stream));
		note(hipStreamSynchronize(stream)); synced();
	}
	void scatter_listed_i32(int32_t *a, const int32_t *val, long long n)
	{
		if (!n) return;
		note(tmp_val.ensure(n * 4));
		h2d(tmp_val.p, val, n * 4);
		hipLaunchKernelGGL(k_scatter_i32, dim3(grid_for(n)), dim3(kBlock), 0, stream, a, (const int32_t *)tmp_idx.p, (const int32_t *)tmp_val.p, n);
	}
	void st_special_class(const Ctx &c, const SpecialPair *sp, long long n, uint8_t *mask, uint8_t *cls)
	{
		if (n > 0) hipLaunchKernelGGL(k_special_class, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, sp, n, mask, cls);
		note(hipGetLastError());
	}
	// ... and, in the same launch, the adoptions the host's walk decided for the pairs it resolves itself
	void st_adopt_auto(const Ctx &c, const SpecialPair *sp, long long n, const uint8_t *cls, const uint8_t *mask, const long long *noff, int32_t *adopted, long long *adopted_at, unsigned long long *count,
	                   const int32_t *host_pairs, const int32_t *host_slots, long long n_host)
	{
		if (n + n_host <= 0) return;
		if (n_host) {
			note(tmp_idx.ensure(n_host * 4)), note(tmp_val.ensure(n_host * 4));
			h2d(tmp_idx.p, host_pairs, n_host * 4), h2d(tmp_val.p, host_slots, n_host * 4);
		}
		hipLaunchKernelGGL(k_adopt_auto, dim3(grid_for(n + n_host, kBlock / kAdoptLanes)), dim3(kBlock), 0, stream, c, sp, n, cls, mask, noff, adopted, adopted_at, count,
		                   (const int32_t *)tmp_idx.p, (const int32_t *)tmp_val.p, n_host);
		note(hipGetLastError());
	}
	void st_adopt(const Ctx &c, const int32_t *pairs, const int32_t *slots, long long n, const long long *noff)
	{
		if (!n) return;
		note(tmp_idx.ensure(n * 4)), note(tmp_val.ensure(n * 4));
		h2d(tmp_idx.p, pairs, n * 4), h2d(tmp_val.p, slots, n * 4);
		hipLaunchKernelGGL(k_adopt, dim3(grid_for(n, kBlock / 64)), dim3(kBlock), 0, stream, c, (const int32_t *)tmp_idx.p, (const int32_t *)tmp_val.p, n, noff);
		note(hipGetLastError());
	}
	void copy_hoff_to_shadows(const Ctx &c, long long P, long long n) { if (n) hipLaunchKernelGGL(k_hoff_shadows, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, P, n); }
	void st_totals(const Ctx &c, const int32_t *w, long long n, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt, bool detect)
	{
		if (n > 0) hipLaunchKernelGGL(k_totals, dim3(grid_for(n)), dim3(kBlock), 0, stream, c, w, n, ctot, hprev, sens, slist, cnt, detect ? 1 : 0);
		note(hipGetLastError());
	}
	void st_totals_dev(const Ctx &c, const int32_t *list, const unsigned long long *n_dev, long long n_max, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt)
	{
		const long long blocks = n_max / kBlock + 1;
		hipLaunchKernelGGL(k_totals_dev, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(kBlock), 0, stream, c, list, n_dev, ctot, hprev, sens, slist, cnt);
		note(hipGetLastError());
	}
	void st_assemble(const Ctx &c, long long b, long long e)
	{
		if (e > b) { t0("k_assemble"); hipLaunchKernelGGL(k_assemble, dim3(grid_for(e - b)), dim3(kBlock), 0, stream, c, b, e); t1(); }
		note(hipGetLastError());
	}
	DevBuf scan_tmp;
	void st_scan_set(const ScanSet &S, int n_sets, long long n)
	{
		if (n <= 0 || n_sets <= 0) return;
		const long long ntile = (n + kScanTile - 1) / kScanTile;
		note(scan_tmp.ensure((3 * ntile + 1) * 8));
		long long *ts = scan_tmp.as<long long>();
		hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)ntile, (unsigned)n_sets), dim3(256), 0, stream, S, n, ts);
		hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)n_sets), dim3(1024), 0, stream, ts, ntile, S);
		hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)ntile, (unsigned)n_sets), dim3(256), 0, stream, S, n, (const long long *)ts);
		note(hipGetLastError());
	}
	void st_scan(const int32_t *cnt, long long n, int stride, int off, long long base, long long *out)
	{
		ScanSet S = {};
		S.cnt[0] = cnt, S.out[0] = out, S.stride[0] = stride, S.off[0] = off, S.base[0] = base;
		st_scan_set(S, 1, n);
	}
	// the same scan over two / three arrays of one length, in the same three launches
	void st_scan2(const int32_t *c0, int stride0, int off0, long long base0, long long *o0, const int32_t *c1, int stride1, int off1, long long base1, long long *o1, long long n)
	{
		ScanSet S = {};
		S.cnt[0] = c0, S.out[0] = o0, S.stride[0] = stride0, S.off[0] = off0, S.base[0] = base0;
		S.cnt[1] = c1, S.out[1] = o1, S.stride[1] = stride1, S.off[1] = off1, S.base[1] = base1;
		st_scan_set(S, 2, n);
	}
	void st_scan3(const int32_t *c0, long long *o0, const int32_t *c1, long long *o1, const int32_t *c2, long long *o2, long long n)
	{
		ScanSet S = {};
		S.cnt[0] = c0, S.out[0] = o0, S.cnt[1] = c1, S.out[1] = o1, S.cnt[2] = c2, S.out[2] = o2;
		for (int k = 0; k < 3; ++k) S.stride[k] = 1;
		st_scan_set(S, 3, n);
	}
	void st_mask_totals(const int32_t *ctot, const uint8_t *mask, long long n, int32_t *out)
	{
		if (n > 0) hipLaunchKernelGGL(k_mask_totals, dim3(grid_for(n)), dim3(kBlock), 0, stream, ctot, mask, n, out);
	}
	void scatter_u8(uint8_t *a, const int32_t *idx, long long n, uint8_t v)
	{
		if (!n) return;
		note(tmp_idx.ensure(n * 4));
		h2d(tmp_idx.p, idx, n * 4);
		hipLaunchKernelGGL(k_scatter_u8, dim3(grid_for(n)), dim3(kBlock), 0, stream, a, (const int32_t *)tmp_idx.p, n, v);
	}
	void st_dirty(const Ctx &c, const long long *noff, const long long *nhoff, int32_t *out, unsigned long long *cnt, int32_t *outp, unsigned long long *cntp,
	              const uint8_t *has_n, int32_t *out3, unsigned long long *cnt3, long long cap3, int32_t *out4, unsigned long long *cnt4)
	{
		hipLaunchKernelGGL(k_dirty, dim3(grid_for(c.n_pairs, kBlock * kDirtyItems)), dim3(kBlock), 0, stream, c, noff, nhoff, out, cnt, outp, cntp, has_n, out3, cnt3, (unsigned long long)cap3);
		// the tie-only pairs, resolved on the spot (their number stays on the device: a fixed small grid walks the list)
		hipLaunchKernelGGL(k_reselect, dim3(64), dim3(64), 0, stream, c, (const int32_t *)out3, (const unsigned long long *)cnt3, (unsigned long long)cap3, out4, cnt4, outp, cntp);
		note(hipGetLastError());
	}

	// queue -> lens -> offsets -> byte sequences -> size classes -> one DP launch per class
	template <class Core> int st_dp(Core &core)
	{
		const Ctx &c = core.c;
		DpIO &d = core.dp;
		const long long n = d.end - d.begin;
		if (!dp_ready) {
			psvr_ksw_params_t kp;
			memset(&kp, 0, sizeof kp);
			kp.m = 5;
			memcpy(kp.mat, c.mat, 25);
			kp.q = (int8_t)c.par.gap_open, kp.e = (int8_t)c.par.gap_ex, kp.q2 = (int8_t)c.par.gap_open2, kp.e2 = (int8_t)c.par.gap_ex2;
			kp.w = 200, kp.zdrop = c.par.zdrop, kp.end_bonus = -1, kp.flag = 0;   // KSW_ALN_handler::copy_option, rr.cpp:817-827 (bandwith = 200)
			int rc = make_dp_params(&kp, 0, &dpP);
			if (rc) return rc;
			note(dp_allow_big_lds());
			// the engine reads score, mqe and the CIGAR of its pieces, never ez.max / max_q / max_t: when the z-drop rule cannot trigger for these
			// scoring parameters (the reference's defaults), the team kernel runs without the per-diagonal maximum (PSVR_DP_NO_LEAN=1: A/B runs)
			dp_lean = dp_zdrop_inert(dpP) && getenv("PSVR_DP_NO_LEAN") == nullptr;
			dp_ready = true;
		}
		// upper bounds for the sequence buffers: every problem has qlen, tlen < 1600; size from the actual lens
		if (!core.ensure_dp(n, 0, 0, 0)) return set_error(PSVR_ERR_NOMEM, "DP buffers");
		PSVR_HIP(plan_bucket.ensure(n * 4)); PSVR_HIP(plan_idx.ensure(n * 4)); PSVR_HIP(plan_plen.ensure((n + 1) * 4)); PSVR_HIP(plan_poff.ensure((n + 1) * 8));
		PSVR_HIP(plan_qpad.ensure((n + 1) * 4)); PSVR_HIP(plan_tpad.ensure((n + 1) * 4));
		PSVR_HIP(plan_hist.ensure(1056 * 8)); PSVR_HIP(plan_bstart.ensure(512 * 8));
		PSVR_HIP(hipMemsetAsync(plan_hist.p, 0, 1056 * 8, stream));
		DpPlanDev pd;
		pd.desc = c.dp.base + d.begin, pd.n = n, pd.qlen = d.qlen, pd.tlen = d.tlen, pd.q_off = d.q_off, pd.t_off = d.t_off;
		pd.p_off = plan_poff.as<long long>(), pd.plen = plan_plen.as<int32_t>(), pd.bucket = plan_bucket.as<int32_t>();
		pd.hist = plan_hist.as<unsigned long long>(), pd.idx = plan_idx.as<int32_t>(), pd.ez = d.ez;
		pd.qpad = plan_qpad.as<int32_t>(), pd.tpad = plan_tpad.as<int32_t>();
		// a round with few problems (the re-runs after the first) cannot fill the chip at 16 alignments per wavefront: its time would be one
		// wavefront's strips x (qlen + 15) steps; a wavefront per alignment needs qlen + tlen steps
		hipLaunchKernelGGL(k_dp_lens, dim3(grid_for(n)), dim3(kBlock), 0, stream, pd, 200, dp_tiny_ok(dpP, true) ? 1 : 0, n >= kTeamMinProblems ? 1 : 0);
		st_scan3((const int32_t *)pd.qpad, d.q_off, (const int32_t *)pd.tpad, d.t_off, (const int32_t *)pd.plen, pd.p_off, n + 1);
		PSVR_HIP(hipGetLastError());
		unsigned long long hist[512], qmax[18];                // qmax[17]: query + target bytes of the round's problems
		long long tot[3];
		{
			// one synchronisation for all five readbacks, through the pinned staging buffer when it exists
			char stackbuf[512 * 8 + 18 * 8 + 24];
			char *hb = pinned() ? (char *)pin : stackbuf;
			PSVR_HIP(hipMemcpyAsync(hb, plan_hist.p, 512 * 8, hipMemcpyDeviceToHost, stream));
			PSVR_HIP(hipMemcpyAsync(hb + 4096, (char *)plan_hist.p + 1024 * 8, 18 * 8, hipMemcpyDeviceToHost, stream));
			PSVR_HIP(hipMemcpyAsync(hb + 4240, d.q_off + n, 8, hipMemcpyDeviceToHost, stream));
			PSVR_HIP(hipMemcpyAsync(hb + 4248, d.t_off + n, 8, hipMemcpyDeviceToHost, stream));
			PSVR_HIP(hipMemcpyAsync(hb + 4256, pd.p_off + n, 8, hipMemcpyDeviceToHost, stream));
			PSVR_HIP(hipStreamSynchronize(stream));
			synced();
			memcpy(hist, hb, 4096), memcpy(qmax, hb + 4096, 144), memcpy(tot, hb + 4240, 24);
		}
		// NB: the scans ran over n+1 entries, element n of qlen/tlen/plen is scratch: its value only lands in slot n+1 (never read)
		core.stats.dp_seq_bytes += (long long)qmax[17];              // query + target bytes the DP launches of this round read (k_dp_lens sums them)
		if (!core.ensure_dp(n, tot[0], tot[1], tot[0] + tot[1] + 2 * n)) return set_error(PSVR_ERR_NOMEM, "DP sequence buffers");
		PSVR_HIP(pslab.ensure((size_t)(tot[2] << 8) + 256));
		// scratch of the team kernel: a wavefront's slice starts at an offset computed from its class (TeamLaunch::add), sized by the class's
		// longest query -- the same clamp (>= 1) as there
		unsigned long long ws_bytes = 0, team_cnt[PSVR_DP_NUM_LDS_CLASSES];
		for (int cls = 0; cls < PSVR_DP_NUM_LDS_CLASSES; ++cls) {
			team_cnt[cls] = 0;
			for (int qb = 0; qb < 16; ++qb) team_cnt[cls] += hist[256 + cls * 16 + qb];
			if (team_cnt[cls]) { const int lanes = dp_team_lanes(cls + 1); ws_bytes += (team_cnt[cls] * lanes + 63) / 64 * dp_team_ws_bytes(qmax[cls] > 0 ? (int)qmax[cls] : 1, cls + 1, lanes); }
		}
		PSVR_HIP(strip_ws.ensure((size_t)ws_bytes + 256));
		long long bstart[512], acc = 0;
		memset(bstart, 0, sizeof bstart);
		std::vector<Launch3> ls;
		const int kind_order[PSVR_DP_NUM_KINDS - 1] = {0, 14, 13, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 11};
		for (int ko = 0; ko < PSVR_DP_NUM_KINDS - 1; ++ko)
			for (int cls = PSVR_DP_NUM_LDS_CLASSES - 1; cls >= 0; --cls) {
				int b = kind_order[ko] * PSVR_DP_NUM_LDS_CLASSES + cls;
				bstart[b] = acc;
				if (hist[b]) ls.push_back(Launch3{kind_order[ko], dp_lds_class_bytes(cls), acc, (long long)hist[b]});
				acc += (long long)hist[b];
			}
		// the team kernel's classes, longest first; inside a class the query-length bins in descending order
		for (int cls = PSVR_DP_NUM_LDS_CLASSES - 1; cls >= 0; --cls) {
			if (team_cnt[cls]) ls.push_back(Launch3{PSVR_DP_KIND_STRIP, dp_lds_class_bytes(cls), acc, (long long)team_cnt[cls]});
			for (int qb = 15; qb >= 0; --qb) bstart[256 + cls * 16 + qb] = acc, acc += (long long)hist[256 + cls * 16 + qb];
		}
		// no wait for this upload: the staging slot (last 4 KB of the pinned buffer) is not written again before the round's later synchronisations
		if (pinned()) { memcpy((char *)pin + kPin - 4096, bstart, 512 * 8); PSVR_HIP(hipMemcpyAsync(plan_bstart.p, (char *)pin + kPin - 4096, 512 * 8, hipMemcpyHostToDevice, stream)); }
		else h2d(plan_bstart.p, bstart, 512 * 8);
		pd.qlen = d.qlen, pd.tlen = d.tlen, pd.q_off = d.q_off, pd.t_off = d.t_off, pd.ez = d.ez;
		hipLaunchKernelGGL(k_dp_scatter, dim3(grid_for(n)), dim3(kBlock), 0, stream, pd, (const long long *)plan_bstart.p);
		t0("k_dp_fetch");
		hipLaunchKernelGGL(k_dp_fetch, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, c, d.begin, n, (const long long *)d.q_off, (const long long *)d.t_off, d.qbuf, d.tbuf);
		t1();
		PSVR_HIP(hipGetLastError());
		DpBatch B;
		B.qseq = d.qbuf, B.q_off = (const int64_t *)d.q_off, B.qlen = d.qlen;
		B.tseq = d.tbuf, B.t_off = (const int64_t *)d.t_off, B.tlen = d.tlen;
		B.ez = d.ez, B.cigar = d.cig, B.pslab = (uint8_t *)pslab.p, B.p_off = (const int64_t *)plan_poff.p, B.p_unit_shift = 8;   // slab offsets in 256-byte units
		B.ws = (uint8_t *)strip_ws.p, B.ws_cap = ws_bytes;
		B.err = c.err;
		TeamLaunch team;
		size_t n_other = 0;
		for (const Launch3 &L : ls) {
			if (L.kind == PSVR_DP_KIND_STRIP) team.add(dp_class_of(L.lds) + 1, L.first, L.count, (int)qmax[dp_class_of(L.lds)]);
			else ++n_other;
		}
		// The launches of a round work on disjoint problems, and all but the team kernel's are short of wavefronts (the thread-per-alignment
		// kernel: a few hundred that run for ~0.1 ms each): they are dealt to side streams and run beside each other and beside the team
		// kernel.  Not while kernels are being timed.
		// The team kernel's sweep fills every SIMD's registers (three wavefronts of 168 VGPRs), so the others start when it is through and run
		// beside its second launch, a thread per alignment chasing records and direction bytes through memory with the SIMDs idle (beside
		// the sweep they took 1.1 ms for 0.3 ms of work and cost it 0.09 ms; the phase as a whole is the same 1.93 ms either way).
		const bool fan = !timing && n_other + (team.T.n_classes ? 1 : 0) > 1 && side_streams();
		int used = 0, flip = 0;
		bool side2_forked = false;
		B.idx = plan_idx.as<int32_t>();
		if (team.T.n_classes) {
			t0("extd2_team_kernel");
			team.launch_sweep(stream, B, dpP, dp_lean);
			if (!fan) team.launch_finish(stream, B, dpP, dp_lean);
			t1();
			PSVR_HIP(hipGetLastError());
		}
		if (fan) PSVR_HIP(hipEventRecord(ev_fork, stream));
		for (const Launch3 &L : ls) {
			if (L.kind == PSVR_DP_KIND_STRIP) continue;
			hipStream_t s2 = stream;
			if (fan) {
				// a round without the team kernel leaves this stream idle: its first two launches (the largest classes: wavefronts that run
				// for ~0.3 ms each) go to side streams, everything else runs here beside them (the side streams share few hardware queues:
				// dealt round-robin the short kernels queued up behind the long ones, 0.62 ms for what is 0.35 ms of critical path)
				if (!team.T.n_classes && used >= 2) {
					// the short ones alternate between this stream and a third side stream
					if (used & 1) {
						if (!side2_forked) { PSVR_HIP(hipStreamWaitEvent(side[2], ev_fork, 0)); side2_forked = true; }
						s2 = side[2];
					}
					++used;
				} else {
					// beside the team kernel's finish launch: two side streams, the launches dealt to them in turn (largest classes first).  The finish
					// launch's wavefronts hold 128 KB of a CU's LDS (4 KB each for the traceback window) and these kernels want up to 40 KB a
					// block: while it runs they get a block in now and then, whatever the stream -- the two largest run beside it, the short ones
					// behind them two at a time.  (A third side stream's launches waited for a hardware queue and ran one after the other
					// behind everything, profiles/r04p_step_timeline.txt; every launch cut in two halves, one per stream: the large classes'
					// halves ran beside each other as slowly as the whole had, and the short ones queued behind them, +0.07 ms.)
					if (used == 0) { PSVR_HIP(hipStreamWaitEvent(side[0], ev_fork, 0)); PSVR_HIP(hipStreamWaitEvent(side[1], ev_fork, 0)); used = 2; }
					s2 = side[flip], flip ^= 1;
				}
			}
			B.idx = plan_idx.as<int32_t>() + L.first;
			t0(dp_kind_name(L.kind, 0));
			dp_launch_kind(L.kind, 0, (unsigned)L.count, L.lds, s2, B, dpP);
			t1();
			PSVR_HIP(hipGetLastError());
		}
		if (!team.T.n_classes && used > 2) used = side2_forked ? 3 : 2;   // (the side streams that were used)

		for (int k = 0; k < kSide && k < used; ++k) PSVR_HIP(hipEventRecord(ev_join[k], side[k]));
		B.idx = plan_idx.as<int32_t>();
		if (team.T.n_classes && fan) { team.launch_finish(stream, B, dpP, dp_lean); PSVR_HIP(hipGetLastError()); }
		for (int k = 0; k < kSide && k < used; ++k) PSVR_HIP(hipStreamWaitEvent(stream, ev_join[k], 0));
		return PSVR_OK;
	}
	struct Launch3 { int kind, lds; long long first, count; };
};

} // namespace psvr

using namespace psvr;

// ------------------------------------------------------------------------------------------------
// index
// ------------------------------------------------------------------------------------------------
__global__ void k_build_occupancy(const uint64_t *hash, uint32_t *occ, long long nwords)
{
	long long w = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	uint32_t bits = 0;
	const uint64_t *h = hash + w * 32;
	uint64_t prev = h[0];
	for (int b = 0; b < 32; ++b) { uint64_t nx = h[b + 1]; bits |= (uint32_t)(nx != prev) << b; prev = nx; }
	occ[w] = bits;
}

// the Bloom filter over the index's 20-mers (kmer_maybe_present, aln_device.h): a thread per first-level bucket, which holds the 22-mers
// whose first 14 bases are its number; the low 16 bits of an entry are the 22-mer's last 8 bases
__global__ void k_build_bloom(const uint64_t *hash, const uint32_t *kmer, long long nbuckets, unsigned long long *bloom, uint32_t shift)
{
	const long long h = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (h >= nbuckets) return;
	const uint64_t lo = hash[h], hi = hash[h + 1];
	for (uint64_t i = lo; i < hi; ++i) {
		uint64_t w, m;
		bloom_slot(((uint64_t)h << 12) | (uint64_t)(kmer[i] >> 4), shift, w, m);
		atomicOr(bloom + w, (unsigned long long)m);
	}
}

// DevIndex::hitrec: a thread per index entry
__global__ void k_build_hitrec(DevIndex ix, uint64_t n, HitRec *out)
{
	const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
	if (i < n) out[i] = hit_record(ix, i);
}
struct psvr_index {
	int device = 0;
	HostIndex host;          // small tables + strings stay on the host too (SAM formatting)
	DevBuf ref_seq, seq, seqf, pos, posp, hash, off, kmer, chr_end, chr_idx, sv, occ, uid_hint, bloom, hitrec;
	DevIndex dev;
	int64_t bytes = 0;
};

__global__ void k_scatter_counts(const uint32_t *ids, const uint32_t *cnts, long long n, int32_t *dense)
{
	const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i < n) dense[ids[i]] = (int32_t)cnts[i];
}

// `sparse` (optional, when v->hash is null): the non-empty first-level buckets as (id, count) -- the dense prefix-sum table is
// then built in HBM (scatter + device scan) instead of being uploaded: 2 GiB that never exist on the host
// `src_on_device`: the eight arrays of the view are device pointers on ix->device (psvr_index_create_from_device)
static int index_upload(psvr_index *ix, const psvr_index_view_t *v, const uint32_t *sparse_id = nullptr, const uint32_t *sparse_cnt = nullptr, long long n_sparse = 0,
                        bool src_on_device = false)
{
	PSVR_HIP(hipSetDevice(ix->device));
	bool from_dev = src_on_device;
	auto up = [&](DevBuf &b, const void *src, size_t n, size_t pad) -> hipError_t {
		hipError_t e = b.alloc(n + pad);
		if (e != hipSuccess) return e;
		ix->bytes += (int64_t)(n + pad);
		if (pad) { e = hipMemset((char *)b.p + n, 0, pad); if (e != hipSuccess) return e; }
		return n ? hipMemcpy(b.p, src, n, from_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) : hipSuccess;
	};
	PSVR_HIP(up(ix->ref_seq, v->ref_seq, v->n_ref_seq * 8, 544));   // load_index_file pads ref.seq with 536 zero bytes
	PSVR_HIP(up(ix->seq, v->seq, v->n_seq * 8, 16));
	PSVR_HIP(up(ix->seqf, v->seqf, v->n_seqf * 8, 0));
	PSVR_HIP(up(ix->pos, v->pos, v->n_pos * 8, 0));
	PSVR_HIP(up(ix->posp, v->posp, v->n_posp * 8, 0));
	if (v->hash) PSVR_HIP(up(ix->hash, v->hash, v->n_hash * 8, 0));
	else {
		const long long NB = (long long)1 << 28;
		PSVR_HIP(ix->hash.alloc((size_t)(NB + 1) * 8));
		ix->bytes += (NB + 1) * 8;
		DevBuf cnt, ids, cn;
		PSVR_HIP(cnt.alloc((size_t)(NB + 1) * 4)); PSVR_HIP(ids.alloc((size_t)n_sparse * 4 + 4)); PSVR_HIP(cn.alloc((size_t)n_sparse * 4 + 4));
		PSVR_HIP(hipMemset(cnt.p, 0, (size_t)(NB + 1) * 4));
		PSVR_HIP(hipMemcpy(ids.p, sparse_id, (size_t)n_sparse * 4, hipMemcpyHostToDevice)); PSVR_HIP(hipMemcpy(cn.p, sparse_cnt, (size_t)n_sparse * 4, hipMemcpyHostToDevice));
		if (n_sparse) hipLaunchKernelGGL(k_scatter_counts, dim3(grid_for(n_sparse)), dim3(kBlock), 0, nullptr, ids.as<uint32_t>(), cn.as<uint32_t>(), n_sparse, cnt.as<int32_t>());
		GpuBE be;                                            // its three-kernel scan: hash[i] = number of 22-mers in buckets < i
		be.st_scan(cnt.as<int32_t>(), NB + 1, 1, 0, 0ll, ix->hash.as<long long>());
		PSVR_HIP(hipGetLastError());
		PSVR_HIP(hipDeviceSynchronize());
		if (be.last != hipSuccess) return set_error(PSVR_ERR_DEVICE, "index build: %s", hipGetErrorString(be.last));
	}
	PSVR_HIP(up(ix->off, v->off, v->n_off * 8, 0));
	PSVR_HIP(up(ix->kmer, v->kmer, v->n_kmer * 4, 16));
	from_dev = false;                                      // the derived tables below come from the host
	const HostIndex &h = ix->host;
	PSVR_HIP(up(ix->chr_end, h.chr_end_n.data(), h.chr_end_n.size() * 4, 0));
	PSVR_HIP(up(ix->chr_idx, h.chr_search_index.data(), h.chr_search_index.size() * 4, 0));
	PSVR_HIP(up(ix->sv, h.sv.data(), h.sv.size() * sizeof(SvDev), 0));
	DevIndex &d = ix->dev;
	memset(&d, 0, sizeof d);
	d.ref_seq = ix->ref_seq.as<uint64_t>(), d.seq = ix->seq.as<uint64_t>(), d.seqf = ix->seqf.as<uint64_t>(), d.pos = ix->pos.as<uint64_t>();
	d.posp = ix->posp.as<uint64_t>(), d.hash = ix->hash.as<uint64_t>(), d.off = ix->off.as<uint64_t>(), d.kmer = ix->kmer.as<uint32_t>();
	d.n_seqf = v->n_seqf, d.chr_end_n = ix->chr_end.as<uint32_t>(), d.chr_search_index = ix->chr_idx.as<uint32_t>(), d.sv = ix->sv.as<SvDev>();
	d.chr_file_n = h.chr_file_n;
	// occupancy bitmap of the first level, derived on the device from the uploaded table
	const long long nwords = ((long long)1 << 28) / 32;
	PSVR_HIP(ix->occ.alloc(nwords * 4));
	ix->bytes += nwords * 4;
	hipLaunchKernelGGL(k_build_occupancy, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, nullptr, d.hash, ix->occ.as<uint32_t>(), nwords);
	PSVR_HIP(hipGetLastError());
	PSVR_HIP(hipDeviceSynchronize());
	d.occ = ix->occ.as<uint32_t>();
	// the Bloom filter over the 20-mers, sized to ~10 bits per k-mer (a power of two of 64-bit words; PSVR_BLOOM_LOG2=<log2 bytes> fixes
	// the size, 0 leaves the occupancy bitmap as the only filter)
	{
		int lg = 0;
		if (const char *e = getenv("PSVR_BLOOM_LOG2")) lg = atoi(e);
		else { lg = 20; while (lg < 30 && ((uint64_t)8 << lg) < v->n_kmer * 10) ++lg; }
		if (lg >= 16 && lg <= 32) {
			const size_t bytes = (size_t)1 << lg;
			PSVR_HIP(ix->bloom.alloc(bytes));
			ix->bytes += (int64_t)bytes;
			PSVR_HIP(hipMemset(ix->bloom.p, 0, bytes));
			const uint32_t shift = (uint32_t)(64 - (lg - 3));
			const long long NB = (long long)1 << 28;
			hipLaunchKernelGGL(k_build_bloom, dim3((unsigned)((NB + 255) / 256)), dim3(256), 0, nullptr, d.hash, d.kmer, NB, (unsigned long long *)ix->bloom.p, shift);
			PSVR_HIP(hipGetLastError());
			PSVR_HIP(hipDeviceSynchronize());
			d.bloom = ix->bloom.as<uint64_t>(), d.bloom_shift = shift;
		}
	}
	// bracket table for the unipath-of-position search (aln_device.h mem_for_hit): one entry per 1024 positions
	{
		const uint32_t sh = 10;
		std::vector<uint64_t> seqf_host;
		const uint64_t *seqf = v->seqf;
		if (src_on_device) {                               // (U + 1 words: the unipath starts, read back for the table)
			seqf_host.resize((size_t)v->n_seqf);
			PSVR_HIP(hipMemcpy(seqf_host.data(), v->seqf, (size_t)v->n_seqf * 8, hipMemcpyDeviceToHost));
			seqf = seqf_host.data();
		}
		const uint64_t last = v->n_seqf ? seqf[v->n_seqf - 1] : 0;
		std::vector<uint32_t> hint((size_t)(last >> sh) + 3);
		uint64_t u = 0;
		for (size_t b = 0; b < hint.size(); ++b) {
			const uint64_t p = (uint64_t)b << sh;
			while (u + 1 < v->n_seqf && seqf[u + 1] <= p) ++u;
			hint[b] = (uint32_t)u;
		}
		PSVR_HIP(up(ix->uid_hint, hint.data(), hint.size() * 4, 0));
		d.uid_hint = ix->uid_hint.as<uint32_t>(), d.uid_shift = sh;
	}
	// the per-entry records of UNITIG_MEM_search's index-only part (32 B per 22-mer occurrence; PSVR_NO_HITREC=1: derived per hit as before)
	if (v->n_off && !getenv("PSVR_NO_HITREC")) {
		PSVR_HIP(ix->hitrec.alloc((size_t)v->n_off * sizeof(HitRec)));
		ix->bytes += (int64_t)(v->n_off * sizeof(HitRec));
		hipLaunchKernelGGL(k_build_hitrec, dim3((unsigned)((v->n_off + 255) / 256)), dim3(256), 0, nullptr, d, (uint64_t)v->n_off, ix->hitrec.as<HitRec>());
		PSVR_HIP(hipGetLastError());
		PSVR_HIP(hipDeviceSynchronize());
		d.hitrec = ix->hitrec.as<HitRec>();
	}
	return PSVR_OK;
}

extern "C" int psvr_index_create(const psvr_index_view_t *v, int device, psvr_index_t **out)
{
	if (!v || !out || !v->ref_seq || !v->seq || !v->seqf || !v->pos || !v->posp || !v->hash || !v->kmer || !v->off || !v->chr_text)
		return set_error(PSVR_ERR_ARG, "psvr_index_create: null pointer in view");
	if (v->n_hash != ((uint64_t)1 << 28) + 1) return set_error(PSVR_ERR_IO, "unipath_g.hash must hold 4^14+1 entries, got %llu", (unsigned long long)v->n_hash);
	if (psvr_device_count() <= 0) return set_error(PSVR_ERR_DEVICE, "no HIP device visible: the engine has no CPU path");
	psvr_index *ix = new psvr_index;
	ix->device = device;
	std::vector<std::string> names;
	for (int i = 0; i < v->n_header; ++i) names.push_back(v->header_names[i]);
	std::string err;
	if (!ix->host.parse_chr(v->chr_text, names, &err)) { delete ix; return set_error(PSVR_ERR_IO, "%s", err.c_str()); }
	int rc = index_upload(ix, v);
	if (rc) { delete ix; return rc; }
	*out = ix;
	return PSVR_OK;
}

// Multi-GPU, one process per GPU: the arrays arrive in this device's memory through a collective (rank 0 uploads once, an RCCL broadcast
// over xGMI brings them to the others: bench.py) and become the index without touching the host again
extern "C" int psvr_index_create_from_device(const psvr_index_view_t *v, int device, psvr_index_t **out)
{
	if (!v || !out || !v->ref_seq || !v->seq || !v->seqf || !v->pos || !v->posp || !v->hash || !v->kmer || !v->off || !v->chr_text)
		return set_error(PSVR_ERR_ARG, "psvr_index_create_from_device: null pointer in view");
	if (v->n_hash != ((uint64_t)1 << 28) + 1) return set_error(PSVR_ERR_IO, "unipath_g.hash must hold 4^14+1 entries, got %llu", (unsigned long long)v->n_hash);
	if (psvr_device_count() <= 0) return set_error(PSVR_ERR_DEVICE, "no HIP device visible: the engine has no CPU path");
	psvr_index *ix = new psvr_index;
	ix->device = device;
	std::vector<std::string> names;
	for (int i = 0; i < v->n_header; ++i) names.push_back(v->header_names[i]);
	std::string err;
	if (!ix->host.parse_chr(v->chr_text, names, &err)) { delete ix; return set_error(PSVR_ERR_IO, "%s", err.c_str()); }
	int rc = index_upload(ix, v, nullptr, nullptr, 0, true);
	if (rc) { delete ix; return rc; }
	*out = ix;
	return PSVR_OK;
}

// f1, "direct-to-HBM build": anchor FASTA -> the index, resident in HBM, without the nine files in between (the host builder of
// `panSVR index` produces the small arrays; the dense first-level table is expanded on the device)
extern "C" int psvr_index_build(const char *anchors_fa, const char *header_sam, int device, psvr_index_t **out)
{
	if (!anchors_fa || !header_sam || !out) return set_error(PSVR_ERR_ARG, "psvr_index_build: null argument");
	if (psvr_device_count() <= 0) return set_error(PSVR_ERR_DEVICE, "no HIP device visible: the engine has no CPU path");
	IndexBuilder b;
	BuiltIndex bi;
	if (!b.build(anchors_fa, &bi)) return set_error(PSVR_ERR_IO, "index build: %s", b.error().c_str());
	std::vector<std::string> names;
	if (!HostIndex::header_names_of(header_sam, &names)) return set_error(PSVR_ERR_IO, "cannot read header %s", header_sam);
	psvr_index *ix = new psvr_index;
	ix->device = device;
	std::string err;
	if (!ix->host.parse_chr(bi.chr_text, names, &err)) { delete ix; return set_error(PSVR_ERR_IO, "%s", err.c_str()); }
	bi.ref_seq.resize(bi.ref_seq.size());
	psvr_index_view_t v;
	memset(&v, 0, sizeof v);
	v.ref_seq = bi.ref_seq.data(), v.n_ref_seq = bi.ref_seq.size(), v.seq = bi.seqb.data(), v.n_seq = bi.seqb.size();
	v.seqf = bi.seqf.data(), v.n_seqf = bi.seqf.size(), v.pos = bi.pos.data(), v.n_pos = bi.pos.size(), v.posp = bi.posp.data(), v.n_posp = bi.posp.size();
	v.hash = nullptr, v.n_hash = ((uint64_t)1 << 28) + 1, v.kmer = bi.kmer.data(), v.n_kmer = bi.kmer.size(), v.off = bi.off.data(), v.n_off = bi.off.size();
	std::vector<uint32_t> sid(bi.hash_sparse_id.size()), scn(bi.hash_sparse_cnt.size());
	for (size_t i = 0; i < sid.size(); ++i) sid[i] = (uint32_t)bi.hash_sparse_id[i], scn[i] = (uint32_t)bi.hash_sparse_cnt[i];
	int rc = index_upload(ix, &v, sid.data(), scn.data(), (long long)sid.size());
	if (rc) { delete ix; return rc; }
	*out = ix;
	return PSVR_OK;
}

extern "C" int psvr_index_load(const char *dir, const char *header_sam, int device, psvr_index_t **out)
{
	if (!dir || !header_sam || !out) return set_error(PSVR_ERR_ARG, "psvr_index_load: null argument");
	if (psvr_device_count() <= 0) return set_error(PSVR_ERR_DEVICE, "no HIP device visible: the engine has no CPU path");
	psvr_index *ix = new psvr_index;
	ix->device = device;
	std::string err;
	ix->host.defer_dense = true;
	if (!ix->host.load_dir(dir, header_sam, &err)) { delete ix; return set_error(PSVR_ERR_IO, "%s", err.c_str()); }
	HostIndex &h = ix->host;
	psvr_index_view_t v;
	memset(&v, 0, sizeof v);
	v.ref_seq = h.ref_seq.data(), v.n_ref_seq = h.ref_seq.size(), v.seq = h.seq.data(), v.n_seq = h.seq.size();
	v.seqf = h.seqf.data(), v.n_seqf = h.seqf.size(), v.pos = h.pos.data(), v.n_pos = h.pos.size(), v.posp = h.posp.data(), v.n_posp = h.posp.size();
	v.hash = h.hash.data(), v.n_hash = h.hash.size(), v.kmer = h.kmer.data(), v.n_kmer = h.kmer.size(), v.off = h.off.data(), v.n_off = h.off.size();
	int rc;
	if (!h.sparse_pairs.empty()) {                 // fixture / `panSVR index --sparse-hash` form: expand on the device
		const size_t ns = h.sparse_pairs.size() / 2;
		std::vector<uint32_t> sid(ns), scn(ns);
		for (size_t i = 0; i < ns; ++i) sid[i] = h.sparse_pairs[2 * i], scn[i] = h.sparse_pairs[2 * i + 1];
		{   // the device scatters counts[id]: a foreign or damaged file must not reach it
			unsigned long long tot = 0;
			bool ok = true;
			for (size_t i = 0; i < ns && ok; ++i) { ok = sid[i] < (1u << 28) && (i == 0 || sid[i] > sid[i - 1]); tot += scn[i]; }
			if (!ok || tot != (unsigned long long)h.kmer.size()) { delete ix; return set_error(PSVR_ERR_IO, "unipath_g.hash.sparse: bucket ids must be < 4^14 and ascending, counts must add up to the %zu entries of unipath_g.kmer", h.kmer.size()); }
		}
		v.hash = nullptr, v.n_hash = ((uint64_t)1 << 28) + 1;
		rc = index_upload(ix, &v, sid.data(), scn.data(), (long long)ns);
		std::vector<uint32_t>().swap(h.sparse_pairs);
	} else rc = index_upload(ix, &v);
	std::vector<uint64_t>().swap(h.hash);          // the 2 GiB table now lives in HBM only
	if (rc) { delete ix; return rc; }
	*out = ix;
	return PSVR_OK;
}

// every buffer of `src` copied device to device (hipMemcpyPeer: xGMI between GPUs of one node), then the derived pointers
extern "C" int psvr_index_clone(const psvr_index_t *src, int device, psvr_index_t **out)
{
	if (!src || !out) return set_error(PSVR_ERR_ARG, "psvr_index_clone: null argument");
	psvr_index *ix = new psvr_index;
	ix->device = device;
	ix->host = src->host;
	hipError_t he = hipSetDevice(device);
	auto cp = [&](DevBuf &d, const DevBuf &s0) {
		if (he != hipSuccess || !s0.p) return;
		he = d.alloc(s0.bytes);
		if (he == hipSuccess) he = hipMemcpyPeer(d.p, device, s0.p, src->device, s0.bytes);
		ix->bytes += (int64_t)s0.bytes;
	};
	cp(ix->ref_seq, src->ref_seq), cp(ix->seq, src->seq), cp(ix->seqf, src->seqf), cp(ix->pos, src->pos), cp(ix->posp, src->posp), cp(ix->hash, src->hash);
	cp(ix->off, src->off), cp(ix->kmer, src->kmer), cp(ix->chr_end, src->chr_end), cp(ix->chr_idx, src->chr_idx), cp(ix->sv, src->sv), cp(ix->occ, src->occ), cp(ix->uid_hint, src->uid_hint), cp(ix->bloom, src->bloom), cp(ix->hitrec, src->hitrec);
	if (he == hipSuccess) he = hipDeviceSynchronize();
	if (he != hipSuccess) { delete ix; return set_error(PSVR_ERR_DEVICE, "psvr_index_clone: %s", hipGetErrorString(he)); }
	DevIndex &d = ix->dev;
	d = src->dev;
	d.ref_seq = ix->ref_seq.as<uint64_t>(), d.seq = ix->seq.as<uint64_t>(), d.seqf = ix->seqf.as<uint64_t>(), d.pos = ix->pos.as<uint64_t>();
	d.posp = ix->posp.as<uint64_t>(), d.hash = ix->hash.as<uint64_t>(), d.off = ix->off.as<uint64_t>(), d.kmer = ix->kmer.as<uint32_t>();
	d.chr_end_n = ix->chr_end.as<uint32_t>(), d.chr_search_index = ix->chr_idx.as<uint32_t>(), d.sv = ix->sv.as<SvDev>();
	d.occ = ix->occ.as<uint32_t>(), d.uid_hint = ix->uid_hint.as<uint32_t>();
	if (src->bloom.p) d.bloom = ix->bloom.as<uint64_t>();
	if (src->hitrec.p) d.hitrec = ix->hitrec.as<HitRec>();
	*out = ix;
	return PSVR_OK;
}

extern "C" void psvr_index_destroy(psvr_index_t *ix) { delete ix; }
extern "C" int64_t psvr_index_device_bytes(const psvr_index_t *ix) { return ix ? ix->bytes : 0; }
extern "C" int32_t psvr_index_n_anchor(const psvr_index_t *ix) { return ix ? ix->host.chr_file_n : 0; }
extern "C" const char *psvr_index_sv_print_string(const psvr_index_t *ix, int32_t sv)
{
	return ix && sv >= 0 && sv < (int)ix->host.svh.size() ? ix->host.svh[sv].vcf_print_string.c_str() : nullptr;
}
extern "C" const char *psvr_index_sv_vcf_id(const psvr_index_t *ix, int32_t sv)
{
	return ix && sv >= 0 && sv < (int)ix->host.svh.size() ? ix->host.svh[sv].vcf_id.c_str() : nullptr;
}

// ------------------------------------------------------------------------------------------------
// seam B3: the seed loop's two look-ups on their own (the device functions are the ones k_seed runs)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_b3_search_kmer(DevIndex ix, long long n, const uint64_t *kmers, long long *range, uint8_t *found)
{
	const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint64_t kmer = kmers[i] & 0xffffffffffull;
	uint64_t first = 0;
	const uint32_t nh = kmer_maybe_present(ix, kmer) ? probe_kmer(ix, kmer, first) : 0;
	found[i] = nh != 0;
	range[2 * i] = nh ? (long long)first : 0, range[2 * i + 1] = nh ? (long long)(first + nh - 1) : -1;
}
__global__ __launch_bounds__(kBlock) void k_b3_mem(DevIndex ix, long long n, const uint64_t *kmer_index, const uint64_t *read_bits, const long long *word_off, const uint32_t *read_off,
                                                   const uint32_t *read_len, psvr_vertex_mem_t *out)
{
	const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
	if (i >= n) return;
	VMem m;
	const uint32_t ri = mem_for_hit(ix, kmer_index[i], read_bits + word_off[i], read_off[i], (int)read_len[i], m);
	psvr_vertex_mem_t &o = out[i];
	o.uid = m.uid, o.seed_id = 0, o.read_pos = m.read_pos, o.uni_pos_off = m.uni_pos_off, o.length = m.length, o.pos_n = m.pos_n, o.right_i = ri;
}

extern "C" int psvr_seed_search_kmer_batch(const psvr_index_t *ix, int64_t n, const uint64_t *kmers, int64_t *range, uint8_t *found)
{
	if (!ix || n < 0 || (n && (!kmers || !range || !found))) return set_error(PSVR_ERR_ARG, "psvr_seed_search_kmer_batch: bad argument");
	if (n == 0) return PSVR_OK;
	PSVR_HIP(hipSetDevice(ix->device));
	DevBuf dk, dr, df;
	PSVR_HIP(dk.alloc((size_t)n * 8)); PSVR_HIP(dr.alloc((size_t)n * 16)); PSVR_HIP(df.alloc((size_t)n));
	PSVR_HIP(hipMemcpy(dk.p, kmers, (size_t)n * 8, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_b3_search_kmer, dim3(grid_for(n)), dim3(kBlock), 0, nullptr, ix->dev, (long long)n, dk.as<uint64_t>(), dr.as<long long>(), df.as<uint8_t>());
	PSVR_HIP(hipGetLastError());
	PSVR_HIP(hipMemcpy(range, dr.p, (size_t)n * 16, hipMemcpyDeviceToHost));
	PSVR_HIP(hipMemcpy(found, df.p, (size_t)n, hipMemcpyDeviceToHost));
	return PSVR_OK;
}

extern "C" int psvr_seed_mem_batch(const psvr_index_t *ix, int64_t n, const uint64_t *kmer_index, const uint64_t *read_bits, int64_t n_words, const int64_t *word_off,
                                   const uint32_t *read_off, const uint32_t *read_len, psvr_vertex_mem_t *out)
{
	if (!ix || n < 0 || n_words < 0 || (n && (!kmer_index || !read_bits || !word_off || !read_off || !read_len || !out))) return set_error(PSVR_ERR_ARG, "psvr_seed_mem_batch: bad argument");
	if (n == 0) return PSVR_OK;
	const uint64_t n_index = ix->off.bytes / 8;
	for (int64_t i = 0; i < n; ++i) {                      // the kernel indexes with these: check them here, on the host
		const int64_t words = ((int64_t)read_len[i] + 31) / 32 + 1;
		if (kmer_index[i] >= n_index || word_off[i] < 0 || word_off[i] + words > n_words || read_len[i] < (uint32_t)kLenKmer || read_off[i] + (uint32_t)kLenKmer > read_len[i])
			return set_error(PSVR_ERR_ARG, "psvr_seed_mem_batch: item %lld is out of range (index entry, word window or read offset)", (long long)i);
	}
	PSVR_HIP(hipSetDevice(ix->device));
	DevBuf dk, db, dw, dro, drl, dout;
	PSVR_HIP(dk.alloc((size_t)n * 8)); PSVR_HIP(db.alloc((size_t)(n_words + 2) * 8)); PSVR_HIP(dw.alloc((size_t)n * 8)); PSVR_HIP(dro.alloc((size_t)n * 4)); PSVR_HIP(drl.alloc((size_t)n * 4));
	PSVR_HIP(dout.alloc((size_t)n * sizeof(psvr_vertex_mem_t)));
	PSVR_HIP(hipMemset((char *)db.p + (size_t)n_words * 8, 0, 16));
	PSVR_HIP(hipMemcpy(dk.p, kmer_index, (size_t)n * 8, hipMemcpyHostToDevice)); PSVR_HIP(hipMemcpy(db.p, read_bits, (size_t)n_words * 8, hipMemcpyHostToDevice));
	PSVR_HIP(hipMemcpy(dw.p, word_off, (size_t)n * 8, hipMemcpyHostToDevice)); PSVR_HIP(hipMemcpy(dro.p, read_off, (size_t)n * 4, hipMemcpyHostToDevice));
	PSVR_HIP(hipMemcpy(drl.p, read_len, (size_t)n * 4, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_b3_mem, dim3(grid_for(n)), dim3(kBlock), 0, nullptr, ix->dev, (long long)n, dk.as<uint64_t>(), db.as<uint64_t>(), dw.as<long long>(), dro.as<uint32_t>(), drl.as<uint32_t>(),
	                   dout.as<psvr_vertex_mem_t>());
	PSVR_HIP(hipGetLastError());
	PSVR_HIP(hipMemcpy(out, dout.p, (size_t)n * sizeof(psvr_vertex_mem_t), hipMemcpyDeviceToHost));
	return PSVR_OK;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
extern "C" int psvr_device_warmup(int device, int n_streams)
{
	if (device < 0 || device >= 64 || n_streams < 1) return set_error(PSVR_ERR_ARG, "psvr_device_warmup: bad argument");
	if (n_streams > 16) n_streams = 16;
	std::thread([device, n_streams]() {
		if (hipSetDevice(device) != hipSuccess) return;
		std::vector<hipStream_t> got;
		void *h = nullptr, *d = nullptr;
		const bool buf = hipHostMalloc(&h, 4096, hipHostMallocDefault) == hipSuccess && hipMalloc(&d, 4096) == hipSuccess;
		if (buf) memset(h, 0, 4096);
		for (int i = 0; i < n_streams; ++i) {
			hipStream_t s = nullptr;
			if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
			// a copy each way and a fill: the queue, the copy engines' paths and the fill kernel are set up by their first use
			if (buf) (void)hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, s), (void)hipMemsetAsync(d, 0, 4096, s), (void)hipMemcpyAsync(h, d, 4096, hipMemcpyDeviceToHost, s);
			(void)hipStreamSynchronize(s);
			got.push_back(s);
		}
		if (h) (void)hipHostFree(h);
		if (d) (void)hipFree(d);
		(void)hipGetLastError();
		StreamPool &P = stream_pool();
		std::lock_guard<std::mutex> lk(P.mu);
		for (hipStream_t s : got) P.idle[device].push_back(s);
	}).detach();
	return PSVR_OK;
}

struct psvr_engine {
	const psvr_index *ix;
	GpuBE be;
	EngineCore<GpuBE> core;
	bool committed = true;
	// hand-over buffers: the materialised ABI records, or the compact form (valid until the next upload / run / rebase)
	DevBuf full_out, cmp_cnt_c, cmp_cnt_w, cmp_off_c, cmp_off_w, cmp_hdr, cmp_cand, cmp_cig;
	bool compact_valid = false;
	long long compact_nc = 0, compact_nw = 0;
	// the engine's own queue (non-blocking: engines of one process, on one device or several, run beside each other); a caller's
	// stream, when psvr_engine_run / _rebase is given one, takes its place for that call
	hipStream_t own = nullptr;
	explicit psvr_engine(const psvr_index *i) : ix(i), core(be) {}
};

extern "C" void psvr_aln_params_default(psvr_aln_params_t *p) { if (p) aln_params_default(p); }

extern "C" int psvr_engine_create(const psvr_index_t *ix, const psvr_aln_params_t *par, psvr_engine_t **out)
{
	if (!ix || !par || !out) return set_error(PSVR_ERR_ARG, "psvr_engine_create: null argument");
	PSVR_HIP(hipSetDevice(ix->device));
	psvr_engine *e = new psvr_engine(ix);
	e->own = stream_pool().get(ix->device);                  // (nullptr: the default queue then)
	e->be.stream = e->own, e->be.device = ix->device;
	e->core.init(ix->dev, *par);
	*out = e;
	return PSVR_OK;
}

extern "C" void psvr_engine_destroy(psvr_engine_t *e)
{
	if (!e) return;
	(void)hipSetDevice(e->ix->device);
	e->core.free_all();
	const int dev = e->ix->device;
	hipStream_t own = e->own;
	delete e;                                                // (its backend gives the side streams back)
	stream_pool().put(dev, own);
}

static int engine_status(psvr_engine *e, int rc)
{
	if (e->be.last != hipSuccess) { hipError_t x = e->be.last; e->be.last = hipSuccess; return set_error(PSVR_ERR_DEVICE, "HIP error: %s", hipGetErrorString(x)); }
	if (rc) return set_error(rc, "%s", e->core.err.empty() ? psvr_last_error() : e->core.err.c_str());
	return PSVR_OK;
}

extern "C" int psvr_engine_upload(psvr_engine_t *e, int64_t n_pairs, const char *bases, const int64_t *base_off, const psvr_ori_t *ori)
{
	if (!e || n_pairs < 0 || (n_pairs && (!bases || !base_off || !ori))) return set_error(PSVR_ERR_ARG, "psvr_engine_upload: bad argument");
	PSVR_HIP(hipSetDevice(e->ix->device));
	e->be.stream = e->own;
	if (!e->committed) { e->core.commit(); e->committed = true; }
	e->compact_valid = false;
	int rc = e->core.upload(n_pairs, bases, base_off, ori);
	return engine_status(e, rc);
}

extern "C" int psvr_engine_run(psvr_engine_t *e, int trace, void *stream)
{
	if (!e) return set_error(PSVR_ERR_ARG, "psvr_engine_run: null engine");
	PSVR_HIP(hipSetDevice(e->ix->device));
	e->be.stream = stream ? (hipStream_t)stream : e->own;
	e->be.timing = (trace & 4) != 0;
	e->be.timed.clear();
	e->compact_valid = false;
	int rc = e->core.run(trace & 1, (trace & 2) != 0);
	hipError_t s = hipStreamSynchronize(e->be.stream);
	if (s != hipSuccess) e->be.note(s);
	e->be.synced();             // (a caller's stream is not looked at again after this call)
	e->be.collect_timing();
	e->committed = false;       // the rand streams advance when the next batch is uploaded (or a stream position is set)
	return engine_status(e, rc);
}

extern "C" int psvr_engine_set_stream_pos(psvr_engine_t *e, const int64_t pos[3])
{
	if (!e || !pos) return set_error(PSVR_ERR_ARG, "psvr_engine_set_stream_pos: null argument");
	PSVR_HIP(hipSetDevice(e->ix->device));
	// (what commit() would do -- read where the last run ended and move there -- minus the readbacks: the position is given)
	e->core.have_run = false, e->committed = true;
	e->core.grand_pos = pos[0], e->core.hrand_pos[0] = pos[1], e->core.hrand_pos[1] = pos[2];
	return PSVR_OK;
}

extern "C" int psvr_engine_stream_end(psvr_engine_t *e, int64_t end[3])
{
	if (!e || !end) return set_error(PSVR_ERR_ARG, "psvr_engine_stream_end: null argument");
	PSVR_HIP(hipSetDevice(e->ix->device));
	long long t[3];
	e->core.stream_end(t);
	end[0] = t[0], end[1] = t[1], end[2] = t[2];
	return engine_status(e, PSVR_OK);
}

extern "C" int psvr_engine_rebase(psvr_engine_t *e, const int64_t pos[3], void *stream)
{
	if (!e || !pos) return set_error(PSVR_ERR_ARG, "psvr_engine_rebase: null argument");
	PSVR_HIP(hipSetDevice(e->ix->device));
	e->be.stream = stream ? (hipStream_t)stream : e->own;
	e->compact_valid = false;
	int rc = e->core.rebase(pos[0], pos[1], pos[2], e->core.c.trace, false);
	hipError_t s = hipStreamSynchronize(e->be.stream);
	if (s != hipSuccess) e->be.note(s);
	e->be.synced();
	return engine_status(e, rc);
}

extern "C" int psvr_engine_download(psvr_engine_t *e, psvr_read_result_t *reads, psvr_pair_result_t *pairs, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_used)
{
	if (!e) return set_error(PSVR_ERR_ARG, "psvr_engine_download: null engine");
	PSVR_HIP(hipSetDevice(e->ix->device));
	auto &c = e->core;
	if (c.P == 0) { if (cigar_used) *cigar_used = 0; return PSVR_OK; }
	unsigned long long top = 0;
	PSVR_HIP(hipMemcpy(&top, c.c.cig.top, 8, hipMemcpyDeviceToHost));
	if (cigar_used) *cigar_used = (int64_t)top;
	if ((int64_t)top > cigar_cap) return set_error(PSVR_ERR_OVERFLOW, "cigar arena too small: need %llu uint32, have %lld", top, (long long)cigar_cap);
	if (reads) {
		// the engine keeps compact headers + a candidate list; the fixed 12-slot records are built here, on request
		PSVR_HIP(e->full_out.ensure((size_t)c.R * sizeof(psvr_read_result_t)));
		hipLaunchKernelGGL(k_materialize, dim3(grid_for(c.R)), dim3(kBlock), 0, e->own, c.c, c.R, e->full_out.as<psvr_read_result_t>());
		PSVR_HIP(hipGetLastError());
		PSVR_HIP(hipMemcpyAsync(reads, e->full_out.p, c.R * sizeof(psvr_read_result_t), hipMemcpyDeviceToHost, e->own));
		PSVR_HIP(hipStreamSynchronize(e->own));
	}
	if (pairs) PSVR_HIP(hipMemcpy(pairs, c.c.pres, c.P * sizeof(psvr_pair_result_t), hipMemcpyDeviceToHost));
	if (cigar && top) PSVR_HIP(hipMemcpy(cigar, c.c.cig.base, top * 4, hipMemcpyDeviceToHost));
	return PSVR_OK;
}

extern "C" int psvr_engine_download_compact(psvr_engine_t *e, psvr_read_hdr_t *hdr, psvr_pair_result_t *pairs, psvr_cand_t *cands, int64_t cand_cap, int64_t *cand_used,
                                            uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_used)
{
	if (!e) return set_error(PSVR_ERR_ARG, "psvr_engine_download_compact: null engine");
	PSVR_HIP(hipSetDevice(e->ix->device));
	auto &c = e->core;
	if (c.P == 0) { if (cand_used) *cand_used = 0; if (cigar_used) *cigar_used = 0; return PSVR_OK; }
	const long long R = c.R;
	if (!e->compact_valid) {
		e->be.stream = e->own;
		PSVR_HIP(e->cmp_cnt_c.ensure((size_t)(R + 1) * 4)); PSVR_HIP(e->cmp_cnt_w.ensure((size_t)(R + 1) * 4));
		PSVR_HIP(e->cmp_off_c.ensure((size_t)(R + 2) * 8)); PSVR_HIP(e->cmp_off_w.ensure((size_t)(R + 2) * 8));
		hipLaunchKernelGGL(k_compact_count, dim3(grid_for(R + 1)), dim3(kBlock), 0, e->own, c.c, R, e->cmp_cnt_c.as<int32_t>(), e->cmp_cnt_w.as<int32_t>());
		e->be.st_scan(e->cmp_cnt_c.as<int32_t>(), R + 1, 1, 0, 0ll, e->cmp_off_c.as<long long>());
		e->be.st_scan(e->cmp_cnt_w.as<int32_t>(), R + 1, 1, 0, 0ll, e->cmp_off_w.as<long long>());
		PSVR_HIP(hipGetLastError());
		long long tot[2] = {0, 0};
		PSVR_HIP(hipMemcpyAsync(&tot[0], e->cmp_off_c.as<long long>() + R, 8, hipMemcpyDeviceToHost, e->own));
		PSVR_HIP(hipMemcpyAsync(&tot[1], e->cmp_off_w.as<long long>() + R, 8, hipMemcpyDeviceToHost, e->own));
		PSVR_HIP(hipStreamSynchronize(e->own));
		PSVR_HIP(e->cmp_hdr.ensure((size_t)R * sizeof(psvr_read_hdr_t)));
		PSVR_HIP(e->cmp_cand.ensure((size_t)(tot[0] + 1) * sizeof(psvr_cand_t)));
		PSVR_HIP(e->cmp_cig.ensure((size_t)(tot[1] + 1) * 4));
		hipLaunchKernelGGL(k_compact_copy, dim3(grid_for(R, kBlock / 16)), dim3(kBlock), 0, e->own, c.c, R, (const long long *)e->cmp_off_c.p, (const long long *)e->cmp_off_w.p,
		                   e->cmp_hdr.as<psvr_read_hdr_t>(), e->cmp_cand.as<psvr_cand_t>(), e->cmp_cig.as<uint32_t>());
		PSVR_HIP(hipGetLastError());
		e->compact_nc = tot[0], e->compact_nw = tot[1], e->compact_valid = true;
	}
	if (cand_used) *cand_used = e->compact_nc;
	if (cigar_used) *cigar_used = e->compact_nw;
	if ((cands && e->compact_nc > cand_cap) || (cigar && e->compact_nw > cigar_cap))
		return set_error(PSVR_ERR_OVERFLOW, "compact download: need %lld candidates / %lld cigar words", e->compact_nc, e->compact_nw);
	// (hipMemcpyDefault: the destinations may be host memory or memory of this device -- the ordered gather of a one-process-per-GPU host
	// hands the blocks on over RCCL without a trip through the host)
	if (hdr) PSVR_HIP(hipMemcpyAsync(hdr, e->cmp_hdr.p, (size_t)R * sizeof(psvr_read_hdr_t), hipMemcpyDefault, e->own));
	if (pairs) PSVR_HIP(hipMemcpyAsync(pairs, c.c.pres, (size_t)c.P * sizeof(psvr_pair_result_t), hipMemcpyDefault, e->own));
	if (cands && e->compact_nc) PSVR_HIP(hipMemcpyAsync(cands, e->cmp_cand.p, (size_t)e->compact_nc * sizeof(psvr_cand_t), hipMemcpyDefault, e->own));
	if (cigar && e->compact_nw) PSVR_HIP(hipMemcpyAsync(cigar, e->cmp_cig.p, (size_t)e->compact_nw * 4, hipMemcpyDefault, e->own));
	PSVR_HIP(hipStreamSynchronize(e->own));
	return PSVR_OK;
}

extern "C" int psvr_engine_align_batch(psvr_engine_t *e, int64_t n_pairs, const char *bases, const int64_t *base_off, const psvr_ori_t *ori,
                                       psvr_read_result_t *reads, psvr_pair_result_t *pairs, uint32_t *cigar, int64_t cigar_cap, int trace)
{
	int rc = psvr_engine_upload(e, n_pairs, bases, base_off, ori);
	if (rc) return rc;
	rc = psvr_engine_run(e, trace, nullptr);
	if (rc) return rc;
	return psvr_engine_download(e, reads, pairs, cigar, cigar_cap, nullptr);
}

extern "C" int psvr_engine_stats(const psvr_engine_t *e, char *buf, size_t n)
{
	if (!e || !buf || !n) return set_error(PSVR_ERR_ARG, "psvr_engine_stats: bad argument");
	const RunStats &s = e->core.stats;
	snprintf(buf, n,
	         "{\"pairs\":%lld,\"rounds\":%lld,\"pair_runs\":%lld,\"pair_only_runs\":%lld,\"shadow_runs\":%lld,\"sensitive_pairs\":%lld,\"window_misses\":%lld,\"adopted_pairs\":%lld,\"stale_open\":%lld,\"dp_problems\":%lld,\"dp_seq_bytes\":%lld,\"candidates\":%lld,\"walk_pairs\":%lld,\"walk_us\":%lld,\"n_special\":%lld,\"special_const\":%lld,\"special_nomove\":%lld,"
	         "\"probes\":%llu,\"hits\":%llu,\"seeds\":%llu,\"dp_cells\":%llu,\"simple\":%llu,\"reads_aligned\":%llu}",
	         e->core.P, s.rounds, s.pairs_run, s.pair_only, s.shadow_runs, s.sensitive, s.window_miss, s.adopted, s.stale_open, s.dp_problems, s.dp_seq_bytes, s.cands, s.walk_pairs, s.walk_us, s.n_special, s.special_const, s.special_nomove, s.counters[ST_PROBES], s.counters[ST_HITS], s.counters[ST_SEEDS],
	         s.counters[ST_CELLS], s.counters[ST_SIMPLE], s.counters[ST_READS]);
	std::string t = buf;
	t.pop_back();
	{
		size_t fr = 0, tot = 0;
		char b[96];
		if (hipSetDevice(e->ix->device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess) { snprintf(b, sizeof b, ",\"hbm_used_bytes\":%zu", tot - fr); t += b; }
	}
	t += ",\"kernels\":{";
	bool first = true;
	for (auto &k : e->be.timed) {
		char b[256];
		snprintf(b, sizeof b, "%s\"%s\":{\"ms\":%.4f,\"launches\":%lld}", first ? "" : ",", k.first.c_str(), k.second.first, k.second.second);
		t += b, first = false;
	}
	t += "}}";
	snprintf(buf, n, "%s", t.c_str());
	return PSVR_OK;
}
