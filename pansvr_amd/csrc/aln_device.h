// aln_device.h -- the per-item stages of the `aln` hot path (seam B1) as __host__ __device__ functions.
//
// Every stage is "one GPU thread owns one item" (read, read-strand, candidate or pair); the only
// wave-cooperative stage is the banded DP (ksw_kernels.hip).  Data lives in flat HBM arrays and
// bump arenas so a batch of millions of reads runs as a handful of launches:
//
//   prep      read       ASCII -> 2-bit fwd/rev-comp bytes + packed words, N -> rand() draw
//   seed      strand     STR mask, stride-5 20-mer probes of the HBM hash, MEM extension in unipaths
//   chain     strand     merge MEMs, expand to reference positions, sort, sparse chaining DP
//   select    read       iterative best-chain extraction (rand() tie-breaks), candidate cut
//   walk      candidate  chain -> extension / between-seed sub-problems; simple ones scored inline,
//                        the rest queued for the DP kernel
//   (fetch + DP kernels: ksw_kernels.hip)
//   assemble  candidate  scores + CIGAR pieces -> merged CIGAR, align score
//   finalize  read       sort, thresholds, anchor -> genome coordinates, mapq
//   pair      pair       PE_score (rand() tie-breaks), primary / secondary / mate
//
// The functions are also compiled for the host by tests/emu (test-only) to check this logic against
// the oracle without a GPU; the product library never calls them on the host.
//
// Reference: src/PanSVgenerateVCF/read_realignment.{cpp,hpp} (== src/jlra_aln.{cpp,hpp}),
// src/deBGA_index.{cpp,hpp}, src/cpp_lib/graph.cpp, src/clib/binarys_qsort.c.  `rr` below =
// src/PanSVgenerateVCF/read_realignment.
#pragma once
#include <stdint.h>
#include "../../include/psvr_engine.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PSVR_HD __host__ __device__ __forceinline__
#define PSVR_HDN __host__ __device__
#else
#define PSVR_HD inline
#define PSVR_HDN
#endif

namespace psvr {

static const int kLenKmer = 20, kSeedStep = 5, kUniPosNMax = 32;   // rr.hpp:26-29, deBGA_index.hpp:17
static const int kMaxOut = 6;                                       // MAX_OUTPUT_NUMBER
static const int kMaxReadLen = 1600;                                // MAX_READ_LEN, rr.hpp:322
static const int kFwd = 1, kRev = 0;                                // clib/utils.h:72-73
static const int kSegMax = 160;                                     // CIGAR pieces per candidate (a DP piece takes three slots until walk_read has numbered it)
static const int kCigMax = 256;                                     // merged CIGAR ops per candidate
static const int kMemSlot = 32;                                     // MEMs per read-strand before spilling to the bump region

struct SvDev { uint32_t chr_id; uint32_t st_pos; int32_t end_offset; int32_t pad; };

struct HitRec { uint64_t kp; uint32_t uid, ul, ur, pos_n; uint32_t pad[2]; };   // 32 bytes: one sector, two 16-byte loads

struct DevIndex {
	const uint64_t *ref_seq, *seq, *seqf, *pos, *posp, *hash, *off;
	const uint32_t *kmer;
	uint64_t n_seqf;
	const uint32_t *chr_end_n, *chr_search_index;
	const SvDev *sv;
	int32_t chr_file_n;
	// 1 bit per first-level bucket (32 MiB, stays resident in the 256 MiB Infinity Cache): set iff the bucket holds a k-mer.
	// ~85 % of all probes (wrong strand, unrelated reads, mismatching positions) end here without touching the 2 GiB table.
	const uint32_t *occ;
	// optional: uid_hint[b] = last unipath that starts at or before position b << uid_shift, so the unipath of a position is found by
	// bisecting between two neighbouring hints (a couple of loads) instead of over all n_seqf starts (14+ dependent loads per MEM)
	const uint32_t *uid_hint; uint32_t uid_shift;
	// optional, in front of (instead of) the occupancy bitmap: a Bloom filter over the 20-mers the index holds (the first 20 bases of its
	// 22-mers), three bits in ONE 64-bit word per key (kmer_maybe_present).  16 MiB for a 12 M-k-mer index against the bitmap's 32 MiB:
	// twice the share of it stays in an XCD's 4 MiB L2 -- a probe is one random 64-byte sector either way, and those sectors are what
	// bounds the seeding kernel --, and it answers for the 20-mer itself, not for its 14-base bucket: a read k-mer that differs from an
	// indexed one in its last six bases (every k-mer over a mismatch) passes the bitmap and fails here.
	const uint64_t *bloom; uint32_t bloom_shift;    // word = (kmer * K) >> bloom_shift
	// optional: per index entry (22-mer occurrence) what UNITIG_MEM_search derives from it -- its position in the unipath sequence array, its
	// unipath, the room to the unipath's two ends, the unipath's number of reference positions -- in one 32-byte record.  Looked up per hit
	// these are off[hit], two bracket-table entries, three or four steps of a bisection over the unipath starts, the two starts again and
	// two position-list offsets: ten loads each waiting for the one before, in a kernel whose lanes wait for memory four cycles of five.
	const struct HitRec *hitrec;
	// tests/emu only (PSVR_EMU_SPARSE_HASH): non-empty first-level buckets instead of the dense 2 GiB table
	const uint32_t *sp_id; const uint64_t *sp_start; uint64_t sp_n, n_kmer;
};

// hash[h], hash[h+1] of the first-level table (one 16-byte gather on the device)
PSVR_HD void hash_pair(const DevIndex &ix, uint64_t h, uint64_t &lo, uint64_t &hi);

struct VMem { uint64_t uid; uint32_t seed_id, read_pos, uni_pos_off, length, pos_n, pad; };   // vertex_MEM
struct VU { uint64_t uid; uint32_t read_pos, uni_pos_off, length1, length2, pos_n, cov; };      // vertex_U
struct USeed { uint32_t read_begin, read_end, seed_id, ref_begin, ref_end, cov; };              // UNI_SEED
struct PathN { int32_t dist, pre_node; uint32_t brk; uint32_t used; };                          // PATH_t (+ scan break)

// Bump arena.  `nshard` > 1 (GPU backend, pure storage arenas): the arena is cut into nshard equal regions, each with its own counter in
// its own cache line (kArenaTopStride words apart); a workgroup bumps the counter of region blockIdx % nshard.  Atomics on one line are
// served one wavefront instruction after the other (~12 ns each, tools/atomic_rate_bench.hip): one counter bumped once per read or
// strand was what bounded the kernels doing it.
static const int kArenaTopStride = 32, kArenaMaxShards = 16;
template <class T> struct Arena {
	T *base;
	unsigned long long *top;
	unsigned long long cap;
	int *overflow;
	unsigned int nshard;
};

PSVR_HD unsigned long long atomic_bump(unsigned long long *p, unsigned long long n)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return atomicAdd(p, n);
#else
	unsigned long long o = *p;
	*p += n;
	return o;
#endif
}

template <class T> PSVR_HD long long arena_alloc(const Arena<T> &a, unsigned long long n)
{
#if defined(__HIP_DEVICE_COMPILE__)
	if (a.nshard > 1) {
		// region blockIdx % nshard first; a full region (a launch of few workgroups uses few regions: ADVICE r2) sends the request on to the
		// next ones before the arena counts as full
		const unsigned long long cap_s = a.cap / a.nshard;
		for (unsigned int k = 0; k < a.nshard; ++k) {
			const unsigned long long s = (blockIdx.x + k) % a.nshard;
			if (k && a.top[s * kArenaTopStride] + n > cap_s) continue;          // (a look before the bump: a full region's counter is left alone)
			const unsigned long long o = atomic_bump(a.top + s * kArenaTopStride, n);
			if (o + n <= cap_s) return (long long)(s * cap_s + o);
		}
		*a.overflow = 1;
		return -1;
	}
#endif
	unsigned long long o = atomic_bump(a.top, n);
	if (o + n > a.cap) { *a.overflow = 1; return -1; }
	return (long long)o;
}

struct Strand {                  // per read-strand bookkeeping
	long long mem_off; uint32_t mem_n;
	long long us_off;  uint32_t us_n;      // USeed + PathN arenas share offsets
	uint64_t seed_hash, chain_hash;
};

struct ChainCand {               // one sort_output() pick (rr.cpp:212-293)
	uint32_t chain_score, max_index, read_bg, ref_bg;
	int32_t chr_id;              // anchor id from get_chromosome_ID
	int32_t direction;
};

struct Seg {                     // one piece of KSW_ALN_handler::cigar_tmp, in push order
	int32_t kind;                // 0 literal op, 1 DP result, 2 payload of the DP piece in front of it (skipped by assemble_candidate)
	int32_t a, b;                // literal: type,size ; DP: problem id, emit order (0 forward i=0.., 1 reverse)
};

struct DpDesc {                  // one queued ksw_extd2_sse call (rr.cpp:872-891,910-986)
	int32_t read, strand;        // query source
	int32_t q_st, qlen;          // read_str + q_st, qlen bases (reversed for left extension)
	uint32_t ref_st; int32_t tlen;
	int32_t type;                // 0 left, 1 right, 2 end-to-end
	int32_t pad;
};

struct CandWork {                // a candidate between walk and assemble
	int32_t read, k;             // read id, slot in the read's candidate list
	int32_t n_seg;
	int32_t read_score;          // partial: simple pieces + gap penalties + tail terms
	int32_t rba;                 // read_begin_alignment
	int32_t bad;
	long long seg_off;
};

struct Ctx {
	DevIndex idx;
	psvr_aln_params_t par;
	int8_t mat[25];
	// batch inputs
	long long n_pairs;
	const char *bases; const long long *base_off; const psvr_ori_t *ori;
	int32_t lmax;                // padded per-strand byte stride
	int32_t wmax;                // packed words per strand
	// rand streams: rand() and the two per-handler random_r streams (rr.hpp:340)
	const int32_t *grand; long long grand_n; long long grand_base;
	const int32_t *hrand[2]; long long hrand_n; long long hrand_base[2];
	// Work items are SLOTS: slot < n_pairs is a real pair, slot >= n_pairs a shadow evaluation of pair src[slot]
	// at another rand() offset (engine_core.h).  All per-read state below is indexed by 2*slot + mate.
	long long n_slots;
	const int32_t *src;                      // slot -> input pair (nullptr: identity)
	long long *poff;                         // per slot: rand() stream offset of the pair (mate 0 draws first, then mate 1, then pairing)
	int32_t *rcnt;                           // per slot x {mate0, mate1, pairing}: draws consumed
	long long *hoff; int32_t *hcnt;          // per read: random_r offsets / counts (one stream per mate)
	// forced N-substitution draws of "variant" shadow slots (engine_core.h): read r = 2*slot+mate replaces its first
	// force[4r] N draws by the residues force[4r+1..3]
	const uint8_t *force;
	// per read
	uint8_t *active; uint8_t *unmapped; uint8_t *is_str; uint8_t *has_n4;
	uint8_t *has_mem;            // per read: a strand found MEMs (set by the seeding stage; the GPU backend compacts its chain / select work by it)
	int32_t *str_list; unsigned int *str_cnt;   // GPU backend: reads whose STR screen was inconclusive (is_str == 2), for the exact count
	int32_t *read_l;
	uint8_t *bin;                // [read][2][lmax]
	uint64_t *rb;                // [read][2][wmax]
	uint8_t *seed_list;          // [read][lmax]
	Strand *strand;              // [read][2]
	ChainCand *ccand; int32_t *n_ccand;      // [read][12]
	psvr_read_hdr_t *rh; psvr_pair_result_t *pres;   // per read: compact header; per slot: pairing result
	psvr_cand_t *cand;           // candidate records, parallel to the CandWork arena (a read's candidates are contiguous: rh.cand_off)
	// arenas
	Arena<VMem> mem; Arena<USeed> us; PathN *path;   // path shares the us arena's offsets
	Arena<Seg> seg; Arena<DpDesc> dp; Arena<CandWork> cw; Arena<uint32_t> cig;
	// DP results
	const psvr_extz_t *dp_ez; const uint32_t *dp_cig;
	// the reference's per-handler scratch buffer as earlier reads left it ([mate][1600], see stale_compare); stale_open: set when a
	// compare needed a position this read's own calls had not written
	const uint8_t *tseq_in; int32_t *stale_open;
	int32_t *any_h;              // set when a read draws from random_r (expand_seed's sampling)
	int32_t trace;
	int32_t *err;                // sticky error word (reference would xassert/abort)
	unsigned long long *stats;   // [16] work counters
};

enum { ST_PROBES = 0, ST_HITS, ST_SEEDS, ST_DP, ST_SIMPLE, ST_CELLS, ST_READS, ST_CAND, ST_N };

PSVR_HD void stat_add(const Ctx &c, int k, unsigned long long v)
{
	if (!c.stats) return;
#if defined(__HIP_DEVICE_COMPILE__)
	atomicAdd(c.stats + k, v);
#else
	c.stats[k] += v;
#endif
}

PSVR_HD long long src_read(const Ctx &c, long long read)
{
	return c.src ? (long long)c.src[read >> 1] * 2 + (read & 1) : read;
}

PSVR_HD uint64_t fnv1a(uint64_t h, uint64_t v)
{
	for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ULL; }
	return h;
}

// false: first-level bucket h is certainly empty (1-bit probe of the occupancy bitmap)
PSVR_HD bool bucket_occupied(const DevIndex &ix, uint64_t h)
{
#ifdef PSVR_EMU_SPARSE_HASH
	(void)ix, (void)h;
	return true;
#else
	return !ix.occ || ((ix.occ[h >> 5] >> (h & 31)) & 1u);
#endif
}

// word and bit mask of a 20-mer in the Bloom filter
PSVR_HD void bloom_slot(uint64_t kmer, uint32_t shift, uint64_t &word, uint64_t &mask)
{
	const uint64_t h = kmer * 0x9E3779B97F4A7C15ull;
	const uint64_t g = (h ^ (h >> 32)) * 0xD6E8FEB86659FD93ull;
	word = h >> shift;
	mask = (1ull << (g >> 58)) | (1ull << ((g >> 52) & 63)) | (1ull << ((g >> 46) & 63));
}
// false: the index certainly holds no 22-mer that starts with this 20-mer (no false negatives: every indexed 20-mer has its bits set)
PSVR_HD bool kmer_maybe_present(const DevIndex &ix, uint64_t kmer)
{
#ifdef PSVR_EMU_SPARSE_HASH
	(void)ix, (void)kmer;
	return true;
#else
	if (ix.bloom) {
		uint64_t w, m;
		bloom_slot(kmer, ix.bloom_shift, w, m);
		return (ix.bloom[w] & m) == m;
	}
	return bucket_occupied(ix, kmer >> 12);
#endif
}

PSVR_HD void hash_pair(const DevIndex &ix, uint64_t h, uint64_t &lo, uint64_t &hi)
{
#ifdef PSVR_EMU_SPARSE_HASH
	auto at = [&](uint64_t x) {
		uint64_t l = 0, r = ix.sp_n;
		while (l < r) { uint64_t m = (l + r) >> 1; if (ix.sp_id[m] < x) l = m + 1; else r = m; }
		return l < ix.sp_n ? ix.sp_start[l] : ix.n_kmer;
	};
	lo = at(h), hi = at(h + 1);
#else
	lo = ix.hash[h], hi = ix.hash[h + 1];
#endif
}

PSVR_HD int base_at(const uint64_t *w, uint64_t i) { return (int)((w[i >> 5] >> ((31 - (i & 0x1f)) << 1)) & 3); }

// 32 bases starting at base offset `i` of a 2-bit packed sequence (MSB first), as one word
PSVR_HD uint64_t window32(const uint64_t *w, uint64_t i)
{
	uint64_t k = i >> 5, sh = (i & 31) << 1;
	uint64_t a = w[k];
	return sh ? (a << sh) | (w[k + 1] >> (64 - sh)) : a;
}
// number of positions j < len with A[ia + j] != B[ib + j] (both 2-bit packed), at most `cap`.
// 128 bases per turn: the five words of each sequence a turn needs are requested together and only then looked at -- a loop that
// loads a window, counts, and decides whether to go on costs one memory round trip per 32 bases (the reference sequence is a random
// access per piece), and the early exit it buys is just min(count, cap).
PSVR_HD int mismatches_packed(const uint64_t *A, uint64_t ia, const uint64_t *B, uint64_t ib, int len, int cap)
{
	int nm = 0;
	for (int j = 0; j < len && nm < cap; j += 128) {
		const uint64_t ka = (ia + (uint64_t)j) >> 5, kb = (ib + (uint64_t)j) >> 5;
		const unsigned sa = (unsigned)((ia + (uint64_t)j) & 31) << 1, sb = (unsigned)((ib + (uint64_t)j) & 31) << 1;
		const int rem = len - j, nw = rem >= 128 ? 4 : (rem + 31) >> 5;          // 32-base windows of this turn
		uint64_t wa[5], wb[5];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
		for (int q = 0; q < 5; ++q) { const bool in = q <= nw; wa[q] = in ? A[ka + q] : 0, wb[q] = in ? B[kb + q] : 0; }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
		for (int q = 0; q < 4; ++q) {
			if (q >= nw) break;
			const uint64_t xa = sa ? (wa[q] << sa) | (wa[q + 1] >> (64 - sa)) : wa[q], xb = sb ? (wb[q] << sb) | (wb[q + 1] >> (64 - sb)) : wb[q];
			uint64_t x = xa ^ xb;
			x = (x | (x >> 1)) & 0x5555555555555555ull;
			const int r = rem - 32 * q;
			if (r < 32) x &= ~0ull << ((32 - r) << 1);
#if defined(__HIP_DEVICE_COMPILE__)
			nm += __popcll(x);
#else
			nm += __builtin_popcountll(x);
#endif
		}
	}
	return nm < cap ? nm : cap;
}

// ---------------------------------------------------------------------------------------------
// prep: parse-independent part of single_end_handler::align up to binary_read_2_bit (rr.cpp:406-416,646-654)
// ---------------------------------------------------------------------------------------------
PSVR_HDN inline void prep_read(const Ctx &c, long long read)
{
	const long long sr = src_read(c, read);
	const psvr_ori_t &o = c.ori[sr];
	const int L = (int)(c.base_off[sr + 1] - c.base_off[sr]);
	c.read_l[read] = L;
	bool unm = o.unmapped != 0;
	if ((uint32_t)o.chr_id > 24u) unm = true;                          // rr.cpp:413
	c.unmapped[read] = unm;
	c.is_str[read] = 0, c.has_mem[read] = 0;
	long long item = (read >> 1) * 3 + (read & 1);
	c.rcnt[item] = 0;
	c.hcnt[read] = 0;
	c.n_ccand[read] = 0;
	for (int s = 0; s < 2; ++s) { Strand &st = c.strand[read * 2 + s]; st.mem_n = st.us_n = 0; st.mem_off = st.us_off = 0; st.seed_hash = st.chain_hash = 1469598103934665603ULL; }
	if (L > kMaxReadLen || L < kLenKmer) { c.active[read] = 0; if (L > kMaxReadLen) *c.err = 1; return; }
	if (!unm && o.align_score == (uint32_t)(L * c.par.match)) { c.active[read] = 0; return; }   // rr.cpp:414
	c.active[read] = 1;
	const char *s = c.bases + c.base_off[sr];
	uint8_t *b0 = c.bin + (read * 2) * (long long)c.lmax, *b1 = b0 + c.lmax;
	uint64_t *w0 = c.rb + (read * 2) * (long long)c.wmax, *w1 = w0 + c.wmax;
	for (int i = 0; i < c.wmax; ++i) w0[i] = 0, w1[i] = 0;
	// mate 1 draws after mate 0 of the same pair (mate 0's stages have completed: engine_core.h runs the mates in turn)
	long long ro = c.poff[read >> 1] + ((read & 1) ? c.rcnt[(read >> 1) * 3] : 0);
	int draws = 0;
	for (int i = 0; i < L; ++i) {
		char ch = s[i];
		if (ch == 'N') {
			long long k = ro + draws - c.grand_base;
			int32_t r;
			if (c.force && draws < (int)c.force[4 * read]) r = c.force[4 * read + 1 + draws];
			else r = (k >= 0 && k < c.grand_n) ? c.grand[k] : (*c.err = 2, 0);
			ch = "ACGT"[r % 4];
			++draws;
		}
		uint8_t code = (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : (ch == 'n') ? 4 : 0; // charToDna5n
		b0[i] = code;
		b1[L - 1 - i] = code ^ 3;
	}
	for (int i = 0; i < L; ++i) {                                       // binary_read_64_bit, rr.cpp:295-300
		w0[i >> 5] |= ((uint64_t)b0[i]) << ((31 - (i & 0x1f)) << 1);
		w1[i >> 5] |= ((uint64_t)b1[i]) << ((31 - (i & 0x1f)) << 1);
	}
	c.rcnt[item] = draws;
	{ uint8_t any4 = 0; for (int i = 0; i < L; ++i) any4 |= b0[i] > 3; c.has_n4[read] = any4; }
	stat_add(c, ST_READS, 1);
}

PSVR_HD uint64_t get_kmer(uint32_t off, const uint64_t *rb)           // getKmer, rr.cpp:204-210
{
	uint32_t w = off >> 5, iw = off & 0x1f;
	uint64_t full = (rb[w] << (iw << 1)) | (iw == 0 ? 0 : (rb[w + 1] >> ((32 - iw) << 1)));
	return full >> ((32 - kLenKmer) << 1);
}

// STR detection on the forward strand (rr.cpp:549-598).  O(n^2) distinct/count over <= L-19 k-mers.
PSVR_HDN inline void str_detect(const Ctx &c, long long read)
{
	if (!c.active[read]) return;
	const int L = c.read_l[read];
	const uint64_t *rb = c.rb + (read * 2) * (long long)c.wmax;
	uint8_t *sl = c.seed_list + read * (long long)c.lmax;
	const uint32_t kn = L - kLenKmer + 1;
	uint32_t distinct = 0;
	for (uint32_t i = 0; i < kn; ++i) {
		uint64_t ki = get_kmer(i, rb);
		bool first = true;
		for (uint32_t j = 0; j < i; ++j) if (get_kmer(j, rb) == ki) { first = false; break; }
		distinct += first;
	}
	if (!(distinct < kn - 15)) { c.is_str[read] = 0; return; }
	c.is_str[read] = 1;
	for (uint32_t i = 0; i < kn; ++i) {
		uint64_t ki = get_kmer(i, rb);
		int cnt = 0;
		for (uint32_t j = 0; j < kn; ++j) cnt += get_kmer(j, rb) == ki;
		sl[i] = cnt >= 4 ? 0 : 1;
	}
	int bg = 0, ed = 0;
	for (uint32_t o = 0; o < (uint32_t)kSeedStep; ++o) {
		bg += sl[o] == 0, ed += sl[L - kLenKmer - o] == 0;
		sl[o] += 2, sl[L - kLenKmer - o] += 4;
	}
	if (bg < kSeedStep && ed < kSeedStep) {
		int tot = 0;
		for (uint32_t o = 0; tot < kSeedStep && o < kn; ++o) {
			if (sl[o] > 0) continue;
			sl[o] += 8, tot++;
		}
	}
}

// seed_list as the reverse strand sees it: getReverseStr_qual (clib/bam_file.c:341-349) swaps i <-> len-1-i for
// i = 0..len/2 INCLUSIVE, so for even len the two middle entries are swapped twice
PSVR_HD uint8_t seed_list_at(const uint8_t *sl, int len, int rev, int i)
{
	if (!rev) return sl[i];
	if (!(len & 1) && (i == len / 2 || i == len / 2 - 1)) return sl[i];
	return sl[len - 1 - i];
}

PSVR_HD int clz64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __clzll((long long)x);
#else
	return __builtin_clzll(x);
#endif
}
PSVR_HD int ctz64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __ffsll((unsigned long long)x) - 1;
#else
	return __builtin_ctzll(x);
#endif
}
// number of consecutive j in [0,max) with A[ia + j] == B[ib + j] (2-bit packed, 32 bases per step)
PSVR_HD uint32_t match_right(const uint64_t *A, uint64_t ia, const uint64_t *B, uint64_t ib, uint32_t max)
{
	// 128 bases per turn, the (up to) five words of A a turn needs requested together: A is the index's unipath sequence, in global memory, and a
	// window at a time was one memory round trip per 32 bases of the MEM (a 150-base read that matches: five in a row); the words are
	// exactly those the window-at-a-time loop reads
	uint32_t n = 0;
	while (n < max) {
		const uint64_t ka = (ia + n) >> 5;
		const unsigned sa = (unsigned)((ia + n) & 31) << 1;
		const uint32_t rem = max - n;
		const int nw = rem >= 128 ? 4 : (int)((rem + 31) >> 5);
		uint64_t wa[5];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
		for (int q = 0; q < 5; ++q) wa[q] = (q < nw || (q == nw && sa != 0)) ? A[ka + q] : 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
		for (int q = 0; q < 4; ++q) {
			if (q >= nw) break;
			const uint64_t xa = sa ? (wa[q] << sa) | (wa[q + 1] >> (64 - sa)) : wa[q];
			uint64_t x = xa ^ window32(B, ib + n + 32u * (uint32_t)q);
			x = (x | (x >> 1)) & 0x5555555555555555ull;
			const uint32_t left = rem - 32u * (uint32_t)q, lim = left < 32 ? left : 32;
			if (x) { const uint32_t k = (uint32_t)clz64(x) >> 1; return n + 32u * (uint32_t)q + (k < lim ? k : lim); }
		}
		n += rem < 128 ? rem : 128;
	}
	return max;
}
// number of consecutive j in [0,max) with A[ia - 1 - j] == B[ib - 1 - j]; requires ia >= max and ib >= max
PSVR_HD uint32_t match_left(const uint64_t *A, uint64_t ia, const uint64_t *B, uint64_t ib, uint32_t max)
{
	uint32_t n = 0;
	while (n < max) {
		uint32_t lim = max - n < 32 ? max - n : 32;
		uint64_t x = window32(A, ia - n - lim) ^ window32(B, ib - n - lim);   // bases 0..lim-1 of the windows are the lim bases left of the cursor
		x = (x | (x >> 1)) & 0x5555555555555555ull;
		if (lim < 32) x &= ~0ull << ((32 - lim) << 1);
		if (x) { uint32_t k = ((uint32_t)ctz64(x) - ((32 - lim) << 1)) >> 1; return n + k; }
		n += lim;
	}
	return max;
}

// search_kmer (deBGA_index.cpp:84-101) + binsearch_range with k_off = 4 (binarys_qsort.c:25-100): index range of the 22-mers
// whose first 20 bases equal `kmer`; returns the number of hits (0 = not found) and the first hit index
PSVR_HD uint32_t probe_kmer(const DevIndex &ix, uint64_t kmer, uint64_t &first_hit)
{
	const uint64_t key = kmer & 0xfff, h = kmer >> 12;
	uint64_t lo, hi;
	hash_pair(ix, h, lo, hi);
	long long l = 0, r = (long long)(hi - lo) - 1, first = -1, last = -1;
	const uint32_t *v = ix.kmer + lo;
	while (l <= r) {
		long long m = (l + r) / 2;
		uint32_t t = v[m] >> 4;
		if (t == key) {
			first = last = m;
			long long sl2 = l, sr = m - 1;
			while (sl2 <= sr) { long long sm = (sl2 + sr) / 2; uint32_t t2 = v[sm] >> 4; if (t2 == key) first = sm, sr = sm - 1; else if (t2 > key) sr = sm - 1; else sl2 = sm + 1; }
			sl2 = m + 1, sr = r;
			while (sl2 <= sr) { long long sm = (sl2 + sr) / 2; uint32_t t2 = v[sm] >> 4; if (t2 == key) last = sm, sl2 = sm + 1; else if (t2 > key) sr = sm - 1; else sl2 = sm + 1; }
			break;
		} else if (t > key) r = m - 1;
		else l = m + 1;
	}
	if (first < 0) return 0;
	first_hit = lo + (uint64_t)first;
	return (uint32_t)(last - first + 1);
}

// UNITIG_MEM_search (deBGA_index.cpp:105-146) for index entry `hit`; returns right_i
// what the index alone says about entry `hit` (the part of UNITIG_MEM_search that does not look at the read)
PSVR_HDN inline HitRec hit_record(const DevIndex &ix, uint64_t hit)
{
	HitRec h;
	const uint64_t kp = ix.off[hit];
	long long lo2 = 0, hi2 = (long long)ix.n_seqf - 1, uid = -1;
	if (ix.uid_hint) {                                      // same search, started from a bracket that is known to hold the answer
		const uint64_t b = kp >> ix.uid_shift;
		lo2 = (long long)ix.uid_hint[b], hi2 = (long long)ix.uid_hint[b + 1];
	}
	while (lo2 <= hi2) {                                    // binsearch_interval_unipath64 (binarys_qsort.c:162-187)
		long long mid = (lo2 + hi2) >> 1;
		uint64_t sv = ix.seqf[mid];
		if (kp < sv) hi2 = mid - 1;
		else if (kp > sv) lo2 = mid + 1;
		else { uid = mid; break; }
	}
	if (uid < 0) uid = hi2;
	const uint64_t f0 = ix.seqf[uid], f1 = ix.seqf[uid + 1];
	h.kp = kp, h.uid = (uint32_t)uid, h.ul = (uint32_t)(kp - f0), h.ur = (uint32_t)(f1 - (kp + kLenKmer));
	h.pos_n = (uint32_t)(ix.posp[uid + 1] - ix.posp[uid]), h.pad[0] = h.pad[1] = 0;
	return h;
}
PSVR_HD uint32_t mem_for_hit(const DevIndex &ix, uint64_t hit, const uint64_t *rb, uint32_t off, int L, VMem &m)
{
	const HitRec h = ix.hitrec ? ix.hitrec[hit] : hit_record(ix, hit);
	const uint64_t kp = h.kp, uid = h.uid;
	const uint32_t ul = h.ul, ur = h.ur;
	const uint32_t lmax = ul < off ? ul : off, rmax0 = (uint32_t)(L - (int)off - kLenKmer), rmax = ur < rmax0 ? ur : rmax0;
	const uint32_t li = 1 + match_left(ix.seq, kp, rb, off, lmax);
	const uint32_t ri = 1 + match_right(ix.seq, kp + kLenKmer, rb, (uint64_t)off + kLenKmer, rmax);
	m.uid = (uint64_t)uid, m.seed_id = 0, m.read_pos = off + 1 - li, m.uni_pos_off = ul + 1 - li;
	m.length = kLenKmer + li + ri - 2, m.pos_n = h.pos_n, m.pad = 0;
	return ri;
}

// seed loop of chainning_one_read (rr.cpp:614-635) for one strand: search_kmer + binsearch_range
// (deBGA_index.cpp:84-101, binarys_qsort.c:25-100) and UNITIG_MEM_search (deBGA_index.cpp:105-146)
// The loop is written as a three-stage state machine so that the lanes of a wavefront regroup: all lanes first run the cheap
// stage (skip rules + occupancy bit) until each has a k-mer whose bucket is non-empty or is done; then all of them do the
// hash gather + bucket bisection together; that repeats until every lane holds a hit or is done, and only then the hits
// are extended.  Written as `for off: probe; if hit: extend` the rare stages run on nearly every iteration with a handful
// of lanes active.  Per strand the order of operations -- and every result -- is unchanged.
// `rb_local`: optional copy of this strand's packed words in fast memory (the GPU kernel stages them in LDS).
template <bool LOCAL> PSVR_HD void seed_strand_t(const Ctx &c, long long rs, const uint64_t *rb_local)
{
	const long long read = rs >> 1;
	const int rev = (int)(rs & 1);
	if (!c.active[read]) return;
	const int L = c.read_l[read];
	const uint64_t *rb = LOCAL ? rb_local : c.rb + rs * (long long)c.wmax;
	const uint8_t *sl = c.seed_list + read * (long long)c.lmax;
	const bool is_str = c.is_str[read] != 0;
	const uint32_t kn = L - kLenKmer + 1;
	const DevIndex &ix = c.idx;
	// MEMs go to this strand's fixed slot of kMemSlot entries; a strand with more re-runs as count + fill
	// passes into the bump region behind the slots (rare: > 32 MEMs needs repeats)
	long long base = rs * (long long)kMemSlot;
	uint32_t total = 0, probes = 0;
	for (int pass = 0; pass < 3; ++pass) {            // 0: write into slot (counting), 1: count only (skipped), 2: fill arena slice
		if (pass == 1) continue;
		uint32_t n = 0, msr = 0, off = 0, nh = 0;
		uint64_t kmer = 0, first_hit = 0;
		int stage = 0;                                // 0: scanning, 1: bucket non-empty, 2: hits to extend
		for (;;) {
			while (stage != 2 && off < kn) {
				while (off < kn) {                        // stage 0
					if (off + kLenKmer - 1 <= msr || (is_str && seed_list_at(sl, (int)kn, rev, off) == 0)) { off += kSeedStep; continue; }
					if (pass == 0) ++probes;
					kmer = get_kmer(off, rb);
					if (kmer_maybe_present(ix, kmer)) { stage = 1; break; }
#if defined(PSVR_DIAG_SEED) && PSVR_DIAG_SEED == 2     /* timing experiment: a strand whose first k-mer is refused is done (results are wrong) */
					if (n == 0) { off = kn; break; }
#endif
					off += kSeedStep;
				}
				if (stage != 1) break;
#if defined(PSVR_DIAG_SEED) && PSVR_DIAG_SEED == 1     /* timing experiment: no hash gather, no MEMs (results are wrong) */
				nh = 0;
#else
				nh = probe_kmer(ix, kmer, first_hit);     // stage 1
#endif
				if (nh == 0 || nh > (uint32_t)kUniPosNMax) { off += kSeedStep; stage = 0; }
				else stage = 2;
			}
			if (stage != 2) break;
			uint32_t mri = 1;                             // stage 2
			for (uint64_t hit = first_hit; hit < first_hit + nh; ++hit) {
				VMem m;
				const uint32_t ri = mem_for_hit(ix, hit, rb, off, L, m);
				m.seed_id = n;
				if (pass == 2 || n < (uint32_t)kMemSlot) c.mem.base[base + n] = m;
				++n;
				if (ri > mri) mri = ri;
			}
			msr = off + kLenKmer + mri - 1;
			off += kSeedStep;
			stage = 0;
		}
		if (pass == 0) {
			total = n;
			if (n <= (uint32_t)kMemSlot) break;
			base = arena_alloc(c.mem, n);
			if (base < 0) return;
		}
	}
	Strand &st = c.strand[rs];
	st.mem_off = base, st.mem_n = total;
	if (total) c.has_mem[read] = 1;
	if (c.stats) { stat_add(c, ST_PROBES, probes); stat_add(c, ST_HITS, total); }
}
PSVR_HDN inline void seed_strand(const Ctx &c, long long rs) { seed_strand_t<false>(c, rs, nullptr); }

// stable bottom-up merge sort of idx[0..n) by less(a,b); tmp has n entries
template <class T, class Less> PSVR_HD void stable_sort(T *a, T *tmp, uint32_t n, Less less)
{
	if (n < 2) return;
	if (n <= 16) {                                                       // stable insertion sort
		for (uint32_t i = 1; i < n; ++i) {
			T x = a[i];
			uint32_t j = i;
			while (j > 0 && less(x, a[j - 1])) { a[j] = a[j - 1]; --j; }
			a[j] = x;
		}
		return;
	}
	T *src = a, *dst = tmp;
	for (uint32_t w = 1; w < n; w <<= 1) {
		for (uint32_t lo = 0; lo < n; lo += 2 * w) {
			uint32_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
			uint32_t i = lo, j = mid, k = lo;
			while (i < mid && j < hi) { if (less(src[j], src[i])) dst[k++] = src[j++]; else dst[k++] = src[i++]; }
			while (i < mid) dst[k++] = src[i++];
			while (j < hi) dst[k++] = src[j++];
		}
		T *t = src; src = dst; dst = t;
	}
	if (src != a) for (uint32_t i = 0; i < n; ++i) a[i] = src[i];
}

// merge_seed_in_unipath + expand_seed + Graph_handler::process (deBGA_index.cpp:151-251, graph.cpp:53-150)
PSVR_HDN inline void chain_strand(const Ctx &c, long long rs, int &hdraw_total)
{
	const long long read = rs >> 1;
	const int rev = (int)(rs & 1);
	if (!c.active[read]) return;
	Strand &st = c.strand[rs];
	const uint32_t mem_i = st.mem_n;
	VMem *vm = c.mem.base + st.mem_off;
	const DevIndex &ix = c.idx;
	if (mem_i > 1) {
		VMem *mtmp = nullptr;
		if (mem_i > 16) {                                                   // merge-sort scratch from the bump region
			long long t = arena_alloc(c.mem, mem_i);
			if (t < 0) return;
			mtmp = c.mem.base + t;
		}
		stable_sort(vm, mtmp, mem_i, [](const VMem &a, const VMem &b) { return a.uid != b.uid ? a.uid < b.uid : a.read_pos < b.read_pos; });
	}
	// pass 0 sizes the seed list, pass 1 fills it
	long long ubase = 0;
	uint32_t total = 0;
	const long long hro = c.hoff[read] + hdraw_total;                     // the reverse strand continues the forward strand's draws
	int hdraw = 0;
	for (int pass = 0; pass < 2; ++pass) {
		uint32_t n = 0, vu_i = 0;
		hdraw = 0;
		bool stop = false;
		uint32_t j = 0;
		while (j < mem_i && !stop) {
			VU u;
			if (mem_i == 1) {
				const VMem &m = vm[0];
				u.uid = m.uid, u.read_pos = m.read_pos, u.uni_pos_off = m.uni_pos_off, u.length1 = u.length2 = m.length, u.pos_n = m.pos_n, u.cov = m.length;
				j = 1;
			} else {
				uint32_t s1 = j, cov = vm[s1].length;
				// NB: uni_id_temp is re-read from vertexm_v[j] after every group, so the uid test below compares
				// against the group's own first element
				const uint64_t uid_t = vm[s1].uid;
				j++;
				while (j < mem_i && uid_t == vm[j].uid && vm[j].uni_pos_off > vm[j - 1].uni_pos_off) {
					int diff = (int)(vm[j].read_pos - vm[j - 1].read_pos - vm[j - 1].length);
					if (diff > 3) break;
					int ce = (int)((vm[j].uni_pos_off - vm[j - 1].uni_pos_off) - (vm[j].read_pos - vm[j - 1].read_pos));
					if ((ce < 0 ? -ce : ce) < 1) { cov += (diff > 0) ? vm[j].length : (uint32_t)(diff + (int)vm[j].length); ++j; }
					else break;
				}
				uint32_t e1 = j - 1;
				u.uid = vm[s1].uid, u.read_pos = vm[s1].read_pos, u.uni_pos_off = vm[s1].uni_pos_off, u.pos_n = vm[s1].pos_n, u.cov = cov;
				if (s1 == e1) u.length1 = u.length2 = vm[s1].length;
				else {
					u.length1 = vm[e1].read_pos + vm[e1].length - vm[s1].read_pos;
					u.length2 = vm[e1].uni_pos_off + vm[e1].length - vm[s1].uni_pos_off;
				}
			}
			// expand_seed for this vertex_U (index vu_i)
			uint32_t cnt = u.pos_n;
			bool sample = false;
			if (u.pos_n > 500) {
				if (u.pos_n > 8000) { stop = true; break; }
				cnt = 500, sample = true;
			}
			for (uint32_t k = 0; k < cnt; ++k) {
				uint32_t m = k;
				if (sample) {
					long long q = hro + hdraw - c.hrand_base[read & 1];
					int32_t rv = (q >= 0 && q < c.hrand_n) ? c.hrand[read & 1][q] : (*c.err = 3, 0);
					m = (uint32_t)rv % u.pos_n;
					++hdraw;
				}
				if (pass == 1) {
					USeed &s = c.us.base[ubase + n];
					s.seed_id = vu_i, s.read_begin = u.read_pos, s.read_end = u.read_pos + u.length1 - 1;
					s.ref_begin = (uint32_t)(ix.pos[m + ix.posp[u.uid]] + u.uni_pos_off - 1);
					s.ref_end = s.ref_begin + u.length2 - 1, s.cov = u.cov;
				}
				++n;
			}
			++vu_i;
		}
		if (pass == 0) {
			total = n;
			if (n == 0) break;
			ubase = arena_alloc(c.us, 2ull * n);       // second half: merge-sort scratch
			if (ubase < 0) return;
		}
	}
	(void)rev;
	st.us_off = ubase, st.us_n = total;
	hdraw_total += hdraw;
	if (total == 0) return;
	USeed *v = c.us.base + ubase;
	PathN *pa = c.path + ubase;
	stable_sort(v, v + total, total, [](const USeed &a, const USeed &b) { return a.ref_end != b.ref_end ? a.ref_end < b.ref_end : a.ref_begin < b.ref_begin; });
	const bool is_str = c.is_str[read] != 0;
	const int max_ref_dis = is_str ? 400 : 50, max_read_dis = is_str ? 400 : 50;
	const uint32_t max_search_step = is_str ? 80 : 40, max_gap = is_str ? 20 : 50;
	const uint32_t n = total;
	const uint32_t search_step = n < max_search_step ? n : max_search_step;
	// forward scan per target: where would the reference's inner loop `break` (graph.cpp:86)?
	for (uint32_t t = 0; t < n; ++t) {
		pa[t].dist = (int32_t)v[t].cov, pa[t].pre_node = -1, pa[t].used = 0;
		uint32_t se = t + search_step < n ? t + search_step : n, brk = se;
		for (uint32_t y = t + 1; y < se; ++y) {
			if (v[y].seed_id == v[t].seed_id) continue;
			if (v[y].ref_end == v[t].ref_end) continue;
			if ((int32_t)(v[y].ref_begin - v[t].ref_end) > max_ref_dis) { brk = y; break; }
		}
		pa[t].brk = brk;
	}
	// dynamic_programming_path over the implicit pre_edge lists (ascending source id)
	for (uint32_t y = 1; y < n; ++y) {
		int32_t cur = 0, pn = -1;
		bool any = false;
		uint32_t t0 = y >= search_step ? y - search_step + 1 : 0;
		for (uint32_t t = t0; t < y; ++t) {
			if (y >= pa[t].brk) continue;
			if (v[y].seed_id == v[t].seed_id) continue;
			if (v[y].ref_end == v[t].ref_end) continue;
			int32_t dis_ref = (int32_t)(v[y].ref_begin - v[t].ref_end);
			int32_t dis_read = (int32_t)(v[y].read_begin - v[t].read_end);
			if (dis_read > max_read_dis) continue;
			uint32_t abs_gap = dis_read > dis_ref ? (uint32_t)(dis_read - dis_ref) : (uint32_t)(dis_ref - dis_read);
			if (abs_gap > max_gap) continue;
			int32_t penalty = abs_gap == 0 ? 0 : (int32_t)((abs_gap >> 3) + 3);
			uint32_t weight;
			if (dis_read == dis_ref) weight = v[y].cov - (uint32_t)((1 - dis_read) > 0 ? (1 - dis_read) : 0);
			else if (dis_read > 0 && dis_ref > 0) weight = v[y].cov;
			else if (dis_read >= -5 && dis_read <= 0 && dis_ref >= -5) weight = v[y].cov + (uint32_t)(dis_read < dis_ref ? dis_read : dis_ref);
			else continue;
			any = true;
			int32_t temp = pa[t].dist + (int32_t)weight - penalty;
			if (cur <= temp) cur = temp, pn = (int32_t)t;
		}
		if (any) pa[y].dist = cur, pa[y].pre_node = pn;
	}
	if (c.trace) {
		uint64_t hs = 1469598103934665603ULL, hd = hs;
		for (uint32_t i = 0; i < n; ++i) {
			hs = fnv1a(hs, v[i].read_begin); hs = fnv1a(hs, v[i].read_end); hs = fnv1a(hs, v[i].seed_id);
			hs = fnv1a(hs, v[i].ref_begin); hs = fnv1a(hs, v[i].ref_end); hs = fnv1a(hs, v[i].cov);
			hd = fnv1a(hd, (uint64_t)(int64_t)pa[i].dist); hd = fnv1a(hd, (uint64_t)(int64_t)pa[i].pre_node);
		}
		st.seed_hash = hs, st.chain_hash = hd;
	}
}

// both strands of one read, forward first: the per-handler random_r stream is consumed in that order
PSVR_HDN inline void chain_read(const Ctx &c, long long read)
{
	if (!c.active[read]) return;
	int hdraw = 0;
	chain_strand(c, read * 2, hdraw);
	chain_strand(c, read * 2 + 1, hdraw);
	c.hcnt[read] = hdraw;
	if (hdraw && c.any_h) *c.any_h = 1;                     // somebody sampled positions with random_r: the batch needs that stream's offsets (engine_core.h)
	if (c.stats) stat_add(c, ST_SEEDS, c.strand[read * 2].us_n + c.strand[read * 2 + 1].us_n);
}

PSVR_HD int get_chromosome_id(const DevIndex &ix, uint32_t position)   // deBGA_index.cpp:369-396
{
	int file_n = 0;
	int pos_index = position / 0x4000;
	int low = (int)ix.chr_search_index[pos_index];
	int high = (int)ix.chr_search_index[pos_index + 1];
	int pos = (int)position + 1;
	while (low <= high) {
		int mid = (low + high) >> 1;
		int e = (int)(ix.chr_end_n[mid] - 1);
		if (pos < e) high = mid - 1;
		else if (pos > e) low = mid + 1;
		else return mid;
		file_n = low;
	}
	return file_n;
}

// sort_output (rr.cpp:212-293), recursion unrolled into a loop; returns 1 and fills `out` on success
PSVR_HD int sort_output(const Ctx &c, const Strand &st, int direction, ChainCand &out, long long ro, int &draws)
{
	const uint32_t n = st.us_n;
	if (n == 0) return 0;
	const USeed *v = c.us.base + st.us_off;
	PathN *pa = c.path + st.us_off;
	for (;;) {
		uint32_t max_index = 0xffffffffu;
		int32_t max_distance = 0;
		uint32_t same = 1;                   // same_top list starts with the sentinel entry
		for (int i = (int)n - 1; i >= 0; i--) {
			if (pa[i].used) continue;
			int32_t d = pa[i].dist;
			if (max_distance < d) max_distance = d, max_index = (uint32_t)i, same = 1;
			else if (max_distance == d) same++;
		}
		if (max_index == 0xffffffffu) return 0;
		if (same > 1) {                      // pick same_top[rand() % same]: entry 0 is the first maximum found (or the sentinel)
			long long k = ro + draws - c.grand_base;
			int32_t r = (k >= 0 && k < c.grand_n) ? c.grand[k] : (*c.err = 2, 0);
			++draws;
			uint32_t pick = (uint32_t)r % same;
			// rebuild the list order: [first maximum (descending scan), then every later index with dist == max]
			uint32_t seen = 0;
			for (int i = (int)n - 1; i >= 0; i--) {
				if (pa[i].used) continue;
				if (pa[i].dist == max_distance) { if (seen == pick) { max_index = (uint32_t)i; break; } ++seen; }
			}
		}
		int used = 0, unused = 0;
		int first_node = (int)max_index, orig_first = first_node;
		for (; first_node != -1;) {
			if (pa[first_node].used) used++;
			else unused++;
			pa[first_node].used = 1;
			int nx = pa[first_node].pre_node;
			if (nx == -1) break;
			first_node = nx;
		}
		int orig_final = first_node;
		if (orig_first - orig_final > ((unused + used + 5) << 1))
			for (int k = orig_final; k < orig_first; k++) pa[k].used = 1;
		if (used >= unused) continue;        // `return sort_output(...)`
		uint32_t ref_begin = v[first_node].ref_begin;
		int chr = get_chromosome_id(c.idx, ref_begin);
		out.direction = direction;
		out.max_index = max_index;
		out.chain_score = (uint32_t)max_distance;
		out.read_bg = v[first_node].read_begin;
		out.chr_id = chr;
		out.ref_bg = ref_begin - c.idx.chr_end_n[chr - 1];
		return 1;
	}
}

// the chain-selection part of single_end_handler::align (rr.cpp:417-442)
PSVR_HDN inline void select_read(const Ctx &c, long long read)
{
	if (!c.active[read]) return;
	const long long item = (read >> 1) * 3 + (read & 1);
	const long long ro = c.poff[read >> 1] + ((read & 1) ? c.rcnt[(read >> 1) * 3] : 0) + c.rcnt[item];   // after the N draws of prep_read
	int draws = 0;
	ChainCand *cc = c.ccand + read * 12;
	int n = 0;
	uint32_t max_chain = 0;
	for (int o = 0; o < 2; ++o) {
		const int direction = o == 0 ? kFwd : kRev;
		for (int i = 0; i < kMaxOut; ++i) {
			ChainCand tmp;
			if (!sort_output(c, c.strand[read * 2 + o], direction, tmp, ro, draws)) break;
			uint32_t cs = tmp.chain_score;
			if (cs > max_chain) max_chain = cs;
			if (cs + 30 < max_chain || cs < 30) break;
			cc[n++] = tmp;
		}
	}
	// qsort by cmp_chain_score (rr.hpp:303-308): chain_score desc, max_index asc, stable
	for (int i = 1; i < n; ++i) {
		ChainCand x = cc[i];
		int j = i;
		while (j > 0 && (x.chain_score != cc[j - 1].chain_score ? x.chain_score > cc[j - 1].chain_score : x.max_index < cc[j - 1].max_index)) { cc[j] = cc[j - 1]; --j; }
		cc[j] = x;
	}
	c.rcnt[item] += draws;
	if (n == 0 || max_chain < 20) { c.n_ccand[read] = 0; return; }
	int keep = n;
	for (int k = 0; k < n; ++k) if (cc[k].chain_score + 30 < max_chain) { keep = k; break; }
	c.n_ccand[read] = keep;
}

// ---- chain + select, the small case ------------------------------------------------------------------------
// Nearly every read that has MEMs at all has them on ONE strand, in (uid, read_pos) order as the seed loop found them, and they merge
// into one or two unipath seeds with a single reference position each (a read that matches its anchor but for substitutions: one; with an
// indel: two; bench batch: 99.8 % of the reads with MEMs).  chain_read + select_read spend their time walking arrays of that size in
// global memory through loops written for hundreds of seeds.  Here the same rules -- merge_seed_in_unipath, expand_seed,
// Graph_handler::process / dynamic_programming_path (deBGA_index.cpp:151-251, graph.cpp:53-150), the sort_output loop and the candidate cut
// of single_end_handler::align (rr.cpp:212-293,417-442) -- run on at most two seeds held in registers, and the records the later stages
// read (USeed / PathN, Strand, ChainCand, counts) are written once.  Nothing is written before the read is known to fit; returns false
// then, and chain_read + select_read take the read.
PSVR_HDN inline bool chain_select_small(const Ctx &c, long long read)
{
	if (!c.active[read]) return true;
	const DevIndex &ix = c.idx;
	Strand &s0 = c.strand[read * 2], &s1 = c.strand[read * 2 + 1];
	const uint32_t n0 = s0.mem_n, n1 = s1.mem_n;
	if (n0 != 0 && n1 != 0) return false;
	const int o = n1 != 0;                                               // the strand that has the MEMs (0 forward, 1 reverse)
	Strand &st = o ? s1 : s0, &sx = o ? s0 : s1;
	const uint32_t mem_i = o ? n1 : n0;
	if (mem_i > 16) return false;
	struct SU { uint64_t uid; uint32_t read_pos, uni_pos_off, length1, length2, pos_n, cov; };
	SU u0, u1;
	u0.uid = u1.uid = 0, u0.read_pos = u1.read_pos = u0.uni_pos_off = u1.uni_pos_off = u0.length1 = u1.length1 = u0.length2 = u1.length2 = u0.pos_n = u1.pos_n = u0.cov = u1.cov = 0;
	int nvu = 0;
	if (mem_i) {
		const VMem *vm = c.mem.base + st.mem_off;
		// merge_seed_in_unipath over MEMs that stand in (uid, read_pos) order already (a stable sort leaves them where they are): a group runs on
		// while the next MEM lies in the same unipath, further right in it, at most 3 read bases behind its predecessor's end and on its diagonal
		VMem first = vm[0], last = first;
		uint32_t cov = first.length;
		for (uint32_t j = 1; j <= mem_i; ++j) {
			bool close = j == mem_i;
			VMem x = last;
			if (!close) {
				x = vm[j];
				if (x.uid != last.uid ? x.uid < last.uid : x.read_pos < last.read_pos) return false;      // not in sorted order: the generic path sorts
				bool on = x.uid == first.uid && x.uni_pos_off > last.uni_pos_off;
				if (on) {
					const int diff = (int)(x.read_pos - last.read_pos - last.length);
					const int ce = (int)((x.uni_pos_off - last.uni_pos_off) - (x.read_pos - last.read_pos));
					on = diff <= 3 && ce == 0;
					if (on) cov += diff > 0 ? x.length : (uint32_t)(diff + (int)x.length);
				}
				close = !on;
			}
			if (close) {
				if (nvu == 2 || first.pos_n != 1) return false;           // a third seed, or a unipath with several reference positions
				SU &u = nvu ? u1 : u0;
				u.uid = first.uid, u.read_pos = first.read_pos, u.uni_pos_off = first.uni_pos_off, u.pos_n = first.pos_n, u.cov = cov;
				u.length1 = last.read_pos + last.length - first.read_pos, u.length2 = last.uni_pos_off + last.length - first.uni_pos_off;   // (a group of one: its length)
				++nvu;
				first = x, cov = x.length;
			}
			last = x;
		}
	}
	// expand_seed: one position per seed
	USeed a, b;
	PathN pa, pb;
	const uint32_t n = (uint32_t)nvu;
	{
		const uint32_t r0 = n > 0 ? (uint32_t)(ix.pos[ix.posp[u0.uid]] + u0.uni_pos_off - 1) : 0u;
		const uint32_t r1 = n > 1 ? (uint32_t)(ix.pos[ix.posp[u1.uid]] + u1.uni_pos_off - 1) : 0u;
		a.seed_id = 0, a.read_begin = u0.read_pos, a.read_end = u0.read_pos + u0.length1 - 1, a.ref_begin = r0, a.ref_end = r0 + u0.length2 - 1, a.cov = u0.cov;
		b.seed_id = 1, b.read_begin = u1.read_pos, b.read_end = u1.read_pos + u1.length1 - 1, b.ref_begin = r1, b.ref_end = r1 + u1.length2 - 1, b.cov = u1.cov;
	}
	if (n == 2 && (b.ref_end != a.ref_end ? b.ref_end < a.ref_end : b.ref_begin < a.ref_begin)) { const USeed t = a; a = b, b = t; }   // sort by (ref_end, ref_begin), stable
	const bool is_str = c.is_str[read] != 0;
	const int max_ref_dis = is_str ? 400 : 50, max_read_dis = is_str ? 400 : 50;
	const uint32_t max_gap = is_str ? 20 : 50;
	// (search_step = min(n, 40 | 80) = n: both scans see the whole list)
	pa.dist = (int32_t)a.cov, pa.pre_node = -1, pa.used = 0, pa.brk = n;
	pb.dist = (int32_t)b.cov, pb.pre_node = -1, pb.used = 0, pb.brk = 2;
	if (n == 2) {
		// the two seeds come from different groups (seed_id 0 and 1)
		if (b.ref_end != a.ref_end && (int32_t)(b.ref_begin - a.ref_end) > max_ref_dis) pa.brk = 1;
		int32_t cur = 0, pn = -1;
		bool any = false;
		if (pa.brk > 1 && b.ref_end != a.ref_end) {
			const int32_t dis_ref = (int32_t)(b.ref_begin - a.ref_end), dis_read = (int32_t)(b.read_begin - a.read_end);
			const uint32_t abs_gap = dis_read > dis_ref ? (uint32_t)(dis_read - dis_ref) : (uint32_t)(dis_ref - dis_read);
			if (!(dis_read > max_read_dis) && !(abs_gap > max_gap)) {
				const int32_t penalty = abs_gap == 0 ? 0 : (int32_t)((abs_gap >> 3) + 3);
				uint32_t weight = 0;
				bool edge = true;
				if (dis_read == dis_ref) weight = b.cov - (uint32_t)((1 - dis_read) > 0 ? (1 - dis_read) : 0);
				else if (dis_read > 0 && dis_ref > 0) weight = b.cov;
				else if (dis_read >= -5 && dis_read <= 0 && dis_ref >= -5) weight = b.cov + (uint32_t)(dis_read < dis_ref ? dis_read : dis_ref);
				else edge = false;
				if (edge) {
					any = true;
					const int32_t temp = pa.dist + (int32_t)weight - penalty;
					if (cur <= temp) cur = temp, pn = 0;
				}
			}
		}
		if (any) pb.dist = cur, pb.pre_node = pn;
	}
	uint64_t hs = 1469598103934665603ULL, hd = hs;
	if (c.trace && n) {
		for (uint32_t i = 0; i < n; ++i) {
			const USeed &v = i ? b : a;
			const PathN &q = i ? pb : pa;
			hs = fnv1a(hs, v.read_begin); hs = fnv1a(hs, v.read_end); hs = fnv1a(hs, v.seed_id);
			hs = fnv1a(hs, v.ref_begin); hs = fnv1a(hs, v.ref_end); hs = fnv1a(hs, v.cov);
			hd = fnv1a(hd, (uint64_t)(int64_t)q.dist); hd = fnv1a(hd, (uint64_t)(int64_t)q.pre_node);
		}
	}
	// ---- the chain selection (select_read / sort_output) on these nodes; the other strand has none
	const long long item = (read >> 1) * 3 + (read & 1);
	const int32_t rc_item = c.rcnt[item];
	const long long ro = c.poff[read >> 1] + ((read & 1) ? c.rcnt[(read >> 1) * 3] : 0) + rc_item;
	int draws = 0, ncc = 0;
	uint32_t max_chain = 0;
	ChainCand c0, c1;
	c0.chain_score = c1.chain_score = 0, c0.max_index = c1.max_index = 0, c0.read_bg = c1.read_bg = 0, c0.ref_bg = c1.ref_bg = 0, c0.chr_id = c1.chr_id = 0, c0.direction = c1.direction = 0;
	for (int it = 0; it < kMaxOut && n; ++it) {
		// sort_output: the best unused chain end; tied ends draw
		bool got = false;
		ChainCand out;
		for (;;) {
			int maxi = -1, same = 1;
			int32_t maxd = 0;
			if (n == 2 && !pb.used) { const int32_t d = pb.dist; if (maxd < d) maxd = d, maxi = 1, same = 1; else if (maxd == d) same++; }
			if (!pa.used) { const int32_t d = pa.dist; if (maxd < d) maxd = d, maxi = 0, same = 1; else if (maxd == d) same++; }
			if (maxi < 0) break;
			if (same > 1) {
				const long long k = ro + draws - c.grand_base;
				const int32_t r = (k >= 0 && k < c.grand_n) ? c.grand[k] : (*c.err = 2, 0);
				++draws;
				const uint32_t pick = (uint32_t)r % (uint32_t)same;
				uint32_t seen = 0;
				bool done = false;
				if (n == 2 && !pb.used && pb.dist == maxd) { if (seen == pick) maxi = 1, done = true; else ++seen; }
				if (!done && !pa.used && pa.dist == maxd) { if (seen == pick) maxi = 0; }
			}
			int used = 0, unused = 0, node = maxi;
			for (;;) {
				PathN &q = node ? pb : pa;
				if (q.used) used++; else unused++;
				q.used = 1;
				if (q.pre_node == -1) break;
				node = q.pre_node;
			}
			// (orig_first - orig_final <= 1 here: the bulk marking of rr.cpp:284 never applies)
			if (used >= unused) continue;
			const USeed &v = node ? b : a;
			const int chr = get_chromosome_id(ix, v.ref_begin);
			out.direction = o == 0 ? kFwd : kRev, out.max_index = (uint32_t)maxi, out.chain_score = (uint32_t)maxd, out.read_bg = v.read_begin, out.chr_id = chr;
			out.ref_bg = v.ref_begin - ix.chr_end_n[chr - 1];
			got = true;
			break;
		}
		if (!got) break;
		const uint32_t cs = out.chain_score;
		if (cs > max_chain) max_chain = cs;
		if (cs + 30 < max_chain || cs < 30) break;
		if (ncc == 0) c0 = out; else c1 = out;
		++ncc;
	}
	if (ncc == 2 && (c1.chain_score != c0.chain_score ? c1.chain_score > c0.chain_score : c1.max_index < c0.max_index)) { const ChainCand t = c0; c0 = c1, c1 = t; }
	int keep = 0;
	if (!(ncc == 0 || max_chain < 20)) {
		keep = ncc;
		if (c0.chain_score + 30 < max_chain) keep = 0;
		else if (ncc == 2 && c1.chain_score + 30 < max_chain) keep = 1;
	}
	// ---- everything fits: write what chain_read + select_read would have left
	long long ubase = 0;
	if (n) {
		ubase = arena_alloc(c.us, 2ull * n);                               // (same footprint as the generic path: second half = its merge-sort scratch)
		if (ubase < 0) return true;                                        // arena full: flagged, the batch runs again with a larger one
		c.us.base[ubase] = a, c.path[ubase] = pa;
		if (n == 2) c.us.base[ubase + 1] = b, c.path[ubase + 1] = pb;
	}
	st.us_off = ubase, st.us_n = n;
	if (c.trace && n) st.seed_hash = hs, st.chain_hash = hd;
	sx.us_off = 0, sx.us_n = 0;
	c.hcnt[read] = 0;
	if (c.stats) stat_add(c, ST_SEEDS, n);
	ChainCand *cc = c.ccand + read * 12;
	if (ncc > 0) cc[0] = c0;
	if (ncc > 1) cc[1] = c1;
	c.rcnt[item] = rc_item + draws;
	c.n_ccand[read] = keep;
	return true;
}

PSVR_HD void get_refseq(const DevIndex &ix, uint8_t *ref, uint32_t len, uint32_t start)
{
	for (uint32_t m = 0; m < len; ++m) ref[m] = (uint8_t)base_at(ix.ref_seq, (uint64_t)m + start);
}

struct WalkState {
	const Ctx *c;
	const uint8_t *read_str;
	const uint64_t *read_w;      // the same strand 2-bit packed (exact unless the read holds a lower-case 'n', code 4)
	bool packed_ok;
	long long read; int strand; int k;
	int32_t read_score; uint32_t total_q_len;
	bool is_simple;
	Seg *seg; int n_seg; int bad; int seg_cap;
	int n_dp;                    // DP pieces queued by this read so far (local numbering until walk_read assigns the ids)
	// the most recent DP piece, kept in registers: a read with exactly one (the common case) gets its descriptor written from here
	Seg *last_seg; int last_q_st, last_qlen, last_ref_st, last_tlen, last_type, last_strand;
};

PSVR_HD void seg_lit(WalkState &w, int type, int size)
{
	if (w.n_seg >= w.seg_cap) { w.bad = 1; return; }
	Seg &s = w.seg[w.n_seg++];
	s.kind = 0, s.a = type, s.b = (int32_t)(int16_t)(uint16_t)size;   // CIGAR_PATH(char, uint16_t) -> int16_t size
}

// KSW_ALN_handler::get_misMatch (rr.cpp:893-908)
PSVR_HD int walk_mismatch(WalkState &w, int read_st, int read_ed, int ref_st, int ref_ed)
{
	uint32_t qlen = read_ed - read_st, tlen = ref_ed - ref_st;
	if (ref_ed < ref_st) tlen = 0, qlen += (ref_st - ref_ed);
	if (!(tlen < 1600)) { w.bad = 2; return 0; }
	int nm = 0;
	const uint32_t n = qlen < tlen ? qlen : tlen;   // (qlen == tlen on this call path; bases past tlen would be stale scratch in the reference)
	if (w.packed_ok) nm = mismatches_packed(w.read_w, (uint64_t)read_st, w.c->idx.ref_seq, (uint64_t)ref_st, (int)n, 0x7fffffff);
	else for (uint32_t i = 0; i < n; ++i) nm += w.read_str[read_st + i] != base_at(w.c->idx.ref_seq, (uint64_t)ref_st + i);
	return nm > 3 ? 3 : nm;
}

PSVR_HDN inline int stale_compare(const Ctx &c, long long read, int k, int strand, int read_st, int qlen, int ref_st, int tlen);

// KSW_ALN_handler::alignment (rr.cpp:910-986): simple pieces are scored here, DP pieces are queued
PSVR_HD void walk_alignment(WalkState &w, int read_st, int read_ed, int ref_st, int ref_ed, int type)
{
	const Ctx &c = *w.c;
	uint32_t qlen = read_ed - read_st, tlen = ref_ed - ref_st;
	if (ref_ed < ref_st) tlen = 0, qlen += (ref_st - ref_ed);
	if (!(tlen < 1600)) { w.bad = 2; return; }
	w.total_q_len += qlen;
	w.is_simple = false;
	uint32_t nm = 0;
	if (qlen == 0 || tlen == 0) {
		w.is_simple = true;
		nm = qlen + tlen;
	} else if (qlen == tlen || type != 2) {
		if (tlen < qlen) {
			// only a left extension clamped at reference position 0 gets here (a right extension's window is qlen + 30, an end-to-end
			// piece is compared when qlen == tlen): the reference goes on comparing against what its scratch buffer still holds
			nm = (uint32_t)stale_compare(c, w.read, w.k, w.strand, read_st, (int)qlen, ref_st, (int)tlen);
		} else if (w.packed_ok) {
			// the reference counts position-wise mismatches up to 6; a left extension compares the reversed sequences, i.e. the
			// read piece against the LAST qlen bases of the reference window
			nm = (uint32_t)mismatches_packed(w.read_w, (uint64_t)read_st, c.idx.ref_seq, (uint64_t)ref_st + (type == 0 ? tlen - qlen : 0), (int)qlen, 6);
		} else if (type == 0) {   // left extension compares the reversed sequences (reads with a lower-case 'n': the per-base bytes exist)
			for (uint32_t i = 0; i < qlen && nm < 6; ++i) nm += w.read_str[read_st + (qlen - 1 - i)] != base_at(c.idx.ref_seq, (uint64_t)ref_st + (tlen - 1 - i));
		} else {
			for (uint32_t i = 0; i < qlen && nm < 6; ++i) nm += w.read_str[read_st + i] != base_at(c.idx.ref_seq, (uint64_t)ref_st + i);
		}
		if (nm == 1 || (nm < 6 && ((nm << 3) < qlen))) w.is_simple = true;
	}
	if (w.is_simple) {
		stat_add(c, ST_SIMPLE, 1);
		if (qlen == 0 || tlen == 0) {
			if (nm != 0) {
				int s1 = c.par.gap_open + (int)(nm - 1) * c.par.gap_ex, s2 = c.par.gap_open2 + (int)(nm - 1) * c.par.gap_ex2;
				w.read_score -= s1 < s2 ? s1 : s2;
			}
		} else w.read_score += (int32_t)(qlen * c.par.match - nm * (c.par.match + c.par.mismatch));
		if (qlen == 0) seg_lit(w, 2, (int)tlen);
		else if (tlen == 0) seg_lit(w, 1, (int)qlen);
		else seg_lit(w, 0, (int)qlen);
		if (ref_ed < ref_st) seg_lit(w, 2, ref_ed - ref_st);
		return;
	}
	if ((long long)tlen * qlen > 1000000) {                              // align_non_splice's fake result (rr.cpp:874-887)
		seg_lit(w, 3, (int)tlen), seg_lit(w, 1, (int)qlen);              // emitted in the order the caller would push them
		if (type == 0) { Seg t = w.seg[w.n_seg - 1]; w.seg[w.n_seg - 1] = w.seg[w.n_seg - 2]; w.seg[w.n_seg - 2] = t; }
		if (type != 2) w.read_score += PSVR_KSW_NEG_INF;                  // ez.mqe after ksw_reset_extz
		return;
	}
	// the DP problem is queued here but numbered later: walk_read reserves the ids of all its read's problems with one
	// allocation (the queue's counter is the hot spot of this stage).  Until then the coordinates ride in two payload slots.
	if (w.n_seg + 3 > w.seg_cap) { w.bad = 1; return; }
	Seg *s = w.seg + w.n_seg;
	w.last_seg = s, w.last_q_st = read_st, w.last_qlen = (int)qlen, w.last_ref_st = ref_st, w.last_tlen = (int)tlen, w.last_type = type, w.last_strand = w.strand;
	s[0].kind = 1, s[0].a = w.n_dp++, s[0].b = type;
	s[1].kind = 2, s[1].a = read_st, s[1].b = (int32_t)qlen;
	s[2].kind = 2, s[2].a = ref_st, s[2].b = (int32_t)tlen;
	w.n_seg += 3;
	stat_add(c, ST_DP, 1);
	stat_add(c, ST_CELLS, (unsigned long long)qlen * tlen);
}

// get_ksw_score (rr.cpp:308-400) for candidate k of `read`
// CIGAR pieces a candidate may need: per chain node at most 'M' + alignment (+ negative D); + tail M + left extension.  A chain
// has at most us_n nodes (sizing by that bound avoids walking the chain twice).
PSVR_HD int walk_seg_cap(const Ctx &c, long long read, int k)
{
	const int is_rev = c.ccand[read * 12 + k].direction == kRev;
	const int n = (int)c.strand[read * 2 + is_rev].us_n;
	const int cap = 3 * n + 4 + 2 * (n + 1);              // + two payload slots per DP piece (at most one per node + the left extension)
	return cap > kSegMax ? kSegMax : cap;
}

// The traversal of get_ksw_score (rr.cpp:308-400) for candidate k of `read`: which pieces of the chain are checked for mismatches
// (get_misMatch), which are aligned (alignment: right extension first, end-to-end pieces, the left extension last) and which gap
// penalties apply.  Its control flow depends on the chain alone, not on what the sink answers, so the same traversal serves the walk
// proper (WalkSink) and the replay of a candidate's reference-window calls (StaleSink, below).
template <class Sink> PSVR_HD void chain_calls(const Ctx &c, long long read, int k, Sink &sk)
{
	const ChainCand &cc = c.ccand[read * 12 + k];
	const int is_rev = cc.direction == kRev;
	const Strand &st = c.strand[read * 2 + is_rev];
	const USeed *va = c.us.base + st.us_off;
	const PathN *dp = c.path + st.us_off;
	const int read_l = c.read_l[read];
	const int BIG = 0x7fffffff;
	int aln_read_begin = read_l, aln_read_end = read_l, aln_ref_begin = BIG, aln_ref_end = BIG;
	int last_aln_begin = read_l, last_ref_begin = BIG;
	for (int node = (int)cc.max_index; node != -1;) {
		int mrb = (int)va[node].read_begin, mre = (int)va[node].read_end, mfb = (int)va[node].ref_begin, mfe = (int)va[node].ref_end;
		aln_read_begin = aln_read_begin < mre ? aln_read_begin : mre;
		aln_ref_begin = aln_ref_begin < mfe ? aln_ref_begin : mfe;
		if (aln_read_begin <= aln_read_end) {
			if (aln_read_end < last_aln_begin) {
				int ml = last_aln_begin - aln_read_end;
				sk.mismatch(aln_read_end, aln_read_end + ml, last_ref_begin, last_ref_begin + ml);
			}
			last_aln_begin = aln_read_begin;
			if (aln_ref_end == BIG) {
				aln_ref_end = aln_ref_begin + (aln_read_end - aln_read_begin) + 30;
				sk.alignment(aln_read_begin, aln_read_end, aln_ref_begin, aln_ref_end, 1);
			} else sk.alignment(aln_read_begin, aln_read_end, aln_ref_begin, aln_ref_end, 2);
		} else {
			int dr = aln_read_end - aln_read_begin, df = aln_ref_end - aln_ref_begin;
			if (dr != df) sk.gap(df - dr);
		}
		aln_read_end = mrb, last_ref_begin = mfb, aln_ref_end = mfb;
		int nx = dp[node].pre_node;
		if (nx == -1) break;
		node = nx;
	}
	if (aln_read_end < last_aln_begin) {
		int ml = last_aln_begin - aln_read_end;
		sk.mismatch(aln_read_end, aln_read_end + ml, last_ref_begin, last_ref_begin + ml);
	}
	aln_read_begin = 0, aln_ref_begin = 0;
	if (aln_read_begin < aln_read_end) {
		aln_ref_begin = aln_ref_end - (aln_read_end - aln_read_begin) - 30;
		aln_ref_begin = aln_ref_begin > 0 ? aln_ref_begin : 0;
		sk.left(aln_read_begin, aln_read_end, aln_ref_begin, aln_ref_end);
	}
}

// ---- the reference's scratch buffer ------------------------------------------------------------
// KSW_ALN_handler keeps ONE 1600-byte buffer `tseq` per handler (one handler per mate at -t 1): every get_misMatch / alignment call
// writes the first tlen bytes (get_refseq; reversed for a left extension) and nothing else, rr.cpp:899,918-925.  When a left
// extension is clamped at reference position 0 and its window is shorter than the read piece (0 < tlen < qlen), the simple-compare
// loop (rr.cpp:939) reads positions tlen..qlen-1 of that buffer: what EARLIER calls left there -- position i holds the byte the
// latest call with tlen' > i wrote.  StaleSink replays the calls of one candidate onto a window [lo, lo + 64) of the buffer.
struct StaleSink {
	const Ctx *c;
	int lo;                      // window start; 64 positions, 2 bits each (reference codes are 0..3)
	uint64_t code_lo, code_hi;   // positions lo..lo+31, lo+32..lo+63
	uint64_t have;               // bit i: position lo + i written by a replayed call
	bool skip_left;              // the candidate under evaluation: its own left extension is the call being answered
	PSVR_HD void put(uint32_t ref_st, int tlen, bool rev)
	{
		const int hi = tlen < lo + 64 ? tlen : lo + 64;
		for (int i = lo; i < hi; ++i) {
			const uint64_t b = (uint64_t)base_at(c->idx.ref_seq, (uint64_t)ref_st + (uint64_t)(rev ? tlen - 1 - i : i));
			const int j = i - lo, sh = (j & 31) << 1;
			if (j < 32) code_lo = (code_lo & ~(3ull << sh)) | (b << sh);
			else code_hi = (code_hi & ~(3ull << sh)) | (b << sh);
			have |= 1ull << j;
		}
	}
	PSVR_HD static int tlen_of(int ref_st, int ref_ed, bool &ok) { const int t = ref_ed < ref_st ? 0 : ref_ed - ref_st; ok = t < 1600; return t; }
	PSVR_HD void mismatch(int, int, int ref_st, int ref_ed) { bool ok; const int t = tlen_of(ref_st, ref_ed, ok); if (ok) put((uint32_t)ref_st, t, false); }
	PSVR_HD void alignment(int, int, int ref_st, int ref_ed, int) { bool ok; const int t = tlen_of(ref_st, ref_ed, ok); if (ok) put((uint32_t)ref_st, t, false); }
	PSVR_HD void left(int, int, int ref_st, int ref_ed) { if (skip_left) return; bool ok; const int t = tlen_of(ref_st, ref_ed, ok); if (ok) put((uint32_t)ref_st, t, true); }
	PSVR_HD void gap(int) {}
	PSVR_HD int at(int i) const { const int j = i - lo, sh = (j & 31) << 1; return (int)(((j < 32 ? code_lo : code_hi) >> sh) & 3); }
};

// the compare of rr.cpp:939 for candidate k's clamped left extension: positions < tlen against the (reversed) window, the rest
// against the scratch buffer as the calls so far have left it.  Resolved here: what this read's own earlier calls wrote (candidates
// 0..k-1 in full, candidate k up to its left extension).  A position none of them reached keeps c.tseq_in (what earlier reads of
// this mate left: see engine_core.h) and is counted in c.stale_open.
PSVR_HDN inline int stale_compare(const Ctx &c, long long read, int k, int strand, int read_st, int qlen, int ref_st, int tlen)
{
	const bool bytes = c.has_n4[read] != 0;
	const uint8_t *rs = c.bin + (read * 2 + strand) * (long long)c.lmax;
	const uint64_t *rw = c.rb + (read * 2 + strand) * (long long)c.wmax;
	auto qrev = [&](int i) { const int p = read_st + (qlen - 1 - i); return bytes ? (int)rs[p] : base_at(rw, (uint64_t)p); };
	int nm = 0;
	for (int i = 0; i < tlen && nm < 6; ++i) nm += qrev(i) != base_at(c.idx.ref_seq, (uint64_t)ref_st + (uint64_t)(tlen - 1 - i));
	for (int lo = tlen; lo < qlen && nm < 6; lo += 64) {
		StaleSink sk;
		sk.c = &c, sk.lo = lo, sk.code_lo = sk.code_hi = 0, sk.have = 0;
		for (int kk = 0; kk <= k; ++kk) { sk.skip_left = kk == k; chain_calls(c, read, kk, sk); }
		const int hi = qlen < lo + 64 ? qlen : lo + 64;
		for (int i = lo; i < hi && nm < 6; ++i) {
			int b;
			if ((sk.have >> (i - lo)) & 1) b = sk.at(i);
			else {
				b = c.tseq_in ? (int)c.tseq_in[(read & 1) * 1600 + i] : 0;
				if (c.stale_open) *c.stale_open = 1;
			}
			nm += qrev(i) != b;
		}
	}
	return nm;
}

// candidate k of `read`; its CandWork slot and its slice of the piece arena were reserved by walk_read
struct DpLast { Seg *seg; int q_st, qlen, ref_st, tlen, type, strand; };
struct WalkSink {
	WalkState w;
	int unitig_mis, rba;
	PSVR_HD void mismatch(int read_st, int read_ed, int ref_st, int ref_ed) { unitig_mis += walk_mismatch(w, read_st, read_ed, ref_st, ref_ed); seg_lit(w, 0, read_ed - read_st); }
	PSVR_HD void alignment(int read_st, int read_ed, int ref_st, int ref_ed, int type) { walk_alignment(w, read_st, read_ed, ref_st, ref_ed, type); }
	PSVR_HD void left(int read_st, int read_ed, int ref_st, int ref_ed)
	{
		walk_alignment(w, read_st, read_ed, ref_st, ref_ed, 0);
		if (ref_ed > ref_st) rba = w.is_simple ? ref_ed - ref_st - 30 : ref_ed - ref_st;
	}
	PSVR_HD void gap(int dl)
	{
		const Ctx &c = *w.c;
		const int a = dl > 0 ? dl : -dl;
		const int s1 = c.par.gap_open + (a - 1) * c.par.gap_ex, s2 = c.par.gap_open2 + (a - 1) * c.par.gap_ex2;
		w.read_score -= s1 < s2 ? s1 : s2;
	}
};
PSVR_HDN inline int walk_candidate(const Ctx &c, long long read, int k, long long cwi, long long so, int seg_cap, int dp_first, DpLast &last)
{
	const ChainCand &cc = c.ccand[read * 12 + k];
	const int is_rev = cc.direction == kRev;
	const int read_l = c.read_l[read];
	WalkSink sk;
	WalkState &w = sk.w;
	w.c = &c, w.read_str = c.bin + (read * 2 + is_rev) * (long long)c.lmax, w.read = read, w.strand = is_rev, w.k = k;
	w.read_w = c.rb + (read * 2 + is_rev) * (long long)c.wmax, w.packed_ok = c.has_n4[read] == 0;
	w.read_score = 0, w.total_q_len = 0, w.is_simple = false, w.seg = c.seg.base + so, w.n_seg = 0, w.bad = 0, w.seg_cap = seg_cap, w.n_dp = dp_first;
	w.last_seg = nullptr;
	sk.unitig_mis = 0, sk.rba = 0;
	chain_calls(c, read, k, sk);
	w.read_score += (int32_t)((read_l - (int)w.total_q_len) * c.par.match);
	w.read_score -= sk.unitig_mis * (c.par.match + c.par.mismatch);
	CandWork &cw = c.cw.base[cwi];
	cw.read = (int32_t)read, cw.k = k, cw.n_seg = w.n_seg, cw.read_score = w.read_score, cw.rba = sk.rba, cw.bad = w.bad, cw.seg_off = so;
	if (w.bad) *c.err = 10 + w.bad;
	stat_add(c, ST_CAND, 1);
	if (w.last_seg) last.seg = w.last_seg, last.q_st = w.last_q_st, last.qlen = w.last_qlen, last.ref_st = w.last_ref_st, last.tlen = w.last_tlen, last.type = w.last_type, last.strand = w.last_strand;
	return w.n_dp;
}

// all candidates of one read, in three phases with an arena reservation in front of each of the last two, so that a backend can
// make the reservations where its lanes are together (the GPU kernel reserves per workgroup: the counters of the candidate and DP
// queues are single lines -- their ids are ranges the host plans on -- and an atomic per read on one line was the limit of this stage):
//   walk_sizes      candidates and piece slots the read needs
//   walk_body       (given the candidate and piece reservations) walks the candidates; returns the number of DP problems queued
//   walk_number_dp  (given the reservation of that many DP ids) writes their descriptors
struct WalkRead { int nc, total; long long cw0, so0; int n_dp; DpLast last; };
PSVR_HD void walk_sizes(const Ctx &c, long long read, WalkRead &wr)
{
	wr.nc = 0, wr.total = 0, wr.n_dp = 0, wr.last.seg = nullptr;
	if (read < 0 || !c.active[read]) return;
	const int nc = c.n_ccand[read];
	if (nc <= 0) return;
	int total = 0;
	for (int k = 0; k < nc; ++k) total += walk_seg_cap(c, read, k);
	wr.nc = nc, wr.total = total;
}
PSVR_HD void walk_body(const Ctx &c, long long read, WalkRead &wr, long long cw0, long long so)
{
	if (wr.nc <= 0) return;
	if (cw0 < 0 || so < 0) { c.n_ccand[read] = 0; wr.nc = 0; return; }       // an arena is full: the batch is run again with larger ones; until then this read has no candidates
	c.rh[read].cand_off = cw0;                       // candidate k of this read lives at cand[cw0 + k]
	wr.cw0 = cw0, wr.so0 = so;
	int n_dp = 0;
	for (int k = 0; k < wr.nc; ++k) {
		const int cap = walk_seg_cap(c, read, k);
		n_dp = walk_candidate(c, read, k, cw0 + k, so, cap, n_dp, wr.last);
		so += cap;
	}
	wr.n_dp = n_dp;
}
PSVR_HD void walk_number_dp(const Ctx &c, long long read, const WalkRead &wr, long long id0)
{
	if (wr.nc <= 0 || wr.n_dp == 0) return;
	const int nc = wr.nc, n_dp = wr.n_dp;
	const long long cw0 = wr.cw0;
	const DpLast &last = wr.last;
	if (n_dp == 1 && id0 >= 0 && last.seg) {
		// one piece (most reads that have any): everything its descriptor needs is still in registers -- no walk back through the
		// pieces just written (a chain of loads from memory this lane stored to a moment ago)
		DpDesc &d = c.dp.base[id0];
		d.read = (int32_t)read, d.strand = last.strand, d.q_st = last.q_st, d.qlen = last.qlen, d.ref_st = (uint32_t)last.ref_st, d.tlen = last.tlen, d.type = last.type, d.pad = 0;
		last.seg->a = (int32_t)id0;
		return;
	}
	long long so = wr.so0;
	for (int k = 0; k < nc; ++k) {
		const CandWork &cw = c.cw.base[cw0 + k];
		Seg *seg = c.seg.base + so;
		for (int i = 0; i < cw.n_seg; ++i) {
			if (seg[i].kind != 1) continue;
			if (id0 < 0) { c.cw.base[cw0 + k].bad = 1; *c.err = 11; seg[i].a = 0; continue; }
			DpDesc &d = c.dp.base[id0 + seg[i].a];
			const ChainCand &cc = c.ccand[read * 12 + k];
			d.read = (int32_t)read, d.strand = cc.direction == kRev, d.q_st = seg[i + 1].a, d.qlen = seg[i + 1].b, d.ref_st = (uint32_t)seg[i + 2].a, d.tlen = seg[i + 2].b;
			d.type = seg[i].b, d.pad = 0;
			seg[i].a = (int32_t)(id0 + seg[i].a);
		}
		so += walk_seg_cap(c, read, k);
	}
}
PSVR_HDN inline void walk_read(const Ctx &c, long long read)
{
	WalkRead wr;
	walk_sizes(c, read, wr);
	if (wr.nc <= 0) return;
	const long long cw0 = arena_alloc(c.cw, (unsigned long long)wr.nc);
	const long long so = arena_alloc(c.seg, (unsigned long long)wr.total);
	walk_body(c, read, wr, cw0, so);
	if (wr.nc <= 0 || wr.n_dp == 0) return;
	walk_number_dp(c, read, wr, arena_alloc(c.dp, (unsigned long long)wr.n_dp));
}

struct CigOp { uint8_t type; int16_t size; };

// CIGAR_PATH::try_merge (rr.hpp:159-178)
PSVR_HD bool cig_try_merge(CigOp &a, const CigOp &cp, int &bad)
{
	if (cp.size < 0) {
		if (cp.type != 2) { bad = 1; return true; }
		if (a.type == 0) { a.size = (int16_t)(a.size + cp.size); if (!(a.size > 0)) bad = 1; return true; }
		if (a.type == 2) { a.size = (int16_t)(a.size - cp.size); if (!(a.size > 0)) bad = 1; return true; }
		bad = 1;
		return true;
	}
	if (a.type == cp.type || cp.size == 0) { a.size = (int16_t)(a.size + cp.size); return true; }
	return false;
}

// second half of the per-candidate loop in single_end_handler::align (rr.cpp:445-452): score sum,
// cigar_tmp reconstruction in push order, reverseGIGAR (rr.hpp:277-301)
// in two parts with the reservation of the CIGAR words between them (see walk_read): assemble_compute builds the CIGAR, assemble_store
// writes it and the candidate record
struct AsmResult { CigOp out[kCigMax]; int n, first, bad; int32_t score; };
PSVR_HD void assemble_compute(const Ctx &c, long long cwi, AsmResult &ar)
{
	const CandWork &cw = c.cw.base[cwi];
	const Seg *seg = c.seg.base + cw.seg_off;
	int32_t score = cw.read_score;
	CigOp *out = ar.out;
	int n = 0, bad = cw.bad;
	bool have = false;
	CigOp back;
	back.type = 0, back.size = 0;
	// cigar_tmp in push order is seg[0..n_seg) with DP pieces expanded; reverseGIGAR walks it from the back
	for (int si = cw.n_seg - 1; si >= 0; --si) {
		const Seg &s = seg[si];
		if (s.kind == 2) continue;
		int cnt = 1;
		const psvr_extz_t *ez = nullptr;
		const uint32_t *cg = nullptr;
		if (s.kind == 1) {
			ez = c.dp_ez + s.a;
			cg = c.dp_cig + ez->cigar_off;
			cnt = ez->n_cigar;
			if (si == cw.n_seg - 1 || true) { /* score added once below */ }
		}
		for (int e = cnt - 1; e >= 0; --e) {
			CigOp op;
			if (s.kind == 0) op.type = (uint8_t)s.a, op.size = (int16_t)s.b;
			else {
				// pushed as i = n-1..0 for end-to-end / right extension, i = 0..n-1 for left extension (rr.cpp:971-984)
				int pi = s.b == 0 ? e : cnt - 1 - e;
				uint32_t b = cg[pi];
				op.type = (uint8_t)(b & 0xf), op.size = (int16_t)(b >> 4);
			}
			if (!have) { back = op, have = true; continue; }
			if (!cig_try_merge(back, op, bad)) {
				if (n < kCigMax) out[n++] = back; else bad = 1;
				back = op;
			}
		}
		if (s.kind == 1) score += s.b == 2 ? ez->score : ez->mqe;
	}
	if (have) { if (n < kCigMax) out[n++] = back; else bad = 1; }
	int first = 0;
	if (n > 0 && out[0].size == 0) first = 1;                      // cigar.erase(cigar.begin())
	ar.n = n, ar.first = first, ar.bad = bad, ar.score = score;
}
PSVR_HD void assemble_store(const Ctx &c, long long cwi, const AsmResult &ar, long long co)
{
	const CandWork &cw = c.cw.base[cwi];
	const CigOp *out = ar.out;
	const int n = ar.n, first = ar.first, bad = ar.bad;
	const int32_t score = ar.score;
	psvr_cand_t pc;                                  // built here, stored in one piece: cand[cwi] == cand[rh.cand_off + cw.k]
	const ChainCand cc = c.ccand[(long long)cw.read * 12 + cw.k];
	pc.align_score = score > 0 ? (uint32_t)score : 0;
	pc.chain_score = cc.chain_score;
	pc.ref_bg = cc.ref_bg - (uint32_t)cw.rba;
	pc.read_bg = cc.read_bg;
	pc.chr_id = cc.chr_id, pc.sv_id = -1;
	pc.max_index = cc.max_index;
	pc.direction = (uint8_t)cc.direction, pc.mapq = 0;
	for (int i = 0; i < 6; ++i) pc.reserved[i] = 0;
	const int m = n - first;
	pc.n_cigar = 0, pc.cigar_off = 0;
	if (co >= 0) {
		for (int i = 0; i < m; ++i) c.cig.base[co + i] = ((uint32_t)(uint16_t)out[first + i].size << 4) | out[first + i].type;
		pc.n_cigar = (uint32_t)m, pc.cigar_off = co;
	}
	c.cand[cwi] = pc;
	if (bad) *c.err = 20;   // the reference would xassert (abort) or print "ERROR cigar"
}
PSVR_HDN inline void assemble_candidate(const Ctx &c, long long cwi)
{
	AsmResult ar;
	assemble_compute(c, cwi, ar);
	const int m = ar.n - ar.first;
	assemble_store(c, cwi, ar, arena_alloc(c.cig, (unsigned long long)(m > 0 ? m : 0)));
}

// PE_score's view of one alignment of a read: a candidate record or the original alignment (rr.hpp:476-534)
struct PeItem { uint32_t align_score, chr_id, ref_bg; int32_t direction, is_ori, sv_id, end_offset; };
PSVR_HD PeItem pe_item_of(const psvr_cand_t &d, int32_t end_offset)
{
	PeItem p;
	p.align_score = d.align_score, p.chr_id = (uint32_t)d.chr_id, p.ref_bg = d.ref_bg, p.direction = d.direction, p.is_ori = 0, p.sv_id = d.sv_id, p.end_offset = end_offset;
	return p;
}

// rest of single_end_handler::align (rr.cpp:453-475)
// `rr`: the read's header -- c.rh[read] itself, or a copy in registers that the caller stores in one piece (k_finalize_pair);
// only its cand_off (set by walk_read) is read here
// `items` (optional, three entries): the pairing stage's view of the read's first three results, handed on in registers -- the
// pairing would otherwise load the records this function has just stored, one dependent round trip per item it looks at.
// Up to three candidates (all but a handful of reads) are loaded together, sorted and finished in registers and stored once; the
// stages around this one are bound by the number of scattered 64-byte sectors they touch and by chains of dependent loads.
PSVR_HD void finalize_read(const Ctx &c, long long read, psvr_read_hdr_t &rr, PeItem *items = nullptr)
{
	rr.unmapped = c.unmapped[read], rr.early_out = !c.active[read], rr.is_str = c.is_str[read], rr.reserved = 0, rr.reserved2 = 0;
	rr.primary = rr.secondary = -1, rr.has_mate = 0, rr.mate_chr_id = 0, rr.mate_ref_bg = 0, rr.prim_sv_id = rr.mate_sv_id = -1;
	int n = c.active[read] ? c.n_ccand[read] : 0;
	if (n <= 0) rr.cand_off = 0;                      // walk_read set it for the reads that have candidates
	psvr_cand_t *cd = c.cand + rr.cand_off;
	auto before = [](const psvr_cand_t &x, const psvr_cand_t &y) { return x.align_score != y.align_score ? x.align_score > y.align_score : x.max_index < y.max_index; };   // cmp_align_score (rr.hpp:310-315)
	if (n <= 3) {
		psvr_cand_t x0 = {}, x1 = {}, x2 = {};
		if (n > 0) x0 = cd[0];
		if (n > 1) x1 = cd[1];
		if (n > 2) x2 = cd[2];
		// (field by field: a copy of the whole struct -- its reserved[] bytes are an array -- keeps all three records in scratch memory)
		auto swap = [](psvr_cand_t &a, psvr_cand_t &b) {
#define PSVR_SW(f) { const auto t = a.f; a.f = b.f; b.f = t; }
			PSVR_SW(align_score) PSVR_SW(chain_score) PSVR_SW(ref_bg) PSVR_SW(read_bg) PSVR_SW(chr_id) PSVR_SW(sv_id) PSVR_SW(max_index) PSVR_SW(n_cigar) PSVR_SW(cigar_off)
			PSVR_SW(direction) PSVR_SW(mapq)                                  // (reserved[]: zero in every record)
#undef PSVR_SW
		};
		bool moved = false;                                                // the stable insertion sort on three registers
		if (n > 1 && before(x1, x0)) { swap(x0, x1); moved = true; }
		if (n > 2 && before(x2, x1)) {
			swap(x1, x2);
			moved = true;
			if (before(x1, x0)) swap(x0, x1);
		}
		const int n_in = n;
		if (n > 0 && x0.align_score < 40) n = 0;
		const uint32_t second = n > 1 ? x1.align_score : 0;
		SvDev s0 = {}, s1 = {}, s2 = {};
		if (n > 0) s0 = c.idx.sv[x0.sv_id < 0 ? x0.chr_id : x0.sv_id];
		if (n > 1) s1 = c.idx.sv[x1.sv_id < 0 ? x1.chr_id : x1.sv_id];
		if (n > 2) s2 = c.idx.sv[x2.sv_id < 0 ? x2.chr_id : x2.sv_id];
		auto finish = [&](psvr_cand_t &x, const SvDev &sd, bool first) {
			if (x.sv_id < 0) {                                              // (a record that went through here before -- reselect_pair puts such
				x.sv_id = x.chr_id;                                         // records into a new list -- has its genome coordinates already)
				x.chr_id = (int32_t)sd.chr_id;
				x.ref_bg += sd.st_pos;
				if (x.ref_bg >= 0x7fffffffu) x.ref_bg = 5;
			}
			x.mapq = 0;
			if (first) { const int32_t d = (int32_t)(x.align_score - second); x.mapq = (uint8_t)(d > 40 ? 40 : d); }
		};
		if (n > 0) finish(x0, s0, true);
		if (n > 1) finish(x1, s1, false);
		if (n > 2) finish(x2, s2, false);
		const int n_st = n > 0 ? n : (moved ? n_in : 0);                    // (a list that is dropped keeps its sorted order in memory, as before)
		if (n_st > 0) cd[0] = x0;
		if (n_st > 1) cd[1] = x1;
		if (n_st > 2) cd[2] = x2;
		if (items) items[0] = pe_item_of(x0, s0.end_offset), items[1] = pe_item_of(x1, s1.end_offset), items[2] = pe_item_of(x2, s2.end_offset);
		rr.n_result = n;
		return;
	}
	for (int i = 1; i < n; ++i) {                                        // stable
		psvr_cand_t x = cd[i];
		int j = i;
		while (j > 0 && before(x, cd[j - 1])) { cd[j] = cd[j - 1]; --j; }
		cd[j] = x;
	}
	if (n > 0 && cd[0].align_score < 40) n = 0;
	for (int i = 0; i < n; ++i) {                                        // a record is loaded, changed and stored as a whole
		psvr_cand_t x = cd[i];
		if (x.sv_id < 0) {
			const int sv = x.chr_id;
			const SvDev &sd = c.idx.sv[sv];
			x.sv_id = sv;
			x.chr_id = (int32_t)sd.chr_id;
			x.ref_bg += sd.st_pos;
			if (x.ref_bg >= 0x7fffffffu) x.ref_bg = 5;
		}
		x.mapq = 0;
		if (i == 0) {
			const int32_t d = (int32_t)(x.align_score - (n > 1 ? cd[1].align_score : 0));
			x.mapq = (uint8_t)(d > 40 ? 40 : d);
		}
		cd[i] = x;
	}
	rr.n_result = n;
	if (items) {                                                         // (static indices: the items stay in registers)
		const psvr_cand_t &d0 = c.cand[rr.cand_off], &d1 = c.cand[rr.cand_off + 1], &d2 = c.cand[rr.cand_off + 2];
		if (n > 0) items[0] = pe_item_of(d0, c.idx.sv[d0.sv_id].end_offset);
		if (n > 1) items[1] = pe_item_of(d1, c.idx.sv[d1.sv_id].end_offset);
		if (n > 2) items[2] = pe_item_of(d2, c.idx.sv[d2.sv_id].end_offset);
	}
}
PSVR_HDN inline void finalize_read(const Ctx &c, long long read) { finalize_read(c, read, c.rh[read]); }

// PE_score::read_get_best_pairing_results + set_primary_secondary_mate (rr.hpp:476-534)
PSVR_HD PeItem pe_item(const Ctx &c, long long read, int i, const psvr_read_hdr_t &rr)
{
	if (i < rr.n_result) {
		const psvr_cand_t &d = c.cand[rr.cand_off + i];
		return pe_item_of(d, c.idx.sv[d.sv_id].end_offset);
	}
	PeItem p;
	const psvr_ori_t &o = c.ori[src_read(c, read)];
	p.align_score = o.align_score, p.chr_id = (uint32_t)o.chr_id, p.ref_bg = o.ref_bg >= 0x7fffffffu ? 1u : o.ref_bg, p.direction = o.direction, p.is_ori = 1, p.sv_id = -1, p.end_offset = 0;
	return p;
}

// h0 / h1: the two reads' headers (c.rh[2 pair], c.rh[2 pair + 1] or copies in registers, see finalize_read)
// pre0 / pre1 (optional, three entries each): the reads' first three result items as finalize_read left them in registers.  Without them
// they are loaded here, together and before the loops -- every combination the loops look at was a pair of dependent loads (candidate
// record, then its SV's end offset) in the middle of the pairing.
PSVR_HD void pair_reads(const Ctx &c, long long pair, psvr_read_hdr_t &h0, psvr_read_hdr_t &h1, const PeItem *pre0 = nullptr, const PeItem *pre1 = nullptr)
{
	const long long r0 = pair * 2, r1 = r0 + 1;
	const long long item = pair * 3 + 2;
	for (int e = 0; e < 2; ++e) {       // re-runnable: the pairing stage alone is repeated when only its draw offset moved
		psvr_read_hdr_t &rr = e == 0 ? h0 : h1;
		rr.primary = rr.secondary = -1, rr.has_mate = 0, rr.mate_chr_id = 0, rr.mate_ref_bg = 0, rr.prim_sv_id = rr.mate_sv_id = -1;
	}
	const long long ro = c.poff[pair] + c.rcnt[pair * 3] + c.rcnt[pair * 3 + 1];
	int draws = 0;
	const int max_isize = c.par.isize_max + 200;
	int min_isize = c.par.isize_min - 200;
	if (min_isize < 0) min_isize = 0;
	const int nrl = c.par.normal_read_length;
	int max_same = 1, max_score = 0, cur_isize = 0, m1 = -1, m2 = -1;   // -1 NULL, else item index
	bool proper = false;
	int n0 = h0.n_result, n1 = h1.n_result;
	const int nr0 = n0, nr1 = n1;
	if (!c.unmapped[r0]) n0++;
	if (!c.unmapped[r1]) n1++;
	PeItem q0[3], q1[3], o0, o1;
	for (int k = 0; k < 3; ++k) q0[k] = PeItem(), q1[k] = PeItem();
	o0 = PeItem(), o1 = PeItem();
	if (pre0) { q0[0] = pre0[0], q0[1] = pre0[1], q0[2] = pre0[2]; }
	else { if (nr0 > 0) q0[0] = pe_item(c, r0, 0, h0); if (nr0 > 1) q0[1] = pe_item(c, r0, 1, h0); if (nr0 > 2) q0[2] = pe_item(c, r0, 2, h0); }
	if (pre1) { q1[0] = pre1[0], q1[1] = pre1[1], q1[2] = pre1[2]; }
	else { if (nr1 > 0) q1[0] = pe_item(c, r1, 0, h1); if (nr1 > 1) q1[1] = pe_item(c, r1, 1, h1); if (nr1 > 2) q1[2] = pe_item(c, r1, 2, h1); }
	if (n0 > nr0) o0 = pe_item(c, r0, nr0, h0);                          // the original alignments
	if (n1 > nr1) o1 = pe_item(c, r1, nr1, h1);
	auto item0 = [&](int i) { return i >= nr0 ? o0 : i == 0 ? q0[0] : i == 1 ? q0[1] : i == 2 ? q0[2] : pe_item(c, r0, i, h0); };
	auto item1 = [&](int j) { return j >= nr1 ? o1 : j == 0 ? q1[0] : j == 1 ? q1[1] : j == 2 ? q1[2] : pe_item(c, r1, j, h1); };
	auto get_isize = [&](int p1, int p2, int d1, int d2) {
		if (d1 == d2) return 0;
		int is = nrl + ((d1 == kFwd) ? (p2 - p1) : (p1 - p2));
		return (is < max_isize && is > min_isize) ? is : 0;
	};
	auto store = [&](int i, int j) {   // i / j = -1 for NULL
		PeItem a = PeItem(), b = PeItem();
		if (i >= 0) a = item0(i);
		if (j >= 0) b = item1(j);
		int ISIZE = 0;
		if (i >= 0 && j >= 0 && a.chr_id == b.chr_id) {
			int s1p1 = (int)a.ref_bg, s1p2 = s1p1 + (a.is_ori ? 0 : a.end_offset);
			int s2p1 = (int)b.ref_bg, s2p2 = s2p1 + (b.is_ori ? 0 : b.end_offset);
			int is;
			if ((is = get_isize(s1p1, s2p1, a.direction, b.direction)) > 0) ISIZE = is;
			else if ((is = get_isize(s1p1, s2p2, a.direction, b.direction)) > 0) ISIZE = is;
			else if ((is = get_isize(s1p2, s2p1, a.direction, b.direction)) > 0) ISIZE = is;
			else if ((is = get_isize(s1p2, s2p2, a.direction, b.direction)) > 0) ISIZE = is;
		}
		int basic = (i >= 0 ? (int)a.align_score : 0) + (j >= 0 ? (int)b.align_score : 0);
		bool one_new = (i >= 0 && !a.is_ori) || (j >= 0 && !b.is_ori);
		int fin = basic + (ISIZE > 0 ? 0 : -60) + (one_new ? 0 : 1);
		if (fin >= max_score) {
			bool st = true;
			if (fin > max_score) max_same = 1;
			else {
				max_same++;
				long long k = ro + draws - c.grand_base;
				int32_t r = (k >= 0 && k < c.grand_n) ? c.grand[k] : (*c.err = 2, 0);
				++draws;
				if (r % max_same != 0) st = false;
			}
			if (st) m1 = i, m2 = j, max_score = fin, cur_isize = ISIZE, proper = cur_isize > 0;
		}
	};
	for (int i = 0; i < n0; i++) store(i, -1);
	for (int j = 0; j < n1; j++) store(-1, j);
	for (int i = 0; i < n0; i++) for (int j = 0; j < n1; j++) store(i, j);
	c.rcnt[item] = draws;
	const bool a_new = m1 >= 0 && m1 < nr0, b_new = m2 >= 0 && m2 < nr1;
	const bool gain = max_score > 0 && (a_new || b_new);
	psvr_pair_result_t pr;
	pr.max_score = max_score, pr.cur_isize = cur_isize, pr.proper = proper, pr.gain = gain;
	pr.max1 = m1 < 0 ? -1 : (m1 < nr0 ? m1 : -2);
	pr.max2 = m2 < 0 ? -1 : (m2 < nr1 ? m2 : -2);
	c.pres[pair] = pr;
	if (!gain) return;
	for (int e = 0; e < 2; ++e) {                                        // set_primary_secondary_mate
		const int mine = e == 0 ? m1 : m2, other = e == 0 ? m2 : m1;
		const int nmine = e == 0 ? nr0 : nr1, nother = e == 0 ? nr1 : nr0;
		if (mine < 0) continue;
		psvr_read_hdr_t &rr = e == 0 ? h0 : h1;
		const bool is_ori = mine >= nmine;
		rr.primary = is_ori ? -2 : mine;
		rr.secondary = -1;
		if (is_ori && nmine > 0) rr.secondary = 0;
		else if (nmine > 1) rr.secondary = mine == 0 ? 1 : 0;          // rst_idx == position after the final sort
		const PeItem me = e == 0 ? item0(mine) : item1(mine);
		rr.prim_sv_id = me.sv_id;
		if (other >= 0) {
			const PeItem mt = e == 0 ? item1(other) : item0(other);
			(void)nother;
			if (mt.chr_id != 0xffffffffu) {
				// The reference handles read 0 first and stores the SV of an ORIGINAL primary in the shared `ori` object
				// (`c_rst->sv_info_p = c_rst->mate_sv_info_p`, rr.hpp:524-526), so when read 1's turn comes an original mate already
				// carries the SV it inherited -- from read 1 itself.  Read 0 looks at read 1's original before that assignment.
				const int32_t mate_sv = (e == 1 && mt.is_ori) ? h0.prim_sv_id : mt.sv_id;
				rr.has_mate = 1, rr.mate_chr_id = (int32_t)mt.chr_id, rr.mate_ref_bg = mt.ref_bg, rr.mate_sv_id = mate_sv;
				if (is_ori) rr.prim_sv_id = mate_sv;
				continue;
			}
		}
		rr.has_mate = 0, rr.mate_chr_id = 0, rr.mate_sv_id = -1;
	}
}
PSVR_HDN inline void pair_reads(const Ctx &c, long long pair) { pair_reads(c, pair, c.rh[pair * 2], c.rh[pair * 2 + 1]); }

// the fixed-size ABI record of one read (psvr_read_result_t) from the compact header, the candidate list and the strand
// bookkeeping; `out` is written completely (unused candidate slots are zero)
PSVR_HD void materialize_read(const Ctx &c, long long read, psvr_read_result_t *out)
{
	const psvr_read_hdr_t &h = c.rh[read];
	psvr_read_result_t &o = *out;
	o.n_result = h.n_result, o.unmapped = h.unmapped, o.early_out = h.early_out, o.is_str = h.is_str, o.reserved = 0;
	o.primary = h.primary, o.secondary = h.secondary, o.has_mate = h.has_mate, o.mate_chr_id = h.mate_chr_id, o.mate_ref_bg = h.mate_ref_bg;
	o.prim_sv_id = h.prim_sv_id, o.mate_sv_id = h.mate_sv_id, o.reserved1 = 0;
	for (int s = 0; s < 2; ++s) {
		const Strand &st = c.strand[read * 2 + s];
		o.n_seed[s] = st.us_n, o.seed_hash[s] = st.seed_hash, o.chain_hash[s] = st.chain_hash;
	}
	uint32_t *w = (uint32_t *)o.cand;
	const uint32_t *src = (const uint32_t *)(c.cand + h.cand_off);
	const int nw = (int)(sizeof(psvr_cand_t) / 4), have = h.n_result * nw;
	for (int i = 0; i < PSVR_MAX_RESULT * nw; ++i) w[i] = i < have ? src[i] : 0u;
}

// which pairs consumed draws from an offset that the scan of the actual draw counts has since moved?
// Also adopts the new offsets.  (engine_core.h, "rand() order")  Real pairs only.
PSVR_HD int mark_dirty(const Ctx &c, long long pair, const long long *noff, const long long *nhoff, const uint8_t *has_n)
{
	int dirty = 0;                   // 0 clean, 1 only the pairing stage must be repeated, 2 the whole pair, 3 chain selection + pairing (reselect_pair)
	// (everything is requested before anything is looked at: a dozen words per pair, one round trip instead of five in a row)
	const long long po = c.poff[pair], no = noff[pair];
	const int32_t r0 = c.rcnt[pair * 3], r1 = c.rcnt[pair * 3 + 1], r2 = c.rcnt[pair * 3 + 2];
	const int32_t h0 = c.hcnt[pair * 2], h1 = c.hcnt[pair * 2 + 1];
	const long long ho0 = c.hoff[pair * 2], ho1 = c.hoff[pair * 2 + 1], nh0 = nhoff[pair * 2], nh1 = nhoff[pair * 2 + 1];
	if (po != no) {
		// draws of a read without N bases are tie draws of its chain selection (rr.cpp:247): everything in front of the selection
		// stands whatever the offset is
		if (r0 > 0 || r1 > 0) dirty = (has_n && !has_n[pair]) ? 3 : 2;
		else if (r2 > 0) dirty = 1;
		c.poff[pair] = no;
	}
	if ((h0 > 0 && ho0 != nh0) || (h1 > 0 && ho1 != nh1)) dirty = 2;
	c.hoff[pair * 2] = nh0, c.hoff[pair * 2 + 1] = nh1;
	return dirty;
}

// A pair whose reads hold no N (or a variant slot, whose N draws are forced): its other draws in front of the pairing stage are the tie
// draws of sort_output.  At another offset the ties are resolved by other values, which changes the ORDER in which tied chains are taken
// -- and hardly ever the candidate list that comes out (it is sorted by score and chain index afterwards).  So the selection alone runs
// again at the new offset.  Returns
//   1  both reads' lists are what they were: every later stage's result stands, only the pairing has to follow;
//   3  a list changed, but every chain on it was a candidate before (a chain is named by its last node and strand, and everything
//      the later stages make of it depends on the chain alone): the read's candidate records are put together from the old ones
//      and its tail (finalize_read) runs again here; only the pairing has to follow;
//   2  a list holds a chain that was not evaluated before: the pair goes on from the walk.
PSVR_HDN inline int reselect_pair(const Ctx &c, long long pair)
{
	int result = 1;
	for (int mate = 0; mate < 2; ++mate) {
		const long long read = pair * 2 + mate, item = pair * 3 + mate;
		if (!c.active[read]) continue;
		const int n_old = c.n_ccand[read];
		ChainCand *cc = c.ccand + read * 12;
		ChainCand save[12];
		for (int i = 0; i < n_old && i < 12; ++i) save[i] = cc[i];
		for (int o = 0; o < 2; ++o) {                                     // sort_output marks the chains it has taken
			const Strand &st = c.strand[read * 2 + o];
			PathN *pa = c.path + st.us_off;
			for (uint32_t i = 0; i < st.us_n; ++i) pa[i].used = 0;
		}
		c.rcnt[item] = c.force ? (int32_t)c.force[4 * read] : 0;         // the N draws (forced in a variant slot, none in a pair without N); select_read adds its tie draws
		select_read(c, read);
		const int n_new = c.n_ccand[read];
		bool same = n_new == n_old;
		for (int i = 0; i < n_old && same; ++i) {
			const ChainCand &a = save[i], &b = cc[i];
			if (a.chain_score != b.chain_score || a.max_index != b.max_index || a.read_bg != b.read_bg || a.ref_bg != b.ref_bg || a.chr_id != b.chr_id || a.direction != b.direction) same = false;
		}
		if (same) continue;
		if (result == 2) continue;                                        // (the other read's selection has to be repeated all the same)
		// every chain of the new list among the old candidate records (walk_read wrote n_old of them at rh.cand_off; finalize_read has sorted them since)?
		psvr_cand_t *cd = c.cand + c.rh[read].cand_off;
		psvr_cand_t old[12];
		for (int i = 0; i < n_old && i < 12; ++i) old[i] = cd[i];
		bool all = n_new <= n_old;
		int where[12];
		for (int j = 0; j < n_new && all; ++j) {
			where[j] = -1;
			for (int i = 0; i < n_old; ++i) if (old[i].max_index == cc[j].max_index && old[i].direction == (uint8_t)cc[j].direction) where[j] = i;
			if (where[j] < 0) all = false;
		}
		if (!all) { result = 2; continue; }
		for (int j = 0; j < n_new; ++j) cd[j] = old[where[j]];
		finalize_read(c, read, c.rh[read]);                               // sort, threshold, mapq again; coordinates stay (sv_id >= 0)
		result = 3;
	}
	return result;
}

// A pair with N bases takes over the records of the variant slot that was evaluated with exactly the residues its draws yield
// (engine_core.h: only when that slot's two reads drew nothing else, so their results do not depend on where in the stream they stand).
// `part` of `parts` workers copy the two read records word by word; worker 0 also moves the counters and the offset.
// A slot whose PAIRING stage drew (tied pair scores, rr.hpp:553) is adopted as well: those draws follow the reads' and their number
// does not depend on their values (a tie leaves max_score where it is), so worker 0 takes the slot's read records and runs the pairing
// again where the pair really stands in the stream -- a third of the pairs with N bases, which used to run again from their first stage.
// Declined (nothing touched) when the slot sampled positions with random_r: that result depends on the other streams' offsets.
PSVR_HDN inline void adopt_variant(const Ctx &c, long long pair, long long slot, const long long *noff, int part, int parts)
{
	if (c.hcnt[2 * slot] != 0 || c.hcnt[2 * slot + 1] != 0) return;
	const int32_t s0 = c.rcnt[3 * slot], s1 = c.rcnt[3 * slot + 1];
	const bool ties = c.force && (s0 != (int32_t)c.force[4 * (2 * slot)] || s1 != (int32_t)c.force[4 * (2 * slot + 1)]);   // a read of the slot drew for tied chains too
	const bool repair = ties || c.rcnt[3 * slot + 2] != 0;
	if (repair) {
		if (part != 0) return;
		if (ties) {
			// the slot's chain selection again where the pair stands (reselect_pair); adopted only if the candidates stand and the
			// number of draws is what the host's walk counted on -- otherwise nothing is touched and the pair runs in full
			c.poff[slot] = noff[pair];
			const int r = reselect_pair(c, slot);
			if (r == 2 || c.rcnt[3 * slot] != s0 || c.rcnt[3 * slot + 1] != s1) {
				// Declined: the pair runs in full.  reselect_pair has rewritten the slot's chain selection (and, on result 3, its candidate
				// records), so the slot is not what the host's table of variant counts (read once per batch) describes any more: its
				// counts go back to what that table holds -- special_slot_at and the host's walk keep counting on them -- and the slot is
				// closed for further adoptions (hcnt != 0 declines above; nothing else looks at a variant slot's hcnt after round 1),
				// so no later attempt compares against, or adopts, the changed records
				c.rcnt[3 * slot] = s0, c.rcnt[3 * slot + 1] = s1;
				c.hcnt[2 * slot] = -1;
				return;
			}
		}
		for (int k = 0; k < 4; ++k) c.strand[4 * pair + k] = c.strand[4 * slot + k];
		c.rcnt[3 * pair] = c.rcnt[3 * slot], c.rcnt[3 * pair + 1] = c.rcnt[3 * slot + 1];
		c.hcnt[2 * pair] = c.hcnt[2 * pair + 1] = 0;
		c.poff[pair] = noff[pair];
		psvr_read_hdr_t h0 = c.rh[2 * slot], h1 = c.rh[2 * slot + 1];   // (cand_off keeps pointing at the variant slot's candidates: they are this pair's now)
		pair_reads(c, pair, h0, h1);                                     // draws from poff[pair] + the two reads' counts; sets rcnt[3 pair + 2], pres[pair]
		c.rh[2 * pair] = h0, c.rh[2 * pair + 1] = h1;
		return;
	}
	// the two headers (cand_off keeps pointing at the variant slot's candidates: they are this pair's now) and the trace hashes
	const uint32_t *src = (const uint32_t *)(c.rh + 2 * slot);
	uint32_t *dst = (uint32_t *)(c.rh + 2 * pair);
	const int nw = (int)(2 * sizeof(psvr_read_hdr_t) / 4);
	for (int i = part; i < nw; i += parts) dst[i] = src[i];
	if (part == 0) {
		c.pres[pair] = c.pres[slot];
		for (int k = 0; k < 4; ++k) c.strand[4 * pair + k] = c.strand[4 * slot + k];      // seed counts / trace hashes travel with the records
		for (int k = 0; k < 3; ++k) c.rcnt[3 * pair + k] = c.rcnt[3 * slot + k];
		c.hcnt[2 * pair] = c.hcnt[2 * pair + 1] = 0;
		c.poff[pair] = noff[pair];                                       // mark_dirty then finds the pair where it belongs
	}
}

// what a run resets before its first round (engine_core.h::run): one launch instead of a dozen fills, copies and memsets of their own
struct RunInit {
	long long *poff, *hoff;
	int32_t *rcnt, *hcnt, *ctot, *hprev, *src;
	uint8_t *sens, *mask;
	long long S, P, V, g, h0, h1;
	const int32_t *vsrc;                 // variant slot -> pair
	const int32_t *spidx; long long nsp; // the special pairs (masked: the host or k_adopt_auto resolves them)
	uint8_t *sp_class; int32_t *sp_adopted; long long *sp_adopted_at;
	unsigned long long *tops; int n_tops; unsigned long long *atops; int n_atops; int32_t *flags; unsigned long long *stats;
	unsigned long long *mem_top; unsigned long long mem0;
};
PSVR_HD void run_init_slot(const RunInit &r, long long s)
{
	if (s < r.S) {
		r.poff[s] = r.g, r.hoff[2 * s] = r.h0, r.hoff[2 * s + 1] = r.h1;
		r.rcnt[3 * s] = 0, r.rcnt[3 * s + 1] = 0, r.rcnt[3 * s + 2] = 0, r.hcnt[2 * s] = 0, r.hcnt[2 * s + 1] = 0, r.ctot[s] = 0, r.hprev[2 * s] = 0, r.hprev[2 * s + 1] = 0;
		r.src[s] = s >= r.P && s < r.P + r.V ? r.vsrc[s - r.P] : (int32_t)s;
		if (s < r.P) r.sens[s] = 0, r.mask[s] = 0;
	}
	if (s < r.nsp) r.sp_class[s] = 0, r.sp_adopted[s] = -1, r.sp_adopted_at[s] = -1;
	if (s < r.n_tops) r.tops[s] = 0;
	if (s < r.n_atops) r.atops[s] = 0;
	if (s < 16) { r.flags[s] = 0; if (r.stats) r.stats[s] = 0; }
	if (s == 0) *r.mem_top = r.mem0;
}

// ---- pairs with N draws whose variant slots make them predictable (engine_core.h) ----
// A pair with 1..3 N bases was also run once per residue assignment (its "variant slots").  If every variant draws the same number from
// the stream, the pair's total does not depend on where in the stream it stands: it needs no place in the host's walk, and which variant's
// records are the pair's follows from the residues at its final offset -- looked up and adopted on the device.
struct SpecialPair { int32_t pair; uint8_t n1, n2; int32_t vslot, nvar; };
PSVR_HD int special_is_const(const Ctx &c, const SpecialPair &sp)
{
	const int32_t *v0 = c.rcnt + 3 * (long long)sp.vslot;
	for (int v = 0; v < sp.nvar; ++v) {
		const long long slot = (long long)sp.vslot + v;
		const int32_t *vc = c.rcnt + 3 * slot;
		if (vc[0] != v0[0] || vc[0] + vc[1] + vc[2] != v0[0] + v0[1] + v0[2]) return 0;
		if (c.hcnt[2 * slot] != 0 || c.hcnt[2 * slot + 1] != 0) return 0;       // (adopt_variant declines those)
	}
	return 1;
}
// the variant slot the residues at stream offset t select (the host walk's `code`, engine_core.h); -1: outside the device's window of the stream
PSVR_HD long long special_slot_at(const Ctx &c, const SpecialPair &sp, long long t)
{
	int code = 0, sh = 0;
	for (int j = 0; j < sp.n1; ++j) {
		const long long k = t + j - c.grand_base;
		if (k < 0 || k >= c.grand_n) return -1;
		code |= (c.grand[k] & 3) << sh, sh += 2;
	}
	const int32_t c1 = c.rcnt[3 * ((long long)sp.vslot + code)];          // mate 0 does not depend on mate 1's residues
	for (int j = 0; j < sp.n2; ++j) {
		const long long k = t + c1 + j - c.grand_base;
		if (k < 0 || k >= c.grand_n) return -1;
		code |= (c.grand[k] & 3) << sh, sh += 2;
	}
	return (long long)sp.vslot + code;
}
// a predictable pair at its (new) offset: adopt the variant its residues select, unless it carries that one already (the host walk's rule for
// the pairs it resolves itself).  All `parts` callers of a pair decide alike; part 0 records the adoption.  Returns 1 if it adopted.
PSVR_HDN inline int adopt_auto(const Ctx &c, const SpecialPair &sp, const long long *noff, int32_t *adopted, long long *adopted_at, int part, int parts)
{
	const long long t = noff[sp.pair];
	const long long slot = special_slot_at(c, sp, t);
	if (slot < 0) { *c.err = 2; return 0; }
	const int32_t *vc = c.rcnt + 3 * slot;
	const int32_t had = *adopted;
	const long long had_at = *adopted_at;
	// (the same slot at another offset is adopted again even if nothing of it depends on the offset: the adoption is also what tells
	// mark_dirty that the pair stands where it belongs -- left alone it would run in full)
	if (!(vc[0] >= sp.n1 && vc[1] >= sp.n2 && (had != (int32_t)slot || had_at != t))) return 0;
	adopt_variant(c, sp.pair, slot, noff, part, parts);
	if (part == 0) *adopted = (int32_t)slot, *adopted_at = t;
	return 1;
}

// materialise the byte sequences of one queued DP problem (get_refseq + the reversal of left extensions,
// rr.cpp:920-928)
PSVR_HD void dp_fetch_base(const Ctx &c, const DpDesc &d, int i, uint8_t *q, uint8_t *t)
{
	if (i < d.qlen) {
		const int qi = d.q_st + (d.type == 0 ? d.qlen - 1 - i : i);
		// the read's bases from its 2-bit words; the byte form exists only for reads with a lower-case 'n' (code 4 does not fit two bits)
		if (c.has_n4[d.read]) q[i] = c.bin[((long long)d.read * 2 + d.strand) * c.lmax + qi];
		else q[i] = (uint8_t)base_at(c.rb + ((long long)d.read * 2 + d.strand) * c.wmax, (uint64_t)qi);
	}
	if (i < d.tlen) t[i] = (uint8_t)base_at(c.idx.ref_seq, (uint64_t)d.ref_st + (d.type == 0 ? d.tlen - 1 - i : i));
}
PSVR_HD void dp_fetch_one(const Ctx &c, const DpDesc &d, uint8_t *q, uint8_t *t)
{
	int n = d.qlen > d.tlen ? d.qlen : d.tlen;
	for (int i = 0; i < n; ++i) dp_fetch_base(c, d, i, q, t);
}

} // namespace psvr
