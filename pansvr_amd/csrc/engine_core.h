// engine_core.h -- batch orchestration of seam B1, independent of where the stages execute.
//
// `BE` (backend) owns memory and runs the stages: the product backend (engine.hip) launches HIP
// kernels on the MI355X; tests/emu provides a host backend that loops over the same
// __host__ __device__ stage functions so this logic can be checked against the oracle without a GPU.
//
// rand() order: the reference consumes one process-wide rand() stream in input order
// (N-base substitution, tied chains, tied pair scores) plus one random_r stream per handler
// (expand_seed sampling).  The streams are precomputed tables in HBM; every pair gets its stream
// offsets from an exclusive scan of per-item draw counts.  Counts are only known after a pair ran, so
// the batch runs speculatively: run all pairs with guessed offsets, scan the counts, re-run exactly
// those pairs that drew from a wrong offset, repeat until no pair is dirty.  The first dirty pair of
// every round is final afterwards, so the loop terminates; in practice it takes 2-3 rounds over
// ~1 % of the pairs.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "aln_device.h"

namespace psvr {

// glibc TYPE_3 additive feedback generator: rand() (seed 1) and initstate_r(seed, 128-byte state)
struct HostRand3 {
	int32_t ring[31];
	int f, b;
	void seed(unsigned s)
	{
		int32_t word = s ? (int32_t)s : 1;
		ring[0] = word;
		for (int i = 1; i < 31; ++i) {
			long hi = word / 127773, lo = word % 127773;
			word = (int32_t)(16807 * lo - 2836 * hi);
			if (word < 0) word += 2147483647;
			ring[i] = word;
		}
		f = 3, b = 0;
		for (int i = 0; i < 310; ++i) next();
	}
	int32_t next()
	{
		uint32_t val = (uint32_t)ring[f] + (uint32_t)ring[b];
		ring[f] = (int32_t)val;
		int32_t r = (int32_t)(val >> 1);
		if (++f >= 31) { f = 0; ++b; }
		else if (++b >= 31) b = 0;
		return r;
	}
};

struct RandStream {           // lazily extended table of one generator + device mirror
	HostRand3 gen;
	std::vector<int32_t> host;   // host[k] = draw number k (absolute)
	void ensure(long long n) { while ((long long)host.size() < n) host.push_back(gen.next()); }
};

struct DpIO {                 // what the DP stage needs beyond Ctx
	long long begin, end;     // descriptor range of this round
	int32_t *qlen, *tlen;
	long long *q_off, *t_off;
	uint8_t *qbuf, *tbuf;
	psvr_extz_t *ez;
	uint32_t *cig;
	long long qbytes, tbytes, cig_words;
};

struct RunStats {
	long long rounds = 0, pairs_run = 0, pair_only = 0, dp_problems = 0, cands = 0;
	unsigned long long counters[16] = {0};
};

template <class BE> struct EngineCore {
	BE &be;
	Ctx c;
	long long P = 0, R = 0;                 // pairs / reads of the uploaded batch
	long long total_bases = 0;
	RandStream grand, hrand[2];
	long long grand_pos = 0, hrand_pos[2] = {0, 0};     // draws consumed by earlier batches
	long long grand_dev_n = 0, hrand_dev_n = 0;
	int32_t *d_grand = nullptr, *d_hrand[2] = {nullptr, nullptr};
	// device buffers owned here
	std::vector<void *> owned;
	long long *d_noff = nullptr, *d_nhoff = nullptr;
	int32_t *d_work = nullptr, *d_work2 = nullptr, *d_workp = nullptr;   // full re-run lists (ping-pong) and the pair-stage-only list
	unsigned long long *d_tops = nullptr;     // [8] arena tops + dirty count
	int32_t *d_flags = nullptr;               // [8] overflow flags + err
	unsigned long long cap_mem = 0, cap_us = 0, cap_seg = 0, cap_dp = 0, cap_cw = 0, cap_cig = 0;
	DpIO dp;
	long long dp_cap_q = 0, dp_cap_t = 0, dp_cap_c = 0, dp_cap_n = 0;
	RunStats stats;
	std::string err;

	explicit EngineCore(BE &b) : be(b) { memset(&c, 0, sizeof c); memset(&dp, 0, sizeof dp); }

	template <class T> T *alloc(unsigned long long n)
	{
		void *p = be.dalloc((n ? n : 1) * sizeof(T));
		if (p) owned.push_back(p);
		return (T *)p;
	}
	void free_all()
	{
		for (void *p : owned) be.dfree(p);
		owned.clear();
		free_arenas();
		for (void *p : {(void *)d_grand, (void *)d_hrand[0], (void *)d_hrand[1], (void *)dp.qlen, (void *)dp.tlen, (void *)dp.q_off,
		                (void *)dp.t_off, (void *)dp.qbuf, (void *)dp.tbuf, (void *)dp.ez, (void *)dp.cig})
			if (p) be.dfree(p);
		d_grand = d_hrand[0] = d_hrand[1] = nullptr;
		memset(&dp, 0, sizeof dp);
		dp_cap_q = dp_cap_t = dp_cap_c = dp_cap_n = 0;
		grand_dev_n = hrand_dev_n = 0;
	}

	void init(const DevIndex &ix, const psvr_aln_params_t &par)
	{
		c.idx = ix;
		c.par = par;
		int k = 0;                                 // ksw_gen_mat_D, rr.cpp:829-843
		for (int l = 0; l < 4; ++l) { for (int m = 0; m < 4; ++m) c.mat[k++] = (int8_t)(l == m ? par.match : -par.mismatch); c.mat[k++] = 0; }
		for (int m = 0; m < 5; ++m) c.mat[k++] = 0;
		// rr.cpp:62-67 at -t 1: the two handlers seed their random_r state with rand() draws #0 and #1
		grand.gen.seed(1);
		grand.ensure(2);
		hrand[0].gen.seed((unsigned)grand.host[0]);
		hrand[1].gen.seed((unsigned)grand.host[1]);
		grand_pos = 2;
	}

	bool upload_rand(long long need_g, long long need_h)
	{
		if (grand_dev_n < need_g) {
			long long n = need_g + need_g / 2 + 4096;
			grand.ensure(grand_pos + n);
			if (d_grand) be.dfree(d_grand);
			d_grand = (int32_t *)be.dalloc(n * 4);
			if (!d_grand) return false;
			be.h2d(d_grand, grand.host.data() + grand_pos, n * 4);
			grand_dev_n = n;
		}
		if (hrand_dev_n < need_h) {
			long long n = need_h + need_h / 2 + 4096;
			for (int k = 0; k < 2; ++k) {
				hrand[k].ensure(hrand_pos[k] + n);
				if (d_hrand[k]) be.dfree(d_hrand[k]);
				d_hrand[k] = (int32_t *)be.dalloc(n * 4);
				if (!d_hrand[k]) return false;
				be.h2d(d_hrand[k], hrand[k].host.data() + hrand_pos[k], n * 4);
			}
			hrand_dev_n = n;
		}
		c.grand = d_grand, c.grand_n = grand_dev_n, c.grand_base = grand_pos;
		for (int k = 0; k < 2; ++k) c.hrand[k] = d_hrand[k], c.hrand_base[k] = hrand_pos[k];
		c.hrand_n = hrand_dev_n;
		return true;
	}

	// ---- batch upload: allocate everything sized by the batch
	int upload(long long n_pairs, const char *bases, const int64_t *base_off, const psvr_ori_t *ori)
	{
		for (void *p : owned) be.dfree(p);
		owned.clear();
		P = n_pairs, R = 2 * n_pairs;
		total_bases = R ? base_off[R] : 0;
		int lmax = 0;
		for (long long r = 0; r < R; ++r) { long long l = base_off[r + 1] - base_off[r]; if (l > lmax) lmax = (int)l; }
		if (lmax > kMaxReadLen) { err = "read longer than MAX_READ_LEN 1600"; return PSVR_ERR_UNSUPPORTED; }
		c.n_pairs = P;
		c.lmax = (lmax + 31) & ~31;
		if (c.lmax < 32) c.lmax = 32;
		c.wmax = c.lmax / 32 + 2;
		char *d_bases = alloc<char>(total_bases + 16);
		long long *d_off = alloc<long long>(R + 1);
		psvr_ori_t *d_ori = alloc<psvr_ori_t>(R);
		c.roff = alloc<long long>(3 * P), c.rcnt = alloc<int32_t>(3 * P), d_noff = alloc<long long>(3 * P);
		c.hoff = alloc<long long>(R), c.hcnt = alloc<int32_t>(R), d_nhoff = alloc<long long>(R);
		c.active = alloc<uint8_t>(R), c.unmapped = alloc<uint8_t>(R), c.is_str = alloc<uint8_t>(R);
		c.read_l = alloc<int32_t>(R);
		c.bin = alloc<uint8_t>((unsigned long long)R * 2 * c.lmax);
		c.rb = alloc<uint64_t>((unsigned long long)R * 2 * c.wmax);
		c.seed_list = alloc<uint8_t>((unsigned long long)R * c.lmax);
		c.strand = alloc<Strand>(2 * R);
		c.ccand = alloc<ChainCand>(12 * R), c.n_ccand = alloc<int32_t>(R);
		c.res = alloc<psvr_read_result_t>(R), c.pres = alloc<psvr_pair_result_t>(P);
		d_work = alloc<int32_t>(P), d_work2 = alloc<int32_t>(P), d_workp = alloc<int32_t>(P);
		d_tops = alloc<unsigned long long>(8), d_flags = alloc<int32_t>(8);
		cap_mem = (unsigned long long)2 * R * kMemSlot + (unsigned long long)R * 16 + 4096;
		cap_us = (unsigned long long)R * 48 + 65536;
		cap_cw = (unsigned long long)R * 3 + 1024;
		cap_seg = cap_cw * 12;
		cap_dp = (unsigned long long)R * 4 + 1024;
		cap_cig = (unsigned long long)R * 48 + 4096;
		c.stats = alloc<unsigned long long>(16);
		for (void *p : owned) if (!p) { err = "device allocation failed"; return PSVR_ERR_NOMEM; }
		free_arenas();
		if (!alloc_arenas()) { err = "device allocation failed (arenas)"; return PSVR_ERR_NOMEM; }
		c.err = d_flags + 6;
		be.h2d(d_bases, bases, total_bases);
		be.h2d(d_off, base_off, (R + 1) * 8);
		be.h2d(d_ori, ori, R * sizeof(psvr_ori_t));
		c.bases = d_bases, c.base_off = d_off, c.ori = d_ori;
		return PSVR_OK;
	}

	void free_arenas()
	{
		for (void *p : {(void *)c.mem.base, (void *)c.us.base, (void *)c.path, (void *)c.seg.base, (void *)c.dp.base, (void *)c.cw.base, (void *)c.cig.base})
			if (p) be.dfree(p);
		c.mem.base = nullptr, c.us.base = nullptr, c.path = nullptr, c.seg.base = nullptr, c.dp.base = nullptr, c.cw.base = nullptr, c.cig.base = nullptr;
	}
	bool alloc_arenas()
	{
		c.mem.base = (VMem *)be.dalloc(cap_mem * sizeof(VMem)), c.us.base = (USeed *)be.dalloc(cap_us * sizeof(USeed)), c.path = (PathN *)be.dalloc(cap_us * sizeof(PathN));
		c.seg.base = (Seg *)be.dalloc(cap_seg * sizeof(Seg)), c.dp.base = (DpDesc *)be.dalloc(cap_dp * sizeof(DpDesc));
		c.cw.base = (CandWork *)be.dalloc(cap_cw * sizeof(CandWork)), c.cig.base = (uint32_t *)be.dalloc(cap_cig * 4);
		c.mem.top = d_tops + 0, c.us.top = d_tops + 1, c.seg.top = d_tops + 2, c.dp.top = d_tops + 3, c.cw.top = d_tops + 4, c.cig.top = d_tops + 5;
		c.mem.cap = cap_mem, c.us.cap = cap_us, c.seg.cap = cap_seg, c.dp.cap = cap_dp, c.cw.cap = cap_cw, c.cig.cap = cap_cig;
		c.mem.overflow = d_flags + 0, c.us.overflow = d_flags + 1, c.seg.overflow = d_flags + 2, c.dp.overflow = d_flags + 3, c.cw.overflow = d_flags + 4, c.cig.overflow = d_flags + 5;
		return c.mem.base && c.us.base && c.path && c.seg.base && c.dp.base && c.cw.base && c.cig.base;
	}
	// a scratch arena overflowed (repeat-rich reads expand to many seeds): grow it 4x and run the batch again
	int grow_and_rerun(const int32_t *flags, int trace, bool want_stats, int depth)
	{
		if (depth > 8) { err = "scratch arena overflow persists after 8 growth steps"; return PSVR_ERR_OVERFLOW; }
		unsigned long long *caps[6] = {&cap_mem, &cap_us, &cap_seg, &cap_dp, &cap_cw, &cap_cig};
		for (int k = 0; k < 6; ++k) if (flags[k]) *caps[k] *= 4;
		if (flags[0]) cap_mem += (unsigned long long)2 * R * kMemSlot;
		fprintf(stderr, "[psvr] scratch arena overflow (mem %d us %d seg %d dp %d cand %d cigar %d): growing 4x and re-running the batch\n", flags[0], flags[1], flags[2], flags[3],
		        flags[4], flags[5]);
		free_arenas();
		if (!alloc_arenas()) { err = "device allocation failed while growing arenas"; return PSVR_ERR_NOMEM; }
		return run(trace, want_stats, depth + 1);
	}

	bool ensure_dp(long long n, long long qb, long long tb, long long cw)
	{
		auto grow = [&](void **p, long long &cap, long long need, size_t el) {
			if (need <= cap && *p) return true;
			if (*p) be.dfree(*p);
			cap = need + need / 4 + 1024;
			*p = be.dalloc(cap * el);
			return *p != nullptr;
		};
		bool ok = true;
		if (n + 2 > dp_cap_n || !dp.qlen) {
			long long cap = 0;
			void **arr[5] = {(void **)&dp.qlen, (void **)&dp.tlen, (void **)&dp.q_off, (void **)&dp.t_off, (void **)&dp.ez};
			const size_t el[5] = {4, 4, 8, 8, sizeof(psvr_extz_t)};
			for (int k = 0; k < 5; ++k) { cap = 0; if (*arr[k]) be.dfree(*arr[k]); *arr[k] = nullptr; ok &= grow(arr[k], cap, n + 2, el[k]); }
			dp_cap_n = cap;
		}
		ok &= grow((void **)&dp.qbuf, dp_cap_q, qb + 64, 1);
		ok &= grow((void **)&dp.tbuf, dp_cap_t, tb + 64, 1);
		ok &= grow((void **)&dp.cig, dp_cap_c, cw + 64, 4);
		return ok;
	}

	// ---- one full run of the uploaded batch (rand state is NOT advanced: call commit() for that)
	int run(int trace, bool want_stats, int depth = 0)
	{
		stats = RunStats();
		c.trace = trace;
		if (P == 0) return PSVR_OK;
		unsigned long long *stats_ptr = c.stats;
		if (!want_stats) c.stats = nullptr;
		be.dzero(d_tops, 8 * 8), be.dzero(d_flags, 8 * 4), be.dzero(stats_ptr, 16 * 8);
		unsigned long long mem0 = (unsigned long long)2 * R * kMemSlot;   // bump region starts behind the per-strand slots
		be.h2d(c.mem.top, &mem0, 8);
		// initial guess: nobody draws.  roff = stream position at batch start, hoff likewise
		if (!upload_rand(total_bases / 64 + 4096, 4096)) { err = "rand table allocation failed"; c.stats = stats_ptr; return PSVR_ERR_NOMEM; }
		be.fill_i64(c.roff, 3 * P, 1, 0, grand_pos);
		be.fill_i64(c.hoff, P, 2, 0, hrand_pos[0]);
		be.fill_i64(c.hoff, P, 2, 1, hrand_pos[1]);
		be.dzero(c.rcnt, 3 * P * 4), be.dzero(c.hcnt, R * 4);
		long long nwork = P, npair_only = 0;
		const int32_t *work = nullptr;                 // nullptr = identity list
		long long dp_done = 0, cw_done = 0;
		int rc = PSVR_OK;
		for (;;) {
			stats.rounds++, stats.pairs_run += nwork;
			be.st_prep(c, work, nwork);
			be.st_str(c, work, nwork);
			be.st_seed(c, work, nwork);
			be.st_chain(c, work, nwork);
			be.st_select(c, work, nwork);
			be.st_walk(c, work, nwork);
			unsigned long long tops[8];
			be.d2h(tops, d_tops, 64);
			long long dp_end = (long long)tops[3], cw_end = (long long)tops[4];
			if (dp_end > (long long)cap_dp || cw_end > (long long)cap_cw) {
				int32_t fl[8] = {0, 0, 0, dp_end > (long long)cap_dp, cw_end > (long long)cap_cw, 0, 0, 0};
				c.stats = stats_ptr;
				return grow_and_rerun(fl, trace, want_stats, depth);
			}
			if (dp_end > dp_done) {
				dp.begin = dp_done, dp.end = dp_end;
				rc = be.st_dp(*this);
				if (rc) break;
			}
			c.dp_ez = dp.ez ? dp.ez - dp_done : nullptr, c.dp_cig = dp.cig;   // DP ids are absolute, the result buffers are per round
			if (cw_end > cw_done) be.st_assemble(c, cw_done, cw_end);
			stats.dp_problems += dp_end - dp_done, stats.cands += cw_end - cw_done;
			dp_done = dp_end, cw_done = cw_end;
			be.st_finalize(c, work, nwork);
			be.st_pair(c, work, nwork);
			if (npair_only) be.st_pair(c, d_workp, npair_only), stats.pair_only += npair_only;
			// new offsets from the draw counts; which pairs drew from a stale offset?
			be.st_scan(c.rcnt, 3 * P, 1, 0, grand_pos, d_noff);
			be.st_scan(c.hcnt, P, 2, 0, hrand_pos[0], d_nhoff);
			be.st_scan(c.hcnt, P, 2, 1, hrand_pos[1], d_nhoff);
			be.dzero(d_tops + 6, 16);
			be.st_dirty(c, d_noff, d_nhoff, d_work2, d_tops + 6, d_workp, d_tops + 7);
			unsigned long long nd[2] = {0, 0};
			int32_t flags[8];
			be.d2h(nd, d_tops + 6, 16);
			const unsigned long long ndirty = nd[0];
			npair_only = (long long)nd[1];
			be.d2h(flags, d_flags, 32);
			if (flags[0] | flags[1] | flags[2] | flags[3] | flags[4] | flags[5]) { c.stats = stats_ptr; return grow_and_rerun(flags, trace, want_stats, depth); }
			if (flags[6] == 2 || flags[6] == 3) {      // a rand table ran out: extend and restart the batch
				be.dzero(d_flags, 32);
				grand_dev_n = hrand_dev_n = 0;
				if (!upload_rand(c.grand_n * 4, c.hrand_n * 4)) { err = "rand table allocation failed"; rc = PSVR_ERR_NOMEM; break; }
				c.stats = stats_ptr;
				return run(trace, want_stats, depth);
			}
			if (flags[6]) { char b[128]; snprintf(b, sizeof b, "device stage error %d (the reference would abort here)", flags[6]); err = b; rc = PSVR_ERR_UNSUPPORTED; break; }
			if (ndirty == 0 && npair_only == 0) break;
			int32_t *t = d_work; d_work = d_work2; d_work2 = t;
			work = d_work, nwork = (long long)ndirty;
		}
		c.stats = stats_ptr;
		if (rc == PSVR_OK && want_stats) be.d2h(stats.counters, stats_ptr, 16 * 8);
		return rc;
	}

	// advance the rand streams past this batch (the reference's generators keep running across batches)
	void commit()
	{
		if (P == 0) return;
		long long last[3];
		int32_t lc;
		be.d2h(&last[0], c.roff + (3 * P - 1), 8);
		be.d2h(&lc, c.rcnt + (3 * P - 1), 4);
		long long new_g = last[0] + lc;
		long long hp[2];
		for (int k = 0; k < 2; ++k) {
			long long ho; int32_t hc;
			be.d2h(&ho, c.hoff + (R - 2 + k), 8);
			be.d2h(&hc, c.hcnt + (R - 2 + k), 4);
			hp[k] = ho + hc;
		}
		grand_pos = new_g, hrand_pos[0] = hp[0], hrand_pos[1] = hp[1];
		grand_dev_n = hrand_dev_n = 0;          // tables are relative to the stream position: refresh on next run
	}
};

} // namespace psvr
