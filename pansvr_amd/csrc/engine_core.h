// engine_core.h -- batch orchestration of seam B1, independent of where the stages execute.
//
// `BE` (backend) owns memory and runs the stages: the product backend (engine.hip) launches HIP
// kernels on the MI355X; tests/emu provides a host backend that loops over the same
// __host__ __device__ stage functions so this logic can be checked against the oracle without a GPU.
//
// rand() order: the reference consumes one process-wide rand() stream in input order
// (N-base substitution, tied chains, tied pair scores) plus one random_r stream per handler
// (expand_seed sampling).  The streams are precomputed tables in HBM.  A pair is the atomic unit: its
// mate 0 runs prep..select first, mate 1 starts where mate 0 stopped, the pairing stage where mate 1
// stopped, so the draws D_p of pair p are a pure function of its stream offset O_p, and
// O_{p+1} = O_p + D_p(O_p).  D_p is almost always independent of O_p, so the batch runs speculatively:
//   1. run every pair at a guessed offset, scan the counts into offsets, re-run the pairs that drew from
//      a stale offset (most of them only need the pairing stage again);
//   2. the pairs whose count does depend on the drawn values form a serial chain that would cost one round
//      per flip.  Almost all of them contain N bases: D_p then depends on the substituted residues, not on
//      the offset, so every pair with 1..3 N draws is ALSO evaluated once per residue combination in
//      round 1 ("variant" shadow slots with forced draws, ~6 % extra work) and the chain
//      O_{s+1} = O_s + D_s(stream content at O_s) is walked exactly on the host through those tables;
//   3. any other pair whose count changes between two evaluations (tied chains in repeats) is evaluated at
//      a window of offsets around its estimate in the next round and joins the same host walk;
//   4. repeat until no pair is dirty (3-4 rounds in practice).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
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

// The process-wide rand() stream (the reference never seeds it: seed 1), generated once for every engine of the process in blocks that never
// move.  Each engine used to keep a table of its own and extend it when its position left the device window: a pipeline's job slots -- or
// the devices of one command -- each generated the same millions of draws again, 17 ms at a time in the middle of a run (the overlapped
// rate through the ABI lost a third to it, the CLI's engine stage ~2 ms per batch).
struct SharedRand {
	static constexpr int kLog = 22;                      // 4 M draws per block
	static constexpr long long kMask = (1ll << kLog) - 1;
	std::mutex mu;
	HostRand3 gen;
	int32_t *blk[1 << 12] = {};                          // 16 G draws
	std::atomic<long long> n{0};
	SharedRand() { gen.seed(1); }
	~SharedRand() { for (int32_t *b : blk) delete[] b; }
	void ensure(long long want)
	{
		if (want <= n.load(std::memory_order_acquire)) return;
		std::lock_guard<std::mutex> lk(mu);
		long long k = n.load(std::memory_order_relaxed);
		want = (want + 0xfffff) & ~0xfffffll;            // a million at a time
		for (; k < want; ++k) {
			int32_t *&b = blk[k >> kLog];
			if (!b) b = new int32_t[(size_t)1 << kLog];
			b[k & kMask] = gen.next();
		}
		n.store(k, std::memory_order_release);
	}
	int32_t at(long long k) const { return blk[k >> kLog][k & kMask]; }          // k below what the caller has ensure()d
	void prefetch(long long k) const { if (k >= 0 && k < n.load(std::memory_order_relaxed)) __builtin_prefetch(blk[k >> kLog] + (k & kMask)); }
	// the draws [from, from + cnt) handed to `put(dst offset, source, entries)` block by block
	template <class F> void pieces(long long from, long long cnt, F &&put) const
	{
		for (long long o = 0; o < cnt;) {
			const long long k = from + o, room = (kMask + 1) - (k & kMask), m = cnt - o < room ? cnt - o : room;
			put(o, blk[k >> kLog] + (k & kMask), m);
			o += m;
		}
	}
};
inline SharedRand &shared_grand() { static SharedRand s; return s; }

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
	long long rounds = 0, pairs_run = 0, pair_only = 0, shadow_runs = 0, sensitive = 0, window_miss = 0, dp_problems = 0, cands = 0, adopted = 0, stale_open = 0, dp_seq_bytes = 0, from_walk = 0;
	long long walk_pairs = 0, walk_us = 0, n_special = 0, special_const = 0, special_nomove = 0;     // the host walk over the N / tie-sensitive pairs: entries, microseconds (all rounds of the batch)
	unsigned long long counters[16] = {0};
};

template <class BE> struct EngineCore {
	BE &be;
	Ctx c;
	long long P = 0, R = 0;                 // pairs / reads of the uploaded batch
	long long total_bases = 0;
	SharedRand &grand = shared_grand();             // rand(): one stream per process, as in the reference
	RandStream hrand[2];                            // random_r: one per handler
	long long grand_pos = 0, hrand_pos[2] = {0, 0};     // draws consumed by earlier batches
	long long grand_dev_n = 0, hrand_dev_n = 0;
	// the device tables hold a WINDOW of the host streams: entries [base, base + n).  A run, a rebase or a new batch whose needs lie inside the
	// window keeps it (a shard that moves behind its predecessors' draws, the next batch of a pipeline: no 28 MB upload, no hipFree)
	long long grand_dev_base = 0, hrand_dev_base[2] = {0, 0}, grand_dev_cap = 0, hrand_dev_cap = 0;
	int32_t *d_grand = nullptr, *d_hrand[2] = {nullptr, nullptr};
	// device buffers owned here
	std::vector<void *> owned;
	long long *d_noff = nullptr, *d_nhoff = nullptr;
	int32_t *d_work = nullptr, *d_workp = nullptr;   // full re-run list (real pairs then shadow slots) and the pairing-only list
	int32_t *d_ctot = nullptr, *d_src = nullptr;     // per slot: total draws of the last evaluation; slot -> input pair
	int32_t *d_hprev = nullptr;                      // per slot x mate: random_r draws of the last evaluation (to see whether a round changed anything)
	uint8_t *d_sens = nullptr;                       // per pair: known count-sensitive
	int32_t *d_slist = nullptr;                      // newly detected sensitive pairs
	char *d_bases = nullptr; long long *d_off = nullptr; psvr_ori_t *d_ori = nullptr;   // the uploaded batch
	long long cap_S = 0, cap_P = 0; int cap_lm = 0;                        // what the per-batch buffers were sized for
	uint8_t *d_force = nullptr, *d_mask = nullptr;   // forced draws per read; per pair: resolved on the host (special or sensitive)
	// the special pairs on the device: the list, which of them draw the same number under every residue assignment (resolved there, not in the
	// host walk), the variant slot each of those carries and the offset it was adopted at
	SpecialPair *d_special = nullptr;
	int32_t *d_vsrc = nullptr, *d_spidx = nullptr;   // variant slot -> pair, the special pairs' numbers: constant for the batch, copied / scattered device to device at every run
	uint8_t *d_sp_class = nullptr;
	int32_t *d_sp_adopted = nullptr;
	long long *d_sp_adopted_at = nullptr;
	std::vector<uint8_t> h_sp_class;
	uint8_t *d_hasn = nullptr;                       // per pair: a read of it draws for N bases (its draws are not just chain-selection ties)
	int32_t *d_resel = nullptr, *d_resel4 = nullptr;   // pairs whose chain selection runs again on its own (reselect_pair); of those, the ones that go on from the walk
	std::vector<int32_t> h_n_idx;                    // pairs with N draws (built by upload())
	static const long long kReselCap = 1 << 16;      // tie-only pairs a round can resolve on the spot; beyond that they run in full
	int32_t *d_cmask = nullptr;                      // totals with the host-resolved pairs masked out
	typedef SpecialPair Special;                     // (aln_device.h: the device looks at them too)
	std::vector<Special> special;                    // pairs with 1..3 N draws, ascending
	long long V = 0;                                 // variant slots [P, P+V)
	long long S = 0;                                 // slots = P real pairs + V variants + window-shadow capacity
	static const int kWin = 32;                      // offsets evaluated per sensitive pair and round
	unsigned long long *d_tops = nullptr;     // [16] dirty counts, totals counters
	// the six arena tops, each in a cache line of its own (kTopStride words apart): a wavefront's atomic on a line costs ~12 ns however many
	// lanes take part, and atomics on one line are served one after the other (tools/atomic_rate_bench.hip) -- three counters that the
	// walk bumps per read shared one line
	static constexpr int kTopStride = kArenaTopStride * kArenaMaxShards;   // words between two arenas' first counters
	unsigned long long *d_atops = nullptr;
	int32_t *d_flags = nullptr;               // [8] six overflow flags, the error word, [7] stale_open (aln_device.h stale_compare)
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
		free_inputs();
		free_arenas();
		for (void *p : {(void *)d_grand, (void *)d_hrand[0], (void *)d_hrand[1], (void *)dp.qlen, (void *)dp.tlen, (void *)dp.q_off,
		                (void *)dp.t_off, (void *)dp.qbuf, (void *)dp.tbuf, (void *)dp.ez, (void *)dp.cig})
			if (p) be.dfree(p);
		d_grand = d_hrand[0] = d_hrand[1] = nullptr;
		memset(&dp, 0, sizeof dp);
		dp_cap_q = dp_cap_t = dp_cap_c = dp_cap_n = 0;
		grand_dev_n = hrand_dev_n = 0, grand_dev_cap = hrand_dev_cap = 0;
	}

	void init(const DevIndex &ix, const psvr_aln_params_t &par)
	{
		c.idx = ix;
		c.par = par;
		int k = 0;                                 // ksw_gen_mat_D, rr.cpp:829-843
		for (int l = 0; l < 4; ++l) { for (int m = 0; m < 4; ++m) c.mat[k++] = (int8_t)(l == m ? par.match : -par.mismatch); c.mat[k++] = 0; }
		for (int m = 0; m < 5; ++m) c.mat[k++] = 0;
		// rr.cpp:62-67 at -t 1: the two handlers seed their random_r state with rand() draws #0 and #1
		grand.ensure(2);
		hrand[0].gen.seed((unsigned)grand.at(0));
		hrand[1].gen.seed((unsigned)grand.at(1));
		grand_pos = 2;
	}

	bool upload_rand(long long need_g, long long need_h)
	{
		// entries beyond what this run needs: room for the position to move before the window is uploaded again (a batch of 1 M pairs needs a
		// window of 4.7 M and draws ~0.4 M times: forty batches; small inputs stay small)
		const long long margin_h = 1 << 20, margin = std::max<long long>(margin_h, 4 * need_g);
		if (!(d_grand && grand_pos >= grand_dev_base && grand_pos + need_g <= grand_dev_base + grand_dev_n)) {
			const long long n = need_g + need_g / 2 + 4096 + margin;
			grand.ensure(grand_pos + n);
			if (n > grand_dev_cap || !d_grand) {
				if (d_grand) be.dfree(d_grand);
				d_grand = (int32_t *)be.dalloc(n * 4);
				if (!d_grand) return false;
				grand_dev_cap = n;
			}
			grand.pieces(grand_pos, n, [&](long long o, const int32_t *src, long long m) { be.h2d(d_grand + o, src, (size_t)m * 4); });
			grand_dev_base = grand_pos, grand_dev_n = n;
		}
		bool hok = d_hrand[0] && d_hrand[1];
		for (int k = 0; k < 2 && hok; ++k) hok = hrand_pos[k] >= hrand_dev_base[k] && hrand_pos[k] + need_h <= hrand_dev_base[k] + hrand_dev_n;
		if (!hok) {
			const long long n = need_h + need_h / 2 + 4096 + margin_h;
			for (int k = 0; k < 2; ++k) {
				hrand[k].ensure(hrand_pos[k] + n);
				if (n > hrand_dev_cap || !d_hrand[k]) {
					if (d_hrand[k]) be.dfree(d_hrand[k]);
					d_hrand[k] = (int32_t *)be.dalloc(n * 4);
					if (!d_hrand[k]) return false;
				}
				be.h2d(d_hrand[k], hrand[k].host.data() + hrand_pos[k], n * 4);
				hrand_dev_base[k] = hrand_pos[k];
			}
			if (n > hrand_dev_cap) hrand_dev_cap = n;
			hrand_dev_n = n;
		}
		c.grand = d_grand, c.grand_n = grand_dev_n, c.grand_base = grand_dev_base;
		for (int k = 0; k < 2; ++k) c.hrand[k] = d_hrand[k], c.hrand_base[k] = hrand_dev_base[k];
		c.hrand_n = hrand_dev_n;
		return true;
	}

	// ---- batch upload: allocate everything sized by the batch
	// the batch's three arrays on the device, in buffers of their own (kept while the next batch fits), and the list the device's pass
	// over the batch leaves: (pair, n0 | n1 << 8) for every pair a read of which will draw for N bases
	long long in_cap_bases = 0, in_cap_R = 0;
	int32_t *d_nlist = nullptr;
	void free_inputs()
	{
		for (void *p : {(void *)d_bases, (void *)d_off, (void *)d_ori, (void *)d_nlist}) if (p) be.dfree(p);
		d_bases = nullptr, d_off = nullptr, d_ori = nullptr, d_nlist = nullptr, in_cap_bases = in_cap_R = 0;
	}
	int upload(long long n_pairs, const char *bases, const int64_t *base_off, const psvr_ori_t *ori)
	{
		P = n_pairs, R = 2 * n_pairs;
		// a block of a larger batch may be handed over as a window of the batch's arrays: offsets then start at base_off[0] > 0
		const long long b0 = R ? base_off[0] : 0;
		total_bases = R ? base_off[R] - b0 : 0;
		// The three arrays go to the device first, and the DEVICE looks at them (scan_batch: the longest read, the pairs whose reads will
		// draw for N bases; early-out reads draw nothing, rr.cpp:414 returns first).  Until round 4 the host made those passes over every
		// base before the transfer started -- 5 of an upload's 12 ms for 1 M pairs on eight threads, which a pipeline's three job slots
		// took from the thread that drives the runs: the overlapped rate through the ABI was bound by it.
		if (!d_bases || total_bases > in_cap_bases || R > in_cap_R) {
			free_inputs();
			in_cap_bases = total_bases + total_bases / 8 + 64, in_cap_R = R + R / 8 + 2;
			d_bases = (char *)be.dalloc((size_t)in_cap_bases + 64);     // slack: the prep kernel loads whole 32-base groups
			d_off = (long long *)be.dalloc((size_t)(in_cap_R + 1) * 8);
			d_ori = (psvr_ori_t *)be.dalloc((size_t)in_cap_R * sizeof(psvr_ori_t));
			d_nlist = (int32_t *)be.dalloc((size_t)(in_cap_R / 2 + 2) * 8);
			if (!d_bases || !d_off || !d_ori || !d_nlist) { free_inputs(); err = "device allocation failed (batch)"; return PSVR_ERR_NOMEM; }
		}
		special.clear(), h_n_idx.clear();
		V = 0;
		int lmax = 0;
		if (R > 0) {
			be.h2d_start(d_bases, bases + b0, total_bases);
			be.h2d_start(d_off, base_off, (R + 1) * 8);
			be.h2d_start(d_ori, ori, R * sizeof(psvr_ori_t));
			std::vector<int32_t> nl;                                     // (pair, counts) pairs as the device appended them
			if (!be.scan_batch(d_bases - b0, d_off, d_ori, P, c.par.match, d_nlist, &lmax, nl)) { err = "device pass over the batch failed"; return PSVR_ERR_DEVICE; }
			// (scan_batch has synchronised: the caller's arrays are free again)
			if (lmax > kMaxReadLen) { err = "read longer than MAX_READ_LEN 1600"; return PSVR_ERR_UNSUPPORTED; }
			std::vector<std::pair<int32_t, int32_t>> srt(nl.size() / 2);
			for (size_t i = 0; i < srt.size(); ++i) srt[i] = {nl[2 * i], nl[2 * i + 1]};
			std::sort(srt.begin(), srt.end());
			for (const auto &e : srt) {
				const int n0 = e.second & 0xff, n1 = (e.second >> 8) & 0xff;   // (counts saturate at 255: anything beyond 3 is "many")
				h_n_idx.push_back(e.first);
				if (n0 + n1 <= 3) { special.push_back(Special{e.first, (uint8_t)n0, (uint8_t)n1, (int32_t)(P + V), 1 << (2 * (n0 + n1))}); V += 1ll << (2 * (n0 + n1)); }
			}
		}
		c.n_pairs = P;
		int lm = (lmax + 31) & ~31;
		if (lm < 32) lm = 32;
		const long long shadow_cap = P / 16 + 8192;
		S = P + V + shadow_cap;
		c.n_slots = S;
		// keep every per-batch buffer (and the arenas) while the new batch fits
		const bool fits = !owned.empty() && S <= cap_S && lm <= cap_lm && P <= cap_P;
		c.bases = d_bases - b0, c.base_off = d_off, c.ori = d_ori;
		if (fits) {
			upload_variants();
			return PSVR_OK;
		}
		for (void *p : owned) be.dfree(p);
		owned.clear();
		cap_S = S, cap_lm = lm, cap_P = P;
		c.lmax = lm;
		c.wmax = c.lmax / 32 + 2;
		const long long RS = 2 * S;                                 // reads incl. shadow slots
		c.poff = alloc<long long>(S), c.rcnt = alloc<int32_t>(3 * S), d_noff = alloc<long long>(S);
		c.hoff = alloc<long long>(RS), c.hcnt = alloc<int32_t>(RS), d_nhoff = alloc<long long>(RS);
		c.active = alloc<uint8_t>(RS), c.unmapped = alloc<uint8_t>(RS), c.is_str = alloc<uint8_t>(RS), c.has_n4 = alloc<uint8_t>(RS);
		c.has_mem = alloc<uint8_t>(RS);
		c.str_list = alloc<int32_t>(RS), c.str_cnt = alloc<unsigned int>(4);
		c.read_l = alloc<int32_t>(RS);
		c.bin = alloc<uint8_t>((unsigned long long)RS * 2 * c.lmax);
		c.rb = alloc<uint64_t>((unsigned long long)RS * 2 * c.wmax);
		c.seed_list = alloc<uint8_t>((unsigned long long)RS * c.lmax);
		c.strand = alloc<Strand>(2 * RS);
		c.ccand = alloc<ChainCand>(12 * RS), c.n_ccand = alloc<int32_t>(RS);
		c.rh = alloc<psvr_read_hdr_t>(RS), c.pres = alloc<psvr_pair_result_t>(S);
		d_work = alloc<int32_t>(S), d_workp = alloc<int32_t>(P);
		d_ctot = alloc<int32_t>(S), d_src = alloc<int32_t>(S), d_sens = alloc<uint8_t>(P), d_slist = alloc<int32_t>(P);
		d_hprev = alloc<int32_t>(2 * S);
		d_force = alloc<uint8_t>(8 * S), d_mask = alloc<uint8_t>(P), d_cmask = alloc<int32_t>(P);
		d_hasn = alloc<uint8_t>(P), d_resel = alloc<int32_t>(P), d_resel4 = alloc<int32_t>(P < kReselCap ? P : kReselCap);
		{
			const long long nsp = S / 4 + 1;                 // (a special pair has >= 4 variant slots: whatever batch fits these buffers later has no more)
			d_vsrc = alloc<int32_t>(S), d_spidx = alloc<int32_t>(nsp);
			d_special = alloc<Special>(nsp), d_sp_class = alloc<uint8_t>(nsp), d_sp_adopted = alloc<int32_t>(nsp), d_sp_adopted_at = alloc<long long>(nsp);
		}
		d_tops = alloc<unsigned long long>(64), d_atops = alloc<unsigned long long>(6 * kTopStride), d_flags = alloc<int32_t>(16);
		const long long R2 = RS;
		cap_mem = (unsigned long long)2 * R2 * kMemSlot + (unsigned long long)R2 * 16 + 4096;
		cap_us = (unsigned long long)R2 * 48 + 65536;
		cap_cw = (unsigned long long)R2 * 3 + 1024;
		cap_seg = cap_cw * 16;
		cap_dp = (unsigned long long)R2 * 4 + 1024;
		cap_cig = (unsigned long long)R2 * 48 + 4096;
		// PSVR_ARENA_SHRINK=<n>: start with 1/n of the scratch arenas, so that a small input walks through the overflow -> grow -> re-run
		// path of every arena (tests)
		if (const char *e = getenv("PSVR_ARENA_SHRINK")) {
			const unsigned long long n = (unsigned long long)atoll(e);
			if (n > 1) {
				for (unsigned long long *cap : {&cap_us, &cap_cw, &cap_seg, &cap_dp, &cap_cig}) *cap = *cap / n + 64;
				const unsigned long long slots = (unsigned long long)2 * R2 * kMemSlot;    // fixed per-strand slots: only the bump region behind them shrinks
				cap_mem = slots + (cap_mem - slots) / n + 64;
			}
		}
		c.stats = alloc<unsigned long long>(16);
		for (void *p : owned) if (!p) { err = "device allocation failed"; return PSVR_ERR_NOMEM; }
		free_arenas();
		if (!alloc_arenas()) { err = "device allocation failed (arenas)"; return PSVR_ERR_NOMEM; }
		c.err = d_flags + 6, c.stale_open = d_flags + 7, c.any_h = d_flags + 8;
		upload_variants();
		return PSVR_OK;
	}

	// variant slots: pair s at every combination of its N-substitution residues.  Only those slots force draws, so the
	// table is zeroed on the device and just their rows are uploaded, once per batch.
	void upload_variants()
	{
		{
			std::vector<uint8_t> force((size_t)8 * V, 0);
			h_vsrc.assign(V, 0), h_sp_idx.assign(special.size(), 0);
			for (size_t i = 0; i < special.size(); ++i) {
				const Special &sp = special[i];
				h_sp_idx[i] = sp.pair;
				for (int v = 0; v < sp.nvar; ++v) {
					long long slot = sp.vslot + v;
					h_vsrc[slot - P] = sp.pair;
					int code = v;
					for (int k = 0; k < 2; ++k) {
						int n = k == 0 ? sp.n1 : sp.n2;
						uint8_t *f = &force[4 * (2 * (slot - P) + k)];
						f[0] = (uint8_t)n;
						for (int j = 0; j < n; ++j) f[1 + j] = (uint8_t)(code & 3), code >>= 2;
					}
				}
			}
			be.dzero(d_force, (size_t)8 * S);
			if (V) be.h2d(d_force + (size_t)8 * P, force.data(), force.size());
			be.dzero(d_hasn, (size_t)P);
			if (!special.empty()) {                                   // (constant for the batch, like d_force)
				be.h2d(d_special, special.data(), special.size() * sizeof(Special));
				be.h2d(d_spidx, h_sp_idx.data(), h_sp_idx.size() * 4);
			}
			if (V) be.h2d(d_vsrc, h_vsrc.data(), V * 4);
			if (!h_n_idx.empty()) be.scatter_u8(d_hasn, h_n_idx.data(), (long long)h_n_idx.size(), 1);
		}
	}

	void free_arenas()
	{
		for (void *p : {(void *)c.mem.base, (void *)c.us.base, (void *)c.path, (void *)c.seg.base, (void *)c.dp.base, (void *)c.cw.base, (void *)c.cig.base, (void *)c.cand})
			if (p) be.dfree(p);
		c.mem.base = nullptr, c.us.base = nullptr, c.path = nullptr, c.seg.base = nullptr, c.dp.base = nullptr, c.cw.base = nullptr, c.cig.base = nullptr, c.cand = nullptr;
	}
	bool alloc_arenas()
	{
		c.mem.base = (VMem *)be.dalloc(cap_mem * sizeof(VMem)), c.us.base = (USeed *)be.dalloc(cap_us * sizeof(USeed)), c.path = (PathN *)be.dalloc(cap_us * sizeof(PathN));
		c.seg.base = (Seg *)be.dalloc(cap_seg * sizeof(Seg)), c.dp.base = (DpDesc *)be.dalloc(cap_dp * sizeof(DpDesc));
		c.cw.base = (CandWork *)be.dalloc(cap_cw * sizeof(CandWork)), c.cig.base = (uint32_t *)be.dalloc(cap_cig * 4);
		c.cand = (psvr_cand_t *)be.dalloc(cap_cw * sizeof(psvr_cand_t));       // candidate records share the CandWork arena's indices
		c.mem.top = d_atops + 0 * kTopStride, c.us.top = d_atops + 1 * kTopStride, c.seg.top = d_atops + 2 * kTopStride, c.dp.top = d_atops + 3 * kTopStride, c.cw.top = d_atops + 4 * kTopStride, c.cig.top = d_atops + 5 * kTopStride;
		c.mem.cap = cap_mem, c.us.cap = cap_us, c.seg.cap = cap_seg, c.dp.cap = cap_dp, c.cw.cap = cap_cw, c.cig.cap = cap_cig;
		c.mem.nshard = c.dp.nshard = c.cw.nshard = c.cig.nshard = 1;          // mem: fixed slots in front; dp, cw: their ids are ranges the host plans on; cig: downloaded as one piece
		c.us.nshard = c.seg.nshard = BE::kArenaShards;
		c.mem.overflow = d_flags + 0, c.us.overflow = d_flags + 1, c.seg.overflow = d_flags + 2, c.dp.overflow = d_flags + 3, c.cw.overflow = d_flags + 4, c.cig.overflow = d_flags + 5;
		return c.mem.base && c.us.base && c.path && c.seg.base && c.dp.base && c.cw.base && c.cig.base && c.cand;
	}
	// a scratch arena overflowed (repeat-rich reads expand to many seeds): grow it 4x and run the batch again
	int grow_and_rerun(const int32_t *flags, int trace, bool want_stats, int depth)
	{
		if (depth > 8) { err = "scratch arena overflow persists after 8 growth steps"; return PSVR_ERR_OVERFLOW; }
		unsigned long long *caps[6] = {&cap_mem, &cap_us, &cap_seg, &cap_dp, &cap_cw, &cap_cig};
		for (int k = 0; k < 6; ++k) if (flags[k]) *caps[k] *= 4;
		if (flags[0]) cap_mem += (unsigned long long)4 * S * kMemSlot;
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

	// the stages of one round over a list of slots: mate 0, then mate 1 (continues mate 0's draws), then the
	// candidate / DP / pairing stages
	// `nwalk`: entries behind the first nwork of `work` that go on from the walk (their chain selection has been repeated already, k_reselect)
	int run_slots(const int32_t *work, long long nwork, long long &dp_done, long long &cw_done, long long nwalk = 0)
	{
		if (nwork + nwalk <= 0) return PSVR_OK;
		for (int mate = 0; mate < 2; ++mate) {
			be.st_prep(c, work, nwork, mate);
			be.st_str(c, work, nwork, mate);
			be.st_seed(c, work, nwork, mate);
			be.st_chain(c, work, nwork, mate);
			be.st_select(c, work, nwork, mate);
		}
		be.st_walk(c, work, nwork + nwalk);
		unsigned long long tops[kTopStride + 1];                             // from the dp counter to the cw counter
		int32_t fl[16];
		be.d2h2(tops, d_atops + 3 * kTopStride, sizeof tops, fl, d_flags, 64);
		if (fl[8]) any_h = true;
		long long dp_end = (long long)tops[0], cw_end = (long long)tops[kTopStride];
		if (fl[7]) stats.stale_open = 1;
		// An arena that filled up in the stages so far ends the round here: what follows (assembly, the reads' tails) would walk records
		// that were never written.  Bits: 1 dp, 2 cw, 4 seg, 8 us, 16 mem.
		{
			const int full = (dp_end > (long long)cap_dp || fl[3] ? 1 : 0) | (cw_end > (long long)cap_cw || fl[4] ? 2 : 0) | (fl[2] ? 4 : 0) | (fl[1] ? 8 : 0) | (fl[0] ? 16 : 0);
			if (full) return -1000 - full;
		}
		if (dp_end > dp_done) {
			dp.begin = dp_done, dp.end = dp_end;
			int rc = be.st_dp(*this);
			if (rc) return rc;
		}
		c.dp_ez = dp.ez ? dp.ez - dp_done : nullptr, c.dp_cig = dp.cig;   // DP ids are absolute, the result buffers are per round
		if (cw_end > cw_done) be.st_assemble(c, cw_done, cw_end);
		stats.dp_problems += dp_end - dp_done, stats.cands += cw_end - cw_done;
		dp_done = dp_end, cw_done = cw_end;
		be.st_finalize_pair(c, work, nwork + nwalk);       // both reads' tails, then the pairing, by the same worker: the records stay close
		return PSVR_OK;
	}

	// ---- one full run of the uploaded batch (rand state is NOT advanced: call commit() for that)
	struct Win { int32_t pair; long long eval_off; int32_t eval_tot; std::vector<long long> off; std::vector<int32_t> tot; };   // window-resolved pair
	// state of the current batch's resolution, kept so that rebase() can continue from it
	std::vector<int32_t> w_listed, w_cur_tot, w_res;    // scratch of the host offset walk, kept across iterations
	std::vector<long long> w_pre, w_cur_off;
	std::vector<int32_t> h_vsrc, h_sp_idx;            // variant slot -> pair; pairs with variant slots (built by upload())
	// per variant slot: c1, c2, c3 -- read back once per batch (1 MB on the bench batch); a plain buffer: a vector's resize() would zero what the
	// readback overwrites
	struct HostTable {
		int32_t *p = nullptr;
		size_t n = 0, cap = 0;
		bool empty() const { return n == 0; }
		void clear() { n = 0; }
		int32_t *data() { return p; }
		const int32_t &operator[](size_t i) const { return p[i]; }
		// the table as a view of memory somebody else owns (the backend's page-locked readback region: no copy out of it)
		bool alias = false;
		void view(int32_t *ext, size_t want) { if (!alias) free(p); p = ext, n = want, cap = 0, alias = true; }
		bool resize(size_t want)
		{
			if (alias) p = nullptr, cap = 0, alias = false;
			if (want > cap) {
				free(p);
				cap = want + want / 4 + 1024;
				p = (int32_t *)malloc(cap * sizeof(int32_t));
				if (!p) { cap = n = 0; return false; }
			}
			n = want;
			return true;
		}
		~HostTable() { if (!alias) free(p); }
		HostTable() = default;
		HostTable(const HostTable &) = delete;
		HostTable &operator=(const HostTable &) = delete;
	} vcnt;
	// The host walk's view of the variant table: per host-resolved special pair the rows of its variant slots, (c1, 2 D + adoptable), copied
	// out of the 1 MB table once per batch in walk order.  The walk visits a few thousand pairs once per step with cold caches: through
	// special[] by search, the table by row and three more per-pair arrays it paid ~170 ns of cache misses per pair (0.5 ms for 2.9 k
	// pairs); over arrays laid out in the order it walks them the prefetchers keep up.
	struct HwRow { int32_t c1, d_ok; };
	std::vector<HwRow> hw_rows;
	std::vector<int32_t> sp_row0;                     // per special pair: its first row in hw_rows (-1: none)
	std::vector<int32_t> l_si;                        // per entry of the walk's list: its index in special[] (-1: a window-resolved pair)
	void build_rows()
	{
		if (vcnt.empty()) return;
		const size_t nl = l_si.size();
		for (size_t k = 0; k < nl; ++k) {
			if (k + 12 < nl && l_si[k + 12] >= 0) __builtin_prefetch(&vcnt[3 * (size_t)(special[(size_t)l_si[k + 12]].vslot - P)]);
			const int32_t si = l_si[k];
			if (si < 0 || sp_row0[(size_t)si] >= 0) continue;
			const Special &sp = special[(size_t)si];
			sp_row0[(size_t)si] = (int32_t)hw_rows.size();
			const int32_t *vc = &vcnt[3 * (size_t)(sp.vslot - P)];
			for (int v = 0; v < sp.nvar; ++v, vc += 3) {
				// adoptable: the two reads of that variant slot drew at least the forced residues (see the walk)
				HwRow r; r.c1 = vc[0], r.d_ok = 2 * (vc[0] + vc[1] + vc[2]) + ((vc[0] >= sp.n1 && vc[1] >= sp.n2) ? 1 : 0);
				hw_rows.push_back(r);
			}
		}
	}
	std::vector<Win> wins;                            // window-resolved (tie-sensitive) pairs, ascending
	std::vector<char> is_special;                     // a special pair whose prediction failed falls back to the window method
	std::vector<int32_t> adopted, adopt_pair, adopt_slot;   // per special pair: the variant slot whose records it carries (-1 none); this walk's new adoptions
	std::vector<long long> adopted_at;                      // ... and the stream offset it was adopted at (a slot whose pairing draws is adopted again when the pair moves)
	long long dp_done = 0, cw_done = 0;
	bool have_run = false;
	bool any_h = false;                               // a read of this batch has drawn from random_r

	int run(int trace, bool want_stats, int depth = 0)
	{
		stats = RunStats();
		c.trace = trace;
		if (P == 0) return PSVR_OK;
		unsigned long long *stats_ptr = c.stats;
		if (!want_stats) c.stats = nullptr;
		if (!upload_rand(total_bases / 64 + 4096, 4096)) { err = "rand table allocation failed"; c.stats = stats_ptr; return PSVR_ERR_NOMEM; }
		// initial guess: nobody draws.  One pass over the slots: poff = grand_pos, hoff = the two random_r positions, rcnt = hcnt = ctot = hprev =
		// sens = mask = 0, src = identity (the variant slots: their pair); the special pairs' classes and adoptions; the counters, flags and
		// statistics; the bump region's start behind the per-strand MEM slots -- a dozen fills, copies and memsets of their own were 0.1 ms per run.
		{
			RunInit ri;
			ri.poff = c.poff, ri.hoff = c.hoff, ri.rcnt = c.rcnt, ri.hcnt = c.hcnt, ri.ctot = d_ctot, ri.hprev = d_hprev, ri.src = d_src, ri.sens = d_sens, ri.mask = d_mask;
			ri.S = S, ri.P = P, ri.V = V, ri.g = grand_pos, ri.h0 = hrand_pos[0], ri.h1 = hrand_pos[1];
			ri.vsrc = d_vsrc, ri.spidx = d_spidx, ri.nsp = (long long)special.size();
			ri.sp_class = d_sp_class, ri.sp_adopted = d_sp_adopted, ri.sp_adopted_at = d_sp_adopted_at;
			ri.tops = d_tops, ri.n_tops = 16, ri.atops = d_atops, ri.n_atops = 6 * kTopStride, ri.flags = d_flags, ri.stats = stats_ptr;
			ri.mem_top = c.mem.top, ri.mem0 = (unsigned long long)4 * S * kMemSlot;   // bump region starts behind the per-strand slots
			be.run_init(ri);
		}
		c.src = d_src, c.force = d_force;
		// (the special pairs are masked after the pass above has cleared every pair's bit)
		if (!h_sp_idx.empty()) be.scatter_u8_dev(d_mask, d_spidx, (long long)h_sp_idx.size(), 1);
		h_sp_class.assign(special.size(), 0);
		dp_done = 0, cw_done = 0, any_h = false;
		vcnt.clear(), wins.clear(), is_special.assign(special.size(), 1);
		hw_rows.clear(), sp_row0.assign(special.size(), -1);
		adopted.assign(special.size(), -1), adopted_at.assign(special.size(), -1), adopt_pair.clear(), adopt_slot.clear();
		have_run = true;
		int rc = iterate(trace, want_stats, depth, stats_ptr, false);
		return rc;
	}

	// The streams of this batch start somewhere else than assumed (another rank's shard precedes it): move every offset
	// and re-run only what drew from a stale position.  Results and draw counts afterwards are those of a run() that had
	// started at the new position.
	int rebase(long long g, long long h0, long long h1, int trace, bool want_stats)
	{
		if (!have_run || P == 0) { grand_pos = g, hrand_pos[0] = h0, hrand_pos[1] = h1; return PSVR_OK; }
		if (g == grand_pos && h0 == hrand_pos[0] && h1 == hrand_pos[1]) return PSVR_OK;
		grand_pos = g, hrand_pos[0] = h0, hrand_pos[1] = h1;
		unsigned long long *stats_ptr = c.stats;
		if (!want_stats) c.stats = nullptr;
		if (!upload_rand(total_bases / 64 + 4096, 4096)) { err = "rand table allocation failed"; c.stats = stats_ptr; return PSVR_ERR_NOMEM; }
		for (Win &w : wins) w.off.clear(), w.tot.clear(), w.eval_off = -1;
		return iterate(trace, want_stats, 0, stats_ptr, true);
	}

	int iterate(int trace, bool want_stats, int depth, unsigned long long *stats_ptr, bool resume)
	{
		int rc = PSVR_OK;
		long long nfull = resume ? 0 : P + V, npair_only = 0, nshadow = 0, nwalk = 0;
		const int32_t *work = nullptr;                    // nullptr = identity: round 1 runs every real pair and every variant slot
		std::vector<int32_t> sh_src; std::vector<long long> sh_off;
		bool skip_eval = resume, pair_done = false;
		bool pre_tot = false;                             // the pairing-only list's totals were taken right behind its launch (and sit in d_tops[8..9] / d_slist)
		for (;;) {
			std::vector<int32_t> &listed = w_listed;
			std::vector<long long> &pre = w_pre, &cur_off = w_cur_off;
			std::vector<int32_t> &cur_tot = w_cur_tot, &res = w_res;
			bool gathered = false;
			// random_r offsets: only when a read has sampled at all (expand_seed beyond POS_N_MAX positions) or the streams' start has moved --
			// otherwise every read stands at the stream's start, where run_init put it, and six scan launches per pass are saved
			// (any_h: run_slots below may set it -- looked at when the scans go out)
			bool h_scans = false;
			// every host-resolved pair: its last evaluation (offset, total) and the masked prefix in front of it
			// (the kernels go out first: the host builds its lists while they run)
			auto enqueue_offset_scans = [&]() {
				h_scans = any_h || resume;
				be.st_mask_totals(d_ctot, d_mask, P, d_cmask);
				be.st_scan(d_cmask, P, 1, 0, 0, d_noff);
				if (h_scans) {
					be.st_scan(c.hcnt, P, 2, 0, hrand_pos[0], d_nhoff);          // (do not depend on the host walk: they run while the host works)
					be.st_scan(c.hcnt, P, 2, 1, hrand_pos[1], d_nhoff);
				}
			};
			auto build_listed = [&]() {
				listed.clear(), l_si.clear();
				size_t wi = 0;                                    // both parts are ascending: merged on the way
				for (size_t i = 0; i < special.size(); ++i) {
					if (is_special[i] != 1) continue;            // (2: resolved on the device; 0: window-resolved, among `wins`)
					for (; wi < wins.size() && wins[wi].pair < special[i].pair; ++wi) listed.push_back(wins[wi].pair), l_si.push_back(-1);
					listed.push_back(special[i].pair), l_si.push_back((int32_t)i);
				}
				for (; wi < wins.size(); ++wi) listed.push_back(wins[wi].pair), l_si.push_back(-1);
				pre.resize(listed.size()), cur_off.resize(listed.size()), cur_tot.resize(listed.size()), res.resize(listed.size());
				build_rows();                                    // (once the table is there; the first round calls it again when it has arrived)
			};
			if (!skip_eval) {
			stats.rounds++;
			if (work == nullptr) stats.pairs_run += P, stats.shadow_runs += V; else stats.pairs_run += nfull, stats.shadow_runs += nshadow;
			stats.pair_only += npair_only;
			// the pairs that only repeat their pairing stage are not among the slots that run in full: a backend with a second queue does
			// them beside the stage chain
			const bool beside = npair_only > 0 && !pair_done && be.side_begin();
			if (beside) { be.st_pair(c, d_workp, npair_only); be.side_end(); }
			stats.from_walk += nwalk;
			rc = run_slots(work, nfull + nshadow, dp_done, cw_done, nwalk);
			if (beside) be.side_wait();
			if (rc <= -1000) {
				const int full = -rc - 1000;
				int32_t fl[8] = {(full >> 4) & 1, (full >> 3) & 1, (full >> 2) & 1, full & 1, (full >> 1) & 1, 0, 0, 0};
				c.stats = stats_ptr;
				return grow_and_rerun(fl, trace, want_stats, depth);
			}
			if (rc) break;
			if (npair_only && !beside && !pair_done) be.st_pair(c, d_workp, npair_only);
			pair_done = false;
			// totals of the evaluated slots; a real pair whose total differs from what the offsets assumed is sensitive
			if (!pre_tot) be.dzero(d_tops + 8, 16);
			be.st_totals(c, work, nfull + nshadow + nwalk, d_ctot, d_hprev, d_sens, d_slist, d_tops + 8, work != nullptr);
			if (npair_only && !pre_tot) be.st_totals(c, d_workp, npair_only, d_ctot, d_hprev, d_sens, d_slist, d_tops + 8, true);
			pre_tot = false;
			unsigned long long nnew_chg[2] = {0, 0};           // newly count-sensitive pairs; did any evaluated slot draw a different number than last time?
			const bool want_vcnt = vcnt.empty() && V && work == nullptr;   // the variant slots' draw counts ride on the same synchronisation
			if (work == nullptr && nshadow == 0) {
				// first round: the offset scans go out behind the totals.  Two short readbacks instead of one long one: first the counters and
				// the special pairs' classes (the pairs every variant of which draws alike leave the host's hands here: unmasked, class 1 --
				// six of seven on the bench batch), with the variant slots' draw counts (1 MB) queued behind them; the host marks the classes
				// and builds its list of the pairs that are left while that copy runs, and the gather for just those pairs brings it in.
				// (One readback of everything had the host walk 20 k pairs, 17 k of them to find they were none of its business, behind a
				// 1.4 MB copy: 0.39 ms of idle GPU per step.)
				const bool classify = want_vcnt && !special.empty();
				if (classify) be.st_special_class(c, d_special, (long long)special.size(), d_mask, d_sp_class);
				enqueue_offset_scans();
				int32_t *late = want_vcnt ? be.d2h_early_late(nnew_chg, d_tops + 8, 16, classify ? h_sp_class.data() : nullptr, d_sp_class, classify ? special.size() : 0,
				                                              c.rcnt + 3 * P, (size_t)3 * V * 4)
				                          : (be.d2h(nnew_chg, d_tops + 8, 16), (int32_t *)nullptr);
				if (classify) for (size_t i = 0; i < special.size(); ++i) if (h_sp_class[i] && is_special[i] == 1) is_special[i] = 2;
				build_listed();
				be.gather_listed(d_noff, c.poff, d_ctot, listed.data(), (long long)listed.size(), pre.data(), cur_off.data(), cur_tot.data(), nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0);
				if (want_vcnt) {
					be.d2h_late_done();                      // (the gather's synchronisation has brought the late copy in as well; no list, no gather: waits here)
					// the table is read at random by the walk below: out of the page-locked region the DMA engine has just written every
					// row is a miss to memory (~170 ns per pair measured: 0.45 ms for 2.9 k pairs); one streaming copy (1 MB, ~35 us) puts it in
					// the CPU's caches (PSVR_VCNT_VIEW=1: read it in place, for A/B runs)
					static const bool in_place = getenv("PSVR_VCNT_VIEW") != nullptr;
					if (late && in_place) vcnt.view(late, (size_t)3 * V);
					else if (!vcnt.resize((size_t)3 * V)) { err = "host allocation failed (variant table)"; rc = PSVR_ERR_NOMEM; break; }
					else if (late) memcpy(vcnt.data(), late, (size_t)3 * V * 4);
					else be.d2h(vcnt.data(), c.rcnt + 3 * P, (size_t)3 * V * 4);
					build_rows();
				}
				gathered = true;
			} else if (want_vcnt) {
				if (!vcnt.resize((size_t)3 * V)) { err = "host allocation failed (variant table)"; rc = PSVR_ERR_NOMEM; break; }
				be.d2h2(nnew_chg, d_tops + 8, 16, vcnt.data(), c.rcnt + 3 * P, 3 * V * 4);
			}
			else be.d2h(nnew_chg, d_tops + 8, 16);
			const unsigned long long nnew = nnew_chg[0];
			// A re-run round in which every slot drew exactly as often as at its previous evaluation leaves every offset where it is: the
			// streams are consistent, the bookkeeping below would find nothing dirty.
			if (work != nullptr && nnew == 0 && nnew_chg[1] == 0 && nshadow == 0 && wins.empty()) break;
			// window tables of the pairs evaluated with offset shadows this round
			if (nshadow > 0) {
				std::vector<int32_t> tot(nshadow);
				be.d2h(tot.data(), d_ctot + (P + V), nshadow * 4);
				size_t sh = 0;
				for (Win &w : wins) {
					w.off.clear(), w.tot.clear();
					while (sh < sh_src.size() && sh_src[sh] == w.pair) w.off.push_back(sh_off[sh]), w.tot.push_back(tot[sh]), ++sh;
				}
			}
			if (nnew) {
				std::vector<int32_t> add(nnew);
				be.d2h(add.data(), d_slist, nnew * 4);
				for (int32_t s : add) {
					for (size_t i = 0; i < special.size(); ++i) if (special[i].pair == s) is_special[i] = 0;
					Win w; w.pair = s, w.eval_off = -1, w.eval_tot = 0;
					wins.push_back(w);
				}
				std::sort(wins.begin(), wins.end(), [](const Win &a, const Win &b) { return a.pair < b.pair; });
				be.scatter_u8(d_mask, add.data(), (long long)add.size(), 1);
				stats.sensitive = (long long)wins.size();
				gathered = false;                               // the mask and the lists have changed
			}
			}   // !skip_eval
			skip_eval = false;
			if (!gathered) {
				enqueue_offset_scans();
				build_listed();
				be.gather_listed(d_noff, c.poff, d_ctot, listed.data(), (long long)listed.size(), pre.data(), cur_off.data(), cur_tot.data(), nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, nullptr, 0);
			}
			// walk O_{s+1} = O_s + D_s through the tables
			{
				if (want_stats && !vcnt.empty() && stats.n_special == 0) {
					// (diagnostic) how many of the N pairs draw the same number whatever residues they get / draw nothing but those residues
					for (const Special &sp : special) {
						bool cst = true, nomove = true;
						const int32_t *v0 = &vcnt[3 * (sp.vslot - P)];
						for (int v = 0; v < sp.nvar; ++v) {
							const int32_t *vc = &vcnt[3 * (sp.vslot - P + v)];
							if (vc[0] != v0[0] || vc[0] + vc[1] + vc[2] != v0[0] + v0[1] + v0[2]) cst = false;
							if (vc[2] != 0 || vc[0] != sp.n1 || vc[1] != sp.n2) nomove = false;
						}
						stats.n_special++, stats.special_const += cst, stats.special_nomove += nomove;
					}
				}
				const auto walk_t0 = std::chrono::steady_clock::now();
				long long acc = 0;
				size_t wi = 0;
				for (size_t i = 0; i < listed.size(); ++i) {
					const int32_t s = listed[i];
					const long long t = grand_pos + pre[i] + acc;
					// (the draws a pair further down will look at: its offset moves by a few entries at most until the walk is there -- each
					// pair's look-up is otherwise a miss to memory, the list being spread over the whole batch's draws)
					if (i + 8 < listed.size()) grand.prefetch(grand_pos + pre[i + 8] + acc);
					int32_t D = cur_tot[i];
					const int32_t sidx = l_si[i];
					if (sidx >= 0 && is_special[(size_t)sidx] == 2) {
						// classified while this list was on its way: its total stands whatever its offset (and is part of `pre` for the pairs behind it)
						res[i] = D;
						continue;
					}
					if (sidx >= 0 && is_special[(size_t)sidx] == 1 && sp_row0[(size_t)sidx] >= 0) {
						const size_t si = (size_t)sidx;
						const Special &sp = special[si];
						const HwRow *rows = &hw_rows[(size_t)sp_row0[si]];
						grand.ensure(t + 64);
						int code = 0, sh2 = 0;
						for (int j = 0; j < sp.n1; ++j) code |= (grand.at(t + j) & 3) << sh2, sh2 += 2;
						const int32_t c1 = rows[code].c1;                           // mate 0 does not depend on mate 1's residues
						for (int j = 0; j < sp.n2; ++j) code |= (grand.at(t + c1 + j) & 3) << sh2, sh2 += 2;
						const HwRow vr = rows[code];
						D = vr.d_ok >> 1;
						// The two reads of that variant slot drew nothing but the forced residues: their records ARE this pair's at any offset
						// (adopted below instead of running the pair again where its draws have moved to).  If the slot's pairing stage drew
						// too, the adoption runs the pairing again at the pair's offset (adopt_variant): again whenever the offset moves.
						// (a read of the slot that drew for tied chains as well -- vc > n -- is adopted if its selection, repeated at this offset,
						// leaves the candidates and the counts as they are; the device declines otherwise and the pair runs in full)
						// (the same slot at another offset is adopted again even if nothing of it depends on the offset: the adoption is also what
						// tells mark_dirty that the pair stands where it belongs -- left alone it ran in full, 3.7 k pairs per rebase of the bench batch)
						if ((vr.d_ok & 1) && (adopted[si] != sp.vslot + code || adopted_at[si] != t)) {
							adopted[si] = sp.vslot + code, adopted_at[si] = t;
							adopt_pair.push_back(s), adopt_slot.push_back(sp.vslot + code);
						}
					} else if (sidx < 0) {
						while (wi < wins.size() && wins[wi].pair < s) ++wi;
						if (wi < wins.size() && wins[wi].pair == s) {
							Win &w = wins[wi];
							w.eval_off = cur_off[i], w.eval_tot = cur_tot[i];
							long long best = std::llabs(t - w.eval_off);
							for (size_t k = 0; k < w.off.size(); ++k) {
								long long d = std::llabs(w.off[k] - t);
								if (d < best) best = d, D = w.tot[k];
							}
							if (best != 0) stats.window_miss++;
						}
					}
					res[i] = D, acc += D;
					cur_off[i] = t;                                  // where the pair must be evaluated next
				}
				stats.walk_pairs += (long long)listed.size();
				stats.walk_us += (long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - walk_t0).count();
			}
			if (!listed.empty()) be.scatter_listed_i32(d_ctot, res.data(), (long long)listed.size());   // same indices as gather_listed
			// new offsets from the totals; which pairs drew from a stale offset?
			be.st_scan(d_ctot, P, 1, 0, grand_pos, d_noff);
			// (the two list counters in cache lines of their own: d_tops[32], d_tops[48])
			be.dzero(d_tops + 8, 56 * 8);                   // [8..9] the totals counters of k_totals_dev below, [32..] the lists' counters, [60] the adoptions: one fill (to the buffer's end: a size the runtime does not split)
			// adoptions: the special pairs the device resolves, and in the same launch the ones this walk decided
			if (!special.empty()) {
				be.st_adopt_auto(c, d_special, (long long)special.size(), d_sp_class, d_mask, d_noff, d_sp_adopted, d_sp_adopted_at, want_stats ? d_tops + 60 : nullptr,
				                 adopt_pair.data(), adopt_slot.data(), (long long)adopt_pair.size());
				stats.adopted += (long long)adopt_pair.size();
				adopt_pair.clear(), adopt_slot.clear();
			}
			be.st_dirty(c, d_noff, h_scans ? d_nhoff : c.hoff, d_work, d_tops + 32, d_workp, d_tops + 48, d_hasn, d_resel, d_tops + 56, P < kReselCap ? P : kReselCap, d_resel4, d_tops + 57);
			// the pairing-only repeats go out at once (the list's length is on the device; the host reads it below for the totals pass)
			be.st_pair_dev(c, d_workp, d_tops + 48);
			pair_done = true;
			// ... and their totals, which the next round would take first thing: when nothing else is left to run (the usual case) that round
			// is over with the readback below instead of costing a synchronisation of its own
			be.st_totals_dev(c, d_workp, d_tops + 48, P, d_ctot, d_hprev, d_sens, d_slist, d_tops + 8);
			pre_tot = true;
			unsigned long long tops[53];                    // d_tops[8..60]: [0] newly count-sensitive, [1] any count changed, from [24] on the lists' counters, [52] adoptions made on the device
			const unsigned long long *nd17 = tops + 24;     // [0] full re-runs, [16] pairing only, [24] tie-only pairs seen, [25] of those: on from the walk
			int32_t flags[16];
			be.d2h2(tops, d_tops + 8, sizeof tops, flags, d_flags, 64);
			if (flags[8]) any_h = true;
			stats.adopted += (long long)tops[52];
			const unsigned long long nd[2] = {nd17[0], nd17[16]};
			if (flags[7]) stats.stale_open = 1;
			if (flags[0] | flags[1] | flags[2] | flags[3] | flags[4] | flags[5]) { c.stats = stats_ptr; return grow_and_rerun(flags, trace, want_stats, depth); }
			if (flags[6] == 2 || flags[6] == 3) {      // a rand table ran out: extend and restart the batch
				be.dzero(d_flags, 64);
				grand_dev_n = hrand_dev_n = 0;
				if (!upload_rand(c.grand_n * 4, c.hrand_n * 4)) { err = "rand table allocation failed"; rc = PSVR_ERR_NOMEM; break; }
				c.stats = stats_ptr;
				return run(trace, want_stats, depth);
			}
			if (flags[6]) { char b[128]; snprintf(b, sizeof b, "device stage error %d (the reference would abort here)", flags[6]); err = b; rc = PSVR_ERR_UNSUPPORTED; break; }
			nfull = (long long)nd[0], npair_only = (long long)nd[1], nshadow = 0, nwalk = (long long)nd17[25];
			if (nfull == 0 && npair_only == 0 && nwalk == 0) break;
			if (nfull == 0 && nwalk == 0 && tops[0] == 0 && tops[1] == 0 && wins.empty()) {
				// the round that only repeats pairing stages: they have run, drew as often as before, nothing is sensitive -- the streams are consistent
				stats.rounds++, stats.pair_only += npair_only;
				break;
			}
			if (stats.rounds > 200) { err = "rand()-order resolution did not converge in 200 rounds"; rc = PSVR_ERR_UNSUPPORTED; break; }
			work = d_work;
			// offset windows for the tie-sensitive pairs that move
			sh_src.clear(), sh_off.clear();
			const long long cap = S - P - V;
			{
				size_t li = 0;
				for (Win &w : wins) {
					while (li < listed.size() && listed[li] < w.pair) ++li;
					const long long t = li < listed.size() && listed[li] == w.pair ? cur_off[li] : w.eval_off;
					if (t == w.eval_off && !w.off.empty()) continue;            // evaluated there already
					if ((long long)sh_src.size() + kWin > cap) break;
					for (int k = -kWin / 2; k < kWin / 2; ++k) {
						if (k == 0 || t + k < grand_pos) continue;
						sh_src.push_back(w.pair), sh_off.push_back(t + k);
					}
				}
			}
			nshadow = (long long)sh_src.size();
			if (nshadow) {
				be.h2d(d_src + P + V, sh_src.data(), nshadow * 4);
				be.h2d(c.poff + P + V, sh_off.data(), nshadow * 8);
				be.copy_hoff_to_shadows(c, P + V, nshadow);
				be.append_iota(d_work, nfull, P + V, nshadow);      // work list: dirty real pairs, then the shadow slots
			}
			if (nwalk) be.append_list(d_work, nfull + nshadow, d_resel4, nwalk);   // ... then the pairs that go on from the walk
		}
		c.stats = stats_ptr;
		if (rc == PSVR_OK && want_stats) be.d2h(stats.counters, stats_ptr, 16 * 8);
		return rc;
	}

	// where the three streams stand after this batch (valid after run()/rebase())
	void stream_end(long long out[3])
	{
		out[0] = grand_pos, out[1] = hrand_pos[0], out[2] = hrand_pos[1];
		if (P == 0) return;
		// the last pair's offset + draw count of each stream (one readback: poff[P-1], then hoff / hcnt of the pair are neighbours)
		long long lo, ho[2];
		int32_t lc, hc[2];
		be.d2h4(&lo, c.poff + (P - 1), 8, &lc, d_ctot + (P - 1), 4, ho, c.hoff + 2 * (P - 1), 16, hc, c.hcnt + 2 * (P - 1), 8);
		out[0] = lo + lc, out[1] = ho[0] + hc[0], out[2] = ho[1] + hc[1];
	}

	// advance the rand streams past this batch (the reference's generators keep running across batches)
	void commit()
	{
		if (P == 0) return;
		long long e[3];
		stream_end(e);
		grand_pos = e[0], hrand_pos[0] = e[1], hrand_pos[1] = e[2];
		have_run = false;
	}
};

} // namespace psvr
