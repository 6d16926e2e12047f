// tests/emu/emu_main.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host backend for pansvr_amd/csrc/engine_core.h: runs the very same stage functions
// (aln_device.h) and batch orchestration the GPU engine uses, but as plain loops on the CPU, with the
// oracle's DP (oracle/ksw_oracle.c) standing in for the HIP DP kernel.  It exists so the stage logic
// and the speculative rand()-offset loop can be checked against the golden records in `-m "not gpu"`
// tests.  It is never linked into libpsvr_engine.so.
//
// With --sam / --ori-sam it also runs the PRODUCT's host-side formatter (pansvr_amd/csrc/sam_emit.h, fastq_batch.h) over these
// results, so the SAM text can be compared with the reference's own files without a GPU.
//
// Usage: emu_aln <fixture_index_dir> <reads.fq> <header.sam> [--trace] [--batch N] [--sam FILE --ori-sam FILE] [--threads N]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../pansvr_amd/csrc/engine_core.h"
#include "../../pansvr_amd/csrc/host_io.h"
#include "../../pansvr_amd/csrc/fastq_batch.h"
#include "../../pansvr_amd/csrc/sam_emit.h"
#include "../../pansvr_amd/csrc/signal_step.h"
#include <thread>
#include "../../oracle/ksw_oracle.h"

using namespace psvr;

struct CpuBE {
	static constexpr unsigned int kArenaShards = 1;
	void *dalloc(size_t n) { return calloc(n ? n : 1, 1); }
	void dfree(void *p) { free(p); }
	void dzero(void *p, size_t n) { memset(p, 0, n); }
	void dfill(void *p, int byte, size_t n) { memset(p, byte, n); }
	void d2d(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }
	void scatter_u8_dev(uint8_t *a, const int32_t *idx, long long n, uint8_t v) { scatter_u8(a, idx, n, v); }
	void h2d(void *d, const void *h, size_t n) { memcpy(d, h, n); }
	void h2d_start(void *d, const void *h, size_t n) { memcpy(d, h, n); }
	void h2d_wait() {}
	// the pass over an uploaded batch (k_scan_batch on the device): longest read, pairs whose reads will draw for N bases
	bool scan_batch(const char *bases, const long long *off, const psvr_ori_t *ori, long long P, int match, int32_t *, int *lmax, std::vector<int32_t> &out)
	{
		out.clear();
		*lmax = 0;
		for (long long p = 0; p < P; ++p) {
			int nn[2] = {0, 0};
			for (int k = 0; k < 2; ++k) {
				const long long r = 2 * p + k, L = off[r + 1] - off[r];
				if (L > *lmax) *lmax = (int)L;
				const bool unm = ori[r].unmapped || (uint32_t)ori[r].chr_id > 24u;
				if ((!unm && ori[r].align_score == (uint32_t)(L * match)) || L < kLenKmer || L > kMaxReadLen) continue;
				int n = 0;
				for (long long i = 0; i < L; ++i) n += bases[off[r] + i] == 'N';
				nn[k] = n < 255 ? n : 255;
			}
			if (nn[0] + nn[1] >= 1) out.push_back((int32_t)p), out.push_back(nn[0] | (nn[1] << 8));
		}
		return true;
	}
	void d2h(void *h, const void *d, size_t n) { memcpy(h, d, n); }
	void d2h2(void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2) { memcpy(h1, d1, n1), memcpy(h2, d2, n2); }
	void d2h4(void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2, void *h3, const void *d3, size_t n3, void *h4, const void *d4, size_t n4) { d2h2(h1, d1, n1, h2, d2, n2), d2h2(h3, d3, n3, h4, d4, n4); }
	// (GPU backend: two small readbacks now, one long one queued behind them; here everything is there at once)
	std::vector<int32_t> late_buf;
	int32_t *d2h_early_late(void *h1, const void *d1, size_t n1, void *h2, const void *d2, size_t n2, const void *dl, size_t nl)
	{
		memcpy(h1, d1, n1);
		if (h2 && n2) memcpy(h2, d2, n2);
		late_buf.resize(nl / 4 + 1);
		memcpy(late_buf.data(), dl, nl);
		return late_buf.data();
	}
	void d2h_late_done() {}
	void fill_i64(long long *p, long long n, int stride, int off, long long v) { for (long long i = 0; i < n; ++i) p[off + i * stride] = v; }
	static long long pr(const int32_t *w, long long i) { return w ? w[i] : i; }
	void fill_iota(int32_t *p, long long n) { for (long long i = 0; i < n; ++i) p[i] = (int32_t)i; }
	void run_init(const RunInit &r)
	{
		long long n = r.S;
		for (long long k : {r.nsp, (long long)r.n_tops, (long long)r.n_atops, 16ll}) if (k > n) n = k;
		for (long long s = 0; s < n; ++s) run_init_slot(r, s);
	}
	void append_iota(int32_t *w, long long at, long long start, long long n) { for (long long i = 0; i < n; ++i) w[at + i] = (int32_t)(start + i); }
	std::vector<int32_t> listed_idx;
	void gather_listed(const long long *a, const long long *b, const int32_t *cc, const int32_t *idx, long long n, long long *oa, long long *ob, int32_t *oc,
	                   void *x1h, const void *x1d, size_t x1n, void *x2h, const void *x2d, size_t x2n, void *x3h, const void *x3d, size_t x3n)
	{
		if (x1h && x1n) memcpy(x1h, x1d, x1n);
		if (x2h && x2n) memcpy(x2h, x2d, x2n);
		if (x3h && x3n) memcpy(x3h, x3d, x3n);
		if (!n) return;                          // (the indices of the gather before stay: scatter_listed_i32 is not called then)
		listed_idx.assign(idx, idx + n);
		for (long long i = 0; i < n; ++i) oa[i] = a[idx[i]], ob[i] = b[idx[i]], oc[i] = cc[idx[i]];
	}
	void st_special_class(const Ctx &c, const SpecialPair *sp, long long n, uint8_t *mask, uint8_t *cls)
	{
		for (long long i = 0; i < n; ++i) { cls[i] = (uint8_t)special_is_const(c, sp[i]); if (cls[i]) mask[sp[i].pair] = 0; }
	}
	void st_adopt_auto(const Ctx &c, const SpecialPair *sp, long long n, const uint8_t *cls, const uint8_t *mask, const long long *noff, int32_t *adopted, long long *adopted_at, unsigned long long *count,
	                   const int32_t *host_pairs, const int32_t *host_slots, long long n_host)
	{
		st_adopt(c, host_pairs, host_slots, n_host, noff);
		for (long long i = 0; i < n; ++i) if (cls[i] && !mask[sp[i].pair]) { const int did = adopt_auto(c, sp[i], noff, adopted + i, adopted_at + i, 0, 1); if (count) *count += (unsigned long long)did; }
	}
	void scatter_listed_i32(int32_t *a, const int32_t *val, long long n) { for (long long i = 0; i < n; ++i) a[listed_idx[i]] = val[i]; }
	void scatter_u8(uint8_t *a, const int32_t *idx, long long n, uint8_t v) { for (long long i = 0; i < n; ++i) a[idx[i]] = v; }
	void st_mask_totals(const int32_t *ctot, const uint8_t *mask, long long n, int32_t *out) { for (long long i = 0; i < n; ++i) out[i] = mask[i] ? 0 : ctot[i]; }
	void copy_hoff_to_shadows(const Ctx &c, long long P, long long n)
	{
		for (long long j = 0; j < n; ++j) for (int k = 0; k < 2; ++k) c.hoff[2 * (P + j) + k] = c.hoff[2 * (long long)c.src[P + j] + k];
	}
	void st_prep(const Ctx &c, const int32_t *w, long long n, int mate) { for (long long i = 0; i < n; ++i) prep_read(c, pr(w, i) * 2 + mate); }
	void st_str(const Ctx &c, const int32_t *w, long long n, int mate) { for (long long i = 0; i < n; ++i) str_detect(c, pr(w, i) * 2 + mate); }
	void st_seed(const Ctx &c, const int32_t *w, long long n, int mate) { for (long long i = 0; i < 2 * n; ++i) seed_strand(c, (pr(w, i >> 1) * 2 + mate) * 2 + (i & 1)); }
	// chaining + chain selection of a read in one go, like the GPU backend: the register-resident small case first, the generic pair for what
	// it declines (PSVR_EMU_NO_SMALL=1: the generic pair for every read)
	long long n_small = 0, n_generic = 0;
	void st_chain(const Ctx &c, const int32_t *w, long long n, int mate)
	{
		static const bool no_small = getenv("PSVR_EMU_NO_SMALL") != nullptr;
		for (long long i = 0; i < n; ++i) {
			const long long r = pr(w, i) * 2 + mate;
			if (!no_small && chain_select_small(c, r)) { ++n_small; continue; }
			++n_generic;
			chain_read(c, r), select_read(c, r);
		}
	}
	void st_select(const Ctx &, const int32_t *, long long, int) {}
	void st_walk(const Ctx &c, const int32_t *w, long long n)
	{
		for (long long i = 0; i < 2 * n; ++i) walk_read(c, pr(w, i >> 1) * 2 + (i & 1));
	}
	void st_totals(const Ctx &c, const int32_t *w, long long n, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt, bool detect)
	{
		for (long long i = 0; i < n; ++i) {
			long long s = pr(w, i);
			int32_t t = c.rcnt[3 * s] + c.rcnt[3 * s + 1] + c.rcnt[3 * s + 2];
			const int32_t h0 = c.hcnt[2 * s], h1 = c.hcnt[2 * s + 1];
			if (t != ctot[s] || h0 != hprev[2 * s] || h1 != hprev[2 * s + 1]) cnt[1] = 1;
			if (detect && s < c.n_pairs && t != ctot[s] && !sens[s]) sens[s] = 1, slist[(*cnt)++] = (int32_t)s;
			ctot[s] = t, hprev[2 * s] = h0, hprev[2 * s + 1] = h1;
		}
	}
	void st_totals_dev(const Ctx &c, const int32_t *list, const unsigned long long *n_dev, long long, int32_t *ctot, int32_t *hprev, uint8_t *sens, int32_t *slist, unsigned long long *cnt)
	{
		st_totals(c, list, (long long)*n_dev, ctot, hprev, sens, slist, cnt, true);
	}
	void st_assemble(const Ctx &c, long long b, long long e) { for (long long i = b; i < e; ++i) assemble_candidate(c, i); }
	void st_finalize(const Ctx &c, const int32_t *w, long long n) { for (long long i = 0; i < 2 * n; ++i) finalize_read(c, pr(w, i >> 1) * 2 + (i & 1)); }
	void st_pair(const Ctx &c, const int32_t *w, long long n) { for (long long i = 0; i < n; ++i) pair_reads(c, pr(w, i)); }
	// as k_finalize_pair does it: both headers built in place, the pairing's items handed on from finalize_read
	void st_finalize_pair(const Ctx &c, const int32_t *w, long long n)
	{
		for (long long i = 0; i < n; ++i) {
			const long long p = pr(w, i);
			PeItem it0[3], it1[3];
			finalize_read(c, 2 * p, c.rh[2 * p], it0), finalize_read(c, 2 * p + 1, c.rh[2 * p + 1], it1);
			pair_reads(c, p, c.rh[2 * p], c.rh[2 * p + 1], it0, it1);
		}
	}
	void st_scan(const int32_t *cnt, long long n, int stride, int off, long long base, long long *out)
	{
		long long acc = base;
		for (long long i = 0; i < n; ++i) { out[off + i * stride] = acc; acc += cnt[off + i * stride]; }
	}
	void st_dirty(const Ctx &c, const long long *noff, const long long *nhoff, int32_t *out, unsigned long long *cnt, int32_t *outp, unsigned long long *cntp,
	              const uint8_t *has_n, int32_t *out3, unsigned long long *cnt3, long long cap3, int32_t *out4, unsigned long long *cnt4)
	{
		for (long long p = 0; p < c.n_pairs; ++p) {
			int d = mark_dirty(c, p, noff, nhoff, has_n);
			if (d == 3 && (long long)*cnt3 >= cap3) d = 2;
			if (d == 3) out3[(*cnt3)++] = (int32_t)p;
			else if (d == 2) out[(*cnt)++] = (int32_t)p;
			else if (d == 1) outp[(*cntp)++] = (int32_t)p;
		}
		for (unsigned long long i = 0; i < *cnt3; ++i) {
			const long long p = out3[i];
			if (reselect_pair(c, p) != 2) outp[(*cntp)++] = (int32_t)p;
			else out4[(*cnt4)++] = (int32_t)p;
		}
	}
	void append_list(int32_t *w, long long at, const int32_t *src, long long n) { for (long long i = 0; i < n; ++i) w[at + i] = src[i]; }
	void st_pair_dev(const Ctx &c, const int32_t *list, const unsigned long long *cnt) { for (unsigned long long i = 0; i < *cnt; ++i) pair_reads(c, list[i]); }
	void st_adopt(const Ctx &c, const int32_t *pairs, const int32_t *slots, long long n, const long long *noff) { for (long long i = 0; i < n; ++i) adopt_variant(c, pairs[i], slots[i], noff, 0, 1); }
	bool side_begin() { return false; }       // one queue
	void side_end() {}
	void side_wait() {}
	template <class Core> int st_dp(Core &core)
	{
		const Ctx &c = core.c;
		DpIO &d = core.dp;
		long long n = d.end - d.begin, qb = 0, tb = 0;
		for (long long i = 0; i < n; ++i) { const DpDesc &x = c.dp.base[d.begin + i]; qb += x.qlen, tb += x.tlen; }
		if (!core.ensure_dp(n, qb, tb, qb + tb + 2 * n)) return PSVR_ERR_NOMEM;
		long long qo = 0, to = 0;
		static FILE *shapes = getenv("EMU_DP_SHAPES") ? fopen(getenv("EMU_DP_SHAPES"), "w") : nullptr;   // one "qlen tlen" line per DP problem, for tools/team_fill.py
		for (long long i = 0; i < n; ++i) {
			const DpDesc &x = c.dp.base[d.begin + i];
			if (shapes) fprintf(shapes, "%d %d\n", x.qlen, x.tlen);
			d.qlen[i] = x.qlen, d.tlen[i] = x.tlen, d.q_off[i] = qo, d.t_off[i] = to;
			dp_fetch_one(c, x, d.qbuf + qo, d.tbuf + to);
			orc_extz_t ez;
			psvr_extz_t &o = d.ez[i];
			o.cigar_off = qo + to + 2 * i;
			orc_extd2(x.qlen, d.qbuf + qo, x.tlen, d.tbuf + to, 5, c.mat, (int8_t)c.par.gap_open, (int8_t)c.par.gap_ex, (int8_t)c.par.gap_open2, (int8_t)c.par.gap_ex2,
			          200, c.par.zdrop, -1, 0, &ez, d.cig + o.cigar_off, x.qlen + x.tlen + 2);
			o.max = ez.max, o.zdropped = ez.zdropped, o.max_q = ez.max_q, o.max_t = ez.max_t, o.mqe = ez.mqe, o.mqe_t = ez.mqe_t;
			o.mte = ez.mte, o.mte_q = ez.mte_q, o.score = ez.score, o.n_cigar = ez.n_cigar, o.reach_end = ez.reach_end;
			qo += x.qlen, to += x.tlen;
		}
		return PSVR_OK;
	}
};

struct HostSvNames : SvNames {
	const HostIndex *h;
	const char *print_string(int sv) const override { return sv >= 0 && sv < (int)h->svh.size() ? h->svh[(size_t)sv].vcf_print_string.c_str() : nullptr; }
	const char *vcf_id(int sv) const override { return sv >= 0 && sv < (int)h->svh.size() ? h->svh[(size_t)sv].vcf_id.c_str() : nullptr; }
};

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: emu_aln <index_dir> <reads.fq> <header.sam> [--trace] [--batch N] [--sam FILE --ori-sam FILE]\n"); return 1; }
	bool trace = false, quiet = false, not_ori = false, sig_n = false, sig_d = false, sig_u = false;
	long long batch = 1 << 20;
	int threads = 1;
	const char *sam_fn = nullptr, *ori_fn = nullptr, *bam_fn = nullptr;
	bool bam_text = false;
	int format_reps = 0;
	long long pos[3] = {-1, -1, -1}, from[3] = {-1, -1, -1};
	for (int i = 4; i < argc; ++i) {
		if (!strcmp(argv[i], "--trace")) trace = true;
		else if (!strcmp(argv[i], "--no-records")) quiet = true;
		else if (!strcmp(argv[i], "-Q")) not_ori = true;            // --not-ori (read_realignment.cpp:485)
		else if (!strcmp(argv[i], "-N")) sig_n = true;
		else if (!strcmp(argv[i], "-D")) sig_d = true;
		else if (!strcmp(argv[i], "-U")) sig_u = true;
		else if (!strcmp(argv[i], "--batch") && i + 1 < argc) batch = atoll(argv[++i]);
		else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
		else if (!strcmp(argv[i], "--sam") && i + 1 < argc) sam_fn = argv[++i];
		else if (!strcmp(argv[i], "--ori-sam") && i + 1 < argc) ori_fn = argv[++i];
		else if (!strcmp(argv[i], "--bam-records") && i + 1 < argc) bam_fn = argv[++i];      // the main file's records as BAM bytes (uncompressed, no header): direct encoder
		else if (!strcmp(argv[i], "--format-reps") && i + 1 < argc) format_reps = atoi(argv[++i]);   // host stages timed on one thread: the batch formatted this many times (stderr)
		else if (!strcmp(argv[i], "--bam-via-text")) bam_text = true;                          // ... through the SAM-line strings and BamWriter::encode instead
		else if (!strcmp(argv[i], "--stream-pos") && i + 1 < argc) sscanf(argv[++i], "%lld,%lld,%lld", &pos[0], &pos[1], &pos[2]);       // start of this shard in the three draw streams
		else if (!strcmp(argv[i], "--rebase-from") && i + 1 < argc) sscanf(argv[++i], "%lld,%lld,%lld", &from[0], &from[1], &from[2]);  // run there first, then rebase to --stream-pos
	}
	const size_t rl = strlen(argv[2]);
	const bool from_bam = rl > 4 && !strcmp(argv[2] + rl - 4, ".bam");
	if (from_bam) {                // the header file is WRITTEN from the BAM's header, like the CLI does
		psvr::BamReader br;
		if (!br.open(argv[2])) { fprintf(stderr, "%s\n", br.error().c_str()); return 2; }
		FILE *h = fopen(argv[3], "w");
		if (!h) return 2;
		fwrite(br.header_text.data(), 1, br.header_text.size(), h);
		fclose(h);
	}
	HostIndex hi;
	hi.keep_sparse = true;       // PSVR_EMU_SPARSE_HASH build: no 2 GiB table on the CPU
	std::string err;
	if (!hi.load_dir(argv[1], argv[3], &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
	DevIndex ix = hi.view();       // host pointers: the CPU backend's "device" is host memory
	psvr_aln_params_t par;
	aln_params_default(&par);
	CpuBE be;
	EngineCore<CpuBE> core(be);
	FastqReader rd;
	FastqBatch fb;
	// <reads> = *.bam: the signal step in this process, its pairs handed to the batch reader without FASTQ text (PairFeed: what `panSVR aln x.bam` does)
	psvr::SignalStep sig;
	psvr::PairFeed feed;
	std::thread sig_thread;
	int sig_rc = 0;
	if (from_bam) {
		sig.o.sort_by_name = sig_n, sig.o.not_use_filter = sig_d, sig.o.discard_full_match = sig_u;
		sig.o.input = argv[2], sig.o.header_fn = argv[3], sig.o.status_fn = std::string(argv[3]) + ".status";
		sig.feed = &feed;
		sig_thread = std::thread([&]() { sig_rc = sig.run(); feed.close(); });
		rd.open_feed(&feed);
	} else if (!rd.open(argv[2])) { fprintf(stderr, "%s\n", rd.error().c_str()); return 2; }
	HeaderInfo H;
	HostSvNames svn;
	svn.h = &hi;
	SamEmitter em;
	FILE *fsam = nullptr, *fori = nullptr, *fbam = nullptr;
	if (bam_fn && !(fbam = fopen(bam_fn, "wb"))) { fprintf(stderr, "cannot open %s\n", bam_fn); return 2; }
	if (sam_fn && ori_fn) {
		if (!H.load(argv[3])) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
		fsam = fopen(sam_fn, "w"), fori = fopen(ori_fn, "w");
		if (!fsam || !fori) { fprintf(stderr, "cannot open the SAM outputs\n"); return 2; }
		fputs(H.text.c_str(), fsam), fputs(H.text.c_str(), fori);
		em.H = &H, em.sv = &svn, em.not_ori = not_ori;
	}
	bool first = true;
	long long pair_base = 0;
	while (rd.read(fb, batch, 100000000, threads)) {
		if (first) {
			rd.stat_params(&par);
			core.init(ix, par);
			first = false;
			em.min_filter_score = par.min_filter_score;
			const long long *st = from[0] >= 0 ? from : pos;
			if (st[0] >= 0) core.grand_pos = st[0], core.hrand_pos[0] = st[1], core.hrand_pos[1] = st[2];
		}
		int rc = core.upload(fb.n_pairs(), fb.bases, fb.base_off, fb.ori);
		if (!rc) rc = core.run(trace, true);
		if (!rc && from[0] >= 0 && pos[0] >= 0) { rc = core.rebase(pos[0], pos[1], pos[2], trace, true); from[0] = -1; }
		if (rc) { fprintf(stderr, "emu error %d: %s\n", rc, core.err.c_str()); return 3; }
		for (long long p = 0; p < fb.n_pairs() && !quiet; ++p) {
			const char *t; int lens[2];
			fb.seq(2 * p, t, lens[0]), fb.seq(2 * p + 1, t, lens[1]);
			psvr_read_result_t rr[2];
			materialize_read(core.c, 2 * p, &rr[0]), materialize_read(core.c, 2 * p + 1, &rr[1]);
			puts(record_json(pair_base + p, rr, core.c.pres[p], &fb.ori[2 * p], lens, core.c.cig.base, trace).c_str());
		}
		if (fsam) {       // the engine's arrays ARE the compact form (headers + candidate list + CIGAR arena)
			ResultView V;
			V.hdr = core.c.rh, V.pairs = core.c.pres, V.cands = core.c.cand, V.cig = core.c.cig.base;
			Bytes a, b;
			if (format_reps > 0) {
				auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
				for (int mode = 0; mode < 2; ++mode) {
					SamEmitter e2 = em;
					e2.as_bam = mode == 1;
					size_t bytes = 0;
					const double t0 = now();
					for (int rep = 0; rep < format_reps; ++rep) {
						a.clear(), b.clear();
						for (long long p = 0; p < fb.n_pairs(); ++p) e2.main_pair(fb, V, p, a), e2.ori_pair(fb, V, p, b);
						bytes = a.size() + b.size();
					}
					const double dt = (now() - t0) / format_reps;
					fprintf(stderr, "[emu] format (%s): %.1f ns per read, %.2f GB/s of output, %zu bytes\n", mode ? "BAM records" : "SAM text", dt * 1e9 / (2.0 * fb.n_pairs()), bytes / dt / 1e9, bytes);
				}
				a.clear(), b.clear();
			}
			for (long long p = 0; p < fb.n_pairs(); ++p) em.main_pair(fb, V, p, a), em.ori_pair(fb, V, p, b);
			fwrite(a.data(), 1, a.size(), fsam), fwrite(b.data(), 1, b.size(), fori);
			if (fbam) {
				SamEmitter eb = em;
				eb.as_bam = true, eb.bam_via_text = bam_text;
				Bytes m;
				for (long long p = 0; p < fb.n_pairs(); ++p) eb.main_pair(fb, V, p, m);
				fwrite(m.data(), 1, m.size(), fbam);
			}
		}
		core.commit();
		fprintf(stderr, "[emu] stream_end %lld %lld %lld\n", core.grand_pos, core.hrand_pos[0], core.hrand_pos[1]);
		fprintf(stderr, "[emu] chain+select: %lld reads by the small case, %lld by the generic pair\n", be.n_small, be.n_generic);
		pair_base += fb.n_pairs();
		fprintf(stderr, "[emu] batch of %lld pairs: %lld rounds, %lld pair-runs (+%lld pairing-only, +%lld shadow, %lld sensitive, %lld window misses), %lld DP problems, %lld candidates\n", fb.n_pairs(), core.stats.rounds,
		        core.stats.pairs_run, core.stats.pair_only, core.stats.shadow_runs, core.stats.sensitive, core.stats.window_miss, core.stats.dp_problems, core.stats.cands);
	}
	if (fsam) fclose(fsam), fclose(fori);
	if (fbam) fclose(fbam);
	feed.abort();
	if (sig_thread.joinable()) sig_thread.join();
	if (sig_rc) { fprintf(stderr, "the signal step failed\n"); return 4; }
	return 0;
}
