// fastq_batch.h -- step 0 of the reference's pipeline (load_reads, src/PanSVgenerateVCF/read_realignment.cpp:121-152) for the
// MI355X engine: the interleaved FASTQ of the `signal` step, cut into batches and parsed straight into the buffers
// psvr_engine_upload takes.
//
// The reference builds four kstrings per read (kseq_read, clib/utils.c:953); at the engine's rate that allocation work
// is the whole command's wall.  Here nothing is copied except the bases: a batch keeps the raw text (a window of the
// memory-mapped file, or the chunk read from a pipe / gz stream) and one line index; the per-read work -- bases into the
// page-locked upload buffer, parse_ori_mapping_rst's five numbers + flag token -- runs on the `-t` threads; names,
// comments and qualities are looked at again only for the records that are written.
#pragma once
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include "../../include/psvr_engine.h"
#include "worker_pool.h"

namespace psvr {

template <class F> inline void parallel_ranges(long long n, int threads, F &&fn)   // fn(begin, end) over [0, n) on `threads` threads
{
	const int nt = threads < 1 ? 1 : threads;
	if (n <= 0) return;
	const long long per = (n + nt - 1) / nt;
	const int use = (int)((n + per - 1) / per);
	thread_pool().run(use, [&](int t) { fn(t * per, (t + 1) * per < n ? (t + 1) * per : n); });
}

// page-locked when the engine library can provide it (psvr_host_alloc), pageable otherwise; kept by its owner across batches
struct HostBuf {
	void *p = nullptr; size_t cap = 0; bool locked = false;
	void *reserve(size_t bytes)
	{
		if (bytes <= cap) return p;
		release();
		const size_t want = bytes + bytes / 8 + 64;
#ifndef PSVR_NO_ENGINE_LIB                            // (tests/emu builds this header without the engine library)
		if ((p = psvr_host_alloc(want))) locked = true;
		else
#endif
		if (!(p = malloc(want))) { fprintf(stderr, "[panSVR-amd] out of host memory\n"); abort(); }
		cap = want;
		return p;
	}
	void release()
	{
#ifndef PSVR_NO_ENGINE_LIB
		if (p && locked) psvr_host_free(p), p = nullptr;
#endif
		if (p) free(p);
		p = nullptr, cap = 0, locked = false;
	}
	HostBuf() = default;
	HostBuf(const HostBuf &) = delete;
	HostBuf &operator=(const HostBuf &) = delete;
	~HostBuf() { release(); }
};

// atoi() on a token that ends at `e` (the reference calls atoi on strtok_r tokens)
inline int atoi_span(const char *p, const char *e)
{
	while (p < e && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
	bool neg = false;
	if (p < e && (*p == '-' || *p == '+')) neg = *p == '-', ++p;
	unsigned v = 0;                                  // wraps like a 32-bit accumulator instead of being undefined
	while (p < e && *p >= '0' && *p <= '9') v = v * 10u + (unsigned)(*p - '0'), ++p;
	return neg ? (int)(0u - v) : (int)v;
}

// single_end_handler::parse_ori_mapping_rst (rr.hpp:392-429) on the comment [cm, cm + L): strtok_r tokens 0-4 and 9.
// cut[k] = offset of the separator strtok_r overwrote behind token k (-1: none): the reference then turns those into ','
// (rr.hpp:425-427), which is what the RC:Z tag shows; rewrite_comment() reproduces it for the records that are written.
inline psvr_ori_t parse_ori_span(const char *cm, int L, int32_t cut[10])
{
	psvr_ori_t o;
	memset(&o, 0, sizeof o);
	const char *p = cm, *e = cm + L;
	const char *ts[10], *te[10];
	int nt = 0;
	for (; nt < 10; ++nt) {
		while (p < e && *p == '_') ++p;
		if (p >= e) break;
		ts[nt] = p;
		while (p < e && *p != '_') ++p;
		te[nt] = p;
		cut[nt] = p < e ? (int32_t)(p - cm) : -1;
		if (p < e) ++p;
	}
	for (int k = nt; k < 10; ++k) cut[k] = -1, ts[k] = te[k] = e;
	auto num = [&](int k) { return k < nt ? atoi_span(ts[k], te[k]) : 0; };
	o.chr_id = num(0), o.ref_bg = (uint32_t)num(1), o.read_bg = (uint32_t)num(2), o.align_score = (uint32_t)num(3), o.mapq = (uint8_t)num(4);
	o.direction = (nt > 9 && te[9] - ts[9] >= 1 && ts[9][0] == 'F') ? 1 : 0;
	o.unmapped = (nt > 9 && te[9] - ts[9] >= 2 && ts[9][1] == 'Y') ? 1 : 0;
	return o;
}
// the separators alone (what the writer of a record needs of parse_ori_span)
inline void ori_cuts(const char *cm, int L, int32_t cut[10])
{
	const char *p = cm, *e = cm + L;
	int nt = 0;
	for (; nt < 10; ++nt) {
		while (p < e && *p == '_') ++p;
		if (p >= e) break;
		const char *q = (const char *)memchr(p, '_', (size_t)(e - p));
		p = q ? q : e;
		cut[nt] = p < e ? (int32_t)(p - cm) : -1;
		if (p < e) ++p;
	}
	for (int k = nt; k < 10; ++k) cut[k] = -1;
}
// the comment as the reference leaves it after parse_ori_mapping_rst: every separator strtok_r cut becomes ',' except one at
// the very last position, which stays a NUL and ends the C string there
inline void rewrite_comment(const char *cm, int L, std::string &out)
{
	int32_t cut[10];
	parse_ori_span(cm, L, cut);
	out.assign(cm, (size_t)L);
	for (int k = 0; k < 10; ++k) {
		if (cut[k] < 0) continue;
		if (cut[k] < L - 1) out[(size_t)cut[k]] = ',';
		else out.resize((size_t)cut[k]);
	}
}

// where the text comes from: a memory-mapped regular file (nothing is copied; the parse threads fault the pages in), or a
// sequential stream (stdin, a pipe, a .gz file through zlib: the reference's xzopen reads those too, clib/utils.c:44-53)
class TextSource {
	int fd_ = -1;
	gzFile gz_ = nullptr;
	const char *map_ = nullptr;
	size_t map_len_ = 0, map_pos_ = 0;
	std::vector<char> carry_;                        // stream mode: text read past the previous batch's end
	bool eof_ = false;
	std::string err_;

public:
	bool mapped() const { return map_ != nullptr; }
	const std::string &error() const { return err_; }
	bool open(const char *path, bool allow_map = true)
	{
		const bool is_stdin = !strcmp(path, "-");
		fd_ = is_stdin ? 0 : ::open(path, O_RDONLY);
		if (fd_ < 0) { err_ = std::string("fail to open file '") + path + "'"; return false; }
		struct stat st;
		unsigned char magic[2] = {0, 0};
		const bool regular = fstat(fd_, &st) == 0 && S_ISREG(st.st_mode);
		if (regular && pread(fd_, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
			gz_ = gzdopen(fd_, "rb");
			if (!gz_) { err_ = "gzdopen failed"; return false; }
			gzbuffer(gz_, 1 << 20);
			return true;
		}
		if (!regular) {
			// stdin, a pipe, a FIFO: the reference's xzopen / gzdopen decodes a gzip stream there too and passes plain text through
			// (zlib's transparent mode); nothing can be peeked at without consuming it, so zlib gets the descriptor
			gz_ = gzdopen(fd_, "rb");
			if (!gz_) { err_ = "gzdopen failed"; return false; }
			gzbuffer(gz_, 4 << 20);
			return true;
		}
		if (regular && allow_map && st.st_size > 0) {
			void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
			if (m != MAP_FAILED) {
				map_ = (const char *)m, map_len_ = (size_t)st.st_size;
				madvise(m, map_len_, MADV_SEQUENTIAL);
			}
		}
		return true;
	}
	void close()
	{
		if (map_) munmap((void *)map_, map_len_);
		if (gz_) gzclose(gz_);
		else if (fd_ > 0) ::close(fd_);
		map_ = nullptr, gz_ = nullptr, fd_ = -1;
	}
	~TextSource() { close(); }

	// mapped mode: the unread window
	const char *window(size_t *len) const { *len = map_len_ - map_pos_; return map_ + map_pos_; }
	void consume(size_t n) { map_pos_ += n; }
	// stream mode: `buf` starts with the carried tail; appends up to `want` more bytes; false at the end of the stream
	void take_carry(std::vector<char> &buf) { buf.swap(carry_); carry_.clear(); }
	void give_carry(const char *p, size_t n) { carry_.assign(p, p + n); }
	bool more(std::vector<char> &buf, size_t want)
	{
		if (eof_) return false;
		const size_t old = buf.size();
		buf.resize(old + want);
		size_t got = 0;
		while (got < want) {
			long k;
			if (gz_) k = gzread(gz_, buf.data() + old + got, (unsigned)((want - got) < (1u << 30) ? (want - got) : (1u << 30)));
			else k = (long)::read(fd_, buf.data() + old + got, want - got);
			if (k <= 0) { eof_ = true; break; }
			got += (size_t)k;
		}
		buf.resize(old + got);
		return got > 0;
	}
};

// one batch of read pairs: raw text + line index + the engine's input arrays
struct FastqBatch {
	const char *text = nullptr;                      // base of the line offsets (a window of the map, or own.data())
	std::vector<char> own;                           // stream mode: this batch's text
	std::vector<uint64_t> ls;                        // start of every line; ls[4R] = end of the last one
	std::vector<uint16_t> name_end;                  // per read: offset of the name's end inside its header line
	HostBuf bases_buf, off_buf, ori_buf;             // what psvr_engine_upload reads (page-locked)
	char *bases = nullptr; int64_t *base_off = nullptr; psvr_ori_t *ori = nullptr;
	long long R = 0;
	long long n_pairs() const { return R / 2; }

	void line(long long li, const char *&b, int &n) const
	{
		b = text + ls[(size_t)li];
		size_t m = (size_t)(ls[(size_t)li + 1] - ls[(size_t)li]);
		while (m > 0 && (b[m - 1] == '\n' || b[m - 1] == '\r')) --m;
		n = (int)m;
	}
	void name(long long r, const char *&b, int &n) const { int m; line(4 * r, b, m); n = m ? name_end[(size_t)r] - 1 : 0; if (m) ++b; }
	void comment(long long r, const char *&b, int &n) const
	{
		int m;
		line(4 * r, b, m);
		const int ne = name_end[(size_t)r];
		if (ne < m) b += ne + 1, n = m - ne - 1; else b += m, n = 0;
	}
	void seq(long long r, const char *&b, int &n) const { line(4 * r + 1, b, n); }
	void qual(long long r, const char *&b, int &n) const { line(4 * r + 3, b, n); }
};

// f2, the `signal` step fused with `aln`: when <reads> is a BAM, the signal step runs in this process and hands every
// record pair it selects straight to the batch being built -- the record's text (kept for the records that are written: name, comment,
// quality), its bases into the upload array and its original alignment as the numbers the signal step has just computed.  No FASTQ
// text travels through a pipe, no line index is built, no comment is parsed again.  The batch limits are load_reads' (pairs or bases,
// rr.cpp:24,109,126).  One producer (the signal step's thread), one consumer (the pipeline's reader stage).
struct FastqBatch;
class PairFeed {
	std::mutex mu_;
	std::condition_variable cv_;
	FastqBatch *target_ = nullptr;
	long long max_pairs_ = 0, max_bases_ = 0, pairs_ = 0, total_bases_ = 0;
	int in_pair_ = 0;
	bool full_ = false, eof_ = false, aborted_ = false;
	std::vector<char> bases_;
	std::vector<int64_t> off_;
	std::vector<psvr_ori_t> ori_;
	std::vector<uint16_t> name_end_;

public:
	std::string first_comment;
	bool have_first = false;
	// producer: one record (the pair is complete with its second one)
	void put(const char *name, const std::string &comment, const std::string &seq, const std::string &qual, const psvr_ori_t &ori);
	void close() { { std::lock_guard<std::mutex> lk(mu_); eof_ = true; } cv_.notify_all(); }
	void abort() { { std::lock_guard<std::mutex> lk(mu_); aborted_ = true; } cv_.notify_all(); }     // the consumer stopped early (-R): the producer runs to its end unheard
	// consumer: fills B up to the limits; false at the end of the input
	bool fill(FastqBatch &B, long long max_pairs, long long max_bases);
};

class FastqReader {
	TextSource src_;
	PairFeed *feed_ = nullptr;
	size_t est_pair_bytes_ = 1024;                   // bytes of text per pair, refined from the previous batch
	std::vector<int32_t> slen_;                      // scratch: sequence length per candidate read of the batch being cut
	std::string first_comment_;
	bool have_first_ = false;

	// appends the offsets (relative to `base`) of the line starts that follow the newlines in [from, to)
	static void index_lines(const char *base, size_t from, size_t to, int threads, std::vector<uint64_t> &ls)
	{
		const int nt = threads < 1 ? 1 : threads;
		std::vector<std::vector<uint64_t>> part((size_t)nt);
		const size_t span = to - from, per = (span + (size_t)nt - 1) / (size_t)nt;
		auto scan = [&](int t) {
			const size_t a = from + (size_t)t * per, b = a + per < to ? a + per : to;
			std::vector<uint64_t> &v = part[(size_t)t];
			v.reserve((b > a ? b - a : 0) / 64 + 16);
			const char *p = base + a, *e = base + b;
			while (p < e && (p = (const char *)memchr(p, '\n', (size_t)(e - p)))) { ++p; v.push_back((uint64_t)(p - base)); }
		};
		thread_pool().run(per ? (int)((span + per - 1) / per) : 0, scan);
		for (auto &v : part) ls.insert(ls.end(), v.begin(), v.end());
	}

public:
	bool open(const char *path) { return src_.open(path); }
	void open_feed(PairFeed *f) { feed_ = f; }
	const std::string &error() const { return src_.error(); }

	// up to max_pairs pairs or max_bases bases (load_reads stops at 2 M pairs / 100 MB of bases, rr.cpp:24,109,126); false at the
	// end of the input.  `B` keeps its buffers across calls.
	bool read(FastqBatch &B, long long max_pairs, long long max_bases, int threads)
	{
		if (feed_) {
			if (!feed_->fill(B, max_pairs, max_bases)) return false;
			if (!have_first_ && feed_->have_first) first_comment_ = feed_->first_comment, have_first_ = true;
			return true;
		}
		B.R = 0;
		B.ls.clear();
		B.ls.push_back(0);
		const size_t want_lines = (size_t)max_pairs * 8;
		size_t text_len = 0, scanned = 0;
		bool at_end = false;
		if (src_.mapped()) {
			size_t avail;
			B.text = src_.window(&avail);
			size_t upto = 0;
			while (B.ls.size() - 1 < want_lines && upto < avail) {
				const size_t more = est_pair_bytes_ * (size_t)max_pairs / 8 * 9 + (1 << 20);
				const size_t to = upto + more < avail ? upto + more : avail;
				index_lines(B.text, upto, to, threads, B.ls);
				upto = to;
			}
			text_len = upto, scanned = upto, at_end = upto == avail;
		} else {
			B.own.clear();
			src_.take_carry(B.own);
			for (;;) {
				index_lines(B.own.data(), scanned, B.own.size(), threads, B.ls);
				scanned = B.own.size();
				if (B.ls.size() - 1 >= want_lines) break;
				if (!src_.more(B.own, est_pair_bytes_ * (size_t)max_pairs / 8 * 9 + (1 << 20))) { at_end = true; break; }
			}
			B.text = B.own.data(), text_len = B.own.size();
		}
		if (at_end && B.ls.back() < text_len) B.ls.push_back(text_len);          // last line without a newline
		size_t nlines = B.ls.size() - 1;
		if (nlines > want_lines) nlines = want_lines;
		long long npairs = (long long)(nlines / 8);
		// sequence length of every candidate read, on the threads (a line's end is a cache miss each: not a loop for one core)
		std::vector<int32_t> &slen = slen_;
		slen.resize((size_t)(2 * npairs));
		parallel_ranges(2 * npairs, threads, [&](long long r0, long long r1) {
			for (long long r = r0; r < r1; ++r) {
				const size_t li = (size_t)(4 * r + 1);
				size_t m = (size_t)(B.ls[li + 1] - B.ls[li]);
				const char *b = B.text + B.ls[li];
				while (m > 0 && (b[m - 1] == '\n' || b[m - 1] == '\r')) --m;
				slen[(size_t)r] = (int32_t)m;
			}
		});
		// the 100 MB limit: load_reads stops BEFORE a pair once the bases loaded so far reach it
		{
			long long total = 0, keep = 0;
			for (long long p = 0; p < npairs && total < max_bases; ++p, ++keep) total += slen[(size_t)(2 * p)] + slen[(size_t)(2 * p + 1)];
			npairs = keep;
		}
		const size_t used = (size_t)B.ls[(size_t)npairs * 8];
		B.ls.resize((size_t)npairs * 8 + 1);
		if (src_.mapped()) src_.consume(used);
		else src_.give_carry(B.own.data() + used, text_len - used);
		if (npairs == 0) return false;
		est_pair_bytes_ = used / (size_t)npairs + 1;
		const long long R = 2 * npairs;
		B.R = R;
		B.name_end.resize((size_t)R);
		B.base_off = (int64_t *)B.off_buf.reserve((size_t)(R + 1) * 8);
		B.ori = (psvr_ori_t *)B.ori_buf.reserve((size_t)R * sizeof(psvr_ori_t));
		B.base_off[0] = 0;
		for (long long r = 0; r < R; ++r) B.base_off[r + 1] = B.base_off[r] + slen[(size_t)r];
		B.bases = (char *)B.bases_buf.reserve((size_t)B.base_off[R] + 16);
		parallel_ranges(R, threads, [&](long long r0, long long r1) {
			for (long long r = r0; r < r1; ++r) {
				const char *b; int n;
				B.line(4 * r, b, n);
				int sp = n ? 1 : 0;
				while (sp < n && b[sp] != ' ' && b[sp] != '\t') ++sp;
				B.name_end[(size_t)r] = (uint16_t)(sp > 65535 ? 65535 : sp);
				int32_t cut[10];
				B.ori[r] = sp < n ? parse_ori_span(b + sp + 1, n - sp - 1, cut) : parse_ori_span(b + n, 0, cut);
				B.seq(r, b, n);
				memcpy(B.bases + B.base_off[r], b, (size_t)n);
			}
		});
		B.bases[B.base_off[R]] = 0;
		if (!have_first_) { const char *b; int n; B.comment(0, b, n); first_comment_.assign(b, (size_t)n); have_first_ = true; }
		return true;
	}

	// STAT_ of the very first read (load_reads, rr.cpp:134-148)
	void stat_params(psvr_aln_params_t *p) const
	{
		int rl = 150, mn = 100, mid = 500, mx = 900;
		const char *st = strstr(first_comment_.c_str(), "STAT_");
		if (!st || sscanf(st + 5, "%d_%d_%d_%d_", &rl, &mn, &mid, &mx) == -1) rl = 150, mn = 100, mid = 500, mx = 900;
		p->normal_read_length = rl, p->isize_min = mn, p->isize_max = mx;
		int mfs = rl * p->match * 2 - 80;
		p->min_filter_score = mfs > 50 ? mfs : 50;
	}
};

inline void PairFeed::put(const char *name, const std::string &comment, const std::string &seq, const std::string &qual, const psvr_ori_t &ori)
{
	std::unique_lock<std::mutex> lk(mu_);
	cv_.wait(lk, [&] { return aborted_ || (target_ && !full_); });
	if (aborted_) return;
	FastqBatch &B = *target_;
	if (!have_first) first_comment = comment, have_first = true;
	// the record as the FASTQ file would hold it: @name comment \n seq \n + \n qual \n  (bam2fastqWrite_additional_str_gz, getSignalRead.cpp:15-34)
	std::vector<char> &t = B.own;
	const size_t nl = strlen(name);
	t.push_back('@');
	t.insert(t.end(), name, name + nl);
	t.push_back(' ');
	t.insert(t.end(), comment.begin(), comment.end());
	t.push_back('\n');
	B.ls.push_back(t.size());
	t.insert(t.end(), seq.begin(), seq.end());
	t.push_back('\n');
	B.ls.push_back(t.size());
	t.push_back('+'), t.push_back('\n');
	B.ls.push_back(t.size());
	t.insert(t.end(), qual.begin(), qual.end());
	t.push_back('\n');
	B.ls.push_back(t.size());
	name_end_.push_back((uint16_t)(nl + 1 > 65535 ? 65535 : nl + 1));
	bases_.insert(bases_.end(), seq.begin(), seq.end());
	off_.push_back((int64_t)bases_.size());
	ori_.push_back(ori);
	total_bases_ += (long long)seq.size();
	if (++in_pair_ == 2) {
		in_pair_ = 0;
		++pairs_;
		if (pairs_ >= max_pairs_ || total_bases_ >= max_bases_) { full_ = true; lk.unlock(); cv_.notify_all(); }
	}
}

inline bool PairFeed::fill(FastqBatch &B, long long max_pairs, long long max_bases)
{
	std::unique_lock<std::mutex> lk(mu_);
	B.R = 0;
	B.own.clear(), B.ls.clear(), B.ls.push_back(0);
	bases_.clear(), off_.clear(), off_.push_back(0), ori_.clear(), name_end_.clear();
	pairs_ = 0, total_bases_ = 0, full_ = false;
	max_pairs_ = max_pairs, max_bases_ = max_bases;
	target_ = &B;
	cv_.notify_all();
	cv_.wait(lk, [&] { return full_ || eof_; });
	target_ = nullptr;
	const long long R = 2 * pairs_;                           // (a trailing record without its mate cannot arrive: the signal step writes pairs)
	if (R == 0) return false;
	B.R = R;
	B.text = B.own.data();
	B.ls.resize((size_t)(4 * R + 1));
	B.name_end.assign(name_end_.begin(), name_end_.begin() + R);
	B.base_off = (int64_t *)B.off_buf.reserve((size_t)(R + 1) * 8);
	B.ori = (psvr_ori_t *)B.ori_buf.reserve((size_t)R * sizeof(psvr_ori_t));
	memcpy(B.base_off, off_.data(), (size_t)(R + 1) * 8);
	memcpy(B.ori, ori_.data(), (size_t)R * sizeof(psvr_ori_t));
	B.bases = (char *)B.bases_buf.reserve((size_t)B.base_off[R] + 16);
	memcpy(B.bases, bases_.data(), (size_t)B.base_off[R]);
	B.bases[B.base_off[R]] = 0;
	return true;
}

} // namespace psvr
