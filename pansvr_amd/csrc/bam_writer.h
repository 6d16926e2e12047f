// bam_writer.h -- BGZF + BAM record encoder for the CLI's two output files.
//
// Replaces the reference's text -> sam_parse1 -> sam_write1 route when the output is BAM (`output_BAM`,
// `output_ori_bam` rr.cpp:479-536,656-719 write through htslib; `init_run` opens the files "wb" unless -S).
// The vendored htslib cannot be built in this image, so this encoder follows the published
// format (SAM/BAM specification v1, sections 4.1 BGZF and 4.2 BAM) and the typing rules of htslib's sam_parse1:
//   * integer tags take the smallest type that holds the value (c/C/s/S/i/I),
//   * bin = reg2bin(pos, pos + reference length of the CIGAR, or +1 without one),
//   * sequence as 4-bit codes of "=ACMGRSVTWYHKDBN", qualities as phred (text - 33), '*' -> 0xff.
// Byte-identity with htslib's BGZF blocks is not claimed (block boundaries and deflate output depend on the
// zlib build); the records inside are what the specification prescribes.  tests/test_aln_gpu.py decodes the
// BAM with an independent reader and compares it with the SAM text of the same run.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <atomic>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "worker_pool.h"
#include "deflate_device.h"
#ifdef PSVR_BGZF_ON_DEVICE                            /* (the CLI, which links the engine library: psvr_bgzf_compress) */
#include "../../include/psvr_engine.h"
#endif

namespace psvr {

// A byte buffer whose resize() does not zero what it adds: the record formatters reserve a bound, write through a raw pointer and
// shrink to what they wrote (value-initialising the slack was a second pass over every output byte).
template <class T> struct NoInitAlloc : std::allocator<T> {
	template <class U> struct rebind { typedef NoInitAlloc<U> other; };
	NoInitAlloc() = default;
	template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
	template <class U> void construct(U *p) { ::new ((void *)p) U; }
	template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};
typedef std::vector<uint8_t, NoInitAlloc<uint8_t>> Bytes;

class BgzfWriter {
	FILE *f_ = nullptr;
	std::vector<uint8_t> buf_;
	std::vector<uint8_t> out_;                    // compressed blocks of a flush
	static const size_t kBlock = 0xff00;      // uncompressed bytes per BGZF block (htslib's BGZF_BLOCK_SIZE)
	static const size_t kOut = 0x10000 + 64;
	bool ok_ = true;
	int threads_ = 1;
	int level_ = Z_DEFAULT_COMPRESSION;           // htslib's "wb" is zlib's default level too
	int device_ = -1;                             // >= 0: BGZF members come from psvr_bgzf_compress on that device
	uint8_t *pin_in_ = nullptr, *pin_out_ = nullptr;   // page-locked: the records of a batch, its members
	size_t pin_n_ = 0, pin_out_cap_ = 0;
	// one member: gzip header with the BC extra field, raw deflate, CRC32, ISIZE (SAMv1 4.1); returns the member size
	static size_t compress_block(const uint8_t *p, size_t n, uint8_t *out, int level = Z_DEFAULT_COMPRESSION)
	{
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
		zs.next_in = (Bytef *)p, zs.avail_in = (uInt)n;
		zs.next_out = out + 18, zs.avail_out = (uInt)(kOut - 18 - 8);
		int rc = deflate(&zs, Z_FINISH);
		deflateEnd(&zs);
		if (rc != Z_STREAM_END) return 0;
		const size_t clen = zs.total_out, bsize = clen + 18 + 8 - 1;
		static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
		memcpy(out, hdr, 16);
		out[16] = (uint8_t)(bsize & 0xff), out[17] = (uint8_t)(bsize >> 8);
		const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
		uint8_t *t = out + 18 + clen;
		for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i)), t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
		return clen + 26;
	}
	// the same member from this repository's own encoder (deflate_device.h, the one the device route runs a lane per member, here a host
	// thread per member): greedy LZ77 + one dynamic-Huffman block.  1.6 x zlib level 1's speed per thread for members 8 - 10 % larger.
	static size_t compress_block_fast(const uint8_t *p, size_t n, uint8_t *out)
	{
		static const int hbits = 13;
		static thread_local std::vector<uint8_t> fast;
		static thread_local std::vector<uint32_t> tok;
		if (fast.empty()) fast.resize(df_fast_bytes(hbits) + 16), tok.resize(kDfMaxIn + 16);
		if (n > kDfMaxIn) return 0;
		uint32_t *tk = (uint32_t *)(((uintptr_t)tok.data() + 15) & ~(uintptr_t)15);
		const size_t clen = deflate_block(p, (uint32_t)n, out + 18, (uint32_t)(0x10000 - 26), fast.data(), hbits, tk);
		if (!clen) return 0;
		const size_t bsize = clen + 18 + 8 - 1;
		static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
		memcpy(out, hdr, 16);
		out[16] = (uint8_t)(bsize & 0xff), out[17] = (uint8_t)(bsize >> 8);
		const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
		uint8_t *t = out + 18 + clen;
		for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i)), t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
		return clen + 26;
	}
public:
	static const int kLevelFast = -2;             // `level`: zlib's levels, Z_DEFAULT_COMPRESSION (-1), or this: compress_block_fast
	static size_t compress_block_public(const uint8_t *p, size_t n, uint8_t *out) { return compress_block(p, n, out); }   // out: 0x10000 + 64 bytes
private:
	// blocks are independent: compress them on `threads_` threads, write in order
	void flush_blocks(const uint8_t *p, size_t n)
	{
		const size_t nb = (n + kBlock - 1) / kBlock;
#ifdef PSVR_BGZF_ON_DEVICE
		static const size_t dev_min = getenv("PSVR_BGZF_DEVICE_MIN_BLOCKS") ? (size_t)atoll(getenv("PSVR_BGZF_DEVICE_MIN_BLOCKS")) : 64;   // (tests: small files through the device too)
		if (device_ >= 0 && nb >= dev_min) {
			const size_t need = (size_t)psvr_bgzf_bound((int64_t)n);
			if (need > pin_out_cap_) { if (pin_out_) psvr_host_free(pin_out_); pin_out_ = (uint8_t *)psvr_host_alloc(need), pin_out_cap_ = pin_out_ ? need : 0; }
			int64_t got = 0;
			if (pin_out_ && psvr_bgzf_compress(device_, p, (int64_t)n, pin_out_, (int64_t)pin_out_cap_, &got) == 0) {
				if (fwrite(pin_out_, 1, (size_t)got, f_) != (size_t)got) ok_ = false;
				return;
			}
			fprintf(stderr, "[panSVR-amd] BGZF on the device failed (%s): compressing on the host\n", psvr_last_error());
			device_ = -1;
		}
#endif
		if (out_.size() < nb * kOut) out_.resize(nb * kOut);      // (kept across calls: a fresh vector was 0.8 GB of zeroing per 1 M pairs)
		std::vector<uint8_t> &out = out_;
		std::vector<size_t> len(nb, 0);
		std::atomic<size_t> next(0);
		auto work = [&]() {
			for (size_t b = next++; b < nb; b = next++) {
				const size_t o = b * kBlock, m = n - o < kBlock ? n - o : kBlock;
				len[b] = level_ == kLevelFast ? compress_block_fast(p + o, m, out.data() + b * kOut) : compress_block(p + o, m, out.data() + b * kOut, level_);
			}
		};
		const int nt = threads_ < 1 ? 1 : (size_t)threads_ > nb ? (int)nb : threads_;
		thread_pool().run(nt, [&](int) { work(); });
		for (size_t b = 0; b < nb; ++b) {
			if (!len[b] || fwrite(out.data() + b * kOut, 1, len[b], f_) != len[b]) ok_ = false;
		}
	}
public:
	bool open(const char *fn, int threads = 1, int level = Z_DEFAULT_COMPRESSION) { f_ = fopen(fn, "wb"); threads_ = threads; level_ = level; return f_ != nullptr; }
	// compress on HIP device `d` (psvr_bgzf_compress: a lane per block; the members decode like any other, their bytes are not zlib's)
	void set_device(int d) { device_ = d; }
	void write(const void *p, size_t n)
	{
		const uint8_t *b = (const uint8_t *)p;
#ifdef PSVR_BGZF_ON_DEVICE
		if (device_ >= 0) {
			// the records gather in page-locked memory (the transfer starts from where they lie: out of pageable memory the runtime copies
			// them once more), whole batches go to the device, what is left at close() to the host's zlib
			const size_t cap = kBlock * (size_t)3072;
			if (!pin_in_ && !(pin_in_ = (uint8_t *)psvr_host_alloc(cap))) { device_ = -1; }
			else {
				if (!buf_.empty()) { std::vector<uint8_t> first; first.swap(buf_); write(first.data(), first.size()); }   // (what was written before set_device: the BAM header)
				while (n) {
					const size_t m = cap - pin_n_ < n ? cap - pin_n_ : n;
					memcpy(pin_in_ + pin_n_, b, m), pin_n_ += m, b += m, n -= m;
					if (pin_n_ == cap) flush_blocks(pin_in_, pin_n_), pin_n_ = 0;
				}
				return;
			}
		}
#endif
		buf_.insert(buf_.end(), b, b + n);
		// enough whole blocks to keep every thread busy; on the device a call lasts as long as ONE block takes a lane (tens of ms) however
		// many blocks it holds, so the batches are large
		const size_t batch = device_ >= 0 ? kBlock * (size_t)3072 : kBlock * (size_t)(threads_ < 1 ? 1 : threads_) * 8;
		if (buf_.size() < batch) return;
		const size_t whole = buf_.size() / kBlock * kBlock;
		flush_blocks(buf_.data(), whole);
		buf_.erase(buf_.begin(), buf_.begin() + whole);
	}
	bool close()
	{
		if (!f_) return false;
#ifdef PSVR_BGZF_ON_DEVICE
		if (pin_n_) flush_blocks(pin_in_, pin_n_), pin_n_ = 0;
		if (pin_in_) psvr_host_free(pin_in_), pin_in_ = nullptr;
		if (pin_out_) psvr_host_free(pin_out_), pin_out_ = nullptr, pin_out_cap_ = 0;
#endif
		if (!buf_.empty()) flush_blocks(buf_.data(), buf_.size());
		static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		if (fwrite(eof, 1, 28, f_) != 28) ok_ = false;
		if (fclose(f_) != 0) ok_ = false;
		f_ = nullptr;
		return ok_;
	}
};

// htslib's seq_nt16_table (hts.c:62-80): the 4-bit code a SEQ character is stored as -- IUPAC letters in either case, '=' and
// the digits 0-3 (A, C, G, T); everything else is N.  A record that went through BAM comes back as "=ACMGRSVTWYHKDBN"[code].
inline int nt16_code(char ch)
{
	static const char *nt16 = "=ACMGRSVTWYHKDBN";
	if (ch >= '0' && ch <= '3') return 1 << (ch - '0');
	if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
	const char *q = ch ? strchr(nt16, ch) : nullptr;
	return q ? (int)(q - nt16) : 15;
}
inline char nt16_char(char ch) { return "=ACMGRSVTWYHKDBN"[nt16_code(ch)]; }
struct Nt16Table { uint8_t code[256]; Nt16Table() { for (int c = 0; c < 256; ++c) code[c] = (uint8_t)nt16_code((char)c); } };
inline const uint8_t *nt16_table() { static const Nt16Table t; return t.code; }        // (nt16_code per base is a strchr per base)

struct BamRef { std::string name; uint32_t len; };

inline int bam_reg2bin(int64_t beg, int64_t end)             // SAMv1 section 5.3
{
	--end;
	if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
	if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
	if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
	if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
	if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
	return 0;
}

// one alignment record in SAM terms (what the CLI's emit_record prints)
struct SamFields {
	std::string qname, cigar, seq, qual, tags;   // tags: "\tXX:t:value..." as in the SAM line; cigar/seq/qual may be "*"
	int flag = 0, tid = -1, mapq = 0, mtid = -1, isize = 0;
	int64_t pos1 = 0, mpos1 = 0;                 // 1-based; 0 = unset
};

class BamWriter {
	BgzfWriter z_;
	template <class V> static void put32(V &v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i))); }
	template <class V> static void put16(V &v, uint16_t x) { v.push_back((uint8_t)x), v.push_back((uint8_t)(x >> 8)); }
	static int reg2bin(int64_t beg, int64_t end)             // SAMv1 section 5.3
	{
		--end;
		if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
		if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
		if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
		if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
		if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
		return 0;
	}
	template <class V> static void put_int_tag(V &v, long long x)   // smallest type, as sam_parse1 chooses
	{
		if (x < 0) {
			if (x >= -128) v.push_back('c'), v.push_back((uint8_t)(int8_t)x);
			else if (x >= -32768) v.push_back('s'), put16(v, (uint16_t)(int16_t)x);
			else v.push_back('i'), put32(v, (uint32_t)(int32_t)x);
		} else {
			if (x <= 255) v.push_back('C'), v.push_back((uint8_t)x);
			else if (x <= 65535) v.push_back('S'), put16(v, (uint16_t)x);
			else v.push_back('I'), put32(v, (uint32_t)x);
		}
	}
	template <class V> static bool put_tags(V &v, const std::string &tags)
	{
		size_t i = 0;
		while (i < tags.size()) {
			if (tags[i] == '\t') { ++i; continue; }
			size_t e = tags.find('\t', i);
			if (e == std::string::npos) e = tags.size();
			if (e - i < 5 || tags[i + 2] != ':' || tags[i + 4] != ':') return false;
			v.push_back((uint8_t)tags[i]), v.push_back((uint8_t)tags[i + 1]);
			const char ty = tags[i + 3];
			const std::string val = tags.substr(i + 5, e - (i + 5));
			if (ty == 'i') put_int_tag(v, strtoll(val.c_str(), nullptr, 10));
			else if (ty == 'A') { v.push_back('A'); v.push_back(val.empty() ? ' ' : (uint8_t)val[0]); }
			else if (ty == 'f') { v.push_back('f'); float fl = strtof(val.c_str(), nullptr); uint32_t u; memcpy(&u, &fl, 4); put32(v, u); }
			else if (ty == 'Z' || ty == 'H') { v.push_back((uint8_t)ty); v.insert(v.end(), val.begin(), val.end()); v.push_back(0); }
			else if (ty == 'B') {
				if (val.empty()) return false;
				const char st = val[0];
				std::vector<std::string> items;
				size_t p = 1;
				while (p < val.size()) { size_t q = val.find(',', p + 1); if (q == std::string::npos) q = val.size(); items.push_back(val.substr(p + 1, q - p - 1)); p = q; }
				v.push_back('B'), v.push_back((uint8_t)st), put32(v, (uint32_t)items.size());
				for (const std::string &it : items) {
					if (st == 'f') { float fl = strtof(it.c_str(), nullptr); uint32_t u; memcpy(&u, &fl, 4); put32(v, u); }
					else {
						long long x = strtoll(it.c_str(), nullptr, 10);
						if (st == 'c' || st == 'C') v.push_back((uint8_t)x);
						else if (st == 's' || st == 'S') put16(v, (uint16_t)x);
						else put32(v, (uint32_t)x);
					}
				}
			} else return false;
			i = e;
		}
		return true;
	}
public:
	bool open(const char *fn, const std::string &header_text, const std::vector<BamRef> &refs, int threads = 1, int level = Z_DEFAULT_COMPRESSION)
	{
		if (!z_.open(fn, threads, level)) return false;
		std::vector<uint8_t> h = {'B', 'A', 'M', 1};
		put32(h, (uint32_t)header_text.size());
		h.insert(h.end(), header_text.begin(), header_text.end());
		put32(h, (uint32_t)refs.size());
		for (const BamRef &r : refs) {
			put32(h, (uint32_t)r.name.size() + 1);
			h.insert(h.end(), r.name.begin(), r.name.end());
			h.push_back(0);
			put32(h, r.len);
		}
		z_.write(h.data(), h.size());
		return true;
	}
	// appends one encoded record (block_size + body) to `out`; thread-safe (touches no writer state)
	template <class V> static bool encode(const SamFields &s, V &rec_)
	{
		const size_t base = rec_.size();
		if (s.qname.empty() || s.qname.size() > 254) return false;   // l_read_name is one byte (htslib: "query name too long")
		std::vector<uint32_t> cig;
		int64_t rlen = 0;
		if (!s.cigar.empty() && s.cigar != "*") {
			long long n = 0;
			bool neg = false;
			for (char ch : s.cigar) {
				if (ch == '-') { neg = true; continue; }
				if (ch >= '0' && ch <= '9') { n = n * 10 + (ch - '0'); continue; }
				const char *ops = "MIDNSHP=X", *q = strchr(ops, ch);
				if (!q) return false;
				if (neg) n = -n;
				const int op = (int)(q - ops);
				cig.push_back((uint32_t)n << 4 | (uint32_t)op);
				if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (uint32_t)n & 0xfffffff;
				n = 0, neg = false;
			}
		}
		const bool no_seq = s.seq.empty() || s.seq == "*";
		const uint32_t l_seq = no_seq ? 0 : (uint32_t)s.seq.size();
		const int64_t pos0 = s.pos1 - 1;
		put32(rec_, 0);                                             // block_size, patched below
		put32(rec_, (uint32_t)s.tid), put32(rec_, (uint32_t)(int32_t)pos0);
		rec_.push_back((uint8_t)(s.qname.size() + 1)), rec_.push_back((uint8_t)s.mapq);
		put16(rec_, (uint16_t)reg2bin(pos0 < 0 ? 0 : pos0, (pos0 < 0 ? 0 : pos0) + (rlen > 0 ? rlen : 1)));
		if (cig.size() > 0xffff) { rec_.resize(base); return false; }     // n_cigar_op is 16 bits (no CG:B long-CIGAR tag is written)
		put16(rec_, (uint16_t)cig.size()), put16(rec_, (uint16_t)s.flag);
		put32(rec_, l_seq);
		put32(rec_, (uint32_t)s.mtid), put32(rec_, (uint32_t)(int32_t)(s.mpos1 - 1)), put32(rec_, (uint32_t)s.isize);
		rec_.insert(rec_.end(), s.qname.begin(), s.qname.end());
		rec_.push_back(0);
		for (uint32_t c : cig) put32(rec_, c);
		if (l_seq) {
			const uint8_t *T = nt16_table();
			const size_t at = rec_.size(), nb = (l_seq + 1) / 2;
			rec_.resize(at + nb + l_seq);
			uint8_t *q = &rec_[at];
			const uint8_t *sq = (const uint8_t *)s.seq.data();
			for (uint32_t i = 0; i + 1 < l_seq; i += 2) *q++ = (uint8_t)(T[sq[i]] << 4 | T[sq[i + 1]]);
			if (l_seq & 1) *q++ = (uint8_t)(T[sq[l_seq - 1]] << 4);
			if (s.qual.empty() || s.qual == "*" || s.qual.size() != l_seq) memset(q, 0xff, l_seq);
			else for (uint32_t i = 0; i < l_seq; ++i) q[i] = (uint8_t)(s.qual[i] - 33);
		}
		if (!put_tags(rec_, s.tags)) { rec_.resize(base); return false; }
		const uint32_t bs = (uint32_t)(rec_.size() - base) - 4;
		for (int i = 0; i < 4; ++i) rec_[base + i] = (uint8_t)(bs >> (8 * i));
		return true;
	}
	bool write(const SamFields &s)
	{
		std::vector<uint8_t> rec;
		if (!encode(s, rec)) return false;
		z_.write(rec.data(), rec.size());
		return true;
	}
	void write_raw(const void *p, size_t n) { z_.write(p, n); }
	void set_device(int d) { z_.set_device(d); }
	bool close() { return z_.close(); }
};

} // namespace psvr
