// sam_emit.h -- step 2 of the reference's pipeline for the MI355X engine: the two output files' records from the engine's
// compact results.  Mirrors single_end_handler::output_BAM / output_ori_bam and the filter of align_read_pair
// (src/PanSVgenerateVCF/read_realignment.cpp:479-536, 656-719, 776-797) as they come out of htslib 1.9's
// sam_parse1 -> sam_format1 round trip (POS <= 0 drops the record, RNEXT collapses to '=', a trailing tab is ignored ...).
// Pinned against the reference's own output: tests/golden/*/<reads>.sam.gz and .ori.sam.gz are `fc_aln -t 1 -S` files
// written by the reference objects (oracle/ref_harness); tests/test_sam_golden.py compares byte for byte.
// Only the pairs that are written are looked at: names, comments and qualities stay in the batch's raw text until here.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <atomic>
#include <string>
#include <vector>
#include "../../include/psvr_engine.h"
#include "bam_writer.h"
#include "fastq_batch.h"

namespace psvr {

struct HeaderInfo {
	std::string text;
	std::vector<std::string> names;
	std::vector<uint32_t> lens;
	const char *name(int id) const { return id >= 0 && id < (int)names.size() ? names[(size_t)id].c_str() : "*"; }
	bool load(const std::string &fn)
	{
		FILE *f = fopen(fn.c_str(), "r");
		if (!f) return false;
		char *buf = nullptr;
		size_t cap = 0;
		while (getline(&buf, &cap, f) > 0) {                 // header lines of any length
			if (buf[0] != '@') continue;
			text += buf;
			if (strncmp(buf, "@SQ", 3)) continue;
			char *p = strstr(buf, "SN:");
			if (!p) continue;
			p += 3;
			char *e = p;
			while (*e && *e != '\t' && *e != '\n') ++e;
			names.emplace_back(p, e - p);
			const char *ln = strstr(buf, "LN:");
			lens.push_back(ln ? (uint32_t)strtoul(ln + 3, nullptr, 10) : 0u);
		}
		free(buf);
		fclose(f);
		return true;
	}
};

// the results of a run of pairs, in the engine's compact form (psvr_engine_download_compact, or the emulation's arrays)
struct ResultView {
	const psvr_read_hdr_t *hdr = nullptr;            // [2 * pairs]
	const psvr_pair_result_t *pairs = nullptr;       // [pairs]
	const psvr_cand_t *cands = nullptr;              // hdr.cand_off indexes this
	const uint32_t *cig = nullptr;                   // cand.cigar_off indexes this
	long long pair0 = 0;                             // batch-local index of pairs[0] (a device's block of the batch)
};

struct SvNames {                                     // SV_chr_info::vcf_print_string / vcf_id per anchor (deBGA_index.hpp:116-119)
	virtual const char *print_string(int sv) const = 0;
	virtual const char *vcf_id(int sv) const = 0;
	virtual ~SvNames() {}
};

inline char sam_rc_char(char c)                      // getReverseChar, clib/bam_file.c:316-327
{
	switch (c) {
	case 'A': case 'a': return 'T';
	case 'C': case 'c': return 'G';
	case 'G': case 'g': return 'C';
	case 'T': case 't': return 'A';
	}
	return 'N';
}
inline void sam_rev_seq(std::string &s)              // getReverseStr_char, clib/bam_file.c:329-339
{
	const int len = (int)s.size(), half = len >> 1;
	for (int i = 0; i < half; i++) { const char t = s[(size_t)i]; s[(size_t)i] = sam_rc_char(s[(size_t)(len - 1 - i)]); s[(size_t)(len - 1 - i)] = sam_rc_char(t); }
	if (len & 1) s[(size_t)half] = sam_rc_char(s[(size_t)half]);
}
inline void sam_rev_qual(std::string &q)             // getReverseStr_qual_char, clib/bam_file.c:351-359: bound len/2 + 1 (even len: the middle pair is swapped back)
{
	const int len = (int)q.size(), half = len >> 1;
	for (int i = 0; i < half + 1; i++) { const int ri = len - 1 - i; if (ri < 0 || i >= len) break; const char t = q[(size_t)i]; q[(size_t)i] = q[(size_t)ri]; q[(size_t)ri] = t; }
}

struct EmitStats { std::atomic<long long> dropped{0}; };

// the SAM text of a record goes into the output buffer through a raw pointer: the constructor reserves a bound for the record,
// the fields are appended without a capacity check each, close() gives back what was not used
struct RawOut {
	Bytes &v;
	size_t at;
	uint8_t *p;
	RawOut(Bytes &buf, size_t bound) : v(buf), at(buf.size())
	{
		if (v.capacity() < at + bound) v.reserve((at + bound) * 2 + (1 << 16));
		v.resize(at + bound);                          // (Bytes does not zero what it adds)
		p = v.data() + at;
	}
	void close() { v.resize((size_t)(p - v.data())); }
	void put(const char *s, size_t n) { memcpy(p, s, n), p += n; }
	void put(const std::string &s) { put(s.data(), s.size()); }
	void ch(char c) { *p++ = (uint8_t)c; }
	void num(long long x)                              // %lld
	{
		static const char d2[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869707172737475767778798081828384858687888990919293949596979899";
		unsigned long long u = (unsigned long long)x;
		if (x < 0) *p++ = '-', u = 0ull - u;
		// (flags, qualities, scores, insert sizes: mostly one to three digits)
		if (u < 10) { *p++ = (uint8_t)('0' + u); return; }
		if (u < 100) { p[0] = (uint8_t)d2[2 * u], p[1] = (uint8_t)d2[2 * u + 1], p += 2; return; }
		if (u < 1000) { const unsigned h = (unsigned)u / 100, r = (unsigned)u - 100 * h; p[0] = (uint8_t)('0' + h), p[1] = (uint8_t)d2[2 * r], p[2] = (uint8_t)d2[2 * r + 1], p += 3; return; }
		char b[24];
		int n = 24;
		while (u >= 100) { const unsigned r = (unsigned)(u % 100); u /= 100; b[--n] = d2[2 * r + 1], b[--n] = d2[2 * r]; }
		if (u >= 10) b[--n] = d2[2 * u + 1], b[--n] = d2[2 * u]; else b[--n] = (char)('0' + u);
		for (; n < 24; ++n) *p++ = (uint8_t)b[n];
	}
	void tag_int(const char *tag5, long long x) { ch('\t'), put(tag5, 5), num(x); }
	// ---- BAM (little-endian binary)
	void u8(unsigned x) { *p++ = (uint8_t)x; }
	void u16(unsigned x) { p[0] = (uint8_t)x, p[1] = (uint8_t)(x >> 8), p += 2; }
	void u32(uint32_t x) { p[0] = (uint8_t)x, p[1] = (uint8_t)(x >> 8), p[2] = (uint8_t)(x >> 16), p[3] = (uint8_t)(x >> 24), p += 4; }
	void bam_int(const char *tag2, long long x)        // an integer tag in the smallest type, as sam_parse1 chooses (BamWriter::put_int_tag)
	{
		put(tag2, 2);
		if (x < 0) {
			if (x >= -128) u8('c'), u8((unsigned)(int8_t)x);
			else if (x >= -32768) u8('s'), u16((unsigned)(uint16_t)(int16_t)x);
			else u8('i'), u32((uint32_t)(int32_t)x);
		} else {
			if (x <= 255) u8('C'), u8((unsigned)x);
			else if (x <= 65535) u8('S'), u16((unsigned)x);
			else u8('I'), u32((uint32_t)x);
		}
	}
	void bam_z(const char *tag2) { put(tag2, 2), u8('Z'); }   // the value and its NUL follow
};
// SEQ / QUAL of a record: forward = the read through htslib's 4-bit code (nt16_char), reverse = getReverseStr_char /
// getReverseStr_qual_char (the even-length quirk: the middle pair is swapped back) -- byte tables instead of a switch per base
struct SeqTables {
	uint8_t rc[256], n16[256];
	uint8_t c16[256], rc16[256];                     // the 4-bit BAM code of a base / of its reverse-strand character
	SeqTables()
	{
		for (int c = 0; c < 256; ++c) {
			rc[c] = (uint8_t)sam_rc_char((char)c), n16[c] = (uint8_t)nt16_char((char)c);
			c16[c] = (uint8_t)nt16_code((char)c), rc16[c] = (uint8_t)nt16_code(sam_rc_char((char)c));
		}
	}
};
inline const SeqTables &seq_tables() { static const SeqTables t; return t; }
#if defined(__SSE2__)
// sixteen bytes at a time where every one of them is A, C, G, T or N (what the table maps to itself / to its complement); a block with
// anything else goes through the tables
struct Seq16 {
	__m128i a, c, g, t, n;
	Seq16() : a(_mm_set1_epi8('A')), c(_mm_set1_epi8('C')), g(_mm_set1_epi8('G')), t(_mm_set1_epi8('T')), n(_mm_set1_epi8('N')) {}
	static __m128i reversed(__m128i v)
	{
		v = _mm_shuffle_epi32(v, _MM_SHUFFLE(0, 1, 2, 3));
		v = _mm_shufflelo_epi16(v, _MM_SHUFFLE(2, 3, 0, 1)), v = _mm_shufflehi_epi16(v, _MM_SHUFFLE(2, 3, 0, 1));
		return _mm_or_si128(_mm_slli_epi16(v, 8), _mm_srli_epi16(v, 8));
	}
	// all sixteen plain?  *comp = their complements
	bool plain(__m128i v, __m128i *comp) const
	{
		const __m128i ia = _mm_cmpeq_epi8(v, a), ic = _mm_cmpeq_epi8(v, c), ig = _mm_cmpeq_epi8(v, g), it = _mm_cmpeq_epi8(v, t), in = _mm_cmpeq_epi8(v, n);
		if (_mm_movemask_epi8(_mm_or_si128(_mm_or_si128(_mm_or_si128(ia, ic), _mm_or_si128(ig, it)), in)) != 0xffff) return false;
		if (comp) *comp = _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_and_si128(ia, t), _mm_and_si128(ic, g)), _mm_or_si128(_mm_and_si128(ig, c), _mm_and_si128(it, a))), _mm_and_si128(in, n));
		return true;
	}
};
#endif
inline void put_seq_qual(RawOut &o, const char *t, const char *qt, int n, bool reverse)
{
	const SeqTables &T = seq_tables();
	uint8_t *sq = o.p, *ql = sq + n + 1;
	int i = 0;
	if (reverse) {
#if defined(__SSE2__)
		const Seq16 K;
		for (; i + 16 <= n; i += 16) {
			const __m128i v = _mm_loadu_si128((const __m128i *)(t + n - 16 - i));
			__m128i comp;
			if (K.plain(v, &comp)) _mm_storeu_si128((__m128i *)(sq + i), Seq16::reversed(comp));
			else for (int j = i; j < i + 16; ++j) sq[j] = T.rc[(uint8_t)t[n - 1 - j]];
			_mm_storeu_si128((__m128i *)(ql + i), Seq16::reversed(_mm_loadu_si128((const __m128i *)(qt + n - 16 - i))));
		}
#endif
		for (int j = i; j < n; ++j) sq[j] = T.rc[(uint8_t)t[n - 1 - j]];      // A C G T N only: already what the 4-bit code gives back
		for (int j = i; j < n; ++j) ql[j] = (uint8_t)qt[n - 1 - j];
		if (!(n & 1) && n >= 2) ql[n / 2 - 1] = (uint8_t)qt[n / 2 - 1], ql[n / 2] = (uint8_t)qt[n / 2];   // loop bound len/2 + 1: the middle pair is swapped back
	} else {
#if defined(__SSE2__)
		const Seq16 K;
		for (; i + 16 <= n; i += 16) {
			const __m128i v = _mm_loadu_si128((const __m128i *)(t + i));
			if (K.plain(v, nullptr)) _mm_storeu_si128((__m128i *)(sq + i), v);
			else for (int j = i; j < i + 16; ++j) sq[j] = T.n16[(uint8_t)t[j]];
		}
#endif
		for (; i < n; ++i) sq[i] = T.n16[(uint8_t)t[i]];
		memcpy(ql, qt, (size_t)n);
	}
	sq[n] = '\t';
	o.p += 2 * n + 1;
}

class SamEmitter {
	friend struct EmitCheck;                         // tests/tools/emit_check.cpp: the fast scanners against the C library's
public:
	const HeaderInfo *H = nullptr;
	const SvNames *sv = nullptr;
	bool as_bam = false, not_ori = false;
	bool bam_via_text = false;                       // tests: every BAM record through the SAM-line strings and BamWriter::encode (the direct encoder must give the same bytes)
	int min_filter_score = 520;
	EmitStats *stats = nullptr;

private:
	static void put(Bytes &d, const char *p, size_t n) { d.insert(d.end(), (const uint8_t *)p, (const uint8_t *)p + n); }
	static void put(Bytes &d, const std::string &s) { put(d, s.data(), s.size()); }
	static void put_int(Bytes &d, long long v)          // %lld without the format machinery
	{
		char b[24];
		int n = 24;
		unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
		do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
		if (v < 0) b[--n] = '-';
		put(d, b + n, (size_t)(24 - n));
	}
	static void put_tag_int(Bytes &d, const char *tag5, long long v) { put(d, "\t", 1), put(d, tag5, 5), put_int(d, v); }
	static void cigar_text(const psvr_cand_t &c, const uint32_t *cig, std::string &s)
	{
		char b[32];
		for (uint32_t j = 0; j < c.n_cigar; ++j) { const uint32_t w = cig[c.cigar_off + j]; const int n = snprintf(b, sizeof b, "%d%c", (int)(int16_t)(w >> 4), "MIDNSHP=XB"[w & 0xf]); s.append(b, (size_t)n); }
	}
	void drop(const char *what) const
	{
		if (stats) stats->dropped++;
		fprintf(stderr, "%s\n", what);                // the reference reports a record sam_parse1 refuses the same way (rr.cpp:532,716)
	}
	// what sam_parse1 refuses (htslib 1.9 sam.c:1197-1424) among the texts this step can produce.  One rule for both output
	// modes, so the BAM and the SAM file of the same input hold the same records.
	static bool acceptable(const std::string &qname, const std::string &cigar, const std::string &seq, const std::string &qual)
	{
		if (qname.empty() || qname.size() > 254) return false;                    // "query name too long"
		if (!acceptable_cigar(cigar)) return false;
		if (seq != "*" && qual != "*" && seq.size() != qual.size()) return false;  // "SEQ and QUAL are of different length"
		return true;
	}
	static bool acceptable_cigar(const std::string &cigar)
	{
		if (!cigar.empty() && cigar != "*") {
			bool digit = false;
			for (char ch : cigar) {
				if (ch >= '0' && ch <= '9') { digit = true; continue; }
				if (ch == '-' && !digit) continue;                                   // strtol takes a sign (the reference can print negative lengths)
				if (!digit || !strchr("MIDNSHP=XB", ch)) return false;               // "unrecognized CIGAR operator"
				digit = false;
			}
			if (digit) return false;
		}
		return true;
	}
	// one record after the sam_parse1 -> sam_format1 round trip
	bool emit(Bytes &dst, const std::string &name, int flag, int chr_id, uint32_t ref_bg, int mapq, const std::string &cigar, bool has_mate, int mate_chr, uint32_t mate_pos,
	          int isize, const std::string &seq, const std::string &qual, const std::string &tags, const char *err_line) const
	{
		const int pos = (int)ref_bg;                               // printed with %d
		if (chr_id < 0 || chr_id >= (int)H->names.size()) return false;   // target_name[] would be indexed out of range in the reference
		if (pos - 1 < 0) return false;                             // "mapped query cannot have zero coordinate; treated as unmapped" -> tid = -1 -> not written
		if (!acceptable(name, cigar, seq, qual)) { drop(err_line); return false; }
		const char *rnext = "*";
		long pnext = 0;
		int mtid = -1;
		if (has_mate) {
			const int mp = (int)mate_pos;
			const bool mate_ok = mate_chr >= 0 && mate_chr < (int)H->names.size() && !(mp - 1 < 0);
			if (mate_ok) rnext = mate_chr == chr_id ? "=" : H->name(mate_chr), mtid = mate_chr;
			pnext = mp;
		}
		if (as_bam) {
			SamFields f;
			f.qname = name, f.flag = flag, f.tid = chr_id, f.pos1 = pos, f.mapq = mapq, f.cigar = cigar.empty() ? "*" : cigar;
			f.mtid = mtid, f.mpos1 = pnext, f.isize = isize, f.seq = seq, f.qual = qual, f.tags = tags;
			if (!BamWriter::encode(f, dst)) { drop(err_line); return false; }
			return true;
		}
		put(dst, name), put(dst, "\t", 1), put_int(dst, flag), put(dst, "\t", 1), put(dst, H->name(chr_id), strlen(H->name(chr_id))), put(dst, "\t", 1);
		put_int(dst, pos), put(dst, "\t", 1), put_int(dst, mapq), put(dst, "\t", 1);
		if (cigar.empty()) put(dst, "*", 1); else put(dst, cigar);
		put(dst, "\t", 1), put(dst, rnext, strlen(rnext)), put(dst, "\t", 1), put_int(dst, pnext), put(dst, "\t", 1), put_int(dst, isize), put(dst, "\t", 1);
		// SEQ as it comes back from the 4-bit BAM encoding sam_parse1 stores it in (lower case -> upper, non-IUPAC -> N)
		const size_t at = dst.size();
		put(dst, seq);
		if (seq != "*") for (size_t i = at; i < dst.size(); ++i) dst[i] = (uint8_t)nt16_char((char)dst[i]);
		put(dst, "\t", 1), put(dst, qual), put(dst, tags), put(dst, "\n", 1);
		return true;
	}

	struct OriRecord { int flag = 0, mapq = 0, mate_chr = -1, mate_pos = 0, isize = 0; std::string cigar, tags; };
	// single_end_handler::output_ori_bam (rr.cpp:656-717): the ORIGINAL alignment from the comment's FLAG_/CIGAR_/MATE_/TAG_ sections
	static bool parse_ori_record(const std::string &comment, OriRecord *r)
	{
		const char *c = comment.c_str();
		const char *f = strstr(c, "FLAG_");
		if (!f) return false;
		unsigned fl = 0, q = 0;
		if (sscanf(f + 5, "%u_%u_", &fl, &q) < 2) return false;
		r->flag = (int)fl, r->mapq = (int)q;
		const char *cg = strstr(f + 5, "CIGAR_");
		if (!cg) return false;
		cg += 6;
		const char *ce = strchr(cg, '_');
		if (!ce) return false;
		r->cigar.assign(cg, ce - cg);
		const char *mate = ce + 1 + 5;                           // skips "MATE_"
		if (strlen(ce) < 6 || sscanf(mate, "%d_%d_%d_", &r->mate_chr, &r->mate_pos, &r->isize) < 3) return false;
		r->mate_pos += 1;
		const char *tg = strstr(mate, "TAG_");
		if (!tg) return false;
		std::string tags = tg + 4;
		const int tl = (int)tags.size();
		for (int i = 0; i < tl - 5; i++) if (tags[(size_t)i] == '_' && tags[(size_t)i + 3] == ':' && tags[(size_t)i + 5] == ':') tags[(size_t)i] = '\t';
		if (tl > 0) tags.resize((size_t)tl - 1);
		r->tags = tags;
		return true;
	}
	// the same from the comment's span without the C library's scanners (two sscanf and three strstr per read were most of what the
	// second file's records cost).  Numbers that are plain runs of one to nine digits (a '-' in front where %d takes one) are read here;
	// anything else -- white space, a '+', more digits -- is left to the function above, so both give the same record.
	static bool parse_ori_record(const char *ct, int cn, OriRecord *r)
	{
		const char *c = ct, *e = ct + strnlen(ct, (size_t)cn);
		auto slow = [&]() { return parse_ori_record(std::string(ct, (size_t)cn), r); };
		auto find = [&](const char *from, const char *lit, size_t n) { return from <= e ? (const char *)memmem(from, (size_t)(e - from), lit, n) : nullptr; };
		// 0: not a plain number (caller falls back); 1: read
		auto number = [&](const char *&p, bool sign, long long *v) {
			const char *q = p;
			bool neg = false;
			if (sign && q < e && *q == '-') neg = true, ++q;
			const char *d0 = q;
			long long x = 0;
			while (q < e && *q >= '0' && *q <= '9' && q - d0 < 10) x = x * 10 + (*q - '0'), ++q;
			if (q == d0 || q - d0 > 9) return 0;
			*v = neg ? -x : x, p = q;
			return 1;
		};
		const char *f = find(c, "FLAG_", 5);
		if (!f) return false;
		const char *p = f + 5;
		long long fl, q, a, b, d;
		if (!number(p, false, &fl) || p >= e || *p != '_') return slow();
		++p;
		if (!number(p, false, &q)) return slow();
		r->flag = (int)fl, r->mapq = (int)q;
		const char *cg = find(f + 5, "CIGAR_", 6);
		if (!cg) return false;
		cg += 6;
		const char *ce = cg <= e ? (const char *)memchr(cg, '_', (size_t)(e - cg)) : nullptr;
		if (!ce) return false;
		r->cigar.assign(cg, (size_t)(ce - cg));
		if (e - ce < 6) return false;
		p = ce + 1 + 5;                                          // skips "MATE_"
		if (!number(p, true, &a) || p >= e || *p != '_') return slow();
		++p;
		if (!number(p, true, &b) || p >= e || *p != '_') return slow();
		++p;
		if (!number(p, true, &d)) return slow();
		r->mate_chr = (int)a, r->mate_pos = (int)b + 1, r->isize = (int)d;
		const char *tg = find(ce + 1 + 5, "TAG_", 4);
		if (!tg) return false;
		r->tags.assign(tg + 4, (size_t)(e - (tg + 4)));
		std::string &tags = r->tags;
		const int tl = (int)tags.size();
		for (int i = 0; i < tl - 5; i++) if (tags[(size_t)i] == '_' && tags[(size_t)i + 3] == ':' && tags[(size_t)i + 5] == ':') tags[(size_t)i] = '\t';
		if (tl > 0) tags.resize((size_t)tl - 1);
		return true;
	}
	// bam_has_clip_or_unmapped_ori (rr.cpp:721-733) on the CIGAR text
	static bool ori_has_clip(const std::string &cigar, int min_clip)
	{
		if (cigar.empty() || cigar == "*") return true;
		int first_len = 0, last_len = 0, n = 0, ops = 0;
		char first_op = 0, last_op = 0;
		for (char ch : cigar) {
			if (ch >= '0' && ch <= '9') { n = n * 10 + (ch - '0'); continue; }
			if (ops++ == 0) first_op = ch, first_len = n;
			last_op = ch, last_len = n, n = 0;
		}
		if (!ops) return true;
		int tot = 0;
		if (first_op == 'S' || first_op == 'H') tot += first_len;
		if (last_op == 'S' || last_op == 'H') tot += last_len;
		return tot >= min_clip;
	}

public:
	// output_BAM for both reads of pair p (batch-local index); r = local read index base 2 * p
	void main_pair(const FastqBatch &B, const ResultView &V, long long p, Bytes &dst) const
	{
		const psvr_pair_result_t &pr = V.pairs[p - V.pair0];
		if (!pr.gain) return;
		std::string name, cm, seq, qual, cg, tags;
		char b[256];
		for (int k = 0; k < 2; ++k) {
			const long long r = 2 * p + k;
			const psvr_read_hdr_t &rr = V.hdr[r - 2 * V.pair0];
			const psvr_ori_t &ori = B.ori[r];
			if (rr.primary == -1) continue;                          // primary_result == NULL
			const bool is_ori = rr.primary == -2;
			if (not_ori && is_ori) continue;
			int chr_id, direction, mapq;
			uint32_t ref_bg, align_score, chain_score = 0;
			const char *t; int n;
			B.seq(r, t, n);
			const int read_l = n;
			cg.clear();
			const psvr_cand_t *pcd = nullptr;
			if (is_ori) {
				chr_id = ori.chr_id, direction = ori.direction, mapq = ori.mapq, ref_bg = ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg, align_score = ori.align_score;
				if (ori.read_bg > 0) { snprintf(b, sizeof b, "%dS", (int)(int16_t)(uint16_t)ori.read_bg); cg += b; }
				snprintf(b, sizeof b, "%dM", (int)(int16_t)(uint16_t)(read_l - (int)ori.read_bg));
				cg += b;
			} else {
				const psvr_cand_t &cd = V.cands[rr.cand_off + rr.primary];
				chr_id = cd.chr_id, direction = cd.direction, mapq = cd.mapq, ref_bg = cd.ref_bg, align_score = cd.align_score, chain_score = cd.chain_score;
				if (!as_bam) pcd = &cd;                                       // (both paths print / encode the operations straight into the record)
			}
			if ((uint32_t)chr_id == 0xffffffffu) continue;           // primary_result->chrID == MAX_uint32_t
			const int flag = (uint8_t)((k == 0 ? 0x40 : 0) + (direction == 0 ? 0x10 : 0) + (rr.has_mate ? 0 : 0x8));
			const int isize = direction == 1 ? pr.cur_isize : -pr.cur_isize;
			if (!as_bam) {
				// SAM text straight into the output buffer (the string-building path below serves the BAM encoder): same fields, same rules
				const char *qt, *nt, *ct; int qn, nn, cn;
				B.qual(r, qt, qn), B.name(r, nt, nn), B.comment(r, ct, cn);
				const int pos = (int)ref_bg;
				if (chr_id < 0 || chr_id >= (int)H->names.size() || pos - 1 < 0) continue;
				if (nn <= 0 || nn > 254 || qn != read_l) { drop("@sam_parse1 ERROR"); continue; }
				const char *svs = sv->print_string(rr.prim_sv_id);
				const char *mvs = rr.has_mate ? sv->print_string(rr.mate_sv_id) : nullptr;
				const psvr_cand_t *sc = rr.secondary >= 0 ? &V.cands[rr.cand_off + rr.secondary] : nullptr;
				const char *vid = sc ? sv->vcf_id(sc->sv_id) : nullptr;
				const std::string &rn = H->names[(size_t)chr_id];
				const bool mate_named = rr.has_mate && rr.mate_chr_id >= 0 && rr.mate_chr_id < (int)H->names.size();
				RawOut o(dst, (size_t)nn + (size_t)cn + 2 * (size_t)read_l + cg.size() + (pcd ? (size_t)pcd->n_cigar * 8 : 0) + rn.size() + (mate_named ? H->names[(size_t)rr.mate_chr_id].size() : 0) +
				                  (svs ? strlen(svs) : 0) + (mvs ? strlen(mvs) : 0) + (vid ? strlen(vid) : 0) + 512);
				o.put(nt, (size_t)nn), o.ch('\t'), o.num(flag), o.ch('\t');
				o.put(rn), o.ch('\t'), o.num(pos), o.ch('\t'), o.num(mapq), o.ch('\t');
				if (pcd && pcd->n_cigar) {
					for (uint32_t j = 0; j < pcd->n_cigar; ++j) { const uint32_t w = V.cig[pcd->cigar_off + j]; o.num((int)(int16_t)(w >> 4)), o.ch("MIDNSHP=XB"[w & 0xf]); }
				} else if (cg.empty()) o.ch('*'); else o.put(cg);
				o.ch('\t');
				if (rr.has_mate) {
					const int mp = (int)rr.mate_ref_bg, mc = rr.mate_chr_id;
					const bool mate_ok = mate_named && !(mp - 1 < 0);
					if (!mate_ok) o.ch('*'); else if (mc == chr_id) o.ch('='); else o.put(H->names[(size_t)mc]);
					o.ch('\t'), o.num(mp);
				} else o.put("*\t0", 3);
				o.ch('\t'), o.num(isize), o.ch('\t');
				put_seq_qual(o, t, qt, read_l, direction == 0);   // SEQ through the reverse complement and htslib's 4-bit code; QUAL through getReverseStr_qual_char
				o.tag_int("AS:i:", (int)align_score), o.tag_int("OS:i:", (int)ori.align_score);
				o.put("\tOA:Z:", 6), o.num(ori.chr_id), o.ch(','), o.num((int)(ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg)), o.ch(',');
				o.num((int)ori.read_bg), o.ch(','), o.num((int)ori.mapq), o.put(rr.unmapped ? ",U;" : ",M;", 3);
				if (!is_ori) o.tag_int("CS:i:", (int)chain_score);
				if (svs) o.put("\tSV:Z:", 6), o.put(svs, strlen(svs));
				if (mvs) o.put("\tMV:Z:", 6), o.put(mvs, strlen(mvs));
				if (sc) {
					o.put("\tXA:Z:", 6), o.num(sc->chr_id), o.ch(','), o.num((int)sc->ref_bg), o.ch(','), o.num((int)sc->read_bg), o.ch(',');
					o.num((int)sc->align_score), o.put(sc->direction == 1 ? ",F," : ",R,", 3);
					if (vid) o.put(vid, strlen(vid)); else o.ch('*');
					o.ch(';');
				}
				{   // RC:Z = the comment as parse_ori_mapping_rst leaves it (see rewrite_comment)
					o.put("\tRC:Z:", 6);
					uint8_t *at = o.p;
					memcpy(at, ct, (size_t)cn);
					int32_t cut[10];
					ori_cuts(ct, cn, cut);
					size_t len = (size_t)cn;
					for (int q = 0; q < 10; ++q) {
						if (cut[q] < 0) continue;
						if (cut[q] < cn - 1) at[(size_t)cut[q]] = ',';
						else if ((size_t)cut[q] < len) len = (size_t)cut[q];
					}
					if (const void *z = memchr(at, 0, len)) len = (size_t)((const uint8_t *)z - at);       // (%s stops at a NUL)
					o.p += len;
				}
				o.ch('\n');
				o.close();
				continue;
			}
			{
				// BAM record straight into the output buffer: the fields of the SAM line above in BAM's binary layout (SAMv1 4.2), i.e. what
				// BamWriter::encode makes of that line -- which stays the path for the records this one declines (a tab inside a string
				// tag's value, a CIGAR operator beyond 'X': the text path decides what becomes of those).  Building six strings per record
				// and parsing them back was 0.45 s per 1 M pairs of the command's BAM route.
				const char *qt, *nt, *ct; int qn, nn, cn;
				B.qual(r, qt, qn), B.name(r, nt, nn), B.comment(r, ct, cn);
				const int pos = (int)ref_bg;
				const char *svs = sv->print_string(rr.prim_sv_id);
				const char *mvs = rr.has_mate ? sv->print_string(rr.mate_sv_id) : nullptr;
				const psvr_cand_t *sc = rr.secondary >= 0 ? &V.cands[rr.cand_off + rr.secondary] : nullptr;
				const char *vid = sc ? sv->vcf_id(sc->sv_id) : nullptr;
				const psvr_cand_t *cd = is_ori ? nullptr : &V.cands[rr.cand_off + rr.primary];
				bool plain = !bam_via_text && chr_id >= 0 && chr_id < (int)H->names.size() && pos - 1 >= 0 && nn > 0 && nn <= 254 && qn == read_l && !memchr(ct, '\t', (size_t)cn) &&
				             !(svs && strchr(svs, '\t')) && !(mvs && strchr(mvs, '\t')) && !(vid && strchr(vid, '\t')) && (!cd || cd->n_cigar <= 0xffff);
				if (plain && cd) for (uint32_t j = 0; j < cd->n_cigar; ++j) if ((V.cig[cd->cigar_off + j] & 0xf) > 8) { plain = false; break; }
				if (plain) {
					const SeqTables &T = seq_tables();
					RawOut o(dst, (size_t)nn + (size_t)cn + 2 * (size_t)read_l + (cd ? (size_t)cd->n_cigar * 4 : 8) + (svs ? strlen(svs) : 0) + (mvs ? strlen(mvs) : 0) + (vid ? strlen(vid) : 0) + 512);
					uint8_t *const rec0 = o.p;
					o.u32(0);                                                   // block_size, patched below
					o.u32((uint32_t)chr_id), o.u32((uint32_t)(pos - 1));
					o.u8((unsigned)nn + 1), o.u8((unsigned)mapq);
					uint8_t *const bin_at = o.p;
					o.u16(0);                                                   // bin, patched when the reference length is known
					const uint32_t ncig = cd ? cd->n_cigar : (ori.read_bg > 0 ? 2u : 1u);
					o.u16(ncig), o.u16((unsigned)flag), o.u32((uint32_t)read_l);
					int mtid = -1;
					long pnext = 0;
					if (rr.has_mate) {
						const int mp = (int)rr.mate_ref_bg, mc = rr.mate_chr_id;
						if (mc >= 0 && mc < (int)H->names.size() && !(mp - 1 < 0)) mtid = mc;
						pnext = mp;
					}
					o.u32((uint32_t)mtid), o.u32((uint32_t)(int32_t)(pnext - 1)), o.u32((uint32_t)isize);
					o.put(nt, (size_t)nn), o.u8(0);
					int64_t rlen = 0;
					auto cig_op = [&](long long len, unsigned op) {            // (lengths are printed as int16 and parsed back: negative ones keep their sign bits)
						o.u32((uint32_t)len << 4 | op);
						if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += (uint32_t)len & 0xfffffff;
					};
					if (cd) for (uint32_t j = 0; j < cd->n_cigar; ++j) { const uint32_t w = V.cig[cd->cigar_off + j]; cig_op((int)(int16_t)(w >> 4), w & 0xf); }
					else {
						if (ori.read_bg > 0) cig_op((int)(int16_t)(uint16_t)ori.read_bg, 4);
						cig_op((int)(int16_t)(uint16_t)(read_l - (int)ori.read_bg), 0);
					}
					{
						const int64_t p0 = pos - 1;
						const int bin = bam_reg2bin(p0, p0 + (rlen > 0 ? rlen : 1));
						bin_at[0] = (uint8_t)bin, bin_at[1] = (uint8_t)(bin >> 8);
					}
					// SEQ: two 4-bit codes per byte; QUAL: phred values (the reverse strand through getReverseStr_char / getReverseStr_qual_char)
					const bool rev = direction == 0;
					int i = 0;
#if defined(__SSE2__)
					{
						// sixteen plain bases (A C G T N) -> eight bytes of 4-bit codes; a block with anything else through the tables
						const Seq16 K;
						const __m128i k1 = _mm_set1_epi8(1), k2 = _mm_set1_epi8(2), k4 = _mm_set1_epi8(4), k8 = _mm_set1_epi8(8), k15 = _mm_set1_epi8(15), lo8 = _mm_set1_epi16(0x00ff);
						for (; i + 16 <= read_l; i += 16) {
							__m128i v = _mm_loadu_si128((const __m128i *)(rev ? t + read_l - 16 - i : t + i));
							if (rev) v = Seq16::reversed(v);
							const __m128i ia = _mm_cmpeq_epi8(v, K.a), ic = _mm_cmpeq_epi8(v, K.c), ig = _mm_cmpeq_epi8(v, K.g), it = _mm_cmpeq_epi8(v, K.t), in = _mm_cmpeq_epi8(v, K.n);
							if (_mm_movemask_epi8(_mm_or_si128(_mm_or_si128(_mm_or_si128(ia, ic), _mm_or_si128(ig, it)), in)) != 0xffff) break;
							const __m128i code = rev ? _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_and_si128(ia, k8), _mm_and_si128(ic, k4)), _mm_or_si128(_mm_and_si128(ig, k2), _mm_and_si128(it, k1))), _mm_and_si128(in, k15))
							                         : _mm_or_si128(_mm_or_si128(_mm_or_si128(_mm_and_si128(ia, k1), _mm_and_si128(ic, k2)), _mm_or_si128(_mm_and_si128(ig, k4), _mm_and_si128(it, k8))), _mm_and_si128(in, k15));
							const __m128i w = _mm_or_si128(_mm_slli_epi16(_mm_and_si128(code, lo8), 4), _mm_srli_epi16(code, 8));
							_mm_storel_epi64((__m128i *)o.p, _mm_packus_epi16(w, w));
							o.p += 8;
						}
					}
#endif
					for (; i < read_l; i += 2) {
						const unsigned hi = rev ? T.rc16[(uint8_t)t[read_l - 1 - i]] : T.c16[(uint8_t)t[i]];
						const unsigned lo = i + 1 < read_l ? (rev ? T.rc16[(uint8_t)t[read_l - 2 - i]] : T.c16[(uint8_t)t[i + 1]]) : 0u;
						o.u8(hi << 4 | lo);
					}
					{
						uint8_t *ql = o.p;
						int j = 0;
#if defined(__SSE2__)
						const __m128i k33 = _mm_set1_epi8(33);
						for (; j + 16 <= read_l; j += 16) {
							const __m128i v = _mm_loadu_si128((const __m128i *)(rev ? qt + read_l - 16 - j : qt + j));
							_mm_storeu_si128((__m128i *)(ql + j), _mm_sub_epi8(rev ? Seq16::reversed(v) : v, k33));
						}
#endif
						if (rev) {
							for (; j < read_l; ++j) ql[j] = (uint8_t)(qt[read_l - 1 - j] - 33);
							if (!(read_l & 1) && read_l >= 2) ql[read_l / 2 - 1] = (uint8_t)(qt[read_l / 2 - 1] - 33), ql[read_l / 2] = (uint8_t)(qt[read_l / 2] - 33);
						} else for (; j < read_l; ++j) ql[j] = (uint8_t)(qt[j] - 33);
						o.p += read_l;
					}
					o.bam_int("AS", (int)align_score), o.bam_int("OS", (int)ori.align_score);
					o.bam_z("OA"), o.num(ori.chr_id), o.ch(','), o.num((int)(ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg)), o.ch(','), o.num((int)ori.read_bg), o.ch(',');
					o.num((int)ori.mapq), o.put(rr.unmapped ? ",U;" : ",M;", 3), o.u8(0);
					if (!is_ori) o.bam_int("CS", (int)chain_score);
					if (svs) o.bam_z("SV"), o.put(svs, strlen(svs)), o.u8(0);
					if (mvs) o.bam_z("MV"), o.put(mvs, strlen(mvs)), o.u8(0);
					if (sc) {
						o.bam_z("XA"), o.num(sc->chr_id), o.ch(','), o.num((int)sc->ref_bg), o.ch(','), o.num((int)sc->read_bg), o.ch(','), o.num((int)sc->align_score);
						o.put(sc->direction == 1 ? ",F," : ",R,", 3);
						if (vid) o.put(vid, strlen(vid)); else o.ch('*');
						o.ch(';'), o.u8(0);
					}
					{   // RC:Z = the comment as parse_ori_mapping_rst leaves it (see rewrite_comment), up to a NUL
						o.bam_z("RC");
						uint8_t *at = o.p;
						memcpy(at, ct, (size_t)cn);
						int32_t cut[10];
						ori_cuts(ct, cn, cut);
						size_t len = (size_t)cn;
						for (int q = 0; q < 10; ++q) {
							if (cut[q] < 0) continue;
							if (cut[q] < cn - 1) at[(size_t)cut[q]] = ',';
							else if ((size_t)cut[q] < len) len = (size_t)cut[q];
						}
						if (const void *z = memchr(at, 0, len)) len = (size_t)((const uint8_t *)z - at);
						o.p += len;
						o.u8(0);
					}
					const uint32_t bs = (uint32_t)(o.p - rec0) - 4;
					rec0[0] = (uint8_t)bs, rec0[1] = (uint8_t)(bs >> 8), rec0[2] = (uint8_t)(bs >> 16), rec0[3] = (uint8_t)(bs >> 24);
					o.close();
					continue;
				}
				if (!cd) {}                                                   // (is_ori: cg holds the text already)
				else cigar_text(*cd, V.cig, cg);
			}
			seq.assign(t, (size_t)n);
			B.qual(r, t, n), qual.assign(t, (size_t)n);
			if (direction == 0) sam_rev_seq(seq), sam_rev_qual(qual);
			B.name(r, t, n), name.assign(t, (size_t)n);
			B.comment(r, t, n), rewrite_comment(t, n, cm);
			tags.clear();
			snprintf(b, sizeof b, "\tAS:i:%d", (int)align_score), tags += b;
			snprintf(b, sizeof b, "\tOS:i:%d\tOA:Z:%d,%d,%d,%d,%c;", (int)ori.align_score, ori.chr_id, (int)(ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg), (int)ori.read_bg, (int)ori.mapq,
			         rr.unmapped ? 'U' : 'M');
			tags += b;
			if (!is_ori) { snprintf(b, sizeof b, "\tCS:i:%d", (int)chain_score); tags += b; }
			const char *svs = sv->print_string(rr.prim_sv_id);
			if (svs) tags += "\tSV:Z:", tags += svs;
			const char *mvs = rr.has_mate ? sv->print_string(rr.mate_sv_id) : nullptr;
			if (mvs) tags += "\tMV:Z:", tags += mvs;
			if (rr.secondary >= 0) {
				const psvr_cand_t &sc = V.cands[rr.cand_off + rr.secondary];
				const char *vid = sv->vcf_id(sc.sv_id);
				snprintf(b, sizeof b, "\tXA:Z:%d,%d,%d,%d,%c,", sc.chr_id, (int)sc.ref_bg, (int)sc.read_bg, (int)sc.align_score, sc.direction == 1 ? 'F' : 'R');
				tags += b;
				tags += vid ? vid : "*";
				tags += ";";
			}
			tags += "\tRC:Z:", tags += cm.c_str();                   // %s: up to a NUL the rewrite may have left
			emit(dst, name, flag, chr_id, ref_bg, mapq, cg, rr.has_mate != 0, rr.mate_chr_id, rr.mate_ref_bg, isize, seq, qual, tags, "@sam_parse1 ERROR");
		}
	}
	// the second file (rr.cpp:776-797): pairs neither the original aligner nor the re-aligner placed well
	void ori_pair(const FastqBatch &B, const ResultView &V, long long p, Bytes &dst) const
	{
		const psvr_pair_result_t &pr = V.pairs[p - V.pair0];
		if (!(pr.max_score <= min_filter_score && B.ori[2 * p].chr_id != -1 && B.ori[2 * p + 1].chr_id != -1)) return;
		OriRecord orr[2];
		for (int k = 0; k < 2; ++k) { const char *t; int n; B.comment(2 * p + k, t, n); if (!parse_ori_record(t, n, &orr[k])) return; }
		bool proper = pr.proper != 0;
		for (int k = 0; proper && k < 2; ++k) {
			const int mx = k == 0 ? pr.max1 : pr.max2;
			if (mx == -1) { proper = false; break; }
			if (mx == -2) { if (ori_has_clip(orr[k].cigar, 25)) proper = false; }
			else {                                               // bam_has_clip_or_unmapped_new (rr.cpp:735-743): sums the 'I' ops
				const psvr_cand_t &cd = V.cands[V.hdr[2 * (p - V.pair0) + k].cand_off + mx];
				int tot = 0;
				for (uint32_t j = 0; j < cd.n_cigar; ++j) { const uint32_t wv = V.cig[cd.cigar_off + j]; if ((wv & 0xf) == 1) tot += (int)(int16_t)(wv >> 4); }
				if (cd.n_cigar == 0 || tot >= 25) proper = false;
			}
		}
		if (proper) return;
		std::string name, seq, qual, tags;
		for (int k = 0; k < 2; ++k) {
			const long long r = 2 * p + k;
			const psvr_ori_t &ori = B.ori[r];
			const char *t; int n;
			if (!as_bam) {
				// SAM text straight into the output buffer (same fields and rules as the BAM path below)
				const char *qt, *nt; int qn, nn;
				B.seq(r, t, n), B.qual(r, qt, qn), B.name(r, nt, nn);
				const uint32_t rb = (ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg) + 1;
				const int pos = (int)rb, chr_id = ori.chr_id;
				if (chr_id < 0 || chr_id >= (int)H->names.size() || pos - 1 < 0) continue;
				if (nn <= 0 || nn > 254 || !acceptable_cigar(orr[k].cigar) || qn != n) { drop("@ori_bam_sam_parse1 ERROR"); continue; }
				const int mp = orr[k].mate_pos, mc = orr[k].mate_chr;
				const bool mate_named = mc >= 0 && mc < (int)H->names.size();
				RawOut o(dst, (size_t)nn + 2 * (size_t)n + orr[k].cigar.size() + orr[k].tags.size() + H->names[(size_t)chr_id].size() + (mate_named ? H->names[(size_t)mc].size() : 0) + 256);
				o.put(nt, (size_t)nn), o.ch('\t'), o.num(orr[k].flag), o.ch('\t'), o.put(H->names[(size_t)chr_id]), o.ch('\t');
				o.num(pos), o.ch('\t'), o.num(orr[k].mapq), o.ch('\t');
				if (orr[k].cigar.empty()) o.ch('*'); else o.put(orr[k].cigar);
				o.ch('\t');
				const bool mate_ok = mate_named && !(mp - 1 < 0);
				if (!mate_ok) o.ch('*'); else if (mc == chr_id) o.ch('='); else o.put(H->names[(size_t)mc]);
				o.ch('\t'), o.num(mp), o.ch('\t'), o.num(orr[k].isize), o.ch('\t');
				put_seq_qual(o, t, qt, n, (orr[k].flag & 0x10) != 0);
				if (!orr[k].tags.empty()) o.ch('\t'), o.put(orr[k].tags);
				o.tag_int("MS:i:", pr.max_score);
				o.ch('\n');
				o.close();
				continue;
			}
			B.seq(r, t, n), seq.assign(t, (size_t)n);
			B.qual(r, t, n), qual.assign(t, (size_t)n);
			B.name(r, t, n), name.assign(t, (size_t)n);
			if (orr[k].flag & 0x10) sam_rev_seq(seq), sam_rev_qual(qual);
			tags.clear();
			if (!orr[k].tags.empty()) tags += "\t" + orr[k].tags;
			char b[64];
			snprintf(b, sizeof b, "\tMS:i:%d", pr.max_score);
			tags += b;
			const uint32_t ref_bg = ori.ref_bg >= 0x7fffffffu ? 1u : ori.ref_bg;
			emit(dst, name, orr[k].flag, ori.chr_id, ref_bg + 1, orr[k].mapq, orr[k].cigar, true, orr[k].mate_chr, (uint32_t)orr[k].mate_pos, orr[k].isize, seq, qual, tags,
			     "@ori_bam_sam_parse1 ERROR");
		}
	}
};

} // namespace psvr
