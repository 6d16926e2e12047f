// bam_reader.h -- sequential BAM reader for the `signal` step (SAM/BAM specification v1, sections 4.1-4.2): BGZF members
// inflated with zlib, the header, then one alignment record after the other.  Host C++ only; it replaces the reference's use of
// htslib (sam_read1 / bam_aux_get / bam_aux2i) for the fields that step looks at.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace psvr {

class BgzfReader {
	FILE *f = nullptr;
	std::vector<uint8_t> in, out;
	size_t pos = 0;                       // read position in `out`
	bool eof = false;
	std::string err;
	bool next_block()
	{
		uint8_t h[18];
		size_t got = fread(h, 1, 18, f);
		if (got == 0) { eof = true; return false; }
		if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block"; eof = true; return false; }
		const unsigned xlen = h[10] | (unsigned)h[11] << 8;
		// the BC subfield is the first one in every BGZF writer's output; walk the extra field to be safe
		std::vector<uint8_t> extra(xlen);
		memcpy(extra.data(), h + 12, xlen < 6 ? xlen : 6);
		if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, f) != xlen - 6) { err = "truncated BGZF block"; eof = true; return false; }
		unsigned bsize = 0;
		for (unsigned o = 0; o + 4 <= xlen;) {
			const unsigned slen = extra[o + 2] | (unsigned)extra[o + 3] << 8;
			if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2 && o + 6 <= xlen) bsize = (extra[o + 4] | (unsigned)extra[o + 5] << 8) + 1;
			o += 4 + slen;
		}
		if (bsize < 12 + xlen + 8) { err = "BGZF block without a BC field"; eof = true; return false; }
		const size_t clen = bsize - 12 - xlen - 8;
		in.resize(clen + 8);
		if (fread(in.data(), 1, clen + 8, f) != clen + 8) { err = "truncated BGZF block"; eof = true; return false; }
		const uint32_t isize = in[clen + 4] | (uint32_t)in[clen + 5] << 8 | (uint32_t)in[clen + 6] << 16 | (uint32_t)in[clen + 7] << 24;
		out.resize(isize);
		pos = 0;
		if (isize == 0) return true;      // the end-of-file marker block (or an empty one)
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		if (inflateInit2(&zs, -15) != Z_OK) { err = "zlib"; eof = true; return false; }
		zs.next_in = in.data(), zs.avail_in = (uInt)clen, zs.next_out = out.data(), zs.avail_out = isize;
		const int rc = inflate(&zs, Z_FINISH);
		inflateEnd(&zs);
		if (rc != Z_STREAM_END || zs.total_out != isize) { err = "corrupt BGZF block"; eof = true; return false; }
		return true;
	}

public:
	bool open(const char *path) { f = !strcmp(path, "-") ? stdin : fopen(path, "rb"); return f != nullptr; }
	void close() { if (f && f != stdin) fclose(f); f = nullptr; }
	~BgzfReader() { close(); }
	const std::string &error() const { return err; }
	// exactly n bytes, or false at the end of the file (err is set when the end comes inside a request)
	bool read(void *dst, size_t n)
	{
		uint8_t *d = (uint8_t *)dst;
		size_t done = 0;
		while (done < n) {
			if (pos == out.size()) {
				if (eof || !next_block()) { if (done) err = "truncated BAM stream"; return false; }
				continue;
			}
			const size_t k = out.size() - pos < n - done ? out.size() - pos : n - done;
			memcpy(d + done, out.data() + pos, k);
			pos += k, done += k;
		}
		return true;
	}
};

struct BamRecord {                     // one alignment, fields as in bam1_core_t
	int32_t tid = -1, pos = -1, mtid = -1, mpos = -1, isize = 0, l_qseq = 0;
	uint16_t flag = 0, n_cigar = 0;
	uint8_t mapq = 0, l_qname = 0;
	std::vector<uint8_t> data;         // qname | cigar | seq (4-bit) | qual | aux
	const char *qname() const { return (const char *)data.data(); }
	uint32_t cig(unsigned k) const { uint32_t v; memcpy(&v, data.data() + l_qname + 4 * (size_t)k, 4); return v; }   // the CIGAR words are not aligned (the name's length decides)
	const uint8_t *seq() const { return data.data() + l_qname + 4 * (size_t)n_cigar; }
	const uint8_t *qual() const { return seq() + (l_qseq + 1) / 2; }
	const uint8_t *aux() const { return qual() + l_qseq; }
	const uint8_t *aux_end() const { return data.data() + data.size(); }
	// bam_aux_get: pointer to the type byte of the tag, or null.  A tag is only returned when its whole value lies inside the
	// record (Z/H: including the terminating NUL), so num_tag()/string_tag() and their callers never read past the end of a
	// malformed record.
	const uint8_t *aux_get(const char tag[2]) const
	{
		const uint8_t *p = aux(), *e = aux_end();
		while (p + 3 <= e) {
			const bool hit = p[0] == (uint8_t)tag[0] && p[1] == (uint8_t)tag[1];
			const uint8_t *v = p + 2;
			const char t = (char)v[0];
			const uint8_t *val = v + 1;
			size_t sz = 0;
			switch (t) {
			case 'A': case 'c': case 'C': sz = 1; break;
			case 's': case 'S': sz = 2; break;
			case 'i': case 'I': case 'f': sz = 4; break;
			case 'd': sz = 8; break;
			case 'Z': case 'H': {
				const uint8_t *q = val;
				while (q < e && *q) ++q;
				if (q >= e) return nullptr;                      // no terminator inside the record
				sz = (size_t)(q - val) + 1;
				break;
			}
			case 'B': {
				if (val + 5 > e) return nullptr;
				const char st = (char)val[0];
				const uint32_t cnt = val[1] | (uint32_t)val[2] << 8 | (uint32_t)val[3] << 16 | (uint32_t)val[4] << 24;
				const size_t es = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
				sz = 5 + es * (size_t)cnt;
				break;
			}
			default: return nullptr;
			}
			if (sz > (size_t)(e - val)) return nullptr;          // the value runs past the record
			if (hit) return v;
			p = val + sz;
		}
		return nullptr;
	}
	// bam_get_string_tag (clib/bam_file.c:427-438): only 'Z'
	const char *string_tag(const char tag[2]) const
	{
		const uint8_t *p = aux_get(tag);
		return p && p[0] == 'Z' ? (const char *)(p + 1) : nullptr;
	}
	// bam_get_num_tag (clib/bam_file.c:456-468): integer codes only
	bool num_tag(const char tag[2], int32_t *num) const
	{
		const uint8_t *p = aux_get(tag);
		if (!p) return false;
		switch ((char)p[0]) {
		case 'c': *num = (int8_t)p[1]; return true;
		case 'C': *num = p[1]; return true;
		case 's': *num = (int16_t)(p[1] | (uint16_t)p[2] << 8); return true;
		case 'S': *num = (uint16_t)(p[1] | (uint16_t)p[2] << 8); return true;
		case 'i': case 'I': *num = (int32_t)(p[1] | (uint32_t)p[2] << 8 | (uint32_t)p[3] << 16 | (uint32_t)p[4] << 24); return true;
		default: return false;
		}
	}
};

class BamReader {
	BgzfReader z;
	std::string err;

public:
	std::string header_text;
	std::vector<std::pair<std::string, int32_t>> refs;
	const std::string &error() const { return err.empty() ? z.error() : err; }
	bool open(const char *path)
	{
		if (!z.open(path)) { err = std::string("cannot open ") + path; return false; }
		char magic[4];
		int32_t l_text = 0, n_ref = 0;
		if (!z.read(magic, 4) || memcmp(magic, "BAM\1", 4)) { err = "not a BAM file"; return false; }
		if (!z.read(&l_text, 4) || l_text < 0) { err = "bad BAM header"; return false; }
		header_text.resize((size_t)l_text);
		if (l_text && !z.read(&header_text[0], (size_t)l_text)) { err = "bad BAM header"; return false; }
		while (!header_text.empty() && header_text.back() == '\0') header_text.pop_back();
		if (!z.read(&n_ref, 4) || n_ref < 0) { err = "bad BAM header"; return false; }
		for (int i = 0; i < n_ref; ++i) {
			int32_t l_name = 0, l_ref = 0;
			if (!z.read(&l_name, 4) || l_name <= 0) { err = "bad BAM header"; return false; }
			std::string name((size_t)l_name, '\0');
			if (!z.read(&name[0], (size_t)l_name) || !z.read(&l_ref, 4)) { err = "bad BAM header"; return false; }
			name.resize(strlen(name.c_str()));
			refs.push_back({name, l_ref});
		}
		return true;
	}
	// sam_read1: false at the end of the file (error() is empty then) or on a malformed record
	bool next(BamRecord &r)
	{
		int32_t block = 0;
		if (!z.read(&block, 4)) return false;
		if (block < 32) { err = "bad BAM record"; return false; }
		uint8_t c[32];
		if (!z.read(c, 32)) { err = "truncated BAM record"; return false; }
		auto i32 = [&](int o) { return (int32_t)(c[o] | (uint32_t)c[o + 1] << 8 | (uint32_t)c[o + 2] << 16 | (uint32_t)c[o + 3] << 24); };
		r.tid = i32(0), r.pos = i32(4);
		r.l_qname = c[8], r.mapq = c[9];
		r.n_cigar = (uint16_t)(c[12] | (uint16_t)c[13] << 8), r.flag = (uint16_t)(c[14] | (uint16_t)c[15] << 8);
		r.l_qseq = i32(16), r.mtid = i32(20), r.mpos = i32(24), r.isize = i32(28);
		r.data.resize((size_t)block - 32);
		if (block > 32 && !z.read(r.data.data(), (size_t)block - 32)) { err = "truncated BAM record"; return false; }
		if (r.l_qseq < 0 || (size_t)r.l_qname + 4 * (size_t)r.n_cigar + (size_t)((size_t)r.l_qseq + 1) / 2 + (size_t)r.l_qseq > r.data.size()) { err = "bad BAM record"; return false; }
		if (r.l_qname == 0 || r.data[(size_t)r.l_qname - 1] != 0) { err = "bad BAM record (read name not NUL-terminated)"; return false; }   // qname() is used as a C string
		return true;
	}
};

} // namespace psvr
