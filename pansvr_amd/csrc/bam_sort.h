// bam_sort.h -- `panSVR sort`: what panSVR_run.sh does after the `aln` step with `samtools sort` + `samtools index`
// (panSVR_run.sh:53-54), so the drop-in does not depend on an external binary.  Host C++ only.
//   panSVR sort [-n] [-t threads] [-o out.bam] in.bam      coordinate order (default) + out.bam.bai, or name order (-n)
// Coordinate order is samtools' (bam_sort.c bam1_lt): reference id as unsigned (unplaced records last), position, forward strand
// before reverse, ties in input order; name order compares the names with strcmp, first read before second.  The whole file is held
// in memory (the aln step's output is the signal subset of a run, not the full BAM).  The .bai follows SAMv1 section 5.2: bins with
// their chunk lists (virtual file offsets), the 16 kbp linear index, the per-reference metadata pseudo-bin 37450 and n_no_coor.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "bam_reader.h"
#include "bam_writer.h"

namespace psvr {

inline int bam_sort_main(int argc, char **argv)
{
	bool by_name = false;
	int threads = 4;
	std::string out_fn, in_fn;
	for (int i = 2; i < argc; ++i) {
		if (!strcmp(argv[i], "-n")) by_name = true;
		else if ((!strcmp(argv[i], "-t") || !strcmp(argv[i], "-@")) && i + 1 < argc) threads = atoi(argv[++i]);
		else if (!strcmp(argv[i], "-o") && i + 1 < argc) out_fn = argv[++i];
		else in_fn = argv[i];
	}
	if (in_fn.empty()) { fprintf(stderr, "usage: panSVR sort [-n] [-t threads] [-o out.bam] in.bam\n"); return 1; }
	if (out_fn.empty()) out_fn = in_fn + (by_name ? ".nsorted.bam" : ".sorted.bam");
	if (threads < 1) threads = 1;
	BamReader rd;
	if (!rd.open(in_fn.c_str())) { fprintf(stderr, "[panSVR-amd] sort: %s\n", rd.error().c_str()); return 2; }
	// every record as it stands in the file (block_size + body), one buffer
	std::vector<uint8_t> blob;
	struct Rec { uint64_t off; uint32_t len; uint32_t tid; int32_t pos; uint16_t flag; uint8_t l_qname; };
	std::vector<Rec> recs;
	BamRecord r;
	while (rd.next(r)) {
		Rec x;
		x.off = blob.size(), x.len = (uint32_t)(36 + r.data.size()), x.tid = (uint32_t)r.tid, x.pos = r.pos, x.flag = r.flag, x.l_qname = r.l_qname;
		uint8_t h[36];
		auto p32 = [&](int o, uint32_t v) { for (int k = 0; k < 4; ++k) h[o + k] = (uint8_t)(v >> (8 * k)); };
		// the fixed part is re-encoded from the parsed fields (bin recomputed below would be identical: it is kept from the CIGAR's span)
		p32(0, (uint32_t)(32 + r.data.size())), p32(4, (uint32_t)r.tid), p32(8, (uint32_t)r.pos);
		h[12] = r.l_qname, h[13] = r.mapq;
		int64_t rlen = 0;
		for (unsigned k = 0; k < r.n_cigar; ++k) { const uint32_t c = r.cig(k); const int op = (int)(c & 0xf); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4; }
		const int64_t beg = r.pos < 0 ? 0 : r.pos, end = beg + (rlen > 0 ? rlen : 1);
		int bin;
		{
			int64_t e = end - 1;
			if (beg >> 14 == e >> 14) bin = (int)(((1 << 15) - 1) / 7 + (beg >> 14));
			else if (beg >> 17 == e >> 17) bin = (int)(((1 << 12) - 1) / 7 + (beg >> 17));
			else if (beg >> 20 == e >> 20) bin = (int)(((1 << 9) - 1) / 7 + (beg >> 20));
			else if (beg >> 23 == e >> 23) bin = (int)(((1 << 6) - 1) / 7 + (beg >> 23));
			else if (beg >> 26 == e >> 26) bin = (int)(((1 << 3) - 1) / 7 + (beg >> 26));
			else bin = 0;
		}
		h[14] = (uint8_t)bin, h[15] = (uint8_t)(bin >> 8);
		h[16] = (uint8_t)r.n_cigar, h[17] = (uint8_t)(r.n_cigar >> 8), h[18] = (uint8_t)r.flag, h[19] = (uint8_t)(r.flag >> 8);
		p32(20, (uint32_t)r.l_qseq), p32(24, (uint32_t)r.mtid), p32(28, (uint32_t)r.mpos), p32(32, (uint32_t)r.isize);
		blob.insert(blob.end(), h, h + 36);
		blob.insert(blob.end(), r.data.begin(), r.data.end());
		recs.push_back(x);
	}
	if (!rd.error().empty()) { fprintf(stderr, "[panSVR-amd] sort: %s\n", rd.error().c_str()); return 2; }
	std::vector<uint32_t> ord(recs.size());
	for (size_t i = 0; i < ord.size(); ++i) ord[i] = (uint32_t)i;
	if (by_name)
		std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
			const int c = strcmp((const char *)&blob[recs[a].off + 36], (const char *)&blob[recs[b].off + 36]);
			if (c) return c < 0;
			return (recs[a].flag & 0xc0) < (recs[b].flag & 0xc0);
		});
	else
		std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
			const Rec &x = recs[a], &y = recs[b];
			if (x.tid != y.tid) return x.tid < y.tid;
			if (x.pos != y.pos) return x.pos < y.pos;
			return (x.flag & 0x10) < (y.flag & 0x10);
		});
	// header with the sort order stated, as samtools rewrites it
	std::string text = rd.header_text;
	{
		const std::string so = by_name ? "queryname" : "coordinate";
		if (text.compare(0, 3, "@HD") == 0) {
			const size_t eol = text.find('\n');
			std::string hd = text.substr(0, eol);
			const size_t p = hd.find("\tSO:");
			if (p != std::string::npos) { size_t e = hd.find('\t', p + 1); hd.erase(p, (e == std::string::npos ? hd.size() : e) - p); }
			hd += "\tSO:" + so;
			text = hd + text.substr(eol == std::string::npos ? text.size() : eol);
		} else text = "@HD\tVN:1.6\tSO:" + so + "\n" + text;
	}
	std::vector<uint8_t> stream = {'B', 'A', 'M', 1};
	auto put32 = [&](uint32_t v) { for (int k = 0; k < 4; ++k) stream.push_back((uint8_t)(v >> (8 * k))); };
	put32((uint32_t)text.size());
	stream.insert(stream.end(), text.begin(), text.end());
	put32((uint32_t)rd.refs.size());
	for (auto &rf : rd.refs) { put32((uint32_t)rf.first.size() + 1); stream.insert(stream.end(), rf.first.begin(), rf.first.end()); stream.push_back(0); put32((uint32_t)rf.second); }
	std::vector<uint64_t> ustart(ord.size() + 1);
	for (size_t i = 0; i < ord.size(); ++i) {
		ustart[i] = stream.size();
		const Rec &x = recs[ord[i]];
		stream.insert(stream.end(), blob.begin() + (long)x.off, blob.begin() + (long)(x.off + x.len));
	}
	ustart[ord.size()] = stream.size();
	std::vector<uint8_t>().swap(blob);
	// BGZF blocks of 0xff00 uncompressed bytes, compressed on `threads` threads; their file offsets give the virtual offsets
	const size_t kBlock = 0xff00, nb = (stream.size() + kBlock - 1) / kBlock;
	std::vector<std::vector<uint8_t>> comp(nb);
	{
		std::atomic<size_t> next(0);
		auto work = [&]() {
			std::vector<uint8_t> tmp(0x10000 + 64);
			for (size_t b = next++; b < nb; b = next++) {
				const size_t o = b * kBlock, m = stream.size() - o < kBlock ? stream.size() - o : kBlock;
				const size_t n = BgzfWriter::compress_block_public(stream.data() + o, m, tmp.data());
				comp[b].assign(tmp.begin(), tmp.begin() + (long)n);
			}
		};
		std::vector<std::thread> th;
		for (int t = 1; t < threads; ++t) th.emplace_back(work);
		work();
		for (auto &t : th) t.join();
	}
	std::vector<uint64_t> cstart(nb + 1, 0);
	for (size_t b = 0; b < nb; ++b) { if (comp[b].empty()) { fprintf(stderr, "[panSVR-amd] sort: compression failed\n"); return 2; } cstart[b + 1] = cstart[b] + comp[b].size(); }
	FILE *fo = fopen(out_fn.c_str(), "wb");
	if (!fo) { fprintf(stderr, "fail to open file '%s'\n", out_fn.c_str()); return 2; }
	for (size_t b = 0; b < nb; ++b) fwrite(comp[b].data(), 1, comp[b].size(), fo);
	static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	fwrite(eof, 1, 28, fo);
	if (fclose(fo) != 0) { fprintf(stderr, "fail to write file '%s'\n", out_fn.c_str()); return 2; }
	fprintf(stderr, "[panSVR-amd] sort: %zu records -> %s (%s order)\n", ord.size(), out_fn.c_str(), by_name ? "name" : "coordinate");
	if (by_name) return 0;
	// ---- .bai
	auto voff = [&](uint64_t u) { const size_t b = (size_t)(u / kBlock); return b < nb ? (cstart[b] << 16) | (u % kBlock) : (cstart[nb] << 16); };   // (the end of the data = the EOF block)
	struct RefIdx { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin; uint64_t beg = ~0ull, end = 0, n_mapped = 0, n_unmapped = 0; };
	std::vector<RefIdx> ri(rd.refs.size());
	uint64_t n_no_coor = 0;
	for (size_t i = 0; i < ord.size(); ++i) {
		const Rec &x = recs[ord[i]];
		if ((int32_t)x.tid < 0 || x.tid >= ri.size()) { ++n_no_coor; continue; }
		const uint8_t *h = &stream[ustart[i]];
		const uint32_t bin = h[14] | (uint32_t)h[15] << 8;
		const uint64_t vb = voff(ustart[i]), ve = voff(ustart[i + 1]);
		RefIdx &R = ri[x.tid];
		auto &ch = R.bins[bin];
		if (!ch.empty() && ch.back().second == vb) ch.back().second = ve;       // adjacent records of a bin share a chunk
		else ch.push_back({vb, ve});
		// reference span from the CIGAR (1 base without one), for the linear index
		const uint32_t l_qname = h[12], n_cig = h[16] | (uint32_t)h[17] << 8;
		int64_t rlen = 0;
		for (uint32_t k = 0; k < n_cig; ++k) { uint32_t c; memcpy(&c, h + 36 + l_qname + 4 * k, 4); const int op = (int)(c & 0xf); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4; }
		const int64_t beg = x.pos < 0 ? 0 : x.pos, end = beg + (rlen > 0 ? rlen : 1);
		for (int64_t w = beg >> 14; w <= (end - 1) >> 14; ++w) {
			if ((size_t)w >= R.lin.size()) R.lin.resize((size_t)w + 1, 0);
			if (R.lin[(size_t)w] == 0) R.lin[(size_t)w] = vb;
		}
		if (vb < R.beg) R.beg = vb;
		if (ve > R.end) R.end = ve;
		if (x.flag & 0x4) ++R.n_unmapped; else ++R.n_mapped;
	}
	std::vector<uint8_t> bai = {'B', 'A', 'I', 1};
	auto b32 = [&](uint32_t v) { for (int k = 0; k < 4; ++k) bai.push_back((uint8_t)(v >> (8 * k))); };
	auto b64 = [&](uint64_t v) { for (int k = 0; k < 8; ++k) bai.push_back((uint8_t)(v >> (8 * k))); };
	b32((uint32_t)ri.size());
	for (RefIdx &R : ri) {
		const bool any = !R.bins.empty();
		b32((uint32_t)R.bins.size() + (any ? 1 : 0));
		for (auto &kv : R.bins) { b32(kv.first); b32((uint32_t)kv.second.size()); for (auto &c : kv.second) b64(c.first), b64(c.second); }
		if (any) { b32(37450), b32(2), b64(R.beg), b64(R.end), b64(R.n_mapped), b64(R.n_unmapped); }
		for (size_t w = 1; w < R.lin.size(); ++w) if (R.lin[w] == 0) R.lin[w] = R.lin[w - 1];          // empty windows point at the previous one, as samtools fills them
		b32((uint32_t)R.lin.size());
		for (uint64_t v : R.lin) b64(v);
	}
	b64(n_no_coor);
	FILE *fi = fopen((out_fn + ".bai").c_str(), "wb");
	if (!fi || fwrite(bai.data(), 1, bai.size(), fi) != bai.size() || fclose(fi) != 0) { fprintf(stderr, "fail to write file '%s.bai'\n", out_fn.c_str()); return 2; }
	return 0;
}

} // namespace psvr
