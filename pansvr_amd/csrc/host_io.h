// host_io.h -- host-side C++ of the `aln` step that sits above the C ABI: the index files and the per-pair decision
// record the parity tests compare (the FASTQ reader is fastq_batch.h, the SAM/BAM records sam_emit.h).
// Mirrors (does not copy) the reference's host behaviour; citations inline.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "aln_device.h"

namespace psvr {

inline void aln_params_default(psvr_aln_params_t *p)   // rr.hpp:34-41 + load_reads defaults rr.cpp:138-143
{
	p->match = 2, p->mismatch = 12, p->gap_open = 16, p->gap_ex = 1, p->gap_open2 = 32, p->gap_ex2 = 0, p->zdrop = 400;
	p->normal_read_length = 150, p->isize_min = 100, p->isize_max = 900;
	p->min_filter_score = 150 * 2 * 2 - 80;
}

template <class T> static bool slurp_file(const std::string &fn, std::vector<T> *out, size_t pad_bytes = 0)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) return false;
	fseek(f, 0, SEEK_END);
	size_t n = ftell(f);
	rewind(f);
	out->assign((n + pad_bytes + sizeof(T) - 1) / sizeof(T), 0);
	bool ok = fread(out->data(), 1, n, f) == n;
	fclose(f);
	return ok;
}

struct SvHost { std::string vcf_print_string, vcf_id; };

// deBGA_INDEX on the host (deBGA_index.cpp:33-80 load_index_file, :354-366 building_chr_index,
// :398-431 building_bam_header)
struct HostIndex {
	std::vector<uint64_t> ref_seq, seq, seqf, pos, posp, hash, off;
	std::vector<uint32_t> kmer;
	std::vector<uint32_t> sp_id; std::vector<uint64_t> sp_start;   // sparse first level (tests/emu only)
	bool keep_sparse = false;
	bool defer_dense = false;                 // GPU engine: leave a sparse first level as (bucket, count) pairs; the dense table is built in HBM
	std::vector<uint32_t> sparse_pairs;
	std::vector<uint32_t> chr_end_n, chr_search_index;
	std::vector<std::string> chr_names;
	std::vector<SvDev> sv;
	std::vector<SvHost> svh;
	int chr_file_n = 0;

	bool parse_chr(const std::string &text, const std::vector<std::string> &header_names, std::string *err)
	{
		// the reference xcalloc()s deBGA_INDEX (rr.cpp:36) so chr_file_n starts at 0, names/ends fill from slot 0
		// and chr_end_n[0] is then overwritten with START_POS_REF + 1 (deBGA_index.cpp:60-72)
		chr_names.clear(), chr_end_n.clear(), chr_file_n = 0;
		size_t p = 0;
		uint32_t line_n = 0;
		while (p < text.size()) {
			while (p < text.size() && strchr(" \t\r\n", text[p])) ++p;
			if (p >= text.size()) break;
			size_t e = p;
			while (e < text.size() && !strchr(" \t\r\n", text[e])) ++e;
			std::string tok = text.substr(p, e - p);
			if ((line_n & 1) == 0) chr_names.push_back(tok);
			else { chr_end_n.push_back((uint32_t)strtoul(tok.c_str(), 0, 10)); chr_file_n++; }
			line_n++;
			p = e;
		}
		if (chr_file_n == 0) { *err = "empty unipath.chr"; return false; }
		chr_end_n[0] = 1;
		chr_end_n.resize(chr_file_n + 1, 0);
		chr_names.resize(chr_file_n + 1);
		chr_names[chr_file_n] = "*";
		uint64_t reference_len = chr_end_n[chr_file_n - 1];
		chr_search_index.assign((reference_len >> 14) + 2, 0);
		uint32_t pis = 0;
		for (int i = 0; i < chr_file_n; i++) {
			int pi = chr_end_n[i] / 0x4000;
			while (pi >= (int)pis) chr_search_index[pis++] = i;
		}
		chr_search_index[pis] = chr_file_n;
		sv.clear(), svh.clear();
		for (int i = 0; i < chr_file_n; i++) {                    // ID_chr_st_len_TYPE_bp1_bp2_end_vcfid, split on '_' (strtok)
			std::vector<std::string> t;
			const std::string &nm = chr_names[i];
			size_t q = 0;
			while (q <= nm.size()) {
				while (q < nm.size() && nm[q] == '_') ++q;
				if (q >= nm.size()) break;
				size_t e = nm.find('_', q);
				if (e == std::string::npos) e = nm.size();
				t.push_back(nm.substr(q, e - q));
				q = e + 1;
			}
			if (t.size() < 9) { *err = "anchor name '" + nm + "' is not ID_chr_st_len_TYPE_bp1_bp2_end_vcfid"; return false; }
			SvDev s;
			int id = atoi(t[0].c_str()), cid = -1;
			for (size_t k = 0; k < header_names.size(); ++k) if (header_names[k] == t[1]) { cid = (int)k; break; }
			s.chr_id = (uint32_t)cid;
			s.st_pos = (uint32_t)atoi(t[2].c_str());
			int region_len = atoi(t[3].c_str());
			uint64_t ed = (uint64_t)(int64_t)atoi(t[7].c_str());
			s.end_offset = (int)(ed - (uint64_t)s.st_pos - region_len);   // SV_chr_info::add_node, deBGA_index.hpp:113
			s.pad = 0;
			sv.push_back(s);
			char b[1200];
			snprintf(b, sizeof b, "%d_%d_%ld_%d_%s_%s", id, cid, (long)s.st_pos, region_len, t[4].c_str(), t[8].c_str());
			svh.push_back(SvHost{b, t[8]});
		}
		return true;
	}

	static bool header_names_of(const std::string &path, std::vector<std::string> *names)
	{
		FILE *h = fopen(path.c_str(), "r");
		if (!h) return false;
		char *buf = nullptr;
		size_t cap = 0;
		while (getline(&buf, &cap, h) > 0) {                   // header lines of any length
			if (strncmp(buf, "@SQ", 3)) continue;
			char *p = strstr(buf, "SN:");
			if (!p) continue;
			p += 3;
			char *e = p;
			while (*e && *e != '\t' && *e != '\n') ++e;
			names->emplace_back(p, e - p);
		}
		free(buf);
		fclose(h);
		return true;
	}

	// index_dir holds the nine deBGA files; the 2 GiB unipath_g.hash may be replaced by the fixture form
	// unipath_g.hash.sparse ((bucket,count) uint32 pairs), which is expanded to the dense prefix-sum table here
	bool load_dir(const std::string &dir_, const std::string &header_sam, std::string *err)
	{
		std::string dir = dir_;
		if (dir.empty()) { *err = "empty index dir"; return false; }
		if (dir.back() != '/') dir += '/';
		std::vector<std::string> names;
		if (!header_names_of(header_sam, &names)) { *err = "cannot read header " + header_sam; return false; }
		if (!slurp_file(dir + "ref.seq", &ref_seq, 536) || !slurp_file(dir + "unipath.seqb", &seq, 16) || !slurp_file(dir + "unipath.seqfb", &seqf) ||
		    !slurp_file(dir + "unipath.pos", &pos) || !slurp_file(dir + "unipath.posp", &posp) || !slurp_file(dir + "unipath_g.kmer", &kmer) ||
		    !slurp_file(dir + "unipath_g.offset", &off)) { *err = "missing index file in " + dir; return false; }
		std::vector<uint32_t> sparse;
		if (keep_sparse && slurp_file(dir + "unipath_g.hash.sparse", &sparse)) {
			uint64_t acc = 0;
			for (size_t i = 0; i + 1 < sparse.size(); i += 2) { sp_id.push_back(sparse[i]); sp_start.push_back(acc); acc += sparse[i + 1]; }
			hash.assign(((size_t)1 << 28) + 1 > 0 ? 0 : 0, 0);
		} else if (defer_dense && slurp_file(dir + "unipath_g.hash.sparse", &sparse)) {
			sparse_pairs.swap(sparse);
		} else if (slurp_file(dir + "unipath_g.hash.sparse", &sparse)) {
			hash.assign(((size_t)1 << 28) + 1, 0);
			for (size_t i = 0; i + 1 < sparse.size(); i += 2) hash[(size_t)sparse[i] + 1] = sparse[i + 1];
			for (size_t i = 1; i < hash.size(); ++i) hash[i] += hash[i - 1];
		} else if (!slurp_file(dir + "unipath_g.hash", &hash)) { *err = "missing unipath_g.hash in " + dir; return false; }
		if (!(keep_sparse && !sp_id.empty()) && sparse_pairs.empty() && hash.size() != ((size_t)1 << 28) + 1) { *err = "unipath_g.hash has the wrong size"; return false; }
		std::vector<char> txt;
		if (!slurp_file(dir + "unipath.chr", &txt)) { *err = "missing unipath.chr"; return false; }
		return parse_chr(std::string(txt.begin(), txt.end()), names, err);
	}

	DevIndex view() const      // pointers into THIS object's memory (host); the GPU engine builds its own from device copies
	{
		DevIndex d = DevIndex();    // everything not set below (occupancy bitmap, bracket table) stays null
		d.ref_seq = ref_seq.data(), d.seq = seq.data(), d.seqf = seqf.data(), d.pos = pos.data(), d.posp = posp.data();
		d.hash = hash.data(), d.off = off.data(), d.kmer = kmer.data(), d.n_seqf = seqf.size();
		d.chr_end_n = chr_end_n.data(), d.chr_search_index = chr_search_index.data(), d.sv = sv.data(), d.chr_file_n = chr_file_n;
		d.sp_id = sp_id.data(), d.sp_start = sp_start.data(), d.sp_n = sp_id.size(), d.n_kmer = kmer.size();
		return d;
	}
};

// the record line of the oracle / reference harness (tests compare these verbatim)
inline std::string record_json(long long pair_i, const psvr_read_result_t *rr, const psvr_pair_result_t &pr, const psvr_ori_t *ori,
                               const int *read_len, const uint32_t *cig, bool trace)
{
	std::string o;
	char b[512];
	snprintf(b, sizeof b, "{\"i\":%lld,\"reads\":[", pair_i);
	o += b;
	for (int k = 0; k < 2; ++k) {
		const psvr_read_result_t &r = rr[k];
		snprintf(b, sizeof b, "%s{\"n\":%d,\"unmapped\":%d,\"res\":[", k ? "," : "", r.n_result, (int)r.unmapped);
		o += b;
		for (int i = 0; i < r.n_result; ++i) {
			const psvr_cand_t &c = r.cand[i];
			snprintf(b, sizeof b, "%s[%u,%u,%d,%u,%u,%d,%d,\"", i ? "," : "", c.align_score, c.chain_score, c.chr_id, c.ref_bg, c.read_bg, (int)c.direction, (int)c.mapq);
			o += b;
			for (uint32_t j = 0; j < c.n_cigar; ++j) { uint32_t w = cig[c.cigar_off + j]; snprintf(b, sizeof b, "%d%c", (int)(int16_t)(w >> 4), "MIDNSHP=XB"[w & 0xf]); o += b; }
			o += "\"]";
		}
		uint32_t ref_bg = ori[k].ref_bg >= 0x7fffffffu ? 1u : ori[k].ref_bg;
		snprintf(b, sizeof b, "],\"ori\":[%u,0,%d,%u,%u,%d,%d,\"", ori[k].align_score, ori[k].chr_id, ref_bg, ori[k].read_bg, (int)ori[k].direction, (int)ori[k].mapq);
		o += b;
		if (ori[k].read_bg > 0) { snprintf(b, sizeof b, "%dS", (int)(int16_t)(uint16_t)ori[k].read_bg); o += b; }
		snprintf(b, sizeof b, "%dM\"]", (int)(int16_t)(uint16_t)(read_len[k] - (int)ori[k].read_bg));
		o += b;
		if (pr.gain) {
			snprintf(b, sizeof b, ",\"prim\":%d,\"sec\":%d", r.primary, r.secondary);
			o += b;
			if (r.primary != -1) { snprintf(b, sizeof b, ",\"mate\":[%d,%u,%u]", r.has_mate, r.has_mate ? (uint32_t)r.mate_chr_id : 0u, r.has_mate ? r.mate_ref_bg : 0u); o += b; }
		}
		if (trace) {
			snprintf(b, sizeof b, ",\"str\":%d,\"tr\":[[%u,\"%016llx\",\"%016llx\"],[%u,\"%016llx\",\"%016llx\"]]", (int)r.is_str, r.n_seed[0],
			         (unsigned long long)r.seed_hash[0], (unsigned long long)r.chain_hash[0], r.n_seed[1], (unsigned long long)r.seed_hash[1], (unsigned long long)r.chain_hash[1]);
			o += b;
		}
		o += "}";
	}
	snprintf(b, sizeof b, "],\"pe\":[%d,%d,%d,%d,%d,%d]}", pr.max_score, pr.cur_isize, pr.proper, pr.gain, pr.max1, pr.max2);
	o += b;
	return o;
}

} // namespace psvr
