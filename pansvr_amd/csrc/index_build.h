// index_build.h -- anchor FASTA -> the deBGA index files `panSVR aln` loads.
//
// Replaces `deBGA index -k 22 <anchors.fa> <dir>` (reference: deBGA_release/src/index_build.c -- load_reffile_kmer_fa
// :411-909, file_kmer_qsort :1013-1860, build_pos_unipath :1862-2390).  Host C++, no GPU: the index is built once per
// anchor set.  The arrays must come out identical to the reference builder's, because unipath numbering and k-mer offsets
// feed the aligner's sort orders; tests/test_index_build.py compares every array with the committed reference-built
// fixtures (tests/golden/*/idx), repeats included.
//
// What the files hold (k_t = 22, first level k = 14 bases):
//   ref.seq            2-bit reference, 32 bases per uint64, MSB first
//   unipath.chr        name / cumulative end + 1, alternating lines
//   unipath_g.hash     uint64[4^14 + 1]: index of the first distinct 22-mer whose first 14 bases are >= the slot
//   unipath_g.kmer     uint32 per distinct 22-mer (sorted): its last 8 bases
//   unipath_g.offset   uint64 per distinct 22-mer: where it starts in the concatenated unipath sequence
//   unipath.seqb       the unipath sequences, concatenated, 2-bit packed like ref.seq
//   unipath.seqfb      uint64[U + 1]: unipath boundaries in that sequence
//   unipath.pos/.posp  for every unipath the ascending 1-based reference positions where the WHOLE unipath occurs
// Unipaths are the non-branching paths of the de Bruijn graph of the 22-mers, cut and numbered by the reference's walk:
// distinct 22-mers are visited in sorted order; a node that is not "one in, one out" starts / ends / is a path of its own
// depending on its in- and out-degree, and its successors are followed in A,C,G,T order (the walk in build()).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

namespace psvr {

struct BuiltIndex {
	std::vector<uint64_t> ref_seq, seqb, seqf, pos, posp, off;
	std::vector<uint32_t> kmer;
	std::vector<uint64_t> hash_sparse_id, hash_sparse_cnt;   // non-empty first-level buckets (the dense table is their prefix sum)
	std::string chr_text;
	uint64_t n_kmer = 0;
};

class IndexBuilder {
	static const int KT = 22, K1 = 14;
	struct Occ { uint64_t kmer, pos1; uint8_t in, out; };
	std::vector<uint64_t> kv_, point_, pos_arr_;     // distinct 22-mers (sorted), their slices of pos_arr_
	std::vector<uint8_t> edge_, eflag_;
	std::string err_;

	// the reference's character folding (bit_operation.c charTochar): IUPAC codes containing G -> G, Y -> T, N kept, the rest -> A;
	// lower-case a c g t n keep their case
	static char fold(unsigned char c)
	{
		switch (c) {
		case 'C': case 'G': case 'T': case 'N': case 'a': case 'c': case 'g': case 't': case 'n': return (char)c;
		case 'B': case 'D': case 'K': case 'R': case 'S': case 'V': case 'b': case 'd': case 'k': case 'r': case 's': case 'v': return 'G';
		case 'Y': case 'y': return 'T';
		default: return 'A';
		}
	}
	static int code5(char c)      // charToDna5: A C G T -> 0..3 (either case), everything else 4
	{
		switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
	}
	static int popc4(unsigned x) { return (int)((x & 1) + ((x >> 1) & 1) + ((x >> 2) & 1) + ((x >> 3) & 1)); }
	// node_indentity (index_build.c:2541-2563)
	int node_type(size_t i) const
	{
		if (eflag_[i]) return 2;
		const int in = popc4(edge_[i] >> 4), out = popc4(edge_[i] & 0xf);
		if (in == 1 && out == 1) return 1;
		if (in == 1 && out > 1) return 2;
		if (in > 1 && out == 1) return 3;
		if (in > 1 && out > 1) return 4;
		if (in == 0 && out == 1) return 5;
		if (in == 0 && out > 1) return 6;
		if (in == 1 && out == 0) return 7;
		if (in > 1 && out == 0) return 8;
		return 0;                                         // no edge at all: the reference leaves such a node untouched
	}
	std::vector<uint32_t> top_;                       // first distinct 22-mer of every 12-base prefix: find() bisects inside one slot
	static const int TOPB = 24;
	void build_top()
	{
		top_.assign(((size_t)1 << TOPB) + 1, 0);
		for (uint64_t k : kv_) top_[(size_t)(k >> (2 * KT - TOPB)) + 1]++;
		for (size_t i = 1; i < top_.size(); ++i) top_[i] += top_[i - 1];
	}
	long long find(uint64_t k) const
	{
		const size_t h = (size_t)(k >> (2 * KT - TOPB));
		auto b = kv_.begin() + top_[h], e = kv_.begin() + top_[h + 1];
		auto it = std::lower_bound(b, e, k);
		return it != e && *it == k ? (long long)(it - kv_.begin()) : -1;
	}
	static uint64_t next_kmer(uint64_t k, int e) { return ((k << 2) | (uint64_t)e) & ((1ull << (2 * KT)) - 1); }
	static void push_bases(std::vector<uint8_t> &s, uint64_t k, int n) { for (int i = n - 1; i >= 0; --i) s.push_back((uint8_t)((k >> (2 * i)) & 3)); }
	static void pack(const std::vector<uint8_t> &codes, std::vector<uint64_t> &w)
	{
		w.assign(codes.size() / 32 + 1, 0);
		for (size_t i = 0; i < codes.size(); ++i) w[i >> 5] |= (uint64_t)codes[i] << ((31 - (i & 31)) << 1);
	}
public:
	const std::string &error() const { return err_; }

	bool build(const char *fasta, BuiltIndex *out)
	{
		FILE *f = fopen(fasta, "r");
		if (!f) { err_ = std::string("cannot open ") + fasta; return false; }
		// ---- sequences (load_reffile_kmer_fa): names = first token of the header, characters folded
		std::vector<std::string> names;
		std::vector<std::string> seqs;
		{
			char *line = nullptr;
			size_t cap = 0;
			ssize_t n;
			while ((n = getline(&line, &cap, f)) > 0) {
				while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
				if (line[0] == '>') {
					std::string nm(line + 1);
					size_t sp = nm.find(' ');
					if (sp != std::string::npos) nm.resize(sp);
					names.push_back(nm), seqs.emplace_back();
				} else if (!seqs.empty()) {
					for (ssize_t i = 0; i < n; ++i) seqs.back().push_back(fold((unsigned char)line[i]));
				}
			}
			free(line);
			fclose(f);
		}
		if (seqs.empty()) { err_ = "no sequence in the FASTA"; return false; }
		// reference words, chromosome table, and one record per 22-mer without N: (k-mer, 1-based global position, the base in
		// front of it, the base behind it).  The base in front is forgotten after a k-mer with an N (:676-682) and at a sequence
		// start; the base behind the last k-mer of a sequence is 'N' (:543).
		std::vector<uint8_t> ref_codes;
		std::vector<Occ> occ;
		uint64_t gpos = 0;
		out->chr_text.clear();
		for (size_t s = 0; s < seqs.size(); ++s) {
			const std::string &q = seqs[s];
			for (char c : q) { int v = code5(c); ref_codes.push_back((uint8_t)(v > 3 ? 2 : v)); }   // charToDna5_N2: an N is packed as 2 (G)
			const long long L = (long long)q.size();
			int in = 4;
			uint64_t k = 0;
			long long last_n = -1;                      // position of the latest N seen
			for (long long j = 0; j < KT - 1 && j < L; ++j) { int v = code5(q[j]); if (v > 3) last_n = j; k = (k << 2) | (uint64_t)(v & 3); }
			for (long long i = 0; i + KT <= L; ++i) {
				{ int v = code5(q[i + KT - 1]); if (v > 3) last_n = i + KT - 1; k = ((k << 2) | (uint64_t)(v & 3)) & ((1ull << (2 * KT)) - 1); }
				if (last_n >= i) { in = 4; continue; }
				const int o = i + KT < L ? code5(q[i + KT]) : 4;
				occ.push_back(Occ{k, gpos + (uint64_t)i + 1, (uint8_t)in, (uint8_t)o});
				in = code5(q[i]);
			}
			gpos += (uint64_t)(L < KT ? KT : L);       // a sequence shorter than a k-mer still advances the coordinate by k_t (:538, pos += k_t)
			char b[64];
			snprintf(b, sizeof b, "\n%llu\n", (unsigned long long)(gpos + 1));
			out->chr_text += names[s] + b;
		}
		pack(ref_codes, out->ref_seq);
		// ---- distinct 22-mers in sorted order with their edges and position lists (file_kmer_qsort :1196-1290)
		std::sort(occ.begin(), occ.end(), [](const Occ &a, const Occ &b) { return a.kmer != b.kmer ? a.kmer < b.kmer : a.pos1 < b.pos1; });
		kv_.clear(), point_.clear(), pos_arr_.clear(), edge_.clear(), eflag_.clear();
		for (size_t i = 0; i < occ.size(); ++i) {
			if (i == 0 || occ[i].kmer != occ[i - 1].kmer) kv_.push_back(occ[i].kmer), point_.push_back(i), edge_.push_back(0), eflag_.push_back(0);
			pos_arr_.push_back(occ[i].pos1);
			if (occ[i].in <= 3) edge_.back() |= (uint8_t)(1u << (7 - occ[i].in));
			if (occ[i].out <= 3) edge_.back() |= (uint8_t)(1u << (3 - occ[i].out));
			else eflag_.back() = 1;
		}
		point_.push_back(occ.size());
		const size_t K = kv_.size();
		if (K >= 0xffffffffull) { err_ = "more than 2^32 distinct 22-mers"; return false; }
		build_top();
		out->n_kmer = K;
		out->kmer.resize(K);
		for (size_t i = 0; i < K; ++i) out->kmer[i] = (uint32_t)(kv_[i] & 0xffff);
		out->hash_sparse_id.clear(), out->hash_sparse_cnt.clear();
		for (size_t i = 0; i < K; ++i) {
			const uint64_t h = kv_[i] >> (2 * (KT - K1));
			if (out->hash_sparse_id.empty() || out->hash_sparse_id.back() != h) out->hash_sparse_id.push_back(h), out->hash_sparse_cnt.push_back(0);
			out->hash_sparse_cnt.back()++;
		}
		// ---- the unipath walk (file_kmer_qsort :1448-1760)
		std::vector<uint8_t> useq;                      // concatenated unipath bases
		std::vector<uint64_t> &seqf = out->seqf, &uoff = out->off;
		seqf.assign(1, 0);
		uoff.assign(K, 0);
		uint64_t kmer_off = 0;                          // offset the next k-mer of the current unipath gets
		for (size_t idx = 0; idx < K; ++idx) {
			const int node = node_type(idx);
			if (node == 1 || node == 0) continue;
			const bool fy = node == 2;
			bool ry = false, d = false;
			if (node == 3 || node == 5) {               // more than one way in (or none) and one way out: a unipath starts here
				uoff[idx] = kmer_off++;
				push_bases(useq, kv_[idx], KT);
				ry = true;
			}
			if (node == 4 || node == 6 || node == 8) {  // branching on both sides / a dead end with several ways in: a unipath of its own
				uoff[idx] = kmer_off, kmer_off += KT;
				push_bases(useq, kv_[idx], KT);
				seqf.push_back(useq.size());
				d = node != 8;
				if (node == 8) continue;
			}
			for (int e = 0; e < 4; ++e) {
				if (!((edge_[idx] >> (3 - e)) & 1)) continue;
				uint64_t cur = next_kmer(kv_[idx], e);
				long long r = find(cur);
				if (r < 0) { err_ = "inconsistent k-mer graph (successor not found)"; return false; }
				int nt = node_type((size_t)r);
				bool n_flag = false, l_flag = false;
				const bool joins = nt == 1 || nt == 2 || nt == 7;
				if ((d || fy) && joins) {              // the successor opens a new unipath
					uoff[(size_t)r] = kmer_off++;
					push_bases(useq, cur, KT);
					n_flag = true;
				}
				if (ry && joins) {                     // the successor continues the unipath opened at this node
					useq.push_back((uint8_t)e);
					uoff[(size_t)r] = kmer_off++;
				}
				uint64_t off_before = 0;
				int last_e = e;
				while (node_type((size_t)r) == 1) {    // through the one-in-one-out nodes
					const unsigned oe = edge_[(size_t)r] & 0xf;
					last_e = oe == 1 ? 3 : oe == 2 ? 2 : oe == 4 ? 1 : 0;
					useq.push_back((uint8_t)last_e);
					l_flag = true;
					cur = next_kmer(cur, last_e);
					r = find(cur);
					if (r < 0) { err_ = "inconsistent k-mer graph (successor not found)"; return false; }
					off_before = uoff[(size_t)r];
					uoff[(size_t)r] = kmer_off++;
				}
				nt = node_type((size_t)r);
				if ((nt == 3 || nt == 4 || nt == 8) && l_flag) {   // that node starts / is its own unipath: take it back
					useq.pop_back();
					uoff[(size_t)r] = off_before;
					--kmer_off;
				}
				if (n_flag || l_flag || ry) { seqf.push_back(useq.size()); kmer_off += KT - 1; }
			}
		}
		pack(useq, out->seqb);
		out->seqb.resize((out->seqb.size() + 1023) / 1024 * 1024, 0);     // the reference writes its 1024-word buffer whole (:2278)
		// ---- where every unipath occurs (build_pos_unipath :2100-2270): positions of its first 22-mer that are followed, base by
		// base, by positions of all its other 22-mers
		out->pos.clear();
		out->posp.assign(1, 0);
		const size_t U = seqf.size() - 1;
		std::vector<uint64_t> cand;
		for (size_t u = 0; u < U; ++u) {
			const size_t b = seqf[u], len = seqf[u + 1] - seqf[u];
			if (len < (size_t)KT) { err_ = "unipath shorter than a k-mer"; return false; }
			uint64_t k = 0;
			for (int j = 0; j < KT; ++j) k = (k << 2) | useq[b + j];
			long long r = find(k);
			if (r < 0) { err_ = "unipath k-mer not in the table"; return false; }
			cand.assign(pos_arr_.begin() + point_[(size_t)r], pos_arr_.begin() + point_[(size_t)r + 1]);
			size_t i = 1;
			for (; i + KT <= len; ++i) {
				k = next_kmer(k, useq[b + i + KT - 1]);
				r = find(k);
				if (r < 0) { err_ = "unipath k-mer not in the table"; return false; }
				const uint64_t *lo = pos_arr_.data() + point_[(size_t)r], *hi = pos_arr_.data() + point_[(size_t)r + 1];
				size_t m = 0;
				for (size_t c = 0; c < cand.size(); ++c) if (std::binary_search(lo, hi, cand[c] + 1)) cand[m++] = cand[c] + 1;
				cand.resize(m);
				if (cand.empty()) break;
			}
			for (uint64_t p : cand) out->pos.push_back(p + 1 - i);
			out->posp.push_back(out->pos.size());
		}
		return true;
	}

	// writes the nine files; `dense_hash` also writes the 2 GiB unipath_g.hash the reference's own loader needs, otherwise only
	// unipath_g.hash.sparse ((bucket, count) uint32 pairs), which psvr_index_load accepts
	static bool write_dir(const BuiltIndex &ix, const std::string &dir, bool dense_hash, std::string *err)
	{
		auto put = [&](const char *name, const void *p, size_t n) {
			FILE *f = fopen((dir + "/" + name).c_str(), "wb");
			if (!f || (n && fwrite(p, 1, n, f) != n)) { if (f) fclose(f); *err = std::string("cannot write ") + name; return false; }
			return fclose(f) == 0;
		};
		std::vector<uint32_t> sp;
		for (size_t i = 0; i < ix.hash_sparse_id.size(); ++i) sp.push_back((uint32_t)ix.hash_sparse_id[i]), sp.push_back((uint32_t)ix.hash_sparse_cnt[i]);
		bool ok = put("ref.seq", ix.ref_seq.data(), ix.ref_seq.size() * 8) && put("unipath.seqb", ix.seqb.data(), ix.seqb.size() * 8) &&
		          put("unipath.seqfb", ix.seqf.data(), ix.seqf.size() * 8) && put("unipath.pos", ix.pos.data(), ix.pos.size() * 8) &&
		          put("unipath.posp", ix.posp.data(), ix.posp.size() * 8) && put("unipath_g.kmer", ix.kmer.data(), ix.kmer.size() * 4) &&
		          put("unipath_g.offset", ix.off.data(), ix.off.size() * 8) && put("unipath.chr", ix.chr_text.data(), ix.chr_text.size()) &&
		          put("unipath_g.hash.sparse", sp.data(), sp.size() * 4);
		if (ok && dense_hash) {
			const size_t NB = (size_t)1 << (2 * K1);
			std::vector<uint64_t> h(NB + 1, 0);
			for (size_t i = 0; i < ix.hash_sparse_id.size(); ++i) h[ix.hash_sparse_id[i] + 1] = ix.hash_sparse_cnt[i];
			for (size_t i = 1; i <= NB; ++i) h[i] += h[i - 1];
			ok = put("unipath_g.hash", h.data(), h.size() * 8);
		}
		return ok;
	}
};

} // namespace psvr
