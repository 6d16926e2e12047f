/*
 * oracle/aln_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see aln_oracle.h).
 * Each function cites the reference lines it restates; `rr` = src/PanSVgenerateVCF/read_realignment.
 */
#include "aln_oracle.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include "ksw_oracle.h"

namespace orc {

#define O_MAX(a, b) (((a) > (b)) ? (a) : (b))
#define O_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define O_ABS(a) (((a) > 0) ? (a) : (-(a)))
#define O_ABS_U(a, b) (((a) > (b)) ? ((a) - (b)) : ((b) - (a)))
static const int FORWARD = 1, REVERSE = 0;           // clib/utils.h:72-73
static const uint32_t MAX_U32 = 0xffffffffu;
static const int MAX_I32 = 0x7fffffff;
static const int LEN_KMER = 20, SEED_STEP = 5, UNI_POS_N_MAX = 32;   // rr.hpp:26-29, deBGA_index.hpp:17
static const int MAX_OUTPUT_NUMBER = 6;

// ---------------------------------------------------------------------------------------------
// glibc random_r / srandom_r, TYPE_3 (x**31 + x**3 + 1)
// ---------------------------------------------------------------------------------------------
void Rand3::seed(unsigned s)
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

int32_t Rand3::next()
{
	uint32_t val = (uint32_t)ring[f] + (uint32_t)ring[b];
	ring[f] = (int32_t)val;
	int32_t result = (int32_t)(val >> 1);
	if (++f >= 31) { f = 0; ++b; }
	else if (++b >= 31) b = 0;
	return result;
}

// ---------------------------------------------------------------------------------------------
// index (deBGA_index.cpp:33-80, 354-431)
// ---------------------------------------------------------------------------------------------
template <class T> static bool slurp(const std::string &fn, std::vector<T> *out, size_t pad_bytes = 0)
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

uint64_t Index::hash_at(uint64_t h) const
{
	// buffer_hash_g[h] = number of k-mers in buckets < h (prefix sums written by deBGA)
	auto it = std::lower_bound(bucket_id.begin(), bucket_id.end(), (uint32_t)O_MIN(h, (uint64_t)0xffffffffu));
	if (it == bucket_id.end()) return n_kmer;
	return bucket_start[it - bucket_id.begin()];
}

bool Index::load(const std::string &dir_, const std::vector<std::string> &header_names, std::string *err)
{
	std::string dir = dir_;
	if (dir.back() != '/') dir += '/';
	if (!slurp(dir + "ref.seq", &ref_seq, 536) || !slurp(dir + "unipath.seqb", &seq, 16) || !slurp(dir + "unipath.seqfb", &seqf) ||
	    !slurp(dir + "unipath.pos", &pos) || !slurp(dir + "unipath.posp", &posp) || !slurp(dir + "unipath_g.kmer", &kmer_g) ||
	    !slurp(dir + "unipath_g.offset", &off_g)) { *err = "missing index file in " + dir; return false; }
	n_kmer = kmer_g.size();
	std::vector<uint32_t> sparse;
	if (slurp(dir + "unipath_g.hash.sparse", &sparse)) { // (bucket, count) pairs, see tests/index_fixture.py
		uint64_t acc = 0;
		for (size_t i = 0; i + 1 < sparse.size(); i += 2) {
			bucket_id.push_back(sparse[i]);
			bucket_start.push_back(acc);
			acc += sparse[i + 1];
		}
	} else {
		FILE *f = fopen((dir + "unipath_g.hash").c_str(), "rb");
		if (!f) { *err = "missing unipath_g.hash"; return false; }
		std::vector<uint64_t> buf(1 << 20);
		uint64_t base = 0, prev = 0;
		bool first = true;
		size_t n;
		while ((n = fread(buf.data(), 8, buf.size(), f)) > 0) {
			for (size_t i = 0; i < n; ++i) {
				if (!first && buf[i] != prev) { bucket_id.push_back((uint32_t)(base + i - 1)); bucket_start.push_back(prev); }
				prev = buf[i], first = false;
			}
			base += n;
		}
		fclose(f);
	}
	// unipath.chr: the reference calloc()s the struct, so chr_file_n starts at 0 (rr.cpp:36) and
	// chr_end_n[0] is then overwritten with 1 (deBGA_index.cpp:60-72)
	FILE *fp = fopen((dir + "unipath.chr").c_str(), "r");
	if (!fp) { *err = "missing unipath.chr"; return false; }
	char tok[4096];
	uint32_t line_n = 0;
	chr_file_n = 0;
	chr_names.clear(), chr_end_n.clear();
	while (fscanf(fp, "%4095s", tok) == 1) {
		if ((line_n & 1) == 0) chr_names.push_back(tok);
		else { chr_end_n.push_back((uint32_t)strtoul(tok, 0, 10)); chr_file_n++; }
		line_n++;
	}
	fclose(fp);
	if (chr_file_n == 0) { *err = "empty unipath.chr"; return false; }
	chr_end_n[0] = 1;
	chr_names.resize(chr_file_n + 1);
	chr_names[chr_file_n] = "*";
	chr_end_n.resize(chr_file_n + 1, 0);
	reference_len = chr_end_n[chr_file_n - 1];
	// building_chr_index (deBGA_index.cpp:354-366)
	chr_search_index.assign((reference_len >> 14) + 2, 0);
	uint32_t pos_index_size = 0;
	for (int i = 0; i < chr_file_n; i++) {
		int pos_index = chr_end_n[i] / 0x4000;
		while (pos_index >= (int)pos_index_size) chr_search_index[pos_index_size++] = i;
	}
	chr_search_index[pos_index_size] = chr_file_n;
	// building_bam_header (deBGA_index.cpp:398-431): anchor names ID_chr_st_len_TYPE_bp1_bp2_end_vcfid
	sv_info.clear();
	for (int i = 0; i < chr_file_n; i++) {
		std::vector<std::string> t;
		size_t p = 0;
		const std::string &nm = chr_names[i];
		while (p <= nm.size()) { // strtok semantics: runs of '_' collapse
			while (p < nm.size() && nm[p] == '_') ++p;
			if (p >= nm.size()) break;
			size_t e = nm.find('_', p);
			if (e == std::string::npos) e = nm.size();
			t.push_back(nm.substr(p, e - p));
			p = e + 1;
		}
		t.resize(9);
		SvInfo s;
		s.ID = atoi(t[0].c_str());
		int cid = -1;
		for (size_t k = 0; k < header_names.size(); ++k) if (header_names[k] == t[1]) { cid = (int)k; break; }
		s.chr_ID = (uint32_t)cid;
		s.st_pos = (uint32_t)atoi(t[2].c_str());
		s.region_len = atoi(t[3].c_str());
		s.sv_type = t[4];
		uint64_t ed_pos = (uint64_t)(int64_t)atoi(t[7].c_str());
		s.end_offset = (int)(ed_pos - s.st_pos - s.region_len);
		s.vcf_id = t[8];
		char b[1000];
		snprintf(b, sizeof b, "%d_%d_%ld_%d_%s_%s", s.ID, s.chr_ID, (long)s.st_pos, s.region_len, t[4].c_str(), t[8].c_str());
		s.vcf_print_string = b;
		sv_info.push_back(s);
	}
	return true;
}

int Index::get_chromosome_ID(uint32_t position) const // deBGA_index.cpp:369-396
{
	int file_n = 0;
	int pos_index = position / 0x4000;
	int low = chr_search_index[pos_index];
	int high = chr_search_index[pos_index + 1];
	int pos = position + 1;
	while (low <= high) {
		int mid = (low + high) >> 1;
		if (pos < (int)(chr_end_n[mid] - 1)) high = mid - 1;
		else if (pos > (int)(chr_end_n[mid] - 1)) low = mid + 1;
		else return mid;
		file_n = low;
	}
	return file_n;
}

void Index::get_refseq(uint8_t *ref, uint32_t len, uint32_t start) const // deBGA_index.cpp:307-315
{
	for (uint32_t m = 0; m < len; ++m)
		ref[m] = (ref_seq[(m + start) >> 5] >> ((31 - ((m + start) & 0x1f)) << 1)) & 0x3;
}

// ---------------------------------------------------------------------------------------------
// seeding structures (deBGA_index.hpp:24-72, cpp_lib/graph.hpp:41-107)
// ---------------------------------------------------------------------------------------------
struct VMem { uint64_t uid; uint32_t seed_id, read_pos, uni_pos_off, length, pos_n; };
struct VU { uint64_t uid; uint32_t read_pos, uni_pos_off, length1, length2, pos_n, cov; };
struct USeed { uint32_t read_begin, read_end, seed_id, ref_begin, ref_end, cov; };
struct Path { float dist; int32_t pre_node; uint8_t used; };
struct Edge { uint32_t adj; int weight; float penalty; };

struct Graph { // Graph_handler, cpp_lib/graph.cpp:53-150
	std::vector<USeed> *arr = nullptr;
	uint64_t n = 0;
	std::vector<std::vector<Edge>> pre;
	std::vector<Path> path;
	bool readIsSTR = false;
	float max_distance = 0;
	uint32_t max_index = 0;
	std::vector<int> same_top;
	void process(std::vector<USeed> &v);
};

void Graph::process(std::vector<USeed> &v)
{
	n = v.size();
	arr = &v;
	if (n == 0) return;
	std::stable_sort(v.begin(), v.end(), [](const USeed &a, const USeed &b) { // UNI_SEED::cmp (graph.cpp:14-32), glibc qsort = stable mergesort
		if (a.ref_end != b.ref_end) return a.ref_end < b.ref_end;
		return a.ref_begin < b.ref_begin;
	});
	int max_ref_dis = readIsSTR ? 400 : 50, max_read_dis = readIsSTR ? 400 : 50;
	uint32_t max_search_step = readIsSTR ? 80 : 40, max_gap = readIsSTR ? 20 : 50;
	int search_step = (int)O_MIN(n, (uint64_t)max_search_step);
	bool non_isolated = true;
	if (pre.size() < n) pre.resize(n), path.resize(n);
	for (uint32_t i = 0; i < n; ++i) {
		pre[i].clear();
		path[i].dist = v[i].cov, path[i].pre_node = -1, path[i].used = 0;
	}
	for (uint32_t target = 0; target < n - 1; ++target) {
		uint32_t read_end = v[target].read_end, ref_end = v[target].ref_end, seed_id = v[target].seed_id;
		uint32_t search_end = (uint32_t)O_MIN(n, (uint64_t)(target + search_step));
		for (uint32_t t = target + 1; t < search_end; ++t) {
			if (v[t].seed_id == seed_id) continue;
			if (v[t].ref_end == ref_end) continue;
			int32_t dis_ref = (int32_t)(v[t].ref_begin - ref_end);
			if (dis_ref > max_ref_dis) break;
			int32_t dis_read = (int32_t)(v[t].read_begin - read_end);
			if (dis_read > max_read_dis) continue;
			uint32_t abs_gap = O_ABS_U(dis_read, dis_ref);
			if (abs_gap > max_gap) continue;
			float penalty = (abs_gap == 0) ? 0 : ((abs_gap >> 3) + 3);
			uint32_t weight = 0;
			if (dis_read == dis_ref) weight = v[t].cov - O_MAX(1 - dis_read, 0);
			else if (dis_read > 0 && dis_ref > 0) weight = v[t].cov;
			else if (dis_read >= -5 && dis_read <= 0 && dis_ref >= -5) weight = v[t].cov + O_MIN(dis_read, dis_ref);
			else continue;
			pre[t].push_back(Edge{target, (int)weight, penalty});
			non_isolated = false;
		}
	}
	if (!non_isolated) { // dynamic_programming_path, graph.cpp:125-150
		for (uint32_t target = 0; target < n; ++target) {
			if (pre[target].empty()) continue;
			float cur = 0;
			int32_t pn = -1;
			for (const Edge &e : pre[target]) {
				float temp = path[e.adj].dist + e.weight - e.penalty;
				if (cur <= temp) cur = temp, pn = e.adj;
			}
			path[target].dist = cur, path[target].pre_node = pn;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// KSW_ALN_handler (rr.cpp:815-986)
// ---------------------------------------------------------------------------------------------
enum { ALN_LEFT = 0, ALN_RIGHT = 1, ALN_E2E = 2 };

struct Ksw {
	const Index *idx;
	const Params *P;
	Counters *C;
	const uint8_t *read_str;
	int32_t read_score;
	uint32_t total_q_len;
	bool is_simple_aln;
	std::vector<Cigar> cigar_tmp;
	uint8_t tseq[1600], qseq_rev[1600];
	int8_t mat[25];
	void init(const Index *i, const Params *p, Counters *c)
	{
		idx = i, P = p, C = c;
		int k = 0;
		for (int l = 0; l < 4; ++l) { for (int m = 0; m < 4; ++m) mat[k++] = l == m ? P->match : -P->mismatch; mat[k++] = 0; }
		for (int m = 0; m < 5; ++m) mat[k++] = 0;
		memset(tseq, 0, sizeof tseq);
	}
	void setRead(const uint8_t *r) { cigar_tmp.clear(), read_str = r, read_score = 0, total_q_len = 0; }
	static Cigar mk(char t, uint16_t sz)
	{
		static const char *ops = "MIDNSHP=XB";
		const char *p = strchr(ops, t);
		return Cigar{(uint8_t)(p ? p - ops : 0), (int16_t)sz};
	}
	int get_misMatch(int read_st, int read_ed, int ref_st, int ref_ed) // rr.cpp:893-908
	{
		uint32_t qlen = read_ed - read_st;
		const uint8_t *qseq = read_str + read_st;
		uint32_t tlen = ref_ed - ref_st;
		if (ref_ed < ref_st) tlen = 0, qlen += (ref_st - ref_ed);
		if (!(tlen < 1600)) abort();
		idx->get_refseq(tseq, tlen, ref_st);
		C->ref_bytes += (tlen / 32 + 1) * 8;
		int nm = 0;
		for (uint32_t i = 0; i < qlen; ++i) nm += qseq[i] != tseq[i];
		return nm > 3 ? 3 : nm;
	}
	void alignment(int read_st, int read_ed, int ref_st, int ref_ed, int type) // rr.cpp:910-986
	{
		uint32_t qlen = read_ed - read_st;
		const uint8_t *qseq = read_str + read_st;
		uint32_t tlen = ref_ed - ref_st;
		if (ref_ed < ref_st) tlen = 0, qlen += (ref_st - ref_ed);
		if (!(tlen < 1600)) abort();
		idx->get_refseq(tseq, tlen, ref_st);
		C->ref_bytes += (tlen / 32 + 1) * 8;
		if (type == ALN_LEFT) {
			std::reverse(tseq, tseq + tlen);
			memcpy(qseq_rev, qseq, qlen);
			std::reverse(qseq_rev, qseq_rev + qlen);
			qseq = qseq_rev;
		}
		total_q_len += qlen;
		is_simple_aln = false;
		uint32_t simple_NM = 0;
		if (qlen == 0 || tlen == 0) {
			is_simple_aln = true;
			simple_NM = qlen + tlen;
		} else if (qlen == tlen || type != ALN_E2E) {
			for (uint32_t i = 0; i < qlen && simple_NM < 6; ++i) simple_NM += qseq[i] != tseq[i];
			if (simple_NM == 1 || (simple_NM < 6 && ((simple_NM << 3) < qlen))) is_simple_aln = true;
		}
		orc_extz_t ez;
		std::vector<uint32_t> cig;
		if (!is_simple_aln) { // align_non_splice, rr.cpp:872-891
			if ((int64_t)tlen * qlen > 1000000) {
				memset(&ez, 0, sizeof ez);
				ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
				ez.mqe = ez.mte = ORC_NEG_INF;
				cig = {qlen << 4 | 1, tlen << 4 | 3};
				ez.n_cigar = 2, ez.score = 0;
			} else {
				cig.resize(qlen + tlen + 2);
				orc_extd2(qlen, qseq, tlen, tseq, 5, mat, P->gap_open, P->gap_ex, P->gap_open2, P->gap_ex2, 200, P->zdrop, -1, 0, &ez, cig.data(), (int)cig.size());
				C->dp_calls++, C->dp_cells += (uint64_t)qlen * tlen, C->dp_out_bytes += 4 * ez.n_cigar + 40;
				if (getenv("ORC_DP_SHAPES")) fprintf(stderr, "DP %d %d %d\n", qlen, tlen, (int)type);
			}
		} else C->simple_calls++;
		if (is_simple_aln) {
			if (qlen == 0 || tlen == 0) {
				if (simple_NM != 0) {
					int s1 = P->gap_open + (simple_NM - 1) * P->gap_ex, s2 = P->gap_open2 + (simple_NM - 1) * P->gap_ex2;
					read_score -= O_MIN(s1, s2);
				}
			} else read_score += qlen * P->match - simple_NM * (P->match + P->mismatch);
			if (qlen == 0) cigar_tmp.push_back(mk('D', tlen));
			else if (tlen == 0) cigar_tmp.push_back(mk('I', qlen));
			else cigar_tmp.push_back(mk('M', qlen));
			if (ref_ed < ref_st) cigar_tmp.push_back(mk('D', ref_ed - ref_st));
		} else {
			auto bin = [](uint32_t b) { return Cigar{(uint8_t)(b & 0xf), (int16_t)(b >> 4)}; };
			if (type == ALN_E2E) {
				read_score += ez.score;
				for (int i = ez.n_cigar - 1; i >= 0; i--) cigar_tmp.push_back(bin(cig[i]));
			} else if (type == ALN_LEFT) {
				read_score += ez.mqe;
				for (int i = 0; i < ez.n_cigar; i++) cigar_tmp.push_back(bin(cig[i]));
			} else {
				read_score += ez.mqe;
				for (int i = ez.n_cigar - 1; i >= 0; i--) cigar_tmp.push_back(bin(cig[i]));
			}
		}
	}
};

// get_ksw_score, rr.cpp:308-400
static int get_ksw_score(Graph &g, int first_node, int read_l, Ksw &kswh)
{
	std::vector<Path> &dp = g.path;
	std::vector<USeed> &va = *g.arr;
	int aln_read_begin = read_l, aln_read_end = read_l, aln_ref_begin = MAX_I32, aln_ref_end = MAX_I32;
	int last_aln_begin = read_l, last_ref_begin = MAX_I32, UNITIG_MIS = 0;
	for (; first_node != -1;) {
		int MEM_read_beg = va[first_node].read_begin, MEM_read_end = va[first_node].read_end;
		int MEM_ref_beg = va[first_node].ref_begin, MEM_ref_end = va[first_node].ref_end;
		(void)MEM_read_end;
		aln_read_begin = O_MIN(aln_read_begin, (int)va[first_node].read_end);
		aln_ref_begin = O_MIN(aln_ref_begin, MEM_ref_end);
		if (aln_read_begin <= aln_read_end) {
			if (aln_read_end < last_aln_begin) {
				int MEM_LEN = last_aln_begin - aln_read_end;
				UNITIG_MIS += kswh.get_misMatch(aln_read_end, aln_read_end + MEM_LEN, last_ref_begin, last_ref_begin + MEM_LEN);
				kswh.cigar_tmp.push_back(Ksw::mk('M', (uint16_t)MEM_LEN));
			}
			last_aln_begin = aln_read_begin;
			if (aln_ref_end == MAX_I32) {
				aln_ref_end = aln_ref_begin + (aln_read_end - aln_read_begin) + 30;
				kswh.alignment(aln_read_begin, aln_read_end, aln_ref_begin, aln_ref_end, ALN_RIGHT);
			} else kswh.alignment(aln_read_begin, aln_read_end, aln_ref_begin, aln_ref_end, ALN_E2E);
		} else {
			int distance_read = aln_read_end - aln_read_begin, distance_ref = aln_ref_end - aln_ref_begin;
			if (distance_read != distance_ref) {
				int deletion_len = distance_ref - distance_read;
				int a = O_ABS(deletion_len);
				int s1 = kswh.P->gap_open + (a - 1) * kswh.P->gap_ex, s2 = kswh.P->gap_open2 + (a - 1) * kswh.P->gap_ex2;
				kswh.read_score -= O_MIN(s1, s2);
			}
		}
		aln_read_end = MEM_read_beg;
		last_ref_begin = MEM_ref_beg;
		aln_ref_end = MEM_ref_beg;
		int next_node = dp[first_node].pre_node;
		if (next_node == -1) break;
		first_node = next_node;
	}
	if (aln_read_end < last_aln_begin) {
		int MEM_LEN = last_aln_begin - aln_read_end;
		UNITIG_MIS += kswh.get_misMatch(aln_read_end, aln_read_end + MEM_LEN, last_ref_begin, last_ref_begin + MEM_LEN);
		kswh.cigar_tmp.push_back(Ksw::mk('M', MEM_LEN));
	}
	aln_read_begin = 0, aln_ref_begin = 0;
	int read_begin_alignment = 0;
	if (aln_read_begin < aln_read_end) {
		aln_ref_begin = aln_ref_end - (aln_read_end - aln_read_begin) - 30;
		aln_ref_begin = O_MAX(0, aln_ref_begin);
		kswh.alignment(aln_read_begin, aln_read_end, aln_ref_begin, aln_ref_end, ALN_LEFT);
		if (aln_ref_end > aln_ref_begin) {
			if (kswh.is_simple_aln) read_begin_alignment = aln_ref_end - aln_ref_begin - 30;
			else read_begin_alignment = aln_ref_end - aln_ref_begin;
		}
	}
	kswh.read_score += (read_l - kswh.total_q_len) * kswh.P->match;
	kswh.read_score -= UNITIG_MIS * (kswh.P->match + kswh.P->mismatch);
	return read_begin_alignment;
}

// ---------------------------------------------------------------------------------------------
// single_end_handler (rr.hpp:324-432, rr.cpp:212-293, 406-476, 538-654)
// ---------------------------------------------------------------------------------------------
struct SE {
	const Index *idx;
	Params *P;
	Counters *C;
	Rand3 *grand;         // the process-wide rand() stream
	Rand3 rand_buff;      // initstate_r(rand(), ...) per handler (rr.hpp:340)
	int result_num = 0;
	Result result[2 * MAX_OUTPUT_NUMBER];
	Result *primary_result = nullptr, *secondary_result = nullptr;
	Result ori;
	bool ORI_is_UNMAPPED = false;
	uint64_t read_l = 0;
	Read *c_read = nullptr;
	uint8_t bin_read[2][1600];
	Ksw kswh;
	Graph g[2];
	std::vector<VMem> vm;
	std::vector<VU> vu;
	std::vector<USeed> us[2];
	uint8_t seed_list[1600];
	bool readIsSTR = false;

	void init(const Index *i, Params *p, Counters *c, Rand3 *gr)
	{
		idx = i, P = p, C = c, grand = gr;
		rand_buff.seed((unsigned)grand->next());
		kswh.init(i, p, c);
	}
	void parse_ori(std::string &cm, int read_len); // rr.hpp:392-429
	void binary_read_2_bit();
	void chain_one(int rev);
	int sort_output(Graph &gr, Result &rst, int direction);
	void align();
};

static int atoi_tok(const char *s) { return s ? atoi(s) : 0; }

void SE::parse_ori(std::string &cm, int read_len)
{
	ori.is_ori = true;
	// strtok_r on '_' (first 10 tokens); the touched separators are then rewritten as ',' (rr.hpp:425-427)
	std::vector<char> buf(cm.begin(), cm.end());
	buf.push_back(0);
	int L = (int)cm.size();
	char *save = nullptr;
	char *tok = strtok_r(buf.data(), "_", &save);
	ori.chrID = atoi_tok(tok);
	tok = strtok_r(NULL, "_", &save); ori.ref_bg = atoi_tok(tok);
	tok = strtok_r(NULL, "_", &save); ori.read_bg = atoi_tok(tok);
	tok = strtok_r(NULL, "_", &save); ori.align_score = atoi_tok(tok);
	tok = strtok_r(NULL, "_", &save); ori.mapq = (uint8_t)atoi_tok(tok);
	for (int k = 0; k < 4; ++k) tok = strtok_r(NULL, "_", &save);
	tok = strtok_r(NULL, "_", &save);
	ori.direction = (tok && tok[0] == 'F') ? FORWARD : REVERSE;
	ORI_is_UNMAPPED = (tok && tok[1] == 'Y');
	ori.cigar.clear();
	if (ori.read_bg > 0) ori.cigar.push_back(Ksw::mk('S', ori.read_bg));
	ori.cigar.push_back(Ksw::mk('M', read_len - ori.read_bg));
	ori.sv_id = -1;
	ori.has_mate = false;
	if (ori.ref_bg >= (uint32_t)MAX_I32) ori.ref_bg = 1;
	for (int i = 0; i < L - 1; i++) if (buf[i] == 0) buf[i] = ',';
	cm.assign(buf.data(), L);
}

static const uint8_t *char2dna()
{
	static uint8_t t[256];
	static bool init = false;
	if (!init) { // charToDna5n, rr.cpp:180-202
		memset(t, 0, sizeof t);
		t['C'] = t['c'] = 1, t['G'] = t['g'] = 2, t['T'] = t['t'] = 3, t['n'] = 4;
		init = true;
	}
	return t;
}

void SE::binary_read_2_bit() // rr.cpp:646-654
{
	const uint8_t *tab = char2dna();
	const char *s = c_read->seq.c_str();
	for (int i = 0; s[i]; i++) {
		char ch = s[i];
		if (ch == 'N') ch = "ACGT"[grand->next() % 4];
		uint8_t c = tab[(uint8_t)ch];
		bin_read[0][i] = c;
		bin_read[1][read_l - i - 1] = c ^ 0x3;
	}
}

static inline uint64_t getKmer(uint32_t off, const uint64_t *rb) // rr.cpp:204-210
{
	uint32_t w = off >> 5, iw = off & 0x1f;
	uint64_t full = (rb[w] << (iw << 1)) | (iw == 0 ? 0 : (rb[w + 1] >> ((32 - iw) << 1)));
	return full >> ((32 - LEN_KMER) << 1);
}

static void reverse_qual_quirk(uint8_t *q, int len) // getReverseStr_qual, clib/bam_file.c:341-349 (loop runs to len/2 inclusive)
{
	int half = len >> 1;
	for (int i = 0; i < half + 1; i++) {
		int ri = len - 1 - i;
		uint8_t t = q[i];
		q[i] = q[ri], q[ri] = t;
	}
}

void SE::chain_one(int rev) // chainning_one_read, rr.cpp:538-644
{
	uint64_t rb[52];
	memset(rb, 0, sizeof rb);
	const uint8_t *rs = bin_read[rev];
	for (int i = 0; i < (int)read_l; i++) rb[i >> 5] |= ((uint64_t)rs[i]) << ((31 - (i & 0x1f)) << 1); // binary_read_64_bit, rr.cpp:295-300
	vm.clear(), vu.clear(), us[rev].clear();
	uint32_t kmer_number = read_l - LEN_KMER + 1;
	if (!rev) {
		readIsSTR = false;
		std::map<uint64_t, int> ks;
		for (uint32_t o = 0; o < kmer_number; o++) ks[getKmer(o, rb)]++;
		if (ks.size() < kmer_number - 15) {
			readIsSTR = true;
			for (uint32_t o = 0; o < kmer_number; o++) seed_list[o] = ks[getKmer(o, rb)] >= 4 ? 0 : 1;
			int bg = 0, ed = 0;
			for (uint32_t o = 0; o < (uint32_t)SEED_STEP; o++) {
				bg += seed_list[o] == 0, ed += seed_list[read_l - LEN_KMER - o] == 0;
				seed_list[o] += 2, seed_list[read_l - LEN_KMER - o] += 4;
			}
			if (bg < SEED_STEP && ed < SEED_STEP) {
				int tot = 0;
				for (uint32_t o = 0; tot < SEED_STEP && o < kmer_number; o++) {
					if (seed_list[o] > 0) continue;
					seed_list[o] += 8, tot++;
				}
			}
		}
	} else if (readIsSTR) reverse_qual_quirk(seed_list, read_l - LEN_KMER + 1);
	g[rev].readIsSTR = readIsSTR;

	uint32_t max_search_right = 0;
	for (uint32_t off = 0; off < kmer_number; off += SEED_STEP) {
		if (off + LEN_KMER - 1 <= max_search_right) continue;
		if (readIsSTR && seed_list[off] == 0) continue;
		uint64_t kmer = getKmer(off, rb);
		// search_kmer (deBGA_index.cpp:84-101) + binsearch_range (binarys_qsort.c:25-100)
		uint64_t key = kmer & 0xfff, h = kmer >> 12;
		uint64_t lo = idx->hash_at(h), hi = idx->hash_at(h + 1);
		C->probes++, C->probe_bytes += 16;
		int64_t nb = (int64_t)(hi - lo);
		{ int lg = 0; while ((1ll << lg) < nb + 1) ++lg; C->probe_bytes += 4 * lg; }
		const uint32_t *v = idx->kmer_g.data() + lo;
		int64_t first = -1, last = -1;
		for (int64_t i = 0; i < nb; ++i) if ((v[i] >> 4) == key) { if (first < 0) first = i; last = i; }
		if (first < 0) continue;
		uint64_t hit_bg = lo + first, hit_ed = lo + last;
		if ((hit_ed - hit_bg + 1) > (uint64_t)UNI_POS_N_MAX) continue;
		uint32_t max_right_i = 1;
		for (uint64_t hit = hit_bg; hit <= hit_ed; ++hit) { // UNITIG_MEM_search, deBGA_index.cpp:105-146
			uint64_t kp = idx->off_g[hit];
			int64_t lo2 = 0, hi2 = (int64_t)idx->seqf.size() - 1, uid = -1;
			while (lo2 <= hi2) { // binsearch_interval_unipath64, binarys_qsort.c:162-187
				int64_t mid = (lo2 + hi2) >> 1;
				if (kp < idx->seqf[mid]) hi2 = mid - 1;
				else if (kp > idx->seqf[mid]) lo2 = mid + 1;
				else { uid = mid; break; }
			}
			if (uid < 0) uid = hi2;
			uint64_t ref_pos_n = idx->posp[uid + 1] - idx->posp[uid];
			uint32_t ul = kp - idx->seqf[uid], ur = idx->seqf[uid + 1] - (kp + LEN_KMER);
			uint32_t li, ri;
			const uint64_t *sq = idx->seq.data();
			for (li = 1; li <= ul && li <= off; li++)
				if (((sq[(kp - li) >> 5] >> ((31 - ((kp - li) & 0x1f)) << 1)) & 3) != ((rb[(off - li) >> 5] >> ((31 - ((off - li) & 0x1f)) << 1)) & 3)) break;
			for (ri = 1; ri <= ur && ri <= read_l - off - LEN_KMER; ri++)
				if (((sq[(kp + LEN_KMER - 1 + ri) >> 5] >> ((31 - ((kp + LEN_KMER - 1 + ri) & 0x1f)) << 1)) & 3) !=
				    ((rb[(off + LEN_KMER - 1 + ri) >> 5] >> ((31 - ((off + LEN_KMER - 1 + ri) & 0x1f)) << 1)) & 3)) break;
			VMem m;
			m.uid = uid, m.seed_id = vm.size(), m.read_pos = off + 1 - li, m.uni_pos_off = ul + 1 - li;
			m.length = LEN_KMER + li + ri - 2, m.pos_n = (uint32_t)ref_pos_n;
			vm.push_back(m);
			if (ri > max_right_i) max_right_i = ri;
			{ int lg = 0; while ((1ull << lg) < idx->seqf.size()) ++lg;
			  C->hits++, C->hit_bytes += 8 + 8 * lg + 8 * ((li + ri - 2 + 20) / 32 + 1) + 16; }
		}
		max_search_right = off + LEN_KMER + max_right_i - 1;
	}
	// merge_seed_in_unipath, deBGA_index.cpp:151-217
	uint32_t mem_i = vm.size();
	if (mem_i == 1) {
		VMem &m = vm.back();
		vu.push_back(VU{m.uid, m.read_pos, m.uni_pos_off, m.length, m.length, m.pos_n, m.length});
	} else if (mem_i > 1) {
		std::stable_sort(vm.begin(), vm.end(), [](const VMem &a, const VMem &b) { // vertex_MEM::cmp, deBGA_index.hpp:33-51
			if (a.uid != b.uid) return a.uid < b.uid;
			return a.read_pos < b.read_pos;
		});
		vm.push_back(VMem{~0ull, 0, 0, 0, 0, 0}); // the reference reads one past the end (UB, value unused)
		uint64_t uid_t = vm[0].uid;
		uint32_t j = 0;
		while (j < mem_i) {
			uint32_t s1 = j, cov = vm[s1].length;
			j++;
			while (j < mem_i && uid_t == vm[j].uid && vm[j].uni_pos_off > vm[j - 1].uni_pos_off) {
				int diff = (int)(vm[j].read_pos - vm[j - 1].read_pos - vm[j - 1].length);
				if (diff > 3) break;
				int ce = (vm[j].uni_pos_off - vm[j - 1].uni_pos_off) - (vm[j].read_pos - vm[j - 1].read_pos);
				if (std::abs(ce) < 1) { cov += (diff > 0) ? vm[j].length : (diff + vm[j].length); ++j; }
				else break;
			}
			uint32_t e1 = j - 1;
			VU u;
			u.uid = vm[s1].uid, u.read_pos = vm[s1].read_pos, u.uni_pos_off = vm[s1].uni_pos_off, u.pos_n = vm[s1].pos_n, u.cov = cov;
			if (s1 == e1) u.length1 = u.length2 = vm[s1].length;
			else {
				u.length1 = vm[e1].read_pos + vm[e1].length - vm[s1].read_pos;
				u.length2 = vm[e1].uni_pos_off + vm[e1].length - vm[s1].uni_pos_off;
			}
			vu.push_back(u);
			uid_t = vm[j].uid;
		}
		vm.pop_back();
	}
	// expand_seed, deBGA_index.cpp:219-251 (POS_N_MAX 500 in the built variant)
	for (uint32_t i = 0; i < vu.size(); i++) {
		VU &U = vu[i];
		auto emit = [&](uint32_t m) {
			USeed s;
			s.seed_id = i, s.read_begin = U.read_pos, s.read_end = U.read_pos + U.length1 - 1;
			s.ref_begin = (uint32_t)(idx->pos[m + idx->posp[U.uid]] + U.uni_pos_off - 1);
			s.ref_end = s.ref_begin + U.length2 - 1, s.cov = U.cov;
			us[rev].push_back(s);
			C->seeds++, C->pos_bytes += 8;
		};
		if (U.pos_n > 500) {
			if (U.pos_n > 8000) break;
			for (uint32_t ri = 0; ri < 500; ri++) emit((uint32_t)rand_buff.next() % U.pos_n);
		} else for (uint32_t m = 0; m < U.pos_n; m++) emit(m);
	}
	g[rev].process(us[rev]);
}

int SE::sort_output(Graph &gr, Result &rst, int direction) // rr.cpp:212-293
{
	if (gr.n == 0) return 0;
	gr.max_index = MAX_U32, gr.max_distance = 0;
	gr.same_top.clear();
	gr.same_top.push_back((int)gr.max_index);
	for (int i = (int)gr.n - 1; i >= 0; i--) {
		if (gr.path[i].used) continue;
		float c = gr.path[i].dist;
		if (gr.max_distance < c) {
			gr.max_distance = c, gr.max_index = i;
			gr.same_top.clear();
			gr.same_top.push_back(i);
		} else if (gr.max_distance == c) gr.same_top.push_back(i);
	}
	if (gr.max_index == MAX_U32) return 0;
	int used = 0, unused = 0;
	uint32_t same = gr.same_top.size();
	if (same > 1) gr.max_index = gr.same_top[grand->next() % same];
	int first_node = gr.max_index, orig_first = first_node;
	for (; first_node != -1;) {
		if (gr.path[first_node].used) used++;
		else unused++;
		gr.path[first_node].used = 1;
		int nx = gr.path[first_node].pre_node;
		if (nx == -1) break;
		first_node = nx;
	}
	int orig_final = first_node;
	if (orig_first - orig_final > ((unused + used + 5) << 1))
		for (int k = orig_final; k < orig_first; k++) gr.path[k].used = 1;
	if (used >= unused) return sort_output(gr, rst, direction);
	int ref_begin = (*gr.arr)[first_node].ref_begin;
	int chr_ID = idx->get_chromosome_ID(ref_begin);
	rst.direction = direction;
	rst.max_index = gr.max_index;
	rst.chain_score = gr.max_distance;
	rst.read_bg = (*gr.arr)[first_node].read_begin;
	rst.chrID = chr_ID;
	rst.ref_bg = ref_begin - idx->chr_end_n[chr_ID - 1];
	return 1;
}

static bool try_merge(Cigar &a, const Cigar &cp, bool *bad) // CIGAR_PATH::try_merge, rr.hpp:159-178
{
	if (cp.size < 0) {
		if (cp.type != 2) { *bad = true; return true; }
		if (a.type == 0) { a.size += cp.size; if (!(a.size > 0)) *bad = true; return true; }
		else if (a.type == 2) { a.size -= cp.size; if (!(a.size > 0)) *bad = true; return true; }
		else { *bad = true; return true; }
	} else if (a.type == cp.type || cp.size == 0) {
		a.size += cp.size;
		return true;
	}
	return false;
}

static bool reverse_cigar(Result &r, std::vector<Cigar> &tmp, int read_len, bool *bad) // reverseGIGAR, rr.hpp:277-301
{
	r.cigar.clear();
	r.cigar.push_back(tmp.back());
	for (int i = (int)tmp.size() - 2; i >= 0; i--)
		if (!try_merge(r.cigar.back(), tmp[i], bad)) r.cigar.push_back(tmp[i]);
	if (!r.cigar.empty() && r.cigar[0].size == 0) r.cigar.erase(r.cigar.begin());
	int tot = 0;
	for (auto &c : r.cigar) if (c.type == 0 || c.type == 1 || c.type == 3 || c.type == 4) tot += c.size;
	return tot == read_len;
}

void SE::align() // rr.cpp:406-476
{
	result_num = 0;
	primary_result = secondary_result = nullptr;
	parse_ori(c_read->comment, read_l);
	if (ori.chrID > 24) ORI_is_UNMAPPED = true;
	if (!ORI_is_UNMAPPED && ori.align_score == read_l * P->match) return;
	C->reads++, C->read_bytes += read_l;
	binary_read_2_bit();
	for (int o = 0; o < 2; o++) chain_one(o);
	uint32_t max_chain = 0;
	for (int o = 0; o < 2; o++) {
		int direction = o == 0 ? FORWARD : REVERSE;
		for (int i = 0; i < MAX_OUTPUT_NUMBER; i++) {
			int rst = sort_output(g[o], result[result_num], direction);
			if (rst == 0) break;
			uint32_t c = result[result_num].chain_score;
			max_chain = O_MAX(c, max_chain);
			if (c + 30 < max_chain || c < 30) break;
			result_num++;
		}
	}
	std::stable_sort(result, result + result_num, [](const Result &a, const Result &b) { // cmp_chain_score, rr.hpp:303-308
		if (a.chain_score != b.chain_score) return a.chain_score > b.chain_score;
		return a.max_index < b.max_index;
	});
	if (result_num == 0 || max_chain < 20) return;
	for (int k = 0; k < result_num; k++) {
		Result &c = result[k];
		if (c.chain_score + 30 < max_chain) { result_num = k; break; }
		int is_rev = c.direction == REVERSE;
		kswh.setRead(bin_read[is_rev]);
		int rba = get_ksw_score(g[is_rev], c.max_index, read_l, kswh);
		c.ref_bg -= rba;
		c.align_score = O_MAX(kswh.read_score, 0);
		bool bad = false;
		reverse_cigar(c, kswh.cigar_tmp, read_l, &bad);
		C->cand_bytes += 64;
	}
	std::stable_sort(result, result + result_num, [](const Result &a, const Result &b) { // cmp_align_score, rr.hpp:310-315
		if (a.align_score != b.align_score) return a.align_score > b.align_score;
		return a.max_index < b.max_index;
	});
	if (result[0].align_score < 40) { result_num = 0; return; }
	for (int i = 0; i < result_num; i++) {
		uint32_t sv = result[i].chrID;
		result[i].sv_id = sv;
		const SvInfo &s = idx->sv_info[sv];
		result[i].chrID = s.chr_ID;
		result[i].ref_bg += s.st_pos;
		if (result[i].ref_bg >= (uint32_t)MAX_I32) result[i].ref_bg = 5;
		result[i].is_ori = false, result[i].rst_idx = i, result[i].mapq = 0, result[i].has_mate = false;
	}
	if (result_num > 0) {
		int32_t d = result[0].align_score - (result_num > 1 ? result[1].align_score : 0);
		result[0].mapq = d > 40 ? 40 : d;
	}
}

// ---------------------------------------------------------------------------------------------
// PE_score (rr.hpp:434-628)
// ---------------------------------------------------------------------------------------------
struct PE {
	const Index *idx;
	Rand3 *grand;
	int max_same, max_score;
	bool proper;
	int cur_isize;
	bool gain;
	Result *max_1, *max_2;
	int max_isize, min_isize, normal_read_len, min_filter_score;
	void init(int mx, int mn, int rl, int mfs)
	{
		max_isize = mx + 200, min_isize = mn - 200;
		min_isize = O_MAX(0, min_isize);
		normal_read_len = rl, min_filter_score = mfs;
		clear();
	}
	void clear() { max_same = 1, max_score = 0, max_1 = max_2 = nullptr, cur_isize = 0, proper = false, gain = false; }
	int get_isize(int p1, int p2, int d1, int d2)
	{
		if (d1 == d2) return 0;
		int isize = normal_read_len + ((d1 == FORWARD) ? (p2 - p1) : (p1 - p2));
		return (isize < max_isize && isize > min_isize) ? isize : 0;
	}
	int proper_mated(Result *a, Result *b)
	{
		if (!a || !b || a->chrID != b->chrID) return 0;
		int s1p1 = a->ref_bg, s1p2 = s1p1 + (a->is_ori ? 0 : idx->sv_info[a->sv_id].end_offset);
		int s2p1 = b->ref_bg, s2p2 = s2p1 + (b->is_ori ? 0 : idx->sv_info[b->sv_id].end_offset);
		int is;
		if ((is = get_isize(s1p1, s2p1, a->direction, b->direction)) > 0) return is;
		if ((is = get_isize(s1p1, s2p2, a->direction, b->direction)) > 0) return is;
		if ((is = get_isize(s1p2, s2p1, a->direction, b->direction)) > 0) return is;
		if ((is = get_isize(s1p2, s2p2, a->direction, b->direction)) > 0) return is;
		return 0;
	}
	void store(Result *a, Result *b)
	{
		int ISIZE = proper_mated(a, b);
		int basic = (a ? a->align_score : 0) + (b ? b->align_score : 0);
		bool one_new = (a && !a->is_ori) || (b && !b->is_ori);
		int fin = basic + (ISIZE > 0 ? 0 : -60) + (one_new ? 0 : 1);
		if (fin >= max_score) {
			bool st = true;
			if (fin > max_score) max_same = 1;
			else if (fin == max_score) { max_same++; if (grand->next() % max_same != 0) st = false; }
			if (st) max_1 = a, max_2 = b, max_score = fin, cur_isize = ISIZE, proper = cur_isize > 0;
		}
	}
	void best(SE *h)
	{
		clear();
		int n0 = h[0].result_num, n1 = h[1].result_num;
		if (!h[0].ORI_is_UNMAPPED) n0++;
		if (!h[1].ORI_is_UNMAPPED) n1++;
		auto R = [&](int k, int i) { return i < h[k].result_num ? &h[k].result[i] : &h[k].ori; };
		for (int i = 0; i < n0; i++) store(R(0, i), nullptr);
		for (int j = 0; j < n1; j++) store(nullptr, R(1, j));
		for (int i = 0; i < n0; i++) for (int j = 0; j < n1; j++) store(R(0, i), R(1, j));
		gain = max_score > 0 && ((max_1 && !max_1->is_ori) || (max_2 && !max_2->is_ori));
	}
	void set_primary(SE *h) // set_primary_secondary_mate, rr.hpp:501-534
	{
		for (int i = 0; i < 2; i++) {
			Result *c = i == 0 ? max_1 : max_2;
			if (!c) continue;
			SE *s = h + i;
			s->primary_result = c;
			s->secondary_result = nullptr;
			if (c->is_ori && s->result_num > 0) s->secondary_result = &s->result[0];
			else if (s->result_num > 1) s->secondary_result = c->rst_idx == 0 ? &s->result[1] : &s->result[0];
			Result *m = i == 0 ? max_2 : max_1;
			if (m && m->chrID != MAX_U32) {
				c->has_mate = true, c->mate_chrID = m->chrID, c->mate_ref_bg = m->ref_bg, c->mate_sv_id = m->sv_id;
				if (c->is_ori) c->sv_id = c->mate_sv_id;
			} else c->has_mate = false, c->mate_chrID = 0, c->mate_sv_id = -1;
		}
	}
};

// ---------------------------------------------------------------------------------------------
struct Aligner {
	const Index *idx;
	Params P;
	Counters C;
	Rand3 grand;
	SE h[2];
	PE ps;
};

Aligner *aligner_create(const Index *idx, const Params *par)
{
	Aligner *a = new Aligner;
	a->idx = idx;
	if (par) a->P = *par;                   // MAP_PARA::get_option's scoring options (-M -m -O -E -P -F -z), before the handlers read them
	a->grand.seed(1);                       // rand() is never seeded by the reference
	a->ps.idx = idx, a->ps.grand = &a->grand;
	a->ps.init(0, 0, 0, 0);                  // rr.cpp:64 (options still zero)
	a->h[0].init(idx, &a->P, &a->C, &a->grand); // rr.cpp:65-66: two rand() draws at -t 1
	a->h[1].init(idx, &a->P, &a->C, &a->grand);
	return a;
}
void aligner_destroy(Aligner *a) { delete a; }
const Counters &aligner_counters(const Aligner *a) { return a->C; }

static uint64_t fnv(uint64_t h, uint64_t v)
{
	for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ULL; }
	return h;
}

static void print_result(std::string &o, const Result &r, bool is_ori)
{
	char b[256];
	snprintf(b, sizeof b, "[%u,%u,%d,%u,%u,%d,%d,\"", r.align_score, is_ori ? 0 : r.chain_score, (int)r.chrID, r.ref_bg, r.read_bg, r.direction, (int)r.mapq);
	o += b;
	for (auto &c : r.cigar) { snprintf(b, sizeof b, "%d%c", c.size, "MIDNSHP=XB"[c.type]); o += b; }
	o += "\"]";
}

static int which(SE &h, Result *p)
{
	if (!p) return -1;
	if (p == &h.ori) return -2;
	return (int)(p - h.result);
}

std::string aligner_pair(Aligner *a, Read &r1, Read &r2, long pair_i, bool trace)
{
	Params &P = a->P;
	if (!P.stat_set) { // load_reads, rr.cpp:134-148
		const char *st = strstr(r1.comment.c_str(), "STAT_");
		if (!st || sscanf(st + 5, "%d_%d_%d_%d_", &P.normal_read_length, &P.isize_min, &P.isize_mid, &P.isize_max) == -1)
			P.normal_read_length = 150, P.isize_min = 100, P.isize_mid = 500, P.isize_max = 900;
		int mfs = P.normal_read_length * P.match * 2 - 80;
		mfs = O_MAX(mfs, 50);
		a->ps.init(P.isize_max, P.isize_min, P.normal_read_length, mfs);
		P.stat_set = true;
	}
	SE *h = a->h;
	for (int k = 0; k < 2; k++) {
		h[k].c_read = k == 0 ? &r1 : &r2;
		h[k].read_l = h[k].c_read->seq.size();
		h[k].align();
	}
	PE &ps = a->ps;
	ps.best(h);
	if (ps.gain) ps.set_primary(h);
	std::string o;
	char b[256];
	snprintf(b, sizeof b, "{\"i\":%ld,\"reads\":[", pair_i);
	o += b;
	for (int k = 0; k < 2; ++k) {
		SE &s = h[k];
		snprintf(b, sizeof b, "%s{\"n\":%d,\"unmapped\":%d,\"res\":[", k ? "," : "", s.result_num, (int)s.ORI_is_UNMAPPED);
		o += b;
		for (int i = 0; i < s.result_num; ++i) { if (i) o += ","; print_result(o, s.result[i], false); }
		o += "],\"ori\":";
		print_result(o, s.ori, true);
		if (ps.gain) {
			Result *pr = s.primary_result;
			snprintf(b, sizeof b, ",\"prim\":%d,\"sec\":%d", which(s, pr), which(s, s.secondary_result));
			o += b;
			if (pr) { snprintf(b, sizeof b, ",\"mate\":[%d,%u,%u]", (int)pr->has_mate, pr->has_mate ? pr->mate_chrID : 0, pr->has_mate ? pr->mate_ref_bg : 0); o += b; }
		}
		if (trace) {
			snprintf(b, sizeof b, ",\"str\":%d,\"tr\":[", (int)s.readIsSTR);
			o += b;
			for (int d = 0; d < 2; ++d) {
				uint64_t hs = 1469598103934665603ULL, hd = 1469598103934665603ULL;
				for (auto &u : s.us[d]) { hs = fnv(hs, u.read_begin); hs = fnv(hs, u.read_end); hs = fnv(hs, u.seed_id); hs = fnv(hs, u.ref_begin); hs = fnv(hs, u.ref_end); hs = fnv(hs, u.cov); }
				for (size_t i = 0; i < s.us[d].size(); ++i) { hd = fnv(hd, (uint64_t)(int64_t)s.g[d].path[i].dist); hd = fnv(hd, (uint64_t)(int64_t)s.g[d].path[i].pre_node); }
				snprintf(b, sizeof b, "%s[%zu,\"%016llx\",\"%016llx\"]", d ? "," : "", s.us[d].size(), (unsigned long long)hs, (unsigned long long)hd);
				o += b;
			}
			o += "]";
		}
		o += "}";
	}
	snprintf(b, sizeof b, "],\"pe\":[%d,%d,%d,%d,%d,%d]}", ps.max_score, ps.cur_isize, (int)ps.proper, (int)ps.gain, which(h[0], ps.max_1), which(h[1], ps.max_2));
	o += b;
	return o;
}

bool read_header_names(const std::string &path, std::vector<std::string> *names)
{
	FILE *h = fopen(path.c_str(), "r");
	if (!h) return false;
	char buf[4096];
	while (fgets(buf, sizeof buf, h)) {
		if (strncmp(buf, "@SQ", 3)) continue;
		char *p = strstr(buf, "SN:");
		if (!p) continue;
		p += 3;
		char *e = p;
		while (*e && *e != '\t' && *e != '\n') ++e;
		names->emplace_back(p, e - p);
	}
	fclose(h);
	return true;
}

bool read_fastq_record(FILE *f, Read *r)
{
	static thread_local char *line = nullptr;
	static thread_local size_t cap = 0;
	std::string l[4];
	for (int i = 0; i < 4; ++i) {
		ssize_t n = getline(&line, &cap, f);
		if (n <= 0) return false;
		while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
		l[i] = line;
	}
	size_t sp = l[0].find_first_of(" \t");
	r->name = l[0].substr(1, sp == std::string::npos ? std::string::npos : sp - 1);
	r->comment = sp == std::string::npos ? "" : l[0].substr(sp + 1);
	r->seq = l[1], r->qual = l[3];
	return true;
}

} // namespace orc

#ifdef ORC_ALN_MAIN
// aln_oracle <index_dir> <reads.fq> <header.sam> [--trace] [--limit N] [--stats [--print]] [--score M,m,O,E,P,F,z]
// stderr: "ALIGN_SECONDS <s>" = wall of the per-pair loop alone (index load excluded), which bench.py reports.
int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: aln_oracle <index_dir> <reads.fq> <header.sam> [--trace] [--limit N] [--stats]\n"); return 1; }
	bool trace = false, stats = false, print = false;
	long limit = -1;
	orc::Params par;
	for (int i = 4; i < argc; ++i) {
		if (!strcmp(argv[i], "--trace")) trace = true;
		else if (!strcmp(argv[i], "--stats")) stats = true;
		else if (!strcmp(argv[i], "--print")) print = true;
		else if (!strcmp(argv[i], "--limit") && i + 1 < argc) limit = atol(argv[++i]);
		else if (!strcmp(argv[i], "--score") && i + 1 < argc) {
			if (sscanf(argv[++i], "%d,%d,%d,%d,%d,%d,%d", &par.match, &par.mismatch, &par.gap_open, &par.gap_ex, &par.gap_open2, &par.gap_ex2, &par.zdrop) != 7) { fprintf(stderr, "--score wants seven integers\n"); return 1; }
		}
	}
	std::vector<std::string> names;
	if (!orc::read_header_names(argv[3], &names)) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
	orc::Index idx;
	std::string err;
	if (!idx.load(argv[1], names, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
	orc::Aligner *a = orc::aligner_create(&idx, &par);
	FILE *fq = fopen(argv[2], "r");
	if (!fq) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
	orc::Read r1, r2;
	long i = 0;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	while ((limit < 0 || i < limit) && orc::read_fastq_record(fq, &r1) && orc::read_fastq_record(fq, &r2)) {
		std::string line = orc::aligner_pair(a, r1, r2, i, trace);
		if (!stats || print) puts(line.c_str());
		++i;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	fprintf(stderr, "ALIGN_SECONDS %.6f\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
	if (stats) {
		const orc::Counters &c = orc::aligner_counters(a);
		printf("{\"pairs\":%ld,\"reads_aligned\":%llu,\"probes\":%llu,\"hits\":%llu,\"seeds\":%llu,\"dp_calls\":%llu,\"simple_calls\":%llu,\"dp_cells\":%llu,"
		       "\"bytes\":{\"read\":%llu,\"probe\":%llu,\"hit\":%llu,\"pos\":%llu,\"ref\":%llu,\"dp_out\":%llu,\"cand\":%llu,\"total\":%llu}}\n",
		       i, (unsigned long long)c.reads, (unsigned long long)c.probes, (unsigned long long)c.hits, (unsigned long long)c.seeds,
		       (unsigned long long)c.dp_calls, (unsigned long long)c.simple_calls, (unsigned long long)c.dp_cells,
		       (unsigned long long)c.read_bytes, (unsigned long long)c.probe_bytes, (unsigned long long)c.hit_bytes, (unsigned long long)c.pos_bytes,
		       (unsigned long long)c.ref_bytes, (unsigned long long)c.dp_out_bytes, (unsigned long long)c.cand_bytes, (unsigned long long)c.total());
	}
	return 0;
}
#endif

extern "C" int orc_rand_selftest(unsigned seed, int32_t *out, int n)
{
	orc::Rand3 r;
	r.seed(seed);
	for (int i = 0; i < n; ++i) out[i] = r.next();
	return 0;
}
