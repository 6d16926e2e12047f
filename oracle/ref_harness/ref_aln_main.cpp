// oracle/ref_harness/ref_aln_main.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the REFERENCE's own aligner objects (compiled by oracle/Makefile from the sources where
// they lie under /root/reference/src: PanSVgenerateVCF/read_realignment.cpp, deBGA_index.cpp,
// cpp_lib/graph.cpp, clib/binarys_qsort.c, clib/bam_file.c, clib/utils.c, kswlib/*) through the
// body of align_read_pair (read_realignment.cpp:745-775) and prints what it decided, one line per
// read pair.  Only the htslib-dependent formatting (output_BAM / sam_parse1) is not exercised:
// the vendored htslib cannot be built here (cram_io.c needs <lzma.h>), so those functions are
// dropped by --gc-sections and never referenced.
//
// Usage: ref_aln <index_dir> <reads.fq> <header.sam> [--trace] [--limit N]   (--limit: stop after N pairs; bench.py times it)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <sys/mman.h>
#include <string>
#include <vector>
#include "PanSVgenerateVCF/read_realignment.hpp"

static std::vector<std::string> header_names;

static int name2id(const char *nm)
{
	for (size_t i = 0; i < header_names.size(); ++i)
		if (header_names[i] == nm) return (int)i;
	return -1;
}

static uint64_t load_file(const std::string &fn, void **data, size_t pad)
{
	FILE *f = fopen(fn.c_str(), "rb");
	if (!f) { fprintf(stderr, "cannot open %s\n", fn.c_str()); exit(2); }
	fseek(f, 0, SEEK_END);
	uint64_t n = ftell(f);
	rewind(f);
	if (pad == 0 && n > (64u << 20)) { // the 2 GiB first-level table: map it instead of copying (load time is not what is measured)
		void *m = mmap(NULL, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_POPULATE, fileno(f), 0); // prefaulted: no first-touch faults inside the timed loop
		if (m == MAP_FAILED) { fprintf(stderr, "mmap %s failed\n", fn.c_str()); exit(2); }
		*data = m;
		fclose(f);
		return n;
	}
	*data = calloc(n + pad, 1);
	if (fread(*data, 1, n, f) != n) { fprintf(stderr, "short read %s\n", fn.c_str()); exit(2); }
	fclose(f);
	return n;
}

// what deBGA_INDEX::load_index_file (deBGA_index.cpp:33-80) does, minus building_bam_header's htslib call
static void load_index(deBGA_INDEX *idx, std::string dir)
{
	if (dir.back() != '/') dir += '/';
	idx->result_ref_seq = load_file(dir + "ref.seq", (void **)&idx->buffer_ref_seq, 536) >> 3;
	idx->result_seq = load_file(dir + "unipath.seqb", (void **)&idx->buffer_seq, 0) >> 3;
	idx->result_seqf = load_file(dir + "unipath.seqfb", (void **)&idx->buffer_seqf, 0) >> 3;
	idx->result_p = load_file(dir + "unipath.pos", (void **)&idx->buffer_p, 0) >> 3;
	idx->result_pp = load_file(dir + "unipath.posp", (void **)&idx->buffer_pp, 0) >> 3;
	idx->result_hash_g = load_file(dir + "unipath_g.hash", (void **)&idx->buffer_hash_g, 0) >> 3;
	idx->result_kmer_g = load_file(dir + "unipath_g.kmer", (void **)&idx->buffer_kmer_g, 0) >> 2;
	idx->result_off_g = load_file(dir + "unipath_g.offset", (void **)&idx->buffer_off_g, 0) >> 2;
	FILE *fp = fopen((dir + "unipath.chr").c_str(), "r");
	if (!fp) { fprintf(stderr, "cannot open unipath.chr\n"); exit(2); }
	uint32_t line_n = 0;
	// NB: the reference xcalloc()s deBGA_INDEX (read_realignment.cpp:36), so the `chr_file_n = 1` member
	// initialiser never runs and names/ends are filled from slot 0; chr_end_n[0] is then overwritten below.
	if (fscanf(fp, "%s", idx->chr_line_content) != 1) exit(2);
	while (!feof(fp)) {
		if ((line_n & 1) == 0) strcpy(idx->chr_names[idx->chr_file_n], idx->chr_line_content);
		else sscanf(idx->chr_line_content, "%u", &idx->chr_end_n[idx->chr_file_n++]);
		line_n++;
		if (fscanf(fp, "%s", idx->chr_line_content) != 1) break;
	}
	idx->chr_end_n[0] = START_POS_REF + 1;
	strcpy(idx->chr_names[idx->chr_file_n], "*");
	idx->reference_len = idx->chr_end_n[idx->chr_file_n - 1];
	fclose(fp);
	idx->building_chr_index();
	// building_bam_header (deBGA_index.cpp:398-431) with bam_name2id replaced by a lookup in header.sam
	char tmp[1024];
	for (int i = 0; i < idx->chr_file_n; i++) {
		strcpy(tmp, idx->chr_names[i]);
		char *token = strtok(tmp, "_");
		int id = token ? atoi(token) : 0;
		token = strtok(NULL, "_"); int char_ID = token ? name2id(token) : -1;
		token = strtok(NULL, "_"); uint32_t pos = token ? atoi(token) : 0;
		token = strtok(NULL, "_"); int region_len = token ? atoi(token) : 0;
		token = strtok(NULL, "_"); char *sv_type = token ? token : (char *)"";
		token = strtok(NULL, "_"); int bp1 = token ? atoi(token) : 0;
		token = strtok(NULL, "_"); int bp2 = token ? atoi(token) : 0;
		token = strtok(NULL, "_"); int ed = token ? atoi(token) : 0;
		token = strtok(NULL, "_"); char *vcf_id = token ? token : (char *)"";
		idx->sv_info.emplace_back(id, char_ID, pos, region_len, sv_type, bp1, bp2, ed, vcf_id);
	}
}

static void set_ks(kstring_t *k, const std::string &s)
{
	k->l = s.size();
	k->m = s.size() + 1;
	k->s = (char *)realloc(k->s, k->m);
	memcpy(k->s, s.c_str(), k->m);
}

static bool read_record(FILE *f, kseq_t *ks)
{
	static char *line = NULL;
	static size_t cap = 0;
	std::string l[4];
	for (int i = 0; i < 4; ++i) {
		ssize_t n = getline(&line, &cap, f);
		if (n <= 0) return false;
		while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
		l[i] = line;
	}
	size_t sp = l[0].find_first_of(" \t");
	set_ks(&ks->name, l[0].substr(1, sp == std::string::npos ? std::string::npos : sp - 1));
	set_ks(&ks->comment, sp == std::string::npos ? "" : l[0].substr(sp + 1));
	set_ks(&ks->seq, l[1]);
	set_ks(&ks->qual, l[3]);
	return true;
}

static uint64_t fnv(uint64_t h, uint64_t v)
{
	for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ULL; }
	return h;
}

static void print_result(MAX_IDX_OUTPUT &r, bool is_ori)
{
	printf("[%u,%u,%d,%u,%u,%d,%d,\"", r.align_score, is_ori ? 0 : r.chain_score, (int)r.chrID, r.ref_bg, r.read_bg, r.direction, (int)r.mapq);
	for (auto &c : r.cigar) printf("%d%c", c.size, OUT_BAM_CIGAR_STR[c.type]);
	printf("\"]");
}

static int which(single_end_handler &h, MAX_IDX_OUTPUT *p)
{
	if (p == NULL) return -1;
	if (p == &h.ori) return -2;
	return (int)(p - h.result);
}

int main(int argc, char **argv)
{
	if (argc < 4) { fprintf(stderr, "usage: ref_aln <index_dir> <reads.fq> <header.sam> [--trace]\n"); return 1; }
	bool trace = false;
	long limit = -1;
	for (int a = 4; a < argc; ++a) {
		if (!strcmp(argv[a], "--trace")) trace = true;
		else if (!strcmp(argv[a], "--limit") && a + 1 < argc) limit = atol(argv[++a]);
	}
	{
		FILE *h = fopen(argv[3], "r");
		if (!h) { fprintf(stderr, "cannot open %s\n", argv[3]); return 2; }
		char buf[4096];
		while (fgets(buf, sizeof buf, h)) {
			if (strncmp(buf, "@SQ", 3)) continue;
			char *p = strstr(buf, "SN:");
			if (!p) continue;
			p += 3;
			char *e = p;
			while (*e && *e != '\t' && *e != '\n') ++e;
			header_names.emplace_back(p, e - p);
		}
		fclose(h);
	}
	MAP_PARA *o = (MAP_PARA *)calloc(1, sizeof(MAP_PARA));
	o->thread_n = 1;
	o->match_D = MATCH_SCORE, o->mismatch_D = MISMATCH_SCORE;
	o->gap_open_D = GAP_OPEN, o->gap_ex_D = GAP_EXT, o->gap_open2_D = GAP_OPEN2, o->gap_ex2_D = GAP_EXT2;
	o->zdrop_D = ZDROP_SCORE, o->bw = BANDWIDTH, o->max_use_read = MAX_int32t;
	deBGA_INDEX *idx = (deBGA_INDEX *)calloc(1, sizeof(deBGA_INDEX));
	new (&idx->sv_info) std::vector<SV_chr_info>();
	load_index(idx, argv[1]);
	// read_realignment.cpp:62-67 with thread_n == 1: two rand() draws seed the two handlers
	Classify_buff_pool *buff = (Classify_buff_pool *)calloc(1, sizeof(Classify_buff_pool));
	new (buff) Classify_buff_pool();
	buff->ps.init(o->ISIZE_MAX, o->ISIZE_MIN, o->normal_read_length, 0);
	buff->SE_h[0].init(o, idx);
	buff->SE_h[1].init(o, idx);

	FILE *fq = fopen(argv[2], "r");
	if (!fq) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
	kseq_t r1, r2;
	memset(&r1, 0, sizeof r1), memset(&r2, 0, sizeof r2);
	long pair_i = 0;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	while ((limit < 0 || pair_i < limit) && read_record(fq, &r1) && read_record(fq, &r2)) {
		if (!o->read_status_options_already_set) { // read_realignment.cpp:134-148
			char *statu_str = strstr(r1.comment.s, "STAT_");
			if (statu_str == NULL || sscanf(statu_str + 5, "%d_%d_%d_%d_", &(o->normal_read_length), &(o->ISIZE_MIN), &(o->ISIZE_MID), &(o->ISIZE_MAX)) == -1) {
				o->normal_read_length = 150, o->ISIZE_MIN = 100, o->ISIZE_MID = 500, o->ISIZE_MAX = 900;
			}
			int min_filter_score = o->normal_read_length * o->match_D * 2 - 80;
			min_filter_score = MAX(min_filter_score, 50);
			buff->ps.init(o->ISIZE_MAX, o->ISIZE_MIN, o->normal_read_length, min_filter_score);
			o->read_status_options_already_set = true;
		}
		// body of align_read_pair, read_realignment.cpp:750-767
		single_end_handler *SE_h = &(buff->SE_h[0]);
		for (int read_id = 0; read_id < 2; read_id++) {
			SE_h[read_id].read_register((read_id == 0) ? &r1 : &r2);
			SE_h[read_id].align();
		}
		PE_score *ps = &(buff->ps);
		ps->read_get_best_pairing_results(SE_h);
		if (ps->pan_genome_gain_better_result) ps->set_primary_secondary_mate(SE_h);
		printf("{\"i\":%ld,\"reads\":[", pair_i);
		for (int k = 0; k < 2; ++k) {
			single_end_handler &h = SE_h[k];
			printf("%s{\"n\":%d,\"unmapped\":%d,\"res\":[", k ? "," : "", h.result_num, (int)h.ORI_is_UNMAPPED);
			for (int i = 0; i < h.result_num; ++i) { if (i) printf(","); print_result(h.result[i], false); }
			printf("],\"ori\":");
			print_result(h.ori, true);
			if (ps->pan_genome_gain_better_result) {
				MAX_IDX_OUTPUT *pr = h.primary_result;
				printf(",\"prim\":%d,\"sec\":%d", which(h, pr), which(h, h.secondary_result));
				if (pr) printf(",\"mate\":[%d,%u,%u]", (int)pr->has_mate, pr->has_mate ? pr->mate_chrID : 0, pr->has_mate ? pr->mate_ref_bg : 0);
			}
			if (trace) {
				printf(",\"str\":%d,\"tr\":[", (int)h.readIsSTR);
				for (int s = 0; s < 2; ++s) {
					uint64_t hs = 1469598103934665603ULL, hd = 1469598103934665603ULL;
					auto &us = h.uniseed_v[s];
					for (auto &u : us) { hs = fnv(hs, u.read_begin); hs = fnv(hs, u.read_end); hs = fnv(hs, u.seed_id); hs = fnv(hs, u.ref_begin); hs = fnv(hs, u.ref_end); hs = fnv(hs, u.cov); }
					for (size_t i = 0; i < us.size(); ++i) { hd = fnv(hd, (uint64_t)(int64_t)h.g[s].dist_path[i].dist); hd = fnv(hd, (uint64_t)(int64_t)h.g[s].dist_path[i].pre_node); }
					printf("%s[%zu,\"%016llx\",\"%016llx\"]", s ? "," : "", us.size(), (unsigned long long)hs, (unsigned long long)hd);
				}
				printf("]");
			}
			printf("}");
		}
		printf("],\"pe\":[%d,%d,%d,%d,%d,%d]}\n", ps->max_score, ps->cur_isize, (int)ps->read_pair_is_proper_mated, (int)ps->pan_genome_gain_better_result,
		       which(SE_h[0], ps->max_1), which(SE_h[1], ps->max_2));
		pair_i++;
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	// wall of the per-pair loop alone (index load excluded): what bench.py reports as the reference CPU baseline
	fprintf(stderr, "ALIGN_SECONDS %.6f\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));
	return 0;
}
