// oracle/ref_harness/ref_aln_main.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the REFERENCE's own `fc_aln` objects, compiled by oracle/Makefile from the sources where they lie
// under /root/reference/src (never copied): PanSVgenerateVCF/read_realignment.cpp, PanSVgenerateVCF/deBGA_index.cpp,
// cpp_lib/graph.cpp, clib/{binarys_qsort,bam_file,utils,kthread}.c, kswlib/*, and htslib's sam.c / kstring.c / hts.c.
// Everything on the path is the reference's code:
//   MAP_PARA::get_option             the option parser (-t -O -P -E -F -M -m -z -w -o -p -Q -S -R)      rr.hpp:82-128
//   sam_hdr_parse                    the original header (what sam_hdr_read does for SAM text)           htslib sam.c
//   deBGA_INDEX::load_index_file     incl. building_chr_index / building_bam_header (bam_name2id)        deBGA_index.cpp:33-80
//   deCOY_CLASSIFY_MAIN::load_reads  kseq_read over xzopen, STAT_ parsing, batching limits               rr.cpp:121-152
//   kt_for -> align_read_pair        incl. output_BAM / output_ori_bam -> sam_parse1 -> bam1_t           rr.cpp:745-803, kthread.c:61-86
//   sam_format1                      the text sam_write1 writes for every b with core.tid != -1          rr.cpp:165-176
// What this driver adds is init_run's glue (rr.cpp:26-108) minus hts_open: htslib's file layer (hts_open -> cram_open ->
// cram_io.c, which needs <lzma.h>) cannot be built in this image, so the driver reads header.sam itself, hands the '@' lines to
// sam_hdr_parse, and writes sam_format1's text with fputs.  -ffunction-sections + --gc-sections drops init_run and the
// htslib functions nothing here reaches.  No stand-in header or library is involved.
//
// Usage: ref_aln [fc_aln options] <IndexDir> <reads.fq> <header.sam> [--records FILE|-] [--trace] [--limit N] [--batch N] [--quiet]
//                [--stream-pos G,H0,H1]
//   --stream-pos       : (-t 1) the input is the continuation of a longer run: before the first pair, rand() is called until G draws
//                        have been made in total and handler k's random_r until Hk -- where a shard of a sharded run stands
//   -S -o FILE -p FILE : SAM text of the two output files (without -S nothing is formatted; BAM needs htslib's bgzf/hfile layer)
//   --records          : one JSON line per pair with what align_read_pair decided (default: stdout when no -S is given)
//   -t N               : the reference's own kt_for over N threads (timing only: output is non-deterministic for N > 1)
// stderr: "ALIGN_SECONDS s" = wall of the kt_for calls alone; "TOTAL_SECONDS s" = load_reads + kt_for + formatting.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include "PanSVgenerateVCF/read_realignment.hpp"
extern "C" {
#include "clib/kthread.h"
}

extern void align_read_pair(kseq_t *read1, kseq_t *read2, Classify_buff_pool *buff, bam1_t *b1, bam1_t *b2, bam1_t *b1_ori, bam1_t *b2_ori);

static uint64_t fnv(uint64_t h, uint64_t v)
{
	for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ULL; }
	return h;
}

static void print_result(FILE *f, MAX_IDX_OUTPUT &r, bool is_ori)
{
	fprintf(f, "[%u,%u,%d,%u,%u,%d,%d,\"", r.align_score, is_ori ? 0 : r.chain_score, (int)r.chrID, r.ref_bg, r.read_bg, r.direction, (int)r.mapq);
	for (auto &c : r.cigar) fprintf(f, "%d%c", c.size, OUT_BAM_CIGAR_STR[c.type]);
	fprintf(f, "\"]");
}

static int which(single_end_handler &h, MAX_IDX_OUTPUT *p)
{
	if (p == NULL) return -1;
	if (p == &h.ori) return -2;
	return (int)(p - h.result);
}

// the state the two handlers and the pairing object are left in by align_read_pair (valid at -t 1, right after the call)
static void print_record(FILE *f, long pair_i, Classify_buff_pool *buff, bool trace)
{
	single_end_handler *SE_h = &(buff->SE_h[0]);
	PE_score *ps = &(buff->ps);
	fprintf(f, "{\"i\":%ld,\"reads\":[", pair_i);
	for (int k = 0; k < 2; ++k) {
		single_end_handler &h = SE_h[k];
		fprintf(f, "%s{\"n\":%d,\"unmapped\":%d,\"res\":[", k ? "," : "", h.result_num, (int)h.ORI_is_UNMAPPED);
		for (int i = 0; i < h.result_num; ++i) { if (i) fprintf(f, ","); print_result(f, h.result[i], false); }
		fprintf(f, "],\"ori\":");
		print_result(f, h.ori, true);
		if (ps->pan_genome_gain_better_result) {
			MAX_IDX_OUTPUT *pr = h.primary_result;
			fprintf(f, ",\"prim\":%d,\"sec\":%d", which(h, pr), which(h, h.secondary_result));
			if (pr) fprintf(f, ",\"mate\":[%d,%u,%u]", (int)pr->has_mate, pr->has_mate ? pr->mate_chrID : 0, pr->has_mate ? pr->mate_ref_bg : 0);
		}
		if (trace) {
			fprintf(f, ",\"str\":%d,\"tr\":[", (int)h.readIsSTR);
			for (int s = 0; s < 2; ++s) {
				uint64_t hs = 1469598103934665603ULL, hd = 1469598103934665603ULL;
				auto &us = h.uniseed_v[s];
				for (auto &u : us) { hs = fnv(hs, u.read_begin); hs = fnv(hs, u.read_end); hs = fnv(hs, u.seed_id); hs = fnv(hs, u.ref_begin); hs = fnv(hs, u.ref_end); hs = fnv(hs, u.cov); }
				for (size_t i = 0; i < us.size(); ++i) { hd = fnv(hd, (uint64_t)(int64_t)h.g[s].dist_path[i].dist); hd = fnv(hd, (uint64_t)(int64_t)h.g[s].dist_path[i].pre_node); }
				fprintf(f, "%s[%zu,\"%016llx\",\"%016llx\"]", s ? "," : "", us.size(), (unsigned long long)hs, (unsigned long long)hd);
			}
			fprintf(f, "]");
		}
		fprintf(f, "}");
	}
	fprintf(f, "],\"pe\":[%d,%d,%d,%d,%d,%d]}\n", ps->max_score, ps->cur_isize, (int)ps->read_pair_is_proper_mated, (int)ps->pan_genome_gain_better_result,
	        which(SE_h[0], ps->max_1), which(SE_h[1], ps->max_2));
}

struct Batch {
	kseq_t *seqs1, *seqs2;
	bam1_t *b1, *b2, *ori_b1, *ori_b2;
	Classify_buff_pool *buff;
	FILE *frec; bool trace; long pair_base;
};

static void worker(void *data, long i, int tid)           // deCOY_CLASSIFY_MAIN::worker_for, rr.cpp:156-161
{
	Batch *d = (Batch *)data;
	align_read_pair(d->seqs1 + i, d->seqs2 + i, d->buff + tid, d->b1 + i, d->b2 + i, d->ori_b1 + i, d->ori_b2 + i);
	if (d->frec) print_record(d->frec, d->pair_base + i, d->buff + tid, d->trace);      // only offered at -t 1
}

static double now()
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
	// the driver's own switches are taken out; everything else is the reference parser's business
	std::vector<char *> av;
	const char *records = NULL;
	bool trace = false, quiet = false;
	long limit = -1, batch = 200000;
	long long spos[3] = {-1, -1, -1};
	av.push_back(argv[0]);
	av.push_back((char *)"fc_aln");                        // get_option is entered with argv pointing behind the sub-command (main.cpp:18-25)
	for (int a = 1; a < argc; ++a) {
		if (!strcmp(argv[a], "--trace")) trace = true;
		else if (!strcmp(argv[a], "--quiet")) quiet = true;
		else if (!strcmp(argv[a], "--limit") && a + 1 < argc) limit = atol(argv[++a]);
		else if (!strcmp(argv[a], "--batch") && a + 1 < argc) batch = atol(argv[++a]);
		else if (!strcmp(argv[a], "--records") && a + 1 < argc) records = argv[++a];
		else if (!strcmp(argv[a], "--stream-pos") && a + 1 < argc) sscanf(argv[++a], "%lld,%lld,%lld", &spos[0], &spos[1], &spos[2]);
		else av.push_back(argv[a]);
	}
	av.push_back(NULL);
	FILE *real_stderr = stderr;
	if (quiet) stderr = fopen("/dev/null", "w");           // the parser dumps its parameter table
	MAP_PARA *o = (MAP_PARA *)xcalloc(1, sizeof(MAP_PARA));
	if (o->get_option((int)av.size() - 2, av.data() + 1) != 0) return 1;
	if (limit >= 0 && limit < o->max_use_read) o->max_use_read = (int)limit;
	if (!o->output_sam && !records) records = "-";

	deBGA_INDEX *idx = (deBGA_INDEX *)xcalloc(1, sizeof(deBGA_INDEX));
	{   // init_run, rr.cpp:38-42, with the text handed to sam_hdr_parse directly (sam_hdr_read's SAM branch collects the '@' lines and does the same)
		FILE *h = xopen(o->ori_header_fn, "r");
		std::string text;
		char *line = NULL; size_t cap = 0; ssize_t n;
		while ((n = getline(&line, &cap, h)) > 0) if (line[0] == '@') text.append(line, n);
		fclose(h);
		bam_hdr_t *hdr = sam_hdr_parse((int)text.size(), text.c_str());
		hdr->l_text = text.size();
		hdr->text = strdup(text.c_str());
		idx->ori_header = hdr;
	}
	idx->load_index_file(o->indexDir);

	const int nt = o->thread_n;
	Classify_buff_pool *buff = (Classify_buff_pool *)xcalloc(nt, sizeof(Classify_buff_pool));     // rr.cpp:60-67
	for (int i = 0; i < nt; i++) {
		buff[i].ps.init(o->ISIZE_MAX, o->ISIZE_MIN, o->normal_read_length, 0);
		buff[i].SE_h[0].init(o, idx);
		buff[i].SE_h[1].init(o, idx);
	}
	if (spos[0] >= 0 && nt == 1) {                           // the two init() calls above made draws #0 and #1 of rand()
		for (long long k = 2; k < spos[0]; ++k) (void)rand();
		int32_t tmp;
		for (int h = 0; h < 2; ++h) for (long long k = 0; k < spos[1 + h]; ++k) random_r(&buff[0].SE_h[h].rand_buff, &tmp);
	}
	Batch B;
	B.seqs1 = (kseq_t *)xcalloc(batch, sizeof(kseq_t)), B.seqs2 = (kseq_t *)xcalloc(batch, sizeof(kseq_t));
	B.b1 = (bam1_t *)xcalloc(batch, sizeof(bam1_t)), B.b2 = (bam1_t *)xcalloc(batch, sizeof(bam1_t));
	B.ori_b1 = (bam1_t *)xcalloc(batch, sizeof(bam1_t)), B.ori_b2 = (bam1_t *)xcalloc(batch, sizeof(bam1_t));
	B.buff = buff, B.trace = trace, B.pair_base = 0;
	B.frec = NULL;
	if (records && nt == 1) B.frec = !strcmp(records, "-") ? stdout : xopen(records, "w");
	FILE *fo = NULL, *fo_ori = NULL;
	if (o->output_sam) {
		fo = xopen(o->sam_path, "w"), fo_ori = xopen(o->sam_path_signal_ori, "w");
		fputs(idx->ori_header->text, fo), fputs(idx->ori_header->text, fo_ori);       // sam_hdr_write's SAM branch: the header text as is
	}
	gzFile fp1 = xzopen(o->read_fastq1, "rb");
	kstream_t *ks = ks_init(fp1);
	kstring_t str = {0, 0, NULL};
	double t_align = 0, t_all = now();
	for (;;) {
		const int n = deCOY_CLASSIFY_MAIN::load_reads(ks, B.seqs1, B.seqs2, (int)batch, o, buff);
		if (n == 0) break;
		double t0 = now();
		kt_for(nt, worker, &B, n);
		t_align += now() - t0;
		if (fo) {                                           // output_results, rr.cpp:165-176, sam_write1's text branch
			for (int i = 0; i < n; i++) {
				if (B.b1[i].core.tid != -1) { sam_format1(idx->ori_header, B.b1 + i, &str); fputs(str.s, fo), fputc('\n', fo); }
				if (B.b2[i].core.tid != -1) { sam_format1(idx->ori_header, B.b2 + i, &str); fputs(str.s, fo), fputc('\n', fo); }
			}
			for (int i = 0; i < n; i++) {
				if (B.ori_b1[i].core.tid != -1) { sam_format1(idx->ori_header, B.ori_b1 + i, &str); fputs(str.s, fo_ori), fputc('\n', fo_ori); }
				if (B.ori_b2[i].core.tid != -1) { sam_format1(idx->ori_header, B.ori_b2 + i, &str); fputs(str.s, fo_ori), fputc('\n', fo_ori); }
			}
		}
		B.pair_base += n;
	}
	t_all = now() - t_all;
	if (fo) fclose(fo), fclose(fo_ori);
	if (B.frec && B.frec != stdout) fclose(B.frec);
	fflush(stdout);
	fprintf(real_stderr, "ALIGN_SECONDS %.6f\nTOTAL_SECONDS %.6f\nTHREADS %d\nPAIRS %ld\n", t_align, t_all, nt, B.pair_base);
	return 0;
}
