// oracle/ref_harness/ref_signal_main.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the REFERENCE's own per-pair function of the `signal` step, READ_SIGNAL_HANDLER::all_signal_records_read_pair
// (src/PanSVgenerateVCF/getSignalRead.cpp:100-256, with the helpers of clib/bam_file.c and htslib's aux accessors it calls:
// the signal filter, the scores, the FASTQ comment wire format, the strand handling of sequence and qualities), compiled from
// the sources where they lie by oracle/Makefile.  htslib's file layer cannot be built in this image (cram_io.c needs <lzma.h>),
// so the records do not come from sam_read1 on a BAM but from the reference's own sam_parse1 on the SAM text of the same records;
// the insert-size / read-length statistics that sampling_analysis_stat would take from the file are passed in (--stat).
// Usage: ref_signal <pairs.sam> <header.sam> --stat READLEN,MIN,MID,MAX [-D] [-U] [-I max_tid]  > reads.fq
//   pairs.sam: records in name order, mates adjacent (secondary / supplementary records are skipped as the reference does)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "PanSVgenerateVCF/getSignalRead.hpp"

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: ref_signal <pairs.sam> <header.sam> --stat L,MIN,MID,MAX [-D] [-U] [-I n]\n"); return 1; }
	int st[4] = {150, 100, 500, 900}, max_tid = MAX_TID;
	bool all = false, discard = false;
	for (int a = 3; a < argc; ++a) {
		if (!strcmp(argv[a], "--stat") && a + 1 < argc) sscanf(argv[++a], "%d,%d,%d,%d", &st[0], &st[1], &st[2], &st[3]);
		else if (!strcmp(argv[a], "-D")) all = true;
		else if (!strcmp(argv[a], "-U")) discard = true;
		else if (!strcmp(argv[a], "-I") && a + 1 < argc) max_tid = atoi(argv[++a]);
	}
	bam_hdr_t *hdr;
	{
		FILE *h = xopen(argv[2], "r");
		std::string text;
		char *line = NULL; size_t cap = 0; ssize_t n;
		while ((n = getline(&line, &cap, h)) > 0) if (line[0] == '@') text.append(line, n);
		fclose(h);
		hdr = sam_hdr_parse((int)text.size(), text.c_str());
	}
	READ_SIGNAL_HANDLER *H = new READ_SIGNAL_HANDLER();
	// init_run's option defaults (getSignalRead.hpp:240-262) and what it derives from the sampled statistics (:283-296)
	H->gap_open = GAP_OPEN, H->gap_ex = GAP_EXT, H->gap_open2 = GAP_OPEN2, H->gap_ex2 = GAP_EXT2, H->match = MATCH_SCORE, H->mismatch = MISMATCH_SCORE;
	H->maxTid = max_tid, H->NOT_USING_FILTER = all, H->discard_both_full_match = discard, H->sample_rate = 1, H->not_filter_low_quality = false;
	H->bs.init();
	H->bs.analysis_read_length = st[0], H->bs.minInsertLen = (uint32_t)st[1], H->bs.middleInsertLen = (uint32_t)st[2], H->bs.maxInsertLen = (uint32_t)st[3];
	H->isize_max = (int)H->bs.maxInsertLen + 150;
	H->isize_min = (int)H->bs.minInsertLen - 150;
	if (H->isize_min < 1) H->isize_min = 1;
	H->output_file1 = stdout, H->output_file2 = stdout;
	H->hdr = hdr;
	FILE *f = xopen(argv[1], "r");
	bam1_t *b[2] = {bam_init1(), bam_init1()};
	int have = 0;
	char *line = NULL; size_t cap = 0; ssize_t n;
	kstring_t ks = {0, 0, NULL};
	while ((n = getline(&line, &cap, f)) > 0) {
		if (line[0] == '@') continue;
		while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
		if ((size_t)n + 1 > ks.m) { ks.m = (size_t)n + 1024; ks.s = (char *)realloc(ks.s, ks.m); }
		memcpy(ks.s, line, (size_t)n + 1);
		ks.l = (size_t)n;
		if (sam_parse1(&ks, hdr, b[have]) != 0) { fprintf(stderr, "sam_parse1 failed on: %s\n", line); return 2; }
		if (bam_is_secondary(b[have]) || bam_is_supplementary(b[have])) continue;
		if (++have == 2) {
			H->all_signal_records_read_pair(*b[0], *b[1], true);
			have = 0;
		}
	}
	fflush(stdout);
	return 0;
}
