// oracle/ref_harness/ref_seed_main.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Known-answer vectors for seam B3 from the REFERENCE's own functions: deBGA_INDEX::load_index_file, ::search_kmer and
// ::UNITIG_MEM_search (src/PanSVgenerateVCF/deBGA_index.cpp:33-146, compiled from the reference tree by oracle/Makefile).
// For the first N reads of a FASTQ (forward strand as given, N bases read as A) every 20-mer offset is probed:
//   {"r":read,"off":offset,"kmer":K,"found":0|1,"range":[a,b],"mems":[[hit,uid,read_pos,uni_pos_off,length,pos_n,right_i],...]}
// plus one line per read with its packed words: {"r":read,"len":L,"words":[...]}.
// Usage: ref_seed <IndexDir> <reads.fq> <header.sam> <n_reads>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "PanSVgenerateVCF/read_realignment.hpp"

int main(int argc, char **argv)
{
	if (argc < 5) { fprintf(stderr, "usage: ref_seed <IndexDir> <reads.fq> <header.sam> <n_reads>\n"); return 1; }
	deBGA_INDEX *idx = (deBGA_INDEX *)xcalloc(1, sizeof(deBGA_INDEX));
	{
		FILE *h = xopen(argv[3], "r");
		std::string text;
		char *line = NULL; size_t cap = 0; ssize_t n;
		while ((n = getline(&line, &cap, h)) > 0) if (line[0] == '@') text.append(line, n);
		fclose(h);
		idx->ori_header = sam_hdr_parse((int)text.size(), text.c_str());
	}
	std::string dir = argv[1];
	if (dir.back() != '/') dir += '/';
	idx->load_index_file((char *)dir.c_str());
	const long n_reads = atol(argv[4]);
	gzFile fp = xzopen(argv[2], "rb");
	kstream_t *ks = ks_init(fp);
	kseq_t rd;
	memset(&rd, 0, sizeof rd);
	rd.f = ks;
	for (long r = 0; r < n_reads && kseq_read(&rd) >= 0; ++r) {
		const int L = (int)rd.seq.l;
		std::vector<uint64_t> w((size_t)L / 32 + 3, 0);
		for (int i = 0; i < L; ++i) {
			const char ch = rd.seq.s[i];
			const uint64_t code = (ch == 'C' || ch == 'c') ? 1 : (ch == 'G' || ch == 'g') ? 2 : (ch == 'T' || ch == 't') ? 3 : 0;
			w[(size_t)i >> 5] |= code << ((31 - (i & 31)) << 1);
		}
		printf("{\"r\":%ld,\"len\":%d,\"words\":[", r, L);
		for (size_t i = 0; i < w.size(); ++i) printf("%s%llu", i ? "," : "", (unsigned long long)w[i]);
		printf("]}\n");
		for (int off = 0; off + 20 <= L; ++off) {
			uint64_t kmer = 0;
			for (int j = 0; j < 20; ++j) kmer = (kmer << 2) | ((w[(size_t)(off + j) >> 5] >> ((31 - ((off + j) & 31)) << 1)) & 3);
			int64_t range[2] = {0, -1};
			const bool found = idx->search_kmer(20, kmer, range, 2);
			printf("{\"r\":%ld,\"off\":%d,\"kmer\":%llu,\"found\":%d,\"range\":[%lld,%lld],\"mems\":[", r, off, (unsigned long long)kmer, (int)found, found ? (long long)range[0] : 0ll,
			       found ? (long long)range[1] : -1ll);
			if (found && range[1] - range[0] + 1 <= 32) {
				for (int64_t hit = range[0]; hit <= range[1]; ++hit) {
					std::vector<vertex_MEM> v;
					uint32_t mri = 1;
					idx->UNITIG_MEM_search((uint64_t)hit, v, w.data(), (uint32_t)off, (uint32_t)L, 20, mri);
					printf("%s[%lld,%llu,%u,%u,%u,%u,%u]", hit > range[0] ? "," : "", (long long)hit, (unsigned long long)v[0].uid, v[0].read_pos, v[0].uni_pos_off, v[0].length, v[0].pos_n, mri);
				}
			}
			printf("]}\n");
		}
	}
	return 0;
}
