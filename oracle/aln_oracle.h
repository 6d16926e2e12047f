/*
 * oracle/aln_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's `fc_aln` per-pair work
 * (src/PanSVgenerateVCF/read_realignment.{cpp,hpp}, deBGA_index.{cpp,hpp}, cpp_lib/graph.{cpp,hpp},
 * clib/binarys_qsort.c) at `-t 1`.  Pinned by tests/test_oracle_aln.py against the records the
 * reference's own objects produce (oracle/_ref/ref_aln) and the committed tests/golden/ fixtures.
 * Nothing in the product path may include, link or call this.
 */
#ifndef PSVR_ALN_ORACLE_H_
#define PSVR_ALN_ORACLE_H_
#include <stdint.h>
#include <string>
#include <vector>

namespace orc {

// glibc TYPE_3 additive-feedback generator (rand(), and random_r on a 128-byte state), host independent
struct Rand3 {
	int32_t r[34];
	int f, b;     // front / rear indices into the 31-word ring
	int32_t ring[31];
	void seed(unsigned s);
	int32_t next();
};

struct SvInfo { // SV_chr_info, deBGA_index.hpp:74-155
	uint32_t ID, chr_ID;
	uint64_t st_pos;
	int region_len, end_offset;
	std::string sv_type, vcf_id, vcf_print_string;
};

struct Index { // deBGA_INDEX, deBGA_index.hpp:157-219; the 2 GiB prefix-sum hash is kept sparse
	std::vector<uint64_t> ref_seq, seq, seqf, pos, posp, off_g;
	std::vector<uint32_t> kmer_g;
	std::vector<uint32_t> bucket_id;    // non-empty first-level buckets, ascending
	std::vector<uint64_t> bucket_start; // hash[bucket_id[i]]
	uint64_t n_kmer = 0;
	std::vector<std::string> chr_names;
	std::vector<uint32_t> chr_end_n;
	std::vector<uint32_t> chr_search_index;
	int chr_file_n = 0;
	uint64_t reference_len = 0;
	std::vector<SvInfo> sv_info;
	uint64_t hash_at(uint64_t h) const;
	bool load(const std::string &dir, const std::vector<std::string> &header_names, std::string *err);
	int get_chromosome_ID(uint32_t position) const;
	void get_refseq(uint8_t *ref, uint32_t len, uint32_t start) const;
};

struct Cigar { uint8_t type; int16_t size; };

struct Result { // MAX_IDX_OUTPUT, read_realignment.hpp:243-319
	uint32_t align_score = 0, chain_score = 0, max_index = 0, read_bg = 0;
	int sv_id = -1;          // index into Index::sv_info, -1 = none
	uint8_t mapq = 0;
	bool has_mate = false;
	uint32_t mate_chrID = 0, mate_ref_bg = 0;
	int mate_sv_id = -1;
	bool is_ori = false;
	uint32_t chrID = 0, ref_bg = 0;
	int direction = 0;
	std::vector<Cigar> cigar;
	int rst_idx = 0;
};

struct Params {
	int match = 2, mismatch = 12, gap_open = 16, gap_ex = 1, gap_open2 = 32, gap_ex2 = 0, zdrop = 400;
	int normal_read_length = 150, isize_min = 100, isize_mid = 500, isize_max = 900;
	bool stat_set = false;
};

struct Counters { // algorithmic-byte accounting of SURVEY 8(d)
	uint64_t reads = 0, probes = 0, probe_bytes = 0, hits = 0, hit_bytes = 0, seeds = 0, pos_bytes = 0;
	uint64_t dp_calls = 0, simple_calls = 0, ref_bytes = 0, dp_out_bytes = 0, dp_cells = 0, cand_bytes = 0, read_bytes = 0;
	uint64_t total() const { return read_bytes + probe_bytes + hit_bytes + pos_bytes + ref_bytes + dp_out_bytes + cand_bytes; }
};

struct Read {
	std::string name, comment, seq, qual;
};

struct Aligner;
Aligner *aligner_create(const Index *idx, const Params *par = nullptr);
void aligner_destroy(Aligner *);
// processes one pair exactly as align_read_pair does (read_realignment.cpp:750-767) and returns the
// record line the reference harness prints for it
std::string aligner_pair(Aligner *, Read &r1, Read &r2, long pair_i, bool trace);
const Counters &aligner_counters(const Aligner *);

bool read_header_names(const std::string &path, std::vector<std::string> *names);
bool read_fastq_record(FILE *f, Read *r);

} // namespace orc
#endif
