// signal_step.h -- `panSVR signal` (= the reference's `fc_signal`): BAM -> interleaved FASTQ whose comment carries
// the original alignment, i.e. the wire format `aln` parses.  Host C++ only (the work is record parsing; nothing here runs on the
// GPU).  Restated from PanSVgenerateVCF/getSignalRead.cpp / .hpp and the helpers of clib/bam_file.c it calls; PARITY UNPINNED:
// the reference's build of this step needs htslib, which cannot be built in this image, so the only checker is the independent
// restatement in oracle/signal_oracle.py (tests/test_signal.py).
//
// Both input orders are implemented: name-sorted (`-N`, SURVIVOR_SV_region_get_all_signal_records_SORT_BY_NAME, getSignalRead.cpp:491-519)
// and the default, position-sorted one (..._SORT_BY_pos, :285-489: blocks of up to 1 M primary records / one chromosome / 60 Mbp, mates
// found through a 64 bp position index, the records left over -- mate on another chromosome or in another block, and always the last
// record of a block -- paired by name afterwards).  Two deliberate differences in that mode: the left-over records are kept in memory
// instead of going through the temporary BAM (the reference reads that file back into memory in one piece anyway; `-t` is accepted and
// unused), and the insert-size quantiles of the STAT_ field come from the first 100 000 primary records like in the name-sorted mode,
// not from the Manta-derived genome-wide sampler of cpp_lib/statistics (`--isize MIN,MID,MAX` overrides them, e.g. with the
// reference's numbers).  A mate the index search cannot find is left to the by-name pass; the reference aborts there.
// Where the reference reads uninitialised memory this code uses 0 (soft-clip lengths of a record without CIGAR,
// getSignalRead.cpp:129 with clib/bam_file.c:1033-1034; the sampled depth in the status file, getSignalRead.hpp:170).
#pragma once
#include <getopt.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "bam_reader.h"
#include "fastq_batch.h"

namespace psvr {

struct SignalOpt {
	int gap_open = 16, gap_ex = 1, gap_open2 = 32, gap_ex2 = 0, match = 2, mismatch = 12;   // getSignalRead.hpp:19-24
	int max_tid = 24;                                                                       // MAX_TID
	bool sort_by_name = false, not_use_filter = false, discard_full_match = false;
	double sample_rate = 1;
	std::string header_fn = "./header.sam", status_fn = "./status.sam", input;
	int isize_override[3] = {-1, -1, -1};                                                   // --isize MIN,MID,MAX
};

struct BamStat {                                            // BAM_STAT, getSignalRead.hpp:33-190
	static const int kMaxIsize = 100000, kMaxLen = 1000;
	std::vector<uint64_t> isize_n = std::vector<uint64_t>(kMaxIsize, 0), len_n = std::vector<uint64_t>(kMaxLen, 0);
	uint64_t total = 0;
	int read_len = -1;
	double normal_percent = 0, ave_len = 0, depth = 0;
	uint32_t min_l2 = 0, max_l2 = 0, min_i = 0, mid_i = 0, max_i = 0;
	std::vector<float> dist;
	uint32_t reason_n[1024] = {0};
	void reset() { total = 0; std::fill(isize_n.begin(), isize_n.end(), 0); std::fill(len_n.begin(), len_n.end(), 0); memset(reason_n, 0, sizeof reason_n); }
	void collect(const BamRecord &r)                        // collect_signal, :76-86
	{
		const int is = r.isize < 0 ? -r.isize : r.isize;
		if (is > 0 && is < kMaxIsize) isize_n[is]++;
		if (r.l_qseq < kMaxLen) len_n[r.l_qseq]++;
	}
	void global()                                           // global_analysis_stat, :88-124
	{
		read_len = -1;
		double tot_len = 0;
		if (total == 0) { normal_percent = 0, ave_len = 0, read_len = 0, min_l2 = max_l2 = 0; return; }   // an empty BAM: the reference divides by zero here
		for (int i = 0; i < kMaxLen; ++i) {
			tot_len += (double)i * (double)len_n[i];
			if ((double)len_n[i] > 0.6 * (double)total) { read_len = i, normal_percent = (double)len_n[i] / (double)total; break; }
		}
		ave_len = tot_len / (double)total;
		if (read_len == -1) read_len = (int)ave_len;
		min_l2 = max_l2 = 0;
		const float pct = 0.01f;
		const int lim = (int)(pct * (float)total);
		int sum = 0;
		for (int i = 0; i < kMaxIsize; ++i) { sum += (int)isize_n[i]; if (sum > lim) { min_l2 = (uint32_t)i; break; } }
		sum = 0;
		for (int i = kMaxIsize - 1; i > 0; --i) { sum += (int)isize_n[i]; if (sum > lim) { max_l2 = (uint32_t)i; break; } }
	}
	void final_stat(FILE *o) const                          // output_final_stat, :184-189
	{
		fprintf(o, "%f_%d_%d_%d_%d_%d\n", depth, read_len, min_l2, max_l2, min_i, max_i);
		for (uint32_t i = min_i; i < max_i; ++i) fprintf(o, "%f\n", dist[i - min_i]);
	}
};

inline char signal_rev_char(char c)                         // getReverseChar, clib/bam_file.c:320-328
{
	switch (c) {
	case 'A': case 'a': return 'T';
	case 'C': case 'c': return 'G';
	case 'G': case 'g': return 'C';
	case 'T': case 't': return 'A';
	}
	return 'N';
}

struct SignalStep {
	SignalOpt o;
	BamStat bs;
	int isize_min = 1, isize_max = 0;
	bool stat_written = false;
	int sample_max = 0;
	FILE *out = stdout;
	PairFeed *feed = nullptr;          // fused with `aln`: records go to the batch being built instead of FASTQ text on `out`

	static bool primary(const BamRecord &r) { return !(r.flag & 0x100) && !(r.flag & 0x800); }

	int score_by_cigar(const BamRecord &b) const            // getScoreByCigar, getSignalRead.cpp:36-77
	{
		int score = 0, gap_len = 0;
		for (unsigned i = 0; i < b.n_cigar; ++i) {
			const int op = (int)(b.cig(i) & 0xf), len = (int)(b.cig(i) >> 4);
			if (op == 0 || op == 7) score += len * o.match;                    // M, =
			else if (op == 1 || op == 2 || op == 4 || op == 5) {               // I, D, S, H
				if (op == 1 || op == 2) gap_len += len;
				const int p1 = o.gap_open + len * o.gap_ex, p2 = o.gap_open2 + len * o.gap_ex2;
				score -= p1 < p2 ? p1 : p2;
			}
		}
		int32_t nm = 0;
		b.num_tag("NM", &nm);
		score -= (o.mismatch + o.match) * (nm - gap_len);
		return score > 0 ? score : 0;
	}
	static int xa_number(const BamRecord &b)                // get_XA_number, :81-92
	{
		if (b.mapq > 0) return 0;
		const char *xa = b.string_tag("XA");
		if (!xa) return 6;
		int n = 0;
		for (const char *p = xa; *p; ++p) n += *p == ';';
		return n;
	}
	// bam2fastqWrite_additional_str_gz, :15-34
	void write_fastq(const BamRecord &b, const std::string &comment, const psvr_ori_t &ori) const
	{
		std::string seq, qual((size_t)b.l_qseq, '\0');
		const uint8_t *s4 = b.seq(), *q = b.qual();
		for (int i = 0; i < b.l_qseq; ++i) {
			const int c = (s4[i >> 1] >> ((~i & 1) << 2)) & 0xf;
			switch (c) {
			case 1: seq += 'A'; break;
			case 2: seq += 'C'; break;
			case 4: seq += 'G'; break;
			case 8: seq += 'T'; break;
			case 15: seq += 'N'; break;
			default: fprintf(stderr, "Wrong base!");                           // get_bam_seq emits nothing for the other codes
			}
			qual[(size_t)i] = (char)(uint8_t)(q[i] + 33);
		}
		if (!(b.flag & 0x4) && (b.flag & 0x10)) {
			// getReverseStr_char over read_len (= l_qseq) and getReverseStr_qual with its len/2 + 1 bound (the two middle entries of an
			// even-length string are swapped twice), clib/bam_file.c:330-350
			const int len = b.l_qseq, half = len >> 1;
			if ((int)seq.size() == len) {
				for (int i = 0; i < half; ++i) { const char t = seq[(size_t)i]; seq[(size_t)i] = signal_rev_char(seq[(size_t)(len - 1 - i)]); seq[(size_t)(len - 1 - i)] = signal_rev_char(t); }
				if (len & 1) seq[(size_t)half] = signal_rev_char(seq[(size_t)half]);
			}
			for (int i = 0; i < half + 1 && len > 0; ++i) { const int ri = len - 1 - i; std::swap(qual[(size_t)i], qual[(size_t)ri]); }
		}
		if (feed) feed->put(b.qname(), comment, seq, qual, ori);
		else fprintf(out, "@%s %s\n%s\n+\n%s\n", b.qname(), comment.c_str(), seq.c_str(), qual.c_str());
	}

	// all_signal_records_read_pair, :100-256
	void pair(const BamRecord &r1, const BamRecord &r2, bool used)
	{
		bs.collect(r1), bs.collect(r2);
		if (!used) return;
		const BamRecord *b[2] = {&r1, &r2};
		bool unmapped[2], direction[2];
		int mapq[2], isize_read[2], lowq[2], soft_l[2] = {0, 0}, soft_r[2] = {0, 0}, clip[2], indel_nm[2], score[2], xa[2], tid[2];
		for (int i = 0; i < 2; ++i) {
			const BamRecord &x = *b[i];
			unmapped[i] = (x.flag & 0x4) != 0, mapq[i] = x.mapq, isize_read[i] = x.isize, direction[i] = !(x.flag & 0x10);
			lowq[i] = 0;
			for (int k = 0; k < x.l_qseq && k < 100000; ++k) lowq[i] += x.qual()[k] < (uint8_t)'/';      // get_bam_low_quality_num(0, 100000, '/')
			if (x.n_cigar) {                                                                            // bam_has_SH_cigar, clib/bam_file.c:1031-1053
				const uint32_t f = x.cig(0), l = x.cig(x.n_cigar - 1u);
				if ((f & 0xf) == 4 || (f & 0xf) == 5) soft_l[i] = (int)(f >> 4);
				if ((l & 0xf) == 4 || (l & 0xf) == 5) soft_r[i] = (int)(l >> 4);
			}
			clip[i] = soft_l[i] + soft_r[i];
			int indel = 0;                                                                              // bam_has_INDEL_NM, :1056-1069
			for (unsigned k = 0; k < x.n_cigar; ++k) if ((x.cig(k) & 0xf) == 1 || (x.cig(k) & 0xf) == 2) indel += (int)(x.cig(k) >> 4);
			int32_t nm = 0;
			x.num_tag("NM", &nm);
			indel_nm[i] = indel + nm;
			score[i] = score_by_cigar(x), tid[i] = x.tid, xa[i] = xa_number(x);
		}
		const int isize = isize_read[0] < 0 ? -isize_read[0] : isize_read[0];
		if (o.discard_full_match) {
			const int min_score = (b[0]->l_qseq + b[1]->l_qseq) * o.match - 4 * (o.match + o.mismatch);
			const bool near_full = score[0] + score[1] >= min_score, isize_ok = isize != 0 && isize > isize_min && isize < isize_max;
			if (near_full && isize_ok && tid[0] == tid[1] && tid[0] <= o.max_tid && tid[1] <= o.max_tid) return;
		}
		if (isize_read[0] + isize_read[1] != 0 && isize_read[0] != isize_read[1]) fprintf(stderr, " wrong ISIZE: %d  %d \n", isize_read[0], isize_read[1]);
		if (b[0]->pos > b[1]->pos) std::swap(direction[0], direction[1]);
		if (isize == b[0]->l_qseq && isize == b[1]->l_qseq && direction[0] == false && direction[1] == true) std::swap(direction[0], direction[1]);
		std::string reason[2];
		char buf[256];
		for (int i = 0; i < 2; ++i) {
			snprintf(buf, sizeof buf, "%d_%d_%d_%d_%d_%d_%d_%d_%d_", tid[i], b[i]->pos, soft_l[i], score[i], mapq[i], mapq[1 - i], xa[i], xa[1 - i], isize);
			reason[i] = buf;
		}
		char flags[2][5];
		for (int i = 0; i < 2; ++i) {
			flags[i][0] = !(b[i]->flag & 0x10) ? 'F' : 'R', flags[i][1] = unmapped[i] ? 'Y' : 'N';
			flags[i][2] = indel_nm[i] > 8 ? 'Y' : 'N', flags[i][3] = clip[i] > 10 ? 'Y' : 'N', flags[i][4] = 0;
		}
		for (int i = 0; i < 2; ++i) reason[i] += std::string(flags[i]) + "_" + flags[1 - i] + "_";
		uint32_t reason_flag = 0;
		bool pass = true;
		for (int i = 0; i < 2; ++i) {
			clip[i] -= lowq[i];
			if (clip[i] < 0) lowq[i] = -clip[i], clip[i] = 0;
			lowq[i] >>= 1;                                                                              // 1 NM or INDEL in 2 low-quality bases
			indel_nm[i] -= lowq[i];
			if (indel_nm[i] < 0) lowq[i] = -indel_nm[i], indel_nm[i] = 0;
		}
		if (mapq[0] < 10 && mapq[1] < 10) pass = false, reason_flag += 1;
		if (unmapped[0] || unmapped[1]) pass = false, reason_flag += 2;
		if (isize > 1000) pass = false, reason_flag += 4;
		if (direction[0] != true || direction[1] != false) pass = false, reason_flag += 8;
		if (indel_nm[0] + indel_nm[1] > 15) pass = false, reason_flag += 16;
		if (clip[0] + clip[1] > 10) pass = false, reason_flag += 32;
		if (tid[0] != tid[1] || tid[0] > o.max_tid || tid[1] > o.max_tid) pass = false, reason_flag += 64;
		if (pass && !o.not_use_filter) return;
		if (!(b[0]->flag & 0x40) || !(b[1]->flag & 0x80)) { fprintf(stderr, "[panSVR-amd] signal: records of a pair are not first/second in template\n"); abort(); }   // xassert, :195-196
		bs.reason_n[reason_flag]++;
		if (!(b[0]->l_qseq < 2048)) { fprintf(stderr, "[panSVR-amd] signal: read longer than 2047 bases\n"); abort(); }                                                // xassert, :199
		if (!stat_written) {
			snprintf(buf, sizeof buf, "STAT_%d_%d_%d_%d_", bs.read_len, bs.min_i, bs.mid_i, bs.max_i);
			reason[0] += buf;
			stat_written = true;
		}
		static const char *tags_z[3] = {"XA", "MC", "SA"};
		for (int i = 0; i < 2; ++i) {
			const BamRecord &x = *b[i];
			snprintf(buf, sizeof buf, "FLAG_%d_%d_CIGAR_", (int)x.flag, (int)x.mapq);
			reason[i] += buf;
			for (unsigned k = 0; k < x.n_cigar; ++k) { snprintf(buf, sizeof buf, "%d%c", (int)(x.cig(k) >> 4), "MIDNSHP=XB"[x.cig(k) & 0xf]); reason[i] += buf; }
			reason[i] += "_";
			snprintf(buf, sizeof buf, "MATE_%d_%d_%d_", x.mtid, x.mpos, x.isize);
			reason[i] += buf;
			reason[i] += "TAG_";
			for (int t = 0; t < 3; ++t) {
				const char *v = x.string_tag(tags_z[t]);
				if (v) reason[i] += std::string(tags_z[t]) + ":Z:" + v + "_";
			}
			int32_t nm = 0;
			if (x.num_tag("NM", &nm)) { snprintf(buf, sizeof buf, "NM:i:%d_", nm); reason[i] += buf; }
		}
		// the original alignment as `aln` reads it out of the comment (parse_ori_mapping_rst, rr.hpp:392-429: tokens 0-4 and the flag token)
		psvr_ori_t ori[2];
		for (int i = 0; i < 2; ++i) {
			memset(&ori[i], 0, sizeof ori[i]);
			ori[i].chr_id = tid[i], ori[i].ref_bg = (uint32_t)b[i]->pos, ori[i].read_bg = (uint32_t)soft_l[i], ori[i].align_score = (uint32_t)score[i], ori[i].mapq = (uint8_t)mapq[i];
			ori[i].direction = flags[i][0] == 'F', ori[i].unmapped = flags[i][1] == 'Y';
		}
		write_fastq(*b[0], reason[0], ori[0]);
		write_fastq(*b[1], reason[1], ori[1]);
	}

	// sampling_analysis_stat with bam_sort_by_name (getSignalRead.hpp:126-181): the first 100 000 primary records
	bool sample_stats()
	{
		BamReader rd;
		if (!rd.open(o.input.c_str())) { fprintf(stderr, "[panSVR-amd] signal: %s\n", rd.error().c_str()); return false; }
		bs.reset();
		BamRecord r;
		for (;;) {
			bool got;
			do { got = rd.next(r); } while (got && !primary(r));
			if (!got) break;
			bs.total++;
			if (bs.total == 100000) break;
			bs.collect(r);
		}
		if (!rd.error().empty()) { fprintf(stderr, "[panSVR-amd] signal: %s\n", rd.error().c_str()); return false; }
		bs.global();
		bs.min_i = bs.min_l2, bs.mid_i = (bs.min_l2 + bs.max_l2) / 2, bs.max_i = bs.max_l2;
		if (o.isize_override[0] >= 0) bs.min_i = (uint32_t)o.isize_override[0], bs.mid_i = (uint32_t)o.isize_override[1], bs.max_i = (uint32_t)o.isize_override[2];
		bs.dist.clear();
		for (uint32_t i = bs.min_i; i < bs.max_i; ++i) bs.dist.push_back((float)bs.isize_n[i] / (float)(bs.total + 1));
		bs.depth = 0;
		fprintf(stderr, "BAM/CRAM status: read length: [Normal: %d @ %f%%, AVE: %f] ISIZE: [MIN: %d MIDDLE:%d MAX: %d] ave_read_depth [%f]\n", bs.read_len, bs.normal_percent * 100, bs.ave_len,
		        bs.min_i, bs.mid_i, bs.max_i, bs.depth);
		return true;
	}

	// SURVIVOR_SV_region_get_all_signal_records_SORT_BY_pos, getSignalRead.cpp:285-489
	bool run_pos_sorted(BamReader &rd)
	{
		static const int kBuf = 1000000, kStep = 64, kIndex = 1000000, kRegion = 60000000;   // SAM_LOAD_BUFF_SIZE, SEARCH_STEP, SEARCH_POS_INDEX_SIZE, SEARCH_REGION_MAX
		std::vector<BamRecord> buf, left;                        // the block; the records no mate was found for
		std::vector<uint32_t> mate, index((size_t)kIndex);
		const uint32_t none = 0xffffffffu;
		long long unmated_warned = 0;
		bool eof = false;
		for (;;) {
			// part 1: a block of primary records -- until the chromosome changes, 60 Mbp are spanned or the buffer is full
			buf.clear();
			while ((int)buf.size() < kBuf && !eof) {
				BamRecord r;
				if (!rd.next(r)) { eof = true; break; }
				if (!primary(r)) continue;
				buf.push_back(std::move(r));
				if (buf.back().tid != buf[0].tid) break;
				if (buf.back().pos - buf[0].pos > kRegion) break;
			}
			const int n = (int)buf.size();
			if (n < 2) break;                                     // (a single left-over record is dropped, like in the reference)
			// part 2: position index in steps of 64 bp, then the mate of every record but the last
			mate.assign((size_t)n, none);
			int index_n = 0;
			const int32_t start_pos = buf[0].pos, final_pos = buf[(size_t)n - 2].pos;
			for (int i = 0; i < n; ++i) {
				const int pi = (buf[(size_t)i].pos - start_pos) / kStep;
				while (pi >= index_n) {
					if (!(index_n < kIndex - 2)) { fprintf(stderr, "[panSVR-amd] signal: position index overflow (is the input sorted by position?)\n"); abort(); }   // xassert, :327
					index[(size_t)index_n++] = (uint32_t)i;
				}
			}
			index[(size_t)index_n] = (uint32_t)n;
			for (int i = 0; i < n - 1; ++i) {
				if (mate[(size_t)i] != none) continue;
				const BamRecord &c = buf[(size_t)i];
				if (c.tid != c.mtid) continue;
				int m = -1;
				if (c.tid == -1) {                                 // both unmapped: the mate is a neighbour
					if (i == 0 || i > n - 2) continue;
					if (!strcmp(buf[(size_t)i + 1].qname(), c.qname())) m = i + 1;
					else if (!strcmp(buf[(size_t)i - 1].qname(), c.qname())) m = i - 1;
				} else {
					const int mpos = c.mpos;
					if (mpos <= start_pos || mpos >= final_pos) continue;
					const int pi = (mpos - start_pos) / kStep;
					for (uint32_t k = index[(size_t)pi]; k < index[(size_t)pi + 1]; ++k) {
						const BamRecord &t = buf[k];
						if (t.pos < mpos) continue;
						if (t.pos > mpos) break;
						if (t.mpos != c.pos) continue;
						if ((int)k != i && !strcmp(t.qname(), c.qname())) { m = (int)k; break; }
					}
				}
				if (m < 0 || mate[(size_t)m] != none) { if (++unmated_warned % 1000 == 0) fprintf(stderr, "NUM: [%lld]: Mate failed, index:[%d] in [%d] size block, tid: [ %d ], pos [%d] name [%s]\n", unmated_warned, i, n, c.tid, c.pos, c.qname()); continue; }
				mate[(size_t)i] = (uint32_t)m, mate[(size_t)m] = (uint32_t)i;
			}
			// part 3: what found no mate waits for the by-name pass
			int unmated = 0;
			for (int i = 0; i < n; ++i) if (mate[(size_t)i] == none) ++unmated;
			if ((long long)unmated * 20 > n) fprintf(stderr, "WARNING, too many total_unmated_read![ %d %d %f]\n", unmated, n, (float)unmated / (float)n);
			// part 4: the pairs, in the order of their first reads
			for (int i = 0; i < n - 1; ++i) {
				if (mate[(size_t)i] == none) continue;
				const BamRecord &b1 = buf[(size_t)i];
				if (b1.flag & 0x80) continue;
				const BamRecord &b2 = buf[mate[(size_t)i]];
				bs.total += 2;
				if (bs.total % 100000 == 0) fprintf(stderr, "%ld\n", (long)bs.total);
				if (!(b1.flag & 0x40) || !(b2.flag & 0x80)) { fprintf(stderr, "[panSVR-amd] signal: records of [%s] are not first/second in template\n", b1.qname()); abort(); }
				bool used = true;
				if (o.sample_rate < 0.9999 && rand() > sample_max) used = false;
				pair(b1, b2, used);
			}
			for (int i = 0; i < n; ++i) if (mate[(size_t)i] == none) left.push_back(std::move(buf[(size_t)i]));
		}
		if (!rd.error().empty()) return false;
		// phase 2: the left-over records by name, first before second (sam_cmp_by_name, :3-12; glibc's qsort is a stable merge sort)
		fprintf(stderr, "phase 2:\nSort all unpaired records: [%zu] in total\n", left.size());
		if (left.size() % 2) { fprintf(stderr, "[panSVR-amd] signal: an odd number of records found no mate: the BAM is incomplete\n"); abort(); }   // xassert, :437
		std::vector<uint32_t> ord(left.size());
		for (size_t i = 0; i < ord.size(); ++i) ord[i] = (uint32_t)i;
		std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
			const int c = strcmp(left[a].qname(), left[b].qname());
			if (c) return c < 0;
			return (left[a].flag & 0x40) != 0 && !(left[b].flag & 0x40);
		});
		for (size_t i = 0; i + 1 < ord.size(); i += 2) {
			bs.total += 2;
			if (bs.total % 100000 == 0) fprintf(stderr, "%ld\n", (long)bs.total);
			const BamRecord &b1 = left[ord[i]], &b2 = left[ord[i + 1]];
			if (strcmp(b1.qname(), b2.qname())) {
				fprintf(stderr, "Read Pairing error!, reads names are not same, please confirm the completeness of BAM/CRAM file.\n");
				--i;                                               // (the reference steps on by one record)
				continue;
			}
			if (!(b1.flag & 0x40) || !(b2.flag & 0x80)) { fprintf(stderr, "[panSVR-amd] signal: records of [%s] are not first/second in template\n", b1.qname()); abort(); }
			bool used = true;
			if (o.sample_rate < 0.9999 && rand() > sample_max) used = false;
			pair(b1, b2, used);
		}
		return true;
	}

	int run()                                               // init_run, getSignalRead.hpp:228-330
	{
		if (!sample_stats()) return 1;
		bs.final_stat(stderr);
		FILE *st = fopen(o.status_fn.c_str(), "w");
		if (!st) { fprintf(stderr, "fail to open file '%s'\n", o.status_fn.c_str()); return 1; }
		bs.depth *= o.sample_rate;
		bs.final_stat(st);
		fclose(st);
		bs.reset();
		isize_max = (int)(bs.max_i + 150);
		isize_min = (int)(bs.min_i - 150);                  // unsigned arithmetic in the reference, then `< 1 -> 1`
		if (isize_min < 1) isize_min = 1;
		if (o.sample_rate < 0.9999) {
			sample_max = (int)(o.sample_rate * RAND_MAX);
			fprintf(stderr, "Sample_rate: [%f] sample_max_number_int: [%d]\n", o.sample_rate, sample_max);
		}
		BamReader rd;
		if (!rd.open(o.input.c_str())) { fprintf(stderr, "[panSVR-amd] signal: %s\n", rd.error().c_str()); return 1; }
		if (!o.header_fn.empty()) {                         // sam_hdr_write in SAM text mode: the header text as stored
			FILE *h = fopen(o.header_fn.c_str(), "w");
			if (!h) { fprintf(stderr, "fail to open file '%s'\n", o.header_fn.c_str()); return 1; }
			fwrite(rd.header_text.data(), 1, rd.header_text.size(), h);
			fclose(h);
		}
		if (!o.sort_by_name) {
			if (!run_pos_sorted(rd)) { fprintf(stderr, "[panSVR-amd] signal: %s\n", rd.error().c_str()); return 1; }
			fflush(out);
			bs.global();
			fprintf(stderr, "BAM/CRAM status: ave_read_depth: [%f] read length: [Normal: %d @ %f%%, AVE: %f] ISIZE: [MIN: %d MAX: %d]\n", bs.depth, bs.read_len, bs.normal_percent * 100, bs.ave_len,
			        bs.min_l2, bs.max_l2);
			return 0;
		}
		BamRecord b1, b2;
		for (;;) {                                          // SURVIVOR_SV_region_get_all_signal_records_SORT_BY_NAME
			bool g1, g2;
			do { g1 = rd.next(b1); } while (g1 && !primary(b1));
			do { g2 = rd.next(b2); } while (g2 && !primary(b2));
			if (!g1 || !g2) break;
			if (strcmp(b1.qname(), b2.qname())) { fprintf(stderr, "[panSVR-amd] signal: consecutive records [%s] [%s] are not a pair: is the input sorted by name?\n", b1.qname(), b2.qname()); abort(); }
			if (!(b1.flag & 0x40) || !(b2.flag & 0x80)) { fprintf(stderr, "[panSVR-amd] signal: records of [%s] are not first/second in template\n", b1.qname()); abort(); }
			bs.total += 2;
			if (bs.total % 100000 == 0) fprintf(stderr, "%ld\r", (long)bs.total);
			bool used = true;
			if (o.sample_rate < 0.9999 && rand() > sample_max) used = false;   // signal_be_used, :268-274
			pair(b1, b2, used);
		}
		if (!rd.error().empty()) { fprintf(stderr, "[panSVR-amd] signal: %s\n", rd.error().c_str()); return 1; }
		fflush(out);
		bs.global();
		fprintf(stderr, "BAM/CRAM status: ave_read_depth: [%f] read length: [Normal: %d @ %f%%, AVE: %f] ISIZE: [MIN: %d MAX: %d]\n", bs.depth, bs.read_len, bs.normal_percent * 100, bs.ave_len,
		        bs.min_l2, bs.max_l2);
		return 0;
	}
};

inline int signal_main(int argc, char **argv)
{
	SignalStep S;
	static struct option lo[] = {{"gap-open1", 1, 0, 'O'}, {"gap-open2", 1, 0, 'P'}, {"gap-extension1", 1, 0, 'E'}, {"gap-extension2", 1, 0, 'F'}, {"match-score", 1, 0, 'M'},
	                             {"mis-score", 1, 0, 'm'}, {"max-tid-filter", 1, 0, 'I'}, {"sort-by-name", 0, 0, 'N'}, {"not-ignore-low-q", 0, 0, 'L'}, {"reference", 1, 0, 'r'},
	                             {"header-file", 1, 0, 'H'}, {"status-file", 1, 0, 'S'}, {"tmp_file_pairing", 1, 0, 't'}, {"not-use-filter", 0, 0, 'D'}, {"discard-full-match", 0, 0, 'U'},
	                             {"sample-rate", 1, 0, 'R'}, {"isize", 1, 0, 1000}, {0, 0, 0, 0}};
	int c;
	optind = 2;
	while ((c = getopt_long(argc, argv, "O:P:E:F:M:m:I:NLr:H:S:t:DUR:", lo, NULL)) >= 0) {
		switch (c) {
		case 'O': S.o.gap_open = atoi(optarg); break;
		case 'P': S.o.gap_open2 = atoi(optarg); break;
		case 'E': S.o.gap_ex = atoi(optarg); break;
		case 'F': S.o.gap_ex2 = atoi(optarg); break;
		case 'M': S.o.match = atoi(optarg); break;
		case 'm': S.o.mismatch = atoi(optarg); break;
		case 'I': S.o.max_tid = atoi(optarg); break;
		case 'N': S.o.sort_by_name = true; break;
		case 'L': case 'r': case 't': break;                // accepted and unused (-t: the left-over records stay in memory)
		case 'H': S.o.header_fn = optarg; break;
		case 'S': S.o.status_fn = optarg; break;
		case 'D': S.o.not_use_filter = true; break;
		case 'U': S.o.discard_full_match = true; break;
		case 'R': S.o.sample_rate = atof(optarg); break;
		case 1000: if (sscanf(optarg, "%d,%d,%d", &S.o.isize_override[0], &S.o.isize_override[1], &S.o.isize_override[2]) != 3) { fprintf(stderr, "--isize wants MIN,MID,MAX\n"); return 1; } break;
		default: fprintf(stderr, "usage: panSVR signal|fc_signal [-N] [options] <in.bam>  > reads.fq\n"); return 1;
		}
	}
	if (argc - optind < 1) { fprintf(stderr, "usage: panSVR signal|fc_signal [-N] [options] <in.bam>  > reads.fq\n"); return 1; }
	S.o.input = argv[optind];
	return S.run();
}

} // namespace psvr
