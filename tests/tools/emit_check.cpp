// emit_check.cpp -- TEST TOOL.  The record formatter's fast paths (pansvr_amd/csrc/sam_emit.h) against plain restatements:
//   RawOut::num against snprintf("%lld"); put_seq_qual (16 bytes at a time) against the per-byte tables and getReverseStr_qual_char's
//   loop; SamEmitter::parse_ori_record on a span against its sscanf / strstr version, on well-formed comments and on mutated ones
//   (white space, signs, long digit runs, missing sections, NUL bytes).  Exit status 0 = every case agrees.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#define PSVR_NO_ENGINE_LIB 1
#include "../../pansvr_amd/csrc/sam_emit.h"

namespace psvr {
struct EmitCheck {
	typedef SamEmitter::OriRecord Rec;
	static bool fast(const char *c, int n, Rec *r) { return SamEmitter::parse_ori_record(c, n, r); }
	static bool slow(const std::string &c, Rec *r) { return SamEmitter::parse_ori_record(c, r); }
};
}
using namespace psvr;

static unsigned long long rng_state = 88172645463325252ull;
static unsigned long long rnd() { rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17; return rng_state; }

int main()
{
	long long bad = 0, n_cases = 0;
	// ---- numbers
	{
		Bytes buf;
		std::vector<long long> xs = {0, 1, 9, 10, 11, 99, 100, 101, 999, 1000, 1001, 9999, 10000, 99999, 100000, 999999, 1000000, 2147483647ll, -2147483647ll - 1, -1, -9, -10, -99, -100, -999, -1000,
		                             4294967295ll, 9223372036854775807ll, -9223372036854775807ll - 1};
		for (int i = 0; i < 200000; ++i) { long long x = (long long)(rnd() >> (rnd() % 64)); if (rnd() & 1) x = -x; xs.push_back(x); }
		for (long long x : xs) {
			buf.clear();
			RawOut o(buf, 64);
			o.num(x);
			o.close();
			char ref[32];
			const int m = snprintf(ref, sizeof ref, "%lld", x);
			++n_cases;
			if ((size_t)m != buf.size() || memcmp(ref, buf.data(), (size_t)m)) { if (bad++ < 5) fprintf(stderr, "num(%lld) differs\n", x); }
		}
	}
	// ---- SEQ / QUAL
	{
		const char *alpha = "ACGTNacgtnRYKMSWBDHV=.*XU0123\t \x01\xff";
		const int na = (int)strlen(alpha) + 1;           // (+ the NUL)
		for (int it = 0; it < 40000; ++it) {
			const int n = (int)(rnd() % 200);
			std::string t((size_t)n, 'A'), q((size_t)n, 'I');
			const int mode = (int)(rnd() % 3);            // 0: plain bases, 1: a few odd ones, 2: anything
			for (int i = 0; i < n; ++i) {
				t[(size_t)i] = mode == 0 || (mode == 1 && rnd() % 23) ? "ACGTN"[rnd() % 5] : alpha[rnd() % na];
				q[(size_t)i] = (char)(33 + rnd() % 60);
			}
			for (int rev = 0; rev < 2; ++rev) {
				Bytes buf;
				RawOut o(buf, 2 * (size_t)n + 64);
				put_seq_qual(o, t.data(), q.data(), n, rev != 0);
				o.close();
				std::string want_s = t, want_q = q;
				if (rev) sam_rev_seq(want_s), sam_rev_qual(want_q);      // (getReverseStr_char gives A C G T N only: what the 4-bit code gives back)
				else for (char &c : want_s) c = nt16_char(c);
				const std::string want = want_s + "\t" + want_q;
				++n_cases;
				if (want.size() != buf.size() || memcmp(want.data(), buf.data(), want.size())) { if (bad++ < 5) fprintf(stderr, "put_seq_qual differs (n %d, rev %d, mode %d)\n", n, rev, mode); }
			}
		}
	}
	// ---- the second file's comment sections
	{
		auto same = [&](const std::string &c) {
			EmitCheck::Rec a, b;
			const bool ra = EmitCheck::fast(c.data(), (int)c.size(), &a), rb = EmitCheck::slow(c, &b);
			++n_cases;
			if (ra != rb || (ra && (a.flag != b.flag || a.mapq != b.mapq || a.mate_chr != b.mate_chr || a.mate_pos != b.mate_pos || a.isize != b.isize || a.cigar != b.cigar || a.tags != b.tags))) {
				if (bad++ < 5) fprintf(stderr, "parse_ori_record differs on [%s] (%d / %d)\n", c.c_str(), (int)ra, (int)rb);
			}
		};
		const char *junk = "_: \t+-0123456789FLAGCIGARMATETAG_NMiZ,;*SMX";
		const int nj = (int)strlen(junk) + 1;
		for (int it = 0; it < 300000; ++it) {
			char b[512];
			const int flag = (int)(rnd() % 4096), mq = (int)(rnd() % 61), mc = (int)(rnd() % 30) - 2, mp = (int)(rnd() % 250000000) - 3, is = (int)(rnd() % 2000) - 1000;
			const char *cg[] = {"150M", "40S110M", "*", "", "10M2I138M", "5H20M1D100M25S", "-3M"};
			const char *tg[] = {"NM:i:3_", "NM:i:0_MD:Z:150_AS:i:150_XS:i:20_", "", "SA:Z:chr1,5,+,50M100S,60,0;_XA:Z:x_y_", "NM:i:1", "_"};
			snprintf(b, sizeof b, "0_%d_40_140_20_20_0_0_494_RNNY_FNNY_STAT_150_200_400_600_FLAG_%d_%d_CIGAR_%s_MATE_%d_%d_%d_TAG_%s", (int)(rnd() % 1000000), flag, mq, cg[rnd() % 7], mc, mp, is, tg[rnd() % 6]);
			std::string c = b;
			same(c);
			// mutations: a character replaced / inserted / removed somewhere behind "FLAG_", a truncation, a NUL
			for (int m = 0; m < 3; ++m) {
				std::string d = c;
				const size_t from = d.find("FLAG_");
				const size_t at = from + rnd() % (d.size() - from + 1);
				switch (rnd() % 5) {
				case 0: if (at < d.size()) d[at] = junk[rnd() % nj]; break;
				case 1: d.insert(at, 1, junk[rnd() % nj]); break;
				case 2: if (at < d.size()) d.erase(at, 1 + rnd() % 3); break;
				case 3: d.resize(at); break;
				default: d.insert(at, std::string((size_t)(1 + rnd() % 12), (char)('0' + rnd() % 10))); break;
				}
				same(d);
			}
		}
		same(""), same("FLAG_"), same("FLAG_1_2_CIGAR_"), same("FLAG_1_2_CIGAR_5M_"), same("FLAG_1_2_CIGAR_5M_MATE_"), same("FLAG_1_2_CIGAR_5M_MATE_1_2_3_TAG_"), same("FLAG_ 1_2_CIGAR_5M_MATE_1_2_3_TAG_x");
		same("FLAG_+1_2_CIGAR_5M_MATE_+1_-2_3_TAG_NM:i:1_"), same("FLAG_99999999999_2_CIGAR_5M_MATE_1_2_3_TAG_NM:i:1_"), same("xFLAG_1_2_FLAG_3_4_CIGAR_CIGAR_5M_MATE_1_2_3_TAG_TAG_");
	}
	// ---- the stages' worker threads: every index of every call exactly once, calls of different widths in a row, a call from inside a worker
	{
		for (int it = 0; it < 2000; ++it) {
			const long long n = (long long)(rnd() % 5000);
			const int threads = 1 + (int)(rnd() % 24);
			std::vector<int> hit((size_t)n, 0);
			parallel_ranges(n, threads, [&](long long a, long long b) { for (long long i = a; i < b; ++i) hit[(size_t)i]++; });
			++n_cases;
			for (int h : hit) if (h != 1) { if (bad++ < 5) fprintf(stderr, "parallel_ranges(%lld, %d): an index %d times\n", n, threads, h); break; }
		}
		std::atomic<long long> sum(0);
		thread_pool().run(6, [&](int t) { parallel_ranges(1000, 4, [&](long long a, long long b) { sum += (b - a) * (t + 1); }); });
		++n_cases;
		if (sum != 1000 * 21) { ++bad; fprintf(stderr, "nested pool calls: %lld\n", (long long)sum); }
	}
	printf("%lld cases, %lld differ\n", n_cases, bad);
	return bad ? 1 : 0;
}
