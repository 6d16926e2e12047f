// gen_signal_fastq.cpp -- TEST TOOL.  Synthetic `signal | aln` input at BASELINE configs[2] size, fast enough to feed the
// drop-in command through a pipe (no 21 GB file): SV anchor FASTA + interleaved FASTQ in fc_signal's wire format
// (reference src/PanSVgenerateVCF/getSignalRead.cpp:158-247; the same record tests/synth.py and tests/bench_data.py write).
//
//   gen_signal_fastq anchors <n_anchors> <seed>                      > anchors.fa
//   gen_signal_fastq reads   <n_anchors> <seed> <n_pairs> <rseed> [threads] [first_pair]   > reads.fq
//
// Workload of SURVEY 8(d): anchors = 2 x 500 bp flanks + an allele of 60..300 random bases; 150 bp pairs from fragments of
// 300..500 bp; per read 30 % 1-4 substitutions, 20 % a 1-8 bp deletion, 20 % a 1-8 bp insertion, 30 % exact; 20 % of the pairs
// from random sequence; 1 % of the reads with an N.  Every pair is a pure function of (rseed, pair index) -- splitmix64 --, so any
// prefix or slice can be regenerated on its own, with any number of threads.  Anchor 0 is never sampled (tests/datasets.py).
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t mix(uint64_t &s)
{
	uint64_t z = (s += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
static inline uint32_t below(uint64_t &s, uint32_t n) { return (uint32_t)((mix(s) >> 32) * (uint64_t)n >> 32); }
// an independent stream per (seed, item): the state is a hash of both -- states that differ by the generator's own increment would give
// the SAME stream shifted by one draw (every anchor the previous one moved by a base)
static inline uint64_t stream_of(uint64_t seed, uint64_t item)
{
	uint64_t a = seed ^ 0xA0761D6478BD642Full, b = item * 0xE7037ED1A0B428DBull + 0x8EBC6AF09C88C6E3ull;
	const uint64_t x = mix(a) ^ mix(b);
	uint64_t c = x;
	return mix(c);
}

struct Anchors {
	std::vector<std::string> name;
	std::vector<std::string> seq;
	std::vector<long long> st_pos;
};

static void make_anchors(int n, uint64_t seed, Anchors *A)
{
	const int edge = 500;
	for (int i = 0; i < n; ++i) {
		uint64_t s = stream_of(seed, (uint64_t)i);
		const int alen = 60 + (int)below(s, 241);
		std::string q((size_t)(2 * edge + alen), 'A');
		for (char &c : q) c = "ACGT"[mix(s) >> 62];
		const long long st = 10000ll * (i + 1);
		char nm[160];
		snprintf(nm, sizeof nm, "%d_chr1_%lld_%d_INS_%lld_%lld_%lld_sv.INS.%d", i, st, (int)q.size(), st + edge, st + edge, st + 2 * edge, i);
		A->name.push_back(nm), A->seq.push_back(q), A->st_pos.push_back(st);
	}
}

static char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; }

static void one_pair(const Anchors &A, uint64_t rseed, long long p, bool stat, std::string &out)
{
	const int L = 150, maxindel = 8, W = L + maxindel;
	uint64_t s = stream_of(rseed ^ 0x5851F42D4C957F2Dull, (uint64_t)p);
	const bool miss = below(s, 100) < 20;
	const int a = 1 + (int)below(s, (uint32_t)A.seq.size() - 1);
	const std::string &ref = A.seq[(size_t)a];
	const int flen = 300 + (int)below(s, 201);
	int room = (int)ref.size() - flen - 2 * maxindel - 2;
	if (room < 1) room = 1;
	const int off = (int)below(s, (uint32_t)room);
	char src[2][W + 1];
	if (miss) {
		for (int e = 0; e < 2; ++e) for (int i = 0; i < W; ++i) src[e][i] = "ACGT"[mix(s) >> 62];
	} else {
		for (int i = 0; i < W; ++i) src[0][i] = ref[(size_t)(off + i)];
		for (int i = 0; i < W; ++i) src[1][i] = comp(ref[(size_t)(off + flen - 1 - i)]);
	}
	char rd[2][L + 1];
	for (int e = 0; e < 2; ++e) {
		const uint32_t kind = below(s, 10);           // 0-2 substitutions, 3-4 deletion, 5-6 insertion, 7-9 exact
		const int d = 1 + (int)below(s, maxindel), q = 10 + (int)below(s, L - 20 - maxindel);
		if (kind >= 3 && kind <= 4) { for (int i = 0; i < L; ++i) rd[e][i] = src[e][i < q ? i : i + d]; }
		else if (kind >= 5 && kind <= 6) { for (int i = 0; i < L; ++i) rd[e][i] = i < q ? src[e][i] : i < q + d ? "ACGT"[mix(s) >> 62] : src[e][i - d]; }
		else memcpy(rd[e], src[e], L);
		if (kind <= 2) {
			const int ns = 1 + (int)below(s, 4);
			for (int j = 0; j < ns; ++j) { const int at = (int)below(s, L); const char *t = strchr("ACGT", rd[e][at]); rd[e][at] = "ACGT"[((t ? t - "ACGT" : 0) + 1 + below(s, 3)) & 3]; }
		}
		if (below(s, 100) < 1) rd[e][below(s, L)] = 'N';
		rd[e][L] = 0;
	}
	const bool swap = below(s, 2) != 0;
	const long long pos1 = A.st_pos[(size_t)a] + off, pos2 = pos1 + flen - L;
	char buf[1024];
	for (int k = 0; k < 2; ++k) {
		const int e = k ^ (swap ? 1 : 0);
		const bool fw = e == 0;
		const int flag = (k == 0 ? 0x40 : 0x80) | 0x1 | (fw ? 0 : 0x10) | (fw ? 0x20 : 0);
		int n = snprintf(buf, sizeof buf, "@r%08lld 0_%lld_40_140_20_20_0_0_%d_%sNNY_%sNNY_", p, fw ? pos1 : pos2, flen, fw ? "F" : "R", fw ? "R" : "F");
		if (stat && k == 0) n += snprintf(buf + n, sizeof buf - n, "STAT_150_200_400_600_");
		n += snprintf(buf + n, sizeof buf - n, "FLAG_%d_20_CIGAR_40S110M_MATE_0_%lld_%d_TAG_NM:i:3_\n", flag, fw ? pos2 : pos1, fw ? flen : -flen);
		out.append(buf, (size_t)n);
		out.append(rd[e], L);
		out.append("\n+\n", 3);
		out.append((size_t)L, 'I');
		out.push_back('\n');
	}
}

int main(int argc, char **argv)
{
	if (argc >= 4 && !strcmp(argv[1], "anchors")) {
		Anchors A;
		make_anchors(atoi(argv[2]), strtoull(argv[3], nullptr, 10), &A);
		for (size_t i = 0; i < A.seq.size(); ++i) printf(">%s\n%s\n", A.name[i].c_str(), A.seq[i].c_str());
		return 0;
	}
	if (argc >= 6 && !strcmp(argv[1], "reads")) {
		Anchors A;
		make_anchors(atoi(argv[2]), strtoull(argv[3], nullptr, 10), &A);
		const long long n_pairs = atoll(argv[4]);
		const uint64_t rseed = strtoull(argv[5], nullptr, 10);
		int nt = argc > 6 ? atoi(argv[6]) : 4;
		const long long first = argc > 7 ? atoll(argv[7]) : 0;
		if (nt < 1) nt = 1;
		const long long chunk = 32768;
		// rounds of nt chunks, formatted in parallel and written in order; round r + 1 is formatted while round r is written
		std::vector<std::string> bufs[2] = {std::vector<std::string>((size_t)nt), std::vector<std::string>((size_t)nt)};
		auto fill = [&](std::vector<std::string> &bs, long long base) {
			std::vector<std::thread> th;
			for (int t = 0; t < nt; ++t) {
				th.emplace_back([&, t]() {
					std::string &o = bs[(size_t)t];
					o.clear();
					const long long p0 = base + t * chunk, p1 = p0 + chunk < n_pairs ? p0 + chunk : n_pairs;
					for (long long p = p0; p < p1; ++p) one_pair(A, rseed, first + p, first + p == 0, o);
				});
			}
			for (std::thread &x : th) x.join();
		};
		int cur = 0;
#ifdef F_SETPIPE_SZ
		(void)fcntl(1, F_SETPIPE_SZ, 1 << 20);
#endif
		fill(bufs[0], 0);
		for (long long base = 0; base < n_pairs; base += chunk * nt, cur ^= 1) {
			std::thread next;
			const long long nb = base + chunk * nt;
			if (nb < n_pairs) next = std::thread([&, nb]() { fill(bufs[cur ^ 1], nb); });
			bool gone = false;
			for (int t = 0; t < nt && !gone; ++t) {
				const std::string &o = bufs[cur][(size_t)t];
				for (size_t at = 0; at < o.size() && !gone;) {                                       // straight to the pipe, in large pieces
					const ssize_t w = write(1, o.data() + at, o.size() - at);
					if (w <= 0) gone = true;                                                          // the reader stopped (-R)
					else at += (size_t)w;
				}
			}
			if (next.joinable()) next.join();
			if (gone) return 0;
		}
		return 0;
	}
	fprintf(stderr, "usage: gen_signal_fastq anchors <n> <seed> | reads <n_anchors> <seed> <n_pairs> <rseed> [threads] [first_pair]\n");
	return 1;
}
