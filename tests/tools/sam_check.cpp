// sam_check.cpp -- TEST TOOL.  Reads the SAM text `panSVR aln -S` writes (from a file or a FIFO, so that the 20 GB of a
// configs[2]-size run never touch a disk) and checks, for EVERY record, what must hold whatever the input was:
//   * 11 mandatory fields + the tags of output_BAM (reference src/PanSVgenerateVCF/read_realignment.cpp:479-536): AS:i first,
//     OS:i and OA:Z present, RC:Z last;
//   * SEQ and QUAL have the read length, the CIGAR is well formed and consumes exactly that many read bases (M I S = X);
//   * FLAG has only the bits the reference sets, POS >= 1, MAPQ <= 40;
//   * records come in input order: the pair number in the read name (r%08d) never decreases.
// Prints one JSON line: records, header lines, bytes, a checksum of all bytes (FNV-1a over the 8-byte words of each 64 MB block), mapped records, sum of AS, first violation (if any).
//   sam_check <in.sam|-> <read_len> [--head-bytes N --head-file PATH]   (the first N bytes behind the header are also copied out)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
	if (argc < 3) { fprintf(stderr, "usage: sam_check <in.sam|-> <read_len> [--head-bytes N --head-file PATH]\n"); return 2; }
	FILE *in = strcmp(argv[1], "-") ? fopen(argv[1], "rb") : stdin;
	if (!in) { perror(argv[1]); return 2; }
	const int L = atoi(argv[2]);
	long long head_bytes = 0;
	FILE *head = nullptr;
	for (int i = 3; i + 1 < argc; i += 2) {
		if (!strcmp(argv[i], "--head-bytes")) head_bytes = atoll(argv[i + 1]);
		else if (!strcmp(argv[i], "--head-file")) head = fopen(argv[i + 1], "wb");
	}
	std::vector<char> buf((size_t)64 << 20);
	std::string carry, bad;
	long long n_rec = 0, n_hdr = 0, n_bytes = 0, n_mapped = 0, sum_as = 0, last_pair = -1, line_no = 0, body_bytes = 0;
	uint64_t fnv = 1469598103934665603ull;
	auto violation = [&](const char *what, const char *line, size_t len) {
		if (!bad.empty()) return;
		char b[256];
		snprintf(b, sizeof b, "line %lld: %s: ", line_no, what);
		bad = b;
		bad.append(line, len < 300 ? len : 300);
		for (char &c : bad) if (c == '"' || c == '\\' || (unsigned char)c < 32) c = ' ';
	};
	auto check = [&](const char *s, size_t len) {
		++line_no;
		if (len && s[0] == '@') { ++n_hdr; return; }
		if (head && body_bytes < head_bytes) {
			const long long take = (long long)len + 1 < head_bytes - body_bytes ? (long long)len + 1 : head_bytes - body_bytes;
			fwrite(s, 1, (size_t)(take > (long long)len ? len : take), head);
			if (take > (long long)len) fputc('\n', head);
		}
		body_bytes += (long long)len + 1;
		++n_rec;
		const char *f[16];
		size_t fl[16];
		int nf = 0;
		const char *p = s, *e = s + len, *last_tag = nullptr;
		size_t last_len = 0;
		int n_tags = 0;
		bool has_os = false, has_oa = false, as_first = false;
		while (p <= e) {
			const char *t = (const char *)memchr(p, '\t', (size_t)(e - p));
			const size_t l = t ? (size_t)(t - p) : (size_t)(e - p);
			if (nf < 11) f[nf] = p, fl[nf] = l, ++nf;
			else {
				if (n_tags == 0) as_first = l >= 5 && !memcmp(p, "AS:i:", 5);
				if (l >= 5 && !memcmp(p, "AS:i:", 5)) sum_as += atoll(p + 5);
				if (l >= 5 && !memcmp(p, "OS:i:", 5)) has_os = true;
				if (l >= 5 && !memcmp(p, "OA:Z:", 5)) has_oa = true;
				last_tag = p, last_len = l, ++n_tags;
			}
			if (!t) break;
			p = t + 1;
		}
		if (nf < 11 || n_tags < 3) return violation("fewer than 14 fields", s, len);
		if (!as_first || !has_os || !has_oa || !(last_len >= 5 && !memcmp(last_tag, "RC:Z:", 5))) return violation("tags (AS first, OS, OA, RC last)", s, len);
		if ((int)fl[9] != L || (int)fl[10] != L) return violation("SEQ / QUAL length", s, len);
		const long flag = strtol(f[1], nullptr, 10), pos = strtol(f[3], nullptr, 10), mapq = strtol(f[4], nullptr, 10);
		if (flag & ~(0x40 | 0x10 | 0x8)) return violation("FLAG bits", s, len);
		if (pos < 1 || mapq < 0 || mapq > 40) return violation("POS / MAPQ range", s, len);
		long num = 0, q = 0;
		bool any = false;
		for (size_t i = 0; i < fl[5]; ++i) {
			const char c = f[5][i];
			if (c >= '0' && c <= '9') { num = num * 10 + (c - '0'); any = true; continue; }
			if (!strchr("MIDNSHP=X", c) || !any || num <= 0) return violation("CIGAR syntax", s, len);
			if (strchr("MIS=X", c)) q += num;
			num = 0, any = false;
		}
		if (q != L || any) return violation("CIGAR does not consume the read", s, len);
		if (fl[0] >= 2 && f[0][0] == 'r') {
			const long long pr = atoll(f[0] + 1);
			if (pr < last_pair) return violation("records out of input order", s, len);
			last_pair = pr;
		}
		++n_mapped;
	};
	for (;;) {
		const size_t n = fread(buf.data(), 1, buf.size(), in);
		if (n == 0) break;
		n_bytes += (long long)n;
		{   // FNV-1a over 8-byte words (a byte at a time is a 20 GB serial chain); the tail of a block byte by byte
			size_t i = 0;
			for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, buf.data() + i, 8); fnv = (fnv ^ w) * 1099511628211ull; }
			for (; i < n; ++i) fnv = (fnv ^ (unsigned char)buf[i]) * 1099511628211ull;
		}
		size_t at = 0;
		while (at < n) {
			const char *nl = (const char *)memchr(buf.data() + at, '\n', n - at);
			if (!nl) { carry.append(buf.data() + at, n - at); break; }
			const size_t l = (size_t)(nl - (buf.data() + at));
			if (!carry.empty()) { carry.append(buf.data() + at, l); check(carry.data(), carry.size()); carry.clear(); }
			else check(buf.data() + at, l);
			at += l + 1;
		}
	}
	if (!carry.empty()) check(carry.data(), carry.size());
	if (head) fclose(head);
	printf("{\"records\":%lld,\"header_lines\":%lld,\"bytes\":%lld,\"fnv1a\":\"%016llx\",\"checked\":%lld,\"sum_as\":%lld,\"last_pair\":%lld,\"violation\":\"%s\"}\n", n_rec, n_hdr, n_bytes,
	       (unsigned long long)fnv, n_mapped, sum_as, last_pair, bad.c_str());
	return bad.empty() ? 0 : 1;
}
