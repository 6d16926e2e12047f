// deflate_check.cpp -- the one-thread-per-block DEFLATE encoder of pansvr_amd/csrc/deflate_device.h, compiled for the host: every
// 0xff00-byte block of the input files (and of a few synthetic buffers) is compressed and inflated again with zlib; prints the ratio.
// usage: deflate_check [hash bits] [file ...]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <string>
#include <vector>
#include "../../pansvr_amd/csrc/deflate_device.h"
using namespace psvr;

static bool check(const std::vector<uint8_t> &data, int hbits, const char *what)
{
	std::vector<uint8_t> fast(df_fast_bytes(hbits)), out(0x10000 + 64), back(kDfMaxIn + 16);
	std::vector<uint32_t> tok(kDfMaxIn + 8);
	size_t total_in = 0, total_out = 0, stored = 0;
	for (size_t o = 0; o < data.size() || o == 0; o += kDfMaxIn) {
		const uint32_t n = (uint32_t)(data.size() - o < kDfMaxIn ? data.size() - o : kDfMaxIn);
		const uint32_t c = deflate_block(data.data() + o, n, out.data(), 0x10000 - 26, fast.data(), hbits, tok.data());
		if (!c) { fprintf(stderr, "%s: block at %zu did not fit\n", what, o); return false; }
		z_stream zs;
		memset(&zs, 0, sizeof zs);
		inflateInit2(&zs, -15);
		zs.next_in = out.data(), zs.avail_in = c, zs.next_out = back.data(), zs.avail_out = (uInt)back.size();
		const int rc = inflate(&zs, Z_FINISH);
		const bool ok = rc == Z_STREAM_END && zs.total_out == n && zs.total_in == c && !memcmp(back.data(), data.data() + o, n);
		inflateEnd(&zs);
		if (!ok) { fprintf(stderr, "%s: block at %zu: inflate rc %d, %lu of %u bytes back, %lu of %u consumed (%s)\n", what, o, rc, zs.total_out, n, zs.total_in, c, zs.msg ? zs.msg : "-"); return false; }
		total_in += n, total_out += c, stored += (out[0] & 6) == 0;
		if (data.empty()) break;
	}
	printf("%-28s %10zu -> %10zu bytes  ratio %.3f  (%zu stored blocks)\n", what, total_in, total_out, total_in ? (double)total_in / (double)total_out : 0.0, stored);
	return true;
}

int main(int argc, char **argv)
{
	const int hbits = argc > 1 ? atoi(argv[1]) : 10;
	bool ok = true;
	srand(7);
	{ std::vector<uint8_t> v; ok &= check(v, hbits, "empty"); }
	{ std::vector<uint8_t> v(1, 'x'); ok &= check(v, hbits, "one byte"); }
	{ std::vector<uint8_t> v(200000, 0); ok &= check(v, hbits, "zeros"); }
	{ std::vector<uint8_t> v(200000); for (auto &x : v) x = (uint8_t)rand(); ok &= check(v, hbits, "random bytes"); }
	{ std::vector<uint8_t> v(300000); for (size_t i = 0; i < v.size(); ++i) v[i] = "ACGT"[rand() & 3]; ok &= check(v, hbits, "random ACGT"); }
	{ std::vector<uint8_t> v; for (int i = 0; i < 20000; ++i) { char b[64]; int n = snprintf(b, sizeof b, "read%07d\tAS:i:%d\tXA:Z:chr%d,%d;\n", i, rand() % 300, rand() % 24, rand()); v.insert(v.end(), b, b + n); } ok &= check(v, hbits, "tag-like text"); }
	{ std::vector<uint8_t> v(70000); for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)(i % 259 < 258 ? 'a' : 'b'); ok &= check(v, hbits, "period 259 (length 258 matches)"); }
	{ std::vector<uint8_t> v(65280); for (size_t i = 0; i < v.size(); ++i) v[i] = (uint8_t)((i * 2654435761u) >> 24); for (size_t i = 40000; i < 40300; ++i) v[i] = v[i - 32768]; ok &= check(v, hbits, "a match at distance 32768"); }
	for (int a = 2; a < argc; ++a) {
		FILE *f = fopen(argv[a], "rb");
		if (!f) { fprintf(stderr, "cannot open %s\n", argv[a]); return 2; }
		std::vector<uint8_t> v;
		uint8_t buf[1 << 16];
		size_t k;
		while ((k = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + k);
		fclose(f);
		ok &= check(v, hbits, argv[a]);
	}
	return ok ? 0 : 1;
}
