// deflate_device.h -- one DEFLATE stream (RFC 1951) per BGZF block, written so that ONE THREAD compresses one block: the body of
// k_bgzf_deflate (engine.hip: a lane per block, thousands of blocks of a BAM batch in flight) and, compiled for the host, of the test that
// inflates its output with zlib (tests/tools/deflate_check.cpp).  htslib compresses BGZF blocks with zlib on the host
// (bgzf.c: bgzf_compress -> deflate); at the engine's rate that deflate is the drop-in command's BAM route (7-18 CPU-seconds per 1 M
// pairs against 0.02 s of alignment).  The format only fixes what a decoder must accept, so this is a small encoder of its own:
//   * LZ77: greedy, one candidate per position from a hash table of the last position of every 3-byte hash (the table is the caller's:
//     LDS on the device), matches of 3..258 bytes at distances up to 32768, every position of a match entered into the table;
//   * one dynamic-Huffman block per BGZF block: code lengths by Moffat's in-place minimum-redundancy algorithm on the sorted
//     frequencies, limited to 15 bits the way miniz does it, canonical codes; the code-length alphabet is sent with fixed 4-bit codes
//     for the lengths 0..15 (no run-length symbols: ~160 header bytes per 64 KB block);
//   * a stored block when that does not pay (the member must fit BGZF's 64 KB).
// The gzip wrapper, CRC32 and ISIZE of a BGZF member are the host's (bam_writer.h).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PSVR_DF __host__ __device__ inline
#else
#define PSVR_DF inline
#endif

namespace psvr {

static const uint32_t kDfMaxIn = 0xff00;           // bytes per BGZF block (htslib's BGZF_BLOCK_SIZE)
static const int kDfLit = 286, kDfDist = 30;

struct DfBits {                                     // LSB-first bit writer into [p, end): four bytes at a time, one test per put
	uint8_t *p, *end;
	uint64_t acc;
	int n;
	bool over;
	PSVR_DF void put(uint32_t v, int bits)          // bits <= 16
	{
		acc |= (uint64_t)v << n;
		n += bits;
		if (n >= 32) {
			if (p + 4 <= end) { const uint32_t w = (uint32_t)acc; __builtin_memcpy(p, &w, 4); p += 4; } else over = true;
			acc >>= 32, n -= 32;
		}
	}
	PSVR_DF void flush()
	{
		while (n > 0) { if (p < end) *p++ = (uint8_t)acc; else over = true; acc >>= 8, n -= 8; }
		acc = 0, n = 0;
	}
};

// A block's small tables live in `fast` memory of the caller's (LDS on the device, 64 lanes' worth per workgroup): a lane waits for every
// access to them, and in global memory that wait was the encoder's time (a frequency count per literal alone: 60 ms per call).  Layout
// (bytes, hbits = log2 of the hash table's entries):
//   LZ77 pass:        hash   u16[1 << hbits] @ 0            freq  u16[316] @ F
//   code lengths:     S      u16[286] @ 0,  A  u16[286] @ 576   freq @ F   len  u8[316] @ F + 640
//   encoding pass:    codes  u32[316] @ 0   (code | length << 16)
// with F = max(2 << hbits, 1280); df_fast_bytes(hbits) in all.  The tokens (4 bytes per input byte at most) stay in global memory: written
// once, read once, in order.
PSVR_DF uint32_t df_fast_freq_at(int hbits) { const uint32_t h = 2u << hbits; return h > 1280u ? h : 1280u; }
PSVR_DF uint32_t df_fast_bytes(int hbits) { return df_fast_freq_at(hbits) + 640u + 320u; }

// length 3..258 -> symbol 257..285 and its extra bits; distance 1..32768 -> symbol 0..29 and its extra bits (RFC 1951 3.2.5)
PSVR_DF uint32_t df_ctz64(uint64_t x)               // trailing zero bits, x != 0
{
#if defined(__HIP_DEVICE_COMPILE__)
	return (uint32_t)(__ffsll((unsigned long long)x) - 1);
#else
	return (uint32_t)__builtin_ctzll(x);
#endif
}
PSVR_DF uint32_t df_log2(uint32_t x)                  // floor(log2 x), x > 0
{
#if defined(__HIP_DEVICE_COMPILE__)
	return 31u - (uint32_t)__clz((int)x);
#else
	return 31u - (uint32_t)__builtin_clz(x);
#endif
}
// (no loops, no branches on the value: the lanes of a wavefront each compress a block of their own, and every branch one of them takes
// is executed by all)
PSVR_DF void df_len_code(uint32_t len, uint32_t &sym, uint32_t &ebits, uint32_t &eval)
{
	const uint32_t l = len - 3;                     // 0..255
	const uint32_t e = l < 8 ? 0u : df_log2(l) - 2u;            // 4 << e <= l < 8 << e: four codes per extra-bit count, (l >> e) & 3 picks one
	const uint32_t s = l < 8 ? 257u + l : 257u + 4u * (e + 1u) + ((l >> e) & 3u);
	const bool top = len == 258;                    // its own code, no extra bits
	sym = top ? 285u : s, ebits = top ? 0u : e, eval = top ? 0u : l & ((1u << e) - 1u);
}
PSVR_DF void df_dist_code(uint32_t dist, uint32_t &sym, uint32_t &ebits, uint32_t &eval)
{
	const uint32_t d = dist - 1;                    // 0..32767
	const uint32_t e = d < 4 ? 0u : df_log2(d) - 1u;            // 2 << e <= d < 4 << e
	sym = d < 4 ? d : 2u * (e + 1u) + ((d >> e) & 1u);
	ebits = e, eval = d & ((1u << e) - 1u);
}

// code lengths (at most `maxbits`) for the n symbols with the frequencies f[0..n); len[] gets 0 for unused symbols.  At least two
// symbols get a code (zlib does the same: a lone code confuses some decoders).
PSVR_DF void df_code_lengths(const uint16_t *f, int n, int maxbits, uint8_t *len, uint16_t *A, uint16_t *S)
{
	int m = 0;
	for (int i = 0; i < n; ++i) { len[i] = 0; if (f[i]) S[m++] = (uint16_t)i; }
	// force two symbols
	for (int i = 0; m < 2 && i < n; ++i) { bool have = false; for (int k = 0; k < m; ++k) have |= S[k] == i; if (!have) S[m++] = (uint16_t)i; }
	// sort by (frequency, symbol): insertion sort, n <= 286
	for (int i = 1; i < m; ++i) {
		const uint16_t s = S[i];
		const uint32_t fs = f[s] ? f[s] : 1u;
		int j = i;
		while (j > 0) { const uint32_t fp = f[S[j - 1]] ? f[S[j - 1]] : 1u; if (fp < fs || (fp == fs && S[j - 1] < s)) break; S[j] = S[j - 1]; --j; }
		S[j] = s;
	}
	for (int i = 0; i < m; ++i) A[i] = f[S[i]] ? f[S[i]] : (uint16_t)1;      // (a block's symbols number at most 65281: the sums fit)
	if (m == 2) { len[S[0]] = len[S[1]] = 1; return; }
	// Moffat & Katajainen, in-place calculation of minimum-redundancy codes: A[i] becomes the code length of the i-th smallest frequency
	{
		int root, leaf, next, avbl, used, dpth;
		A[0] = (uint16_t)(A[0] + A[1]), root = 0, leaf = 2;
		for (next = 1; next < m - 1; ++next) {
			if (leaf >= m || A[root] < A[leaf]) A[next] = A[root], A[root++] = (uint16_t)next; else A[next] = A[leaf++];
			if (leaf >= m || (root < next && A[root] < A[leaf])) A[next] = (uint16_t)(A[next] + A[root]), A[root++] = (uint16_t)next; else A[next] = (uint16_t)(A[next] + A[leaf++]);
		}
		A[m - 2] = 0;
		for (next = m - 3; next >= 0; --next) A[next] = (uint16_t)(A[A[next]] + 1);
		avbl = 1, used = dpth = 0, root = m - 2, next = m - 1;
		while (avbl > 0) {
			while (root >= 0 && (int)A[root] == dpth) ++used, --root;
			while (avbl > used) A[next--] = (uint16_t)dpth, --avbl;
			avbl = 2 * used, ++dpth, used = 0;
		}
	}
	// limit to maxbits (miniz: tdefl_huffman_enforce_max_code_size): count per length, fold the long ones, repair the Kraft sum
	{
		int cnt[33];
		for (int i = 0; i <= 32; ++i) cnt[i] = 0;
		for (int i = 0; i < m; ++i) cnt[A[i] > 32 ? 32 : A[i]]++;
		for (int i = maxbits + 1; i <= 32; ++i) cnt[maxbits] += cnt[i];
		uint32_t total = 0;
		for (int i = maxbits; i > 0; --i) total += (uint32_t)cnt[i] << (maxbits - i);
		while (total != (1u << maxbits)) {
			cnt[maxbits]--;
			for (int i = maxbits - 1; i > 0; --i) if (cnt[i]) { cnt[i]--, cnt[i + 1] += 2; break; }
			total--;
		}
		// the symbols in frequency order take the lengths from the longest down
		int k = 0;
		for (int l = maxbits; l > 0; --l) for (int c = 0; c < cnt[l]; ++c) len[S[k++]] = (uint8_t)l;
	}
}
// canonical codes, bit-reversed for the LSB-first stream, as code | length << 16
PSVR_DF void df_codes(const uint8_t *len, int n, uint32_t *code)
{
	uint32_t cnt[16], next[16];
	for (int i = 0; i < 16; ++i) cnt[i] = 0;
	for (int i = 0; i < n; ++i) cnt[len[i]]++;
	cnt[0] = 0;
	uint32_t c = 0;
	next[0] = 0;
	for (int l = 1; l < 16; ++l) c = (c + cnt[l - 1]) << 1, next[l] = c;
	for (int i = 0; i < n; ++i) {
		const uint32_t l = len[i];
		if (!l) { code[i] = 0; continue; }
		uint32_t v = next[l]++, r = 0;
		for (uint32_t b = 0; b < l; ++b) r = (r << 1) | ((v >> b) & 1u);
		code[i] = r | (l << 16);
	}
}

// The raw DEFLATE stream of in[0..n) (n <= kDfMaxIn) into out[0..cap); returns its size, 0 if even a stored block does not fit.
// `fast`: df_fast_bytes(hbits) bytes of the caller's fast memory (any content); tok: n + 4 words of scratch, 16-byte aligned.
PSVR_DF uint32_t deflate_block(const uint8_t *in, uint32_t n, uint8_t *out, uint32_t cap, uint8_t *fast, int hbits, uint32_t *tok)
{
	const uint32_t hmask = (1u << hbits) - 1u;
	uint16_t *head = (uint16_t *)fast;
	uint16_t *lf = (uint16_t *)(fast + df_fast_freq_at(hbits)), *df = lf + kDfLit;
	for (uint32_t i = 0; i <= hmask; ++i) head[i] = 0;                        // 0 = no position yet (positions are kept + 1)
	for (int i = 0; i < kDfLit + kDfDist; ++i) lf[i] = 0;
	// four bytes at a time (one unaligned load: a position's three hashed bytes and the first byte a match must extend over); the loads of
	// a lane are what this loop waits for -- a byte at a time it was several dependent round trips per position
	auto ld32 = [&](uint32_t p) { uint32_t v; __builtin_memcpy(&v, in + p, 4); return v; };
	// the bytes at the LZ77 cursor come from an 8-byte window that is loaded once per five positions of a literal run
	uint64_t win = 0;
	uint32_t wpos = 0x7fffffffu;                                              // (no window yet: every position is "more than four" away)
	auto cur32 = [&](uint32_t p) {                                            // bytes p .. p+3, p + 4 <= n
		if (p - wpos > 4u) {                                                  // (also when p < wpos: the difference wraps)
			if (p + 8 <= n) __builtin_memcpy(&win, in + p, 8); else { win = 0; for (uint32_t k = 0; p + k < n; ++k) win |= (uint64_t)in[p + k] << (8 * k); }
			wpos = p;
		}
		return (uint32_t)(win >> (8 * (p - wpos)));
	};
	auto hash_of = [&](uint32_t tri) { return ((tri & 0xffffffu) * 0x9E3779B1u >> (32 - hbits)) & hmask; };   // the three bytes at a position, little endian
	uint32_t nt = 0;
	for (uint32_t i = 0; i < n;) {
		uint32_t best = 0, dist = 0, lit = 0;
		if (i + 4 <= n) {
			const uint32_t cur = cur32(i);
			lit = cur & 0xffu;
			const uint32_t h = hash_of(cur);
			const uint32_t c = head[h];
			head[h] = (uint16_t)(i + 1);
			if (c && i + 1 - c <= 32768u) {
				const uint32_t cp = c - 1, lim = n - i < 258u ? n - i : 258u;
				if (((ld32(cp) ^ cur) & 0xffffffu) == 0) {                       // (cp < i: cp + 4 <= n)
					uint32_t k = 3;
					while (k + 8 <= lim) {                                        // eight bytes a step
						uint64_t a, b;
						__builtin_memcpy(&a, in + cp + k, 8), __builtin_memcpy(&b, in + i + k, 8);
						const uint64_t x = a ^ b;
						if (x) { k += (uint32_t)(df_ctz64(x) >> 3); break; }
						k += 8;
					}
					if (k + 8 > lim) while (k < lim && in[cp + k] == in[i + k]) ++k;
					best = k, dist = i - cp;
				}
			}
		} else lit = in[i];                                                   // (the last three bytes of a block go out as literals)
		// (positions inside a match are not entered into the table: 1.6 % of the ratio on BAM records for a loop every lane would wait for)
		tok[nt++] = best ? 0x80000000u | ((best - 3) << 16) | (dist - 1) : lit;
		i += best ? best : 1u;
	}
	// frequencies: a pass of its own over the tokens, the same few instructions for a literal and a match
	for (uint32_t t = 0; t < nt; ++t) {
		const uint32_t x = tok[t];
		const bool m = (x & 0x80000000u) != 0;
		uint32_t s, d, eb, ev;
		df_len_code(m ? ((x >> 16) & 0xff) + 3 : 3u, s, eb, ev);
		df_dist_code(m ? (x & 0xffff) + 1 : 1u, d, eb, ev);
		uint16_t &c1 = lf[m ? s : x];
		if (c1 != 0xffff) ++c1;                                               // (a block holds at most 65280 symbols)
		if (m && df[d] != 0xffff) ++df[d];
	}
	lf[256] = 1;                                                               // end of block
	uint8_t *ll = fast + df_fast_freq_at(hbits) + 640, *dl = ll + kDfLit;
	{
		uint16_t *S = (uint16_t *)fast, *A = (uint16_t *)(fast + 576);           // (the hash table is through)
		df_code_lengths(lf, kDfLit, 15, ll, A, S);
		df_code_lengths(df, kDfDist, 15, dl, A, S);
	}
	uint32_t *lc = (uint32_t *)fast, *dc = lc + kDfLit;
	df_codes(ll, kDfLit, lc), df_codes(dl, kDfDist, dc);
	int nl = kDfLit, nd = kDfDist;
	while (nl > 257 && !ll[nl - 1]) --nl;
	while (nd > 1 && !dl[nd - 1]) --nd;
	DfBits b;
	b.p = out, b.end = out + cap, b.acc = 0, b.n = 0, b.over = false;
	b.put(1, 1), b.put(2, 2);                                                  // BFINAL, BTYPE = dynamic Huffman
	b.put((uint32_t)(nl - 257), 5), b.put((uint32_t)(nd - 1), 5), b.put(15, 4);    // HLIT, HDIST, HCLEN: all 19 code-length codes
	// code-length alphabet in its transmission order (16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15): 16..18 unused, 0..15 four bits
	// each, whose canonical codes are the values themselves
	b.put(0, 3), b.put(0, 3), b.put(0, 3);
	for (int i = 0; i < 16; ++i) b.put(4, 3);
	auto put_len = [&](uint32_t v) { b.put(((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3), 4); };   // 4-bit code, reversed
	for (int i = 0; i < nl; ++i) put_len(ll[i]);
	for (int i = 0; i < nd; ++i) put_len(dl[i]);
	auto put_sym = [&](uint32_t cl) { b.put(cl & 0xffffu, (int)(cl >> 16)); };
	uint32_t t4[4] = {0, 0, 0, 0};
	for (uint32_t t = 0; t < nt && !b.over; ++t) {
		if ((t & 3u) == 0) __builtin_memcpy(t4, tok + t, 16);                 // (the scratch holds whole groups of four: its size is a multiple of 16 bytes and at least n + 3 words)
		const uint32_t x = t4[t & 3u];
		if (!(x & 0x80000000u)) { put_sym(lc[x]); continue; }
		uint32_t s, eb, ev;
		df_len_code(((x >> 16) & 0xff) + 3, s, eb, ev);
		put_sym(lc[s]);
		if (eb) b.put(ev, (int)eb);
		df_dist_code((x & 0xffff) + 1, s, eb, ev);
		put_sym(dc[s]);
		if (eb) b.put(ev, (int)eb);
	}
	put_sym(lc[256]);
	b.flush();
	const uint32_t used = (uint32_t)(b.p - out);
	if (!b.over && used < n + 5) return used;
	// stored: BFINAL = 1, BTYPE = 00, LEN, ~LEN, the bytes
	if (n + 5 > cap) return 0;
	out[0] = 1, out[1] = (uint8_t)n, out[2] = (uint8_t)(n >> 8), out[3] = (uint8_t)~n, out[4] = (uint8_t)(~n >> 8);
	for (uint32_t i = 0; i < n; ++i) out[5 + i] = in[i];
	return n + 5;
}

} // namespace psvr
