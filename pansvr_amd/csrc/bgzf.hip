// bgzf.hip -- BGZF members on the device: psvr_bgzf_compress (include/psvr_engine.h).  The reference writes its BAM through htslib, whose
// bgzf_compress deflates 0xff00-byte blocks with zlib on the host (htslib bgzf.c: bgzf_write -> bgzf_flush -> bgzf_compress); behind
// the MI355X engine that deflate is what the drop-in command's default output costs.  Blocks are independent, so a batch's thousands of
// blocks are compressed side by side, ONE LANE PER BLOCK (deflate_device.h: greedy LZ77 with the hash table in LDS, one dynamic-Huffman
// block, CRC32 from an LDS table), and a second launch packs the members next to each other for one transfer back.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "../../include/psvr_engine.h"
#include "common.h"
#include "deflate_device.h"

namespace psvr {

static const int kBgzfHashBits = 9;                          // 512 x uint16 per block: with its other tables 2240 bytes of LDS per lane, 140 KB per wavefront
static const uint32_t kBgzfLaneLds = 2240 + 4;               // (an odd number of words: the lanes' tables start in different banks)
static_assert(kBgzfLaneLds >= 2240, "lane tables");

__global__ __launch_bounds__(64) void k_bgzf_deflate(const uint8_t *in, long long n_bytes, long long n_blocks, uint32_t blk, uint32_t slot, uint8_t *slots, uint8_t *work, uint32_t work_stride, int32_t *len)
{
	extern __shared__ __align__(16) uint8_t bgzf_lds[];
	uint32_t *crc_tab = (uint32_t *)bgzf_lds;                                // [256]
	uint8_t *fast = bgzf_lds + 1024 + (size_t)threadIdx.x * kBgzfLaneLds;   // this lane's tables (deflate_device.h)
	for (int i = threadIdx.x; i < 256; i += 64) {
		uint32_t c = (uint32_t)i;
		for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
		crc_tab[i] = c;
	}
	__builtin_amdgcn_wave_barrier();
	const long long b = blockIdx.x * 64ll + threadIdx.x;
	if (b >= n_blocks) return;
	const uint8_t *src = in + b * (long long)blk;
	const uint32_t n = (uint32_t)(n_bytes - b * (long long)blk < (long long)blk ? n_bytes - b * (long long)blk : (long long)blk);
	uint8_t *out = slots + b * (long long)slot;
	const uint32_t c = deflate_block(src, n, out + 18, slot - 26, fast, kBgzfHashBits, (uint32_t *)(work + (size_t)b * work_stride));
	// the member around it: gzip header with the BC extra field (BSIZE = member size - 1), CRC32 and ISIZE of the uncompressed bytes (SAMv1 4.1)
	const uint32_t bsize = c + 18 + 8 - 1;
	const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
	for (int i = 0; i < 16; ++i) out[i] = hdr[i];
	out[16] = (uint8_t)bsize, out[17] = (uint8_t)(bsize >> 8);
	uint32_t crc = 0xffffffffu;
	{
		uint32_t i = 0;
		for (; i + 16 <= n; i += 16) {                                        // sixteen bytes per load: the lane waits for every load it issues
			uint32_t v[4];
			__builtin_memcpy(v, src + i, 16);
#pragma unroll
			for (int k = 0; k < 16; ++k) crc = crc_tab[(crc ^ (v[k >> 2] >> (8 * (k & 3)))) & 0xffu] ^ (crc >> 8);
		}
		for (; i < n; ++i) crc = crc_tab[(crc ^ src[i]) & 0xffu] ^ (crc >> 8);
	}
	crc = ~crc;
	uint8_t *t = out + 18 + c;
	for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i)), t[4 + i] = (uint8_t)(n >> (8 * i));
	len[b] = c ? (int32_t)(c + 26) : 0;
}
// the members side by side: a workgroup per block
__global__ __launch_bounds__(256) void k_bgzf_pack(const uint8_t *slots, uint32_t slot, const int32_t *len, const long long *off, uint8_t *packed)
{
	const long long b = blockIdx.x;
	const uint8_t *s = slots + b * (long long)slot;
	uint8_t *d = packed + off[b];
	const int n = len[b];
	for (int i = threadIdx.x; i < n; i += 256) d[i] = s[i];
}

struct BgzfCtx {
	std::mutex mu;
	int device = -1;
	DevBuf in, slots, work, len, off, packed;
	hipStream_t stream = nullptr;
};
static BgzfCtx &bgzf_ctx() { static BgzfCtx c; return c; }

} // namespace psvr

using namespace psvr;

// input bytes per member: BGZF allows anything up to 64 KB.  A call lasts as long as ONE block takes its lane, however many blocks it holds, and
// the chip has 16 k lanes per wavefront slot: 16 KB blocks (PSVR_BGZF_BLOCK=<bytes>) are a quarter of htslib's 0xff00, four times the lanes
static uint32_t bgzf_block_bytes()
{
	static const uint32_t v = [] { const char *e = getenv("PSVR_BGZF_BLOCK"); long x = e ? atol(e) : 0x4000; return (uint32_t)(x < 256 ? 256 : x > (long)kDfMaxIn ? (long)kDfMaxIn : x); }();
	return v;
}
extern "C" int64_t psvr_bgzf_bound(int64_t n_bytes) { const int64_t blk = bgzf_block_bytes(), nb = (n_bytes + blk - 1) / blk; return n_bytes + nb * 64 + 64; }

extern "C" int psvr_bgzf_compress(int device, const void *in, int64_t n_bytes, void *out, int64_t out_cap, int64_t *out_bytes)
{
	if (!in || !out || !out_bytes || n_bytes < 0) return set_error(PSVR_ERR_ARG, "psvr_bgzf_compress: bad argument");
	*out_bytes = 0;
	if (n_bytes == 0) return PSVR_OK;
	if (psvr_device_count() <= 0) return set_error(PSVR_ERR_DEVICE, "no HIP device visible: the engine has no CPU path");
	BgzfCtx &c = bgzf_ctx();
	std::lock_guard<std::mutex> lk(c.mu);
	PSVR_HIP(hipSetDevice(device));
	if (c.device != device) {
		c.in.release(), c.slots.release(), c.work.release(), c.len.release(), c.off.release(), c.packed.release();
		if (c.stream) (void)hipStreamDestroy(c.stream), c.stream = nullptr;
		c.device = device;
		PSVR_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
		PSVR_HIP(hipFuncSetAttribute((const void *)k_bgzf_deflate, hipFuncAttributeMaxDynamicSharedMemorySize, 1024 + 64 * (int)kBgzfLaneLds));
	}
	const uint32_t blk = bgzf_block_bytes(), slot = blk + 64;             // (a member never exceeds its input by more than the stored block's 5 + 26 bytes)
	const long long nb = (n_bytes + blk - 1) / blk;
	const uint32_t wstride = (blk * 4u + 16u + 255u) & ~255u;             // the tokens: a word per input byte at most (+ a group of four)
	PSVR_HIP(c.in.ensure((size_t)n_bytes + 64));
	PSVR_HIP(c.slots.ensure((size_t)nb * slot));
	PSVR_HIP(c.work.ensure((size_t)nb * wstride));
	PSVR_HIP(c.len.ensure((size_t)nb * 4));
	PSVR_HIP(c.off.ensure((size_t)nb * 8));
	PSVR_HIP(c.packed.ensure((size_t)psvr_bgzf_bound(n_bytes)));
	PSVR_HIP(hipMemcpyAsync(c.in.p, in, (size_t)n_bytes, hipMemcpyHostToDevice, c.stream));
	hipLaunchKernelGGL(k_bgzf_deflate, dim3((unsigned)((nb + 63) / 64)), dim3(64), (size_t)1024 + (size_t)64 * kBgzfLaneLds, c.stream, c.in.as<uint8_t>(), (long long)n_bytes, nb, blk, slot,
	                   c.slots.as<uint8_t>(), c.work.as<uint8_t>(), wstride, c.len.as<int32_t>());
	PSVR_HIP(hipGetLastError());
	std::vector<int32_t> len((size_t)nb);
	PSVR_HIP(hipMemcpyAsync(len.data(), c.len.p, (size_t)nb * 4, hipMemcpyDeviceToHost, c.stream));
	PSVR_HIP(hipStreamSynchronize(c.stream));
	std::vector<long long> off((size_t)nb);
	long long total = 0;
	for (long long b = 0; b < nb; ++b) { if (len[(size_t)b] <= 0) return set_error(PSVR_ERR_DEVICE, "psvr_bgzf_compress: block %lld did not fit its member", b); off[(size_t)b] = total, total += len[(size_t)b]; }
	if (total > out_cap) return set_error(PSVR_ERR_OVERFLOW, "psvr_bgzf_compress: need %lld bytes, have %lld", total, (long long)out_cap);
	PSVR_HIP(hipMemcpyAsync(c.off.p, off.data(), (size_t)nb * 8, hipMemcpyHostToDevice, c.stream));
	hipLaunchKernelGGL(k_bgzf_pack, dim3((unsigned)nb), dim3(256), 0, c.stream, c.slots.as<uint8_t>(), slot, c.len.as<int32_t>(), c.off.as<long long>(), c.packed.as<uint8_t>());
	PSVR_HIP(hipGetLastError());
	PSVR_HIP(hipMemcpyAsync(out, c.packed.p, (size_t)total, hipMemcpyDeviceToHost, c.stream));
	PSVR_HIP(hipStreamSynchronize(c.stream));
	*out_bytes = total;
	return PSVR_OK;
}
