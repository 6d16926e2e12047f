// pinned_read.hip -- what the CPU pays to READ page-locked host memory on the GPU box, by the way it was allocated: a streaming memcpy of a
// 1.4 MB readback and 200 k dependent 4-byte loads at random places.  (profiles/r03f_pinned_read.txt)
//   hipcc --offload-arch=gfx950 -O2 -o build/pinned_read tools/microbench/pinned_read.hip && ./build/pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
	const size_t n = 1400 << 10, cap = 4 << 20;
	void *d;
	CK(hipMalloc(&d, n));
	CK(hipMemset(d, 1, n));
	void *pin, *pin2, *pin3;
	CK(hipHostMalloc(&pin, cap, hipHostMallocDefault));
	void *reg = aligned_alloc(4096, cap);
	memset(reg, 0, cap);
	CK(hipHostRegister(reg, cap, hipHostRegisterDefault));
	CK(hipHostMalloc(&pin2, cap, hipHostMallocNonCoherent));
	CK(hipHostMalloc(&pin3, cap, hipHostMallocPortable));
	void *plain = malloc(cap);
	memset(plain, 0, cap);
	std::vector<char> dst(n);
	hipStream_t s;
	CK(hipStreamCreate(&s));
	const char *names[5] = {"hipHostMalloc default", "malloc + hipHostRegister", "hipHostMalloc non-coherent", "hipHostMalloc portable", "malloc (pageable)"};
	void *src[5] = {pin, reg, pin2, pin3, plain};
	for (int rep = 0; rep < 2; ++rep)
		for (int k = 0; k < 5; ++k) {
			auto t0 = std::chrono::steady_clock::now();
			CK(hipMemcpyAsync(src[k], d, n, hipMemcpyDeviceToHost, s));
			CK(hipStreamSynchronize(s));
			auto t1 = std::chrono::steady_clock::now();
			memcpy(dst.data(), src[k], n);
			auto t2 = std::chrono::steady_clock::now();
			// dependent random loads (an LCG walks the buffer; the loaded value feeds the next index)
			const volatile int *v = (const volatile int *)src[k];
			unsigned x = 12345u;
			long long acc = 0;
			for (int i = 0; i < 200000; ++i) { x = x * 1664525u + 1013904223u + (unsigned)(acc & 1); acc += v[(x >> 8) % (unsigned)(n / 4)]; }
			auto t3 = std::chrono::steady_clock::now();
			printf("%-28s d2h+sync %7.1f us   memcpy %6.1f us (%.1f GB/s)   random 4-byte load %6.1f ns  [%lld]\n", names[k], std::chrono::duration<double, std::micro>(t1 - t0).count(),
			       std::chrono::duration<double, std::micro>(t2 - t1).count(), n / std::chrono::duration<double, std::micro>(t2 - t1).count() / 1e3,
			       std::chrono::duration<double, std::nano>(t3 - t2).count() / 200000, acc);
		}
	return 0;
}
