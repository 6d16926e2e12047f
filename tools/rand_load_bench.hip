// Divergent-load ceiling of the vector memory pipeline: every lane loads one dword from a pseudo-random index of a table of a given size.
// usage: rand_load_bench            (prints G loads/s for table sizes 64 KiB .. 2 GiB, 4-byte and 16-byte loads, 1 or 4 loads in flight per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int N, class T> __global__ __launch_bounds__(256) void k_rand(const T *tab, uint32_t mask, uint32_t *out, int rounds)
{
	uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
	uint32_t acc = 0;
	for (int r = 0; r < rounds; ++r) {
		uint32_t idx[N];
#pragma unroll
		for (int k = 0; k < N; ++k) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; idx[k] = x & mask; }
#pragma unroll
		for (int k = 0; k < N; ++k) { const T v = tab[idx[k]]; acc += *(const uint32_t *)&v; }
	}
	if (acc == 0x12345678u) out[0] = acc;
}

template <int N, class T> static double run(const T *tab, size_t n_el, uint32_t *out)
{
	const uint32_t mask = (uint32_t)(n_el - 1);
	const int blocks = 256 * 32, rounds = 64 / N * 4;
	hipEvent_t a, b;
	(void)hipEventCreate(&a), (void)hipEventCreate(&b);
	hipLaunchKernelGGL((k_rand<N, T>), dim3(blocks), dim3(256), 0, 0, tab, mask, out, rounds);
	(void)hipEventRecord(a, 0);
	for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_rand<N, T>), dim3(blocks), dim3(256), 0, 0, tab, mask, out, rounds);
	(void)hipEventRecord(b, 0);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	return 5.0 * blocks * 256.0 * rounds * N / (ms * 1e-3) / 1e9;
}

int main()
{
	const size_t maxb = (size_t)2 << 30;
	void *tab; uint32_t *out;
	if (hipMalloc(&tab, maxb) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
	(void)hipMemset(tab, 1, maxb);
	printf("%12s %14s %14s %14s %14s\n", "table", "4B x1 G/s", "4B x4 G/s", "16B x1 G/s", "16B x4 G/s");
	for (size_t bytes = (size_t)64 << 10; bytes <= maxb; bytes <<= 2) {
		const double a = run<1, uint32_t>((const uint32_t *)tab, bytes / 4, out), b = run<4, uint32_t>((const uint32_t *)tab, bytes / 4, out);
		const double c = run<1, uint4>((const uint4 *)tab, bytes / 16, out), d = run<4, uint4>((const uint4 *)tab, bytes / 16, out);
		printf("%9zu KiB %14.1f %14.1f %14.1f %14.1f\n", bytes >> 10, a, b, c, d);
		fflush(stdout);
	}
	return 0;
}
