// Same-address device-scope atomic throughput: every lane / one lane per wavefront / one per workgroup adds to ONE counter (or to 3 counters in one line).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE> __global__ __launch_bounds__(256) void k_atomic(unsigned long long *ctr, unsigned long long *out, int per_thread)
{
	unsigned long long acc = 0;
	for (int r = 0; r < per_thread; ++r) {
		if (MODE == 0) acc += atomicAdd(ctr, 1ull);                                     // every lane
		else if (MODE == 1) { if ((threadIdx.x & 63) == 0) acc += atomicAdd(ctr, 64ull); } // one per wavefront
		else if (MODE == 2) { acc += atomicAdd(ctr, 1ull); acc += atomicAdd(ctr + 1, 1ull); acc += atomicAdd(ctr + 2, 1ull); }   // three counters of one line per lane
		else if (MODE == 3) { acc += atomicAdd(ctr + 32 * (blockIdx.x & 7), 1ull); }     // eight counters in different lines
	}
	if (acc == 0x123456789ull) out[0] = acc;
}
template <int MODE> static void run(const char *name, unsigned long long *ctr, unsigned long long *out, double per_thread_atomics)
{
	const int blocks = 8192, per = 4;
	hipEvent_t a, b;
	(void)hipEventCreate(&a), (void)hipEventCreate(&b);
	hipLaunchKernelGGL(k_atomic<MODE>, dim3(blocks), dim3(256), 0, 0, ctr, out, per);
	(void)hipEventRecord(a, 0);
	hipLaunchKernelGGL(k_atomic<MODE>, dim3(blocks), dim3(256), 0, 0, ctr, out, per);
	(void)hipEventRecord(b, 0);
	(void)hipEventSynchronize(b);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, a, b);
	const double n = (double)blocks * 256 * per * per_thread_atomics;
	printf("%-44s %8.3f ms  %8.2f G atomics/s  (%.2f ns each)\n", name, ms, n / (ms * 1e-3) / 1e9, ms * 1e6 / n);
}
int main()
{
	unsigned long long *ctr, *out;
	if (hipMalloc(&ctr, 4096) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
	(void)hipMemset(ctr, 0, 4096);
	run<0>("every lane, one counter", ctr, out, 1.0);
	run<1>("one lane per wavefront, one counter", ctr, out, 1.0 / 64);
	run<2>("every lane, three counters of one line", ctr, out, 3.0);
	run<3>("every lane, eight counters in eight lines", ctr, out, 1.0);
	return 0;
}
