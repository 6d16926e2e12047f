// Host-visible completion latency of a tiny readback: (a) kernel + hipMemcpyAsync(D2H, pinned) + hipStreamSynchronize, against
// (b) a kernel that writes its words and a sequence flag straight into mapped page-locked host memory while the host spins on the flag.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_work(unsigned long long *d, unsigned long long v) { if (threadIdx.x == 0) d[0] = v; }
__global__ void k_publish(const unsigned long long *d, volatile unsigned long long *h, unsigned long long seq)
{
	if (threadIdx.x < 8) h[threadIdx.x] = d[0] + threadIdx.x;
	__threadfence_system();
	if (threadIdx.x == 0) h[8] = seq;
}
int main()
{
	hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	unsigned long long *d, *h;
	(void)hipMalloc(&d, 64); (void)hipHostMalloc(&h, 4096, hipHostMallocDefault);
	const int N = 2000;
	for (int mode = 0; mode < 2; ++mode) {
		h[8] = 0;
		auto t0 = std::chrono::steady_clock::now();
		unsigned long long sum = 0;
		for (int i = 1; i <= N; ++i) {
			hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, s, d, (unsigned long long)i);
			if (mode == 0) { (void)hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); sum += h[0]; }
			else {
				hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, d, (volatile unsigned long long *)h, (unsigned long long)i);
				while (((volatile unsigned long long *)h)[8] != (unsigned long long)i) { }
				sum += h[0];
			}
		}
		const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
		printf("%s: %.1f us per round trip (check %llu)\n", mode == 0 ? "kernel + memcpyAsync + streamSynchronize" : "kernel + publish kernel + host spin", us, sum);
	}
	(void)hipStreamSynchronize(s);
	return 0;
}
