#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define OPS(X) \
 X(0, "v_add_u32", 1, asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(1, "v_sub_u32", 1, asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(2, "v_or_b32", 1, asm volatile("v_or_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(3, "v_xor_b32", 1, asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(4, "v_lshlrev_b32 imm", 1, asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(5, "v_lshrrev_b32 imm", 1, asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(6, "v_ashrrev_i32 imm", 1, asm volatile("v_ashrrev_i32_e32 %0, 1, %0" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(7, "v_bfe_i32", 1, asm volatile("v_bfe_i32 %0, %0, 4, 8" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(8, "v_bfe_u32", 1, asm volatile("v_bfe_u32 %0, %0, 4, 8" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(9, "v_min_i32", 1, asm volatile("v_min_i32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(10, "v_max_u32", 1, asm volatile("v_max_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(11, "v_max3_i32", 1, asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(12, "v_lshl_or_b32", 1, asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(13, "v_lshl_add_u32", 1, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(14, "v_and_or_b32", 1, asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(15, "v_or3_b32", 1, asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(16, "v_mov_b32", 1, asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(17, "v_mov_dpp row_shr1", 1, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(18, "v_mov_dpp quad_perm", 1, asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(19, "v_add_u32 inline const", 1, asm volatile("v_add_u32_e32 %0, 7, %0" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(20, "v_add_u32 literal", 1, asm volatile("v_add_u32_e32 %0, 0x12345, %0" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(21, "v_cmp_gt_i32 vcc", 1, asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(22, "v_cmp_gt_i32 sgpr", 1, asm volatile("v_cmp_gt_i32_e64 s[20:21], %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(23, "v_cmp_eq_u32 vcc", 1, asm volatile("v_cmp_eq_u32_e32 vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(24, "cmp+cndmask", 2, asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(25, "cmp+cndmask sgpr", 2, asm volatile("v_cmp_gt_i32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(26, "cmp + 4 nops + cndmask", 2, asm volatile("v_cmp_gt_i32_e32 vcc, %0, %1\n s_nop 3\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(27, "v_pk_max_i16", 1, asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(28, "v_perm_b32", 1, asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(29, "v_mad_u32_u24", 1, asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(30, "v_mul_u32_u24", 1, asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(31, "v_readfirstlane", 1, asm volatile("v_readfirstlane_b32 s20, %0" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(32, "s_add_u32 (salu)", 1, asm volatile("s_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(33, "v_add + s_add pair", 2, asm volatile("v_add_u32_e32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(34, "v_max + s_add pair", 2, asm volatile("v_max_i32_e32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21")) \
 X(35, "v_add x2 chain", 2, asm volatile("v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc", "s20", "s21"))
template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed)
{
	unsigned a[8];
	for (int i = 0; i < 8; ++i) a[i] = seed * (i + 1) + threadIdx.x;
	unsigned b = seed ^ 0x10203 ^ threadIdx.x;
	unsigned sb = seed;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < REP; ++r) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
#define X(n, name, cnt, code) if (OP == n) code;
				OPS(X)
#undef X
			}
		}
	}
	unsigned s = 0;
	for (int i = 0; i < 8; ++i) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> void run(const char *name, unsigned *d, int cnt)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
	const int blocks = 256 * 8, iters = 200;
	k<OP><<<blocks, 256>>>(d, 10, 1);
	(void)hipEventRecord(e0);
	k<OP><<<blocks, 256>>>(d, iters, 1);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	double winstr = (double)blocks * 4 * iters * REP * 8 * cnt;     // wave instructions
	printf("%-26s %8.3f ms  %7.2f G wave-instr/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, ms, winstr / ms / 1e6, 1024 * 2.4e9 / (winstr / (ms * 1e-3)));
}
int main()
{
	unsigned *d;
	(void)hipMalloc(&d, 256 * 8 * 256 * 4);
#define X(n, name, cnt, code) run<n>(name, d, cnt);
	OPS(X)
#undef X
	return 0;
}
