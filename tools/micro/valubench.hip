// tools/micro/valubench.hip -- VALU issue rate of one SIMD on gfx950 by waves per SIMD and instruction kind (integer ops of k_encode)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k_valu(uint32_t* out, int iters, unsigned long long* clk)
{
	uint32_t a[8];
#pragma unroll
	for (int i = 0; i < 8; i++) a[i] = threadIdx.x * (i + 1);
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 8; r++) {
#pragma unroll
			for (int i = 0; i < 8; i++) {
				if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
				if (KIND == 1) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
				if (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
				if (KIND == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
				if (KIND == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
				if (KIND == 5) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
				if (KIND == 6) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
				if (KIND == 7) asm volatile("v_bfe_u32 %0, %0, 5, 11" : "+v"(a[i]));
				if (KIND == 8) asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "vcc");
			}
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	uint32_t s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) s += a[i];
	if (s == 0x12345u) out[0] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int KIND> static void run(const char* name, uint32_t* out, unsigned long long* clk, int ncu)
{
	const int iters = 2000;
	for (int wps : {1, 2, 4, 8}) {        // waves per SIMD = blocks of 256 threads per CU
		hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
		hipLaunchKernelGGL(k_valu<KIND>, dim3(ncu * wps), dim3(256), 0, 0, out, iters, clk);
		CK(hipDeviceSynchronize());
		CK(hipEventRecord(a));
		hipLaunchKernelGGL(k_valu<KIND>, dim3(ncu * wps), dim3(256), 0, 0, out, iters, clk);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
		const double ninstr = (double)iters * 64 * (KIND == 8 ? 2 : 1);
		printf("%-22s %d wave(s)/SIMD: %.2f clk per instruction per wave, %.2f clk per instruction per SIMD; clock %.2f GHz\n", name, wps,
		       c / ninstr, c / ninstr / wps, c / (ms * 1e6));
	}
}

int main()
{
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	uint32_t* out; unsigned long long* clk;
	CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
	run<0>("v_add_u32", out, clk, prop.multiProcessorCount);
	run<1>("v_and_or_b32", out, clk, prop.multiProcessorCount);
	run<2>("v_alignbit_b32", out, clk, prop.multiProcessorCount);
	run<3>("v_mul_u32_u24", out, clk, prop.multiProcessorCount);
	run<4>("v_perm_b32", out, clk, prop.multiProcessorCount);
	run<5>("v_pk_min_u16", out, clk, prop.multiProcessorCount);
	run<6>("v_add_u32 dpp", out, clk, prop.multiProcessorCount);
	run<7>("v_bfe_u32", out, clk, prop.multiProcessorCount);
	run<8>("v_cmp + v_cndmask", out, clk, prop.multiProcessorCount);
	return 0;
}
