// tools/micro/scratchcheck.hip -- does a kernel that keeps values in scratch memory (private segment) read back what it wrote?
// Round 3 saw k_fp_walk give sporadically wrong results with a 12-byte spill (one-wave workgroups, amdgpu_waves_per_eu(8, 8),
// > 100 k workgroups per launch, next to PyTorch's HIP runtime in the process); the same source without the spill was always
// right.  This is the smallest stand-alone form of that situation: every lane parks 16 values in a dynamically indexed
// private array (= scratch), streams some global memory, and checks them.  Prints the number of mismatching lanes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_check(const uint32_t* __restrict__ src, size_t n, uint32_t* bad, uint32_t salt)
{
	volatile uint32_t park[16];                                // private: scratch
	const uint32_t id = blockIdx.x * 64 + threadIdx.x;
	for (int k = 0; k < 16; k++) park[(k + id) & 15] = id * 2654435761u + (uint32_t)k + salt;
	uint32_t acc = 0;
	for (size_t i = id; i < n; i += (size_t)gridDim.x * 64) acc += src[i];
	uint32_t wrong = 0;
	for (int k = 0; k < 16; k++) wrong |= park[(k + id) & 15] ^ (id * 2654435761u + (uint32_t)k + salt);
	if (wrong) atomicAdd(bad, 1u);
	if (acc == 0x12345678u) bad[1] = acc;
}

int main()
{
	const size_t n = (size_t)64 << 20;
	uint32_t *src, *bad;
	CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&bad, 8));
	CK(hipMemset(src, 1, n * 4));
	uint32_t total = 0;
	for (int rep = 0; rep < 200; rep++) {
		CK(hipMemset(bad, 0, 8));
		hipLaunchKernelGGL(k_check, dim3(111000), dim3(64), 0, 0, src, n, bad, (uint32_t)rep);
		CK(hipDeviceSynchronize());
		uint32_t h = 0;
		CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
		total += h;
	}
	printf("scratchcheck: %u mismatching lanes in 200 launches of 111000 one-wave workgroups\n", total);
	return 0;
}
