// tools/micro/fetchcal.hip -- calibration of rocprofv3's FETCH_SIZE for the access widths of k_encode (MI355X_MICROARCH.md:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Over one 2 GiB
// buffer (far beyond the 256 MiB Infinity Cache), once each:
//   k_cal<0>  16 bytes per lane, coalesced (the pixel stream)          -> B bytes, known
//   k_cal<1>  one 16-bit load per 128-byte line (a table look-up that misses)   -> B / 128 lines
//   k_cal<2>  one 16-bit load per 64 bytes  (two per line)
//   k_cal<3>  one 16-bit load per 32 bytes  (four per line)
// Event times tell how many bytes the memory really moved per line (against the stream's TB/s), the FETCH_SIZE of the same
// kernels under `rocprofv3 --pmc FETCH_SIZE` how they are tallied.  Measurement aid, not product.
//   hipcc --offload-arch=gfx950 -O3 -o fetchcal fetchcal.hip && ./fetchcal
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_cal(const uint8_t* __restrict__ src, size_t bytes, uint32_t* __restrict__ sink)
{
	const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)gridDim.x * 256;
	uint32_t acc = 0;
	if (MODE == 0) {
		for (size_t o = gid * 16; o + 16 <= bytes; o += n * 16) { const uint4 v = *(const uint4*)(src + o); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
	} else {
		const size_t step = MODE == 1 ? 128 : (MODE == 2 ? 64 : 32);
		for (size_t o = gid * step; o + 2 <= bytes; o += n * step) acc += *(const uint16_t*)(src + o + 2 * (gid & 7));
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
	const size_t B = (size_t)2 << 30;
	uint8_t* d; uint32_t* sink;
	CK(hipMalloc(&d, B)); CK(hipMalloc(&sink, 4));
	CK(hipMemset(d, 1, B));
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	const char* names[4] = {"16 B per lane, coalesced", "u16 per 128-byte line", "u16 per 64 bytes", "u16 per 32 bytes"};
	for (int m = 0; m < 4; m++) {
		float best = 1e9f;
		for (int rep = 0; rep < 3; rep++) {
			CK(hipEventRecord(a));
			const int grid = 256 * 32;
			if (m == 0) hipLaunchKernelGGL(k_cal<0>, dim3(grid), dim3(256), 0, 0, d, B, sink);
			if (m == 1) hipLaunchKernelGGL(k_cal<1>, dim3(grid), dim3(256), 0, 0, d, B, sink);
			if (m == 2) hipLaunchKernelGGL(k_cal<2>, dim3(grid), dim3(256), 0, 0, d, B, sink);
			if (m == 3) hipLaunchKernelGGL(k_cal<3>, dim3(grid), dim3(256), 0, 0, d, B, sink);
			CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
			float ms; CK(hipEventElapsedTime(&ms, a, b));
			if (ms < best) best = ms;
		}
		printf("k_cal<%d> %-28s %8.3f ms   = %6.2f TB/s if all %zu MiB moved   (%zu loads of 2 bytes)\n", m, names[m], best, B / (best * 1e-3) / 1e12,
		       B >> 20, m == 0 ? (size_t)0 : B / (m == 1 ? 128 : (m == 2 ? 64 : 32)));
	}
	return 0;
}
