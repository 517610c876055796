// tools/micro/membench.hip -- what the CU's vector-memory pipeline gives a persistent streaming kernel on gfx950:
// read-only streams with D x 1 KiB in flight per wave, W waves per CU, by cache policy, into VGPRs or by LDS-DMA,
// with and without dependent 16-bit table look-ups (the access mix of k_encode).  Measurement aid, not product.
//   hipcc --offload-arch=gfx950 -O3 -o membench membench.hip && ./membench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave streams chunks of D KiB (D b128 loads per lane); chunk c of wave w: (c * nwaves + w) * D KiB
template <int D, int AUX, int NG>
__global__ __launch_bounds__(256) void k_stream(const uint32_t* __restrict__ src, size_t bytes, const uint16_t* __restrict__ lut,
                                                uint32_t* __restrict__ sink)
{
	const int lane = threadIdx.x & 63;
	const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
	const size_t wid = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
	const size_t chunk = (size_t)D * 1024;
	const size_t nchunks = bytes / chunk;
	const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)lut, 0, 1 << 26, 0x00020000);
	uint32_t acc = 0;
	uint4 cur[D], nxt[D];
	size_t c = wid;
	auto load = [&](size_t cc, uint4 (&dst)[D]) {
		const uint8_t* base = (const uint8_t*)src + cc * chunk;
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)chunk, 0x00020000);
#pragma unroll
		for (int i = 0; i < D; i++) dst[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + i * 1024, 0, AUX));
	};
	if (c < nchunks) load(c, nxt);
	for (; c < nchunks; c += nwaves) {
#pragma unroll
		for (int i = 0; i < D; i++) cur[i] = nxt[i];
		if (NG) {      // NG look-ups per loaded dword, issued BEFORE the next chunk's loads (as k_encode does)
			uint32_t e[D * 4];
#pragma unroll
			for (int i = 0; i < D; i++) {
				const uint32_t v[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
				for (int k = 0; k < 4; k++) e[i * 4 + k] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(lrs, (v[k] & 0x1FFFFFFu) << 1, 0, 0);
			}
			if (c + nwaves < nchunks) load(c + nwaves, nxt);
#pragma unroll
			for (int i = 0; i < D * 4; i++) acc += e[i];
		} else {
			if (c + nwaves < nchunks) load(c + nwaves, nxt);
#pragma unroll
			for (int i = 0; i < D; i++) acc ^= cur[i].x ^ cur[i].y ^ cur[i].z ^ cur[i].w;
		}
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

// LDS-DMA stream: each wave owns R ring slots of 1 KiB * D in LDS; loads land there (no VGPRs), consumed by ds_read
template <int D, int R, int AUX>
__global__ __launch_bounds__(256) void k_stream_lds(const uint32_t* __restrict__ src, size_t bytes, uint32_t* __restrict__ sink)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
	const size_t wid = (size_t)blockIdx.x * (blockDim.x / 64) + wave;
	const size_t chunk = (size_t)D * 1024;
	const size_t nchunks = bytes / chunk;
	uint8_t* ring = smem + (size_t)wave * R * chunk;
	uint32_t acc = 0;
	auto issue = [&](size_t cc, int slot) {
		const uint8_t* base = (const uint8_t*)src + cc * chunk;
#pragma unroll
		for (int i = 0; i < D; i++)
			__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + i * 1024 + lane * 16),
			                                 (__attribute__((address_space(3))) void*)(ring + slot * chunk + i * 1024), 16, 0, AUX);
	};
	size_t c = wid;
	int head = 0;
	for (int r = 0; r < R - 1; r++) { if (c + r * nwaves < nchunks) issue(c + r * nwaves, r); }
	for (; c < nchunks; c += nwaves) {
		const size_t ahead = c + (size_t)(R - 1) * nwaves;
		if (ahead < nchunks) issue(ahead, (head + R - 1) % R);
		// wait until the oldest slot has landed: all but the (R-1)*D youngest
		if (ahead < nchunks) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((R - 1) * D) : "memory"); }
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
		for (int i = 0; i < D; i++) {
			const uint4 v = *(const uint4*)(ring + head * chunk + i * 1024 + lane * 16);
			acc ^= v.x ^ v.y ^ v.z ^ v.w;
		}
		head = (head + 1) % R;
	}
	if (acc == 0x12345678u) sink[0] = acc;
}


// k_encode's access shape: an item = 64 consecutive 4x4 blocks of one block row (256 px x 4 rows) of a W-pixel-wide frame, lane = (block jb of
// 16, row prow), four b128 loads per item (each reads 4 row segments of 16 adjacent blocks) and 16 look-ups per lane through the
// sparse cube-tiled table index of agmv_hip.hip.  NG = 0: no look-ups; 1: look-ups issued, then the next item's loads; 2: the next
// item's loads first.  SKIP: every SKIP-th item does its look-ups (the others reuse) -- what skipping buys.
__device__ __forceinline__ uint32_t lut_offset(uint32_t px)
{
	const uint32_t m = __umul24(px & 0x030303u, 0x10410u);
	const uint32_t hi = px & 0xFCFCFCu;
	uint32_t lo, off;
	asm("v_bfe_u32 %0, %1, 15, 7" : "=v"(lo) : "v"(m));
	asm("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(off) : "v"(hi), "v"(lo));
	return off;
}
template <int NG, int AUX, int LANES, uint32_t MASK = 0xFFFFFFFFu, int BUSY = 0, int LATE = 0>
__global__ __launch_bounds__(256) void k_items(const uint32_t* __restrict__ src, uint32_t W, uint32_t H, uint32_t T, const uint16_t* __restrict__ lut,
                                               uint32_t* __restrict__ sink)
{
	const int lane = threadIdx.x & 63;
	const uint32_t jb = lane >> 2, prow = lane & 3;
	const size_t nwaves = (size_t)gridDim.x * (blockDim.x / 64);
	const size_t wid = (size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
	const uint32_t ipr = W / 256, ipf = ipr * (H / 4);          // items per block row, per frame
	const size_t nitems = (size_t)ipf * T;
	const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)lut, 0, 1 << 29, 0x00020000);
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
	uint32_t acc = 0;
	uint4 px[4];
	auto load = [&](size_t it) {
		const uint32_t f = (uint32_t)(it / ipf), r = (uint32_t)(it % ipf), by = r / ipr, ix = r % ipr;
		const uint32_t base = ((f * H + by * 4 + prow) * W + ix * 256 + jb * 4) * 4u;      // < 2^31 for the sizes used
#pragma unroll
		for (int i = 0; i < 4; i++) px[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, base + 256 * i, 0, AUX));
	};
	size_t it = wid;
	if (it < nitems) load(it);
	for (; it < nitems; it += nwaves) {
		if (NG) {
			uint32_t e[16];
			const bool on = lane < LANES;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t v[4] = {px[i].x, px[i].y, px[i].z, px[i].w};
#pragma unroll
				for (int k = 0; k < 4; k++) { e[i * 4 + k] = 0; if (on) e[i * 4 + k] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(lrs, lut_offset(v[k]) & MASK, 0, 0); }
			}
			asm volatile("" ::: "memory");
			if (it + nwaves < nitems) load(it + nwaves);
#pragma unroll
			for (int i = 0; i < 16; i++) acc += e[i];
		} else {
			uint32_t x = px[0].x ^ px[1].y ^ px[2].z ^ px[3].w ^ px[0].y ^ px[1].z ^ px[2].w ^ px[3].x ^ px[0].z ^ px[1].w ^ px[2].x ^ px[3].y ^ px[0].w ^ px[1].x ^ px[2].y ^ px[3].z;
			if (!LATE && it + nwaves < nitems) load(it + nwaves);
#pragma unroll 16
			for (int b = 0; b < BUSY; b++) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(x));      // a dependent chain: ~5 clocks each for this wave
			if (LATE && it + nwaves < nitems) load(it + nwaves);
			acc ^= x;
		}
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

static float time_it(void (*launch)(void*), void* arg, int reps = 5)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	launch(arg); CK(hipDeviceSynchronize());
	float best = 1e9f;
	for (int i = 0; i < reps; i++) {
		CK(hipEventRecord(a)); launch(arg); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (ms < best) best = ms;
	}
	return best;
}

struct Ctx { const uint32_t* src; size_t bytes; const uint16_t* lut; uint32_t* sink; int wpc, ncu; const uint32_t* img; };

template <int D, int AUX, int NG> static void l_stream(void* p)
{
	Ctx* c = (Ctx*)p;
	const int wg = c->ncu * ((c->wpc + 3) / 4);    // 256-thread workgroups: wpc/4 per CU
	hipLaunchKernelGGL((k_stream<D, AUX, NG>), dim3(wg), dim3(256), 0, 0, c->src, c->bytes, c->lut, c->sink);
}
template <int D, int R, int AUX> static void l_lds(void* p)
{
	Ctx* c = (Ctx*)p;
	const int wg = c->ncu * ((c->wpc + 3) / 4);
	const size_t lds = (size_t)4 * R * D * 1024;
	CK(hipFuncSetAttribute((const void*)k_stream_lds<D, R, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL((k_stream_lds<D, R, AUX>), dim3(wg), dim3(256), lds, 0, c->src, c->bytes, c->sink);
}

template <int NG, int AUX, int LANES, uint32_t MASK = 0xFFFFFFFFu, int BUSY = 0, int LATE = 0> static void l_items(void* p)
{
	Ctx* c = (Ctx*)p;
	const int wg = c->ncu * ((c->wpc + 3) / 4);
	hipLaunchKernelGGL((k_items<NG, AUX, LANES, MASK, BUSY, LATE>), dim3(wg), dim3(256), 0, 0, c->img, 1792u, 1080u, 256u, c->lut, c->sink);
}

int main(int argc, char** argv)
{
	// groups: streams | items | busy   (default: all)
	const char* sel = argc > 1 ? argv[1] : "all";
	auto want = [&](const char* g) { return !strcmp(sel, "all") || !strcmp(sel, g); };
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	const size_t bytes = (size_t)2 << 30;
	uint32_t *src, *sink; uint16_t* lut;
	CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&lut, (size_t)1 << 29));
	{
		std::vector<uint32_t> h(bytes / 4);
		for (size_t i = 0; i < h.size(); i++) { uint32_t x = (uint32_t)(i % 1920), y = (uint32_t)((i / 1920) % 1080), t = (uint32_t)(i / (1920 * 1080)); h[i] = ((x / 8 + 2 * t) & 255) << 16 | ((y / 4 + t) & 255) << 8 | (((x + y) / 16 + 3 * t) & 255); }
		CK(hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice));
	}
	CK(hipMemset(lut, 1, (size_t)1 << 29));
	Ctx c{src, bytes, lut, sink, 16, ncu, nullptr};
	const int wpcs[] = {4, 8, 16, 24, 32};
	printf("%d CUs; TB/s by waves per CU (columns)\n", ncu);
#define ROW(name, fn) do { printf("%-44s", name); for (int w : wpcs) { c.wpc = w; float ms = time_it(fn, &c); printf(" %2d:%5.2f", w, bytes / ms / 1e9); } printf("\n"); fflush(stdout); } while (0)
	if (want("streams")) {
		printf("-- read-only stream of %.1f GB, D KiB in flight per wave\n", bytes / 1e9);
		ROW("vgpr D=1 default", (l_stream<1, 0, 0>));
		ROW("vgpr D=2 default", (l_stream<2, 0, 0>));
		ROW("vgpr D=4 default", (l_stream<4, 0, 0>));
		ROW("vgpr D=8 default", (l_stream<8, 0, 0>));
		ROW("vgpr D=4 nt", (l_stream<4, 2, 0>));
		ROW("vgpr D=8 nt", (l_stream<8, 2, 0>));
		ROW("vgpr D=4 sc1 nt", (l_stream<4, 18, 0>));
		ROW("vgpr D=4 nt + 16 look-ups/lane (linear table)", (l_stream<4, 2, 1>));
		ROW("lds-dma D=4 R=2 default", (l_lds<4, 2, 0>));
		ROW("lds-dma D=4 R=3 default", (l_lds<4, 3, 0>));
		ROW("lds-dma D=4 R=2 nt", (l_lds<4, 2, 2>));
		ROW("lds-dma D=4 R=3 nt", (l_lds<4, 3, 2>));
		ROW("lds-dma D=8 R=2 nt", (l_lds<8, 2, 2>));
	}
	// ---- k_encode-shaped items: 256 frames of 1792 x 1080 (7 items per block row)
	if (want("items") || want("busy")) {
		const uint32_t W = 1792, H = 1080, T = 256;
		const size_t n = (size_t)W * H * T;
		uint32_t* img; CK(hipMalloc(&img, n * 4));
		std::vector<uint32_t> h(n);
		const size_t ibytes = n * 4;
#define IROW(name, fn) do { printf("%-44s", name); for (int w : wpcs) { c.wpc = w; float ms = time_it(fn, &c); printf(" %2d:%5.2f", w, ibytes / ms / 1e9); } printf("\n"); fflush(stdout); } while (0)
		for (int kind = 0; kind < 3 && want("items"); kind++) {
			uint64_t z = 12345;
			for (size_t i = 0; i < n; i++) {
				uint32_t x = (uint32_t)(i % W), y = (uint32_t)((i / W) % H), t = (uint32_t)(i / ((size_t)W * H));
				uint32_t r = (x * 255 / (W - 1) + 2 * t) & 255, g = (y * 255 / (H - 1) + t) & 255, b = ((x + y) / 2 + 3 * t) & 255;
				uint32_t v = r << 16 | g << 8 | b;
				if (kind == 0) v = 0x336699;
				if (kind == 2) { z = z * 6364136223846793005ull + 1442695040888963407ull; v ^= (uint32_t)(z >> 33) & 0x070707u; }
				h[i] = v;
			}
			CK(hipMemcpy(img, h.data(), ibytes, hipMemcpyHostToDevice));
			c.img = img;
			printf("-- items, %s (TB/s of pixel bytes)\n", kind == 0 ? "flat" : kind == 1 ? "gradient" : "gradient^noise3");
			IROW("items nt, no look-ups", (l_items<0, 2, 64>));
			IROW("items nt, 16 look-ups/lane", (l_items<1, 2, 64>));
			IROW("items nt, 16 look-ups, table masked to 16 KB", (l_items<1, 2, 64, 0x3FFEu>));
			IROW("items nt, 16 look-ups, table masked to 2 MB", (l_items<1, 2, 64, 0x1FFFFEu>));
			IROW("items default policy, 16 look-ups", (l_items<1, 0, 64>));
		}
		if (want("busy")) {
			for (size_t i = 0; i < n; i++) h[i] = 0x336699;
			CK(hipMemcpy(img, h.data(), ibytes, hipMemcpyHostToDevice));
			c.img = img;
			printf("-- items, flat, no look-ups, B dependent VALU ops of busy work per item; early = next loads issued before the work, late = after\n");
			IROW("B=0", (l_items<0, 2, 64, 0xFFFFFFFFu, 0, 0>));
			IROW("B=200 early", (l_items<0, 2, 64, 0xFFFFFFFFu, 200, 0>));
			IROW("B=200 late", (l_items<0, 2, 64, 0xFFFFFFFFu, 200, 1>));
			IROW("B=500 early", (l_items<0, 2, 64, 0xFFFFFFFFu, 500, 0>));
			IROW("B=500 late", (l_items<0, 2, 64, 0xFFFFFFFFu, 500, 1>));
			IROW("B=1000 early", (l_items<0, 2, 64, 0xFFFFFFFFu, 1000, 0>));
			IROW("B=1000 late", (l_items<0, 2, 64, 0xFFFFFFFFu, 1000, 1>));
		}
	}
	return 0;
}
