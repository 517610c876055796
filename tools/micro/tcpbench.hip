// tools/micro/tcpbench.hip -- cost model of a 16-bit table look-up instruction in the CU's L1 (TCP) on gfx950, measured inside a
// streaming loop of k_encode's shape (4 x b128 pixel loads + 16 look-ups per wave-item): how the time depends on how the 64 lanes'
// addresses spread over 128-byte lines and inside a quad of lanes, and on lanes masked off.  Finding (profiles/r02): a quad whose
// four addresses lie within two consecutive dwords is ONE access; any other quad is served lane by lane, ~1 clock per active lane.  Patterns are computed from the lane id (footprint <= 16 KB: always L1 hits).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PAT, int NLK, int WIDE, int LMASK = 0>
__global__ __launch_bounds__(256) void k_pat(const uint32_t* __restrict__ src, size_t nitems, const uint16_t* __restrict__ lut, uint32_t* __restrict__ sink)
{
	const int lane = threadIdx.x & 63;
	const size_t nwaves = (size_t)gridDim.x * 4, wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void*)lut, 0, 1 << 20, 0x00020000);
	uint32_t acc = 0;
	uint4 px[4];
	auto load = [&](size_t it) {
		const uint8_t* base = (const uint8_t*)src + it * 4096;
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 4096, 0x00020000);
#pragma unroll
		for (int i = 0; i < 4; i++) px[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + i * 1024, 0, 2));
	};
	// byte offset of lane's look-up k under pattern PAT (k rotates the lines so that successive look-ups are not identical)
	auto addr = [&](int k, uint32_t salt) -> uint32_t {
		const uint32_t q = lane >> 2, l4 = lane & 3;
		uint32_t line, in;
		switch (PAT) {
		case 0: line = 0; in = 0; break;                                   // every lane the same address
		case 1: line = 0; in = lane; break;                                // one line, 64 different entries
		case 2: line = q; in = 0; break;                                   // a line per quad, same address inside the quad
		case 3: line = q; in = l4 * 7; break;                              // a line per quad, different entries inside the quad
		case 4: line = q * 2 + (l4 >> 1); in = l4; break;                  // two lines per quad
		case 5: line = lane; in = 0; break;                                // 64 lines: every lane its own
		case 6: line = lane & 15; in = lane >> 4; break;                   // 16 lines, but the 4 lanes of a quad on 4 different lines
		case 7: line = lane >> 4; in = lane & 15; break;                   // 4 lines, 16 consecutive lanes each
		case 8: line = q; in = l4 * 2; break;                              // quad inside 16 bytes (4 dwords)
		case 9: line = q; in = l4 * 4; break;                              // quad inside 32 bytes
		case 10: line = q; in = l4 * 8; break;                             // quad inside 64 bytes
		case 11: line = q; in = l4 * 16; break;                            // quad spread over the 128-byte line
		case 12: line = q; in = (l4 >> 1) * 16; break;                     // quad: two addresses, two lanes each, same line
		case 13: line = q * 2 + (l4 >> 1); in = 0; break;                  // quad: two addresses on two lines
		case 14: line = q; in = l4; break;                                 // quad: 4 consecutive u16 (8 bytes)
		case 15: line = q; in = l4 * 2 + 1; break;                         // quad inside 16 bytes, odd halves
		case 16: line = q; in = l4 >> 1; break;                            // u16 offsets {0,0,1,1}: one dword
		case 17: line = q; in = (l4 + 1) >> 1; break;                      // {0,1,1,2}: two dwords, 6 bytes
		case 18: line = q; in = (0x1203 >> (l4 * 4)) & 3; break;           // {3,0,2,1}: a permutation inside 8 bytes
		case 19: line = q; in = (l4 & 1) * 2; break;                       // {0,2,0,2}: two dwords of one 8-byte pair
		case 20: line = q; in = l4 + 2; break;                             // {2,3,4,5}: consecutive, crossing an 8-byte boundary
		case 21: line = q; in = (l4 >> 1) * 4; break;                      // {0,0,4,4}: two addresses 8 bytes apart
		case 22: line = q; in = l4 == 3 ? 1 : 0; break;                    // {0,0,0,1}
		case 23: line = q; in = l4 == 3 ? 8 : 0; break;                    // {0,0,0,8}: three lanes one address, one lane 16 bytes on
		case 24: line = q; in = (q & 1) ? (l4 >> 1) * 4 : l4; break;           // even quads {0,1,2,3} (fast), odd quads {0,0,4,4} (slow)
		case 25: line = q; in = (q & 3) == 0 ? (l4 >> 1) * 4 : l4; break;      // one slow quad per 16 lanes
		case 26: line = q; in = q == 0 ? (l4 >> 1) * 4 : l4; break;            // one slow quad per wave
		case 27: line = q; in = lane >= 32 ? (l4 >> 1) * 4 : l4; break;        // lanes 0-31 fast, 32-63 slow
		case 28: line = q; in = (q & 1) ? l4 * 16 : l4; break;                 // even quads fast, odd quads spread over the line
		default: line = 0; in = 0;
		}
		line = (line + 5u * (uint32_t)k + (salt & 1u)) & 127u;             // 128 lines = 16 KB footprint
		return line * 128u + in * 2u;
	};
	size_t it = wid;
	if (it < nitems) load(it);
	for (; it < nitems; it += nwaves) {
		uint32_t e[16];
		const uint32_t salt = px[0].x;                                     // dependence on the loaded data, as in the codec
#pragma unroll
		for (int k = 0; k < 16; k++) {
			e[k] = 0;
			const bool on = LMASK == 0 ? true : LMASK == 1 ? !(lane & 1) : LMASK == 2 ? !(lane & 3) : LMASK == 3 ? lane < 32 : LMASK == 4 ? !(lane & 4) : ((lane * 7 + k) % 3 == 0);
			if (k < NLK && on) {
				if (WIDE == 1) e[k] = __builtin_amdgcn_raw_buffer_load_b32(lrs, addr(k, salt) & ~3u, 0, 0);
				else if (WIDE == 2) { const auto v = __builtin_amdgcn_raw_buffer_load_b64(lrs, addr(k, salt) & ~7u, 0, 0); e[k] = (uint32_t)v[0] ^ (uint32_t)v[1]; }
				else if (WIDE == 3) { const auto v = __builtin_amdgcn_raw_buffer_load_b128(lrs, addr(k, salt) & ~15u, 0, 0); e[k] = (uint32_t)v[0] ^ (uint32_t)v[1] ^ (uint32_t)v[2] ^ (uint32_t)v[3]; }
				else e[k] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(lrs, addr(k, salt), 0, 0);
			}
		}
		asm volatile("" ::: "memory");
		const uint32_t x = px[0].y ^ px[1].y ^ px[2].z ^ px[3].w ^ px[1].x ^ px[2].x ^ px[3].x;
		if (it + nwaves < nitems) load(it + nwaves);
#pragma unroll
		for (int k = 0; k < 16; k++) acc += e[k];
		acc ^= x;
	}
	if (acc == 0x12345678u) sink[0] = acc;
}

template <int PAT, int NLK, int WIDE, int LMASK = 0> static float run(const uint32_t* src, size_t nitems, const uint16_t* lut, uint32_t* sink, int ncu, int wpc)
{
	hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	float best = 1e9f;
	for (int r = 0; r < 4; r++) {
		CK(hipEventRecord(a));
		hipLaunchKernelGGL((k_pat<PAT, NLK, WIDE, LMASK>), dim3(ncu * wpc / 4), dim3(256), 0, 0, src, nitems, lut, sink);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		if (r && ms < best) best = ms;
	}
	return best;
}

int main(int argc, char** argv)
{
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int ncu = prop.multiProcessorCount;
	const size_t bytes = (size_t)2 << 30, nitems = bytes / 4096;
	uint32_t *src, *sink; uint16_t* lut;
	CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&lut, 1 << 20));
	CK(hipMemset(src, 0, bytes)); CK(hipMemset(lut, 1, 1 << 20));
	printf("wave-item = 4 KB of pixels + N look-ups per lane; ns per wave-item per CU (and TB/s) at 16 and 32 waves per CU\n");
#define R(P, N, WD, name) do { for (int w : {16, 32}) { float ms = run<P, N, WD>(src, nitems, lut, sink, ncu, w); \
		printf("%-58s N=%2d w=%2d  %6.1f ns/item/CU  %5.2f TB/s\n", name, N, w, ms * 1e6 / (nitems / (double)ncu), bytes / ms / 1e9); } fflush(stdout); } while (0)
#define RM(P, N, M, name) do { for (int w : {16}) { float ms = run<P, N, 0, M>(src, nitems, lut, sink, ncu, w); \
		printf("%-58s N=%2d w=%2d  %6.1f ns/item/CU  %5.2f TB/s\n", name, N, w, ms * 1e6 / (nitems / (double)ncu), bytes / ms / 1e9); } fflush(stdout); } while (0)
	if (argc > 1) {                                                    // mixed instructions: is the fast path decided per quad or per instruction?
		RM(0, 0, 0, "no look-ups");
		RM(14, 16, 0, "P14 every quad {0,1,2,3} (fast)");
		RM(21, 16, 0, "P21 every quad {0,0,4,4} (slow)");
		RM(24, 16, 0, "P24 even quads fast, odd quads slow");
		RM(28, 16, 0, "P28 even quads fast, odd quads over the line");
		RM(25, 16, 0, "P25 one slow quad per 16 lanes");
		RM(26, 16, 0, "P26 one slow quad per wave");
		RM(27, 16, 0, "P27 lanes 0-31 fast, 32-63 slow");
		return 0;
	}
	RM(0, 0, 0, "no look-ups");
	RM(0, 16, 0, "P0 all lanes one address");
	RM(1, 16, 0, "P1 one line, 64 entries (u16 lane)");
	RM(7, 16, 0, "P7 4 lines of 16 consecutive lanes");
	RM(2, 16, 0, "P2 a line per quad, one address in the quad");
	RM(14, 16, 0, "P14 quad: u16 {0,1,2,3}");
	RM(16, 16, 0, "P16 quad: u16 {0,0,1,1}");
	RM(17, 16, 0, "P17 quad: u16 {0,1,1,2}");
	RM(18, 16, 0, "P18 quad: u16 {3,0,2,1}");
	RM(19, 16, 0, "P19 quad: u16 {0,2,0,2}");
	RM(20, 16, 0, "P20 quad: u16 {2,3,4,5}");
	RM(22, 16, 0, "P22 quad: u16 {0,0,0,1}");
	RM(21, 16, 0, "P21 quad: u16 {0,0,4,4}");
	RM(23, 16, 0, "P23 quad: u16 {0,0,0,8}");
	RM(8, 16, 0, "P8 quad inside 16 B (one u16 per dword)");
	RM(9, 16, 0, "P9 quad inside 32 B");
	RM(10, 16, 0, "P10 quad inside 64 B");
	RM(11, 16, 0, "P11 quad over 128 B");
	RM(12, 16, 0, "P12 quad: 2 addresses x 2 lanes, one line");
	RM(13, 16, 0, "P13 quad: 2 addresses on 2 lines");
	RM(3, 16, 0, "P3 a line per quad, 4 entries in the quad");
	RM(4, 16, 0, "P4 two lines per quad");
	RM(6, 16, 0, "P6 16 lines, each quad on 4 different lines");
	RM(5, 16, 0, "P5 64 lines");
	RM(5, 8, 0, "P5, 8 look-ups");
	RM(5, 16, 1, "P5, even lanes only (2 of each quad)");
	RM(5, 16, 2, "P5, one lane of each quad");
	RM(5, 16, 3, "P5, lanes 0-31");
	RM(5, 16, 4, "P5, even quads only");
	RM(5, 16, 5, "P5, a pseudo-random third of the lanes");
	RM(2, 16, 1, "P2, even lanes only");
	R(5, 16, 1, "P5 as dword loads");
	R(5, 16, 2, "P5 as dwordx2 loads");
	R(5, 16, 3, "P5 as dwordx4 loads");
	R(5, 4, 3, "P5 as dwordx4 loads, 4 per lane");
	R(5, 4, 0, "P5 u16, 4 per lane");
	R(2, 16, 3, "P2 as dwordx4 loads");
	return 0;
}
