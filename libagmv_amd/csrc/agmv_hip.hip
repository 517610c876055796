// libagmv_amd/csrc/agmv_hip.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels for the AGMV
// per-frame hot path, and the C-ABI of include/agmv_hip.h.
//
// What runs where (reference = /root/reference, cited as file:line):
//   k_lut_build     exact colour -> entry table; replaces the 256/512-way search of
//                   AGMV_FindNearestColor / AGMV_FindNearestEntry (src/agmv_utils.c:785-895)
//   k_mtx_build     512x512 bit matrix "palette colours within +-2 on every channel",
//                   the predicate of CompareI/PFrameBlock (src/agmv_encode.c:293,345)
//   k_encode        loops A+B of AGMV_EncodeFrame fused (src/agmv_encode.c:552-565, 240-527):
//                   one lane = one 4x4 block carried through the 4 frames of its GOP, one
//                   workgroup = 512 consecutive blocks; per-frame byte offsets by a decoupled
//                   look-back over tiles (single pass over the pixels)
//   k_parse_serial  block entry positions of a decompressed bitstream (src/agmv_decode.c:224-322)
//   k_decode        block -> RGB reconstruction (src/agmv_decode.c:249-319, 350-396, 401-405)
//   k_fixup         sequential repair of blocks whose value depends on an earlier GOP
//                   (stale tail after `escape`, src/agmv_decode.c:229-232; last-block FILL
//                   quirk :264-266)
// Integer/byte work only: no MFMA. The bound is HBM (4 B/px in, usize out).
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/agmv_hip.h"

#define FILL_FLAG   0x4Eu   /* include/agmv_defines.h:49 */
#define NORMAL_FLAG 0x2Fu   /* :50 */
#define COPY_FLAG   0x5Eu   /* :51 */
#define FILL_COUNT  14u     /* :52 */
#define COPY_COUNT  13u     /* :53 */

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(const char* what, hipError_t e, int line)
{
	snprintf(g_err, sizeof(g_err), "agmv_hip: %s failed: %s (agmv_hip.hip:%d)", what, hipGetErrorString(e), line);
	return -1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, e_, __LINE__); } while (0)
#define CKP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fail(#x, e_, __LINE__); return nullptr; } } while (0)

extern "C" const char* agmv_hip_last_error(void) { return g_err; }

// ----------------------------------------------------------------------------------------------
// geometry constants
// ----------------------------------------------------------------------------------------------
constexpr int ENC_T = 512;          // threads per encode workgroup = 4x4 blocks per tile
constexpr int ENC_WAVES = ENC_T / 64;
constexpr int MROW = 17;            // dwords per matrix row: 16 used + 1 pad (LDS bank spread)
constexpr int DEC_T = 256;          // threads per decode workgroup
constexpr uint32_t LUT_ENTRIES = 1u << 24;

constexpr unsigned long long ST_AGG = 1ull << 32;     // look-back status tags (high word)
constexpr unsigned long long ST_PREFIX = 2ull << 32;

struct agmv_hip_ctx {
	int device;
	int mode512;
	int have_palette;
	uint16_t* d_lut;                // 2^24 entries
	uint32_t* d_mtx;                // 512 * MROW dwords
	uint32_t* d_pal;                // 512 colours (p0 | p1)
	unsigned long long* d_status;   // look-back words
	size_t status_cap;              // in words
	uint32_t* d_ctrl;               // [0] ticket, [1] error, padded to 16 B
	uint32_t* d_dirty;              // decode: bitmap of block positions needing the fix-up
	size_t dirty_cap;               // in words
	int enc_grid;                   // resident workgroups for the persistent encode kernel
};

extern "C" size_t agmv_hip_max_usize(uint32_t w, uint32_t h, int mode512)
{
	size_t nblk = (size_t)(w / 4) * (h / 4);
	size_t n = nblk * (mode512 ? 33 : 17) + 64;
	return (n + 255) & ~(size_t)255;
}

// ----------------------------------------------------------------------------------------------
// K0: exact colour -> entry table.  One thread per colour; the palette index is wave-uniform so
// the palette is read through the scalar cache.  Same argmin + tie rules as the reference:
// strict '<' (lowest index wins, src/agmv_utils.c:810), palette0 on '<=' (src/agmv_utils.c:885).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void nearest_in(const uint32_t* __restrict__ pal, int r, int g, int b,
                                           uint32_t& best, uint32_t& idx)
{
	best = 3u * 255u * 255u + 1u;
	idx = 0;
#pragma unroll 8
	for (int i = 0; i < 256; i++) {
		uint32_t p = pal[i];
		int dr = r - (int)((p >> 16) & 0xff), dg = g - (int)((p >> 8) & 0xff), db = b - (int)(p & 0xff);
		uint32_t d = (uint32_t)(dr * dr + dg * dg + db * db);
		if (d < best) { best = d; idx = (uint32_t)i; }
	}
}

__global__ __launch_bounds__(256) void k_lut_build(const uint32_t* __restrict__ pal, int mode512,
                                                   uint16_t* __restrict__ lut)
{
	uint32_t c = blockIdx.x * 256u + threadIdx.x;
	int r = (int)(c >> 16), g = (int)((c >> 8) & 0xff), b = (int)(c & 0xff);
	uint32_t d0, i0;
	nearest_in(pal, r, g, b, d0, i0);
	uint32_t e = i0;
	if (mode512) {
		uint32_t d1, i1;
		nearest_in(pal + 256, r, g, b, d1, i1);
		if (!(d0 <= d1)) e = 0x100u | i1;
	}
	lut[c] = (uint16_t)e;
}

// K0b: bit (e2) of row (e1) = palette colours of entries e1 and e2 are within +-2 on R, G and B.
__device__ __forceinline__ bool within2(uint32_t a, uint32_t b)
{
	int dr = (int)((a >> 16) & 0xff) - (int)((b >> 16) & 0xff);
	int dg = (int)((a >> 8) & 0xff) - (int)((b >> 8) & 0xff);
	int db = (int)(a & 0xff) - (int)(b & 0xff);
	return (unsigned)(dr + 2) <= 4u && (unsigned)(dg + 2) <= 4u && (unsigned)(db + 2) <= 4u;
}

__global__ __launch_bounds__(256) void k_mtx_build(const uint32_t* __restrict__ pal, uint32_t* __restrict__ mtx)
{
	uint32_t t = blockIdx.x * 256u + threadIdx.x;     // 512 rows * MROW words
	if (t >= 512u * MROW) return;
	uint32_t row = t / MROW, word = t % MROW, bits = 0;
	if (word < 16) {
		uint32_t a = pal[row];
		for (uint32_t k = 0; k < 32; k++)
			bits |= (within2(a, pal[word * 32 + k]) ? 1u : 0u) << k;
	}
	mtx[t] = bits;
}

__global__ __launch_bounds__(256) void k_quantise(const uint32_t* __restrict__ pix, size_t n,
                                                  const uint16_t* __restrict__ lut, uint16_t* __restrict__ out)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	size_t stride = (size_t)gridDim.x * 256;
	for (; i < n; i += stride) out[i] = lut[pix[i] & 0xFFFFFFu];
}

// ----------------------------------------------------------------------------------------------
// K1: fused encode
// ----------------------------------------------------------------------------------------------
struct EncArgs {
	const uint32_t* pix;
	uint8_t* out;
	uint32_t* sizes;
	const uint16_t* lut;
	const uint32_t* mtx;
	unsigned long long* status;
	uint32_t* ctrl;
	uint16_t* ientries;
	unsigned long long out_stride;
	uint32_t n_frames, w, h, bw, nblk, tpf, first_fc, phase, n_groups, last_iframe, total_tiles;
};

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t y = __shfl_up(x, d, 64);
		if (lane >= d) x += y;
	}
	return x;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
	return x;
}

// decoupled look-back over the tiles of one frame (run by ONE wave; returns the exclusive
// prefix of `tile`).  Status words are single 8-byte {tag,value} granules read/written with
// relaxed agent-scope atomics (sc1), so no fence is needed (the data is the flag).
// Forward progress: tiles are handed out by a ticket counter, so every predecessor of a
// running tile is itself running or finished.  Spins are bounded; on timeout ctrl[1] is set.
__device__ __forceinline__ uint32_t lookback(unsigned long long* st, int tile, int lane, uint32_t* ctrl)
{
	uint32_t excl = 0;
	int j = tile - 1;
	for (;;) {
		int idx = j - lane;
		unsigned long long v = ST_PREFIX;          // tiles before the first: prefix 0
		unsigned spins = 0;
		for (;;) {
			if (idx >= 0) v = __hip_atomic_load(st + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (__all((v >> 32) != 0)) break;
			__builtin_amdgcn_s_sleep(1);
			if (++spins > (1u << 24)) {
				if (lane == 0) atomicExch(ctrl + 1, 1u);
				return excl;
			}
		}
		uint32_t val = (uint32_t)v;
		unsigned long long pm = __ballot((v >> 32) == 2);
		if (pm) {
			int first = __ffsll((long long)pm) - 1;
			excl += wave_sum(lane <= first ? val : 0u);
			return excl;
		}
		excl += wave_sum(val);
		j -= 64;
	}
}

template <bool M512>
__global__ __launch_bounds__(ENC_T, 4) void k_encode(EncArgs A)
{
	constexpr int NROWS = M512 ? 512 : 256;
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint32_t* s_mtx = (uint32_t*)smem;                         // NROWS * MROW dwords
	uint8_t* s_stage = smem + NROWS * MROW * 4;                // 4 B front pad + ENC_T*33 bytes
	uint32_t* s_misc = (uint32_t*)(s_stage + 4 + ENC_T * 33 + 12);   // [0..7] wave sums, [8] base, [9] ticket
	static_assert((4 + ENC_T * 33 + 12) % 16 == 0, "misc must stay aligned");

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const uint32_t npx = A.w * A.h;

	for (int i = tid; i < NROWS * MROW; i += ENC_T) s_mtx[i] = A.mtx[i];

	for (;;) {
		__syncthreads();                                       // matrix ready / previous tile fully drained
		if (tid == 0) s_misc[9] = atomicAdd(A.ctrl, 1u);
		__syncthreads();
		const uint32_t t = s_misc[9];
		if (t >= A.total_tiles) break;
		const uint32_t group = t / A.tpf, tile = t - group * A.tpf;
		// batch frames of this GOP: frame_count = first_fc + f, GOP boundary where it is 0 mod 4
		const int f_lo = group == 0 ? 0 : (int)(group * 4 - A.phase);
		int f_hi = (int)(group * 4 - A.phase) + 4;
		if (f_hi > (int)A.n_frames) f_hi = (int)A.n_frames;

		const uint32_t blk = tile * ENC_T + tid;
		const bool valid = blk < A.nblk;
		const uint32_t b = valid ? blk : A.nblk - 1;
		const uint32_t by = b / A.bw, bx = b - by * A.bw;
		const uint32_t poff = by * 4 * A.w + bx * 4;           // top-left pixel of the block

		uint32_t irow[16];                                     // matrix row (dword index) of the I-frame entries
		if (((A.first_fc + f_lo) & 3u) != 0) {                 // GOP started in an earlier batch
#pragma unroll
			for (int r = 0; r < 4; r++) {
				uint2 q = *(const uint2*)(A.ientries + poff + r * A.w);
				irow[r * 4 + 0] = (q.x & 0xffffu) * MROW; irow[r * 4 + 1] = (q.x >> 16) * MROW;
				irow[r * 4 + 2] = (q.y & 0xffffu) * MROW; irow[r * 4 + 3] = (q.y >> 16) * MROW;
			}
		} else {
#pragma unroll
			for (int k = 0; k < 16; k++) irow[k] = 0;
		}

		uint4 px[4];
		{
			const uint32_t* fp = A.pix + (size_t)f_lo * npx + poff;
#pragma unroll
			for (int r = 0; r < 4; r++) px[r] = *(const uint4*)(fp + r * A.w);
		}

		for (int f = f_lo; f < f_hi; f++) {
			const bool is_i = ((A.first_fc + f) & 3u) == 0;
			// ---- loop A: colour -> entry through the exact table (16 gathers in flight)
			uint32_t e[16];
#pragma unroll
			for (int r = 0; r < 4; r++) {
				e[r * 4 + 0] = A.lut[px[r].x & 0xFFFFFFu]; e[r * 4 + 1] = A.lut[px[r].y & 0xFFFFFFu];
				e[r * 4 + 2] = A.lut[px[r].z & 0xFFFFFFu]; e[r * 4 + 3] = A.lut[px[r].w & 0xFFFFFFu];
			}
			if (f + 1 < f_hi) {                                // prefetch the next frame of the GOP
				const uint32_t* fp = A.pix + (size_t)(f + 1) * npx + poff;
#pragma unroll
				for (int r = 0; r < 4; r++) px[r] = *(const uint4*)(fp + r * A.w);
			}
			// ---- loop B: block tests. count1 = CompareIFrameBlock vs the top-left entry colour
			// (src/agmv_encode.c:302-352), count2 = ComparePFrameBlock vs the I-frame entries
			// (src/agmv_encode.c:240-300); one matrix bit per pixel.
			const uint32_t row0 = e[0] * MROW;
			uint32_t acc1 = 0, acc2 = 0, nesc = 0;
#pragma unroll
			for (int k = 0; k < 16; k++) {
				uint32_t w1 = s_mtx[row0 + (e[k] >> 5)];
				acc1 = __builtin_amdgcn_alignbit(w1 >> (e[k] & 31u), acc1, 1);
				if (M512) nesc += ((e[k] & 0xffu) >= 127u) ? 1u : 0u;
			}
			if (!is_i) {
#pragma unroll
				for (int k = 0; k < 16; k++) {
					uint32_t w2 = s_mtx[irow[k] + (e[k] >> 5)];
					acc2 = __builtin_amdgcn_alignbit(w2 >> (e[k] & 31u), acc2, 1);
				}
			}
			const uint32_t count1 = __popc(acc1), count2 = __popc(acc2);
			const bool copy = !is_i && count2 >= COPY_COUNT;       // COPY has priority, :465
			const bool fill = !copy && count1 >= FILL_COUNT;
			uint32_t len;
			if (copy) len = 1;
			else if (fill) len = M512 ? (2u + ((e[0] & 0xffu) >= 127u ? 1u : 0u)) : 2u;
			else len = 17u + nesc;
			if (!valid) len = 0;

			if (is_i) {                                            // iframe_entries = img_entry, :626-630
#pragma unroll
				for (int k = 0; k < 16; k++) irow[k] = e[k] * MROW;
				if (A.ientries && (uint32_t)f == A.last_iframe && valid) {
#pragma unroll
					for (int r = 0; r < 4; r++) {
						uint2 q;
						q.x = e[r * 4 + 0] | (e[r * 4 + 1] << 16);
						q.y = e[r * 4 + 2] | (e[r * 4 + 3] << 16);
						*(uint2*)(A.ientries + poff + r * A.w) = q;
					}
				}
			}

			// ---- byte offsets: wave scan -> workgroup scan -> look-back across tiles
			const uint32_t incl = wave_incl_scan(len, lane);
			if (lane == 63) s_misc[wave] = incl;
			__syncthreads();                                       // (A)
			uint32_t woff = 0, total = 0;
#pragma unroll
			for (int i = 0; i < ENC_WAVES; i++) {
				uint32_t s = s_misc[i];
				if (i < wave) woff += s;
				total += s;
			}
			unsigned long long* st = A.status + (size_t)f * A.tpf;
			if (wave == 0 && lane == 0) {
				unsigned long long v = (tile == 0 ? ST_PREFIX : ST_AGG) | total;
				__hip_atomic_store(st + tile, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}

			// ---- emit this block's bytes into the LDS stage (independent of the global base)
			if (valid) {
				uint8_t* sp = s_stage + 4 + woff + incl - len;
				if (copy) {
					sp[0] = COPY_FLAG;
				} else if (fill) {
					sp[0] = FILL_FLAG;
					if (M512) {
						uint32_t idx = e[0] & 0xffu, p7 = (e[0] >> 1) & 0x80u;
						sp[1] = (uint8_t)(p7 | (idx < 127u ? idx : 127u));        // :382-388
						if (idx >= 127u) sp[2] = (uint8_t)idx;
					} else {
						sp[1] = (uint8_t)e[0];                                     // :421
					}
				} else {
					sp[0] = NORMAL_FLAG;
					uint32_t pos = 1;
#pragma unroll
					for (int k = 0; k < 16; k++) {
						if (M512) {
							uint32_t idx = e[k] & 0xffu, p7 = (e[k] >> 1) & 0x80u;
							sp[pos] = (uint8_t)(p7 | (idx < 127u ? idx : 127u));    // :395-401
							if (idx >= 127u) sp[pos + 1] = (uint8_t)idx;
							pos += 1u + (idx >= 127u ? 1u : 0u);
						} else {
							sp[pos++] = (uint8_t)e[k];                             // :428-429
						}
					}
				}
			}

			if (wave == 0) {
				uint32_t excl = 0;
				if (tile != 0) {
					excl = lookback(st, (int)tile, lane, A.ctrl);
					if (lane == 0)
						__hip_atomic_store(st + tile, ST_PREFIX | (unsigned long long)(excl + total),
						                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				if (lane == 0) {
					s_misc[8] = excl;
					if (tile == A.tpf - 1) A.sizes[f] = excl + total;   // usize of the frame
				}
			}
			__syncthreads();                                       // (B) stage + base ready

			// ---- cooperative copy stage -> frame bitstream, dword-wide on global-aligned dwords
			{
				const uint32_t base = s_misc[8];
				uint8_t* gdst = A.out + (size_t)f * A.out_stride + base;
				const uint32_t s = (uint32_t)((uintptr_t)gdst & 3u);
				uint8_t* g0 = gdst - s;
				const uint32_t ndw = (s + total + 3u) >> 2;
				const uint32_t* s32 = (const uint32_t*)s_stage;   // staged byte i lives at byte 4+i
				for (uint32_t j = tid; j < ndw; j += ENC_T) {
					const int lo_i = (int)(4u * j) - (int)s;          // staged index of this dword's byte 0
					if (lo_i >= 0 && (uint32_t)lo_i + 4u <= total) {
						uint32_t lo = s32[j], hi = s32[j + 1];
						uint32_t v = s ? __builtin_amdgcn_alignbyte(hi, lo, 4u - s) : hi;
						*(uint32_t*)(g0 + 4u * j) = v;
					} else {
#pragma unroll
						for (int q = 0; q < 4; q++) {
							int i = lo_i + q;
							if (i >= 0 && (uint32_t)i < total) g0[4u * j + q] = s_stage[4 + i];
						}
					}
				}
			}
			__syncthreads();                                       // (C) stage may be overwritten
		}
	}
}

// ----------------------------------------------------------------------------------------------
// decode side
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_flag(uint32_t b) { return b == FILL_FLAG || b == NORMAL_FLAG || b == COPY_FLAG; }

struct ByteSrc {
	const uint8_t* p;
	uint32_t cap;
	__device__ __forceinline__ uint32_t operator()(uint32_t pos) const { return pos < cap ? p[pos] : 0u; }
};

// K2 (serial form): one lane walks one frame's bitstream exactly like the reference's block loop
// (src/agmv_decode.c:226-320 / 327-397) but only records where each block is entered.
__global__ __launch_bounds__(64) void k_parse_serial(const uint8_t* __restrict__ bits, unsigned long long stride,
                                                     const uint32_t* __restrict__ bpos_a, uint32_t n_frames,
                                                     uint32_t nblk, int mode512, uint32_t* __restrict__ offsets,
                                                     uint32_t* __restrict__ nentered)
{
	uint32_t f = blockIdx.x * 64u + threadIdx.x;
	if (f >= n_frames) return;
	ByteSrc src{bits + (size_t)f * stride, (uint32_t)stride};
	const uint32_t bpos = bpos_a[f];
	uint32_t* off = offsets + (size_t)f * nblk;
	uint32_t bitpos = 0, k = 0;
	bool escape = false;
	while (k < nblk && !escape) {
		if (bitpos > bpos) break;
		off[k++] = bitpos;
		uint32_t byte = src(bitpos++);
		bool invalid = false;
		while (!is_flag(byte)) {
			byte = src(bitpos++);
			if (bitpos > bpos) { escape = true; break; }
		}
		if (!is_flag(byte)) invalid = true;
		if (byte == FILL_FLAG) {
			uint32_t idx = src(bitpos++);
			if (mode512 && (idx & 0x7fu) == 127u) bitpos++;
			if (bitpos > bpos) escape = true;
		} else if (byte == COPY_FLAG) {
		} else {
			for (int j = 0; j < 4; j++)
				for (int i = 0; i < 4; i++) {
					uint32_t idx = src(bitpos++);
					if (mode512 && (idx & 0x7fu) == 127u) bitpos++;
					if (bitpos > bpos || invalid) { escape = true; invalid = false; break; }
				}
		}
	}
	nentered[f] = k;
}

struct DecArgs {
	const uint8_t* bits;
	unsigned long long stride;
	const uint32_t* bpos;
	const uint32_t* offsets;
	const uint32_t* nentered;
	uint32_t* out;
	const uint32_t* pal;
	const uint32_t* prev;
	const uint32_t* prev_iframe;
	uint32_t* dirty;
	uint32_t n_frames, w, h, bw, nblk, tpf, first_fc, phase, n_groups;
};

// one 4x4 block of D2 (512 colours, src/agmv_decode.c:234-319) or D3 (256 colours, :335-396).
// `cur` is the block's img_data, `icol` the block's iframe->img_data.  fill_written reports a
// FILL that stored pixels (the caller applies the last-block quirk, :264-266).
template <bool M512, class Src>
__device__ __forceinline__ void decode_block(const Src& src, uint32_t bitpos, const uint32_t bpos,
                                             const uint32_t* pal, uint32_t (&cur)[16], const uint32_t (&icol)[16],
                                             bool istale, bool& stale, bool& fill_written)
{
	fill_written = false;
	uint32_t byte = src(bitpos++);
	bool invalid = false;
	while (!is_flag(byte)) {                                   // flag resync, :236-243
		byte = src(bitpos++);
		if (bitpos > bpos) break;
	}
	if (!is_flag(byte)) invalid = true;
	if (byte == FILL_FLAG) {
		uint32_t idx = src(bitpos++), color;
		if (M512) {
			const uint32_t base = (idx & 0x80u) ? 256u : 0u;
			if ((idx & 0x7fu) < 127u) color = pal[base + (idx & 0x7fu)];
			else color = pal[base + src(bitpos++)];
		} else {
			color = pal[idx];
		}
		if (!(bitpos > bpos)) {
#pragma unroll
			for (int k = 0; k < 16; k++) cur[k] = color;
			stale = false;
			fill_written = true;
		}
	} else if (byte == COPY_FLAG) {                            // no over-run check, :281-290
#pragma unroll
		for (int k = 0; k < 16; k++) cur[k] = icol[k];
		stale = istale;
	} else {
		bool dead = false;                                     // once a row broke, nothing more is stored
		uint32_t nwritten = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			bool rowbreak = false;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				if (!rowbreak) {
					uint32_t idx = src(bitpos++), color;
					if (M512) {
						const uint32_t base = (idx & 0x80u) ? 256u : 0u;
						if ((idx & 0x7fu) < 127u) color = pal[base + (idx & 0x7fu)];
						else color = pal[base + src(bitpos++)];
					} else {
						color = pal[idx];
					}
					if (bitpos > bpos || invalid) { invalid = false; rowbreak = true; dead = true; }
					else { cur[j * 4 + i] = color; nwritten++; }
				}
			}
		}
		(void)dead;
		if (nwritten == 16) stale = false;
	}
}

__device__ __forceinline__ void load_block(const uint32_t* frame, uint32_t poff, uint32_t w, uint32_t (&v)[16])
{
#pragma unroll
	for (int r = 0; r < 4; r++) {
		uint4 q = *(const uint4*)(frame + poff + r * w);
		v[r * 4 + 0] = q.x; v[r * 4 + 1] = q.y; v[r * 4 + 2] = q.z; v[r * 4 + 3] = q.w;
	}
}

__device__ __forceinline__ void store_block(uint32_t* frame, uint32_t poff, uint32_t w, const uint32_t (&v)[16])
{
#pragma unroll
	for (int r = 0; r < 4; r++) {
		uint4 q;
		q.x = v[r * 4 + 0]; q.y = v[r * 4 + 1]; q.z = v[r * 4 + 2]; q.w = v[r * 4 + 3];
		*(uint4*)(frame + poff + r * w) = q;
	}
}

// K3: one lane = one 4x4 block carried through the frames of its GOP (img_data and
// iframe->img_data of the block live in registers).  A block whose value depends on a frame
// outside the GOP (not rewritten since the GOP started) is flagged in `dirty` and repaired by
// k_fixup; everything else is final.
template <bool M512>
__global__ __launch_bounds__(DEC_T) void k_decode(DecArgs A)
{
	__shared__ uint32_t s_pal[512];
	__shared__ uint32_t s_nb[DEC_T];        // neighbour exchange for the last-block quirk
	__shared__ uint32_t s_nbstale[DEC_T];
	const int tid = threadIdx.x;
	const uint32_t npx = A.w * A.h;
	for (int i = tid; i < 512; i += DEC_T) s_pal[i] = A.pal[i];
	__syncthreads();

	const uint32_t group = blockIdx.x / A.tpf, tile = blockIdx.x - group * A.tpf;
	const int f_lo = group == 0 ? 0 : (int)(group * 4 - A.phase);
	int f_hi = (int)(group * 4 - A.phase) + 4;
	if (f_hi > (int)A.n_frames) f_hi = (int)A.n_frames;

	const uint32_t blk = tile * DEC_T + tid;
	const bool valid = blk < A.nblk;
	const uint32_t b = valid ? blk : A.nblk - 1;
	const uint32_t by = b / A.bw, bx = b - by * A.bw;
	const uint32_t poff = by * 4 * A.w + bx * 4;
	const bool has_last = (tile == A.tpf - 1);                 // this workgroup holds block nblk-1
	const bool is_last = valid && blk == A.nblk - 1;

	uint32_t cur[16], icol[16];
	bool stale, istale;
	if (group == 0) {                                          // state of the decoder before the batch
		if (A.prev) load_block(A.prev, poff, A.w, cur);
		else {
#pragma unroll
			for (int k = 0; k < 16; k++) cur[k] = 0;
		}
		if (A.prev_iframe) load_block(A.prev_iframe, poff, A.w, icol);
		else {
#pragma unroll
			for (int k = 0; k < 16; k++) icol[k] = 0;
		}
		stale = false; istale = false;
	} else {
#pragma unroll
		for (int k = 0; k < 16; k++) { cur[k] = 0; icol[k] = 0; }
		stale = true; istale = true;
	}

	for (int f = f_lo; f < f_hi; f++) {
		const uint32_t ne = A.nentered[f];
		bool fill_written = false;
		if (valid && blk < ne) {
			ByteSrc src{A.bits + (size_t)f * A.stride, (uint32_t)A.stride};
			decode_block<M512>(src, A.offsets[(size_t)f * A.nblk + blk], A.bpos[f], s_pal, cur, icol,
			                   istale, stale, fill_written);
		}
		if (has_last) {                                        // img_data[(x-1)+(y+1)*w] of the block to the left
			s_nb[tid] = cur[7];
			s_nbstale[tid] = stale ? 1u : 0u;
			__syncthreads();
			if (is_last && fill_written) {
				if (tid > 0) {
					uint32_t c = s_nb[tid - 1];
#pragma unroll
					for (int k = 0; k < 16; k++) cur[k] = c;
					stale = s_nbstale[tid - 1] != 0;
				} else {
					stale = true;                              // neighbour lives in another tile: fix-up
				}
			}
			__syncthreads();
		}
		if (((A.first_fc + f) & 3u) == 0) {                    // I-frame snapshot, :401-405
#pragma unroll
			for (int k = 0; k < 16; k++) icol[k] = cur[k];
			istale = stale;
		}
		if (valid) {
			store_block(A.out + (size_t)f * npx, poff, A.w, cur);
			if (stale) atomicOr(A.dirty + (blk >> 5), 1u << (blk & 31u));
		}
	}
}

// K4: sequential repair.  A single wave gathers up to 64 dirty block positions per pass and
// replays ALL frames for them in order, starting from the true pre-batch state, overwriting the
// output.  The left neighbour of the last block rides along so the FILL quirk sees its value.
template <bool M512>
__global__ __launch_bounds__(64) void k_fixup(DecArgs A)
{
	__shared__ uint32_t s_pal[512];
	__shared__ uint32_t s_list[64];
	__shared__ uint32_t s_cnt, s_next;
	const int lane = threadIdx.x;
	const uint32_t npx = A.w * A.h;
	for (int i = lane; i < 512; i += 64) s_pal[i] = A.pal[i];
	if (lane == 0) s_next = 0;
	__syncthreads();
	const uint32_t nwords = (A.nblk + 31) >> 5;
	for (;;) {
		// ---- collect the next (up to 64) dirty positions, in increasing order (lane 0 does it)
		if (lane == 0) {
			uint32_t cnt = 0, pos = s_next;
			while (pos < A.nblk && cnt < 64) {
				uint32_t wd = A.dirty[pos >> 5] >> (pos & 31u);
				if (wd == 0) { pos = (pos | 31u) + 1; continue; }
				pos += (uint32_t)__ffs((int)wd) - 1;
				if (pos >= A.nblk) break;
				if (pos == A.nblk - 1 && A.nblk >= 2 && (cnt == 0 || s_list[cnt - 1] != A.nblk - 2)) {
					if (cnt >= 63) break;                          // keep the pair in one pass
					s_list[cnt++] = A.nblk - 2;
				}
				s_list[cnt++] = pos;
				pos++;
			}
			s_cnt = cnt;
			s_next = pos;
			(void)nwords;
		}
		__syncthreads();
		const uint32_t cnt = s_cnt;
		if (cnt == 0) break;
		const bool active = (uint32_t)lane < cnt;
		const uint32_t blk = active ? s_list[lane] : 0;
		const uint32_t by = blk / A.bw, bx = blk - by * A.bw;
		const uint32_t poff = by * 4 * A.w + bx * 4;
		const bool is_last = active && blk == A.nblk - 1;
		uint32_t cur[16], icol[16];
		if (A.prev) load_block(A.prev, poff, A.w, cur);
		else {
#pragma unroll
			for (int k = 0; k < 16; k++) cur[k] = 0;
		}
		if (A.prev_iframe) load_block(A.prev_iframe, poff, A.w, icol);
		else {
#pragma unroll
			for (int k = 0; k < 16; k++) icol[k] = 0;
		}
		for (uint32_t f = 0; f < A.n_frames; f++) {
			bool stale = false, fill_written = false;
			if (active && blk < A.nentered[f]) {
				ByteSrc src{A.bits + (size_t)f * A.stride, (uint32_t)A.stride};
				decode_block<M512>(src, A.offsets[(size_t)f * A.nblk + blk], A.bpos[f], s_pal, cur, icol,
				                   false, stale, fill_written);
			}
			uint32_t left = __shfl_up(cur[7], 1, 64);          // neighbour sits in lane-1 by construction
			if (is_last && fill_written && lane > 0) {
#pragma unroll
				for (int k = 0; k < 16; k++) cur[k] = left;
			}
			if (((A.first_fc + f) & 3u) == 0) {
#pragma unroll
				for (int k = 0; k < 16; k++) icol[k] = cur[k];
			}
			if (active) store_block(A.out + (size_t)f * npx, poff, A.w, cur);
		}
		__syncthreads();
	}
}

// ----------------------------------------------------------------------------------------------
// caller-side helpers: synthetic clip, PDIFS midpoint, palette histogram
// ----------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// agmv_synth_v1 (integer-only, also stated in agmv_synth.c and tests/synth.py)
__host__ __device__ __forceinline__ uint32_t synth_pixel(uint32_t W, uint32_t H, uint32_t x, uint32_t y, uint32_t t, uint64_t seed)
{
	const uint64_t te = x < W / 4 ? 0 : t;                     // region A: static
	if (y >= 3 * H / 4) {                                      // region B: flat 32x32 tiles
		uint64_t tileid = ((uint64_t)(y / 32) << 40) | ((uint64_t)(x / 32) << 20) | (te / 8);
		return (uint32_t)(splitmix64(seed ^ tileid) & 0xFFFFFFu);
	}
	const uint64_t h = splitmix64(seed ^ (te * 0x9E3779B97F4A7C15ull) ^ (((uint64_t)y << 32) | x));
	uint32_t r = (uint32_t)(((uint64_t)x * 255 / (W - 1) + 2 * te) & 255);
	uint32_t g = (uint32_t)(((uint64_t)y * 255 / (H - 1) + te) & 255);
	uint32_t b = (uint32_t)((((uint64_t)x + y) / 2 + 3 * te) & 255);
	if ((h & 15) == 0) { r ^= (uint32_t)(h >> 8) & 7; g ^= (uint32_t)(h >> 16) & 7; b ^= (uint32_t)(h >> 24) & 7; }
	return r << 16 | g << 8 | b;
}

__global__ __launch_bounds__(256) void k_synth(uint32_t* __restrict__ pix, uint32_t W, uint32_t H, uint32_t t0,
                                               uint32_t n_frames, uint64_t seed)
{
	const size_t npx = (size_t)W * H, total = npx * n_frames;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	for (; i < total; i += stride) {
		uint32_t f = (uint32_t)(i / npx);
		uint32_t p = (uint32_t)(i - (size_t)f * npx);
		pix[i] = synth_pixel(W, H, p % W, p / W, t0 + f, seed);
	}
}

// AGMV_InterpFrame, src/agmv_utils.c:949-969: c1 + ((c2 - c1) >> 1) per channel, arithmetic shift
__global__ __launch_bounds__(256) void k_interp(uint32_t* __restrict__ out, const uint32_t* __restrict__ f1,
                                                const uint32_t* __restrict__ f2, size_t n)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	for (; i < n; i += stride) {
		uint32_t a = f1[i], b = f2[i];
		int r1 = (a >> 16) & 0xff, g1 = (a >> 8) & 0xff, b1 = a & 0xff;
		int r2 = (b >> 16) & 0xff, g2 = (b >> 8) & 0xff, b2 = b & 0xff;
		int r = r1 + ((r2 - r1) >> 1), g = g1 + ((g2 - g1) >> 1), bb = b1 + ((b2 - b1) >> 1);
		out[i] = (uint32_t)(r << 16 | g << 8 | bb);
	}
}

// AGMV_QuantizeColor, src/agmv_utils.c:695-742
__device__ __forceinline__ uint32_t quantize_color(uint32_t c, int quality)
{
	uint32_t r = (c >> 16) & 0xff, g = (c >> 8) & 0xff, b = c & 0xff;
	if (quality == 2) return (r >> 3) << 12 | (g >> 2) << 6 | (b >> 2);     // MID
	if (quality == 3) return (r >> 3) << 11 | (g >> 2) << 5 | (b >> 3);     // LOW
	return (r >> 2) << 13 | (g >> 2) << 7 | (b >> 1);                       // HIGH / default
}

__global__ __launch_bounds__(256) void k_histogram(const uint32_t* __restrict__ pix, size_t n, int quality,
                                                   uint32_t* __restrict__ hist)
{
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * 256;
	for (; i < n; i += stride) atomicAdd(hist + quantize_color(pix[i], quality), 1u);
}

// ----------------------------------------------------------------------------------------------
// C-ABI
// ----------------------------------------------------------------------------------------------
extern "C" int agmv_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" agmv_hip_ctx* agmv_hip_create(int device)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: no HIP device available (%s); the AGMV hot path has no CPU fallback",
		         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
		return nullptr;
	}
	if (device < 0 || device >= n) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: device %d out of range (0..%d)", device, n - 1);
		return nullptr;
	}
	CKP(hipSetDevice(device));
	agmv_hip_ctx* c = (agmv_hip_ctx*)calloc(1, sizeof(*c));
	c->device = device;
	CKP(hipMalloc(&c->d_lut, (size_t)LUT_ENTRIES * sizeof(uint16_t)));
	CKP(hipMalloc(&c->d_mtx, 512 * MROW * sizeof(uint32_t)));
	CKP(hipMalloc(&c->d_pal, 512 * sizeof(uint32_t)));
	CKP(hipMalloc(&c->d_ctrl, 16));
	hipDeviceProp_t prop;
	CKP(hipGetDeviceProperties(&prop, device));
	c->enc_grid = prop.multiProcessorCount * 2;               // 2 workgroups of 512 per CU (LDS/VGPR bound)
	return c;
}

extern "C" void agmv_hip_destroy(agmv_hip_ctx* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipFree(c->d_lut); (void)hipFree(c->d_mtx); (void)hipFree(c->d_pal); (void)hipFree(c->d_ctrl);
	(void)hipFree(c->d_status); (void)hipFree(c->d_dirty);
	free(c);
}

static int need_ctx(agmv_hip_ctx* c, bool palette)
{
	if (!c) { snprintf(g_err, sizeof(g_err), "agmv_hip: NULL context"); return -1; }
	if (palette && !c->have_palette) { snprintf(g_err, sizeof(g_err), "agmv_hip: agmv_hip_set_palette was not called"); return -1; }
	CK(hipSetDevice(c->device));
	return 0;
}

extern "C" int agmv_hip_set_palette(agmv_hip_ctx* c, const uint32_t p0[256], const uint32_t p1[256], int mode512, void* stream)
{
	if (need_ctx(c, false)) return -1;
	hipStream_t s = (hipStream_t)stream;
	uint32_t pal[512];
	memcpy(pal, p0, 1024);
	if (p1) memcpy(pal + 256, p1, 1024); else memset(pal + 256, 0, 1024);
	CK(hipMemcpyAsync(c->d_pal, pal, sizeof(pal), hipMemcpyHostToDevice, s));
	CK(hipStreamSynchronize(s));                              // pal[] is a stack buffer
	hipLaunchKernelGGL(k_lut_build, dim3(LUT_ENTRIES / 256), dim3(256), 0, s, c->d_pal, mode512 ? 1 : 0, c->d_lut);
	CK(hipGetLastError());
	hipLaunchKernelGGL(k_mtx_build, dim3((512 * MROW + 255) / 256), dim3(256), 0, s, c->d_pal, c->d_mtx);
	CK(hipGetLastError());
	c->mode512 = mode512 ? 1 : 0;
	c->have_palette = 1;
	return 0;
}

extern "C" int agmv_hip_quantise_dev(agmv_hip_ctx* c, const uint32_t* d_pix, size_t n, uint16_t* d_entries, void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (n == 0) return 0;
	size_t blocks = (n + 255) / 256;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(k_quantise, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_pix, n, c->d_lut, d_entries);
	CK(hipGetLastError());
	return 0;
}

static int check_geometry(uint32_t w, uint32_t h)
{
	if (w == 0 || h == 0 || (w & 3u) || (h & 3u)) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: width/height must be non-zero multiples of 4 (got %ux%u); "
		         "the reference's block loops have no edge handling (src/agmv_encode.c:365-366)", w, h);
		return -1;
	}
	if ((uint64_t)w * h >= (1ull << 31)) { snprintf(g_err, sizeof(g_err), "agmv_hip: frame too large"); return -1; }
	return 0;
}

extern "C" int agmv_hip_encode_frames_dev(agmv_hip_ctx* c, const uint32_t* d_pix, uint32_t n_frames, uint32_t w, uint32_t h,
                                          uint32_t first_fc, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                                          uint16_t* d_ientries, void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	if (out_stride < agmv_hip_max_usize(w, h, c->mode512)) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: out_stride %zu < agmv_hip_max_usize %zu", out_stride, agmv_hip_max_usize(w, h, c->mode512));
		return -1;
	}
	if (((uintptr_t)d_pix & 15u) || ((uintptr_t)d_out & 3u)) { snprintf(g_err, sizeof(g_err), "agmv_hip: d_pix must be 16-byte and d_out 4-byte aligned"); return -1; }
	hipStream_t s = (hipStream_t)stream;
	EncArgs A;
	memset(&A, 0, sizeof(A));
	A.pix = d_pix; A.out = d_out; A.sizes = d_sizes; A.lut = c->d_lut; A.mtx = c->d_mtx; A.ientries = d_ientries;
	A.out_stride = out_stride;
	A.n_frames = n_frames; A.w = w; A.h = h; A.bw = w / 4; A.nblk = (w / 4) * (h / 4);
	A.tpf = (A.nblk + ENC_T - 1) / ENC_T;
	A.first_fc = first_fc; A.phase = first_fc & 3u;
	A.n_groups = (n_frames + A.phase + 3) / 4;
	A.total_tiles = A.n_groups * A.tpf;
	if (A.phase != 0 && !d_ientries) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: batch starts inside a GOP (frame_count %u) but no I-frame entries were supplied", first_fc);
		return -1;
	}
	A.last_iframe = 0xffffffffu;
	for (int64_t f = (int64_t)n_frames - 1; f >= 0; f--)
		if (((first_fc + (uint32_t)f) & 3u) == 0) { A.last_iframe = (uint32_t)f; break; }
	size_t need = (size_t)n_frames * A.tpf;
	if (need > c->status_cap) {
		if (c->d_status) CK(hipFree(c->d_status));
		c->d_status = nullptr; c->status_cap = 0;
		CK(hipMalloc(&c->d_status, need * sizeof(unsigned long long)));
		c->status_cap = need;
	}
	A.status = c->d_status; A.ctrl = c->d_ctrl;
	CK(hipMemsetAsync(c->d_status, 0, need * sizeof(unsigned long long), s));
	CK(hipMemsetAsync(c->d_ctrl, 0, 16, s));
	uint32_t grid = (uint32_t)c->enc_grid;
	if (grid > A.total_tiles) grid = A.total_tiles;
	if (c->mode512) {
		size_t lds = 512 * MROW * 4 + 4 + ENC_T * 33 + 12 + 64;
		hipLaunchKernelGGL(k_encode<true>, dim3(grid), dim3(ENC_T), lds, s, A);
	} else {
		size_t lds = 256 * MROW * 4 + 4 + ENC_T * 33 + 12 + 64;
		hipLaunchKernelGGL(k_encode<false>, dim3(grid), dim3(ENC_T), lds, s, A);
	}
	CK(hipGetLastError());
	return 0;
}

extern "C" int agmv_hip_check(agmv_hip_ctx* c, void* stream)
{
	if (need_ctx(c, false)) return -1;
	uint32_t ctrl[4] = {0, 0, 0, 0};
	CK(hipStreamSynchronize((hipStream_t)stream));
	CK(hipMemcpy(ctrl, c->d_ctrl, 16, hipMemcpyDeviceToHost));
	if (ctrl[1]) { snprintf(g_err, sizeof(g_err), "agmv_hip: look-back timed out inside k_encode (device error word %u)", ctrl[1]); return -2; }
	return 0;
}

extern "C" int agmv_hip_encode_frames(agmv_hip_ctx* c, const uint32_t* h_pix, uint32_t n_frames, uint32_t w, uint32_t h,
                                      uint32_t first_fc, uint8_t* h_out, size_t out_stride, uint32_t* h_sizes, uint16_t* h_ient)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	const size_t npx = (size_t)w * h;
	uint32_t *d_pix = nullptr, *d_sizes = nullptr;
	uint8_t* d_out = nullptr;
	uint16_t* d_ient = nullptr;
	int rc = -1;
	do {
		if (hipMalloc(&d_pix, npx * 4 * n_frames) != hipSuccess || hipMalloc(&d_out, out_stride * n_frames) != hipSuccess ||
		    hipMalloc(&d_sizes, 4 * (size_t)n_frames) != hipSuccess || (h_ient && hipMalloc(&d_ient, npx * 2) != hipSuccess)) {
			snprintf(g_err, sizeof(g_err), "agmv_hip: device allocation failed"); break;
		}
		if (hipMemcpy(d_pix, h_pix, npx * 4 * n_frames, hipMemcpyHostToDevice) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: H2D failed"); break; }
		if (h_ient && hipMemcpy(d_ient, h_ient, npx * 2, hipMemcpyHostToDevice) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: H2D failed"); break; }
		if (agmv_hip_encode_frames_dev(c, d_pix, n_frames, w, h, first_fc, d_out, out_stride, d_sizes, d_ient, nullptr)) break;
		if (agmv_hip_check(c, nullptr)) break;
		if (hipMemcpy(h_sizes, d_sizes, 4 * (size_t)n_frames, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: D2H failed"); break; }
		bool ok = true;
		for (uint32_t f = 0; f < n_frames && ok; f++)
			ok = hipMemcpy(h_out + (size_t)f * out_stride, d_out + (size_t)f * out_stride, h_sizes[f], hipMemcpyDeviceToHost) == hipSuccess;
		if (!ok) { snprintf(g_err, sizeof(g_err), "agmv_hip: D2H failed"); break; }
		if (h_ient && hipMemcpy(h_ient, d_ient, npx * 2, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: D2H failed"); break; }
		rc = 0;
	} while (0);
	(void)hipFree(d_pix); (void)hipFree(d_out); (void)hipFree(d_sizes); (void)hipFree(d_ient);
	return rc;
}

extern "C" int agmv_hip_parse_frames_dev(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos,
                                         uint32_t n_frames, uint32_t w, uint32_t h, uint32_t* d_offsets, uint32_t* d_nentered,
                                         void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	uint32_t nblk = (w / 4) * (h / 4);
	hipLaunchKernelGGL(k_parse_serial, dim3((n_frames + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_bits,
	                   (unsigned long long)stride, d_bpos, n_frames, nblk, c->mode512, d_offsets, d_nentered);
	CK(hipGetLastError());
	return 0;
}

extern "C" int agmv_hip_decode_frames_dev(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos,
                                          const uint32_t* d_offsets, const uint32_t* d_nentered, uint32_t n_frames,
                                          uint32_t w, uint32_t h, uint32_t first_fc, uint32_t* d_out,
                                          const uint32_t* d_prev, const uint32_t* d_prev_iframe, void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	if (((uintptr_t)d_out & 15u) || ((uintptr_t)d_prev & 15u) || ((uintptr_t)d_prev_iframe & 15u)) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: pixel buffers must be 16-byte aligned"); return -1;
	}
	hipStream_t s = (hipStream_t)stream;
	DecArgs A;
	memset(&A, 0, sizeof(A));
	A.bits = d_bits; A.stride = stride; A.bpos = d_bpos; A.offsets = d_offsets; A.nentered = d_nentered;
	A.out = d_out; A.pal = c->d_pal; A.prev = d_prev; A.prev_iframe = d_prev_iframe;
	A.n_frames = n_frames; A.w = w; A.h = h; A.bw = w / 4; A.nblk = (w / 4) * (h / 4);
	A.tpf = (A.nblk + DEC_T - 1) / DEC_T;
	A.first_fc = first_fc; A.phase = first_fc & 3u;
	A.n_groups = (n_frames + A.phase + 3) / 4;
	size_t nwords = (A.nblk + 31) / 32;
	if (nwords > c->dirty_cap) {
		if (c->d_dirty) CK(hipFree(c->d_dirty));
		c->d_dirty = nullptr; c->dirty_cap = 0;
		CK(hipMalloc(&c->d_dirty, nwords * 4));
		c->dirty_cap = nwords;
	}
	A.dirty = c->d_dirty;
	CK(hipMemsetAsync(c->d_dirty, 0, nwords * 4, s));
	if (c->mode512) {
		hipLaunchKernelGGL(k_decode<true>, dim3(A.n_groups * A.tpf), dim3(DEC_T), 0, s, A);
		CK(hipGetLastError());
		hipLaunchKernelGGL(k_fixup<true>, dim3(1), dim3(64), 0, s, A);
	} else {
		hipLaunchKernelGGL(k_decode<false>, dim3(A.n_groups * A.tpf), dim3(DEC_T), 0, s, A);
		CK(hipGetLastError());
		hipLaunchKernelGGL(k_fixup<false>, dim3(1), dim3(64), 0, s, A);
	}
	CK(hipGetLastError());
	return 0;
}

extern "C" int agmv_hip_decode_frames(agmv_hip_ctx* c, const uint8_t* h_bits, size_t stride, const uint32_t* h_bpos,
                                      uint32_t n_frames, uint32_t w, uint32_t h, uint32_t first_fc, uint32_t* h_out,
                                      const uint32_t* h_prev, const uint32_t* h_prev_iframe)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	const size_t npx = (size_t)w * h, nblk = npx / 16;
	uint8_t* d_bits = nullptr;
	uint32_t *d_bpos = nullptr, *d_off = nullptr, *d_ne = nullptr, *d_out = nullptr, *d_prev = nullptr, *d_pi = nullptr;
	int rc = -1;
	do {
		if (hipMalloc(&d_bits, stride * n_frames) != hipSuccess || hipMalloc(&d_bpos, 4 * (size_t)n_frames) != hipSuccess ||
		    hipMalloc(&d_off, 4 * nblk * n_frames) != hipSuccess || hipMalloc(&d_ne, 4 * (size_t)n_frames) != hipSuccess ||
		    hipMalloc(&d_out, 4 * npx * n_frames) != hipSuccess || (h_prev && hipMalloc(&d_prev, 4 * npx) != hipSuccess) ||
		    (h_prev_iframe && hipMalloc(&d_pi, 4 * npx) != hipSuccess)) {
			snprintf(g_err, sizeof(g_err), "agmv_hip: device allocation failed"); break;
		}
		if (hipMemcpy(d_bits, h_bits, stride * n_frames, hipMemcpyHostToDevice) != hipSuccess ||
		    hipMemcpy(d_bpos, h_bpos, 4 * (size_t)n_frames, hipMemcpyHostToDevice) != hipSuccess ||
		    (h_prev && hipMemcpy(d_prev, h_prev, 4 * npx, hipMemcpyHostToDevice) != hipSuccess) ||
		    (h_prev_iframe && hipMemcpy(d_pi, h_prev_iframe, 4 * npx, hipMemcpyHostToDevice) != hipSuccess)) {
			snprintf(g_err, sizeof(g_err), "agmv_hip: H2D failed"); break;
		}
		if (agmv_hip_parse_frames_dev(c, d_bits, stride, d_bpos, n_frames, w, h, d_off, d_ne, nullptr)) break;
		if (agmv_hip_decode_frames_dev(c, d_bits, stride, d_bpos, d_off, d_ne, n_frames, w, h, first_fc, d_out, d_prev, d_pi, nullptr)) break;
		if (hipMemcpy(h_out, d_out, 4 * npx * n_frames, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: D2H failed"); break; }
		rc = 0;
	} while (0);
	(void)hipFree(d_bits); (void)hipFree(d_bpos); (void)hipFree(d_off); (void)hipFree(d_ne); (void)hipFree(d_out); (void)hipFree(d_prev); (void)hipFree(d_pi);
	return rc;
}

extern "C" int agmv_hip_synth_dev(agmv_hip_ctx* c, uint32_t* d_pix, uint32_t w, uint32_t h, uint32_t t0, uint32_t n_frames,
                                  uint64_t seed, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (w < 2 || h < 2 || n_frames == 0) { snprintf(g_err, sizeof(g_err), "agmv_hip: bad synth geometry"); return -1; }
	hipLaunchKernelGGL(k_synth, dim3(8192), dim3(256), 0, (hipStream_t)stream, d_pix, w, h, t0, n_frames, seed);
	CK(hipGetLastError());
	return 0;
}

extern "C" int agmv_hip_interp_dev(agmv_hip_ctx* c, uint32_t* d_out, const uint32_t* d_f1, const uint32_t* d_f2, size_t n, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (n == 0) return 0;
	size_t blocks = (n + 255) / 256;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(k_interp, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_out, d_f1, d_f2, n);
	CK(hipGetLastError());
	return 0;
}

extern "C" int agmv_hip_histogram_dev(agmv_hip_ctx* c, const uint32_t* d_pix, size_t n, int quality, uint32_t* d_hist, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (n == 0) return 0;
	size_t blocks = (n + 255) / 256;
	if (blocks > 8192) blocks = 8192;
	hipLaunchKernelGGL(k_histogram, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_pix, n, quality, d_hist);
	CK(hipGetLastError());
	return 0;
}

extern "C" void* agmv_hip_malloc(size_t bytes)
{
	void* p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: hipMalloc(%zu) failed", bytes); return nullptr; }
	return p;
}
extern "C" void agmv_hip_free(void* d) { if (d) (void)hipFree(d); }
extern "C" int agmv_hip_memcpy_h2d(void* d, const void* h, size_t n) { CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return 0; }
extern "C" int agmv_hip_memcpy_d2h(void* h, const void* d, size_t n) { CK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); return 0; }
extern "C" int agmv_hip_memset(void* d, int v, size_t n) { CK(hipMemset(d, v, n)); return 0; }
extern "C" int agmv_hip_sync(void) { CK(hipDeviceSynchronize()); return 0; }
