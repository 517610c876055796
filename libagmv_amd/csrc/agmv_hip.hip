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
//                   look-back over tiles (single pass over the pixels); no workgroup barrier in
//                   the steady state -- the waves exchange tagged LDS words
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
#include <pthread.h>

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
#ifndef ENC_T_OVERRIDE
#define ENC_T_OVERRIDE 512
#endif
#ifndef ENC_WPE
#define ENC_WPE 4
#endif
#ifndef ENC_PRIO
#define ENC_PRIO 2          /* s_setprio while a wave issues its look-ups + pixel loads (0 = off): the memory pipeline is the scarce unit */
#endif
#ifndef ENC_PIXAUX
#define ENC_PIXAUX 2          /* cache policy of the pixel loads: 2 = nt (streamed once; keeps L2 for the table), 0 = default */
#endif
#ifndef ENC_LUTAUX
#define ENC_LUTAUX 0          /* cache policy of the table look-ups */
#endif
constexpr int ENC_PFDEPTH = 1;     // items whose pixels are in flight per wave (the loop is unrolled by this many register sets); 2 measured the same as 1
constexpr int ENC_T = ENC_T_OVERRIDE;          // threads per encode workgroup = 4x4 blocks per tile
constexpr int ENC_WAVES = ENC_T / 64;
static_assert(ENC_WAVES >= 1 && ENC_WAVES <= 16, "the per-wave offsets are scanned inside one 16-lane row");
constexpr int MROW = 17;            // dwords per matrix row: 16 used + 1 pad (LDS bank spread)
#ifndef DEC_T_OVERRIDE
#define DEC_T_OVERRIDE 256
#endif
constexpr int DEC_T = DEC_T_OVERRIDE;          // threads per decode workgroup
#ifndef DEC_BPB
#define DEC_BPB 24
#endif
constexpr int DEC_STAGE = DEC_T * DEC_BPB;   // LDS window for a tile's bitstream bytes in ONE frame (24 B per block; beyond it bytes come from global memory); x4 frames = 24 KB, 5 workgroups per CU
constexpr int DEC_SR = DEC_STAGE / 4 / DEC_T;   // dwords of the window each lane carries from global memory to LDS
constexpr int DEC_MAX_SLICES = 32;           // agmv_hip_parse_decode_frames_dev: GOP ranges whose parse overlaps the reconstruction of the range before
constexpr uint32_t LUT_COLOURS = 1u << 24;
constexpr uint32_t LUT_ENTRIES = 1u << 28;   // index space of the table (see lut_index): 32 MiB populated in 512 MiB

constexpr unsigned long long ST_AGG = 1ull << 32;     // look-back status tags (high word)
constexpr unsigned long long ST_PREFIX = 2ull << 32;

struct agmv_hip_ctx {
	int device;
	int mode512;
	int have_palette;
	uint16_t* d_lut;                // 2^24 entries
	uint32_t* d_mtx;                // 512 * MROW dwords
	uint32_t* d_pal;                // 512 colours (p0 | p1)
	struct lut_share* share;        // owner of the three tables above (shared between the contexts of a device that hold the same palette)
	unsigned long long* d_status;   // look-back words
	size_t status_cap;              // in words
	uint32_t* d_ctrl;               // [0] ticket, [1] error, padded to 16 B
	uint16_t* d_ient_tmp;           // encode: I-frame entries written by a batch that also READS the caller's plane
	size_t ient_cap;                // in entries
	uint32_t* d_dirty;              // decode: bitmap of block positions needing the fix-up
	size_t dirty_cap;               // in words
	int enc_grid;                   // resident workgroups for the persistent encode kernel
	int n_cu;
	uint32_t* d_parse_ws;           // parser workspace: cum | centry | summ
	int timing;                     // record HIP events around the three hot kernels
	hipEvent_t ev[8];               // encode, parse, decode, parse||decode pipeline: start/stop
	size_t parse_ws_cap;            // in dwords
	uint32_t* d_fp_ws;              // fast parser workspace: rec | vm | kb | fstate
	size_t fp_ws_cap;               // in bytes
	uint32_t* d_fp_fstate;          // frame states of the last parse (inside d_fp_ws) and how many
	uint32_t fp_frames;
	unsigned long long* fp_vm;      // bitmap form of the last parse (inside d_fp_ws): entry bitmaps, first block per region, tile entries
	uint32_t* fp_kb;
	uint32_t* fp_tidx;
	uint32_t fp_maxR;
	uint32_t* d_nent_own;           // agmv_hip_decode_bitstreams_dev without a caller's nentered[]
	size_t nent_cap;
	hipEvent_t ev_enc;              // end of the last encode launch (encodes of one context share status / control words)
	hipStream_t enc_stream;         // ... and the stream it went to
	int have_enc;
	uint32_t* d_nn_pal;             // agmv_hip_nearest: palette, pixels, entries (grown on demand)
	uint32_t* d_nn_pix;
	uint16_t* d_nn_ent;
	size_t nn_cap;
	hipStream_t aux_stream;         // decode pipeline: the parser's stream (the reconstruction runs on the caller's)
	hipEvent_t ev_fork;             // ... caller's stream -> parser's stream
	hipEvent_t ev_slice[DEC_MAX_SLICES];   // ... slice parsed
};

extern "C" size_t agmv_hip_max_usize(uint32_t w, uint32_t h, int mode512)
{
	size_t nblk = (size_t)(w / 4) * (h / 4);
	size_t n = nblk * (mode512 ? 33 : 17) + 64;
	return (n + 255) & ~(size_t)255;
}

// LUT layout: one 128-byte line (64 u16 entries) holds a 4x4x4 cube of colour space, so the pixels
// of a neighbourhood (which differ mostly in the low bits of each channel) share lines.  The L1
// services one distinct line per cycle per gather instruction, which is what bounds the encoder
// (measured: gathers were 35 % of k_encode on the synthetic clip and 90 % on noise with a linear
// R,G,B table).  index = R[7:2] G[7:2] B[7:2] | R[1:0] G[1:0] B[1:0]
__host__ __device__ __forceinline__ uint32_t lut_index(uint32_t px)
{
	// same 4x4x4 cubes, but the cube number keeps the 2-bit holes of the masked pixel (R6 .. G6 .. B6): 6 VALU per
	// look-up instead of 12; the table spans 512 MiB of address space, 32 MiB of it populated (8 KiB runs every 32 KiB)
	return ((px & 0xFCFCFCu) << 4) | ((((px & 0x030303u) * 0x10410u) >> 16) & 0x3Fu);
}

// byte offset of a colour's entry, = 2 * lut_index(px), in 5 VALU for the sparse form (and, mul, bfe, and, lshl_or):
// bit 15 of the product is always 0, so the 7-bit field at bit 15 is the in-cube index already doubled
__device__ __forceinline__ uint32_t lut_offset(uint32_t px)
{
	const uint32_t m = __umul24(px & 0x030303u, 0x10410u);       // full-rate 24-bit multiply (a 32-bit v_mul_lo is quarter rate)
	const uint32_t hi = px & 0xFCFCFCu;
	uint32_t lo, off;                                          // spelled out: the compiler turns this into 4 instructions otherwise
	asm("v_bfe_u32 %0, %1, 15, 7" : "=v"(lo) : "v"(m));
	asm("v_lshl_or_b32 %0, %1, 5, %2" : "=v"(off) : "v"(hi), "v"(lo));
	return off;
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
	lut[lut_index(c)] = (uint16_t)e;
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
	for (; i < n; i += stride) out[i] = lut[lut_index(pix[i])];
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
	const uint16_t* ientries_in;    // entries of the GOP's I-frame when the batch starts inside a GOP
	uint16_t* ientries_out;         // receives the entries of the batch's last I-frame (never the buffer read above)
	unsigned long long out_stride;
	uint32_t n_frames, w, h, bw, nblk, tpf, first_fc, phase, n_groups, last_iframe, total_tiles;
};

// Workgroup barrier that orders LDS only.  __syncthreads() also carries a global-memory fence, i.e. an
// s_waitcnt vmcnt(0): every prefetch and every output store in flight would have to land before the barrier.
__device__ __forceinline__ void lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Wave-wide scans on the VALU's DPP lanes (row shifts inside the 16-lane rows, then the two row broadcasts gfx9 has
// for exactly this) -- six dependent adds, no LDS round trips (__shfl_up compiles to ds_bpermute + a wait per step).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or0(uint32_t x)       // lanes without a source (or masked off) read 0
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x, int)
{
	x += dpp_or0<0x111, 0xF>(x);                               // row_shr:1
	x += dpp_or0<0x112, 0xF>(x);                               // row_shr:2
	x += dpp_or0<0x114, 0xF>(x);                               // row_shr:4
	x += dpp_or0<0x118, 0xF>(x);                               // row_shr:8
	x += dpp_or0<0x142, 0xA>(x);                               // row_bcast:15 -> rows 1 and 3
	x += dpp_or0<0x143, 0xC>(x);                               // row_bcast:31 -> rows 2 and 3
	return x;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x)      // the same value in every lane
{
	return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(x, 0), 63);
}

// decoupled look-back over the tiles of one frame (run by ONE wave; returns the exclusive
// prefix of `tile`).  Status words are single 8-byte {tag,value} granules read/written with
// relaxed agent-scope atomics (sc1), so no fence is needed (the data is the flag).
// Forward progress: tiles are handed out by a ticket counter, so every predecessor of a
// running tile is itself running or finished.  Spins are bounded; on timeout ctrl[1] is set.
// `pre` is the first window's status word, loaded one frame earlier (latency hidden).
__device__ __forceinline__ unsigned long long st_load(unsigned long long* st, int idx)
{
	return idx >= 0 ? __hip_atomic_load(st + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ST_PREFIX;
}

__device__ __forceinline__ uint32_t lookback(unsigned long long* st, int tile, int lane, uint32_t* ctrl,
                                             unsigned long long pre)
{
	uint32_t excl = 0;
	int j = tile - 1;
	unsigned long long v = pre;
	for (;;) {
		const int idx = j - lane;
		unsigned spins = 0;
		while (!__all((v >> 32) != 0)) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > (1u << 22)) {
				if (lane == 0) atomicExch(ctrl + 1, 1u);
				return excl;
			}
			v = st_load(st, idx);
		}
		const uint32_t val = (uint32_t)v;
		const unsigned long long pm = __ballot((v >> 32) == 2);
		if (pm) {
			const int first = __ffsll((long long)pm) - 1;
			excl += wave_sum(lane <= first ? val : 0u);
			return excl;
		}
		excl += wave_sum(val);
		j -= 64;
		v = st_load(st, j - lane);
	}
}

constexpr int WBLK = 64;                                       // blocks per wave = slice of the workgroup tile
constexpr size_t CTRL_BYTES = 1024;                            // dwords: [0] ticket, [1] error, [32..42] phase stamps (ENC_PROF builds), the rest: lab builds

// K1.  Barrier-free dataflow form.  One workgroup = one tile of ENC_T consecutive 4x4 blocks (ENC_WAVES waves x 64
// blocks), one lane = one block for classification/emission, carried through the <=4 frames of its GOP.  The stream of
// (tile, frame) items a workgroup processes is ONE software pipeline that runs across tile switches; the waves of a
// workgroup never meet at a barrier inside it -- they exchange single tagged LDS words:
//   item `it`, every wave:  (Q) quantise with lane = (block, row): wide row loads, each LUT gather instruction covers a
//                               64x4-pixel patch; entries transposed to lane = block through the wave's own stage slot
//                           prefetch of the next item's pixels (next frame, or the first frame of the NEXT tile, whose
//                               ticket was drawn one tile earlier)
//                           (C) classify FILL / COPY / NORMAL, wave scan of the byte lengths -> wsum[it][wave];
//                               the LAST wave to arrive (LDS counter) adds the tile up and publishes the aggregate
//                           (E) emit bytes into the wave's OWN stage slot (offsets inside the wave only)
//                           copy-out of item it-1 from its own slot, at gbase[it-1][wave]
//   item `it`, duty wave (rotating): decoupled look-back of item it-1 across tiles (status window prefetched ahead
//                               of the look-ups), then gbase[it-1][w] = tile offset + bytes of the lower waves.
// Waves therefore drift apart by up to ~1.5 items, and the phases (look-up issue, look-up wait, LDS, VALU) of the
// waves sharing a SIMD interleave instead of lining up behind a barrier.  Control words live in 8 slots (it & 7): a
// wave can finish item `it` only after gbase[it-1] exists (it waits for that word even when it has no bytes to copy),
// i.e. after EVERY wave has published wsum[it-1]; so when a slot is rewritten for item it+1 all waves have finished
// item it-2 and with it every read of item it-3's words (4 slots would do).
constexpr int DF_SLOTS = 8;
constexpr int WSLOT = 16 + WBLK * 33 + 16;                     // a wave's stage slot: front pad + worst case + tail pad
static_assert(WSLOT % 16 == 0 && WSLOT >= 16 + WBLK * 16 * 2, "slot alignment / transpose scratch");
constexpr int C_WSUM = 0;                                      // [slot][wave]      tag<<16 | bytes of the wave
constexpr int C_ARRIVE = C_WSUM + DF_SLOTS * ENC_WAVES;        // [slot]            waves that have published wsum
constexpr int C_TTOTAL = C_ARRIVE + DF_SLOTS;                  // [slot]            tag<<16 | bytes of the tile
constexpr int C_GBASE = C_TTOTAL + DF_SLOTS;                   // [slot][wave][2]   frame byte offset of the wave, tag
constexpr int C_TICKET = C_GBASE + DF_SLOTS * ENC_WAVES * 2;   // [slot][2]         ticket of tile sequence number s, s
constexpr int C_END = C_TICKET + DF_SLOTS * 2;
constexpr size_t ENC_LDS_EXTRA = 2 * ENC_WAVES * WSLOT + C_END * 4;

typedef uint16_t __attribute__((aligned(1))) u16u;          // byte-aligned 16/32-bit LDS stores (DS unaligned mode)
typedef uint32_t __attribute__((aligned(1))) u32u;
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // explicit LDS pointers: ds_read / ds_write, never flat
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return *(const volatile lds_u32*)p; }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { *(volatile lds_u32*)p = v; }

// Stores the compiler does not track: hipcc guards the data registers of a store it knows about with s_waitcnt vmcnt(0) before
// they are written again, i.e. it waits for the write to be ACKNOWLEDGED (1-2 us for the status words, which go to the fabric)
// -- three such waits per item in the look-back / copy-out sequence.  The hardware reads the data of a 4- / 8-byte store when
// it issues it, and nothing in the wave waits for these stores, so they go out as asm (the compiler's waits for LOADS can only
// become longer by it, never shorter: the memory counter retires in order).
__device__ __forceinline__ void st_store_untracked(unsigned long long* p, unsigned long long v)
{
	asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");      // = a relaxed agent-scope atomic store
}
__device__ __forceinline__ void store32_untracked(uint32_t* p, uint32_t v)
{
	asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store8_untracked(uint8_t* p, uint32_t v)
{
	asm volatile("global_store_byte %0, %1, off" :: "v"(p), "v"(v) : "memory");
}
// wave_copy_own with untracked stores
__device__ __forceinline__ void wave_copy_own_u(const uint8_t* slot, uint8_t* gdst, uint32_t total, int lane)
{
	const uint32_t s = (uint32_t)((uintptr_t)gdst & 3u);
	uint32_t head = s ? 4u - s : 0u;
	if (head > total) head = total;
	const uint32_t nbody = (total - head) >> 2, tail = (total - head) & 3u;
	const uint8_t* sb = slot + 16 + head;
	uint8_t* gb = gdst + head;
	for (uint32_t d = lane; d < nbody; d += 64) store32_untracked((uint32_t*)(gb + 4u * d), *(const u32u*)(sb + 4u * d));
	if ((uint32_t)lane < head) store8_untracked(gdst + lane, slot[16 + lane]);
	const uint32_t tl = (uint32_t)lane - 8u;
	if (tl < tail) store8_untracked(gb + 4u * nbody + tl, sb[4u * nbody + tl]);
}
// look-back of k_encode's duty wave.  The common case -- every status word of the first window is published and one of them is a
// prefix -- is straight-line code on the window the caller loaded long ago (`pre`); anything else goes through the general
// loop above, out of line: inlined, its conditional reloads leave the compiler unsure whether a load is still pending at
// every later write of those registers, and it answers each with s_waitcnt vmcnt(0) -- which at that point also waits for
// the status / copy-out STORES in flight to be acknowledged (1-2 us each, three times per item).
__device__ __noinline__ uint32_t lookback_slow(unsigned long long* st, int tile, int lane, uint32_t* ctrl, unsigned long long pre)
{
	return lookback(st, tile, lane, ctrl, pre);
}
__device__ __forceinline__ uint32_t lookback_w(unsigned long long* st, int tile, int lane, uint32_t* ctrl, unsigned long long pre)
{
	const uint32_t tag = (uint32_t)(pre >> 32), val = (uint32_t)pre;
	const unsigned long long pm = __ballot(tag == 2u);
	if (__all(tag != 0u) && pm != 0) {
		const int first = __ffsll((long long)pm) - 1;
		return wave_sum(lane <= first ? val : 0u);
	}
	return lookback_slow(st, tile, lane, ctrl, pre);
}
// status word of segment idx without a branch around the load (idx < 0, before the frame, reads word 0)
__device__ __forceinline__ unsigned long long st_load_raw(unsigned long long* st, int idx)     // the caller substitutes ST_PREFIX for idx < 0
{
	return __hip_atomic_load(st + (idx >= 0 ? idx : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-uniform bounded spin until the word at p satisfies (v >> shift) == tag; returns the word
__device__ __forceinline__ uint32_t lds_wait(const uint32_t* p, uint32_t tag, int shift, uint32_t* ctrl, int lane)
{
	uint32_t v = __builtin_amdgcn_readfirstlane(lds_ld(p));
	unsigned spins = 0;
	while ((v >> shift) != tag) {
		__builtin_amdgcn_s_sleep(1);
		if (++spins > (1u << 24)) {                          // a legitimate wait is microseconds; the bound is above the look-back's (1 << 22 polls of global memory), which this wait can transitively wait for
			if (lane == 0) atomicExch(ctrl + 1, 2u);
			break;
		}
		v = __builtin_amdgcn_readfirstlane(lds_ld(p));
	}
	return v;
}

// copy the `total` bytes staged at slot+16 to gdst (one wave): dword stores on GLOBAL-aligned dwords, read from the
// stage at byte-granular LDS addresses (DS unaligned mode); the <= 3 bytes before the first and after the last aligned
// dword (which share a dword with the neighbouring waves / tiles) go out as byte stores from eight lanes in one step
__device__ __forceinline__ void wave_copy_own(const uint8_t* slot, uint8_t* gdst, uint32_t total, int lane)
{
	const uint32_t s = (uint32_t)((uintptr_t)gdst & 3u);
	uint32_t head = s ? 4u - s : 0u;
	if (head > total) head = total;
	const uint32_t nbody = (total - head) >> 2, tail = (total - head) & 3u;
	const uint8_t* sb = slot + 16 + head;
	uint8_t* gb = gdst + head;
	for (uint32_t d = lane; d < nbody; d += 64) *(uint32_t*)(gb + 4u * d) = *(const u32u*)(sb + 4u * d);
	if ((uint32_t)lane < head) gdst[lane] = slot[16 + lane];
	const uint32_t tl = (uint32_t)lane - 8u;
	if (tl < tail) gb[4u * nbody + tl] = sb[4u * nbody + tl];
}

// The two block tests of one 4x4 block (entries packed two per register): acc1 collects one matrix bit per pixel
// against the block's top-left entry (row0), acc2 -- P-frames only -- against the I-frame's entry of the same pixel.
template <bool M512, bool PFRAME>
__device__ __forceinline__ void block_tests(const uint32_t (&ep)[8], const uint32_t (&ip)[8], const uint32_t* s_mtx,
                                            uint32_t row0, uint32_t& acc1, uint32_t& acc2, uint32_t& nesc)
{
	// (requesting all 16 / 32 matrix words before the first use -- 110 VGPRs instead of 96 -- measured the same: 0.723 vs
	//  0.728 ms per 256 frames; the LDS round trips of one wave are covered by the other three of its SIMD)
#pragma unroll
	for (int m = 0; m < 8; m++) {
		const uint32_t p = ep[m], a5 = (p >> 5) & 0x7ffu, b5 = p >> 21, bh = p >> 16;
		const uint32_t wa = s_mtx[row0 + a5], wb = s_mtx[row0 + b5];
		acc1 = __builtin_amdgcn_alignbit(wa >> (p & 31u), acc1, 1);
		acc1 = __builtin_amdgcn_alignbit(wb >> (bh & 31u), acc1, 1);
		if (M512) nesc += ((p & 0xffu) >= 127u ? 1u : 0u) + ((bh & 0xffu) >= 127u ? 1u : 0u);
		if (PFRAME) {
			const uint32_t q = ip[m];
			const uint32_t va = s_mtx[(q & 0xffffu) * MROW + a5], vb = s_mtx[(q >> 16) * MROW + b5];
			acc2 = __builtin_amdgcn_alignbit(va >> (p & 31u), acc2, 1);
			acc2 = __builtin_amdgcn_alignbit(vb >> (bh & 31u), acc2, 1);
		}
	}
}

struct EncGeo {
	uint32_t tile, wbase, wb_c, wbx, wby;                      // wave-uniform
	int f_lo, f_hi, path;
	uint32_t poff, p0b, qx0;                                   // per lane
	bool valid;
};

// ENTRIES: the input planes hold ENTRIES (one per 32-bit word, pal_num << 8 | index) instead of pixels -- the table look-ups
// are skipped and classification + emission run on the caller's entries (AGMV_AssembleIFrameBitstream /
// AGMV_AssemblePFrameBitstream on a given AGMV_ENTRY plane, src/agmv_encode.c:354-527).
typedef uint32_t px4 __attribute__((ext_vector_type(4)));   // a native vector (HIP's uint4 is a struct: asm cannot tie it to a register tuple)
template <bool M512, bool ENTRIES>
__global__ __launch_bounds__(ENC_T, ENC_WPE) void k_encode(EncArgs A)
{
	constexpr int NROWS = M512 ? 512 : 256;
	extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
	uint32_t* s_mtx = (uint32_t*)smem;                         // NROWS * MROW dwords
	uint8_t* s_stage0 = smem + NROWS * MROW * 4;               // [2][ENC_WAVES] stage slots
	uint32_t* s_ctl = (uint32_t*)(s_stage0 + 2 * ENC_WAVES * WSLOT);

	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t npx = A.w * A.h;
	const uint32_t jb = lane >> 2, prow = lane & 3;            // quantise phase: lane = (block jb of 16, row prow)
	const __amdgpu_buffer_rsrc_t lut_rs = __builtin_amdgcn_make_buffer_rsrc((void*)A.lut, 0, (int)(LUT_ENTRIES * 2u), 0x00020000);

	for (int i = tid; i < NROWS * MROW; i += ENC_T) s_mtx[i] = A.mtx[i];
	for (int i = tid; i < C_END; i += ENC_T) s_ctl[i] = 0;
	__syncthreads();
	if (tid == 0) s_ctl[C_TICKET] = atomicAdd(A.ctrl, 1u);
	__syncthreads();                                           // the only workgroup barriers of the kernel

	auto locate = [&](const EncGeo& g, uint32_t B, uint32_t& qx, uint32_t& qy) {   // block index (>= wb_c) -> block column / row
		if (B >= A.nblk) B = A.nblk - 1;                       // blocks past the frame re-use the last valid one
		qx = g.wbx + (B - g.wb_c); qy = g.wby;
		while (qx >= A.bw) { qx -= A.bw; qy++; }
	};
	// tile-major ticket order: consecutive tickets are the SAME tile of different GOPs, so a tile's predecessors (same
	// GOP, lower tile) are n_groups tickets older -> mostly finished when the look-back reads them.
	// geometry: ONE integer division per wave (of its first block, wave-uniform); lane positions follow by adding and
	// wrapping at the end of a block row (no per-lane divisions)
	auto setup = [&](uint32_t t, EncGeo& g) {
		g.tile = t / A.n_groups;
		const uint32_t group = t - g.tile * A.n_groups;
		g.f_lo = group == 0 ? 0 : (int)(group * 4 - A.phase);
		g.f_hi = (int)(group * 4 - A.phase) + 4;
		if (g.f_hi > (int)A.n_frames) g.f_hi = (int)A.n_frames;
		g.wbase = g.tile * ENC_T + wave * WBLK;                // first block of this wave
		g.wb_c = g.wbase < A.nblk ? g.wbase : A.nblk - 1;
		g.wby = __builtin_amdgcn_readfirstlane(g.wb_c / A.bw); g.wbx = g.wb_c - g.wby * A.bw;
		// lane-as-block view (classification, emission, I-frame entry plane)
		const uint32_t blk = g.wbase + lane;
		g.valid = blk < A.nblk;
		uint32_t bx, by;
		locate(g, blk, bx, by);
		g.poff = by * 4 * A.w + bx * 4;                        // top-left pixel of the block
		// quantise view, lane = (block, row): load i (0..3) fetches row `prow` (16 bytes) of block wbase + 16i + jb, so
		// one instruction reads four 256-byte row segments of 16 adjacent blocks and each of its four pixel columns is
		// a 64x4-pixel patch for the LUT gather.  A wave inside one block row uses immediate offsets, one crossing a
		// single row boundary adds 3*w past it, anything else locates each of its four blocks.
		uint32_t qy0;
		locate(g, g.wbase + jb, g.qx0, qy0);
		g.p0b = ((qy0 * 4 + prow) * A.w + g.qx0 * 4) * 4u;
		g.path = (g.wbase + WBLK > A.nblk || g.wbx + WBLK > 2 * A.bw) ? 2 : (g.wbx + WBLK > A.bw ? 1 : 0);
	};
	auto load_frame = [&](const EncGeo& g, const uint32_t* fp, px4 (&dst)[4]) {
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)fp, 0, (int)(npx * 4u), 0x00020000);
		const uint32_t w3b = 12u * A.w;
		if (g.path == 0) {
#pragma unroll
			for (int i = 0; i < 4; i++) dst[i] = __builtin_bit_cast(px4, __builtin_amdgcn_raw_buffer_load_b128(rs, g.p0b + 256 * i, 0, ENC_PIXAUX));
		} else if (g.path == 1) {
#pragma unroll
			for (int i = 0; i < 4; i++)
				dst[i] = __builtin_bit_cast(px4, __builtin_amdgcn_raw_buffer_load_b128(rs, g.p0b + 256 * i + (g.qx0 + 16 * i >= A.bw ? w3b : 0u), 0, ENC_PIXAUX));
		} else {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				uint32_t qx, qy;
				locate(g, g.wbase + jb + 16 * i, qx, qy);
				dst[i] = __builtin_bit_cast(px4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((qy * 4 + prow) * A.w + qx * 4) * 4u, 0, ENC_PIXAUX));
			}
		}
	};

	// Two cursors walk the workgroup's sequence of (tile, frame) items: `cur` is the item being encoded, `pf` the item whose
	// pixels are requested next, ENC_PFDEPTH items ahead (every item's pixels are in flight for that many item times).  A
	// cursor that leaves its tile takes the ticket of the next one from the LDS slot of that tile's sequence number; wave 0
	// draws a tile's ticket when its OWN prefetch cursor enters the tile before it and publishes it behind its next look-ups.
	struct Cursor { EncGeo g; int f; uint32_t seq; bool have; };
	Cursor cur, pf;
	uint32_t tk = 0, tk_seq = 0;
	bool tk_pending = false;
	auto draw_ticket = [&](uint32_t for_seq) {
		if (wave == 0) {
			if (lane == 0) tk = atomicAdd(A.ctrl, 1u);
			tk_seq = for_seq; tk_pending = true;
		}
	};
	auto publish_ticket = [&]() {
		if (wave == 0 && tk_pending) {
			const uint32_t tkv = __builtin_amdgcn_readfirstlane(tk);
			if (lane == 0) {
				uint32_t* tw = &s_ctl[C_TICKET + (tk_seq & (DF_SLOTS - 1)) * 2];
				lds_st(tw, tkv);
				asm volatile("" ::: "memory");
				lds_st(tw + 1, tk_seq);
			}
			tk_pending = false;
		}
	};
	auto advance = [&](Cursor& c, bool lead) -> bool {         // to the next item; true when that is the first item of a tile
		if (c.f + 1 < c.g.f_hi) { c.f++; return false; }
		if (lead) publish_ticket();                            // wave 0 is about to wait for the ticket it drew itself
		const uint32_t* tw = &s_ctl[C_TICKET + ((c.seq + 1) & (DF_SLOTS - 1)) * 2];
		lds_wait(tw + 1, c.seq + 1, 0, A.ctrl, lane);
		asm volatile("" ::: "memory");
		const uint32_t nt = __builtin_amdgcn_readfirstlane(lds_ld(tw));
		c.seq++;
		c.have = nt < A.total_tiles;
		if (c.have) {
			setup(nt, c.g);
			c.f = c.g.f_lo;
			if (lead) draw_ticket(c.seq + 1);
		}
		return true;
	};
	{
		const uint32_t ticket = __builtin_amdgcn_readfirstlane(lds_ld(&s_ctl[C_TICKET]));
		cur.seq = 0; cur.f = 0;
		cur.have = ticket < A.total_tiles;
		if (cur.have) { setup(ticket, cur.g); cur.f = cur.g.f_lo; }
	}
	pf = cur;
	if (pf.have) draw_ticket(1);
	uint32_t it = 0;                                           // item number (tags / slots)
	bool new_tile = true;
	uint32_t ip[8];                                            // the GOP's I-frame entries of this block, two u16 per register
	px4 pxs[ENC_PFDEPTH][4];
#pragma unroll
	for (int d = 0; d < ENC_PFDEPTH; d++)
		if (pf.have) { load_frame(pf.g, A.pix + (size_t)pf.f * npx, pxs[d]); advance(pf, true); }
	bool have_prev = false;                                    // item it-1: tile, frame, bytes of this wave
	uint32_t p_tile = 0, p_len = 0;
	int p_f = 0;
	unsigned long long pre_next = ST_PREFIX;                   // the status window the NEXT item's duty wave will look back through

#ifdef ENC_PROF
	uint32_t prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long pt = __builtin_amdgcn_s_memtime();
#define PSTAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof[k] += (uint32_t)(n_ - pt); pt = n_; } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
	uint32_t prof_sink = 0; (void)prof_sink;
	auto body = [&](px4 (&px)[4]) -> bool {                 // one item: consumes px and refills it with the item ENC_PFDEPTH ahead
		EncGeo& g = cur.g;
		const int f = cur.f;
		const uint32_t slot = it & (DF_SLOTS - 1), tag = (it + 1) & 0xffffu;
		const uint32_t pslot = (it - 1) & (DF_SLOTS - 1), ptag = it & 0xffffu;
		// (L) the look-back of item it-1 is the duty of ONE wave (rotating).  Its status window was requested at the END of
		// item it-1, behind that item's pixel prefetch (pre_next): the status words bypass the caches -- a fabric round trip,
		// several times the latency of a table look-up -- and the memory counter retires in order, so requested here, ahead
		// of the look-ups, the window would hold back every entry of the duty wave, and with it gbase[] for all eight waves
		const bool duty = have_prev && wave == (int)(it % ENC_WAVES);
		const unsigned long long pre = pre_next;
		auto resolve_prev = [&]() {                            // tile offset of item it-1, usize of the frame, gbase[]
			unsigned long long* st = A.status + (size_t)p_f * A.tpf;
			uint32_t excl = 0;
			// (straight-line look-back on the window loaded earlier; the general loop is out of line -- inlined, its conditional
			//  reloads make the compiler drain the memory counter before gbase[] is published, i.e. the seven other waves of the
			//  workgroup would wait for the acknowledgement of this wave's status stores: a fabric round trip per item)
			if (p_tile != 0) excl = lookback_w(st, (int)p_tile, lane, A.ctrl, pre);
			const uint32_t tot = lds_wait(&s_ctl[C_TTOTAL + pslot], ptag, 16, A.ctrl, lane) & 0xffffu;
			// every wave of the tile has published its byte count (the tile total exists): exclusive scan over the waves.
			// gbase[] goes out FIRST: it is what the other waves wait for
			const uint32_t ws = lane < ENC_WAVES ? (lds_ld(&s_ctl[C_WSUM + pslot * ENC_WAVES + lane]) & 0xffffu) : 0u;
			uint32_t inc = ws;
			inc += dpp_or0<0x111, 0xF>(inc);                    // the waves sit in the first lanes of row 0
			if (ENC_WAVES > 2) inc += dpp_or0<0x112, 0xF>(inc);
			if (ENC_WAVES > 4) inc += dpp_or0<0x114, 0xF>(inc);
			if (ENC_WAVES > 8) inc += dpp_or0<0x118, 0xF>(inc);
			if (lane < ENC_WAVES) {
				uint32_t* gb = &s_ctl[C_GBASE + (pslot * ENC_WAVES + lane) * 2];
				lds_st(gb, excl + inc - ws);
				asm volatile("" ::: "memory");
				lds_st(gb + 1, it);
			}
			if (lane == 0) {
				if (p_tile != 0) st_store_untracked(st + p_tile, ST_PREFIX | (unsigned long long)(excl + tot));
				if (p_tile == A.tpf - 1) store32_untracked(A.sizes + p_f, excl + tot);   // usize of the frame
			}
		};
		auto copy_out_prev = [&]() {
			// EVERY wave waits here, also one with nothing to copy: this wait is what bounds the drift between the waves
			// (see the header comment) -- a wave of blocks past the end of the frame must not run rounds ahead and recycle
			// control slots the others still read
			const uint32_t* gb = &s_ctl[C_GBASE + (pslot * ENC_WAVES + wave) * 2];
			lds_wait(gb + 1, it, 0, A.ctrl, lane);
			asm volatile("" ::: "memory");
			if (p_len == 0) return;
			const uint32_t base = __builtin_amdgcn_readfirstlane(lds_ld(gb));
			PSTAMP(6);
			wave_copy_own_u(s_stage0 + (((it - 1) & 1) * ENC_WAVES + wave) * WSLOT, A.out + (size_t)p_f * A.out_stride + base, p_len, lane);
		};

		if (!cur.have) {                                       // final drain: item it-1 is the last one
			if (have_prev) {
				if (duty) resolve_prev();
				copy_out_prev();
			}
			return false;
		}

		const bool is_i = ((A.first_fc + f) & 3u) == 0;
		uint8_t* wslot = s_stage0 + ((it & 1) * ENC_WAVES + wave) * WSLOT;   // free since this wave's copy-out of item it-2
		uint8_t* scratch = wslot + 16;
		if (new_tile) {
			if (((A.first_fc + g.f_lo) & 3u) != 0) {           // GOP started in an earlier batch
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const uint2 q = *(const uint2*)(A.ientries_in + g.poff + r * A.w);
					ip[2 * r] = q.x; ip[2 * r + 1] = q.y;
				}
			} else {
#pragma unroll
				for (int m = 0; m < 8; m++) ip[m] = 0;
			}
		}
		// ---- the pixels of the item ENC_PFDEPTH ahead, into the registers this item's pixels leave
		auto prefetch_next = [&]() {
			asm volatile("" ::: "memory");
			publish_ticket();
			if (pf.have) { load_frame(pf.g, A.pix + (size_t)pf.f * npx, px); advance(pf, true); }
		};
#ifdef ENC_PROF
		PSTAMP(9);                                             // loop top: bookkeeping, duty status prefetch, I-frame entry plane
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		PSTAMP(8);                                             // wait for this item's pixels
		prof[10]++;
#endif
#if ENC_PRIO
		__builtin_amdgcn_s_setprio(ENC_PRIO);
#endif
		uint32_t ep[8], e0, len;
		bool copy, fill;
		{
		// ---- (Q) colour -> entry through the exact table, lane = (block, row)
		uint32_t eq[16];
		const uint32_t pxv[16] = {px[0].x, px[0].y, px[0].z, px[0].w, px[1].x, px[1].y, px[1].z, px[1].w,
		                          px[2].x, px[2].y, px[2].z, px[2].w, px[3].x, px[3].y, px[3].z, px[3].w};
#pragma unroll
		for (int k = 0; k < 16; k++) {
			if (ENTRIES) eq[k] = pxv[k] & (M512 ? 0x1FFu : 0xFFu);
			else eq[k] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(lut_rs, lut_offset(pxv[k]), 0, ENC_LUTAUX);
		}
		PSTAMP(0);
#if ENC_PRIO
		__builtin_amdgcn_s_setprio(0);
#endif
		PSTAMP(2);
		// [block][pixel] u16 table in the wave's scratch; a lane writes its row: 8 bytes at i*512 + lane*8
#pragma unroll
		for (int i = 0; i < 4; i++) {
			uint2 q;
			q.x = eq[i * 4 + 0] | (eq[i * 4 + 1] << 16);
			q.y = eq[i * 4 + 2] | (eq[i * 4 + 3] << 16);
			*(uint2*)(scratch + i * 512 + lane * 8) = q;
		}
		PSTAMP(1);
		if (duty) resolve_prev();
		PSTAMP(3);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// ---- transpose: lane = block reads its 16 entries (32 contiguous bytes), kept PACKED two per register
		{
			const uint4 lo = *(const uint4*)(scratch + lane * 32), hi = *(const uint4*)(scratch + lane * 32 + 16);
			ep[0] = lo.x; ep[1] = lo.y; ep[2] = lo.z; ep[3] = lo.w; ep[4] = hi.x; ep[5] = hi.y; ep[6] = hi.z; ep[7] = hi.w;
		}
		// ---- (C) block tests. count1 = CompareIFrameBlock vs the top-left entry colour
		// (src/agmv_encode.c:302-352), count2 = ComparePFrameBlock vs the I-frame entries
		// (src/agmv_encode.c:240-300); one matrix bit per pixel (the shifter uses the low 5 bits of its amount).
		e0 = ep[0] & 0xffffu;
		const uint32_t row0 = e0 * MROW;
		uint32_t acc1 = 0, acc2 = 0, nesc = 0;
		// (the I / P choice is wave-uniform: unswitched by hand -- with the test inside the unrolled loop the compiler
		//  branches per entry pair and waits for each pair's two matrix words before it issues the next reads)
		if (is_i) block_tests<M512, false>(ep, ip, s_mtx, row0, acc1, acc2, nesc);
		else block_tests<M512, true>(ep, ip, s_mtx, row0, acc1, acc2, nesc);
		const uint32_t count1 = __popc(acc1), count2 = __popc(acc2);
		copy = !is_i && count2 >= COPY_COUNT;                  // COPY has priority, :465
		fill = !copy && count1 >= FILL_COUNT;
		if (copy) len = 1;
		else if (fill) len = M512 ? (2u + ((e0 & 0xffu) >= 127u ? 1u : 0u)) : 2u;
		else len = 17u + nesc;
		if (!g.valid) len = 0;
		}

		if (is_i) {                                            // iframe_entries = img_entry, :626-630
#pragma unroll
			for (int m = 0; m < 8; m++) ip[m] = ep[m];
			if (A.ientries_out && (uint32_t)f == A.last_iframe && g.valid) {
#pragma unroll
				for (int r = 0; r < 4; r++) {
					uint2 q;
					q.x = ep[2 * r]; q.y = ep[2 * r + 1];
					*(uint2*)(A.ientries_out + g.poff + r * A.w) = q;
				}
			}
		}

		// ---- byte offsets inside the wave; the last wave to arrive publishes the tile's aggregate
		const uint32_t incl = wave_incl_scan(len, lane);
		const uint32_t wtot = __builtin_amdgcn_readlane(incl, 63);
		uint32_t arrived = 0;
		if (lane == 63) {
			lds_st(&s_ctl[C_WSUM + slot * ENC_WAVES + wave], (tag << 16) | incl);
			asm volatile("" ::: "memory");
			arrived = atomicAdd(&s_ctl[C_ARRIVE + slot], 1u);
		}
		arrived = __builtin_amdgcn_readlane(arrived, 63);
		if (arrived == ENC_WAVES - 1) {
			const uint32_t ws = lane < ENC_WAVES ? (lds_ld(&s_ctl[C_WSUM + slot * ENC_WAVES + lane]) & 0xffffu) : 0u;
			const uint32_t total = wave_sum(ws);
			if (lane == 0) {
				lds_st(&s_ctl[C_ARRIVE + slot], 0u);
				__hip_atomic_store(A.status + (size_t)f * A.tpf + g.tile, (g.tile == 0 ? ST_PREFIX : ST_AGG) | total,
				                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				lds_st(&s_ctl[C_TTOTAL + slot], (tag << 16) | total);
			}
		}
		PSTAMP(4);
		// ---- (E) emit this block's bytes into the wave's stage slot
		// Codes are built two at a time in packed 16-bit lanes and written as {code, index} byte PAIRS at byte-granular
		// LDS addresses (gfx950 runs DS in unaligned mode): when an entry has no escape byte its pair's second byte is
		// overwritten by the next pair, and the one byte a block may spill past its end is the next block's flag --
		// which is why the flags are written last.  The LDS unit executes a wave's writes in program order.
		if (g.valid) {
			uint8_t* sp = wslot + 16 + incl - len;
			if (!copy && !fill) {
				if (M512) {
					uint32_t n = 0;                                // escape bytes so far
#pragma unroll
					for (int m = 0; m < 8; m++) {
						const uint32_t p = ep[m], idx2 = p & 0x00FF00FFu;
						const u16x2 c2 = __builtin_elementwise_min(__builtin_bit_cast(u16x2, idx2), __builtin_bit_cast(u16x2, 0x007F007Fu));
						const uint32_t code2 = ((p >> 1) & 0x00800080u) | __builtin_bit_cast(uint32_t, c2);   // :395-401
						const uint32_t w = __builtin_amdgcn_perm(code2, p, 0x02060004u);   // code_lo, idx_lo, code_hi, idx_hi
						const uint32_t e2 = idx2 + 0x00810081u;    // bit 8 / bit 24: index >= 127
						*(u16u*)(sp + n + (1 + 2 * m)) = (uint16_t)w;
						n += (e2 >> 8) & 1u;
						*(u16u*)(sp + n + (2 + 2 * m)) = (uint16_t)(w >> 16);
						n += e2 >> 24;
					}
				} else {
#pragma unroll
					for (int m = 0; m < 4; m++)                    // :428-429
						*(u32u*)(sp + 1 + 4 * m) = __builtin_amdgcn_perm(ep[2 * m + 1], ep[2 * m], 0x06040200u);
				}
			} else if (fill) {
				if (M512) {
					const uint32_t idx = e0 & 0xffu, p7 = (e0 >> 1) & 0x80u;
					*(u16u*)(sp + 1) = (uint16_t)(p7 | (idx < 127u ? idx : 127u) | (idx << 8));   // :382-388
				} else {
					sp[1] = (uint8_t)e0;                           // :421
				}
			}
			asm volatile("" ::: "memory");
			sp[0] = copy ? COPY_FLAG : (fill ? FILL_FLAG : NORMAL_FLAG);
		}
		PSTAMP(5);
		if (have_prev) copy_out_prev();
		prefetch_next();                                       // LAST in the item: any later wait of the item (the compiler places conservative ones at branch joins) would drain these loads -- requested right behind the entries they measured 0.98 against 0.72 ms per 256 frames
		PSTAMP(7);

		// ---- next item
		have_prev = true; p_tile = g.tile; p_f = f; p_len = wtot;
		it++;
		pre_next = ST_PREFIX;
		if (wave == (int)(it % ENC_WAVES) && p_tile != 0) pre_next = st_load(A.status + (size_t)p_f * A.tpf, (int)p_tile - 1 - lane);
		new_tile = advance(cur, false);
		return true;
	};
	for (;;) {
#pragma unroll
		for (int d = 0; d < ENC_PFDEPTH; d++)
			if (!body(pxs[d])) goto done;
	}
done:;
#ifdef ENC_PROF
	if (lane == 0)
		for (int k = 0; k < 11; k++) atomicAdd(A.ctrl + 32 + k, k == 10 ? prof[k] : prof[k] >> 6);
#endif
}

#ifdef AGMV_LAB_ENCODE_W
#include "lab/k_encode_w.inc"   // the independent-wave form of k_encode (measured slower; profiles/r03/k_encode_experiments.txt)
#endif

// ----------------------------------------------------------------------------------------------
// decode side
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ bool is_flag(uint32_t b) { return b == FILL_FLAG || b == NORMAL_FLAG || b == COPY_FLAG; }

struct ByteSrc {
	const uint8_t* p;
	uint32_t cap;
	__device__ __forceinline__ uint32_t operator()(uint32_t pos) const { return pos < cap ? p[pos] : 0u; }
};

// bytes of one frame read through an LDS window [lo, lo+len) staged by the workgroup; anything outside falls back to
// global memory (streams full of resync garbage can make a tile's byte range larger than the window)
struct StagedSrc {
	const __attribute__((address_space(3))) uint8_t* lds;      // explicit LDS pointer: ds_read_u8, never flat_load
	uint32_t lo, len;
	const uint8_t* p;
	uint32_t cap;
	__device__ __forceinline__ uint32_t operator()(uint32_t pos) const
	{
		const uint32_t d = pos - lo;
		if (d < len) return lds[d];
		return pos < cap ? p[pos] : 0u;
	}
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

// ----------------------------------------------------------------------------------------------
// K2 (parallel form).  The block loop of the reference (src/agmv_decode.c:226-320) is a chain:
// block k+1 is entered where block k ended, and an entry position that does not hold a flag byte
// slides forward to the next flag-valued byte (the resync of :236-243).  So the only positions
// that can start a block are the flag-valued bytes: the NODES.  A node at p ends at
//   COPY   -> p+1        FILL -> p+2 (+1 after an escape code, 512 colours)
//   NORMAL -> 16 codes of 1 or 2 bytes after p+1
// and its successor is the first node at or after that end.  next(p)-p <= 33, so a chunk of PC
// bytes is summarised by a map {entry offset 0..32} -> (exit offset, blocks counted).
// k_parse_chunks (one WAVE per chunk, no workgroup barriers): flags are found with ballots, ranked
// with popcounts into a dense node list, each node's end is computed by one lane, and the chain
// is resolved by pointer doubling over the node list -- the work is proportional to the number
// of blocks in the chunk, not to its bytes.  One wave per frame then threads the chunk maps
// together (k_parse_stitch), and k_parse_emit rebuilds the node list, marks the nodes of the
// true chain (three doubling levels + a walk in steps of 8 nodes) and writes the entry offsets,
// a node's block number being the popcount of marked nodes before it.
// ----------------------------------------------------------------------------------------------
constexpr int PC = 512;         // bytes per chunk (one wave)
constexpr int PNSEG = PC / 64;  // ballot segments per chunk
#ifndef PEL_OVERRIDE
#define PEL_OVERRIDE 3
#endif
constexpr int PCL = 3;          // k_parse_chunks: doubling rounds before the 33 entry lanes walk 2^PCL nodes at a time
constexpr int PEL = PEL_OVERRIDE;   // k_parse_emit: levels kept for marking; the chain is walked 2^PEL nodes at a time
constexpr int PHALO = 64;       // bytes staged beyond the chunk (a block spans <= 33)
constexpr uint32_t J_EXIT = 0x8000u;    // jump leaves the chunk: J_EXIT | offset into the next chunk
constexpr uint32_t J_END = 0xFFFFu;     // chain left the readable stream (position > bpos)
constexpr uint32_t X_END = 63u;         // chunk map: exit code of an ended chain

// geometry of the fast parser (k_fp_*, below); the robust kernels can deliver their result in its bitmap form
constexpr int FC = 64;                  // bytes per piece (one lane)
constexpr int FH = 4;                   // run-in pieces
constexpr int FOWN = 64 - FH - 1;       // pieces a region owns (lane 63 holds the piece behind it)
constexpr int FRB = FOWN * FC;          // bytes a region owns

struct ParseArgs {
	const uint8_t* bits;
	unsigned long long stride;
	const uint32_t* bpos;
	uint32_t cpf;           // chunk rows per frame in summ / centry (the worst case: frame f's chunk c is row f * cpf + c -- no prefix over the frames, no launch for one)
	uint16_t* summ;         // [chunk][33] exit<<10 | count   (exit X_END: chain ended in the chunk)
	uint32_t* centry;       // [chunk] kbase<<8 | entry offset (0xff: chain never reaches the chunk)
	uint32_t* offsets;
	uint32_t* nentered;
	uint32_t n_frames, nblk;
	const uint32_t* fstate; // != NULL: only the frames the fast path gave up on (fstate[f] == FS_BAD) are parsed here
	const uint32_t* nbad;   // != NULL: how many frames are FS_BAD (0: every kernel of this path returns at once, without a look at the frames)
	unsigned long long* vm; // != NULL: the result goes out as entry BITS (the fast parser's bitmaps, [frame][maxR * FOWN] words of 64 bytes of stream) instead of offsets[]
	uint32_t maxR;
};
constexpr uint32_t FS_OK = 0, FS_TODO = 1, FS_BAD = 2;

// chunks of frame f that the robust kernels parse: ceil((bpos + 1) / PC), or none when the frame is not theirs
__device__ __forceinline__ uint32_t parse_nch(const ParseArgs& A, uint32_t f, uint32_t bpos)
{
	return (!A.fstate || A.fstate[f] == FS_BAD) ? (bpos + PC) / PC : 0u;
}

__device__ __forceinline__ void wave_lds_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// LDS of one wave's chunk.  NLV jump tables: 2 (ping-pong, k_parse_chunks) or PEL+1 (kept levels, k_parse_emit).
// The two tables that only the node build needs (rank_at: position -> nodes before it = index of the first node at
// or after it; npos: node -> position) are not members: the caller lends them the space of tables that are first
// written after the build (4.75 KB / 5.75 KB per wave instead of 6.8 / 8.3: 33 / 27 waves per CU instead of 23 / 19).
template <int NLV>
struct ParseLds {
	uint8_t b[PC + PHALO];                  // staged bytes
	unsigned long long esc[PNSEG + 2];      // per 64 bytes (chunk + halo): which bytes are escape codes ((b & 0x7f) == 127)
	uint16_t jl[NLV][PC];                   // node -> jump target (node index | J_EXIT+offset | J_END)
};

constexpr int PCD = ((PC + PHALO) / 4 + 63) / 64;      // dwords of a chunk (+halo) per lane

// the chunk's bytes, dword-wide (cs and the slab stride are multiples of 4); bytes past the slab read as 0
__device__ __forceinline__ void load_chunk(uint32_t (&raw)[PCD], const uint8_t* fbits, uint32_t cap, uint32_t cs, int lane)
{
#pragma unroll
	for (int q = 0; q < PCD; q++) {
		const uint32_t pos = cs + 4u * (uint32_t)(q * 64 + lane);
		raw[q] = (q * 64 + lane < (PC + PHALO) / 4 && pos < cap) ? *(const uint32_t*)(fbits + pos) : 0u;
	}
}

// Build the node list of chunk [cs, cs+PC): rank_at, npos, and per node the end of its block (eo, bit 15 = the
// block counts, i.e. the next one starts inside the stream) and its successor (jl[0]).  Returns the node count.
template <bool M512, int NLV>
__device__ __forceinline__ uint32_t parse_chunk_nodes(ParseLds<NLV>& S, uint16_t* rank_at, uint16_t* npos, uint16_t* eo, uint16_t* n0,
                                                      const uint32_t (&raw)[PCD], uint32_t bpos, uint32_t cs, int lane)
{
	// ---- stage the chunk (+halo) the caller fetched (load_chunk) while the previous chunk was being parsed
#pragma unroll
	for (int q = 0; q < PCD; q++) {
		const int i = q * 64 + lane;
		if (i < (PC + PHALO) / 4) ((uint32_t*)S.b)[i] = raw[q];
	}
	wave_lds_sync();
	// ---- nodes = flag-valued bytes at positions <= bpos, ranked by ballot + popcount
	uint32_t mtot = 0, by[PNSEG];
#pragma unroll
	for (int sg = 0; sg < PNSEG; sg++) by[sg] = S.b[sg * 64 + lane];
#pragma unroll
	for (int sg = 0; sg < PNSEG; sg++) {
		const uint32_t p = sg * 64 + lane;
		const bool node = is_flag(by[sg]) && cs + p <= bpos;
		const unsigned long long m = __ballot(node);
		const uint32_t r = mtot + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
		rank_at[p] = (uint16_t)r;
		if (node) npos[r] = (uint16_t)p;
		mtot += (uint32_t)__popcll(m);
		if (M512) {
			const unsigned long long em = __ballot((by[sg] & 0x7fu) == 127u);
			if (lane == 0) S.esc[sg] = em;
		}
	}
	if (M512) {
		const unsigned long long em = __ballot((S.b[PC + lane] & 0x7fu) == 127u);    // the halo
		if (lane == 0) { S.esc[PNSEG] = em; S.esc[PNSEG + 1] = 0; }
	}
	wave_lds_sync();
	// ---- one lane per node: where its block ends, and the node that follows
	const bool more = cs + PC <= bpos;                         // the stream continues into the next chunk
	for (uint32_t k = lane; k < mtot; k += 64) {
		const uint32_t p = npos[k], byte = S.b[p];
		uint32_t e = p + 1u;
		if (byte == FILL_FLAG) e += M512 ? 1u + ((S.b[p + 1] & 0x7fu) == 127u ? 1u : 0u) : 1u;
		else if (byte == NORMAL_FLAG) {
			if (M512) {
				// 16 codes of 1 or 2 bytes: the escape bits of the 32 bytes behind the flag, walked in registers
				const uint32_t sg = e >> 6, sh = e & 63u;
				const unsigned long long lo = S.esc[sg], hi = S.esc[sg + 1];
				const uint32_t m = (uint32_t)(sh ? (lo >> sh) | (hi << (64u - sh)) : lo);
				uint32_t pos = 0;
#pragma unroll
				for (int i = 0; i < 16; i++) pos += 1u + ((m >> pos) & 1u);
				e += pos;
			} else e += 16u;
		}
		uint32_t j = J_END, n = 0;
		if (cs + e <= bpos) {                                  // block k+1 starts inside the stream: this one counts
			n = 1;
			if (e >= (uint32_t)PC) j = J_EXIT | (e - PC);
			else {
				const uint32_t nx = rank_at[e];                    // a non-flag entry slides to the next node (:236-243)
				j = nx < mtot ? nx : (more ? J_EXIT : J_END);
			}
		}
		S.jl[0][k] = (uint16_t)j;
		if (eo) eo[k] = (uint16_t)(e | n << 15);
		if (n0) n0[k] = (uint16_t)n;
	}
	wave_lds_sync();
	return mtot;
}

template <bool M512>
__global__ __launch_bounds__(64) void k_parse_chunks(ParseArgs A)
{
	__shared__ ParseLds<2> S;
	__shared__ uint16_t nn[2][PC];                             // node -> blocks counted along its jump (ping-pong)
	const int lane = threadIdx.x;
	if (A.nbad && *A.nbad == 0) return;
	// 2-D grid: y strides over frames, x over the chunks of a frame (no search for the frame of a chunk)
	for (uint32_t f = blockIdx.y; f < A.n_frames; f += gridDim.y) {
	const uint32_t bpos = A.bpos[f], g0 = f * A.cpf, nch = parse_nch(A, f, bpos);
	const uint8_t* fbits = A.bits + (size_t)f * A.stride;
	uint32_t nxt[PCD];
	if (blockIdx.x < nch) load_chunk(nxt, fbits, (uint32_t)A.stride, blockIdx.x * PC, lane);
	for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
		const uint32_t g = g0 + c, cs = c * PC;
		uint32_t raw[PCD];
#pragma unroll
		for (int q = 0; q < PCD; q++) raw[q] = nxt[q];
		if (c + gridDim.x < nch) load_chunk(nxt, fbits, (uint32_t)A.stride, (c + gridDim.x) * PC, lane);   // in flight over this chunk
		if (A.vm && lane < PC / 64) A.vm[(size_t)f * A.maxR * FOWN + c * (PC / 64) + lane] = 0ull;   // k_parse_emit ORs the entry bits in (a later launch); FOWN words per region, linear in the byte position
		const uint32_t mtot = parse_chunk_nodes<M512>(S, /*rank_at*/ nn[1], /*npos*/ S.jl[1], nullptr, nn[0], raw, bpos, cs, lane);
		const uint32_t k0 = lane < 33 ? nn[1][lane] : 0u;         // first node at or after entry offset `lane` (rank_at dies below)
		wave_lds_sync();
		// ---- PCL rounds of pointer doubling over the node list (jump + blocks counted along it), then the 33 entry
		// lanes walk their chains 2^PCL nodes at a time
#pragma unroll
		for (int lv = 0; lv < PCL; lv++) {
			const uint16_t *js = S.jl[lv & 1], *ns = nn[lv & 1];
			uint16_t *jd = S.jl[(lv + 1) & 1], *nd = nn[(lv + 1) & 1];
			for (uint32_t k = lane; k < mtot; k += 64) {
				uint32_t j = js[k], n = ns[k];
				if (j < J_EXIT) { n += ns[j]; j = js[j]; }
				jd[k] = (uint16_t)j;
				nd[k] = (uint16_t)n;
			}
			wave_lds_sync();
		}
		if (lane < 33) {
			uint32_t ex = X_END, cnt = 0;
			if (cs + lane <= bpos) {
				uint32_t k = k0;
				if (k >= mtot) { if (cs + PC <= bpos) ex = 0; }
				else {
					const uint16_t *jf = S.jl[PCL & 1], *nf = nn[PCL & 1];
					do { cnt += nf[k]; k = jf[k]; } while (k < J_EXIT);
					if (k != J_END) ex = k & 0x3Fu;
				}
			}
			A.summ[(size_t)g * 33 + lane] = (uint16_t)(ex << 10 | cnt);
		}
		wave_lds_sync();
	}
	}
}

// one wave per frame: thread the chunk maps together.  Rows are fetched a batch of 32 chunks ahead (lane j holds
// map[j]; unconditional clamped loads so that a whole batch is in flight while the previous one is consumed); the
// chain state (entry offset, blocks so far) is wave-uniform, so the dependent step is a v_readlane and scalar
// arithmetic, not a memory access.
constexpr int PSB = 32;
__global__ __launch_bounds__(64) void k_parse_stitch(ParseArgs A)
{
	const int lane = threadIdx.x;
	if (A.nbad && *A.nbad == 0) return;
	for (uint32_t f = blockIdx.x; f < A.n_frames; f += gridDim.x) {
	const uint32_t c0 = f * A.cpf, nch = parse_nch(A, f, A.bpos[f]);     // nch >= 1 for a frame that is parsed here
	if (nch == 0) continue;
	const uint16_t* rows = A.summ + (size_t)c0 * 33 + (lane < 33 ? lane : 0);
	uint32_t o = 0, kb = 0;
	uint32_t nxt[PSB];
#pragma unroll
	for (int u = 0; u < PSB; u++) nxt[u] = rows[(size_t)min((uint32_t)u, nch - 1u) * 33];
	for (uint32_t c = 0; c < nch; c += PSB) {
		uint32_t row[PSB];
#pragma unroll
		for (int u = 0; u < PSB; u++) row[u] = nxt[u];
#pragma unroll
		for (int u = 0; u < PSB; u++) nxt[u] = rows[(size_t)min(c + PSB + u, nch - 1u) * 33];
		uint32_t mine = 0;                                     // lane u: centry of chunk c+u
#pragma unroll
		for (int u = 0; u < PSB; u++) {
			if (lane == u) mine = kb << 8 | o;
			if (c + u < nch && o != 0xFFu) {
				const uint32_t v = __builtin_amdgcn_readlane(row[u], __builtin_amdgcn_readfirstlane(o));
				kb += v & 0x3FFu;
				o = (v >> 10) == X_END ? 0xFFu : v >> 10;
			}
		}
		if (lane < PSB && c + lane < nch) A.centry[c0 + c + lane] = mine;
	}
	if (lane == 0) A.nentered[f] = min(A.nblk, kb + 1u);
	}
}

template <bool M512>
__global__ __launch_bounds__(64) void k_parse_emit(ParseArgs A)
{
	__shared__ ParseLds<PEL + 1> S;
	__shared__ uint16_t eo[PC];                                // node -> end of its block (chunk-relative, <= PC+32) | counts << 15
	uint8_t* mark = S.b;                                       // chain marks: in the staged bytes' space once the nodes are built
	static_assert(PEL >= 2 && PC + PHALO >= PC, "rank_at / npos borrow jl[PEL-1] / jl[PEL], the marks borrow the bytes");
	const int lane = threadIdx.x;
	if (A.nbad && *A.nbad == 0) return;
	for (uint32_t f = blockIdx.y; f < A.n_frames; f += gridDim.y) {
	const uint32_t bpos = A.bpos[f], g0 = f * A.cpf, nch = parse_nch(A, f, bpos);
	const uint8_t* fbits = A.bits + (size_t)f * A.stride;
	uint32_t* off = A.offsets + (size_t)f * A.nblk;
	uint32_t nxt[PCD], ce_n = 0;
	if (blockIdx.x < nch) { ce_n = A.centry[g0 + blockIdx.x]; load_chunk(nxt, fbits, (uint32_t)A.stride, blockIdx.x * PC, lane); }
	for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
		const uint32_t ce = ce_n, o = ce & 0xFFu, kb = ce >> 8;
		uint32_t raw[PCD];
#pragma unroll
		for (int q = 0; q < PCD; q++) raw[q] = nxt[q];
		if (c + gridDim.x < nch) {                             // next chunk: in flight over this one
			ce_n = A.centry[g0 + c + gridDim.x];
			load_chunk(nxt, fbits, (uint32_t)A.stride, (c + gridDim.x) * PC, lane);
		}
		unsigned long long* fvm = A.vm ? A.vm + (size_t)f * A.maxR * FOWN : nullptr;
		if (c == 0 && lane == 0) {                             // block 0 is entered at byte 0
			if (fvm) atomicOr(fvm, 1ull); else off[0] = 0;
		}
		if (o == 0xFFu) continue;                              // the chain ended before this chunk (uniform)
		if (kb + 1u >= A.nblk) continue;                       // every block this chunk could enter is beyond the frame
		const uint32_t cs = c * PC;
		const uint32_t mtot = parse_chunk_nodes<M512>(S, /*rank_at*/ S.jl[PEL - 1], /*npos*/ S.jl[PEL], eo, nullptr, raw, bpos, cs, lane);
		const uint32_t k0 = cs + o <= bpos ? S.jl[PEL - 1][o] : mtot;
		for (uint32_t k = lane; k < mtot; k += 64) mark[k] = 0;
		wave_lds_sync();
		if (k0 < mtot) {
			// PEL rounds of pointer doubling, then one lane walks the true chain 2^PEL nodes at a time and the kept
			// levels fill in the nodes between (every node 2^lv steps behind a marked one)
#pragma unroll
			for (int lv = 0; lv < PEL; lv++) {
				for (uint32_t k = lane; k < mtot; k += 64) {
					uint32_t j = S.jl[lv][k];
					if (j < J_EXIT) j = S.jl[lv][j];
					S.jl[lv + 1][k] = (uint16_t)j;
				}
				wave_lds_sync();
			}
			if (lane == 0)
				for (uint32_t k = k0; k < J_EXIT; k = S.jl[PEL][k]) mark[k] = 1;
			wave_lds_sync();
#pragma unroll
			for (int lv = PEL - 1; lv >= 0; lv--) {
				for (uint32_t k = lane; k < mtot; k += 64) {
					const uint32_t j = S.jl[lv][k];
					if (mark[k] && j < J_EXIT) mark[j] = 1;
				}
				wave_lds_sync();
			}
			// nodes are in stream order, so a counting node's rank on the chain is the number of marked counting
			// nodes before it; it is block kb+rank and ends where block kb+rank+1 is entered
			uint32_t base = kb + 1u;
			for (uint32_t kg = 0; kg < mtot; kg += 64) {
				const uint32_t k = kg + lane;
				const uint32_t ev = k < mtot ? eo[k] : 0u;
				const bool on = k < mtot && mark[k] && (ev & 0x8000u);
				const unsigned long long m = __ballot(on);
				const uint32_t kk = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				if (on && kk < A.nblk) {
					const uint32_t pos = cs + (ev & 0x7FFFu);
					if (fvm) atomicOr(fvm + (pos >> 6), 1ull << (pos & 63u)); else off[kk] = pos;
				}
				base += (uint32_t)__popcll(m);
			}
		}
		wave_lds_sync();
	}
	}
}

// ----------------------------------------------------------------------------------------------
// K2 (fast form): speculate, then PROVE.  The chain of block entries of a frame is unique: if a set of walks, one per
// 64-byte piece of the stream, is such that every walk starts exactly where the walk of the piece before left off and
// the first one starts at byte 0, their concatenation IS the reference's parse.  Chains that start at different bytes
// merge within a few blocks (every block start the true chain passes is a flag byte the other chain will usually hit),
// so a lane that walks its piece from "the first flag byte of the piece" almost always ends where the true chain ends:
//   k_fp_walk   one wave per REGION of 59 pieces (+4 pieces of run-in before it, +1 behind it for the spill of the
//               last block).  Every lane walks its piece from the first flag byte (types and lengths from bit masks
//               built once per piece: flag bytes F, COPY bytes C, two-byte FILLs L, escape codes E; a step takes a whole
//               run of COPYs or of two-byte FILLs; NORMAL lengths are computed for many lanes at once).  Then each lane
//               takes the exit of the lane before it as its true entry: if that entry slides onto a node of the walk it
//               already has, the walk is trimmed; otherwise it walks from there until it hits a node of the old walk
//               (merge) or leaves the piece.  Repeated until no lane's exit changes (1-2 rounds).
//               Per piece: the bitmap V of block ENTRY positions (what offsets[] holds); per region: E (exit of the
//               run-in = assumed entry of the region), X (exit of its last piece), N (entries).
//   k_fp_finish one wave per frame: region r is proven when E[r] == X[r-1] (region 0 starts at byte 0 by definition).
//               Regions that are not are walked again, in order, with X[r-1] as a FORCED entry (an exit that changes
//               carries on into the next region); a frame that needs more than FP_REPAIRS of those is left to the robust
//               parser above.  Proven frames: exclusive sums of N -> first block number of every region, nentered.
//   k_fp_expand bitmaps -> offsets[]: per piece, the lanes whose bit is set write their position at the rank of the bit.
// Exit / entry codes: 0..33 = the next block is entered at that byte of the next piece; FX_SLIDE = no new entry, the
// resync (src/agmv_decode.c:236-243) continues into the next piece; FX_END = the chain ended.
// ----------------------------------------------------------------------------------------------
constexpr int FROW = FC / 4 + 1;        // LDS dwords per piece (odd stride: no bank conflicts between the lanes' pieces)
constexpr uint32_t FX_SLIDE = 64u, FX_END = 65u, FX_UNSET = 66u, FX_MERGE = 128u;
constexpr int FP_REPAIRS = 24;          // regions of one frame walked again (serially, by the frame's wave) before the frame is given up
constexpr int FP_LDS = 64 * FROW + 1;   // (+1: lane 63's look at "the piece behind" stays inside)

struct FpArgs {
	const uint8_t* bits;
	unsigned long long stride;
	const uint32_t* bpos;
	uint4* rec;                 // [n_frames][maxR]  x = E, y = X, z = N
	unsigned long long* vm;     // [n_frames][maxR][FOWN] entry bitmaps
	uint32_t* kb;               // [n_frames][maxR] first block number of the region
	uint32_t* fstate;           // [n_frames] FS_OK / FS_BAD
	uint32_t* offsets;
	uint32_t* nentered;
	uint32_t n_frames, nblk, maxR;
	uint32_t* nbad;             // frames given up (FS_BAD), counted by k_fp_finish
	uint32_t* tidx;             // [n_frames][tpfd + 1] byte position at which the first block of every k_decode tile is entered (TIDX_NONE: not entered)
	uint32_t tpfd;              // k_decode tiles per frame
	uint32_t* dirty;            // != NULL: k_decode's repair bitmap, cleared by k_fp_tiles (the last parser launch in front of it)
	uint32_t ndirty;
};
constexpr uint32_t TIDX_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t ctz64(unsigned long long m) { return (uint32_t)__builtin_ctzll(m); }       // m != 0
__device__ __forceinline__ unsigned long long above(uint32_t q) { return (~0ull << q) << 1; }     // bits > q (q <= 63)

// one region (see above) by one wave; forced = FX_UNSET: the entry of the region is what the run-in pieces give
template <bool M512>
__device__ __forceinline__ void fp_walk_region(const FpArgs& A, uint32_t* s_b, uint32_t f, uint32_t r, uint32_t forced, uint32_t bpos, int lane)
{
	const uint8_t* fbits = A.bits + (size_t)f * A.stride;
	const uint32_t cap = (uint32_t)A.stride;
	// ---- stage the 64 pieces (run-in, own, one behind); bytes before the frame or past the slab read as 0
	const long sb = (long)r * FRB - FH * FC;
	uint32_t raw[16];
	const long pos0 = sb + 4 * lane;
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const long pos = pos0 + 256 * k;
		raw[k] = (pos >= 0 && pos + 4 <= (long)cap) ? *(const uint32_t*)(fbits + pos) : 0u;
	}
	{
		// dword i = 64 k + lane of the span is dword j = lane & 15 of piece 4 k + (lane >> 4): one base address + constants
		uint32_t* row = s_b + (lane >> 4) * FROW + (lane & 15);
#pragma unroll
		for (int k = 0; k < 16; k++) row[k * 4 * FROW] = raw[k];
	}
	wave_lds_sync();
	// ---- this lane's piece as bit masks, four bytes at a time: F flag bytes, C = 0x5E, L = 0x4E, E escape codes
	const long cb = sb + (long)lane * FC;                      // first byte of the piece
	unsigned long long F, C, L, E = 0;
	{
		// bit 7 of every byte that is 0 (exact per byte)
		auto zero7 = [](uint32_t v) -> uint32_t { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u; };
		// the four bits (bit 7 of each byte) of two dwords as one byte: a dot product of the bytes {0, 0x80} with the weights
		// 1, 2, 4, 8 / 16, 32, 64, 128 is 128 x that byte (v_dot4_u32_u8: two instructions where shifts and ors take a dozen).
		// FILL (0x4E) and COPY (0x5E) differ in bit 4 only: one zero test finds both, bit 4 tells them apart; the packed byte of a
		// pair of dwords comes out of the dot products times 128 and goes to its place with ONE shift
		uint32_t f2[2] = {0, 0}, c2[2] = {0, 0}, lc2[2] = {0, 0}, e2[2] = {0, 0};
		auto place = [](uint32_t acc, uint32_t x128, int sh) -> uint32_t { return acc | (sh == 0 ? x128 >> 7 : x128 << (sh - 7)); };
		auto pair128 = [](uint32_t z0, uint32_t z1) -> uint32_t {
			return __builtin_amdgcn_udot4(z1, 0x80402010u, __builtin_amdgcn_udot4(z0, 0x08040201u, 0u, false), false);
		};
#pragma unroll
		for (int j = 0; j < 16; j += 2) {
			const uint32_t w0 = s_b[lane * FROW + j], w1 = s_b[lane * FROW + j + 1];
			const uint32_t zlc0 = zero7((w0 | 0x10101010u) ^ 0x5E5E5E5Eu), zn0 = zero7(w0 ^ 0x2F2F2F2Fu);
			const uint32_t zlc1 = zero7((w1 | 0x10101010u) ^ 0x5E5E5E5Eu), zn1 = zero7(w1 ^ 0x2F2F2F2Fu);
			const int h = j >> 3, sh = 4 * (j & 7);
			lc2[h] = place(lc2[h], pair128(zlc0, zlc1), sh);
			c2[h] = place(c2[h], pair128(zlc0 & (w0 << 3), zlc1 & (w1 << 3)), sh);
			f2[h] = place(f2[h], pair128(zlc0 | zn0, zlc1 | zn1), sh);
			if (M512) e2[h] = place(e2[h], pair128(((w0 & 0x7F7F7F7Fu) + 0x01010101u) & 0x80808080u, ((w1 & 0x7F7F7F7Fu) + 0x01010101u) & 0x80808080u), sh);   // (byte & 0x7f) == 127
		}
		F = (unsigned long long)f2[1] << 32 | f2[0]; C = (unsigned long long)c2[1] << 32 | c2[0];
		L = ((unsigned long long)lc2[1] << 32 | lc2[0]) ^ C; E = (unsigned long long)e2[1] << 32 | e2[0];
	}
	const long lim_l = (long)bpos - cb;                        // a block of this piece counts when it ends at or before this offset
	const int lim = lim_l > 1000 ? 1000 : (lim_l < -1000 ? -1000 : (int)lim_l);
	unsigned long long safe;                                   // positions at which any COPY / FILL ends at or before bpos
	{
		const long nv = lim_l + 1;                             // bytes of the piece at positions <= bpos: only those are nodes
		const unsigned long long ok = nv <= 0 ? 0ull : (nv < 64 ? (1ull << nv) - 1ull : ~0ull);
		F &= ok;
		safe = ok >> 3;
	}
	const bool more = lim >= FC;                               // the stream goes on behind this piece
	unsigned long long En = 0;                                 // escape codes of the piece behind (a NORMAL body spills <= 33 bytes)
	unsigned long long D;
	{
		unsigned long long L3 = 0;                             // FILLs whose index byte is an escape code: three bytes (L: the two-byte ones)
		if (M512) {
			const uint32_t lo = (uint32_t)__shfl_down((int)(uint32_t)E, 1, 64), hi = (uint32_t)__shfl_down((int)(uint32_t)(E >> 32), 1, 64);
			En = lane < 63 ? ((unsigned long long)hi << 32 | lo) : 0ull;
			L3 = L & ((E >> 1) | (En << 63));
			L &= ~L3;
		}
		D = (F & ~(C | L | L3)) | (F & ((L << 1) | (L3 << 1) | (L3 << 2))) | (F & ~safe);
	}
	// A STRETCH of the stream in which every flag byte is a COPY or a FILL that ends at or before bpos, and no flag-valued byte
	// lies inside the body of one of them, is walked in ONE step: the chain through it is exactly its flag bytes (a block's
	// successor is entered right behind its body and slides to the next flag byte -- which is the next flag byte of the stretch).
	// D = where a stretch must end: NORMAL blocks (their length needs the escape count), flag-valued bytes inside the body of a
	// FILL (which of the two is a block depends on the chain), blocks too close to bpos.  Conservative on purpose: a D bit
	// only hands the block at that byte to the one-block step below.  (D is computed above, where the three-byte FILLs are known.)
	const int first = forced != FX_UNSET ? FH : 0;              // first lane that walks (its entry: forced, or speculative)
	const bool walker = lane >= first && lane < 63 && cb >= 0;
	const uint32_t* mine = s_b + lane * FROW;
	const uint32_t behind = mine[FROW];                        // first dword of the piece behind
	unsigned long long V = 0, Q = 0;                           // entries / nodes of the lane's walk
	uint32_t xo = FX_UNSET, applied = FX_UNSET;
	uint32_t want = (lane == first && forced != FX_UNSET) ? forced : FX_SLIDE;
	for (int round = 0; round < 66; round++) {
		const bool need = walker && want != applied;
		if (__ballot(need) == 0) break;
		// ---- apply the entry `want`: trim the walk the lane has, or walk from the entry until it merges / leaves
		bool go = false;
		uint32_t q = 0, res = xo;
		unsigned long long Vw = 0, Qw = 0;
		if (need) {
			applied = want;
			if (want == FX_END) { V = 0; Q = 0; res = FX_END; }
			else {
				const uint32_t x = want == FX_SLIDE ? 0u : want;
				Vw = want == FX_SLIDE ? 0ull : 1ull << x;
				const unsigned long long m = F >> x;
				if (m == 0) { V = Vw; Q = 0; res = more ? FX_SLIDE : FX_END; }
				else {
					q = x + ctz64(m);
					if ((Q >> q) & 1ull) { V = Vw | (V & above(q)); Q &= ~0ull << q; }   // same chain from q on, same exit
					else go = true;
				}
			}
		}
		// The walk, written without divergent branches (selects on every lane).  One step takes a whole stretch (see D above;
		// it also ends where the old walk has a node) and then, if the chain arrives at a D byte, the one block there.
		uint32_t wres = FX_UNSET;
		for (;;) {
			if (__ballot(go) == 0) break;
			{
				const bool merged = go && ((Q >> q) & 1ull);
				wres = merged ? FX_MERGE + q : wres;
				go = go && !merged;
			}
			const unsigned long long hiq = ~0ull << q;             // bytes >= q
			const unsigned long long dq = (D & hiq) | (Q & (hiq << 1));
			const uint32_t d = dq ? ctz64(dq) : 64u;
			const bool str = go && d != q;
			if (__ballot(str) != 0) {                              // (wave-uniform: NORMAL-heavy streams rarely come here)
				const unsigned long long nodes = F & hiq & (d >= 64u ? ~0ull : ~(~0ull << d));
				const uint32_t last = 63u - (uint32_t)__builtin_clzll(nodes | 1ull);
				const uint32_t e = last + (((C >> last) & 1ull) ? 1u : (((L >> last) & 1ull) ? 2u : 3u));
				Qw |= str ? nodes : 0ull;
				// an entry behind every node (three-byte FILLs: what is neither COPY nor two-byte FILL); those at byte 64 and beyond belong to the next piece
				Vw |= str ? ((nodes & C) << 1) | ((nodes & L) << 2) | ((nodes & ~(C | L)) << 3) : 0ull;
				const unsigned long long m = F >> (e & 63u);
				const bool inside = e < 64u && m != 0;
				const uint32_t stop = e >= 64u ? e - 64u : (more ? FX_SLIDE : FX_END);
				wres = (str && !inside) ? stop : wres;
				const uint32_t p = e + ctz64(m | (1ull << 63));
				const bool arrived = str && inside;
				const bool merged = arrived && ((Q >> (p & 63u)) & 1ull);
				wres = merged ? FX_MERGE + p : wres;
				q = arrived ? p : q;
				go = go && (!str || (inside && !merged));
			}
			const bool sing = go && ((D >> q) & 1ull);
			if (__ballot(sing) != 0) {
				const uint32_t d0 = mine[q >> 2], d1 = mine[(q >> 2) + 1];
				const uint32_t two = __builtin_amdgcn_alignbyte((q >> 2) == 15u ? behind : d1, d0, q & 3u);
				const uint32_t t = two & 0xFFu;
				const bool isN = t == NORMAL_FLAG, isC = t == COPY_FLAG;
				uint32_t l1 = isC ? 1u : 2u + ((M512 && ((two >> 8) & 0x7Fu) == 127u) ? 1u : 0u);
				if (__ballot(sing && isN) != 0) {                  // NORMAL lengths: 16 dependent steps on the escape mask
					uint32_t len = 16;
					if (M512) {
						const uint32_t q1 = q + 1u;
						const uint32_t m = (uint32_t)((q1 < 64u ? E >> q1 : 0ull) | (En << (63u - q)));
						uint32_t pos = 0;
#pragma unroll
						for (int i = 0; i < 16; i++) pos += 1u + ((m >> pos) & 1u);
						len = pos;
					}
					if (isN) l1 = 1u + len;
				}
				const uint32_t e = q + l1;
				const bool over = (int)e > lim;                    // entered, not counted: the chain ends
				const bool cnt = sing && !over;
				Qw |= cnt ? 1ull << q : 0ull;
				Vw |= cnt ? (1ull << q) << l1 : 0ull;
				const unsigned long long m = F >> (e & 63u);
				const bool inside = cnt && e < 64u && m != 0;
				const uint32_t stop = over ? FX_END : (e >= 64u ? e - 64u : (more ? FX_SLIDE : FX_END));
				wres = (sing && !inside) ? stop : wres;
				q = inside ? e + ctz64(m) : q;
				go = go && (!sing || inside);
			}
		}
		if (need && wres != FX_UNSET) {
			if (wres >= FX_MERGE) {
				const uint32_t mq = wres - FX_MERGE;
				V = Vw | (V & above(mq)); Q = Qw | (Q & (~0ull << mq));
			} else { V = Vw; Q = Qw; res = wres; }
		}
		if (need) xo = res;
		// ---- next round: every lane's true entry is the exit of the lane before it
		const uint32_t px = (uint32_t)__shfl_up((int)xo, 1, 64);
		if (walker && lane > first) want = px;
	}
	// ---- results
	const bool own = lane >= FH && lane < 63;
	if (own) A.vm[((size_t)f * A.maxR + r) * FOWN + (lane - FH)] = walker ? V : 0ull;
	const uint32_t n = wave_sum(own && walker ? (uint32_t)__popcll(V) : 0u);
	const uint32_t ein = forced != FX_UNSET ? forced : (uint32_t)__builtin_amdgcn_readlane((int)xo, FH - 1);
	const uint32_t xout = (uint32_t)__builtin_amdgcn_readlane((int)xo, 62);
	if (lane == 0) A.rec[(size_t)f * A.maxR + r] = make_uint4(ein, xout, n, 0u);
	wave_lds_sync();
}

template <bool M512>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_fp_walk(FpArgs A)
{
	__shared__ uint32_t s_b[FP_LDS];
	const int lane = threadIdx.x;
	if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) *A.nbad = 0;   // (k_fp_finish, the next launch, counts the frames it gives up; saves a fill launch)
	for (uint32_t f = blockIdx.y; f < A.n_frames; f += gridDim.y) {   // (many small frames: the grid is capped, rows stride over the frames)
		const uint32_t bpos = A.bpos[f];
		const uint32_t nreg = min(bpos / FRB + 1u, A.maxR);    // positions 0 .. bpos can hold nodes
		for (uint32_t r = blockIdx.x; r < nreg; r += gridDim.x)
			fp_walk_region<M512>(A, s_b, f, r, r == 0 ? 0u : FX_UNSET, bpos, lane);   // block 0 is entered at byte 0
	}
}

// one wave per frame: prove the regions (see above), walk again those that are not, number the blocks
template <bool M512>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fp_finish(FpArgs A)
{
	__shared__ uint32_t s_b[FP_LDS];
	const uint32_t f = blockIdx.x;
	const int lane = threadIdx.x;
	const uint32_t bpos = A.bpos[f];
	const uint32_t nreg = min(bpos / FRB + 1u, A.maxR);
	uint4* rec = A.rec + (size_t)f * A.maxR;
	if (A.tidx)                                                // bitmap form: the frame's tile entries start out as "not entered" (k_fp_tiles, a later launch, writes those that are)
		for (uint32_t t = lane; t <= A.tpfd; t += 64) A.tidx[(size_t)f * (A.tpfd + 1) + t] = TIDX_NONE;
	// the first region at or behind `start` whose entry is not the exit of the region before it is walked again with that
	// exit forced; its own exit may have changed, so the search goes on right behind it
	int budget = FP_REPAIRS;
	for (uint32_t start = 1;;) {
		uint32_t bad = 0xFFFFFFFFu;
		for (uint32_t r0 = start & ~63u; r0 < nreg && bad == 0xFFFFFFFFu; r0 += 64) {
			const uint32_t r = r0 + lane;
			const bool in = r < nreg && r >= start;
			const uint32_t x = in ? __hip_atomic_load(&rec[r].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
			const uint32_t yp = in ? __hip_atomic_load(&rec[r - 1].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
			const unsigned long long bm = __ballot(in && x != yp);
			if (bm) bad = r0 + ctz64(bm);
		}
		if (bad == 0xFFFFFFFFu) break;
		if (budget-- == 0) {
			if (lane == 0) { A.fstate[f] = FS_BAD; atomicAdd(A.nbad, 1u); }
			return;
		}
		const uint32_t want = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&rec[bad - 1].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		fp_walk_region<M512>(A, s_b, f, bad, want, bpos, lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		start = bad + 1;
	}
	uint32_t run = 0;
	for (uint32_t r0 = 0; r0 < nreg; r0 += 64) {
		const uint32_t r = r0 + lane;
		const uint32_t n = r < nreg ? __hip_atomic_load(&rec[r].z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
		const uint32_t incl = wave_incl_scan(n, lane);
		if (r < nreg) A.kb[(size_t)f * A.maxR + r] = run + incl - n;
		run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	}
	if (lane == 0) { A.fstate[f] = FS_OK; A.nentered[f] = run < A.nblk ? run : A.nblk; }
}

// entry bitmaps -> offsets[]: the region's entries are listed in LDS (byte position inside the region, in order) and go
// out as coalesced rows.
__global__ __launch_bounds__(64) void k_fp_expand(FpArgs A)
{
	__shared__ uint16_t s_pos[FRB];
	const int lane = threadIdx.x;
	const uint32_t f = blockIdx.y;
	if (A.fstate[f] != FS_OK) return;
	const uint32_t nreg = min(A.bpos[f] / FRB + 1u, A.maxR);
	uint32_t* off = A.offsets + (size_t)f * A.nblk;
	// a wave takes every gridDim.x-th region of its frame; the next one's bitmaps and block number are requested before
	// this one's entries are listed
	uint32_t kb_n = 0;
	unsigned long long V_n = 0;
	if (blockIdx.x < nreg) {
		kb_n = A.kb[(size_t)f * A.maxR + blockIdx.x];
		V_n = lane < FOWN ? A.vm[((size_t)f * A.maxR + blockIdx.x) * FOWN + lane] : 0ull;
	}
	for (uint32_t r = blockIdx.x; r < nreg; r += gridDim.x) {
		const uint32_t kb = kb_n;
		unsigned long long V = V_n;
		if (r + gridDim.x < nreg) {
			kb_n = A.kb[(size_t)f * A.maxR + r + gridDim.x];
			V_n = lane < FOWN ? A.vm[((size_t)f * A.maxR + r + gridDim.x) * FOWN + lane] : 0ull;
		}
		if (kb >= A.nblk) break;                                // (uniform) blocks beyond the frame are never entered
		const uint32_t vlo = (uint32_t)V, vhi = (uint32_t)(V >> 32);
		const uint32_t n = (uint32_t)__popc(vlo) + (uint32_t)__popc(vhi);
		const uint32_t incl = wave_incl_scan(n, lane);
		const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const uint32_t first = incl - n;
		// pieces with few entries: every lane walks the set bits of its own (32 bits at a time: one-instruction bit scans);
		// dense pieces (a run of COPYs: up to 64 entries) one by one by the whole wave, each lane whose bit is set writing at
		// the rank of its bit
		unsigned long long dense = __ballot(n >= 32u);
		if (n < 32u) {
			uint32_t j = first, m = vlo;
			const uint32_t base = (uint32_t)lane * FC;
			while (m) { s_pos[j++] = (uint16_t)(base + (uint32_t)__builtin_ctz(m)); m &= m - 1u; }
			m = vhi;
			while (m) { s_pos[j++] = (uint16_t)(base + 32u + (uint32_t)__builtin_ctz(m)); m &= m - 1u; }
		}
		while (dense) {                                        // (uniform)
			const int c = (int)ctz64(dense);
			dense &= dense - 1ull;
			const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)vlo, c), hi = (uint32_t)__builtin_amdgcn_readlane((int)vhi, c);
			const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)first, c);
			if ((((unsigned long long)hi << 32 | lo) >> lane) & 1ull)
				s_pos[at + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u))] = (uint16_t)(c * FC + lane);
		}
		wave_lds_sync();
		for (uint32_t i = lane; i < tot; i += 64)
			if (kb + i < A.nblk) off[kb + i] = r * FRB + s_pos[i];
		wave_lds_sync();
	}
}

// position of the k-th set bit (k = 0: the lowest) of v; k < popcount(v)
__device__ __forceinline__ uint32_t select64(unsigned long long v, uint32_t k)
{
	uint32_t w = (uint32_t)v, base = 0, c = (uint32_t)__popc(w);
	if (k >= c) { k -= c; w = (uint32_t)(v >> 32); base = 32; }
	c = (uint32_t)__popc(w & 0xFFFFu); if (k >= c) { k -= c; w >>= 16; base += 16; }
	w &= 0xFFFFu;
	c = (uint32_t)__popc(w & 0xFFu);   if (k >= c) { k -= c; w >>= 8;  base += 8; }
	w &= 0xFFu;
	c = (uint32_t)__popc(w & 0xFu);    if (k >= c) { k -= c; w >>= 4;  base += 4; }
	w &= 0xFu;
	c = (uint32_t)__popc(w & 0x3u);    if (k >= c) { k -= c; w >>= 2;  base += 2; }
	return base + ((k >= (w & 1u)) ? 1u : 0u);
}

// Entry bitmaps -> where the first block of every k_decode tile (DEC_T consecutive blocks) is entered.  This is all
// k_decode needs besides the bitmaps themselves: it ranks its own blocks in the bitmap words between two tile entries
// (offsets[] -- 4 bytes per block written by k_fp_expand and read back -- never exists on this path).
// One wave per region, like k_fp_expand; tidx is pre-filled with TIDX_NONE.
__global__ __launch_bounds__(64) void k_fp_tiles(FpArgs A)
{
	const int lane = threadIdx.x;
	if (A.dirty && blockIdx.x == 0 && blockIdx.y == 0)         // (saves the fill launch in front of k_decode)
		for (uint32_t i = lane; i < A.ndirty; i += 64) A.dirty[i] = 0;
	for (uint32_t f = blockIdx.y; f < A.n_frames; f += gridDim.y) {
	const bool fell_back = A.fstate[f] == FS_BAD;              // its entry bits come from the robust kernels (k_parse_emit), nobody has numbered its blocks yet
	const uint32_t nreg = min(A.bpos[f] / FRB + 1u, A.maxR);
	uint32_t* tx = A.tidx + (size_t)f * (A.tpfd + 1);
	for (uint32_t r = blockIdx.x; r < nreg; r += gridDim.x) {
		uint32_t kb;
		if (fell_back) {
			// first block number of the region = entries in the regions before it: counted here, by the region's own wave (the
			// exception path: a launch of its own for this cost 5 us on every decode call that had nothing to count)
			const unsigned long long* v = A.vm + (size_t)f * A.maxR * FOWN;
			uint32_t n = 0;
			for (uint32_t i = lane; i < r * FOWN; i += 64) n += (uint32_t)__popcll(v[i]);
			kb = wave_sum(n);
			if (lane == 0) A.kb[(size_t)f * A.maxR + r] = kb;
		} else kb = A.kb[(size_t)f * A.maxR + r];
		if (kb >= A.nblk) {                                     // (uniform) blocks beyond the frame are never entered
			if (fell_back) continue;                            // (its later regions still get their number: bm_offset_of searches kb[] of the whole frame)
			break;
		}
		const unsigned long long V = lane < FOWN ? A.vm[((size_t)f * A.maxR + r) * FOWN + lane] : 0ull;
		const uint32_t n = (uint32_t)__popcll(V);
		const uint32_t incl = wave_incl_scan(n, lane);
		const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		const uint32_t first = incl - n;
		uint32_t end = kb + tot;
		if (end > A.nblk) end = A.nblk;
		// tiles whose first block is entered in this region: m0 .. m0 + nt - 1 (uniform).  The kernel is bound by VALU issue, so
		// the bit select runs ONCE, on lane j for tile m0 + j: per tile only the piece that holds its first block is found (the
		// pieces' inclusive counts are monotone: it is the number of pieces that end at or before the target) and its bitmap
		// word and rank are handed to lane j.
		const uint32_t m0 = (kb + DEC_T - 1) / DEC_T;
		uint32_t nt = end > m0 * DEC_T ? (end - m0 * DEC_T + DEC_T - 1) / DEC_T : 0u;
		for (uint32_t j0 = 0; j0 < nt; j0 += 64) {                 // (more than 64 tiles per region: never with DEC_T = 256)
			const uint32_t cnt = min(nt - j0, 64u);
			uint32_t xlo = 0, xhi = 0, xr = 0, xo = 0;
			for (uint32_t j = 0; j < cnt; j++) {
				const uint32_t target = (m0 + j0 + j) * DEC_T - kb;
				const int ol = (int)__popcll(__ballot(incl <= target));     // < 64: target < tot
				const uint32_t vl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)V, ol), vh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(V >> 32), ol);
				const uint32_t fo = (uint32_t)__builtin_amdgcn_readlane((int)first, ol);
				const bool me = (uint32_t)lane == j;
				xlo = me ? vl : xlo; xhi = me ? vh : xhi; xr = me ? target - fo : xr; xo = me ? (uint32_t)ol : xo;
			}
			if ((uint32_t)lane < cnt)
				tx[m0 + j0 + lane] = r * FRB + xo * FC + select64((unsigned long long)xhi << 32 | xlo, xr);
		}
	}
	}
}

// entry position of block b of frame f from the bitmaps (b < nentered[f]); the slow, self-contained form: k_fixup and
// the tiles of k_decode whose entries span more bitmap words than the workgroup has lanes
__device__ uint32_t bm_offset_of(const unsigned long long* vm, const uint32_t* kb, uint32_t maxR, uint32_t bpos, uint32_t f, uint32_t b)
{
	const uint32_t nreg = min(bpos / FRB + 1u, maxR);
	const uint32_t* kbp = kb + (size_t)f * maxR;
	uint32_t lo = 0, hi = nreg - 1;
	while (lo < hi) {                                          // the last region whose first block number is <= b
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (kbp[mid] <= b) lo = mid; else hi = mid - 1;
	}
	uint32_t rem = b - kbp[lo];
	const unsigned long long* v = vm + ((size_t)f * maxR + lo) * FOWN;
	for (int k = 0; k < FOWN; k++) {
		const unsigned long long w = v[k];
		const uint32_t c = (uint32_t)__popcll(w);
		if (rem < c) return lo * FRB + (uint32_t)k * FC + select64(w, rem);
		rem -= c;
	}
	return 0;                                                  // (not reached for b < nentered)
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
	uint32_t grp0;          // first GOP of this launch (the grid covers GOPs grp0 .. grp0 + gridDim.x / tpf - 1)
	// bitmap form (BM kernels): the parser's entry bitmaps, first block number per region, tile entries -- no offsets[]
	const unsigned long long* vm;
	const uint32_t* kb;
	const uint32_t* tidx;
	uint32_t maxR;
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

// decode_block for a block whose bytes sit in the workgroup's LDS window, with the dependent byte -> byte -> palette
// round trips of the reference walk taken apart: ONE round of reads fetches the flag, the two bytes behind it and the
// 36 bytes a NORMAL body can span; which of those are escape codes ((b & 0x7f) == 127) becomes a bit mask, the 16
// code positions are walked in registers, and the 32 code bytes and then the 16 palette entries are fetched as
// independent reads.  Anything unusual (no flag at the entry offset, a block that over-runs bpos, bytes outside the
// window) takes the generic walk above, which is the reference's loop verbatim.
template <bool M512>
__device__ __forceinline__ void decode_block_staged(const StagedSrc& src, uint32_t off, const uint32_t bpos,
                                                    const uint32_t* pal, uint32_t (&cur)[16], const uint32_t (&icol)[16],
                                                    bool istale, bool& stale, bool& fill_written)
{
	typedef const __attribute__((address_space(3))) uint8_t* lds8;
	typedef const __attribute__((address_space(3))) uint32_t* lds32;
	const uint32_t d = off - src.lo;
	bool slow = true;
	fill_written = false;
	if (off >= src.lo && d + 44u <= src.len) {                 // flag + 33 bytes + alignment slack inside the window
		const lds8 p = src.lds + d;
		const uint32_t b0 = p[0], b1 = p[1], b2 = p[2];
		uint32_t m = 0;                                        // bit t: the byte at off+1+t is an escape code
		if (M512) {
			const uint32_t d1 = d + 1u, a = d1 & ~3u;
			unsigned long long em = 0;
#pragma unroll
			for (int k = 0; k < 9; k++) {
				const uint32_t w = *(lds32)(src.lds + a + 4u * k);
				const uint32_t z = ((w & 0x7f7f7f7fu) + 0x01010101u) & 0x80808080u;     // bit 7 of every byte equal to 127
				const uint32_t nib = __builtin_amdgcn_udot4(z, 0x08040201u, 0u, false) >> 7;    // those four bits, adjacent (bytes {0, 0x80} . weights 1, 2, 4, 8)
				em |= (unsigned long long)nib << (4 * k);
			}
			m = (uint32_t)(em >> (d1 & 3u));
		}
		if (b0 == COPY_FLAG) {                                 // no over-run check, :281-290
#pragma unroll
			for (int k = 0; k < 16; k++) cur[k] = icol[k];
			stale = istale;
			slow = false;
		} else if (b0 == FILL_FLAG) {
			uint32_t ci = b1, end = off + 2u;
			if (M512) {
				const bool esc = (b1 & 0x7fu) == 127u;
				ci = ((b1 & 0x80u) << 1) + (esc ? b2 : (b1 & 0x7fu));
				end += esc ? 1u : 0u;
			}
			const uint32_t color = pal[ci];
			if (!(end > bpos)) {
#pragma unroll
				for (int k = 0; k < 16; k++) cur[k] = color;
				stale = false;
				fill_written = true;
			}
			slow = false;
		} else if (b0 == NORMAL_FLAG) {
			uint32_t ci[16], pos = 0;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const uint32_t f0 = p[1u + pos], f1 = p[2u + pos];
				if (M512) {
					const uint32_t esc = (m >> pos) & 1u;
					ci[i] = ((f0 & 0x80u) << 1) + (esc ? f1 : (f0 & 0x7fu));
					pos += 1u + esc;
				} else {
					ci[i] = f0;
					pos += 1u;
				}
			}
			if (off + 1u + pos <= bpos) {                      // every code ends inside the stream: all 16 pixels are stored
#pragma unroll
				for (int k = 0; k < 16; k++) cur[k] = pal[ci[k]];
				stale = false;
				slow = false;
			}
		}
	}
	if (slow) decode_block<M512>(src, off, bpos, pal, cur, icol, istale, stale, fill_written);
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
		// written once, read by nobody on the device: non-temporal (measured 0.474 -> 0.386 ms per 256 x 1080p frames)
		typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
		__builtin_nontemporal_store((u32x4){q.x, q.y, q.z, q.w}, (u32x4*)(frame + poff + r * w));
	}
}

// buffer descriptor from wave-uniform inputs, made PROVABLY uniform (readfirstlane on both pointer halves and the size): otherwise
// hipcc wraps every buffer op in a waterfall loop
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, uint32_t bytes)
{
	const uint64_t a = (uint64_t)p;
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
	return __builtin_amdgcn_make_buffer_rsrc((void*)((uint64_t)hi << 32 | lo), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// K3: one lane = one 4x4 block carried through the <=4 frames of its GOP (img_data and
// iframe->img_data of the block live in registers).  A block whose value depends on a frame
// outside the GOP (not rewritten since the GOP started) is flagged in `dirty` and repaired by
// k_fixup; everything else is final.
//
// Everything the GOP needs is loaded UP FRONT, in two dependent round trips: (1) the block's entry offset in each
// of the four frames (+ nentered/bpos, palette, previous state), (2) after the workgroup has exchanged the byte
// range its entered blocks span in each frame, the four byte windows into a quad-buffered LDS stage.  The frame
// loop then reads LDS only -- no global load, no wait on the vector-memory counter, no barrier (except the last
// tile's neighbour exchange) -- so the 64 B/lane pixel stores of one frame drain while the next is reconstructed.
// (A wave's memory counter retires in order: with per-frame staging every wait for the next frame's bytes also
// waited for the previous frame's stores, and under saturating writes those round trips take 2-3 us.)
#ifndef DEC_WPE
#define DEC_WPE 5         // waves per SIMD: 87 VGPRs, no spills; 6 spills and is slower
#endif
template <bool M512, bool BM>
__global__ __launch_bounds__(DEC_T, DEC_WPE) void k_decode(DecArgs A)
{
	__shared__ uint32_t s_pal[512];
	__shared__ uint32_t s_nb[DEC_T];        // neighbour exchange for the last-block quirk
	__shared__ uint32_t s_nbstale[DEC_T];
	__shared__ __attribute__((aligned(16))) uint8_t s_bytes[4][DEC_STAGE];   // the tile's slice of each frame's bitstream
	__shared__ uint32_t s_rng[4][2];        // per frame: [0] lowest, [1] highest entry offset of the tile's entered blocks
	const int tid = threadIdx.x;
	const uint32_t npx = A.w * A.h;
	for (int i = tid; i < 512; i += DEC_T) s_pal[i] = A.pal[i];

	const uint32_t lgroup = blockIdx.x / A.tpf, tile = blockIdx.x - lgroup * A.tpf, group = lgroup + A.grp0;
	const int f_lo = group == 0 ? 0 : (int)(group * 4 - A.phase);
	int f_hi = (int)(group * 4 - A.phase) + 4;
	if (f_hi > (int)A.n_frames) f_hi = (int)A.n_frames;
	const int nf = f_hi - f_lo;                                // 1..4 frames

	const uint32_t blk = tile * DEC_T + tid;
	const bool valid = blk < A.nblk;
	const uint32_t b = valid ? blk : A.nblk - 1;
	const uint32_t by = b / A.bw, bx = b - by * A.bw;
	const uint32_t poff = by * 4 * A.w + bx * 4;
	const bool has_last = (tile == A.tpf - 1);                 // this workgroup holds block nblk-1
	const bool is_last = valid && blk == A.nblk - 1;

	uint32_t off[4], ne[4], bp[4];
	uint32_t r_lo[4], r_len[4];
	uint32_t cur[16], icol[16];
	bool stale, istale;
	// stale / istale: the block's img_data / iframe->img_data still derive from the state before this GOP.  For the first
	// GOP of the batch that state is the caller's (prev / prev_iframe) and the pixels are right as they are; what the
	// flags then tell is whether the batch DEPENDS on the state handed in (reported through agmv_hip_decode_prior_dependent).
	auto load_prior = [&]() {
		if (group == 0) {                                      // state of the decoder before the batch
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
		} else {
#pragma unroll
			for (int k = 0; k < 16; k++) { cur[k] = 0; icol[k] = 0; }
		}
		stale = true; istale = true;
	};
	uint32_t st[4][DEC_SR];                                    // the byte windows on their way from global memory to LDS
	// The first DEC_T dwords of every window go out as unconditional buffer loads (a lane beyond the window is out of range: 0,
	// no fetch, NO BRANCH), LAST; the rest -- needed only where a tile's blocks average more than 4 bytes -- under uniform
	// branches ahead of them.  The compiler waits for an earlier load with the count of loads that follow it on EVERY path,
	// i.e. those four: the bitmap words are then awaited with the windows still in flight.  (With all of them under branches it
	// has to assume none was issued, and the first wait drains everything.)
	auto load_windows = [&]() {
		__amdgpu_buffer_rsrc_t rs[4];
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const uint8_t* fb = A.bits + (size_t)(i < nf ? f_lo + i : f_lo) * A.stride + r_lo[i];
			uint32_t nrec = r_lo[i] < (uint32_t)A.stride ? (uint32_t)A.stride - r_lo[i] : 0u;
			if (nrec > r_len[i]) nrec = r_len[i];
			rs[i] = uniform_rsrc(fb, nrec & ~3u);
		}
#pragma unroll
		for (int i = 0; i < 4; i++) {
#pragma unroll
			for (int k = 1; k < DEC_SR; k++) {
				st[i][k] = 0;
				if ((uint32_t)(k * DEC_T) * 4u < r_len[i])         // (uniform)
					st[i][k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs[i], (uint32_t)(k * DEC_T + tid) * 4u, 0, 0);
			}
		}
#pragma unroll
		for (int i = 0; i < 4; i++) st[i][0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs[i], (uint32_t)tid * 4u, 0, 0);
	};
	auto store_windows = [&]() {
#pragma unroll
		for (int i = 0; i < 4; i++) {
#pragma unroll
			for (int k = 0; k < DEC_SR; k++) {
				const uint32_t j = (uint32_t)(k * DEC_T + tid) * 4u;
				if (j < r_len[i]) *(uint32_t*)(s_bytes[i] + j) = st[i][k];
			}
		}
	};
	if (!BM) {
		// ---- round trip 1: entry offsets, nentered, bpos of every frame of the GOP; previous state of the block
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int fi = i < nf ? f_lo + i : f_hi - 1;
			off[i] = A.offsets[(size_t)fi * A.nblk + b];            // (non-temporal here and in k_fp_expand's stores: 0.375 -> 0.39-0.41 ms, not kept)
			ne[i] = A.nentered[fi]; bp[i] = A.bpos[fi];
		}
		// the range the tile's entered blocks can touch in frame i: [first entry, last entry + 33 + 8]; entry offsets
		// increase with the block index, so it is [offset of lane 0, offset of the last entered lane]
#pragma unroll
		for (int i = 0; i < 4; i++) {
			if (i < nf && valid && blk < ne[i]) {
				if (tid == 0) s_rng[i][0] = off[i];
				if (blk + 1 == ne[i] || tid == DEC_T - 1 || blk + 1 == A.nblk) s_rng[i][1] = off[i];   // exactly one lane
			}
		}
		__syncthreads();                                       // ranges (and the palette) visible
		// ---- round trip 2: the four byte windows, all in flight together
#pragma unroll
		for (int i = 0; i < 4; i++) {
			r_lo[i] = 0; r_len[i] = 0;
			if (i < nf && tile * DEC_T < ne[i]) {              // uniform: at least the first block of the tile is entered
				r_lo[i] = s_rng[i][0] & ~3u;
				uint32_t len = s_rng[i][1] + 48u - r_lo[i];
				if (len > (uint32_t)DEC_STAGE) len = DEC_STAGE;
				r_len[i] = len & ~3u;
			}
		}
	} else {
		// ---- bitmap form.  Round trip 1 (uniform, scalar loads): where this tile and the next one are entered in each frame.
		// Round trip 2, all in flight together: the entry bitmap words between the two (one per lane, 64 bytes of stream
		// each) and the byte windows.  The lanes then rank their blocks in the bitmap: block j of the tile is entered at
		// the j-th set bit behind the tile's entry -- an exclusive scan of the words' popcounts through LDS, a binary search
		// of the lane's rank in it, a bit select.
		uint32_t t0[4], t1[4], P0[4];
		bool wide[4];
		{
			// one vector load for the sixteen words (lane = kind * 4 + frame), broadcast by v_readlane: as scalar loads they
			// are sixteen scalar-cache misses, and every tile's are different
			const int q = tid & 3, kind = (tid >> 2) & 3;
			const int fq = q < nf ? f_lo + q : f_hi - 1;
			const uint32_t* hp = kind == 0 ? A.nentered + fq : (kind == 1 ? A.bpos + fq : A.tidx + (size_t)fq * (A.tpf + 1) + tile + (kind == 3 ? 1 : 0));
			const uint32_t hv = *hp;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				ne[i] = (uint32_t)__builtin_amdgcn_readlane((int)hv, i); bp[i] = (uint32_t)__builtin_amdgcn_readlane((int)hv, 4 + i);
				t0[i] = (uint32_t)__builtin_amdgcn_readlane((int)hv, 8 + i); t1[i] = (uint32_t)__builtin_amdgcn_readlane((int)hv, 12 + i);
			}
		}
		unsigned long long vw[4];
		bool usew[4];
		const unsigned long long* vp[4];
		uint32_t npmax = 1;                                    // words of the longest range (uniform)
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int fi = i < nf ? f_lo + i : f_hi - 1;
			const bool has = i < nf && tile * DEC_T < ne[i] && t0[i] != TIDX_NONE;   // uniform
			r_lo[i] = 0; r_len[i] = 0; P0[i] = 0; wide[i] = false; off[i] = 0; usew[i] = false;
			uint32_t pc = 0;
			if (has) {
				P0[i] = t0[i] >> 6;
				const uint32_t last = (t1[i] != TIDX_NONE ? t1[i] - 1u : bp[i]) >> 6;   // last word that can hold an entry of this tile
				wide[i] = last - P0[i] >= (uint32_t)DEC_T;     // (garbage between blocks: the resync can skip any number of bytes)
				if (!wide[i] && last - P0[i] + 1u > npmax) npmax = last - P0[i] + 1u;
				pc = P0[i] + (uint32_t)tid;
				usew[i] = !wide[i] && pc <= last;
				r_lo[i] = t0[i] & ~3u;
				uint32_t hi = (t1[i] != TIDX_NONE ? t1[i] : bp[i] + 1u) + 48u;
				uint32_t len = hi > r_lo[i] ? hi - r_lo[i] : 0u;
				if (len > (uint32_t)DEC_STAGE) len = DEC_STAGE;
				r_len[i] = len & ~3u;
			}
			vp[i] = A.vm + (size_t)fi * A.maxR * FOWN + (usew[i] ? pc : 0u);   // always a readable word: the four loads go out back to back, unconditionally
		}
#pragma unroll
		for (int i = 0; i < 4; i++) vw[i] = *vp[i];
		asm volatile("" ::: "memory");                      // the four bitmap loads stay AHEAD of the windows: the first wait then leaves the windows in flight
		load_windows();
#pragma unroll
		for (int i = 0; i < 4; i++) {
			vw[i] = usew[i] ? vw[i] : 0ull;
			if (tid == 0) vw[i] &= ~0ull << (t0[i] & 63u);
		}
		// scan + search, with the LDS of the byte windows (not yet written) as scratch: [frame][lane] prefix (u16) | word (u64)
		uint16_t* s_pre = (uint16_t*)&s_bytes[0][0];               // 4 * DEC_T * 2 bytes
		unsigned long long* s_w = (unsigned long long*)(&s_bytes[0][0] + 4 * DEC_T * 2);
		uint32_t* s_wt = s_nb;                                 // [frame][wave] entries per wave
		static_assert(4 * DEC_T * 10 <= 4 * DEC_STAGE && DEC_T / 64 * 4 <= DEC_T, "scratch fits the stage");
		uint32_t incl[4], cnt[4];
		const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
		for (int i = 0; i < 4; i++) {
			cnt[i] = (uint32_t)__popcll(vw[i]);
			incl[i] = wave_incl_scan(cnt[i], lane);
			if (lane == 63) s_wt[i * (DEC_T / 64) + wave] = incl[i];
		}
		lds_barrier();                                         // wave totals visible (an LDS-only barrier: the byte windows stay in flight under the scan and the search)
#pragma unroll
		for (int i = 0; i < 4; i++) {
			uint32_t base = 0;
#pragma unroll
			for (int wv = 0; wv < DEC_T / 64; wv++) base += wv < wave ? s_wt[i * (DEC_T / 64) + wv] : 0u;
			s_pre[i * DEC_T + tid] = (uint16_t)(base + incl[i] - cnt[i]);
			s_w[i * DEC_T + tid] = vw[i];
		}
		lds_barrier();
		{
			// the four frames' searches step together (four independent chains of LDS reads); a lane that is not entered
			// searches too and its result is not used
			uint32_t st0 = DEC_T / 2;
			while (st0 >= npmax && st0 > 1) st0 >>= 1;             // largest power of two below npmax (index 0 needs no test)
			uint32_t pz[4] = {0, 0, 0, 0};                         // the last word whose prefix is <= the lane's rank
			// (uniform) as many halvings as the longest of the four word ranges needs: a tile of the benchmark clip spans ~14 words
			for (uint32_t st = st0; st >= 1; st >>= 1) {
				uint32_t v[4];
#pragma unroll
				for (int i = 0; i < 4; i++) v[i] = s_pre[i * DEC_T + pz[i] + st];
#pragma unroll
				for (int i = 0; i < 4; i++) pz[i] += v[i] <= (uint32_t)tid ? st : 0u;
			}
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t pre = s_pre[i * DEC_T + pz[i]];
				const unsigned long long wv = s_w[i * DEC_T + pz[i]];
				off[i] = ((P0[i] + pz[i]) << 6) + select64(wv, (uint32_t)tid - pre);
			}
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int fi = i < nf ? f_lo + i : f_hi - 1;
				if (wide[i] && i < nf && valid && blk < ne[i]) off[i] = bm_offset_of(A.vm, A.kb, A.maxR, bp[i], (uint32_t)fi, blk);   // (uniform branch; exotic)
			}
		}
		lds_barrier();                                         // scratch read: the byte windows may land
	}
	if (!BM) { load_windows(); store_windows(); }
	else store_windows();
	load_prior();                                              // (only the first GOP of a batch reads anything here: kept out of the prologue, whose registers hold the byte windows)
	__syncthreads();                                           // the last wait on global loads in this kernel

	bool anystale = false, needfix = false;
#pragma unroll
	for (int i = 0; i < 4; i++) {
		if (i >= nf) break;
		const int f = f_lo + i;
		bool fill_written = false;
		const bool entered = valid && blk < ne[i];
		const uint32_t own3 = cur[3];                          // the block's pixel (3,0) and staleness before this frame
		const bool stale0 = stale;
		if (entered) {
			StagedSrc src{(const __attribute__((address_space(3))) uint8_t*)s_bytes[i], r_lo[i], r_len[i], A.bits + (size_t)f * A.stride, (uint32_t)A.stride};
			decode_block_staged<M512>(src, off[i], bp[i], s_pal, cur, icol, istale, stale, fill_written);
		}
		if (has_last) {                                        // img_data[(x-1)+(y+1)*w] of the block to the left
			s_nb[tid] = cur[7];
			s_nbstale[tid] = stale ? 1u : 0u;
			lds_barrier();
			if (is_last && fill_written) {
				if (A.bw == 1) {                                   // one block per row: the reference's 64-bit (x-1) wraps to the
#pragma unroll                                                     // block's own pixel (3,0), not yet written in this frame
					for (int k = 0; k < 16; k++) cur[k] = own3;
					stale = stale0;
				} else if (tid > 0) {
					uint32_t c = s_nb[tid - 1];
#pragma unroll
					for (int k = 0; k < 16; k++) cur[k] = c;
					stale = s_nbstale[tid - 1] != 0;
				} else {
					stale = true; needfix = true;             // neighbour lives in another tile: fix-up (the pixels here are NOT final)
				}
			}
			lds_barrier();
		}
		if (((A.first_fc + f) & 3u) == 0) {                    // I-frame snapshot, :401-405
#pragma unroll
			for (int k = 0; k < 16; k++) icol[k] = cur[k];
			istale = stale;
		}
		anystale |= stale;
		if (valid) store_block(A.out + (size_t)f * npx, poff, A.w, cur);
	}
	if (valid && (anystale || needfix)) {
		if (group != 0 || needfix) {                           // stale in any frame of a later GOP: k_fixup replays the block
			atomicOr(A.dirty + (blk >> 5), 1u << (blk & 31u));
			A.dirty[(A.nblk + 31) >> 5] = 1u;                   // "anything to repair" word behind the bitmap
		}
		if (group == 0 && anystale) A.dirty[((A.nblk + 31) >> 5) + 1] = 1u;   // the batch depends on the decoder state before it
	}
}

// K4: repair of the blocks k_decode flagged.  The blocks are independent of each other (a block's pixels in frame f derive
// from the SAME block in earlier frames) with one exception, the last block of the frame, whose FILL takes a pixel of its
// left neighbour (src/agmv_decode.c:264-266).  So the grid is one wave per 64 consecutive block positions; a wave whose
// 64 bitmap bits are clear exits at once, the others replay ALL frames in order for their flagged positions from the true
// pre-batch state and overwrite the output.  The wave that holds block nblk-1 also replays block nblk-2 (flagged or not:
// a replay from the true state writes the true pixels), in the lane below when both sit in one wave, else in lane 1.
template <bool M512, bool BM>
__global__ __launch_bounds__(64) void k_fixup(DecArgs A)
{
	__shared__ uint32_t s_pal[512];
	const int lane = threadIdx.x;
	const uint32_t npx = A.w * A.h;
	const uint32_t nwords = (A.nblk + 31) >> 5;
	if (A.dirty[nwords] == 0) return;                          // nothing depends on an earlier GOP: done
	const uint32_t base = blockIdx.x * 64u;
	const uint32_t w0 = A.dirty[base >> 5], w1 = (base >> 5) + 1 < nwords ? A.dirty[(base >> 5) + 1] : 0u;
	if ((w0 | w1) == 0) return;
	uint32_t blk = base + (uint32_t)lane;
	bool active = blk < A.nblk && (((lane < 32 ? w0 : w1) >> (lane & 31)) & 1u);
	// the last block's left neighbour rides along
	const uint32_t last = A.nblk - 1;
	const bool have_last = last >= base && last < base + 64 && ((((last - base) < 32 ? w0 : w1) >> ((last - base) & 31)) & 1u);
	int nb_lane = -1;                                          // lane that holds block nblk-2 when this wave repairs nblk-1
	if (have_last && A.nblk >= 2) {
		if (last > base) { nb_lane = (int)(last - base) - 1; if (lane == nb_lane) active = true; }
		else { nb_lane = 1; if (lane == 1) { blk = last - 1; active = true; } }       // nblk-1 is lane 0: lane 1 (a block beyond the frame) takes nblk-2
	}
	for (int i = lane; i < 512; i += 64) s_pal[i] = A.pal[i];
	__syncthreads();
	if (!active) blk = 0;
	const uint32_t by = blk / A.bw, bx = blk - by * A.bw;
	const uint32_t poff = by * 4 * A.w + bx * 4;
	const bool is_last = active && blk == last;
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
		const uint32_t own3 = cur[3];
		if (active && blk < A.nentered[f]) {
			ByteSrc src{A.bits + (size_t)f * A.stride, (uint32_t)A.stride};
			const uint32_t o = BM ? bm_offset_of(A.vm, A.kb, A.maxR, A.bpos[f], f, blk) : A.offsets[(size_t)f * A.nblk + blk];
			decode_block<M512>(src, o, A.bpos[f], s_pal, cur, icol, false, stale, fill_written);
		}
		const uint32_t left = (uint32_t)__builtin_amdgcn_readlane((int)cur[7], nb_lane < 0 ? 0 : nb_lane);   // img_data[(x-1)+(y+1)*w] of the left neighbour
		if (is_last && fill_written) {
			const uint32_t c = A.bw == 1 ? own3 : left;            // one block per row: see k_decode
#pragma unroll
			for (int k = 0; k < 16; k++) cur[k] = c;
		}
		if (((A.first_fc + f) & 3u) == 0) {
#pragma unroll
			for (int k = 0; k < 16; k++) icol[k] = cur[k];
		}
		if (active) store_block(A.out + (size_t)f * npx, poff, A.w, cur);
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
	// lanes hold consecutive pixels; neighbours mostly fall into the same bin, so each RUN of equal codes inside the
	// wave adds its length with one atomic (flat or static areas: one atomic per 64 pixels)
	const int lane = threadIdx.x & 63;
	const size_t stride = (size_t)gridDim.x * 256;
	for (size_t i0 = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63); i0 < n; i0 += stride) {
		const size_t i = i0 + lane;
		const bool live = i < n;
		const uint32_t c = live ? quantize_color(pix[i], quality) : 0xFFFFFFFFu;
		const uint32_t prev = __shfl_up(c, 1, 64);
		const bool leader = live && (lane == 0 || c != prev);
		const unsigned long long lead = __ballot(leader), alive = __ballot(live);
		if (leader) {
			const unsigned long long rest = lane == 63 ? 0ull : lead >> (lane + 1);
			const uint32_t end = rest ? (uint32_t)lane + 1u + (uint32_t)__builtin_ctzll(rest) : (uint32_t)__popcll(alive);
			atomicAdd(hist + c, end - (uint32_t)lane);
		}
	}
}

// A spin of k_encode that ran into its bound leaves ctrl[1] != 0 and the kernel carries on with a wrong offset: the bytes of
// the batch are not to be used.  So that a caller who skips agmv_hip_check cannot take them for good ones, every size of
// the batch is then overwritten with 0xFFFFFFFF (no frame is that long: agmv_hip_max_usize < 2^32).
__global__ __launch_bounds__(64) void k_encode_verdict(const uint32_t* __restrict__ ctrl, uint32_t* __restrict__ sizes, uint32_t n_frames)
{
	if (ctrl[1] == 0) return;
	for (uint32_t f = threadIdx.x; f < n_frames; f += 64) sizes[f] = 0xFFFFFFFFu;
}

// E2 / E3 without the table: nearest colour / entry of n pixels by the reference's own search (src/agmv_utils.c:785-895).
// For the exported single-colour functions AGMV_FindNearestColor / AGMV_FindNearestEntry, whose palette argument changes from
// call to call: building a 2^24-entry table for one look-up would cost 2 ms.
__global__ __launch_bounds__(64) void k_nearest_direct(const uint32_t* __restrict__ pal, int mode512, const uint32_t* __restrict__ pix,
                                                       size_t n, uint16_t* __restrict__ out)
{
	const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
	if (i >= n) return;
	const uint32_t c = pix[i];
	const int r = (int)((c >> 16) & 0xff), g = (int)((c >> 8) & 0xff), b = (int)(c & 0xff);
	uint32_t d0, i0;
	nearest_in(pal, r, g, b, d0, i0);
	uint32_t e = i0;
	if (mode512) {
		uint32_t d1, i1;
		nearest_in(pal + 256, r, g, b, d1, i1);
		if (!(d0 <= d1)) e = 0x100u | i1;
	}
	out[i] = (uint16_t)e;
}

// E5 / E6 on 16 colour pairs: how many pairs are within +-2 on R, G and B (the predicate of AGMV_CompareIFrameBlock /
// AGMV_ComparePFrameBlock, src/agmv_encode.c:293, :345), for the exported single-block helpers.
__global__ __launch_bounds__(64) void k_within2_count(const uint32_t* __restrict__ ab, uint32_t* __restrict__ out)
{
	const int lane = threadIdx.x;
	const bool in = lane < 16 && within2(ab[lane], ab[16 + lane]);
	const unsigned long long m = __ballot(in);
	if (lane == 0) out[0] = (uint32_t)__popcll(m);
}

// ----------------------------------------------------------------------------------------------
// C-ABI
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// Palette tables (colour -> entry table, +-2 bit matrix, palette) are read-only once built and identical for every
// context that holds the same palette on the same device -- the sequence drivers open two worker contexts per device
// besides the caller's.  They are shared: one 512 MiB allocation and one 2 ms table build per (device, palette).
// ----------------------------------------------------------------------------------------------
struct lut_share {
	int device, mode512, refs;
	uint32_t pal[512];
	uint16_t* d_lut;
	uint32_t* d_mtx;
	uint32_t* d_pal;
	hipEvent_t built;               // recorded behind the build kernels; a context on another stream waits for it
	lut_share* next;
};
static pthread_mutex_t g_lut_mu = PTHREAD_MUTEX_INITIALIZER;
static lut_share* g_luts = nullptr;

static void lut_release(lut_share* sh)                        // (device of the share is current)
{
	if (!sh) return;
	pthread_mutex_lock(&g_lut_mu);
	if (--sh->refs == 0) {
		for (lut_share** pp = &g_luts; *pp; pp = &(*pp)->next)
			if (*pp == sh) { *pp = sh->next; break; }
		(void)hipFree(sh->d_lut); (void)hipFree(sh->d_mtx); (void)hipFree(sh->d_pal);
		if (sh->built) (void)hipEventDestroy(sh->built);
		free(sh);
	}
	pthread_mutex_unlock(&g_lut_mu);
}

extern "C" int agmv_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" void agmv_hip_destroy(agmv_hip_ctx* c);
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
	if (!c) { snprintf(g_err, sizeof(g_err), "agmv_hip: out of host memory"); return nullptr; }
	c->device = device;
	hipDeviceProp_t prop;
	hipError_t e2 = hipMalloc(&c->d_ctrl, CTRL_BYTES);
	if (e2 == hipSuccess) e2 = hipGetDeviceProperties(&prop, device);
	if (e2 != hipSuccess) {                                    // nothing half-built is left behind
		fail("agmv_hip_create", e2, __LINE__);
		agmv_hip_destroy(c);
		return nullptr;
	}
	c->n_cu = prop.multiProcessorCount;
	// persistent grid of k_encode = the workgroups that are resident at once (2 per CU: 69 KB of LDS each)
	{
		int per_cu = 0;
		const size_t lds = (size_t)512 * MROW * 4 + ENC_LDS_EXTRA;
		(void)hipFuncSetAttribute((const void*)k_encode<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_encode<true, false>, ENC_T, lds) != hipSuccess || per_cu < 1) per_cu = 2;
		if (getenv("AGMV_HIP_DEBUG")) fprintf(stderr, "agmv_hip: k_encode: %d workgroup(s) of %d lanes resident per CU (%zu B of LDS each), %d CUs\n", per_cu, ENC_T, lds, c->n_cu);
		c->enc_grid = prop.multiProcessorCount * per_cu;
	}
	return c;
}

extern "C" void agmv_hip_destroy(agmv_hip_ctx* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	lut_release(c->share);
	(void)hipFree(c->d_ctrl);
	(void)hipFree(c->d_status); (void)hipFree(c->d_dirty); (void)hipFree(c->d_parse_ws); (void)hipFree(c->d_fp_ws); (void)hipFree(c->d_ient_tmp);
	(void)hipFree(c->d_nent_own); (void)hipFree(c->d_nn_pal); (void)hipFree(c->d_nn_pix); (void)hipFree(c->d_nn_ent);
	if (c->ev_enc) (void)hipEventDestroy(c->ev_enc);
	for (int i = 0; i < 8; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
	if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	for (int i = 0; i < DEC_MAX_SLICES; i++) if (c->ev_slice[i]) (void)hipEventDestroy(c->ev_slice[i]);
	if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
	free(c);
}

static void ev_mark(agmv_hip_ctx* c, int which, hipStream_t s)
{
	if (c->timing) (void)hipEventRecord(c->ev[which], s);
}

extern "C" int agmv_hip_enable_timing(agmv_hip_ctx* c, int on)
{
	if (!c) return -1;
	CK(hipSetDevice(c->device));
	if (on && !c->ev[0]) for (int i = 0; i < 8; i++) CK(hipEventCreate(&c->ev[i]));
	c->timing = on ? 1 : 0;
	return 0;
}

/* duration in ms of the last launch of kernel group `which` (0 = k_encode, 1 = the parser kernels,
   2 = k_decode + k_fixup, 3 = the whole of agmv_hip_parse_decode_frames_dev), measured with HIP events on the stream it ran on; synchronises on the stop event */
extern "C" float agmv_hip_last_kernel_ms(agmv_hip_ctx* c, int which)
{
	float ms = -1.0f;
	if (!c || !c->timing || which < 0 || which > 3) return -1.0f;
	if (hipEventSynchronize(c->ev[2 * which + 1]) != hipSuccess) return -1.0f;
	if (hipEventElapsedTime(&ms, c->ev[2 * which], c->ev[2 * which + 1]) != hipSuccess) return -1.0f;
	return ms;
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
	mode512 = mode512 ? 1 : 0;
	pthread_mutex_lock(&g_lut_mu);
	lut_share* sh = nullptr;
	for (lut_share* q = g_luts; q; q = q->next)
		if (q->device == c->device && q->mode512 == mode512 && memcmp(q->pal, pal, sizeof(pal)) == 0) { sh = q; break; }
	int rc = 0;
	if (sh) sh->refs++;
	else {
		sh = (lut_share*)calloc(1, sizeof(*sh));
		hipError_t e = sh ? hipSuccess : hipErrorOutOfMemory;
		if (e == hipSuccess) e = hipMalloc(&sh->d_lut, (size_t)LUT_ENTRIES * sizeof(uint16_t));
		if (e == hipSuccess) e = hipMalloc(&sh->d_mtx, 512 * MROW * sizeof(uint32_t));
		if (e == hipSuccess) e = hipMalloc(&sh->d_pal, 512 * sizeof(uint32_t));
		if (e == hipSuccess) e = hipEventCreateWithFlags(&sh->built, hipEventDisableTiming);
		if (e == hipSuccess) e = hipMemcpyAsync(sh->d_pal, pal, sizeof(pal), hipMemcpyHostToDevice, s);
		if (e == hipSuccess) e = hipStreamSynchronize(s);       // pal[] is a stack buffer
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_lut_build, dim3(LUT_COLOURS / 256), dim3(256), 0, s, sh->d_pal, mode512, sh->d_lut);
			e = hipGetLastError();
		}
		if (e == hipSuccess) {
			hipLaunchKernelGGL(k_mtx_build, dim3((512 * MROW + 255) / 256), dim3(256), 0, s, sh->d_pal, sh->d_mtx);
			e = hipGetLastError();
		}
		if (e == hipSuccess) e = hipEventRecord(sh->built, s);
		if (e != hipSuccess) {
			rc = fail("agmv_hip_set_palette", e, __LINE__);
			if (sh) { (void)hipFree(sh->d_lut); (void)hipFree(sh->d_mtx); (void)hipFree(sh->d_pal); if (sh->built) (void)hipEventDestroy(sh->built); free(sh); }
			sh = nullptr;
		} else {
			sh->device = c->device; sh->mode512 = mode512; sh->refs = 1;
			memcpy(sh->pal, pal, sizeof(pal));
			sh->next = g_luts; g_luts = sh;
		}
	}
	pthread_mutex_unlock(&g_lut_mu);
	if (rc) return rc;
	if (hipStreamWaitEvent(s, sh->built, 0) != hipSuccess) { lut_release(sh); return fail("hipStreamWaitEvent", hipErrorUnknown, __LINE__); }   // built on another context's stream?
	lut_share* old_sh = c->share;
	c->share = sh; c->d_lut = sh->d_lut; c->d_mtx = sh->d_mtx; c->d_pal = sh->d_pal;
	c->mode512 = mode512;
	c->have_palette = 1;
	if (old_sh) {                                             // work of this context still in flight may read the old tables
		CK(hipDeviceSynchronize());
		lut_release(old_sh);
	}
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

static int encode_dev(agmv_hip_ctx* c, const uint32_t* d_pix, uint32_t n_frames, uint32_t w, uint32_t h,
                      uint32_t first_fc, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                      uint16_t* d_ientries, void* stream, bool entries)
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
	A.pix = d_pix; A.out = d_out; A.sizes = d_sizes; A.lut = c->d_lut; A.mtx = c->d_mtx; A.ientries_in = d_ientries; A.ientries_out = d_ientries;
	A.out_stride = out_stride;
	A.n_frames = n_frames; A.w = w; A.h = h; A.bw = w / 4; A.nblk = (w / 4) * (h / 4);
#ifdef AGMV_LAB_ENCODE_W
	const char* ek = getenv("AGMV_ENC_KERNEL");                // lab build only: the independent-wave form (lab/k_encode_w.inc)
	const bool wform = ek && strcmp(ek, "w") == 0;
#else
	constexpr bool wform = false;
#endif
	A.tpf = wform ? (A.nblk + WBLK - 1) / WBLK : (A.nblk + ENC_T - 1) / ENC_T;   // tiles of ENC_T blocks per frame
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
	// A batch that starts inside a GOP READS the caller's entry plane (its first tiles) and, if it also holds an I-frame,
	// WRITES the plane (other tiles of the same positions, running at the same time): the new entries go to a scratch
	// plane and are copied over the caller's when the kernel is done.
	const bool ient_both = d_ientries && A.phase != 0 && A.last_iframe != 0xffffffffu;
	const size_t npx_e = (size_t)w * h;
	if (ient_both) {
		if (npx_e > c->ient_cap) {
			if (c->d_ient_tmp) CK(hipFree(c->d_ient_tmp));
			c->d_ient_tmp = nullptr; c->ient_cap = 0;
			CK(hipMalloc(&c->d_ient_tmp, npx_e * sizeof(uint16_t)));
			c->ient_cap = npx_e;
		}
		A.ientries_out = c->d_ient_tmp;
	}
	// the look-back status and the control words belong to the context: an encode on another stream waits for the previous one
	if (!c->ev_enc) CK(hipEventCreateWithFlags(&c->ev_enc, hipEventDisableTiming));
	if (c->have_enc && c->enc_stream != s) CK(hipStreamWaitEvent(s, c->ev_enc, 0));
	CK(hipMemsetAsync(c->d_status, 0, need * sizeof(unsigned long long), s));
	CK(hipMemsetAsync(c->d_ctrl, 0, CTRL_BYTES, s));
	uint32_t grid = (uint32_t)c->enc_grid;
	const uint32_t work = wform ? (A.total_tiles + ENC_WAVES - 1) / ENC_WAVES : A.total_tiles;
	if (grid > work) grid = work;
	size_t lds = (size_t)(c->mode512 ? 512 : 256) * MROW * 4 + ENC_LDS_EXTRA;
	void (*kern)(EncArgs) = c->mode512 ? (entries ? k_encode<true, true> : k_encode<true, false>)
	                                   : (entries ? k_encode<false, true> : k_encode<false, false>);
#ifdef AGMV_LAB_ENCODE_W
	if (wform) {
		lds = (size_t)(c->mode512 ? 512 : 256) * MROW * 4 + ENCW_LDS_EXTRA;
		kern = c->mode512 ? (entries ? k_encode_w<true, true> : k_encode_w<true, false>)
		                  : (entries ? k_encode_w<false, true> : k_encode_w<false, false>);
	}
#endif
	CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	ev_mark(c, 0, s);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(ENC_T), lds, s, A);
	ev_mark(c, 1, s);
	CK(hipGetLastError());
	hipLaunchKernelGGL(k_encode_verdict, dim3(1), dim3(64), 0, s, c->d_ctrl, d_sizes, n_frames);
	CK(hipGetLastError());
	if (ient_both) CK(hipMemcpyAsync(d_ientries, c->d_ient_tmp, npx_e * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
	CK(hipEventRecord(c->ev_enc, s));
	c->enc_stream = s; c->have_enc = 1;
	return 0;
}

extern "C" int agmv_hip_encode_frames_dev(agmv_hip_ctx* c, const uint32_t* d_pix, uint32_t n_frames, uint32_t w, uint32_t h,
                                          uint32_t first_fc, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                                          uint16_t* d_ientries, void* stream)
{
	return encode_dev(c, d_pix, n_frames, w, h, first_fc, d_out, out_stride, d_sizes, d_ientries, stream, false);
}

extern "C" int agmv_hip_encode_entries_dev(agmv_hip_ctx* c, const uint32_t* d_entries, uint32_t n_frames, uint32_t w, uint32_t h,
                                           uint32_t first_fc, uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                                           uint16_t* d_ientries, void* stream)
{
	return encode_dev(c, d_entries, n_frames, w, h, first_fc, d_out, out_stride, d_sizes, d_ientries, stream, true);
}

extern "C" int agmv_hip_check(agmv_hip_ctx* c, void* stream)
{
	if (need_ctx(c, false)) return -1;
	uint32_t ctrl[4] = {0, 0, 0, 0};
	CK(hipStreamSynchronize((hipStream_t)stream));
	CK(hipMemcpy(ctrl, c->d_ctrl, 16, hipMemcpyDeviceToHost));
#ifdef ENC_PROF
	{
		uint32_t pr[12];
		CK(hipMemcpy(pr, c->d_ctrl + 32, 44, hipMemcpyDeviceToHost));
		double tot = 0;
		for (int k = 0; k < 10; k++) tot += pr[k];
		const double per = 64.0 / (pr[10] ? pr[10] : 1);        // shader cycles per item of ONE wave (the counters hold cycles / 64 summed over the waves)
		fprintf(stderr, "k_encode phases, cycles per wave-item (%% of wave time): loop top %.0f (%.1f) | pixel wait %.0f (%.1f) | look-up issue %.0f (%.1f) | prefetch issue + ticket %.0f (%.1f) | "
		        "look-up wait + park %.0f (%.1f) | duty %.0f (%.1f) | transpose + classify + scan %.0f (%.1f) | emit %.0f (%.1f) | wait for gbase %.0f (%.1f) | copy-out %.0f (%.1f) | total %.0f, %u wave-items\n",
		        pr[9] * per, 100 * pr[9] / tot, pr[8] * per, 100 * pr[8] / tot, pr[0] * per, 100 * pr[0] / tot, pr[2] * per, 100 * pr[2] / tot, pr[1] * per, 100 * pr[1] / tot,
		        pr[3] * per, 100 * pr[3] / tot, pr[4] * per, 100 * pr[4] / tot, pr[5] * per, 100 * pr[5] / tot, pr[6] * per, 100 * pr[6] / tot, pr[7] * per, 100 * pr[7] / tot, tot * per, pr[10]);
	}
#endif
	if (ctrl[1]) { snprintf(g_err, sizeof(g_err), "agmv_hip: look-back timed out inside k_encode (device error word %u)", ctrl[1]); return -2; }
	return 0;
}

static int encode_host(agmv_hip_ctx* c, const uint32_t* h_pix, uint32_t n_frames, uint32_t w, uint32_t h,
                       uint32_t first_fc, uint8_t* h_out, size_t out_stride, uint32_t* h_sizes, uint16_t* h_ient, bool entries)
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
		if (encode_dev(c, d_pix, n_frames, w, h, first_fc, d_out, out_stride, d_sizes, d_ient, nullptr, entries)) break;
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

extern "C" int agmv_hip_encode_frames(agmv_hip_ctx* c, const uint32_t* h_pix, uint32_t n_frames, uint32_t w, uint32_t h,
                                      uint32_t first_fc, uint8_t* h_out, size_t out_stride, uint32_t* h_sizes, uint16_t* h_ient)
{
	return encode_host(c, h_pix, n_frames, w, h, first_fc, h_out, out_stride, h_sizes, h_ient, false);
}

extern "C" int agmv_hip_encode_entries(agmv_hip_ctx* c, const uint32_t* h_entries, uint32_t n_frames, uint32_t w, uint32_t h,
                                       uint32_t first_fc, uint8_t* h_out, size_t out_stride, uint32_t* h_sizes, uint16_t* h_ient)
{
	return encode_host(c, h_entries, n_frames, w, h, first_fc, h_out, out_stride, h_sizes, h_ient, true);
}

extern "C" int agmv_hip_within2_count(agmv_hip_ctx* c, const uint32_t a[16], const uint32_t b[16])
{
	if (need_ctx(c, false)) return -1;
	if (!c->d_nn_pal) CK(hipMalloc(&c->d_nn_pal, 512 * sizeof(uint32_t)));
	uint32_t ab[32], n = 0;
	memcpy(ab, a, 64); memcpy(ab + 16, b, 64);
	CK(hipMemcpy(c->d_nn_pal, ab, sizeof(ab), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_within2_count, dim3(1), dim3(64), 0, nullptr, c->d_nn_pal, c->d_nn_pal + 32);
	CK(hipGetLastError());
	CK(hipMemcpy(&n, c->d_nn_pal + 32, 4, hipMemcpyDeviceToHost));
	return (int)n;
}

extern "C" int agmv_hip_nearest(agmv_hip_ctx* c, const uint32_t p0[256], const uint32_t p1[256], int mode512,
                                const uint32_t* h_pix, size_t n, uint16_t* h_entries)
{
	if (need_ctx(c, false)) return -1;
	if (n == 0) return 0;
	if (!c->d_nn_pal) CK(hipMalloc(&c->d_nn_pal, 512 * sizeof(uint32_t)));
	if (n > c->nn_cap) {
		(void)hipFree(c->d_nn_pix); (void)hipFree(c->d_nn_ent);
		c->d_nn_pix = nullptr; c->d_nn_ent = nullptr; c->nn_cap = 0;
		const size_t cap = n < 256 ? 256 : n;
		CK(hipMalloc(&c->d_nn_pix, cap * sizeof(uint32_t)));
		CK(hipMalloc(&c->d_nn_ent, cap * sizeof(uint16_t)));
		c->nn_cap = cap;
	}
	uint32_t pal[512];
	memcpy(pal, p0, 1024);
	if (mode512 && p1) memcpy(pal + 256, p1, 1024); else memset(pal + 256, 0, 1024);
	CK(hipMemcpy(c->d_nn_pal, pal, sizeof(pal), hipMemcpyHostToDevice));
	CK(hipMemcpy(c->d_nn_pix, h_pix, n * sizeof(uint32_t), hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_nearest_direct, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, c->d_nn_pal, mode512 ? 1 : 0, c->d_nn_pix, n, c->d_nn_ent);
	CK(hipGetLastError());
	CK(hipMemcpy(h_entries, c->d_nn_ent, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
	return 0;
}

static int check_slab(const uint8_t* d_bits, size_t stride)
{
	if ((stride & 3u) || stride < 4 || ((uintptr_t)d_bits & 3u)) { snprintf(g_err, sizeof(g_err), "agmv_hip: bitstream slab and stride must be 4-byte aligned"); return -1; }
	return 0;
}

// the robust parser kernels over n_frames frames on stream s (workspace of the context: launches that share it must be
// ordered); fstate != NULL: only the frames marked FS_BAD
static int parse_launch_robust(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos, uint32_t n_frames,
                               uint32_t nblk, uint32_t* d_offsets, uint32_t* d_nentered, size_t ws_frames, const uint32_t* fstate, hipStream_t s,
                               unsigned long long* vm = nullptr, uint32_t maxR = 0, const uint32_t* nbad = nullptr)
{
	const size_t cpf = (stride + PC) / PC, maxchunks = cpf * ws_frames;
	if (maxchunks >> 32) { snprintf(g_err, sizeof(g_err), "agmv_hip: parser batch too large (%zu chunk rows)", maxchunks); return -1; }
	const size_t need = maxchunks + (maxchunks * 33 + 1) / 2 + 16;   // dwords: centry | summ (u16)
	if (need > c->parse_ws_cap) {
		if (c->d_parse_ws) CK(hipFree(c->d_parse_ws));
		c->d_parse_ws = nullptr; c->parse_ws_cap = 0;
		CK(hipMalloc(&c->d_parse_ws, need * 4));
		c->parse_ws_cap = need;
	}
	ParseArgs A;
	memset(&A, 0, sizeof(A));
	A.bits = d_bits; A.stride = stride; A.bpos = d_bpos; A.offsets = d_offsets; A.nentered = d_nentered;
	A.cpf = (uint32_t)cpf; A.centry = c->d_parse_ws; A.summ = (uint16_t*)(A.centry + maxchunks);
	A.n_frames = n_frames; A.nblk = nblk; A.fstate = fstate; A.vm = vm; A.maxR = maxR; A.nbad = nbad;
	// the exception path (fstate): a few rows of workgroups stride over the frames and leave those that are not FS_BAD at
	// once -- with one row per frame the three gated launches cost 0.03 ms per 1024 frames for zero frames to parse
	const dim3 gy(1, fstate ? (n_frames < 64u ? n_frames : 64u) : (n_frames < 65535u ? n_frames : 65535u));
	// one wave per workgroup, each striding over the chunks of one frame: ~512 waves per CU in the grid (measured on
	// 1024 x 1080p: 8 / 16 / 32 / 64 / 128 / 256 per frame -> 2.85 / 2.30 / 1.96 / 1.87 / 1.83 / 1.84 ms)
	uint32_t gx = (uint32_t)(((size_t)c->n_cu * 512 + n_frames - 1) / n_frames);
	if (gx < 32) gx = 32;
	if (gx > 256) gx = 256;
	if (fstate) gx = 32;                                       // the exception path: most frames leave at once
	if (getenv("AGMV_PARSE_GX")) gx = (uint32_t)atoi(getenv("AGMV_PARSE_GX"));   // tuning aid
	if (gx > cpf) gx = (uint32_t)cpf;
	if (gx < 1) gx = 1;
	const dim3 grid(gx, gy.y);
	if (c->mode512) hipLaunchKernelGGL(k_parse_chunks<true>, grid, dim3(64), 0, s, A);
	else            hipLaunchKernelGGL(k_parse_chunks<false>, grid, dim3(64), 0, s, A);
	CK(hipGetLastError());
	hipLaunchKernelGGL(k_parse_stitch, dim3(fstate ? (n_frames < 1024u ? n_frames : 1024u) : n_frames), dim3(64), 0, s, A);
	CK(hipGetLastError());
	if (c->mode512) hipLaunchKernelGGL(k_parse_emit<true>, grid, dim3(64), 0, s, A);
	else            hipLaunchKernelGGL(k_parse_emit<false>, grid, dim3(64), 0, s, A);
	CK(hipGetLastError());
	return 0;
}

// the parser: speculative walks proven per frame (k_fp_*), the robust kernels for the frames that could not be proven.
// AGMV_HIP_PARSE=robust runs the robust kernels alone.
static int parse_launch(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos, uint32_t n_frames,
                        uint32_t nblk, uint32_t* d_offsets, uint32_t* d_nentered, size_t ws_frames, hipStream_t s, bool bitmap = false,
                        uint32_t* dirty = nullptr, uint32_t ndirty = 0)
{
	const char* mode = getenv("AGMV_HIP_PARSE");
	const bool robust_only = mode && strcmp(mode, "robust") == 0;
	c->d_fp_fstate = nullptr; c->fp_frames = 0;
	if (!bitmap && (n_frames > 65535u || robust_only))
		return parse_launch_robust(c, d_bits, stride, d_bpos, n_frames, nblk, d_offsets, d_nentered, ws_frames, nullptr, s);
	if (n_frames > 65535u) { snprintf(g_err, sizeof(g_err), "agmv_hip: more than 65535 frames in one parser launch"); return -1; }
	const size_t maxR = (stride + FRB - 1) / FRB + 1;
	const size_t nreg = maxR * ws_frames;
	const uint32_t tpfd = (nblk + DEC_T - 1) / DEC_T;
	const size_t b_rec = nreg * sizeof(uint4), b_vm = nreg * FOWN * 8, b_kb = nreg * 4, b_fs = ((ws_frames * 4 + 15) & ~(size_t)15);
	const size_t b_tx = bitmap ? (((size_t)ws_frames * (tpfd + 1) * 4 + 15) & ~(size_t)15) : 0;
	const size_t need = b_rec + b_vm + b_kb + b_fs + b_tx + 16;
	if (need > c->fp_ws_cap) {
		if (c->d_fp_ws) CK(hipFree(c->d_fp_ws));
		c->d_fp_ws = nullptr; c->fp_ws_cap = 0;
		CK(hipMalloc(&c->d_fp_ws, need));
		c->fp_ws_cap = need;
	}
	FpArgs A;
	memset(&A, 0, sizeof(A));
	A.bits = d_bits; A.stride = stride; A.bpos = d_bpos; A.offsets = d_offsets; A.nentered = d_nentered;
	uint8_t* w = (uint8_t*)c->d_fp_ws;
	A.rec = (uint4*)w; A.vm = (unsigned long long*)(w + b_rec); A.kb = (uint32_t*)(w + b_rec + b_vm); A.fstate = (uint32_t*)(w + b_rec + b_vm + b_kb);
	A.tidx = bitmap ? (uint32_t*)(w + b_rec + b_vm + b_kb + b_fs) : nullptr; A.tpfd = tpfd;
	A.nbad = (uint32_t*)(w + b_rec + b_vm + b_kb + b_fs + b_tx);
	A.n_frames = n_frames; A.nblk = nblk; A.maxR = (uint32_t)maxR; A.dirty = dirty; A.ndirty = ndirty;
	uint32_t gx = (uint32_t)(((size_t)c->n_cu * 512 + n_frames - 1) / n_frames);
	if (gx < 32) gx = 32;
	if (gx > 256) gx = 256;
	if (getenv("AGMV_PARSE_GX")) gx = (uint32_t)atoi(getenv("AGMV_PARSE_GX"));   // tuning aid
	if (gx > maxR) gx = (uint32_t)maxR;
	if (gx < 1) gx = 1;
	// many small frames (8192 x 320x240: 7 regions each): one row of gx workgroups per frame would launch a quarter of a million
	// one-wave workgroups of which most find nothing to do; the rows stride over the frames instead
	uint32_t gy = n_frames;
	if ((size_t)gx * gy > 131072u) {
		const uint32_t want = (uint32_t)(((size_t)stride / 8 + FRB - 1) / FRB) + 1;   // regions of a frame whose stream is an eighth of the worst case
		if (gx > want) gx = want;
		if ((size_t)gx * gy > 131072u) gy = 131072u / gx;
	}
	const dim3 grid(gx, gy);
	if (robust_only) {                                         // debugging aid: every frame through the robust kernels (bitmap form)
		if (bitmap) CK(hipMemsetAsync(A.tidx, 0xFF, (size_t)n_frames * (tpfd + 1) * 4, s));   // TIDX_NONE (otherwise k_fp_finish's job; nbad: k_fp_walk's)
		CK(hipMemsetD32Async((hipDeviceptr_t)A.fstate, (int)FS_BAD, n_frames, s));
		CK(hipMemsetD32Async((hipDeviceptr_t)A.nbad, (int)n_frames, 1, s));
	} else {
		if (c->mode512) hipLaunchKernelGGL(k_fp_walk<true>, grid, dim3(64), 0, s, A);
		else            hipLaunchKernelGGL(k_fp_walk<false>, grid, dim3(64), 0, s, A);
		CK(hipGetLastError());
		if (c->mode512) hipLaunchKernelGGL(k_fp_finish<true>, dim3(n_frames), dim3(64), 0, s, A);
		else            hipLaunchKernelGGL(k_fp_finish<false>, dim3(n_frames), dim3(64), 0, s, A);
		CK(hipGetLastError());
	}
	c->d_fp_fstate = A.fstate; c->fp_frames = n_frames;
	c->fp_vm = A.vm; c->fp_kb = A.kb; c->fp_tidx = A.tidx; c->fp_maxR = A.maxR;
	if (!bitmap) {
		uint32_t ge = gx;                                      // (a quarter / an eighth of it: 0.179 / 0.188 against 0.169 ms per 256 frames)
		if (getenv("AGMV_EXPAND_GX")) ge = (uint32_t)atoi(getenv("AGMV_EXPAND_GX"));   // tuning aid
		if (ge < 1) ge = 1;
		hipLaunchKernelGGL(k_fp_expand, dim3(ge, n_frames), dim3(64), 0, s, A);   // (one row per frame)
		CK(hipGetLastError());
		return parse_launch_robust(c, d_bits, stride, d_bpos, n_frames, nblk, d_offsets, d_nentered, ws_frames, A.fstate, s, nullptr, 0, A.nbad);
	}
	// bitmap form: the frames that could not be proven get their entry BITS from the robust kernels, are counted and
	// numbered by k_fp_tiles, which then looks up every frame's tile entries
	if (parse_launch_robust(c, d_bits, stride, d_bpos, n_frames, nblk, nullptr, d_nentered, ws_frames, A.fstate, s, A.vm, A.maxR, A.nbad)) return -1;
	uint32_t gt = gx / 2 ? gx / 2 : 1;
	if (getenv("AGMV_TILES_GX")) gt = (uint32_t)atoi(getenv("AGMV_TILES_GX"));   // tuning aid
	if (gt < 1) gt = 1;
	hipLaunchKernelGGL(k_fp_tiles, dim3(gt, gy), dim3(64), 0, s, A);
	CK(hipGetLastError());
	return 0;
}

extern "C" int agmv_hip_parse_frames_dev(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos,
                                         uint32_t n_frames, uint32_t w, uint32_t h, uint32_t* d_offsets, uint32_t* d_nentered,
                                         void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	uint32_t nblk = (w / 4) * (h / 4);
	hipStream_t s = (hipStream_t)stream;
	const char* mode = getenv("AGMV_HIP_PARSE");
	if (mode && strcmp(mode, "serial") == 0) {                 // debugging aid: one lane per frame
		hipLaunchKernelGGL(k_parse_serial, dim3((n_frames + 63) / 64), dim3(64), 0, s, d_bits,
		                   (unsigned long long)stride, d_bpos, n_frames, nblk, c->mode512, d_offsets, d_nentered);
		CK(hipGetLastError());
		return 0;
	}
	if (check_slab(d_bits, stride)) return -1;
	ev_mark(c, 2, s);
	if (parse_launch(c, d_bits, stride, d_bpos, n_frames, nblk, d_offsets, d_nentered, n_frames, s)) return -1;
	ev_mark(c, 3, s);
	return 0;
}

extern "C" int agmv_hip_parse_fallback_frames(agmv_hip_ctx* c, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (!c->d_fp_fstate || c->fp_frames == 0) return 0;
	CK(hipStreamSynchronize((hipStream_t)stream));
	uint32_t* h = (uint32_t*)malloc((size_t)c->fp_frames * 4);
	if (!h) { snprintf(g_err, sizeof(g_err), "agmv_hip: out of host memory"); return -1; }
	if (hipMemcpy(h, c->d_fp_fstate, (size_t)c->fp_frames * 4, hipMemcpyDeviceToHost) != hipSuccess) { free(h); snprintf(g_err, sizeof(g_err), "agmv_hip: D2H failed"); return -1; }
	int n = 0;
	for (uint32_t i = 0; i < c->fp_frames; i++) n += h[i] != FS_OK;
	free(h);
	return n;
}

// arguments of k_decode / k_fixup for a batch; grows and clears the context's bitmap of positions to repair
static int decode_prepare(agmv_hip_ctx* c, DecArgs& A, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos,
                          const uint32_t* d_offsets, const uint32_t* d_nentered, uint32_t n_frames, uint32_t w, uint32_t h,
                          uint32_t first_fc, uint32_t* d_out, const uint32_t* d_prev, const uint32_t* d_prev_iframe, hipStream_t s,
                          uint32_t* ndirty_out = nullptr /* != NULL: the caller has the bitmap cleared (k_fp_tiles) */)
{
	if (((uintptr_t)d_out & 15u) || ((uintptr_t)d_prev & 15u) || ((uintptr_t)d_prev_iframe & 15u)) {
		snprintf(g_err, sizeof(g_err), "agmv_hip: pixel buffers must be 16-byte aligned"); return -1;
	}
	if (check_slab(d_bits, stride)) return -1;
	memset(&A, 0, sizeof(A));
	A.bits = d_bits; A.stride = stride; A.bpos = d_bpos; A.offsets = d_offsets; A.nentered = d_nentered;
	A.out = d_out; A.pal = c->d_pal; A.prev = d_prev; A.prev_iframe = d_prev_iframe;
	A.n_frames = n_frames; A.w = w; A.h = h; A.bw = w / 4; A.nblk = (w / 4) * (h / 4);
	A.tpf = (A.nblk + DEC_T - 1) / DEC_T;
	A.first_fc = first_fc; A.phase = first_fc & 3u;
	A.n_groups = (n_frames + A.phase + 3) / 4;
	size_t nwords = (A.nblk + 31) / 32 + 2;                    // bitmap + the "anything to repair" word + the "depends on the prior state" word
	if (nwords > c->dirty_cap) {
		if (c->d_dirty) CK(hipFree(c->d_dirty));
		c->d_dirty = nullptr; c->dirty_cap = 0;
		CK(hipMalloc(&c->d_dirty, nwords * 4));
		c->dirty_cap = nwords;
	}
	A.dirty = c->d_dirty;
	if (ndirty_out) *ndirty_out = (uint32_t)nwords;
	else CK(hipMemsetAsync(c->d_dirty, 0, nwords * 4, s));
	return 0;
}

static int decode_launch(agmv_hip_ctx* c, DecArgs A, uint32_t g0, uint32_t g1, hipStream_t s)   // GOPs [g0, g1) of the batch
{
	A.grp0 = g0;
	const dim3 grid((g1 - g0) * A.tpf);
	if (A.vm) {
		if (c->mode512) hipLaunchKernelGGL((k_decode<true, true>), grid, dim3(DEC_T), 0, s, A);
		else            hipLaunchKernelGGL((k_decode<false, true>), grid, dim3(DEC_T), 0, s, A);
	} else {
		if (c->mode512) hipLaunchKernelGGL((k_decode<true, false>), grid, dim3(DEC_T), 0, s, A);
		else            hipLaunchKernelGGL((k_decode<false, false>), grid, dim3(DEC_T), 0, s, A);
	}
	CK(hipGetLastError());
	return 0;
}

static int fixup_launch(agmv_hip_ctx* c, const DecArgs& A, hipStream_t s)
{
	const dim3 grid((A.nblk + 63) / 64);
	if (A.vm) {
		if (c->mode512) hipLaunchKernelGGL((k_fixup<true, true>), grid, dim3(64), 0, s, A);
		else            hipLaunchKernelGGL((k_fixup<false, true>), grid, dim3(64), 0, s, A);
	} else {
		if (c->mode512) hipLaunchKernelGGL((k_fixup<true, false>), grid, dim3(64), 0, s, A);
		else            hipLaunchKernelGGL((k_fixup<false, false>), grid, dim3(64), 0, s, A);
	}
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
	hipStream_t s = (hipStream_t)stream;
	DecArgs A;
	if (decode_prepare(c, A, d_bits, stride, d_bpos, d_offsets, d_nentered, n_frames, w, h, first_fc, d_out, d_prev, d_prev_iframe, s)) return -1;
	ev_mark(c, 4, s);
	if (decode_launch(c, A, 0, A.n_groups, s)) return -1;
	if (fixup_launch(c, A, s)) return -1;
	ev_mark(c, 5, s);
	return 0;
}

// Parse + reconstruct as ONE call; optionally (AGMV_DEC_SLICES=n) cut into n ranges of GOPs with the parser on a stream of
// the context's own, so that the parse of range k+1 runs beside the reconstruction of range k (k_fixup needs every
// frame's offsets and runs last).  Measured (profiles/r02/k_decode_experiments.txt): the kernels do run side by side but
// take from each other what they gain -- k_decode needs its full occupancy -- so the default is one range.
extern "C" int agmv_hip_parse_decode_frames_dev(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos,
                                                uint32_t n_frames, uint32_t w, uint32_t h, uint32_t first_fc,
                                                uint32_t* d_offsets, uint32_t* d_nentered, uint32_t* d_out,
                                                const uint32_t* d_prev, const uint32_t* d_prev_iframe, void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	hipStream_t s = (hipStream_t)stream;
	DecArgs A;
	if (decode_prepare(c, A, d_bits, stride, d_bpos, d_offsets, d_nentered, n_frames, w, h, first_fc, d_out, d_prev, d_prev_iframe, s)) return -1;
	uint32_t nsl = 1;
	if (getenv("AGMV_DEC_SLICES")) nsl = (uint32_t)atoi(getenv("AGMV_DEC_SLICES"));
	if (nsl > A.n_groups) nsl = A.n_groups;
	if (nsl > (uint32_t)DEC_MAX_SLICES) nsl = DEC_MAX_SLICES;
	if (nsl < 1) nsl = 1;
	const uint32_t gps = (A.n_groups + nsl - 1) / nsl;         // GOPs per range
	auto first_frame = [&](uint32_t g) -> uint32_t { const long f = (long)g * 4 - (long)A.phase; return f < 0 ? 0u : ((uint32_t)f > n_frames ? n_frames : (uint32_t)f); };
	const size_t ws_frames = (size_t)gps * 4;
	ev_mark(c, 6, s);
	if (nsl == 1) {
		if (parse_launch(c, d_bits, stride, d_bpos, n_frames, A.nblk, d_offsets, d_nentered, n_frames, s)) return -1;
		if (decode_launch(c, A, 0, A.n_groups, s)) return -1;
	} else {
		if (!c->aux_stream) {
			CK(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
			CK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
			for (int i = 0; i < DEC_MAX_SLICES; i++) CK(hipEventCreateWithFlags(&c->ev_slice[i], hipEventDisableTiming));
		}
		CK(hipEventRecord(c->ev_fork, s));                     // the bitstreams are complete on the caller's stream
		CK(hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
		uint32_t k = 0;
		for (uint32_t g0 = 0; g0 < A.n_groups; g0 += gps, k++) {
			const uint32_t g1 = g0 + gps < A.n_groups ? g0 + gps : A.n_groups;
			const uint32_t f0 = first_frame(g0), f1 = first_frame(g1);
			if (parse_launch(c, d_bits + (size_t)f0 * stride, stride, d_bpos + f0, f1 - f0, A.nblk, d_offsets + (size_t)f0 * A.nblk,
			                 d_nentered + f0, ws_frames, c->aux_stream)) return -1;
			CK(hipEventRecord(c->ev_slice[k], c->aux_stream));
			CK(hipStreamWaitEvent(s, c->ev_slice[k], 0));
			if (decode_launch(c, A, g0, g1, s)) return -1;
		}
	}
	if (fixup_launch(c, A, s)) return -1;
	ev_mark(c, 7, s);
	return 0;
}

// Parse + reconstruct without offsets[]: the parser's entry bitmaps go straight to k_decode, which ranks its own blocks in
// them (k_fp_tiles tells every tile where it starts).  Against agmv_hip_parse_decode_frames_dev this drops k_fp_expand and
// the 4 bytes per block it writes and k_decode reads back.  Batches of more than 65532 frames are cut at GOP boundaries
// (the parser's grid has one row per frame); each part continues from the decoder state the part before it left.
extern "C" int agmv_hip_decode_bitstreams_dev(agmv_hip_ctx* c, const uint8_t* d_bits, size_t stride, const uint32_t* d_bpos,
                                              uint32_t n_frames, uint32_t w, uint32_t h, uint32_t first_fc,
                                              uint32_t* d_nentered, uint32_t* d_out,
                                              const uint32_t* d_prev, const uint32_t* d_prev_iframe, void* stream)
{
	if (need_ctx(c, true)) return -1;
	if (check_geometry(w, h)) return -1;
	if (n_frames == 0) return 0;
	hipStream_t s = (hipStream_t)stream;
	if (!d_nentered) {
		if (n_frames > c->nent_cap) {
			if (c->d_nent_own) CK(hipFree(c->d_nent_own));
			c->d_nent_own = nullptr; c->nent_cap = 0;
			CK(hipMalloc(&c->d_nent_own, (size_t)n_frames * 4));
			c->nent_cap = n_frames;
		}
		d_nentered = c->d_nent_own;
	}
	const size_t npx = (size_t)w * h;
	constexpr uint32_t PART = 65532u;                          // a multiple of 4
	ev_mark(c, 6, s);
	for (uint32_t f0 = 0; f0 < n_frames;) {
		uint32_t f1 = n_frames;
		if (f1 - f0 > PART) { f1 = f0 + PART; f1 -= (first_fc + f1) & 3u; }   // the next part starts with an I-frame
		const uint32_t n = f1 - f0, fc = first_fc + f0;
		// state before frame f0: the frame before it, and the snapshot taken at the last I-frame (the decoded I-frame itself, :401-405)
		const uint32_t* prev = f0 == 0 ? d_prev : d_out + (size_t)(f0 - 1) * npx;
		const uint32_t* previ = f0 == 0 ? d_prev_iframe : d_out + (size_t)(f0 - 4) * npx;
		DecArgs A;
		uint32_t ndirty = 0;
		if (decode_prepare(c, A, d_bits + (size_t)f0 * stride, stride, d_bpos + f0, nullptr, d_nentered + f0, n, w, h, fc, d_out + (size_t)f0 * npx, prev, previ, s, &ndirty)) return -1;
		ev_mark(c, 2, s);
		if (parse_launch(c, d_bits + (size_t)f0 * stride, stride, d_bpos + f0, n, A.nblk, nullptr, d_nentered + f0, n, s, true, A.dirty, ndirty)) return -1;
		ev_mark(c, 3, s);
		A.vm = c->fp_vm; A.kb = c->fp_kb; A.tidx = c->fp_tidx; A.maxR = c->fp_maxR;
		ev_mark(c, 4, s);
		if (decode_launch(c, A, 0, A.n_groups, s)) return -1;
		if (fixup_launch(c, A, s)) return -1;
		ev_mark(c, 5, s);
		f0 = f1;
	}
	ev_mark(c, 7, s);
	return 0;
}

extern "C" int agmv_hip_decode_prior_dependent(agmv_hip_ctx* c, uint32_t w, uint32_t h, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (!c->d_dirty) { snprintf(g_err, sizeof(g_err), "agmv_hip: no decode has run on this context"); return -1; }
	const size_t nblk = (size_t)(w / 4) * (h / 4);
	uint32_t v = 0;
	CK(hipStreamSynchronize((hipStream_t)stream));
	CK(hipMemcpy(&v, c->d_dirty + (nblk + 31) / 32 + 1, 4, hipMemcpyDeviceToHost));
	return v ? 1 : 0;
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

// ----------------------------------------------------------------------------------------------
// The exchange step of the GOP-sharded encoder (usize per frame, then the variable-length bitstreams to the
// root): a rank's frames leave as ONE contiguous message -- the used bytes of every slab row back to back, frame f at
// offsets[f] -- and arrive in a slab again.  k_pack_scan: offsets = exclusive sums of the sizes (one workgroup);
// k_pack_copy: rows <-> message, a few workgroups per frame striding over 16 KB chunks.  The unaligned side (the message)
// is addressed byte-granularly with dword accesses: gfx950 runs global memory in unaligned mode.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_scan(const uint32_t* __restrict__ sizes, uint32_t n, unsigned long long* __restrict__ offsets)
{
	__shared__ unsigned long long s_w[4];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	unsigned long long run = 0;
	for (uint32_t i0 = 0; i0 < n; i0 += 256) {
		const uint32_t i = i0 + threadIdx.x;
		const uint32_t v = i < n ? sizes[i] : 0u;
		const uint32_t incl = wave_incl_scan(v, lane);
		if (lane == 63) s_w[wave] = incl;
		__syncthreads();
		unsigned long long base = run;
		for (int k = 0; k < wave; k++) base += s_w[k];
		if (i < n) offsets[i] = base + incl - v;
		run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
		__syncthreads();
	}
	if (threadIdx.x == 0) offsets[n] = run;
}

constexpr uint32_t PACK_CHUNK = 16384;
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_pack_copy(uint8_t* __restrict__ slab, unsigned long long stride, const uint32_t* __restrict__ sizes,
                                                   const unsigned long long* __restrict__ offsets, uint8_t* __restrict__ msg, uint32_t n_frames)
{
	for (uint32_t f = blockIdx.y; f < n_frames; f += gridDim.y) {
		const uint32_t size = sizes[f];
		uint8_t* row = slab + (size_t)f * stride;                  // 256-byte aligned (agmv_hip_max_usize)
		uint8_t* m = msg + offsets[f];                             // any alignment
		for (uint32_t c = blockIdx.x * PACK_CHUNK; c < size; c += gridDim.x * PACK_CHUNK) {
			const uint32_t len = min(PACK_CHUNK, size - c), nd = len >> 2;
			for (uint32_t d = threadIdx.x; d < nd; d += 256) {
				if (UNPACK) *(uint32_t*)(row + c + 4u * d) = *(const u32u*)(m + c + 4u * d);
				else *(u32u*)(m + c + 4u * d) = *(const uint32_t*)(row + c + 4u * d);
			}
			const uint32_t t = 4u * nd + threadIdx.x;              // the <= 3 bytes behind the last whole dword
			if (t < len) {
				if (UNPACK) row[c + t] = m[c + t]; else m[c + t] = row[c + t];
			}
		}
	}
}

static int pack_launch(agmv_hip_ctx* c, bool unpack, uint8_t* d_slab, size_t stride, const uint32_t* d_sizes, uint32_t n_frames,
                       uint8_t* d_msg, unsigned long long* d_offsets, hipStream_t s)
{
	if (!d_slab || !d_sizes || !d_msg || !d_offsets) { snprintf(g_err, sizeof(g_err), "agmv_hip: pack/unpack: NULL argument"); return -1; }
	if (stride & 3u) { snprintf(g_err, sizeof(g_err), "agmv_hip: pack/unpack: the slab stride must be a multiple of 4 (agmv_hip_max_usize is)"); return -1; }
	hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(256), 0, s, d_sizes, n_frames, d_offsets);
	CK(hipGetLastError());
	uint32_t gx = (uint32_t)((stride / 8 + PACK_CHUNK - 1) / PACK_CHUNK);   // chunks of a frame whose stream is an eighth of the worst case
	if (gx < 1) gx = 1;
	uint32_t gy = n_frames;
	if ((size_t)gx * gy > 65536u) { gy = 65536u / gx; if (gy < 1) gy = 1; }
	if (unpack) hipLaunchKernelGGL(k_pack_copy<true>, dim3(gx, gy), dim3(256), 0, s, d_slab, (unsigned long long)stride, d_sizes, d_offsets, d_msg, n_frames);
	else        hipLaunchKernelGGL(k_pack_copy<false>, dim3(gx, gy), dim3(256), 0, s, d_slab, (unsigned long long)stride, d_sizes, d_offsets, d_msg, n_frames);
	CK(hipGetLastError());
	(void)c;
	return 0;
}

extern "C" int agmv_hip_pack_frames_dev(agmv_hip_ctx* c, const uint8_t* d_slab, size_t stride, const uint32_t* d_sizes, uint32_t n_frames,
                                        uint8_t* d_msg, unsigned long long* d_offsets, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (n_frames == 0) return 0;
	return pack_launch(c, false, (uint8_t*)d_slab, stride, d_sizes, n_frames, d_msg, d_offsets, (hipStream_t)stream);
}

extern "C" int agmv_hip_unpack_frames_dev(agmv_hip_ctx* c, const uint8_t* d_msg, const uint32_t* d_sizes, uint32_t n_frames,
                                          uint8_t* d_slab, size_t stride, unsigned long long* d_offsets, void* stream)
{
	if (need_ctx(c, false)) return -1;
	if (n_frames == 0) return 0;
	return pack_launch(c, true, d_slab, stride, d_sizes, n_frames, (uint8_t*)d_msg, d_offsets, (hipStream_t)stream);
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

// ---- streams, pinned staging and asynchronous copies for C hosts (the pipelined drivers of agmv_pipeline.c) ----
extern "C" void* agmv_hip_stream_create(agmv_hip_ctx* c)
{
	if (need_ctx(c, false)) return nullptr;
	hipStream_t s = nullptr;
	CKP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	return (void*)s;
}
extern "C" void agmv_hip_stream_destroy(agmv_hip_ctx* c, void* stream)
{
	if (!c || !stream) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamDestroy((hipStream_t)stream);
}
extern "C" int agmv_hip_stream_sync(agmv_hip_ctx* c, void* stream)
{
	if (need_ctx(c, false)) return -1;
	CK(hipStreamSynchronize((hipStream_t)stream));
	return 0;
}
extern "C" void* agmv_hip_host_alloc(size_t bytes)
{
	void* p = nullptr;
	if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: hipHostMalloc(%zu) failed", bytes); return nullptr; }
	return p;
}
extern "C" void agmv_hip_host_free(void* h) { if (h) (void)hipHostFree(h); }
extern "C" void* agmv_hip_malloc_on(agmv_hip_ctx* c, size_t bytes)
{
	if (need_ctx(c, false)) return nullptr;
	void* p = nullptr;
	if (hipMalloc(&p, bytes) != hipSuccess) { snprintf(g_err, sizeof(g_err), "agmv_hip: hipMalloc(%zu) failed on device %d", bytes, c->device); return nullptr; }
	return p;
}
extern "C" void agmv_hip_free_on(agmv_hip_ctx* c, void* d) { if (c && d) { (void)hipSetDevice(c->device); (void)hipFree(d); } }
extern "C" int agmv_hip_memcpy_async(agmv_hip_ctx* c, void* dst, const void* src, size_t n, int kind, void* stream)
{
	if (need_ctx(c, false)) return -1;
	const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
	CK(hipMemcpyAsync(dst, src, n, k, (hipStream_t)stream));
	return 0;
}
extern "C" int agmv_hip_memset_async(agmv_hip_ctx* c, void* d, int v, size_t n, void* stream)
{
	if (need_ctx(c, false)) return -1;
	CK(hipMemsetAsync(d, v, n, (hipStream_t)stream));
	return 0;
}
extern "C" int agmv_hip_ctx_device(agmv_hip_ctx* c) { return c ? c->device : -1; }

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
