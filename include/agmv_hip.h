/*
 * include/agmv_hip.h -- the C-ABI drop-in boundary of the MI355X hot path.
 *
 * libagmv has no plugin/FFI layer: consumers call its C API directly
 * (reference include/agmv_encode.h:26-37, include/agmv_decode.h:21-26).  The
 * per-frame functions of that API are FILE*-coupled, one frame per call, and
 * take LP64 `unsigned long` pixels (8 B/px, reference include/agmv_defines.h:22)
 * -- a shape no GPU can be fed through.  This header therefore declares the
 * batch entry points that the reference-compatible host layer (include/agmv.h,
 * libagmv_amd/csrc/agmv_api.c) is built on, and that a maintainer of the
 * reference would bind.  Plain pointers and sizes only.
 *
 * Conventions
 *   pixel   4-byte 0x00RRGGBB (bits >= 24 ignored, reference src/agmv_utils.c:632-642)
 *   frame   w*h pixels, row-major, w and h multiples of 4 (reference src/agmv_encode.c:365-366)
 *   entry   u16 = pal_num << 8 | index   (GPU form of AGMV_ENTRY, include/agmv_defines.h:122-126)
 *   d_*     device pointers (HIP), h_* host pointers; `stream` is a hipStream_t (NULL = default)
 *   return  0 on success, negative on error (agmv_hip_last_error() gives the text).
 *           There is NO CPU fallback: without a usable GPU every entry point fails.
 */
#ifndef AGMV_HIP_H
#define AGMV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct agmv_hip_ctx agmv_hip_ctx;

/* bytes a frame's pre-LZ bitstream can take at most: 33 (512-colour) or 17 (256-colour)
   bytes per 4x4 block (reference src/agmv_encode.c:381-403 / :420-431), rounded up to 256.
   NOTE the reference's own buffer (w*h*2, src/agmv_utils.c:338) is too small for the
   512-colour worst case. */
size_t agmv_hip_max_usize(uint32_t w, uint32_t h, int mode512);

int          agmv_hip_device_count(void);
agmv_hip_ctx* agmv_hip_create(int device);
void         agmv_hip_destroy(agmv_hip_ctx* ctx);
const char*  agmv_hip_last_error(void);

/* -- palette ------------------------------------------------------------------------------
 * Replaces the per-pixel 256/512-way search of AGMV_FindNearestColor / AGMV_FindNearestEntry
 * (reference src/agmv_utils.c:785-816, :851-895) by an exact 2^24-entry table built ONCE per
 * palette on the GPU with the same argmin and the same tie rules (lowest index inside a
 * palette, palette0 on cross-palette ties).  Also builds the 512x512 "within +-2 per channel"
 * bit matrix used by the block tests (reference src/agmv_encode.c:293, :345).
 * mode512: 1 = two palettes (OPT_I/III/GBA_I/GBA_III/NDS, container v1/v3),
 *          0 = palette0 only (OPT_II/ANIM/GBA_II, container v2/v4). */
int agmv_hip_set_palette(agmv_hip_ctx* ctx, const uint32_t p0[256], const uint32_t p1[256],
                         int mode512, void* stream);

/* exact nearest entries for n pixels (loop A of AGMV_EncodeFrame, src/agmv_encode.c:556-558 /
   :589-592) -- exposed for parity tests of the table. */
int agmv_hip_quantise_dev(agmv_hip_ctx* ctx, const uint32_t* d_pix, size_t n, uint16_t* d_entries,
                          void* stream);

/* -- encode -------------------------------------------------------------------------------
 * Loops A+B of AGMV_EncodeFrame for n_frames consecutive frames (reference
 * src/agmv_encode.c:552-565 / :589-599, :626-630): quantise, I/P block classification
 * (CompareIFrameBlock :302-352, ComparePFrameBlock :240-300), byte assembly
 * (AssembleIFrameBitstream :354-436, AssemblePFrameBitstream :438-527).
 * Frame f of the batch has frame_count = first_frame_count + f; it is an I-frame when
 * frame_count % 4 == 0.  d_out receives frame f's pre-LZ bitstream at d_out + f*out_stride
 * (out_stride >= agmv_hip_max_usize), d_sizes[f] its length (`usize`).
 * d_iframe_entries (w*h u16, may be NULL): if the batch starts inside a GOP
 * (first_frame_count % 4 != 0) it supplies the entries of that GOP's I-frame
 * (agmv->iframe_entries); on return it holds the entries of the last I-frame of the batch. */
int agmv_hip_encode_frames_dev(agmv_hip_ctx* ctx, const uint32_t* d_pix, uint32_t n_frames,
                               uint32_t w, uint32_t h, uint32_t first_frame_count,
                               uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                               uint16_t* d_iframe_entries, void* stream);
/* same from/to host memory (does the H2D/D2H itself, synchronous) */
int agmv_hip_encode_frames(agmv_hip_ctx* ctx, const uint32_t* h_pix, uint32_t n_frames,
                           uint32_t w, uint32_t h, uint32_t first_frame_count,
                           uint8_t* h_out, size_t out_stride, uint32_t* h_sizes,
                           uint16_t* h_iframe_entries);

/* The same on planes of ENTRIES instead of pixels: word k of frame f holds pal_num << 8 | index of pixel k.  The
   quantisation is skipped; classification and assembly are those of AGMV_AssembleIFrameBitstream /
   AGMV_AssemblePFrameBitstream on that AGMV_ENTRY plane (reference src/agmv_encode.c:354-436, :438-527). */
int agmv_hip_encode_entries_dev(agmv_hip_ctx* ctx, const uint32_t* d_entries, uint32_t n_frames,
                                uint32_t w, uint32_t h, uint32_t first_frame_count,
                                uint8_t* d_out, size_t out_stride, uint32_t* d_sizes,
                                uint16_t* d_iframe_entries, void* stream);
int agmv_hip_encode_entries(agmv_hip_ctx* ctx, const uint32_t* h_entries, uint32_t n_frames,
                            uint32_t w, uint32_t h, uint32_t first_frame_count,
                            uint8_t* h_out, size_t out_stride, uint32_t* h_sizes,
                            uint16_t* h_iframe_entries);

/* nearest colour / entry of n pixels against palettes that need not be the context's (no table is built):
   AGMV_FindNearestColor (mode512 = 0, entry = index in p0) / AGMV_FindNearestEntry (mode512 = 1), reference
   src/agmv_utils.c:785-816, :851-895.  Host buffers, synchronous; the device scratch is cached in the context. */
int agmv_hip_nearest(agmv_hip_ctx* ctx, const uint32_t p0[256], const uint32_t p1[256], int mode512,
                     const uint32_t* h_pix, size_t n, uint16_t* h_entries);

/* number of the 16 colour pairs (a[k], b[k]) that are within +-2 on R, G and B: the count AGMV_CompareIFrameBlock
   (b = the block's reference colour 16 times) and AGMV_ComparePFrameBlock (b = the I-frame entries' colours) return
   (reference src/agmv_encode.c:302-352, :240-300).  Returns 0..16, negative on error. */
int agmv_hip_within2_count(agmv_hip_ctx* ctx, const uint32_t a[16], const uint32_t b[16]);

/* Notes on a context:
 *  - device memory: 512 MiB of address space for the colour -> entry table (32 MiB of it populated, see
 *    k_lut_build in agmv_hip.hip), plus per-batch work areas grown on demand (look-back status 8 B per tile and
 *    frame, parser workspace <= 15 % of the bitstream slab);
 *  - the encode entry points of ONE context share its look-back status and control words: a second encode is
 *    ordered behind the first (on another stream it waits for it through an event); use one context per
 *    concurrent encoder;
 *  - likewise the parse / decode entry points of ONE context share its parser work areas and repair bitmap: calls on
 *    different streams of one context must not overlap; use one context per concurrent decoder;
 *  - a device-side wait that runs into its bound (never observed on a healthy GPU) makes agmv_hip_check fail AND
 *    overwrites every size of that batch with 0xFFFFFFFF, so the bytes cannot be taken for valid ones. */

/* -- decode -------------------------------------------------------------------------------
 * The parse + reconstruct half of AGMV_DecodeFrameChunk (reference src/agmv_decode.c:224-407)
 * for n_frames consecutive frames whose LZ stage (:171-222, host) has already run.
 *   d_bits + f*bits_stride : frame f's decompressed bitstream; bytes [bpos, bpos+16) must hold
 *                            what the reference's persistent buffer holds there (stale bytes of
 *                            earlier frames, or 0) -- they are read on over-run.
 *   d_bpos[f]              : bitstream->pos after the LZ stage (may differ from usize).
 * agmv_hip_parse_frames_dev computes, per frame, the byte position at which each 4x4 block is
 * entered (d_offsets[f*nblk + k]) and how many blocks are entered before the reference raises
 * `escape` (d_nentered[f]) -- by speculative per-piece walks that are PROVEN per frame to be the serial parse, and by the
 * map / stitch / emit kernels for the frames that cannot be proven (AGMV_HIP_PARSE=robust in the
 * environment runs the latter alone, =serial a one-lane walk: debugging aids, same outputs).
 * agmv_hip_decode_frames_dev turns that into pixels:
 * d_pix_out + f*w*h.  d_prev_frame / d_prev_iframe (w*h pixels, may be NULL = zeroed, the state
 * of a fresh decoder) are img_data / iframe->img_data before the first frame of the batch;
 * blocks the bitstream does not reach keep the previous frame's pixels (:229-232). */
int agmv_hip_parse_frames_dev(agmv_hip_ctx* ctx, const uint8_t* d_bits, size_t bits_stride,
                              const uint32_t* d_bpos, uint32_t n_frames, uint32_t w, uint32_t h,
                              uint32_t* d_offsets, uint32_t* d_nentered, void* stream);
int agmv_hip_decode_frames_dev(agmv_hip_ctx* ctx, const uint8_t* d_bits, size_t bits_stride,
                               const uint32_t* d_bpos, const uint32_t* d_offsets,
                               const uint32_t* d_nentered, uint32_t n_frames, uint32_t w,
                               uint32_t h, uint32_t first_frame_count, uint32_t* d_pix_out,
                               const uint32_t* d_prev_frame, const uint32_t* d_prev_iframe,
                               void* stream);
/* How many frames of the last agmv_hip_parse_frames_dev call (of the last range, for agmv_hip_parse_decode_frames_dev)
   the speculative parser could not prove and left to the robust kernels (0 for streams the encoder emits unless a
   long run of FILL blocks carries a flag-valued index; damaged streams typically land here).  A statistic: the
   outputs are the same either way.  Synchronises the stream; negative on error. */
int agmv_hip_parse_fallback_frames(agmv_hip_ctx* ctx, void* stream);
/* Both steps as ONE call.  Same inputs, same outputs (d_offsets / d_nentered are written as by
   agmv_hip_parse_frames_dev), same pixels.  With AGMV_DEC_SLICES=n in the environment the batch is cut into n ranges of
   GOPs, the parser runs on a stream the context owns and the reconstruction of a range waits only for the parse of that
   range (measured: no gain on MI355X, the default is one range).  Everything is ordered after the work already on
   `stream`, and `stream` is complete only when the whole call is. */
int agmv_hip_parse_decode_frames_dev(agmv_hip_ctx* ctx, const uint8_t* d_bits, size_t bits_stride,
                                     const uint32_t* d_bpos, uint32_t n_frames, uint32_t w, uint32_t h,
                                     uint32_t first_frame_count, uint32_t* d_offsets, uint32_t* d_nentered,
                                     uint32_t* d_pix_out, const uint32_t* d_prev_frame,
                                     const uint32_t* d_prev_iframe, void* stream);
/* The same result without offsets[]: the parser's block-entry bitmaps (one bit per byte of stream) go straight to the
   reconstruction kernel, which ranks its own blocks in them; frames the speculative parser cannot prove get their bits from
   the robust kernels.  This is the form the sequence drivers and bench.py use: it saves the 4 bytes per block that
   agmv_hip_parse_frames_dev writes and agmv_hip_decode_frames_dev reads back (reference: the block loop of
   AGMV_DecodeFrameChunk, src/agmv_decode.c:224-407, has no such table either -- it walks).  d_nentered may be NULL.
   Any number of frames (more than 65532 are cut at GOP boundaries internally). */
int agmv_hip_decode_bitstreams_dev(agmv_hip_ctx* ctx, const uint8_t* d_bits, size_t bits_stride,
                                   const uint32_t* d_bpos, uint32_t n_frames, uint32_t w, uint32_t h,
                                   uint32_t first_frame_count, uint32_t* d_nentered, uint32_t* d_pix_out,
                                   const uint32_t* d_prev_frame, const uint32_t* d_prev_iframe, void* stream);
/* After agmv_hip_decode_frames_dev / agmv_hip_parse_decode_frames_dev / agmv_hip_decode_bitstreams_dev: 1 when a pixel of that batch derives from d_prev_frame / d_prev_iframe (a block the
   bitstream did not rewrite before it was read: stale tail after `escape`, COPY in the first GOP, a FILL / NORMAL block cut
   off by bpos -- reference src/agmv_decode.c:229-232, :268-271, :277-285, :310-314), 0 when the batch is a function of its
   own bitstreams alone, negative on error.  What a GOP-sharded decode needs to know before it trusts a range decoded from
   the fresh state (never 1 for a stream the encoder emits).  Synchronises the stream. */
int agmv_hip_decode_prior_dependent(agmv_hip_ctx* ctx, uint32_t w, uint32_t h, void* stream);
/* host-memory convenience: parse on the GPU, reconstruct, copy back (synchronous) */
int agmv_hip_decode_frames(agmv_hip_ctx* ctx, const uint8_t* h_bits, size_t bits_stride,
                           const uint32_t* h_bpos, uint32_t n_frames, uint32_t w, uint32_t h,
                           uint32_t first_frame_count, uint32_t* h_pix_out,
                           const uint32_t* h_prev_frame, const uint32_t* h_prev_iframe);

/* -- exchange step of the GOP-sharded encoder (the reference is single-process: its "gather" is the frame
 * loop of AGMV_EncodeAGMV appending chunk after chunk, src/agmv_encode.c:3610-3612) --------------------------------
 * A rank's bitstreams travel as ONE contiguous message: the used bytes of every slab row back to back.
 * agmv_hip_pack_frames_dev:   d_slab [n_frames][stride] + d_sizes[n_frames] (as agmv_hip_encode_frames_dev leaves them)
 *                             -> d_msg (sum of the sizes bytes; the caller sizes it, e.g. from the sizes it exchanges first)
 *                             and d_offsets[n_frames + 1] (u64: frame f starts at d_offsets[f]; [n_frames] = the total).
 * agmv_hip_unpack_frames_dev: the inverse on the receiving rank: d_msg + d_sizes -> rows of a slab (stride a multiple of 4;
 *                             bytes of a row behind its size are left as they are); d_offsets is scratch of n_frames + 1 u64.
 * Both are asynchronous on `stream`; what moves the message between the ranks (RCCL send/recv, hipMemcpyPeer, a host
 * socket) is the caller's business. */
int agmv_hip_pack_frames_dev(agmv_hip_ctx* ctx, const uint8_t* d_slab, size_t stride, const uint32_t* d_sizes,
                             uint32_t n_frames, uint8_t* d_msg, unsigned long long* d_offsets, void* stream);
int agmv_hip_unpack_frames_dev(agmv_hip_ctx* ctx, const uint8_t* d_msg, const uint32_t* d_sizes, uint32_t n_frames,
                               uint8_t* d_slab, size_t stride, unsigned long long* d_offsets, void* stream);

/* -- helpers on the caller side of the path -------------------------------------------------*/
/* canonical synthetic clip agmv_synth_v1: frames t0..t0+n-1 into d_pix */
int agmv_hip_synth_dev(agmv_hip_ctx* ctx, uint32_t* d_pix, uint32_t w, uint32_t h, uint32_t t0,
                       uint32_t n_frames, uint64_t seed, void* stream);
/* PDIFS midpoint frame, AGMV_InterpFrame (reference src/agmv_utils.c:949-969) */
int agmv_hip_interp_dev(agmv_hip_ctx* ctx, uint32_t* d_out, const uint32_t* d_f1,
                        const uint32_t* d_f2, size_t n_pixels, void* stream);
/* pass-1 colour histogram of the palette build: hist[AGMV_QuantizeColor(px, quality)] += 1
   (reference src/agmv_encode.c:2390-2394, src/agmv_utils.c:695-742); hist has 2^19 bins */
int agmv_hip_histogram_dev(agmv_hip_ctx* ctx, const uint32_t* d_pix, size_t n_pixels, int quality,
                           uint32_t* d_hist, void* stream);

/* optional timing: when enabled the library records HIP events on the caller's stream around its three kernel
   groups; agmv_hip_last_kernel_ms(which) returns the last launch's duration in ms (0 = k_encode, 1 = the parser
   kernels, 2 = k_decode + k_fixup, 3 = the whole of agmv_hip_parse_decode_frames_dev / agmv_hip_decode_bitstreams_dev), or a negative value if
   unavailable */
int   agmv_hip_enable_timing(agmv_hip_ctx* ctx, int on);
float agmv_hip_last_kernel_ms(agmv_hip_ctx* ctx, int which);

/* check for an asynchronous device-side failure (look-back timeout) after synchronising */
int agmv_hip_check(agmv_hip_ctx* ctx, void* stream);

/* streams, pinned host staging and asynchronous copies for C hosts that do not link the HIP runtime themselves (the
   pipelined sequence drivers: H2D || kernels || D2H || host LZ).  The *_on / *_async forms act on the context's device;
   kind: 0 = host to device, 1 = device to host, 2 = device to device.  Host memory of asynchronous copies must come from
   agmv_hip_host_alloc. */
void* agmv_hip_stream_create(agmv_hip_ctx* ctx);
void  agmv_hip_stream_destroy(agmv_hip_ctx* ctx, void* stream);
int   agmv_hip_stream_sync(agmv_hip_ctx* ctx, void* stream);
void* agmv_hip_host_alloc(size_t bytes);
void  agmv_hip_host_free(void* h);
void* agmv_hip_malloc_on(agmv_hip_ctx* ctx, size_t bytes);
void  agmv_hip_free_on(agmv_hip_ctx* ctx, void* d);
int   agmv_hip_memcpy_async(agmv_hip_ctx* ctx, void* dst, const void* src, size_t n, int kind, void* stream);
int   agmv_hip_memset_async(agmv_hip_ctx* ctx, void* d, int v, size_t n, void* stream);
int   agmv_hip_ctx_device(agmv_hip_ctx* ctx);

/* raw device memory for C hosts that do not link the HIP runtime themselves (current device) */
void* agmv_hip_malloc(size_t bytes);
void  agmv_hip_free(void* d);
int   agmv_hip_memcpy_h2d(void* d, const void* h, size_t bytes);
int   agmv_hip_memcpy_d2h(void* h, const void* d, size_t bytes);
int   agmv_hip_memset(void* d, int value, size_t bytes);
int   agmv_hip_sync(void);

#ifdef __cplusplus
}
#endif
#endif
