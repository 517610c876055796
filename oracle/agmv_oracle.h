/*
 * oracle/agmv_oracle.h -- CPU restatement of libagmv's per-frame hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load liboracle.so.  The product
 * (libagmv_amd/) never links, imports or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked against the compiled,
 * unmodified reference (oracle/_ref/libagmv_ref.so, built by oracle/Makefile
 * from /root/reference) in tests/test_oracle_vs_ref.py, and against the golden
 * vectors under tests/golden/ (generated from that same reference build by
 * tests/golden/make_golden.py).
 *
 * Conventions: pixels are 4-byte 0x00RRGGBB words (the reference's `u32` is an
 * 8-byte unsigned long on LP64; only the low 24 bits are ever read,
 * src/agmv_utils.c:632-642).  An "entry" is (pal_num << 8 | index), the GPU form
 * of AGMV_ENTRY (include/agmv_defines.h:122-126).
 */
#ifndef AGMV_ORACLE_H
#define AGMV_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#define ORC_FILL_FLAG   0x4E   /* include/agmv_defines.h:49 */
#define ORC_NORMAL_FLAG 0x2F   /* include/agmv_defines.h:50 */
#define ORC_COPY_FLAG   0x5E   /* include/agmv_defines.h:51 */
#define ORC_FILL_COUNT  14     /* include/agmv_defines.h:52 */
#define ORC_COPY_COUNT  13     /* include/agmv_defines.h:53 */

/* E2  src/agmv_utils.c:785-816 */
uint8_t  orc_find_nearest_color(const uint32_t pal[256], uint32_t color);
/* E3  src/agmv_utils.c:851-895 */
uint16_t orc_find_nearest_entry(const uint32_t p0[256], const uint32_t p1[256], uint32_t color);
/* E4  src/agmv_encode.c:556-558 (512) / :589-592 (256) */
void orc_quantise(const uint32_t p0[256], const uint32_t p1[256], int mode512,
                  const uint32_t* pix, size_t n, uint16_t* entries);
/* E5  src/agmv_encode.c:302-352 */
uint8_t orc_compare_iframe_block(const uint32_t p0[256], const uint32_t p1[256], uint32_t w,
                                 uint32_t x, uint32_t y, uint32_t color, const uint16_t* entries);
/* E6  src/agmv_encode.c:240-300 */
uint8_t orc_compare_pframe_block(const uint32_t p0[256], const uint32_t p1[256], uint32_t w,
                                 uint32_t x, uint32_t y, const uint16_t* entries,
                                 const uint16_t* iframe_entries);
/* E7  src/agmv_encode.c:354-436; returns bytes written */
size_t orc_assemble_iframe(const uint32_t p0[256], const uint32_t p1[256], int mode512,
                           uint32_t w, uint32_t h, const uint16_t* entries, uint8_t* out);
/* E8  src/agmv_encode.c:438-527 */
size_t orc_assemble_pframe(const uint32_t p0[256], const uint32_t p1[256], int mode512,
                           uint32_t w, uint32_t h, const uint16_t* entries,
                           const uint16_t* iframe_entries, uint8_t* out);

/* E9  the stateful part of AGMV_EncodeFrame (src/agmv_encode.c:529-634) minus FILE + LZ:
   I/P choice by frame_count % 4, I-frame entry snapshot, frame_count++. */
typedef struct orc_encoder {
	uint32_t w, h;
	int mode512;
	uint32_t p0[256], p1[256];
	uint32_t frame_count;
	uint16_t* iframe_entries; /* w*h */
	uint16_t* scratch;        /* w*h */
} orc_encoder;

orc_encoder* orc_encoder_new(uint32_t w, uint32_t h, int mode512, const uint32_t* p0,
                             const uint32_t* p1, uint32_t first_frame_count);
void   orc_encoder_free(orc_encoder* e);
/* returns usize; `out` needs 33*w*h/16 bytes; `entries_out` (optional) gets the frame's plane */
size_t orc_encode_frame(orc_encoder* e, const uint32_t* pix, uint8_t* out, uint16_t* entries_out);

/* N1  host LZ stage, restated naively (brute-force window search like the reference).
   src/agmv_encode.c:106-177 (LZSS), :179-238 (LZ77), bit packer src/agmv_utils.c:86-112.
   Writes the bytes the reference writes to the file INCLUDING the flushed partial byte;
   *csize_field = (u32)(outbits / 8.0f) as stored in the chunk header. Returns bytes written. */
size_t orc_lzss_compress(const uint8_t* in, size_t n, uint8_t* out, uint32_t* csize_field);
size_t orc_lz77_compress(const uint8_t* in, size_t n, uint8_t* out, uint32_t* csize_field);

/* D1-D4  decoder with the reference's persistent state (src/agmv_decode.c:145-410).
   All buffers start zeroed (the frozen glibc behaviour). */
typedef struct orc_decoder {
	uint32_t w, h;
	int version;              /* 1..4, header byte 17 */
	uint32_t p0[256], p1[256];
	uint32_t frame_count;
	uint32_t* img;            /* w*h, persists across frames */
	uint32_t* iframe;         /* w*h */
	uint8_t* bitstream;       /* persistent decompression buffer (stale tail semantics) */
	size_t bitstream_cap;
	uint32_t bpos;            /* bitstream->pos after the last LZ stage */
} orc_decoder;

orc_decoder* orc_decoder_new(uint32_t w, uint32_t h, int version, const uint32_t* p0,
                             const uint32_t* p1);
void orc_decoder_free(orc_decoder* d);
/* D1: LZ stage. `payload` points just after the 16-byte chunk header; `avail` = bytes that
   can be read from there to the end of the file (the reference's bit reader runs past csize
   into the 0xFF guard / next chunk). Returns the number of payload bytes consumed. */
size_t orc_decoder_lz(orc_decoder* d, const uint8_t* payload, size_t avail, uint32_t usize,
                      uint32_t csize);
/* test hook: install an already-decompressed bitstream as if the LZ stage had produced it
   (bytes beyond n keep their stale content, exactly like the reference's persistent buffer) */
void orc_decoder_set_bitstream(orc_decoder* d, const uint8_t* bytes, uint32_t n);
/* D2/D3/D4: parse + reconstruct from d->bitstream[0..] with d->bpos, snapshot, frame_count++.
   If entry_offsets != NULL (w*h/16 u32), entry_offsets[k] receives `bitpos` at the moment
   block k is entered (before the flag resync) and *n_entered the number of blocks entered. */
void orc_decoder_parse(orc_decoder* d, uint32_t* entry_offsets, uint32_t* n_entered);

/* whole .agmv file from memory: header (src/agmv_decode.c:91-143), chunk scan
   (src/agmv_utils.c:140-166), frames.  Returns reference error code. */
typedef struct orc_file_info {
	uint32_t num_frames, w, h, fps, version, fmt;
	uint32_t total_audio_duration, sample_rate, audio_size, channels, bits_per_sample;
	size_t first_chunk;
} orc_file_info;
int orc_parse_header(const uint8_t* file, size_t len, orc_file_info* info, uint32_t* p0,
                     uint32_t* p1);
/* scan for 'AGFC' starting at pos the way AGMV_FindNextFrameChunk does; returns offset of the
   FourCC (or len if none). */
size_t orc_find_next_frame_chunk(const uint8_t* file, size_t len, size_t pos);

/* FNV-1a-64 over the little-endian bytes of 4-byte pixels (golden hashing helper) */
uint64_t orc_fnv1a64(const void* data, size_t nbytes, uint64_t seed);

/* N2  PDIFS midpoint, src/agmv_utils.c:949-969 */
void orc_interp_frame(uint32_t* out, const uint32_t* f1, const uint32_t* f2, size_t n);

#endif
