/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin accessors that are compiled TOGETHER WITH the unmodified reference
 * sources (from where they lie under /root/reference, see oracle/Makefile) into
 * oracle/_ref/libagmv_ref.so.  The shim contains no codec logic of its own: it
 * only marshals 4-byte test buffers into the reference's LP64 `u32`
 * (= unsigned long, include/agmv_defines.h:22) arrays and calls the reference's
 * exported functions:
 *
 *   AGMV_FindNearestEntry / AGMV_FindNearestColor   src/agmv_utils.c:785-895
 *   AGMV_AssembleIFrameBitstream / ...PFrame...      src/agmv_encode.c:354-527
 *   AGMV_EncodeFrame                                  src/agmv_encode.c:529-634
 *   AGMV_DecodeHeader / AGMV_DecodeFrameChunk         src/agmv_decode.c:91-410
 *   AGMV_LZSS / AGMV_LZ77                             src/agmv_encode.c:106-238
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the resulting library.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stddef.h>

#include <agmv.h>

u32 AGMV_LZ77(FILE* file, AGMV_BITSTREAM* in); /* defined in agmv_encode.c, not in the header */

size_t refshim_sizeof_agmv(void) { return sizeof(AGMV); }
size_t refshim_sizeof_entry(void) { return sizeof(AGMV_ENTRY); }
size_t refshim_sizeof_u32(void) { return sizeof(u32); }

/* ---- palette / mode plumbing ------------------------------------------------ */

AGMV* refshim_create(uint32_t w, uint32_t h, int opt, int compression,
                     const uint32_t* p0, const uint32_t* p1)
{
	int i;
	AGMV* a = CreateAGMV(1, w, h, 24);
	/* worst case of the v1 coder is 33 bytes per 16 px; CreateAGMV allocates 2 B/px
	   (src/agmv_utils.c:338) which noisy frames overflow -- give the reference room. */
	free(a->bitstream->data);
	a->bitstream->len = (u32)w * h * 3 + 64;
	a->bitstream->data = (u8*)calloc(a->bitstream->len, 1);
	AGMV_SetOPT(a, (AGMV_OPT)opt);
	AGMV_SetCompression(a, (AGMV_COMPRESSION)compression);
	for (i = 0; i < 256; i++) {
		a->header.palette0[i] = p0[i];
		a->header.palette1[i] = p1 ? p1[i] : 0;
	}
	return a;
}

void refshim_destroy(AGMV* a) { DestroyAGMV(a); }

void refshim_set_frame_count(AGMV* a, uint32_t fc) { a->frame_count = fc; }
uint32_t refshim_get_frame_count(AGMV* a) { return (uint32_t)a->frame_count; }

/* ---- E2/E3: nearest colour / entry ------------------------------------------- */

void refshim_nearest_entries(const uint32_t* p0, const uint32_t* p1, int mode512,
                             const uint32_t* pix, size_t n, uint16_t* out)
{
	u32 P0[256], P1[256];
	size_t i;
	for (i = 0; i < 256; i++) { P0[i] = p0[i]; P1[i] = p1 ? p1[i] : 0; }
	for (i = 0; i < n; i++) {
		if (mode512) {
			AGMV_ENTRY e = AGMV_FindNearestEntry(P0, P1, pix[i]);
			out[i] = (uint16_t)(e.pal_num << 8 | e.index);
		} else {
			out[i] = AGMV_FindNearestColor(P0, pix[i]);
		}
	}
}

/* ---- E4-E8: loops A + B of AGMV_EncodeFrame without the FILE + LZ stage -------- */
/* Mirrors src/agmv_encode.c:552-565 / 589-599 and :626-630 by CALLING the
   reference's own functions; returns usize, copies the pre-LZ bitstream and the
   frame's entry plane out. */
size_t refshim_encode_frame_hot(AGMV* a, const uint32_t* pix, uint8_t* out_bytes,
                                uint16_t* out_entries)
{
	AGMV_OPT opt = AGMV_GetOPT(a);
	size_t size = (size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a), i;
	AGMV_ENTRY* img_entry = (AGMV_ENTRY*)malloc(sizeof(AGMV_ENTRY) * size);
	int mode512 = (opt != AGMV_OPT_II && opt != AGMV_OPT_ANIM && opt != AGMV_OPT_GBA_II);

	a->bitstream->pos = 0;
	for (i = 0; i < size; i++) {
		if (mode512) {
			img_entry[i] = AGMV_FindNearestEntry(a->header.palette0, a->header.palette1, pix[i]);
		} else {
			img_entry[i].index = AGMV_FindNearestColor(a->header.palette0, pix[i]);
			img_entry[i].pal_num = 0;
		}
	}
	if (a->frame_count % 4 == 0) AGMV_AssembleIFrameBitstream(a, img_entry);
	else                         AGMV_AssemblePFrameBitstream(a, img_entry);

	if (out_bytes) memcpy(out_bytes, a->bitstream->data, a->bitstream->pos);
	if (out_entries)
		for (i = 0; i < size; i++)
			out_entries[i] = (uint16_t)(img_entry[i].pal_num << 8 | img_entry[i].index);
	if (a->frame_count % 4 == 0)
		for (i = 0; i < size; i++) a->iframe_entries[i] = img_entry[i];
	free(img_entry);
	a->frame_count++;
	return a->bitstream->pos;
}

/* Time-able split of the above (cpu_baseline "reference" leg): quantise only. */
void refshim_quantise_only(AGMV* a, const uint32_t* pix, size_t n)
{
	size_t i; volatile unsigned sink = 0;
	for (i = 0; i < n; i++) {
		AGMV_ENTRY e = AGMV_FindNearestEntry(a->header.palette0, a->header.palette1, pix[i]);
		sink += e.index;
	}
	(void)sink;
}

/* ---- the real AGMV_EncodeFrame through a FILE* -------------------------------- */
/* pixels are widened to the reference's 8-byte u32. Appends the chunk to `path`. */
void refshim_encode_frame_file(AGMV* a, const char* path, const uint32_t* pix, int truncate)
{
	size_t size = (size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a), i;
	u32* wide = (u32*)malloc(sizeof(u32) * size);
	FILE* f = fopen(path, truncate ? "wb+" : "rb+");
	if (!truncate) fseek(f, 0, SEEK_END);
	for (i = 0; i < size; i++) wide[i] = pix[i];
	AGMV_EncodeFrame(f, a, wide);
	fclose(f);
	free(wide);
}

void refshim_write_header(AGMV* a, const char* path)
{
	FILE* f = fopen(path, "wb");
	AGMV_EncodeHeader(f, a);
	fclose(f);
}

/* ---- LZ stage alone (N1) ------------------------------------------------------ */
/* Runs AGMV_LZSS / AGMV_LZ77 + AGMV_FlushWriteBits on a byte buffer, returns the
   bytes written to the file and the csize the reference would store. */
size_t refshim_lz(const uint8_t* in, size_t n, int compression, uint8_t* out, size_t out_cap,
                  uint32_t* csize)
{
	AGMV_BITSTREAM bs;
	FILE* f = tmpfile();
	size_t got;
	bs.data = (u8*)malloc(n + 64);
	memset(bs.data, 0, n + 64);
	memcpy(bs.data, in, n);
	bs.len = n; bs.pos = n;
	if (compression == AGMV_LZSS_COMPRESSION) *csize = (uint32_t)AGMV_LZSS(f, &bs);
	else                                      *csize = (uint32_t)AGMV_LZ77(f, &bs);
	AGMV_FlushWriteBits(f);
	fflush(f);
	fseek(f, 0, SEEK_SET);
	got = fread(out, 1, out_cap, f);
	fclose(f);
	free(bs.data);
	return got;
}

/* ---- decode: header + frame loop of AGMV_DecodeAGMV without the BMP export ---- */
/* Follows src/agmv_decode.c:527-639 (video branch) but allocates zeroed buffers --
   the behaviour the reference gets from fresh glibc mmap pages -- and
   hands every decoded frame (low 32 bits of each img_data word) to the caller. */
typedef struct refshim_decoder {
	AGMV* a;
	FILE* f;
	int has_audio;
} refshim_decoder;

refshim_decoder* refshim_decoder_open(const char* path, int* err, uint32_t* w, uint32_t* h,
                                      uint32_t* nframes, uint32_t* version)
{
	refshim_decoder* d = (refshim_decoder*)calloc(1, sizeof(*d));
	AGMV* a = (AGMV*)calloc(1, sizeof(AGMV));
	size_t npx;
	a->frame_chunk = (AGMV_FRAME_CHUNK*)calloc(1, sizeof(AGMV_FRAME_CHUNK));
	a->audio_chunk = (AGMV_AUDIO_CHUNK*)calloc(1, sizeof(AGMV_AUDIO_CHUNK));
	a->bitstream = (AGMV_BITSTREAM*)calloc(1, sizeof(AGMV_BITSTREAM));
	a->frame = (AGMV_FRAME*)calloc(1, sizeof(AGMV_FRAME));
	a->iframe = (AGMV_FRAME*)calloc(1, sizeof(AGMV_FRAME));
	a->audio_track = (AGMV_AUDIO_TRACK*)calloc(1, sizeof(AGMV_AUDIO_TRACK));
	d->a = a;
	d->f = fopen(path, "rb");
	if (!d->f) { *err = FILE_NOT_FOUND_ERR; return d; }
	*err = AGMV_DecodeHeader(d->f, a);
	if (*err != NO_ERR) return d;
	a->frame->width = a->iframe->width = a->header.width;
	a->frame->height = a->iframe->height = a->header.height;
	npx = (size_t)a->header.width * a->header.height;
	a->frame->img_data = (u32*)calloc(npx, sizeof(u32));
	a->iframe->img_data = (u32*)calloc(npx, sizeof(u32));
	a->bitstream->len = npx * 3 + 64;
	a->bitstream->data = (u8*)calloc(a->bitstream->len, 1);
	d->has_audio = a->header.total_audio_duration != 0;
	*w = a->header.width; *h = a->header.height;
	*nframes = a->header.num_of_frames; *version = a->header.version;
	return d;
}

void refshim_decoder_palettes(refshim_decoder* d, uint32_t* p0, uint32_t* p1)
{
	int i;
	for (i = 0; i < 256; i++) { p0[i] = d->a->header.palette0[i]; p1[i] = d->a->header.palette1[i]; }
}

/* decode the next frame; returns the reference's error code; also reports the chunk
   header fields and bpos (= bitstream->pos after the LZ stage). */
int refshim_decoder_next(refshim_decoder* d, uint32_t* out_pix, uint32_t* usize, uint32_t* csize,
                         uint32_t* bpos)
{
	size_t npx = (size_t)d->a->header.width * d->a->header.height, i;
	int err;
	AGMV_FindNextFrameChunk(d->f);
	err = AGMV_DecodeFrameChunk(d->f, d->a);
	if (usize) *usize = d->a->frame_chunk->uncompressed_size;
	if (csize) *csize = d->a->frame_chunk->compressed_size;
	if (bpos) *bpos = d->a->bitstream->pos;
	if (out_pix) for (i = 0; i < npx; i++) out_pix[i] = (uint32_t)d->a->frame->img_data[i];
	return err;
}

/* copy of the decompressed (pre-parse) bitstream buffer incl. stale tail */
void refshim_decoder_bitstream(refshim_decoder* d, uint8_t* out, size_t n)
{
	memcpy(out, d->a->bitstream->data, n);
}

void refshim_decoder_close(refshim_decoder* d)
{
	if (d->f) fclose(d->f);
	if (d->a) {
		if (d->a->frame) { free(d->a->frame->img_data); free(d->a->frame); }
		if (d->a->iframe) { free(d->a->iframe->img_data); free(d->a->iframe); }
		if (d->a->bitstream) { free(d->a->bitstream->data); free(d->a->bitstream); }
		free(d->a->frame_chunk); free(d->a->audio_chunk); free(d->a->audio_track);
		free(d->a);
	}
	free(d);
}

/* PDIFS helper (N2): AGMV_InterpFrame on 4-byte buffers. */
void refshim_interp(uint32_t* out, const uint32_t* f1, const uint32_t* f2, uint32_t w, uint32_t h)
{
	size_t n = (size_t)w * h, i;
	u32 *a = (u32*)malloc(n * sizeof(u32)), *b = (u32*)malloc(n * sizeof(u32)),
	    *o = (u32*)malloc(n * sizeof(u32));
	for (i = 0; i < n; i++) { a[i] = f1[i]; b[i] = f2[i]; }
	AGMV_InterpFrame(o, a, b, w, h);
	for (i = 0; i < n; i++) out[i] = (uint32_t)o[i];
	free(a); free(b); free(o);
}

/* ---- E5-E8 on a GIVEN entry plane (the exported helpers, src/agmv_encode.c:240-527) ---------- */
/* entries are pal_num << 8 | index; the plane is widened to AGMV_ENTRY and handed to the reference's
   own AGMV_Assemble{I,P}FrameBitstream; returns the bytes appended at bitstream->pos = 0. */
static AGMV_ENTRY* widen_entries(const uint16_t* e, size_t n)
{
	AGMV_ENTRY* w = (AGMV_ENTRY*)calloc(n, sizeof(AGMV_ENTRY));
	size_t i;
	for (i = 0; i < n; i++) { w[i].pal_num = (u8)(e[i] >> 8); w[i].index = (u8)e[i]; }
	return w;
}

void refshim_set_iframe_entries(AGMV* a, const uint16_t* e)
{
	size_t n = (size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a), i;
	for (i = 0; i < n; i++) { a->iframe_entries[i].pal_num = (u8)(e[i] >> 8); a->iframe_entries[i].index = (u8)e[i]; }
}

size_t refshim_assemble(AGMV* a, const uint16_t* entries, int iframe, uint8_t* out)
{
	size_t n = (size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a);
	AGMV_ENTRY* w = widen_entries(entries, n);
	a->bitstream->pos = 0;
	if (iframe) AGMV_AssembleIFrameBitstream(a, w); else AGMV_AssemblePFrameBitstream(a, w);
	memcpy(out, a->bitstream->data, a->bitstream->pos);
	free(w);
	return a->bitstream->pos;
}

unsigned refshim_compare_i(AGMV* a, uint32_t x, uint32_t y, uint32_t color, const uint16_t* entries)
{
	size_t n = (size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a);
	AGMV_ENTRY* w = widen_entries(entries, n);
	unsigned c = AGMV_CompareIFrameBlock(a, x, y, color, w);
	free(w);
	return c;
}

unsigned refshim_compare_p(AGMV* a, uint32_t x, uint32_t y, const uint16_t* entries)
{
	size_t n = (size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a);
	AGMV_ENTRY* w = widen_entries(entries, n);
	unsigned c = AGMV_ComparePFrameBlock(a, x, y, w);
	free(w);
	return c;
}

/* ---- N4: field access for the playback tests (same struct layout in both libraries) ------------ */
uint32_t refshim_offset_table(AGMV* a, uint32_t i) { return (uint32_t)a->offset_table[i]; }
void refshim_frame_pixels(AGMV* a, uint32_t* out)
{
	size_t n = (size_t)a->frame->width * a->frame->height, i;
	for (i = 0; i < n; i++) out[i] = (uint32_t)a->frame->img_data[i];
}
/* a decoder object the way a player sets one up (tools/agmvp: CreateAGMV, then AGMV_DecodeHeader on the open file) */
size_t refshim_offsetof_frame_count(void) { return offsetof(AGMV, frame_count); }
size_t refshim_offsetof_offset_table(void) { return offsetof(AGMV, offset_table); }
