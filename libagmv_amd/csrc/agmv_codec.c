/*
 * libagmv_amd/csrc/agmv_codec.c -- libagmv's encode/decode entry points on top of the GPU hot path.
 *
 *   per frame   AGMV_EncodeFrame (reference src/agmv_encode.c:529-634) and AGMV_DecodeFrameChunk
 *               (src/agmv_decode.c:145-410): same FILE* protocol, one frame per call through the
 *               batch C-ABI of include/agmv_hip.h with n_frames = 1.
 *   sequences   AGMV_EncodeAGMV / AGMV_EncodeFullAGMV / AGMV_EncodeVideo (src/agmv_encode.c:719-4407,
 *               BMP branch) and AGMV_DecodeAGMV / AGMV_DecodeVideo (src/agmv_decode.c:455-647): these
 *               own the loop, so frames go to the GPU in GOP-aligned batches while the host threads
 *               run the LZ stage and the container is written strictly in frame order.
 *
 * Everything on the hot path (quantise, block classification, byte assembly, parse, reconstruct,
 * PDIFS midpoint, palette histogram) runs on the GPU.  There is no CPU fallback: if the GPU path
 * fails these functions print the reason and abort (the void encoders have no error channel).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "agmv_hip.h"
#include "agmv_internal.h"
#include "agmv_pipeline.h"

/* ------------------------------------------------------------------------------------------ */
static agmv_hip_ctx* g_ctx = NULL;
static uint32_t g_pal[512];
static int g_pal_mode = -1;
static unsigned g_batch_frames = 0, g_lz_threads = 0, g_devices = 0;
static unsigned long g_export_count = 0;            /* AGIDL's expcount, extern/agidl/src/agidl_img_export.c:18 */

void agmv_die(const char* what)
{
	fprintf(stderr, "libagmv(amd): %s: %s\n", what, agmv_hip_last_error());
	fprintf(stderr, "libagmv(amd): the AGMV hot path runs on the GPU only (no CPU fallback) -- aborting\n");
	abort();
}

/* the decoders have an error channel (enum Error): a GPU failure is reported on stderr AND returned, never papered over */
static int gpu_failed(const char* what)
{
	fprintf(stderr, "libagmv(amd): %s: %s (the AGMV hot path runs on the GPU only -- no CPU fallback)\n", what, agmv_hip_last_error());
	return MEMORY_CORRUPTION_ERR;
}

static int bad_geometry(uint32_t w, uint32_t h)
{
	return w == 0 || h == 0 || (w & 3u) || (h & 3u) || (unsigned long long)w * h > (1ull << 28);
}

void AGMV_SetBatchFrames(unsigned n) { g_batch_frames = n; }
void AGMV_SetLZThreads(unsigned n) { g_lz_threads = n; }
void AGMV_SetDevices(unsigned n) { g_devices = n; }

/* frames per GPU batch: what the caller asked for, else about 128 MB of source pixels (64 frames at most), whole GOPs */
static unsigned batch_frames(size_t npx)
{
	unsigned n = g_batch_frames;
	const char* e = getenv("AGMV_BATCH_FRAMES");
	if (!n && e) n = (unsigned)atoi(e);
	if (!n) {
		size_t f = ((size_t)128 << 20) / (npx ? npx * 4 : 4);
		n = f < 8 ? 8 : (f > 64 ? 64 : (unsigned)f);
	}
	return (n + 3u) & ~3u;
}

/* GPUs the sequence encoders spread their batches over (AGMV_SetDevices / env AGMV_DEVICES; default 1) */
static unsigned devices(void)
{
	unsigned n = g_devices;
	const char* e = getenv("AGMV_DEVICES");
	if (!n && e) n = (unsigned)atoi(e);
	return n ? n : 1;
}

/* host threads of the pipelines (BMP parse, LZ, BMP export): what the caller asked for, else the cores this process may
   use -- the online count, cut to the cgroup's CPU quota where there is one (a container's share of a big host) */
static unsigned lz_threads(void)
{
	unsigned n = g_lz_threads;
	const char* e = getenv("AGMV_LZ_THREADS");
	if (!n && e) n = (unsigned)atoi(e);
	if (!n) {
		long c = sysconf(_SC_NPROCESSORS_ONLN);
		FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
		n = c > 0 ? (unsigned)c : 1;
		if (f) {
			long long quota = 0, period = 0;
			if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
				unsigned q = (unsigned)((quota + period - 1) / period);
				if (q >= 1 && q < n) n = q;
			}
			fclose(f);
		}
	}
	return n > 64 ? 64 : n;
}

static agmv_hip_ctx* ctx(void)
{
	if (!g_ctx) {
		const char* e = getenv("AGMV_DEVICE");
		g_ctx = agmv_hip_create(e ? atoi(e) : 0);
		if (!g_ctx) agmv_die("cannot open the GPU");
	}
	return g_ctx;
}

static int mode512_of(AGMV_OPT opt) { return opt != AGMV_OPT_II && opt != AGMV_OPT_ANIM && opt != AGMV_OPT_GBA_II; }

/* (re)build the exact LUT when the palette of the object differs from the one on the device */
static void use_palette(const u32* p0, const u32* p1, int mode512)
{
	uint32_t pal[512];
	int i;
	for (i = 0; i < 256; i++) { pal[i] = (uint32_t)p0[i]; pal[256 + i] = mode512 ? (uint32_t)p1[i] : 0; }
	if (g_pal_mode == mode512 && memcmp(pal, g_pal, sizeof(pal)) == 0) { ctx(); return; }
	if (agmv_hip_set_palette(ctx(), pal, pal + 256, mode512, NULL)) agmv_die("palette upload");
	memcpy(g_pal, pal, sizeof(pal));
	g_pal_mode = mode512;
}

/* ------------------------------------------------------------------------------------------
 * container pieces
 * ------------------------------------------------------------------------------------------ */
/* header, reference src/agmv_encode.c:21-94 (palette1 only in the 512-colour versions) */
void AGMV_EncodeHeader(FILE* f, AGMV* a)
{
	AGMV_OPT opt = AGMV_GetOPT(a);
	int pals = mode512_of(opt) ? 2 : 1, p, i;
	AGMV_WriteFourCC(f, 'A', 'G', 'M', 'V');
	AGMV_WriteLong(f, AGMV_GetNumberOfFrames(a));
	AGMV_WriteLong(f, AGMV_GetWidth(a));
	AGMV_WriteLong(f, AGMV_GetHeight(a));
	AGMV_WriteByte(f, 1);
	AGMV_WriteByte(f, AGMV_GetVersionFromOPT(opt, AGMV_GetCompression(a)));
	AGMV_WriteLong(f, AGMV_GetFramesPerSecond(a));
	AGMV_WriteLong(f, AGMV_GetTotalAudioDuration(a));
	AGMV_WriteLong(f, AGMV_GetSampleRate(a));
	AGMV_WriteLong(f, AGMV_GetAudioSize(a));
	AGMV_WriteShort(f, AGMV_GetNumberOfChannels(a));
	AGMV_WriteShort(f, AGMV_GetBitsPerSample(a));
	for (p = 0; p < pals; p++)
		for (i = 0; i < 256; i++) {
			u32 c = p ? a->header.palette1[i] : a->header.palette0[i];
			AGMV_WriteByte(f, AGMV_GetR(c)); AGMV_WriteByte(f, AGMV_GetG(c)); AGMV_WriteByte(f, AGMV_GetB(c));
		}
}

/* reference src/agmv_decode.c:91-143 */
int AGMV_DecodeHeader(FILE* f, AGMV* a)
{
	int pals, p, i;
	AGMV_ReadFourCC(f, a->header.fourcc);
	a->header.num_of_frames = AGMV_ReadLong(f);
	a->header.width = AGMV_ReadLong(f);
	a->header.height = AGMV_ReadLong(f);
	a->header.fmt = AGMV_ReadByte(f);
	a->header.version = AGMV_ReadByte(f);
	a->header.frames_per_second = AGMV_ReadLong(f);
	a->header.total_audio_duration = AGMV_ReadLong(f);
	a->header.sample_rate = AGMV_ReadLong(f);
	a->header.audio_size = AGMV_ReadLong(f);
	a->header.num_of_channels = AGMV_ReadShort(f);
	a->header.bits_per_sample = AGMV_ReadShort(f);
	if (!AGMV_IsCorrectFourCC(a->header.fourcc, 'A', 'G', 'M', 'V') || a->header.version < 1 || a->header.version > 4 ||
	    a->header.frames_per_second >= 200 || !(a->header.bits_per_sample == 16 || a->header.bits_per_sample == 8))
		return INVALID_HEADER_FORMATTING_ERR;
	pals = (a->header.version == 1 || a->header.version == 3) ? 2 : 1;
	for (p = 0; p < pals; p++)
		for (i = 0; i < 256; i++) {
			u32 r = AGMV_ReadByte(f), g = AGMV_ReadByte(f), b = AGMV_ReadByte(f);
			/* AGIDL_RGB(r,g,b,fmt): the encoder always writes fmt 1 = RGB_888; 2 = BGR_888 */
			u32 c = a->header.fmt == 2 ? (b << 16 | g << 8 | r) : (r << 16 | g << 8 | b);
			if (p) a->header.palette1[i] = c; else a->header.palette0[i] = c;
		}
	return NO_ERR;
}

/* zero-length audio chunks are written after every frame by the AGMV drivers (reference
   src/agmv_encode.c:707-717); audio itself is out of scope: only pass-through of what the object holds */
void AGMV_EncodeAudioChunk(FILE* f, AGMV* a)
{
	u32 size = a->audio_chunk ? a->audio_chunk->size : 0, i;
	AGMV_WriteFourCC(f, 'A', 'G', 'A', 'C');
	AGMV_WriteLong(f, size);
	for (i = 0; i < size; i++)
		AGMV_WriteByte(f, a->audio_chunk->atsample ? a->audio_chunk->atsample[a->audio_track->start_point++] : 0);
}

/* audio payloads are skipped, not decoded (out of scope); the chunk framing is honoured so a file
   with audio decodes its video (reference src/agmv_decode.c:412-453) */
int AGMV_DecodeAudioChunk(FILE* f, AGMV* a)
{
	AGMV_ReadFourCC(f, a->audio_chunk->fourcc);
	a->audio_chunk->size = AGMV_ReadLong(f);
	if (!AGMV_IsCorrectFourCC(a->audio_chunk->fourcc, 'A', 'G', 'A', 'C')) return INVALID_HEADER_FORMATTING_ERR;
	fseek(f, (long)a->audio_chunk->size, SEEK_CUR);
	return NO_ERR;
}

/* chunk framing around an already compressed payload, reference src/agmv_encode.c:549-550,
   567-585, 622-624: 'AGFC', frame number, usize, csize, csize payload bytes, 8 x 0xFF.
   (the reference writes the flushed partial byte and then overwrites it with the first 0xFF) */
void agmv_write_frame_chunk(FILE* f, u32 frame_no, u32 usize, u32 csize, const u8* payload)
{
	static const u8 guard[8] = {0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff};
	AGMV_WriteFourCC(f, 'A', 'G', 'F', 'C');
	AGMV_WriteLong(f, frame_no);
	AGMV_WriteLong(f, usize);
	AGMV_WriteLong(f, csize);
	fwrite(payload, 1, csize, f);
	fwrite(guard, 1, 8, f);
}

/* ------------------------------------------------------------------------------------------
 * per-frame entry points
 * ------------------------------------------------------------------------------------------ */
void AGMV_EncodeFrame(FILE* file, AGMV* a, u32* img_data)
{
	const uint32_t w = (uint32_t)AGMV_GetWidth(a), h = (uint32_t)AGMV_GetHeight(a);
	const size_t npx = (size_t)w * h, stride = agmv_hip_max_usize(w, h, 1);
	const int m512 = mode512_of(AGMV_GetOPT(a));
	const int is_i = a->frame_count % 4 == 0;
	uint32_t* pix = (uint32_t*)malloc(npx * 4);
	uint16_t* ient = (uint16_t*)malloc(npx * 2);
	u8* bytes = (u8*)malloc(stride);
	u8* comp;
	uint32_t usize = 0;
	u32 csize;
	size_t i;

	use_palette(a->header.palette0, a->header.palette1, m512);
	for (i = 0; i < npx; i++) pix[i] = (uint32_t)img_data[i];          /* LP64: 8 -> 4 bytes per pixel */
	for (i = 0; i < npx; i++) ient[i] = (uint16_t)(a->iframe_entries[i].pal_num << 8 | a->iframe_entries[i].index);
	if (agmv_hip_encode_frames(ctx(), pix, 1, w, h, (uint32_t)a->frame_count, bytes, stride, &usize, ient))
		agmv_die("AGMV_EncodeFrame");

	AGMV_SyncFrameAndImage(a, img_data);                               /* :548 */
	if ((size_t)usize > a->bitstream->len) {                           /* objects built by foreign code: w*h*2 */
		a->bitstream->data = (u8*)realloc(a->bitstream->data, usize + 64);
		a->bitstream->len = usize + 64;
	}
	memcpy(a->bitstream->data, bytes, usize);
	a->bitstream->pos = usize;

	comp = (u8*)malloc((size_t)usize * 4 + 64);
	/* LZ77 peeks one byte past the stream (:222): hand it the byte the persistent buffer holds there */
	csize = AGMV_GetCompression(a) == AGMV_LZSS_COMPRESSION ? agmv_lzss_mem(a->bitstream->data, usize, comp)
	                                                        : agmv_lz77_mem(a->bitstream->data, usize, comp);
	agmv_write_frame_chunk(file, a->frame_count + 1, usize, csize, comp);

	if (is_i)                                                          /* :626-630 */
		for (i = 0; i < npx; i++) { a->iframe_entries[i].pal_num = (u8)(ient[i] >> 8); a->iframe_entries[i].index = (u8)ient[i]; }
	a->frame_count++;
	free(comp); free(bytes); free(ient); free(pix);
}

int AGMV_DecodeFrameChunk(FILE* file, AGMV* a)
{
	const uint32_t w = (uint32_t)a->frame->width, h = (uint32_t)a->frame->height;
	const size_t npx = (size_t)w * h;
	const int ver = a->header.version, m512 = (ver == 1 || ver == 3);
	uint32_t bpos = 0, *prev, *prev_i, *out;
	size_t i, cap, stride;
	u8* slab;

	if (bad_geometry(w, h)) return INVALID_HEADER_FORMATTING_ERR;      /* the block loops need multiples of 4 (src/agmv_decode.c:226-227) */
	a->bitstream->pos = 0;
	AGMV_ReadFourCC(file, a->frame_chunk->fourcc);
	a->frame_chunk->frame_num = AGMV_ReadLong(file);
	a->frame_chunk->uncompressed_size = AGMV_ReadLong(file);
	a->frame_chunk->compressed_size = AGMV_ReadLong(file);
	if (!AGMV_IsCorrectFourCC(a->frame_chunk->fourcc, 'A', 'G', 'F', 'C')) return INVALID_HEADER_FORMATTING_ERR;

	/* D1: LZ stage on the host straight from the FILE*, same bit reader protocol as the reference
	   (src/agmv_decode.c:171-222) so the file position ends where the reference's does */
	cap = a->bitstream->len;
	{
		u8* data = a->bitstream->data;
		const u32 usize = a->frame_chunk->uncompressed_size, csize = a->frame_chunk->compressed_size;
		unsigned long long bp = 0, lim = cap > 16 ? cap - 16 : 0;
		if (ver == 1 || ver == 2) {
			unsigned long long nbits = (unsigned long long)csize * 8, bits = 0;
			while (bits < nbits && bp < usize && bp < lim) {
				u32 flag = AGMV_ReadBits(file, 1);
				bits++;
				if (flag & 1) { data[bp++] = (u8)AGMV_ReadBits(file, 8); bits += 8; }
				else {
					u32 offset = AGMV_ReadBits(file, 16), len = AGMV_ReadBits(file, 4), k;
					unsigned long long pos = bp;
					bits += 20;
					for (k = 0; k < len; k++) {
						unsigned long long src = pos - offset + k;
						if (src < bp && bp < lim) data[bp++] = data[src];
					}
				}
			}
		} else {
			u32 t;
			for (t = 0; t < csize; t += 4) {
				u32 offset = AGMV_ReadShort(file), len = AGMV_ReadByte(file), k;
				u8 byte = AGMV_ReadByte(file);
				unsigned long long pos = bp;
				for (k = 0; k < len; k++) {
					unsigned long long src = pos - offset + k;
					if (src < bp && bp < lim) data[bp++] = data[src];
				}
				if (bp < lim) data[bp++] = byte;
			}
		}
		bpos = (uint32_t)bp;
	}
	a->bitstream->pos = bpos;
	AGMV_FlushReadBits();

	/* D2-D4 on the GPU: parse + reconstruct one frame on top of img_data / iframe */
	use_palette(a->header.palette0, a->header.palette1, m512);
	stride = ((size_t)bpos + 16 + 255) & ~(size_t)255;
	slab = (u8*)calloc(stride, 1);
	prev = (uint32_t*)malloc(npx * 4); prev_i = (uint32_t*)malloc(npx * 4); out = (uint32_t*)malloc(npx * 4);
	if (!slab || !prev || !prev_i || !out) { free(slab); free(prev); free(prev_i); free(out); return MEMORY_CORRUPTION_ERR; }
	memcpy(slab, a->bitstream->data, (size_t)bpos + 16 <= cap ? (size_t)bpos + 16 : cap);   /* incl. the stale bytes */
	for (i = 0; i < npx; i++) { prev[i] = (uint32_t)a->frame->img_data[i]; prev_i[i] = (uint32_t)a->iframe->img_data[i]; }
	if (agmv_hip_decode_frames(ctx(), slab, stride, &bpos, 1, w, h, (uint32_t)a->frame_count, out, prev, prev_i)) {
		free(slab); free(prev); free(prev_i); free(out);
		return gpu_failed("AGMV_DecodeFrameChunk");
	}
	for (i = 0; i < npx; i++) a->frame->img_data[i] = out[i];
	if (a->frame_count % 4 == 0) memcpy(a->iframe->img_data, a->frame->img_data, npx * sizeof(u32));   /* :401-405 */
	a->frame_count++;
	free(slab); free(prev); free(prev_i); free(out);
	return NO_ERR;
}

/* ------------------------------------------------------------------------------------------
 * helper entry points of the reference API that work on host AGMV_ENTRY planes.  They are steps
 * of AGMV_EncodeFrame; kept exported for source compatibility and routed through the same GPU
 * tables (exact LUT / bit matrix) so no second implementation of the hot path exists.
 * ------------------------------------------------------------------------------------------ */
static uint16_t gpu_entry_of(const u32* p0, const u32* p1, int m512, u32 color)
{
	/* the reference's own search, run on the GPU against the caller's palettes (no table is built for a single
	   colour; the device scratch lives in the context): src/agmv_utils.c:785-816, :851-895 */
	uint32_t pal[512], px = (uint32_t)color;
	uint16_t e = 0;
	int i;
	for (i = 0; i < 256; i++) { pal[i] = (uint32_t)p0[i]; pal[256 + i] = m512 ? (uint32_t)p1[i] : 0; }
	if (agmv_hip_nearest(ctx(), pal, pal + 256, m512, &px, 1, &e)) agmv_die("nearest entry");
	return e;
}

u8 AGMV_FindNearestColor(u32 palette[256], u32 color) { return (u8)gpu_entry_of(palette, palette, 0, color); }

AGMV_ENTRY AGMV_FindNearestEntry(u32 palette0[256], u32 palette1[256], u32 color)
{
	uint16_t e = gpu_entry_of(palette0, palette1, 1, color);
	AGMV_ENTRY r;
	memset(&r, 0, sizeof(r));
	r.pal_num = (u8)(e >> 8); r.index = (u8)e;
	return r;
}

static u32 entry_colour(AGMV* a, const AGMV_ENTRY* e) { return e->pal_num ? a->header.palette1[e->index] : a->header.palette0[e->index]; }

/* the two block predicates (reference src/agmv_encode.c:302-352, :240-300): the 16 palette colours of the block and the
   16 colours they are compared with go to the GPU, which counts the pairs within +-2 on every channel */
static u8 count_within2(const uint32_t* a, const uint32_t* b)
{
	int n = agmv_hip_within2_count(ctx(), a, b);
	if (n < 0) agmv_die("block compare");
	return (u8)n;
}

u8 AGMV_CompareIFrameBlock(AGMV* a, u32 x, u32 y, u32 color, AGMV_ENTRY* e)
{
	u32 w = a->frame->width, i, j;
	uint32_t ca[16], cb[16];
	for (j = 0; j < 4; j++) for (i = 0; i < 4; i++) { ca[j * 4 + i] = (uint32_t)entry_colour(a, &e[(x + i) + (y + j) * w]); cb[j * 4 + i] = (uint32_t)color; }
	return count_within2(ca, cb);
}

u8 AGMV_ComparePFrameBlock(AGMV* a, u32 x, u32 y, AGMV_ENTRY* e)
{
	u32 w = a->frame->width, i, j;
	uint32_t ca[16], cb[16];
	for (j = 0; j < 4; j++)
		for (i = 0; i < 4; i++) {
			size_t k = (x + i) + (size_t)(y + j) * w;
			ca[j * 4 + i] = (uint32_t)entry_colour(a, &e[k]);
			cb[j * 4 + i] = (uint32_t)entry_colour(a, &a->iframe_entries[k]);
		}
	return count_within2(ca, cb);
}

/* entries -> bitstream on the GPU: the entry plane goes to the encoder AS ENTRIES (agmv_hip_encode_entries: no
   quantisation), so classification and codes are exactly those of the given plane -- also for palettes with duplicate
   colours, where re-quantising an entry's colour would return the first of the duplicates
   (reference src/agmv_encode.c:354-436, :438-527; bytes are appended at bitstream->pos like the reference does) */
static void assemble_via_gpu(AGMV* a, AGMV_ENTRY* e, int iframe)
{
	const uint32_t w = (uint32_t)a->frame->width, h = (uint32_t)a->frame->height;
	const int m512 = mode512_of(AGMV_GetOPT(a));
	const size_t npx = (size_t)w * h, stride = agmv_hip_max_usize(w, h, 1);
	uint32_t* ent = (uint32_t*)malloc(npx * 4), usize = 0;
	uint16_t* ient = (uint16_t*)malloc(npx * 2);
	u8* bytes = (u8*)malloc(stride);
	size_t i;
	use_palette(a->header.palette0, a->header.palette1, m512);
	for (i = 0; i < npx; i++) {
		ent[i] = m512 ? (uint32_t)((e[i].pal_num & 1u) << 8 | e[i].index) : (uint32_t)e[i].index;
		ient[i] = (uint16_t)((a->iframe_entries[i].pal_num & 1u) << 8 | a->iframe_entries[i].index);
	}
	if (agmv_hip_encode_entries(ctx(), ent, 1, w, h, iframe ? 0u : 1u, bytes, stride, &usize, ient)) agmv_die("AGMV_Assemble*FrameBitstream");
	if ((size_t)a->bitstream->pos + usize > a->bitstream->len) {
		a->bitstream->len = a->bitstream->pos + usize + 64;
		a->bitstream->data = (u8*)realloc(a->bitstream->data, a->bitstream->len);
	}
	memcpy(a->bitstream->data + a->bitstream->pos, bytes, usize);
	a->bitstream->pos += usize;
	free(ent); free(ient); free(bytes);
}

void AGMV_AssembleIFrameBitstream(AGMV* a, AGMV_ENTRY* e) { assemble_via_gpu(a, e, 1); }
void AGMV_AssemblePFrameBitstream(AGMV* a, AGMV_ENTRY* e) { assemble_via_gpu(a, e, 0); }

/* ------------------------------------------------------------------------------------------
 * sequence encoders: the drivers decide WHICH frames are encoded (PDIFS schedules, frame skipping), the pipelined engine
 * of agmv_pipeline.c loads, encodes, compresses and writes them
 * ------------------------------------------------------------------------------------------ */
static int is_gba(AGMV_OPT o) { return o == AGMV_OPT_GBA_I || o == AGMV_OPT_GBA_II || o == AGMV_OPT_GBA_III; }

static agmv_seq* seq_open(AGMV* a, FILE* file, const char* dir, const char* base, AGMV_OPT opt, int audio_chunks, int use_interp)
{
	uint32_t pal[512];
	int i, sw = 0, sh = 0;
	const int m512 = mode512_of(opt);
	if (is_gba(opt)) { sw = AGMV_GBA_W; sh = AGMV_GBA_H; }
	if (opt == AGMV_OPT_NDS) { sw = AGMV_NDS_W; sh = AGMV_NDS_H; }
	for (i = 0; i < 256; i++) { pal[i] = (uint32_t)a->header.palette0[i]; pal[256 + i] = m512 ? (uint32_t)a->header.palette1[i] : 0; }
	return agmv_seq_open(a, file, dir, base, sw, sh, m512, AGMV_GetCompression(a) != AGMV_LZSS_COMPRESSION, audio_chunks, use_interp,
	                     batch_frames((size_t)AGMV_GetWidth(a) * AGMV_GetHeight(a)), devices(), lz_threads(), pal);
}

/* pass 1 of the palette build on the GPU (histogram), pick on the host */
static void build_palette_from_frames(const char* dir, const char* base, u32 start, u32 end, u32 size, AGMV_QUALITY quality,
                                      AGMV_OPT opt, u32* p0, u32* p1)
{
	uint32_t* hist = (uint32_t*)malloc(4u << 19);
	agmv_histogram_frames(ctx(), dir, base, start, end, size, (int)quality, lz_threads(), hist);
	AGMV_BuildPalette(hist, quality, opt, p0, p1);
	free(hist);
}

static void dump_gba_header(const char* filename)
{
	/* reference src/agmv_encode.c:3627-3656: the finished file as a C array in GBA_GEN_AGMV.h (CWD) */
	FILE *in = fopen(filename, "rb"), *out;
	long n, i;
	u8* data;
	if (!in) return;
	fseek(in, 0, SEEK_END); n = ftell(in); fseek(in, 0, SEEK_SET);
	data = (u8*)malloc((size_t)n);
	if (fread(data, 1, (size_t)n, in) != (size_t)n) { /* short read: dump what we have */ }
	fclose(in);
	out = fopen("GBA_GEN_AGMV.h", "w");
	fprintf(out, "#ifndef GBA_GEN_AGMV_H\n#define GBA_GEN_AGMV_H\n\nconst unsigned char GBA_AGMV_FILE[%ld] = {\n", n);
	for (i = 0; i < n; i++) {
		if (i != 0 && i % 4000 == 0) fprintf(out, "\n");
		fprintf(out, "%d,", data[i]);
	}
	fprintf(out, "};\n\n#endif");
	fclose(out);
	free(data);
}

static int heavy_pdifs(AGMV_OPT o) { return o == AGMV_OPT_I || o == AGMV_OPT_ANIM || o == AGMV_OPT_GBA_I || o == AGMV_OPT_GBA_II; }

static void resize_for_target(AGMV* a, AGMV_OPT opt)
{
	if (is_gba(opt)) { AGMV_SetWidth(a, AGMV_GBA_W); AGMV_SetHeight(a, AGMV_GBA_H); }
	if (opt == AGMV_OPT_NDS) { AGMV_SetWidth(a, AGMV_NDS_W); AGMV_SetHeight(a, AGMV_NDS_H); }
}

static void require_bmp(u8 img_type)
{
	if (img_type != AGMV_IMG_BMP) {
		fprintf(stderr, "libagmv(amd): only AGMV_IMG_BMP input is supported by this build (image type %u is out of scope)\n", img_type);
		abort();
	}
}

/* reference src/agmv_encode.c:2270-3657 (BMP branch) */
void AGMV_EncodeAGMV(AGMV* a, const char* filename, const char* dir, const char* basename, u8 img_type, u32 start_frame,
                     u32 end_frame, u32 width, u32 height, u32 frames_per_second, AGMV_OPT opt, AGMV_QUALITY quality,
                     AGMV_COMPRESSION compression)
{
	u32 p0[256], p1[256], adjusted = end_frame - start_frame, i;
	FILE* file;
	agmv_seq* s;
	u32 written;
	f32 rate;
	(void)frames_per_second;
	require_bmp(img_type);
	AGMV_SetOPT(a, opt);
	AGMV_SetCompression(a, compression);
	AGMV_SetLeniency(a, 0);
	/* :2296-2353: integer halves for the heavy modes, x0.75 in double (float for GBA_III) for the light ones */
	if (heavy_pdifs(opt)) adjusted /= 2;
	else if (opt == AGMV_OPT_GBA_III) adjusted = (u32)(adjusted * 0.75f);
	else adjusted = (u32)(adjusted * 0.75);
	resize_for_target(a, opt);

	build_palette_from_frames(dir, basename, start_frame, end_frame, width * height, quality, opt, p0, p1);
	if (a->audio_chunk) a->audio_chunk->size = (u32)(a->header.audio_size / (f32)adjusted);   /* 0 without an audio track */

	file = fopen(filename, "wb");
	if (!file) { fprintf(stderr, "libagmv(amd): cannot create %s\n", filename); abort(); }
	AGMV_SetICP0(a, p0);
	AGMV_SetICP1(a, p1);
	AGMV_EncodeHeader(file, a);

	s = seq_open(a, file, dir, basename, opt, 1, 1);
	for (i = start_frame; i <= end_frame;) {                  /* :2678, :3610-3612 */
		/* NDS is "light" only for BMP input (:2727 vs :2810) -- BMP is the only input here */
		if (!heavy_pdifs(opt)) { agmv_seq_push(s, i, -1); agmv_seq_push(s, i + 1, i + 2); agmv_seq_push(s, i + 3, -1); i += 4; }
		else { agmv_seq_push(s, i, i + 1); i += 2; }
		if (i + 4 >= end_frame) break;
	}
	written = agmv_seq_close(s);

	fseek(file, 4, SEEK_SET);                                 /* :3615-3620 */
	AGMV_WriteLong(file, written);
	fseek(file, 18, SEEK_SET);
	rate = (f32)adjusted / (AGMV_GetNumberOfFrames(a) + 1);
	AGMV_WriteLong(file, (u32)round(AGMV_GetFramesPerSecond(a) * rate));
	fclose(file);
	DestroyAGMV(a);                                           /* the callee frees the caller's object, :3625 */
	if (is_gba(opt)) dump_gba_header(filename);
}

/* reference src/agmv_encode.c:3659-4407 (BMP branch): every input frame, no PDIFS, no header patch */
void AGMV_EncodeFullAGMV(AGMV* a, const char* filename, const char* dir, const char* basename, u8 img_type, u32 start_frame,
                         u32 end_frame, u32 width, u32 height, u32 frames_per_second, AGMV_OPT opt, AGMV_QUALITY quality,
                         AGMV_COMPRESSION compression)
{
	u32 p0[256], p1[256], i;
	FILE* file;
	agmv_seq* s;
	(void)frames_per_second;
	require_bmp(img_type);
	AGMV_SetOPT(a, opt);
	AGMV_SetCompression(a, compression);
	resize_for_target(a, opt);
	build_palette_from_frames(dir, basename, start_frame, end_frame, width * height, quality, opt, p0, p1);
	if (a->audio_chunk) a->audio_chunk->size = (u32)(a->header.audio_size / (f32)AGMV_GetNumberOfFrames(a));
	file = fopen(filename, "wb");
	if (!file) { fprintf(stderr, "libagmv(amd): cannot create %s\n", filename); abort(); }
	AGMV_SetICP0(a, p0);
	AGMV_SetICP1(a, p1);
	AGMV_EncodeHeader(file, a);
	s = seq_open(a, file, dir, basename, opt, AGMV_GetTotalAudioDuration(a) != 0, 0);
	for (i = start_frame; i <= end_frame; i++) agmv_seq_push(s, i, -1);
	(void)agmv_seq_close(s);
	fclose(file);
	DestroyAGMV(a);
	if (is_gba(opt)) dump_gba_header(filename);
}

/* reference src/agmv_encode.c:719-2268 (BMP branch): frame skipping decided per group by the grey-equality
   ratio of the two middle (light) / first two (heavy) frames against the leniency */
void AGMV_EncodeVideo(const char* filename, const char* dir, const char* basename, u8 img_type, u32 start_frame, u32 end_frame,
                      u32 width, u32 height, u32 frames_per_second, AGMV_OPT opt, AGMV_QUALITY quality, AGMV_COMPRESSION compression)
{
	AGMV* a = CreateAGMV(end_frame - start_frame, width, height, frames_per_second);
	u32 p0[256], p1[256], i;
	FILE* file;
	agmv_seq* s;
	u32 written;
	int sw = 0, sh = 0;
	uint32_t ew, eh;
	f32 rate, len;
	size_t npx;
	u32 *fa, *fb;
	uint32_t* tmp;
	require_bmp(img_type);
	AGMV_SetOPT(a, opt);
	AGMV_SetCompression(a, compression);
	switch (opt) {                                            /* :744-790 */
	case AGMV_OPT_II: len = 0.1282f; break;
	case AGMV_OPT_GBA_I: case AGMV_OPT_GBA_II: case AGMV_OPT_GBA_III: case AGMV_OPT_NDS: len = 0.0f; break;
	default: len = 0.2282f; break;
	}
	AGMV_SetLeniency(a, len);
	resize_for_target(a, opt);
	build_palette_from_frames(dir, basename, start_frame, end_frame, width * height, quality, opt, p0, p1);
	file = fopen(filename, "wb");
	if (!file) { fprintf(stderr, "libagmv(amd): cannot create %s\n", filename); abort(); }
	AGMV_SetICP0(a, p0);
	AGMV_SetICP1(a, p1);
	AGMV_EncodeHeader(file, a);
	s = seq_open(a, file, dir, basename, opt, 0, 1);
	if (is_gba(opt)) { sw = AGMV_GBA_W; sh = AGMV_GBA_H; }
	if (opt == AGMV_OPT_NDS) { sw = AGMV_NDS_W; sh = AGMV_NDS_H; }
	ew = (uint32_t)AGMV_GetWidth(a); eh = (uint32_t)AGMV_GetHeight(a);
	npx = (size_t)ew * eh;
	fa = (u32*)malloc(npx * sizeof(u32)); fb = (u32*)malloc(npx * sizeof(u32)); tmp = (uint32_t*)malloc(npx * 4);
	for (i = start_frame; i <= end_frame;) {
		long x = heavy_pdifs(opt) ? (long)i : (long)i + 1;   /* pair whose similarity decides */
		size_t k;
		f32 ratio;
		agmv_load_source(dir, basename, x, sw, sh, ew, eh, tmp); for (k = 0; k < npx; k++) fa[k] = tmp[k];
		agmv_load_source(dir, basename, x + 1, sw, sh, ew, eh, tmp); for (k = 0; k < npx; k++) fb[k] = tmp[k];
		ratio = AGMV_CompareFrameSimilarity(fa, fb, ew, eh);
		if (ratio >= AGMV_GetLeniency(a)) {
			if (!heavy_pdifs(opt)) { agmv_seq_push(s, i, -1); agmv_seq_push(s, i + 1, i + 2); agmv_seq_push(s, i + 3, -1); i += 4; }
			else { agmv_seq_push(s, i, i + 1); i += 2; }
		} else { agmv_seq_push(s, i, -1); i += 1; }
		if (i + 4 >= end_frame) break;
	}
	written = agmv_seq_close(s);
	free(fa); free(fb); free(tmp);
	fseek(file, 4, SEEK_SET);                                 /* :2225-2231 */
	AGMV_WriteLong(file, written);
	fseek(file, 18, SEEK_SET);
	rate = (f32)written / AGMV_GetNumberOfFrames(a);
	AGMV_WriteLong(file, (u32)round(AGMV_GetFramesPerSecond(a) * rate));
	fclose(file);
	DestroyAGMV(a);
	if (is_gba(opt)) dump_gba_header(filename);
}

/* ------------------------------------------------------------------------------------------
 * sequence decoders (reference src/agmv_decode.c:455-647): host does the chunk scan and the LZ stage
 * frame by frame into ONE persistent buffer (so the stale-tail semantics hold), the GPU parses and
 * reconstructs whole batches, frames are exported as quick_export_<n>.bmp in the CWD.
 * ------------------------------------------------------------------------------------------ */
static int decode_file(const char* filename, u8 img_type)
{
	FILE* f = fopen(filename, "rb");
	AGMV hdr_obj;
	u8* file;
	long flen;
	size_t pos, got, npx;
	uint32_t w, h;
	unsigned cap;
	int err, m512;
	agmv_hip_ctx* c;
	uint32_t pal[512];
	int i;
	if (!f) return FILE_NOT_FOUND_ERR;
	if (img_type != AGMV_IMG_BMP) { fclose(f); require_bmp(img_type); }
	memset(&hdr_obj, 0, sizeof(hdr_obj.header));
	err = AGMV_DecodeHeader(f, &hdr_obj);
	if (err != NO_ERR) { fclose(f); return err; }
	pos = (size_t)ftell(f);
	fseek(f, 0, SEEK_END); flen = ftell(f); fseek(f, 0, SEEK_SET);
	file = (u8*)malloc((size_t)flen + 16);
	if (!file) { fclose(f); return MEMORY_CORRUPTION_ERR; }
	got = fread(file, 1, (size_t)flen, f);
	fclose(f);
	memset(file + got, 0, 16);
	w = (uint32_t)hdr_obj.header.width; h = (uint32_t)hdr_obj.header.height;
	if (bad_geometry(w, h)) { free(file); return INVALID_HEADER_FORMATTING_ERR; }
	npx = (size_t)w * h;
	m512 = hdr_obj.header.version == 1 || hdr_obj.header.version == 3;
	if (!g_ctx) { const char* e = getenv("AGMV_DEVICE"); g_ctx = agmv_hip_create(e ? atoi(e) : 0); }
	c = g_ctx;
	if (!c) { free(file); return gpu_failed("cannot open the GPU"); }
	for (i = 0; i < 256; i++) { pal[i] = (uint32_t)hdr_obj.header.palette0[i]; pal[256 + i] = m512 ? (uint32_t)hdr_obj.header.palette1[i] : 0; }
	if (!(g_pal_mode == m512 && memcmp(pal, g_pal, sizeof(pal)) == 0)) {
		if (agmv_hip_set_palette(c, pal, pal + 256, m512, NULL) || agmv_hip_sync()) { free(file); return gpu_failed("palette upload"); }
		memcpy(g_pal, pal, sizeof(pal));
		g_pal_mode = m512;
	}
	cap = batch_frames(npx);
	err = agmv_decode_stream(c, file, got, pos, w, h, (uint32_t)hdr_obj.header.num_of_frames, hdr_obj.header.version,
	                         hdr_obj.header.total_audio_duration != 0, cap, lz_threads(), &g_export_count);
	free(file);
	return err;
}

int AGMV_DecodeVideo(const char* filename, u8 img_type) { return decode_file(filename, img_type); }

/* audio export (quick_export.wav / .aiff) is out of scope of this build; the video frames are exported
   exactly like the reference does */
int AGMV_DecodeAGMV(const char* filename, u8 img_type, AGMV_AUDIO_TYPE audio_type)
{
	(void)audio_type;
	return decode_file(filename, img_type);
}

/* ------------------------------------------------------------------------------------------
 * playback helpers, restated from reference src/agmv_playback.c:18-115.  They keep the reference's
 * bookkeeping exactly: offset_table[] is filled as frames are played or skipped over (not only by
 * AGMV_ParseAGMV), seeks are relative to frame_count, and the reset offset is 1574 for container
 * version 1 ONLY (a 512-colour LZ77 file, version 3, is sought to 806 and the next chunk scan walks
 * over its second palette -- that is what the reference does, src/agmv_playback.c:19-24).
 * The one deviation is in undefined territory: indices outside offset_table[MAX_OFFSET_TABLE] are
 * not written / read (the reference has no bound), and a backwards skip past frame 0 lands on frame 0
 * (the reference stores the negative count into the unsigned field, :86-92).
 * ------------------------------------------------------------------------------------------ */
void AGMV_ResetVideo(FILE* f, AGMV* a)
{
	fseek(f, AGMV_GetVersion(a) == 1 ? 1574 : 806, SEEK_SET);     /* :18-26 */
	a->frame_count = 0;
}

Bool AGMV_IsVideoDone(AGMV* a) { return a->frame_count >= AGMV_GetNumberOfFrames(a) ? TRUE : FALSE; }   /* :28-33 */

static void note_offset(FILE* f, AGMV* a)
{
	if (a->frame_count < MAX_OFFSET_TABLE) a->offset_table[a->frame_count] = (u32)ftell(f);
}

/* :35-83: walk forward over NextIFrame(n, frame_count) frame chunks, recording where each one starts */
static void skip_forwards(FILE* f, AGMV* a, int n, int decode_audio)
{
	const int audio = AGMV_GetTotalAudioDuration(a) != 0;
	int i;
	n = AGMV_NextIFrame(n, (int)a->frame_count);
	for (i = 0; i < n; i++) {
		AGMV_FindNextFrameChunk(f);
		note_offset(f, a);
		a->frame_count++;
		AGMV_SkipFrameChunk(f);
		if (audio) {
			AGMV_FindNextAudioChunk(f);
			if (decode_audio) AGMV_DecodeAudioChunk(f, a); else AGMV_SkipAudioChunk(f);
		}
	}
}

void AGMV_SkipForwards(FILE* f, AGMV* a, int n) { skip_forwards(f, a, n, 0); }
void AGMV_SkipForwardsAndDecodeAudio(FILE* f, AGMV* a, int n) { skip_forwards(f, a, n, 1); }

void AGMV_SkipBackwards(FILE* f, AGMV* a, int n)                 /* :85-94 */
{
	int fc = (int)a->frame_count;
	n = AGMV_PrevIFrame(n, fc);
	fc -= n;
	if (fc < 0) fc = 0;
	a->frame_count = (u32)fc;
	if (fc < MAX_OFFSET_TABLE) fseek(f, (long)a->offset_table[fc], SEEK_SET);
}

/* :96-104 ("only call after all frames have been read": offset_table must hold entry n) */
void AGMV_SkipTo(FILE* f, AGMV* a, int n)
{
	n = AGMV_SkipToNearestIFrame(n);
	if (n >= 0 && (u32)n < AGMV_GetNumberOfFrames(a) && n < MAX_OFFSET_TABLE) {
		fseek(f, (long)a->offset_table[n], SEEK_SET);
		a->frame_count = (u32)n;
	}
}

void AGMV_PlayAGMV(FILE* f, AGMV* a)                            /* :106-119 */
{
	AGMV_FindNextFrameChunk(f);
	note_offset(f, a);
	AGMV_DecodeFrameChunk(f, a);
	if (AGMV_GetTotalAudioDuration(a) != 0) { AGMV_FindNextAudioChunk(f); AGMV_SkipAudioChunk(f); }
}
