/*
 * libagmv_amd/csrc/agmv_host.c -- host side of the libagmv-compatible API: object lifecycle,
 * attribute accessors, FILE* byte/bit I/O, chunk scanning and the small colour utilities.
 * Plain host C, none of it on the hot path.  Behaviour follows the reference (cited per
 * function as file:line under /root/reference); the code is written from that behaviour.
 */
#include <stdlib.h>
#include <string.h>

#include "agmv.h"

/* ------------------------------------------------------------------------------------------
 * lifecycle (reference src/agmv_utils.c:332-421)
 * ------------------------------------------------------------------------------------------ */
AGMV* CreateAGMV(u32 num_of_frames, u32 width, u32 height, u32 frames_per_second)
{
	AGMV* a = (AGMV*)calloc(1, sizeof(AGMV));
	size_t npx = (size_t)width * height;
	a->frame_chunk = (AGMV_FRAME_CHUNK*)calloc(1, sizeof(AGMV_FRAME_CHUNK));
	a->audio_chunk = (AGMV_AUDIO_CHUNK*)calloc(1, sizeof(AGMV_AUDIO_CHUNK));
	a->bitstream = (AGMV_BITSTREAM*)calloc(1, sizeof(AGMV_BITSTREAM));
	/* the reference allocates w*h*2 bytes (src/agmv_utils.c:338); a 512-colour NORMAL block takes up
	   to 33 bytes per 16 px, so noisy frames overflow that.  Allocate the true worst case. */
	a->bitstream->len = npx * 33 / 16 + 64;
	a->bitstream->pos = 0;
	a->bitstream->data = (u8*)calloc(a->bitstream->len, 1);
	a->frame = (AGMV_FRAME*)calloc(1, sizeof(AGMV_FRAME));
	a->frame->img_data = (u32*)calloc(npx, sizeof(u32));
	a->iframe = (AGMV_FRAME*)calloc(1, sizeof(AGMV_FRAME));
	a->iframe->img_data = (u32*)calloc(npx, sizeof(u32));
	a->audio_track = (AGMV_AUDIO_TRACK*)calloc(1, sizeof(AGMV_AUDIO_TRACK));
	a->iframe_entries = (AGMV_ENTRY*)calloc(npx, sizeof(AGMV_ENTRY));

	a->frame_count = 0;
	AGMV_SetWidth(a, width);
	AGMV_SetHeight(a, height);
	AGMV_SetNumberOfFrames(a, num_of_frames);
	AGMV_SetFramesPerSecond(a, frames_per_second);
	AGMV_SetLeniency(a, 0.1282f);
	AGMV_SetOPT(a, AGMV_OPT_I);
	AGMV_SetCompression(a, AGMV_LZSS_COMPRESSION);
	AGMV_SetVolume(a, 1.0f);
	AGMV_SetBitsPerSample(a, 16);
	return a;
}

void DestroyAGMV(AGMV* a)
{
	if (!a) return;
	free(a->iframe_entries);
	if (a->frame) { free(a->frame->img_data); free(a->frame); }
	if (a->iframe) { free(a->iframe->img_data); free(a->iframe); }
	if (a->bitstream) { free(a->bitstream->data); free(a->bitstream); }
	free(a->frame_chunk);
	if (a->audio_track) {
		if (a->header.total_audio_duration != 0) {
			if (a->header.bits_per_sample == 16) free(a->audio_track->pcm);
			else free(a->audio_track->pcm8);
		}
		free(a->audio_track);
	}
	if (a->audio_chunk) {
		if (a->header.total_audio_duration != 0) free(a->audio_chunk->atsample);
		free(a->audio_chunk);
	}
	free(a);
}

/* ------------------------------------------------------------------------------------------
 * attributes (reference src/agmv_utils.c:247-330, 425-483)
 * ------------------------------------------------------------------------------------------ */
void AGMV_SetWidth(AGMV* a, u32 v) { a->header.width = v; a->frame->width = v; a->iframe->width = v; }
void AGMV_SetHeight(AGMV* a, u32 v) { a->header.height = v; a->frame->height = v; a->iframe->height = v; }
void AGMV_SetICP0(AGMV* a, u32 p[256]) { memcpy(a->header.palette0, p, sizeof(a->header.palette0)); }
void AGMV_SetICP1(AGMV* a, u32 p[256]) { memcpy(a->header.palette1, p, sizeof(a->header.palette1)); }
void AGMV_SetFramesPerSecond(AGMV* a, u32 v) { a->header.frames_per_second = v; }
void AGMV_SetNumberOfFrames(AGMV* a, u32 v) { a->header.num_of_frames = v; }
void AGMV_SetTotalAudioDuration(AGMV* a, u32 v) { a->header.total_audio_duration = v; }
void AGMV_SetSampleRate(AGMV* a, u32 v) { a->header.sample_rate = v; }
void AGMV_SetNumberOfChannels(AGMV* a, u8 v) { a->header.num_of_channels = v; }
void AGMV_SetAudioSize(AGMV* a, u32 v) { a->header.audio_size = v; }
void AGMV_SetLeniency(AGMV* a, f32 v) { a->leniency = v; }
void AGMV_SetOPT(AGMV* a, AGMV_OPT v) { a->opt = v; }
void AGMV_SetVersion(AGMV* a, u8 v) { a->header.version = v; }
void AGMV_SetCompression(AGMV* a, AGMV_COMPRESSION v) { a->compression = v; }
void AGMV_SetAudioState(AGMV* a, Bool v) { a->enable_audio = v; }
void AGMV_SetVolume(AGMV* a, f32 v) { a->volume = AGMV_ClampVolume(v); }
void AGMV_SetBitsPerSample(AGMV* a, u16 v) { a->header.bits_per_sample = v; }

u32 AGMV_GetWidth(AGMV* a) { return a->header.width; }
u32 AGMV_GetHeight(AGMV* a) { return a->header.height; }
u32 AGMV_GetFramesPerSecond(AGMV* a) { return a->header.frames_per_second; }
u32 AGMV_GetNumberOfFrames(AGMV* a) { return a->header.num_of_frames; }
u32 AGMV_GetTotalAudioDuration(AGMV* a) { return a->header.total_audio_duration; }
u32 AGMV_GetSampleRate(AGMV* a) { return a->header.sample_rate; }
u16 AGMV_GetNumberOfChannels(AGMV* a) { return a->header.num_of_channels; }
u32 AGMV_GetAudioSize(AGMV* a) { return a->header.audio_size; }
f32 AGMV_GetLeniency(AGMV* a) { return a->leniency; }
u8 AGMV_GetVersion(AGMV* a) { return a->header.version; }
AGMV_OPT AGMV_GetOPT(AGMV* a) { return a->opt; }
AGMV_COMPRESSION AGMV_GetCompression(AGMV* a) { return a->compression; }
Bool AGMV_GetAudioState(AGMV* a) { return a->enable_audio; }
f32 AGMV_GetVolume(AGMV* a) { return a->volume; }
u16 AGMV_GetBitsPerSample(AGMV* a) { return a->header.bits_per_sample; }

AGMV_INFO AGMV_GetVideoInfo(AGMV* a)
{
	AGMV_INFO i;
	memset(&i, 0, sizeof(i));
	i.width = a->header.width; i.height = a->header.height; i.number_of_frames = a->header.num_of_frames;
	i.version = a->header.version; i.total_audio_duration = a->header.total_audio_duration;
	i.sample_rate = a->header.sample_rate; i.audio_size = a->header.audio_size;
	i.number_of_channels = a->header.num_of_channels; i.bits_per_sample = a->header.bits_per_sample;
	return i;
}

/* container version from the option set (reference src/agmv_utils.c:487-545): 256-colour modes
   are versions 2/4, 512-colour modes 1/3; LZ77 adds 2. */
u8 AGMV_GetVersionFromOPT(AGMV_OPT opt, AGMV_COMPRESSION compression)
{
	int two_fifty_six = (opt == AGMV_OPT_II || opt == AGMV_OPT_ANIM || opt == AGMV_OPT_GBA_II);
	int known = opt >= AGMV_OPT_I && opt <= AGMV_OPT_NDS;
	if (!known) return 1;
	return (u8)((two_fifty_six ? 2 : 1) + (compression == AGMV_LZSS_COMPRESSION ? 0 : 2));
}

/* ------------------------------------------------------------------------------------------
 * FILE* byte I/O (reference src/agmv_utils.c:20-31, 61-131): little-endian, 4-byte longs on
 * disk although u32 is 8 bytes in memory; a short read leaves zeros.
 * ------------------------------------------------------------------------------------------ */
Bool AGMV_EOF(FILE* f)
{
	long pos = ftell(f), end;
	fseek(f, 0, SEEK_END);
	end = ftell(f);
	fseek(f, pos, SEEK_SET);
	return pos >= end ? TRUE : FALSE;
}

u8 AGMV_ReadByte(FILE* f) { u8 b = 0; if (fread(&b, 1, 1, f) != 1) b = 0; return b; }
u16 AGMV_ReadShort(FILE* f) { u8 b[2] = {0, 0}; size_t n = fread(b, 1, 2, f); (void)n; return (u16)(b[0] | b[1] << 8); }
u32 AGMV_ReadLong(FILE* f)
{
	u8 b[4] = {0, 0, 0, 0};
	size_t n = fread(b, 1, 4, f);
	(void)n;
	return (u32)b[0] | (u32)b[1] << 8 | (u32)b[2] << 16 | (u32)b[3] << 24;
}
void AGMV_ReadFourCC(FILE* f, char cc[4]) { int i; for (i = 0; i < 4; i++) cc[i] = (char)AGMV_ReadByte(f); }

void AGMV_WriteByte(FILE* f, u8 b) { fwrite(&b, 1, 1, f); }
void AGMV_WriteShort(FILE* f, u16 w) { u8 b[2] = {(u8)w, (u8)(w >> 8)}; fwrite(b, 2, 1, f); }
void AGMV_WriteLong(FILE* f, u32 d) { u8 b[4] = {(u8)d, (u8)(d >> 8), (u8)(d >> 16), (u8)(d >> 24)}; fwrite(b, 4, 1, f); }
void AGMV_WriteFourCC(FILE* f, char a, char b, char c, char d) { char cc[4] = {a, b, c, d}; fwrite(cc, 1, 4, f); }

Bool AGMV_IsCorrectFourCC(char cc[4], char f, char o, char u, char r)
{
	return (cc[0] == f && cc[1] == o && cc[2] == u && cc[3] == r) ? TRUE : FALSE;
}

/* ------------------------------------------------------------------------------------------
 * bit I/O, LSB first (reference src/agmv_utils.c:32-59, 86-112).  One shared state for reader
 * and writer, like the reference's file-static bitbuf/bitsin.
 * ------------------------------------------------------------------------------------------ */
static unsigned long long g_bitbuf = 0;
static unsigned g_bitsin = 0;

u32 AGMV_ReadBits(FILE* f, u32 n)
{
	unsigned long long v = g_bitbuf >> (8 - g_bitsin);
	while (n > g_bitsin) {
		g_bitbuf = AGMV_ReadByte(f);
		v |= g_bitbuf << g_bitsin;
		g_bitsin += 8;
	}
	g_bitsin -= (unsigned)n;
	return (u32)(v & ((1ull << n) - 1ull));
}

void AGMV_FlushReadBits(void) { g_bitbuf = 0; g_bitsin = 0; }

void AGMV_WriteBits(FILE* f, u32 num, u16 n)
{
	g_bitbuf |= (unsigned long long)num << g_bitsin;
	g_bitsin += n;
	while (g_bitsin >= 8) {
		AGMV_WriteByte(f, (u8)g_bitbuf);
		g_bitbuf >>= 8;
		g_bitsin -= 8;
	}
}

void AGMV_FlushWriteBits(FILE* f)
{
	if (g_bitsin > 0) {
		AGMV_WriteByte(f, (u8)g_bitbuf);
		g_bitbuf = 0;
		g_bitsin = 0;
	}
}

/* ------------------------------------------------------------------------------------------
 * chunk scan (reference src/agmv_utils.c:140-243): compare four bytes, advance one byte
 * ------------------------------------------------------------------------------------------ */
static void find_chunk(FILE* f, char c0, char c1, char c2, char c3)
{
	char cc[4];
	AGMV_ReadFourCC(f, cc);
	if (AGMV_IsCorrectFourCC(cc, c0, c1, c2, c3)) { fseek(f, ftell(f) - 4, SEEK_SET); return; }
	for (;;) {
		if (AGMV_EOF(f)) break;
		AGMV_ReadFourCC(f, cc);
		if (AGMV_IsCorrectFourCC(cc, c0, c1, c2, c3)) break;
		fseek(f, ftell(f) - 3, SEEK_SET);
	}
	fseek(f, ftell(f) - 4, SEEK_SET);
}

void AGMV_FindNextFrameChunk(FILE* f) { find_chunk(f, 'A', 'G', 'F', 'C'); }
void AGMV_FindNextAudioChunk(FILE* f) { find_chunk(f, 'A', 'G', 'A', 'C'); }

void AGMV_SkipFrameChunk(FILE* f)
{
	u32 csize;
	AGMV_FindNextFrameChunk(f);
	AGMV_ReadLong(f); AGMV_ReadLong(f); AGMV_ReadLong(f);
	csize = AGMV_ReadLong(f);
	fseek(f, (long)csize, SEEK_CUR);
}

void AGMV_SkipAudioChunk(FILE* f)
{
	u32 size;
	AGMV_FindNextAudioChunk(f);
	AGMV_ReadLong(f);
	size = AGMV_ReadLong(f);
	fseek(f, (long)size, SEEK_CUR);
}

void AGMV_ParseAGMV(FILE* f, AGMV* a)
{
	u32 n = AGMV_GetNumberOfFrames(a), i;
	int audio = AGMV_GetTotalAudioDuration(a) != 0;
	for (i = 0; i < n; i++) {
		AGMV_FindNextFrameChunk(f);
		if (a->frame_count < MAX_OFFSET_TABLE) a->offset_table[a->frame_count] = (u32)ftell(f);
		a->frame_count++;
		AGMV_SkipFrameChunk(f);
		if (audio) { AGMV_FindNextAudioChunk(f); AGMV_DecodeAudioChunk(f, a); }
	}
	a->frame_count = 0;
}

/* ------------------------------------------------------------------------------------------
 * utilities (reference src/agmv_utils.c:547-642, 695-783, 916-1033)
 * ------------------------------------------------------------------------------------------ */
f32 AGMV_ClampVolume(f32 v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
u16 AGMV_SwapShort(u16 w) { return (u16)((w << 8) | (w >> 8)); }
u32 AGMV_SwapLong(u32 d)
{
	return ((d & 0xffu) << 24) | ((d & 0xff00u) << 8) | ((d >> 8) & 0xff00u) | ((d >> 24) & 0xffu);
}
void AGMV_CopyImageData(u32* dst, u32* src, u32 n) { memcpy(dst, src, (size_t)n * sizeof(u32)); }
void AGMV_SyncFrameAndImage(AGMV* a, u32* img)
{
	memcpy(a->frame->img_data, img, (size_t)a->frame->width * a->frame->height * sizeof(u32));
}
int AGMV_Abs(int a) { return a < 0 ? -a : a; }
int AGMV_Min(int a, int b) { return a < b ? a : b; }
u8 AGMV_GetR(u32 c) { return (u8)(c >> 16); }
u8 AGMV_GetG(u32 c) { return (u8)(c >> 8); }
u8 AGMV_GetB(u32 c) { return (u8)c; }

/* bit budgets of the palette histogram: HIGH r6 g6 b7, MID r5 g6 b6, LOW r5 g6 b5 */
static void qbits(AGMV_QUALITY q, int* rb, int* gb, int* bb)
{
	switch (q) {
	case AGMV_MID_QUALITY: *rb = 5; *gb = 6; *bb = 6; break;
	case AGMV_LOW_QUALITY: *rb = 5; *gb = 6; *bb = 5; break;
	default: *rb = 6; *gb = 6; *bb = 7; break;
	}
}

u8 AGMV_GetQuantizedR(u32 c, AGMV_QUALITY q) { int r, g, b; qbits(q, &r, &g, &b); return (u8)((c >> (g + b)) & ((1u << r) - 1)); }
u8 AGMV_GetQuantizedG(u32 c, AGMV_QUALITY q) { int r, g, b; qbits(q, &r, &g, &b); return (u8)((c >> b) & ((1u << g) - 1)); }
u8 AGMV_GetQuantizedB(u32 c, AGMV_QUALITY q) { int r, g, b; qbits(q, &r, &g, &b); return (u8)(c & ((1u << b) - 1)); }

u32 AGMV_QuantizeColor(u32 c, AGMV_QUALITY q)
{
	int rb, gb, bb;
	qbits(q, &rb, &gb, &bb);
	return (u32)(AGMV_GetR(c) >> (8 - rb)) << (gb + bb) | (u32)(AGMV_GetG(c) >> (8 - gb)) << bb | (u32)(AGMV_GetB(c) >> (8 - bb));
}

u32 AGMV_ReverseQuantizeColor(u32 c, AGMV_QUALITY q)
{
	int rb, gb, bb;
	qbits(q, &rb, &gb, &bb);
	return (u32)AGMV_GetQuantizedR(c, q) << (8 - rb) << 16 | (u32)AGMV_GetQuantizedG(c, q) << (8 - gb) << 8 |
	       (u32)AGMV_GetQuantizedB(c, q) << (8 - bb);
}

/* grey-equality ratio of two frames (reference src/agmv_utils.c:920-947) */
f32 AGMV_CompareFrameSimilarity(u32* f1, u32* f2, u32 w, u32 h)
{
	size_t n = (size_t)w * h, i, same = 0;
	for (i = 0; i < n; i++) {
		u8 g1 = (u8)((AGMV_GetR(f1[i]) + AGMV_GetG(f1[i]) + AGMV_GetB(f1[i])) / 3.0f);
		u8 g2 = (u8)((AGMV_GetR(f2[i]) + AGMV_GetG(f2[i]) + AGMV_GetB(f2[i])) / 3.0f);
		same += g1 == g2;
	}
	return same / (f32)n;
}

/* PDIFS midpoint (reference src/agmv_utils.c:949-969); host form for single frames, the
   sequence drivers use agmv_hip_interp_dev */
void AGMV_InterpFrame(u32* out, u32* f1, u32* f2, u32 w, u32 h)
{
	size_t n = (size_t)w * h, i;
	for (i = 0; i < n; i++) {
		int r1 = AGMV_GetR(f1[i]), g1 = AGMV_GetG(f1[i]), b1 = AGMV_GetB(f1[i]);
		int r2 = AGMV_GetR(f2[i]), g2 = AGMV_GetG(f2[i]), b2 = AGMV_GetB(f2[i]);
		out[i] = (u32)((r1 + ((r2 - r1) >> 1)) << 16 | (g1 + ((g2 - g1) >> 1)) << 8 | (b1 + ((b2 - b1) >> 1)));
	}
}

/* exported for API completeness (reference src/agmv_utils.c:995-1010): n-1 passes of adjacent
   swaps on strict '>', i.e. a stable ascending sort of (data, gram) pairs.  Implemented as a
   stable merge sort -- identical permutation, O(n log n). */
void AGMV_BubbleSort(u32* data, u32* gram, u32 n)
{
	u32 *d2, *g2, *sd = data, *sg = gram, *dd, *dg, width, i;
	if (n < 2) return;
	d2 = (u32*)malloc(sizeof(u32) * n);
	g2 = (u32*)malloc(sizeof(u32) * n);
	dd = d2; dg = g2;
	for (width = 1; width < n; width *= 2) {
		for (i = 0; i < n; i += 2 * width) {
			u32 l = i, m = i + width < n ? i + width : n, r = i + 2 * width < n ? i + 2 * width : n, a = l, b = m, o = l;
			while (a < m && b < r) {
				if (sd[b] < sd[a]) { dd[o] = sd[b]; dg[o++] = sg[b++]; }
				else { dd[o] = sd[a]; dg[o++] = sg[a++]; }
			}
			while (a < m) { dd[o] = sd[a]; dg[o++] = sg[a++]; }
			while (b < r) { dd[o] = sd[b]; dg[o++] = sg[b++]; }
		}
		{ u32* t = sd; sd = dd; dd = t; t = sg; sg = dg; dg = t; }
	}
	if (sd != data) { memcpy(data, sd, sizeof(u32) * n); memcpy(gram, sg, sizeof(u32) * n); }
	free(d2); free(g2);
}

char* AGMV_Error2Str(Error e)
{
	switch (e) {
	case NO_ERR: return "NO ERROR";
	case FILE_NOT_FOUND_ERR: return "FILE NOT FOUND ERROR";
	case INVALID_HEADER_FORMATTING_ERR: return "INVALID HEADER FORMATTING ERROR";
	case MEMORY_CORRUPTION_ERR: return "MEMORY CORRUPTION ERROR";
	}
	return "INVALID ERROR CODE";
}

u32 AGMV_GetNumberOfBytesRead(u32 bits) { return (u32)(bits / 8.0f); }

int AGMV_NextIFrame(int n, int fc) { while ((n + fc) % 4 != 0) n++; return n; }
int AGMV_PrevIFrame(int n, int fc) { while ((n - fc) % 4 != 0) n--; return n; }
int AGMV_SkipToNearestIFrame(int n) { while (n % 4 != 0) n++; return n; }

/* patches the fps field of an existing file (header offset 18) */
int AGMV_ResetFrameRate(const char* filename, u32 fps)
{
	FILE* f = fopen(filename, "rb+");
	if (!f) return FILE_NOT_FOUND_ERR;
	fseek(f, 18, SEEK_SET);
	AGMV_WriteLong(f, fps);
	fclose(f);
	return NO_ERR;
}

/* ---- small helpers of the reference API that nothing on the hot path calls ------------------ */

/* reference src/agmv_utils.c:818-849: like AGMV_FindNearestColor but over slots 0..199 only */
u8 AGMV_FindSmallestColor(u32 palette[256], u32 color)
{
	int r = AGMV_GetR(color), g = AGMV_GetG(color), b = AGMV_GetB(color), i;
	u32 best = 3u * 255u * 255u + 1u;
	u8 at = 0;
	for (i = 0; i < 200; i++) {
		int dr = r - (int)AGMV_GetR(palette[i]), dg = g - (int)AGMV_GetG(palette[i]), db = b - (int)AGMV_GetB(palette[i]);
		u32 d = (u32)(dr * dr + dg * dg + db * db);
		if (d < best) { best = d; at = (u8)i; }
	}
	return at;
}

/* reference src/agmv_utils.c:897-914: compares the two INDICES, not the distances */
AGMV_ENTRY AGMV_FindSmallestEntry(u32 palette0[256], u32 palette1[256], u32 color)
{
	AGMV_ENTRY e;
	u8 i0 = AGMV_FindSmallestColor(palette0, color), i1 = AGMV_FindSmallestColor(palette1, color);
	memset(&e, 0, sizeof(e));
	if (i0 <= i1) { e.index = i0; e.pal_num = 0; }
	else { e.index = i1; e.pal_num = 1; }
	return e;
}

/* reference src/agmv_playback.c:117-134 */
void PlotPixel(u32* vram, int x, int y, int w, int h, u32 color)
{
	if (x >= 0 && y >= 0 && x < w && y < h) vram[x + y * w] = color;
}

void AGMV_DisplayFrame(u32* vram, u16 width, u16 height, AGMV* agmv)
{
	u32 fw = agmv->frame->width, fh = agmv->frame->height, x, y;
	for (y = 0; y < fh; y++)
		for (x = 0; x < fw; x++) PlotPixel(vram, (int)x, (int)y, width, height, agmv->frame->img_data[x + y * fw]);
}

/* reference src/agmv_utils.c:1577-1615: ./agmv.h with the file as `agmv_file[FILE_SIZE]`, a line break every 500 bytes */
void AGMV_ExportAGMVToHeader(const char* filename)
{
	FILE *in = fopen(filename, "rb"), *out;
	long n, i;
	u8* data;
	if (!in) return;
	fseek(in, 0, SEEK_END); n = ftell(in); fseek(in, 0, SEEK_SET);
	data = (u8*)malloc(n > 0 ? (size_t)n : 1);
	if (!data || fread(data, 1, (size_t)n, in) != (size_t)n) { free(data); fclose(in); return; }
	fclose(in);
	out = fopen("agmv.h", "w");
	if (!out) { free(data); return; }
	fprintf(out, "#ifndef AGMV_H\n#define AGMV_H\n\n#define FILE_SIZE %ld\n\nconst unsigned char agmv_file[FILE_SIZE] = {\n", n);
	for (i = 0; i < n; i++) {
		if (i % 500 == 0 && i != 0) fprintf(out, "\n");
		fprintf(out, "%d,", data[i]);
	}
	fprintf(out, "};\n#endif");
	fclose(out);
	free(data);
}
