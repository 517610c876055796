/*
 * include/agmv.h -- libagmv-compatible C API of the MI355X build (drop-in for the hot path).
 *
 * Same symbols, argument meaning, struct layouts and error behaviour as the reference's
 * public headers (reference include/agmv_defines.h, agmv_encode.h, agmv_decode.h,
 * agmv_utils.h, agmv_playback.h), so existing callers (README snippets, the example programs,
 * tools/agmvcli) compile and link unchanged against libagmv_amd/libagmv.so.
 * The per-frame work behind AGMV_EncodeFrame / AGMV_DecodeFrameChunk and the batch drivers
 * runs on the GPU through include/agmv_hip.h; LZSS/LZ77, the container, BMP I/O and the
 * palette build are host C.  There is no CPU fallback for the hot path: without a GPU the
 * encode/decode entry points abort with a message (the void encoders have no error channel,
 * reference src/agmv_encode.c:529).
 *
 * Out of scope in this build (declared by the reference, not provided here): the ten
 * non-BMP image formats, audio import/export (WAV/AIFF), the Win32 player helpers.
 */
#ifndef AGMV_H
#define AGMV_H

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar types (reference include/agmv_defines.h:20-30). NOTE u32 is `unsigned long`:
 * 8 bytes on LP64 -- frame buffers are 8 B/pixel at this boundary and are packed to 4 B/pixel
 * before they go to the GPU. */
typedef unsigned char  u8;
typedef unsigned short u16;
typedef unsigned long  u32;
typedef signed char    s8;
typedef signed short   s16;
typedef signed long    s32;
typedef float          f32;
typedef int            Bool;

#ifndef TRUE
#define TRUE  1
#define FALSE 0
#endif

typedef enum Error {                       /* reference include/agmv_defines.h:37-42 */
	NO_ERR = 0x0,
	INVALID_HEADER_FORMATTING_ERR = 0x1,
	FILE_NOT_FOUND_ERR = 0x2,
	MEMORY_CORRUPTION_ERR = 0x3,
} Error;

#define AGMV_MAX_CLR      524287
#define MAX_OFFSET_TABLE  40000

#define AGMV_FILL_FLAG    0x4E             /* block opcodes, reference include/agmv_defines.h:49-53 */
#define AGMV_NORMAL_FLAG  0x2f
#define AGMV_COPY_FLAG    0x5E
#define AGMV_FILL_COUNT   14
#define AGMV_COPY_COUNT   13

#define AGMV_GBA_W 120                     /* reference include/agmv_encode.h:21-24 */
#define AGMV_GBA_H  80
#define AGMV_NDS_W 128
#define AGMV_NDS_H  96

typedef enum AGMV_OPT {                    /* reference include/agmv_defines.h:56-65 */
	AGMV_OPT_I = 0x1,                      /* 512 colours, heavy PDIFS */
	AGMV_OPT_II = 0x2,                     /* 256 colours, light PDIFS */
	AGMV_OPT_III = 0x3,                    /* 512 colours, light PDIFS */
	AGMV_OPT_ANIM = 0x4,                   /* 256 colours, heavy PDIFS */
	AGMV_OPT_GBA_I = 0x5,                  /* 512 colours, heavy PDIFS, 120x80 */
	AGMV_OPT_GBA_II = 0x6,                 /* 256 colours, heavy PDIFS, 120x80 */
	AGMV_OPT_GBA_III = 0x7,                /* 512 colours, light PDIFS, 120x80 */
	AGMV_OPT_NDS = 0x8,                    /* 512 colours, light (BMP) PDIFS, 128x96 */
} AGMV_OPT;

typedef enum AGMV_QUALITY { AGMV_HIGH_QUALITY = 0x1, AGMV_MID_QUALITY = 0x2, AGMV_LOW_QUALITY = 0x3 } AGMV_QUALITY;
typedef enum AGMV_COMPRESSION { AGMV_LZSS_COMPRESSION = 0x1, AGMV_LZ77_COMPRESSION = 0x2 } AGMV_COMPRESSION;

typedef enum AGMV_IMG_TYPE {
	AGMV_IMG_BMP = 0x1, AGMV_IMG_TGA = 0x2, AGMV_IMG_TIM = 0x3, AGMV_IMG_PCX = 0x4, AGMV_IMG_LMP = 0x5,
	AGMV_IMG_PVR = 0x6, AGMV_IMG_GXT = 0x7, AGMV_IMG_BTI = 0x8, AGMV_IMG_3DF = 0x9, AGMV_IMG_PPM = 0x0A,
	AGMV_IMG_LBM = 0x0B,
} AGMV_IMG_TYPE;

typedef enum AGMV_AUDIO_TYPE { AGMV_AUDIO_WAV = 0x1, AGMV_AUDIO_AIFF = 0x2, AGMV_AUDIO_AIFC = 0x3, AGMV_AUDIO_RAW = 0x4 } AGMV_AUDIO_TYPE;

/* ---- records; member order and types fixed by the reference (include/agmv_defines.h:78-162) */
typedef struct AGMV_MAIN_HEADER {
	char fourcc[4];
	u32 num_of_frames;
	u32 width;
	u32 height;
	u8  fmt;
	u8  version;
	u32 frames_per_second;
	u32 total_audio_duration;
	u32 sample_rate;
	u32 audio_size;
	u16 num_of_channels;
	u16 bits_per_sample;
	u32 palette0[256];
	u32 palette1[256];
} AGMV_MAIN_HEADER;

typedef struct AGMV_FRAME_CHUNK { char fourcc[4]; u32 frame_num; u32 uncompressed_size; u32 compressed_size; } AGMV_FRAME_CHUNK;
typedef struct AGMV_AUDIO_CHUNK { char fourcc[4]; u32 size; u8* atsample; s8* satsample; } AGMV_AUDIO_CHUNK;
typedef struct AGMV_FRAME { u32 width; u32 height; u32* img_data; } AGMV_FRAME;
typedef struct AGMV_AUDIO_TRACK { u32 total_audio_duration; u32 start_point; u16* pcm; u8* pcm8; } AGMV_AUDIO_TRACK;
typedef struct AGMV_ENTRY { u8 pal_num; u8 index; u32 occurence; } AGMV_ENTRY;
typedef struct AGMV_INFO {
	u32 width; u32 height; u32 number_of_frames; u8 version; u32 total_audio_duration; u32 sample_rate;
	u32 audio_size; u16 number_of_channels; u16 bits_per_sample;
} AGMV_INFO;
typedef struct AGMV_BITSTREAM { u8* data; u32 len; u32 pos; } AGMV_BITSTREAM;

typedef struct AGMV {
	AGMV_MAIN_HEADER header;
	AGMV_FRAME_CHUNK* frame_chunk;
	AGMV_AUDIO_CHUNK* audio_chunk;
	AGMV_BITSTREAM* bitstream;
	AGMV_FRAME* frame;
	AGMV_FRAME* iframe;
	AGMV_AUDIO_TRACK* audio_track;
	AGMV_ENTRY* iframe_entries;
	AGMV_OPT opt;
	AGMV_COMPRESSION compression;
	u32 frame_count;
	f32 leniency;
	u32 offset_table[MAX_OFFSET_TABLE];
	Bool enable_audio;
	f32 volume;
} AGMV;

/* ---- object lifecycle + attributes (reference include/agmv_utils.h:53-92) */
AGMV* CreateAGMV(u32 num_of_frames, u32 width, u32 height, u32 frames_per_second);
void  DestroyAGMV(AGMV* agmv);

void AGMV_SetWidth(AGMV* agmv, u32 width);
void AGMV_SetHeight(AGMV* agmv, u32 height);
void AGMV_SetICP0(AGMV* agmv, u32 palette0[256]);
void AGMV_SetICP1(AGMV* agmv, u32 palette1[256]);
void AGMV_SetFramesPerSecond(AGMV* agmv, u32 frames_per_second);
void AGMV_SetNumberOfFrames(AGMV* agmv, u32 num_of_frames);
void AGMV_SetTotalAudioDuration(AGMV* agmv, u32 total_audio_duration);
void AGMV_SetSampleRate(AGMV* agmv, u32 sample_rate);
void AGMV_SetNumberOfChannels(AGMV* agmv, u8 num_of_channels);
void AGMV_SetAudioSize(AGMV* agmv, u32 size);
void AGMV_SetLeniency(AGMV* agmv, f32 leniency);
void AGMV_SetOPT(AGMV* agmv, AGMV_OPT opt);
void AGMV_SetVersion(AGMV* agmv, u8 version);
void AGMV_SetCompression(AGMV* agmv, AGMV_COMPRESSION compression);
void AGMV_SetAudioState(AGMV* agmv, Bool audio);
void AGMV_SetVolume(AGMV* agmv, f32 volume);
void AGMV_SetBitsPerSample(AGMV* agmv, u16 bits_per_sample);

u32 AGMV_GetWidth(AGMV* agmv);
u32 AGMV_GetHeight(AGMV* agmv);
u32 AGMV_GetFramesPerSecond(AGMV* agmv);
u32 AGMV_GetNumberOfFrames(AGMV* agmv);
u32 AGMV_GetTotalAudioDuration(AGMV* agmv);
u32 AGMV_GetSampleRate(AGMV* agmv);
u16 AGMV_GetNumberOfChannels(AGMV* agmv);
u32 AGMV_GetAudioSize(AGMV* agmv);
f32 AGMV_GetLeniency(AGMV* agmv);
u8  AGMV_GetVersion(AGMV* agmv);
AGMV_OPT AGMV_GetOPT(AGMV* agmv);
AGMV_COMPRESSION AGMV_GetCompression(AGMV* agmv);
Bool AGMV_GetAudioState(AGMV* agmv);
f32 AGMV_GetVolume(AGMV* agmv);
u16 AGMV_GetBitsPerSample(AGMV* agmv);
AGMV_INFO AGMV_GetVideoInfo(AGMV* agmv);

/* ---- FILE* byte / bit I/O and chunk scan (reference include/agmv_utils.h:23-49).
 * The bit reader and writer share one file-static state exactly like the reference
 * (src/agmv_utils.c:32-36): one encode or decode stream per process at a time. */
Bool AGMV_EOF(FILE* file);
u32  AGMV_ReadBits(FILE* file, u32 num_of_bits);
u8   AGMV_ReadByte(FILE* file);
u16  AGMV_ReadShort(FILE* file);
u32  AGMV_ReadLong(FILE* file);
void AGMV_ReadFourCC(FILE* file, char fourcc[4]);
void AGMV_WriteBits(FILE* file, u32 num, u16 num_of_bits);
void AGMV_WriteByte(FILE* file, u8 byte);
void AGMV_WriteShort(FILE* file, u16 word);
void AGMV_WriteLong(FILE* file, u32 dword);
void AGMV_WriteFourCC(FILE* file, char f, char o, char u, char r);
void AGMV_FlushReadBits(void);
void AGMV_FlushWriteBits(FILE* file);
void AGMV_FindNextFrameChunk(FILE* file);
void AGMV_FindNextAudioChunk(FILE* file);
void AGMV_SkipFrameChunk(FILE* file);
void AGMV_SkipAudioChunk(FILE* file);
void AGMV_ParseAGMV(FILE* file, AGMV* agmv);
Bool AGMV_IsCorrectFourCC(char fourcc[4], char f, char o, char u, char r);

/* ---- small utilities (reference include/agmv_utils.h:94-131) */
int  AGMV_NextIFrame(int n, int frame_count);
int  AGMV_PrevIFrame(int n, int frame_count);
int  AGMV_SkipToNearestIFrame(int n);
u8   AGMV_GetVersionFromOPT(AGMV_OPT opt, AGMV_COMPRESSION compression);
f32  AGMV_ClampVolume(f32 volume);
u16  AGMV_SwapShort(u16 word);
u32  AGMV_SwapLong(u32 dword);
void AGMV_CopyImageData(u32* dest, u32* src, u32 size);
void AGMV_SyncFrameAndImage(AGMV* agmv, u32* img_data);
int  AGMV_Abs(int a);
int  AGMV_Min(int a, int b);
u8   AGMV_GetR(u32 color);
u8   AGMV_GetG(u32 color);
u8   AGMV_GetB(u32 color);
u8   AGMV_GetQuantizedR(u32 color, AGMV_QUALITY quality);
u8   AGMV_GetQuantizedG(u32 color, AGMV_QUALITY quality);
u8   AGMV_GetQuantizedB(u32 color, AGMV_QUALITY quality);
u32  AGMV_QuantizeColor(u32 color, AGMV_QUALITY quality);
u32  AGMV_ReverseQuantizeColor(u32 color, AGMV_QUALITY quality);
f32  AGMV_CompareFrameSimilarity(u32* frame1, u32* frame2, u32 width, u32 height);
void AGMV_InterpFrame(u32* interp, u32* frame1, u32* frame2, u32 width, u32 height);
void AGMV_BubbleSort(u32* data, u32* gram, u32 num_of_colors);
char* AGMV_Error2Str(Error error);
u32  AGMV_GetNumberOfBytesRead(u32 bits);
int  AGMV_ResetFrameRate(const char* filename, u32 frames_per_second);

/* ---- the hot path, per frame (reference include/agmv_encode.h:26-31, agmv_decode.h:22,
 * agmv_utils.h:114-116).  All of these run on the GPU. */
u8 AGMV_FindNearestColor(u32 palette[256], u32 color);
AGMV_ENTRY AGMV_FindNearestEntry(u32 palette0[256], u32 palette1[256], u32 color);
/* unused by the library itself (reference src/agmv_utils.c:818-849, :897-914): nearest colour among the first 200
   palette slots; the entry picks the palette with the smaller INDEX.  Host code, kept for API completeness. */
u8 AGMV_FindSmallestColor(u32 palette[256], u32 color);
AGMV_ENTRY AGMV_FindSmallestEntry(u32 palette0[256], u32 palette1[256], u32 color);
u8 AGMV_ComparePFrameBlock(AGMV* agmv, u32 x, u32 y, AGMV_ENTRY* entry);
u8 AGMV_CompareIFrameBlock(AGMV* agmv, u32 x, u32 y, u32 color, AGMV_ENTRY* img_entry);
void AGMV_AssembleIFrameBitstream(AGMV* agmv, AGMV_ENTRY* img_entry);
void AGMV_AssemblePFrameBitstream(AGMV* agmv, AGMV_ENTRY* img_entry);
void AGMV_EncodeHeader(FILE* file, AGMV* agmv);
void AGMV_EncodeFrame(FILE* file, AGMV* agmv, u32* img_data);
u32  AGMV_LZSS(FILE* file, AGMV_BITSTREAM* in);
u32  AGMV_LZ77(FILE* file, AGMV_BITSTREAM* in);
void AGMV_EncodeAudioChunk(FILE* file, AGMV* agmv);
int  AGMV_DecodeHeader(FILE* file, AGMV* agmv);
int  AGMV_DecodeFrameChunk(FILE* file, AGMV* agmv);
int  AGMV_DecodeAudioChunk(FILE* file, AGMV* agmv);

/* ---- sequence drivers (reference include/agmv_encode.h:35-37, agmv_decode.h:24-26).
 * Like the reference, the three encoders call DestroyAGMV on `agmv` before returning
 * (src/agmv_encode.c:3625); only AGMV_IMG_BMP input is supported. */
void AGMV_EncodeVideo(const char* filename, const char* dir, const char* basename, u8 img_type, u32 start_frame,
                      u32 end_frame, u32 width, u32 height, u32 frames_per_second, AGMV_OPT opt,
                      AGMV_QUALITY quality, AGMV_COMPRESSION compression);
void AGMV_EncodeAGMV(AGMV* agmv, const char* filename, const char* dir, const char* basename, u8 img_type,
                     u32 start_frame, u32 end_frame, u32 width, u32 height, u32 frames_per_second, AGMV_OPT opt,
                     AGMV_QUALITY quality, AGMV_COMPRESSION compression);
void AGMV_EncodeFullAGMV(AGMV* agmv, const char* filename, const char* dir, const char* basename, u8 img_type,
                         u32 start_frame, u32 end_frame, u32 width, u32 height, u32 frames_per_second,
                         AGMV_OPT opt, AGMV_QUALITY quality, AGMV_COMPRESSION compression);
int AGMV_DecodeVideo(const char* filename, u8 img_type);
int AGMV_DecodeAGMV(const char* filename, u8 img_type, AGMV_AUDIO_TYPE audio_type);

/* ---- playback helpers (reference include/agmv_playback.h:23-31) */
void AGMV_ResetVideo(FILE* file, AGMV* agmv);
Bool AGMV_IsVideoDone(AGMV* agmv);
void AGMV_SkipForwards(FILE* file, AGMV* agmv, int n);
void AGMV_SkipForwardsAndDecodeAudio(FILE* file, AGMV* agmv, int n);
void AGMV_SkipBackwards(FILE* file, AGMV* agmv, int n);
void AGMV_SkipTo(FILE* file, AGMV* agmv, int n);
void AGMV_PlayAGMV(FILE* file, AGMV* agmv);
void PlotPixel(u32* vram, int x, int y, int w, int h, u32 color);
void AGMV_DisplayFrame(u32* vram, u16 width, u16 height, AGMV* agmv);
/* the finished file as a C array in ./agmv.h (reference src/agmv_utils.c:1577-1615) */
void AGMV_ExportAGMVToHeader(const char* filename);

/* ---- extensions of this build (not in the reference) -------------------------------------- */
/* frames per GPU batch of the sequence drivers (default 64; env AGMV_BATCH_FRAMES) and host
   threads for the LZ stage (default: online cores, env AGMV_LZ_THREADS) */
void AGMV_SetBatchFrames(unsigned frames);
void AGMV_SetLZThreads(unsigned threads);
/* GPUs the sequence encoders (AGMV_EncodeAGMV / EncodeFullAGMV / EncodeVideo) spread their GOP-aligned batches over, from this
   one host process (default 1; env AGMV_DEVICES); the output is the same file whatever the number */
void AGMV_SetDevices(unsigned devices);
/* canonical synthetic clip agmv_synth_v1: frame t as 4-byte 0x00RRGGBB pixels */
void AGMV_SynthFrame(unsigned* pix, unsigned w, unsigned h, unsigned t, unsigned long long seed);
/* palette build of the encoders (reference src/agmv_encode.c:2364-2656) from a 2^19-bin histogram
   of AGMV_QuantizeColor codes (the +1 initial count is added inside).  pal0/pal1: 256 words each. */
void AGMV_BuildPalette(const unsigned* hist, AGMV_QUALITY quality, AGMV_OPT opt, u32 pal0[256], u32 pal1[256]);

#ifdef __cplusplus
}
#endif
#endif
