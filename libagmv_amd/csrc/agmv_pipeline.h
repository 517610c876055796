/* libagmv_amd/csrc/agmv_pipeline.h -- the pipelined sequence engine (agmv_pipeline.c), internal to the host library */
#ifndef AGMV_PIPELINE_H
#define AGMV_PIPELINE_H

#include <stdio.h>

#include "agmv_hip.h"
#include "agmv_internal.h"

typedef struct agmv_pool agmv_pool;
agmv_pool* agmv_pool_start(unsigned threads);
void agmv_pool_submit(agmv_pool* p, void (*fn)(void*), void* arg);
void agmv_pool_stop(agmv_pool* p);                       /* runs what is queued, then joins */

void agmv_frame_path(char* out, size_t cap, const char* dir, const char* base, long idx);
void agmv_load_source(const char* dir, const char* base, long idx, int scale_w, int scale_h, uint32_t w, uint32_t h, uint32_t* dst);

/* one open sequence encode: frames are pushed in order (plain, or the PDIFS midpoint of two sources) and leave as AGFC
   (+ AGAC) chunks in `file`, strictly in order; batches of `cap` frames (whole GOPs) go round-robin over two workers per
   device.  `pal` = palette0 | palette1.  agmv_seq_close returns the number of frames written. */
typedef struct agmv_seq agmv_seq;
agmv_seq* agmv_seq_open(AGMV* a, FILE* file, const char* dir, const char* base, int scale_w, int scale_h, int mode512, int lz77,
                        int audio_chunks, int use_interp, unsigned cap, unsigned devices, unsigned threads, const uint32_t pal[512]);
void agmv_seq_push(agmv_seq* s, long a, long b);
u32  agmv_seq_close(agmv_seq* s);

void agmv_histogram_frames(agmv_hip_ctx* ctx, const char* dir, const char* base, u32 start, u32 end, u32 size, int quality,
                           unsigned threads, uint32_t* hist);

int agmv_decode_stream(agmv_hip_ctx* ctx, const u8* file, size_t len, size_t pos, uint32_t w, uint32_t h, uint32_t nframes, int ver,
                       int has_audio, unsigned cap_frames, unsigned threads, unsigned long* export_count);

/* agmv_codec.c */
void agmv_write_frame_chunk(FILE* f, u32 frame_no, u32 usize, u32 csize, const u8* payload);

#endif
