/* libagmv_amd/csrc/agmv_internal.h -- internals shared by the host C files (not installed) */
#ifndef AGMV_INTERNAL_H
#define AGMV_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "agmv.h"

/* agmv_lz.c */
u32 agmv_lzss_mem(const u8* in, size_t n, u8* out);
u32 agmv_lz77_mem(const u8* in, size_t n, u8* out);
u32 agmv_lz_decode_mem(int version, const u8* payload, size_t avail, u32 usize, u32 csize, u8* data, size_t cap,
                       size_t* consumed);

/* agmv_bmp.c: 24-bit BMP in the exact form AGIDL reads/writes it (reference
   extern/agidl/src/agidl_img_bmp.c:585-655, 1041-1110): rows in FILE order (no flip), B,G,R bytes,
   row padding = width % 4.  Pixels are 4-byte 0x00RRGGBB. */
int  agmv_bmp_load(const char* path, uint32_t** pix, uint32_t* w, uint32_t* h);
int  agmv_bmp_load_into(const char* path, uint32_t* dst, size_t max_px, uint32_t* w, uint32_t* h);
int  agmv_bmp_save(const char* path, const uint32_t* pix, uint32_t w, uint32_t h);
/* AGIDL_FastScaleBMP(..., AGIDL_SCALE_NEAREST) as the GBA/NDS drivers call it (reference
   src/agmv_encode.c:2707-2721, extern/agidl/src/agidl_imgp_scale.c:262-291) */
uint32_t* agmv_scale_nearest(const uint32_t* pix, uint32_t w, uint32_t h, float sx, float sy, uint32_t* nw, uint32_t* nh);

/* agmv_codec.c */
void agmv_die(const char* what);

#endif
