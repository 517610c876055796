/*
 * libagmv_amd/csrc/agmv_bmp.c -- the caller side of the path: 24-bit BMP frames in the exact form
 * the reference's vendored image library reads and writes them (only what the configs reach:
 * AGIDL_LoadBMP + AGIDL_ColorConvertBMP, AGIDL_QuickExport, AGIDL_FastScaleBMP nearest).
 *   load   reference extern/agidl/src/agidl_img_bmp.c:585-655,973-1002: 54-byte header, pixels
 *          follow immediately, 3 bytes B,G,R per pixel, rows kept in FILE order (bottom-up files
 *          stay bottom-up in memory -- nothing is flipped), `width % 4` pad bytes skipped per row.
 *   save   reference :1041-1110 + :521-550: same layout, header fields as AGIDL_BMPEncodeHeader.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "agmv_internal.h"

static unsigned rd16(const unsigned char* p) { return p[0] | p[1] << 8; }
static unsigned rd32(const unsigned char* p) { return p[0] | p[1] << 8 | p[2] << 16 | (unsigned)p[3] << 24; }
static void wr16(unsigned char* p, unsigned v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); }
static void wr32(unsigned char* p, unsigned v) { wr16(p, v); wr16(p + 2, v >> 16); }

int agmv_bmp_load(const char* path, uint32_t** pix, uint32_t* w, uint32_t* h)
{
	unsigned char hdr[54], *row;
	uint32_t W, H, x, y, pad;
	size_t rowbytes;
	FILE* f = fopen(path, "rb");
	if (!f) return FILE_NOT_FOUND_ERR;
	if (fread(hdr, 1, 54, f) != 54 || rd16(hdr) != 0x4d42 || rd32(hdr + 14) != 40 || rd16(hdr + 28) != 24 || rd32(hdr + 30) != 0) {
		fclose(f);
		return INVALID_HEADER_FORMATTING_ERR;         /* only uncompressed 24-bit BITMAPINFOHEADER files */
	}
	W = rd32(hdr + 18); H = rd32(hdr + 22);
	if (W == 0 || H == 0 || W > 65535 || H > 65535) { fclose(f); return INVALID_HEADER_FORMATTING_ERR; }
	pad = W % 4;                                      /* AGIDL's rule (not the BMP standard's) */
	rowbytes = (size_t)W * 3 + pad;
	row = (unsigned char*)malloc(rowbytes);
	*pix = (uint32_t*)malloc((size_t)W * H * sizeof(uint32_t));
	for (y = 0; y < H; y++) {
		size_t got = fread(row, 1, rowbytes, f);
		if (got < rowbytes) memset(row + got, 0, rowbytes - got);
		for (x = 0; x < W; x++)
			(*pix)[(size_t)y * W + x] = (uint32_t)row[3 * x + 2] << 16 | (uint32_t)row[3 * x + 1] << 8 | row[3 * x];
	}
	free(row);
	fclose(f);
	*w = W; *h = H;
	return NO_ERR;
}

int agmv_bmp_save(const char* path, const uint32_t* pix, uint32_t W, uint32_t H)
{
	unsigned char hdr[54], *row;
	uint32_t x, y, pad = W % 4;
	size_t rowbytes = (size_t)W * 3 + pad;
	FILE* f = fopen(path, "wb");
	if (!f) return FILE_NOT_FOUND_ERR;
	memset(hdr, 0, sizeof(hdr));
	wr16(hdr, 0x4d42);
	wr32(hdr + 2, 54 + W * H * 3);                    /* file_size */
	wr32(hdr + 10, 54);                               /* offset */
	wr32(hdr + 14, 40);                               /* header_size */
	wr32(hdr + 18, W); wr32(hdr + 22, H);
	wr16(hdr + 26, 1);                                /* planes */
	wr16(hdr + 28, 24);                               /* bits */
	wr32(hdr + 34, W * H * 3);                        /* img_size */
	fwrite(hdr, 1, 54, f);
	row = (unsigned char*)calloc(rowbytes, 1);
	for (y = 0; y < H; y++) {
		for (x = 0; x < W; x++) {
			uint32_t c = pix[(size_t)y * W + x];
			row[3 * x] = (unsigned char)c; row[3 * x + 1] = (unsigned char)(c >> 8); row[3 * x + 2] = (unsigned char)(c >> 16);
		}
		fwrite(row, 1, rowbytes, f);
	}
	free(row);
	fclose(f);
	return NO_ERR;
}

/* new size = (u32)(w*sx) x (u32)(h*sy) in float; source pixel = (u32)(x * (float)(w-1)/nw) */
uint32_t* agmv_scale_nearest(const uint32_t* pix, uint32_t w, uint32_t h, float sx, float sy, uint32_t* nw, uint32_t* nh)
{
	uint32_t W2 = (uint32_t)(w * sx), H2 = (uint32_t)(h * sy), x, y;
	float xs = (float)(w - 1) / W2, ys = (float)(h - 1) / H2;
	uint32_t* out = (uint32_t*)malloc((size_t)W2 * H2 * sizeof(uint32_t));
	for (y = 0; y < H2; y++)
		for (x = 0; x < W2; x++) {
			uint32_t x2 = (uint32_t)(x * xs), y2 = (uint32_t)(y * ys);
			out[(size_t)y * W2 + x] = (x2 < w && y2 < h) ? pix[(size_t)y2 * w + x2] : 0;
		}
	*nw = W2; *nh = H2;
	return out;
}
