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

/* rows are converted in strips through one buffer of a few hundred KB (one fread per strip) */
static int bmp_open(const char* path, FILE** fp, uint32_t* W, uint32_t* H)
{
	unsigned char hdr[54];
	FILE* f = fopen(path, "rb");
	if (!f) return FILE_NOT_FOUND_ERR;
	if (fread(hdr, 1, 54, f) != 54 || rd16(hdr) != 0x4d42 || rd32(hdr + 14) != 40 || rd16(hdr + 28) != 24 || rd32(hdr + 30) != 0) {
		fclose(f);
		return INVALID_HEADER_FORMATTING_ERR;         /* only uncompressed 24-bit BITMAPINFOHEADER files */
	}
	*W = rd32(hdr + 18); *H = rd32(hdr + 22);
	if (*W == 0 || *H == 0 || *W > 65535 || *H > 65535) { fclose(f); return INVALID_HEADER_FORMATTING_ERR; }
	*fp = f;
	return NO_ERR;
}

static void bmp_read_rows(FILE* f, uint32_t W, uint32_t H, uint32_t* dst, size_t max_px)
{
	const uint32_t pad = W % 4;                           /* AGIDL's rule (not the BMP standard's) */
	const size_t rowbytes = (size_t)W * 3 + pad;
	uint32_t strip = (uint32_t)((256u << 10) / rowbytes), y, x, r;
	unsigned char* buf;
	if (strip < 1) strip = 1;
	buf = (unsigned char*)malloc(rowbytes * strip);
	for (y = 0; y < H && (size_t)y * W < max_px; y += strip) {
		const uint32_t rows = y + strip <= H ? strip : H - y;
		const size_t want = rowbytes * rows, got = fread(buf, 1, want, f);
		if (got < want) memset(buf + got, 0, want - got);
		for (r = 0; r < rows; r++) {
			const unsigned char* row = buf + rowbytes * r;
			uint32_t* out = dst + (size_t)(y + r) * W;
			size_t n = W;
			if ((size_t)(y + r) * W >= max_px) break;
			if ((size_t)(y + r) * W + n > max_px) n = max_px - (size_t)(y + r) * W;
			for (x = 0; x < n; x++) out[x] = (uint32_t)row[3 * x + 2] << 16 | (uint32_t)row[3 * x + 1] << 8 | row[3 * x];
		}
	}
	free(buf);
}

int agmv_bmp_load(const char* path, uint32_t** pix, uint32_t* w, uint32_t* h)
{
	uint32_t W, H;
	FILE* f;
	int err = bmp_open(path, &f, &W, &H);
	if (err != NO_ERR) return err;
	*pix = (uint32_t*)malloc((size_t)W * H * sizeof(uint32_t));
	bmp_read_rows(f, W, H, *pix, (size_t)W * H);
	fclose(f);
	*w = W; *h = H;
	return NO_ERR;
}

/* the same straight into a caller's buffer of max_px pixels (pinned staging): the first max_px pixels in file order; returns the
   file's dimensions */
int agmv_bmp_load_into(const char* path, uint32_t* dst, size_t max_px, uint32_t* w, uint32_t* h)
{
	uint32_t W, H;
	FILE* f;
	int err = bmp_open(path, &f, &W, &H);
	if (err != NO_ERR) return err;
	bmp_read_rows(f, W, H, dst, max_px);
	fclose(f);
	*w = W; *h = H;
	return NO_ERR;
}

int agmv_bmp_save(const char* path, const uint32_t* pix, uint32_t W, uint32_t H)
{
	unsigned char hdr[54], *buf;
	uint32_t x, y, pad = W % 4, strip, r;
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
	strip = (uint32_t)((256u << 10) / rowbytes);
	if (strip < 1) strip = 1;
	buf = (unsigned char*)calloc(rowbytes * strip, 1);
	for (y = 0; y < H; y += strip) {
		const uint32_t rows = y + strip <= H ? strip : H - y;
		for (r = 0; r < rows; r++) {
			unsigned char* row = buf + rowbytes * r;
			const uint32_t* in = pix + (size_t)(y + r) * W;
			for (x = 0; x < W; x++) {
				const uint32_t c = in[x];
				row[3 * x] = (unsigned char)c; row[3 * x + 1] = (unsigned char)(c >> 8); row[3 * x + 2] = (unsigned char)(c >> 16);
			}
		}
		fwrite(buf, 1, rowbytes * rows, f);
	}
	free(buf);
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
