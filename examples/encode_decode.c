/*
 * examples/encode_decode.c -- the reference's README flow (README.md:29-86) against the drop-in library:
 * write a short synthetic BMP sequence, encode it with AGMV_EncodeAGMV, decode it with AGMV_DecodeAGMV.
 *
 *   gcc examples/encode_decode.c -Iinclude -Llibagmv_amd -lagmv -lagmv_hip -Wl,-rpath,$PWD/libagmv_amd -o /tmp/agmv_example
 *   (cd /tmp/work && /tmp/agmv_example)        # needs an MI355X; writes frames/, example.agmv, quick_export_<n>.bmp
 */
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>

#include <agmv.h>

/* 24-bit BMP the way AGIDL reads it: 54-byte header, B,G,R, rows in file order */
static void write_bmp(const char* path, const unsigned* pix, unsigned w, unsigned h)
{
	unsigned char hdr[54] = {'B', 'M'};
	unsigned size = 54 + w * h * 3, x, y;
	FILE* f = fopen(path, "wb");
	hdr[2] = size; hdr[3] = size >> 8; hdr[4] = size >> 16; hdr[5] = size >> 24;
	hdr[10] = 54; hdr[14] = 40;
	hdr[18] = w; hdr[19] = w >> 8; hdr[22] = h; hdr[23] = h >> 8;
	hdr[26] = 1; hdr[28] = 24;
	fwrite(hdr, 1, 54, f);
	for (y = 0; y < h; y++)
		for (x = 0; x < w; x++) {
			unsigned c = pix[y * w + x];
			fputc(c & 255, f); fputc((c >> 8) & 255, f); fputc((c >> 16) & 255, f);
		}
	fclose(f);
}

int main(void)
{
	const unsigned W = 320, H = 240, T = 28;
	unsigned* pix = malloc(sizeof(unsigned) * W * H), t;
	char path[64];
	AGMV* agmv;
	int err;

	mkdir("frames", 0755);
	for (t = 1; t <= T; t++) {
		AGMV_SynthFrame(pix, W, H, t, 0xA6D5);
		snprintf(path, sizeof(path), "frames/f%u.bmp", t);
		write_bmp(path, pix, W, H);
	}
	free(pix);

	agmv = CreateAGMV(T, W, H, 24);                                  /* reference README.md:35 */
	AGMV_EncodeAGMV(agmv, "example.agmv", "frames", "f", AGMV_IMG_BMP, 1, T, W, H, 24, AGMV_OPT_III, AGMV_LOW_QUALITY,
	                AGMV_LZSS_COMPRESSION);                          /* frees agmv, like the reference */
	err = AGMV_DecodeAGMV("example.agmv", AGMV_IMG_BMP, AGMV_AUDIO_WAV);   /* reference README.md:67 */
	printf("decode: %s\n", AGMV_Error2Str((Error)err));
	return err;
}
