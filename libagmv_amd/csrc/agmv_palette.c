/*
 * libagmv_amd/csrc/agmv_palette.c -- palette build of the sequence encoders, host side
 * (reference src/agmv_encode.c:2364-2367, 2570-2656; "next" row N3 of SURVEY.md 8f).
 *
 * Input: the pass-1 histogram of AGMV_QuantizeColor codes over all frames (counted on the GPU by
 * agmv_hip_histogram_dev or on the host).  The reference then
 *   - sorts (count, colour) pairs ascending with AGMV_BubbleSort, an O(n^2) STABLE sort of the
 *     first max_clr entries (strict '>' swaps): reproduced by a stable counting-free merge sort;
 *   - walks the colours from the most frequent down and greedily keeps a colour unless it lies
 *     within (2,2,3) [HIGH] / (1,1,1) quantised steps of ANY of the 512 slots -- including the
 *     still-empty ones, which hold colour 0, so near-black codes are always rejected;
 *   - its first candidate is colorgram[max_clr], one past the sorted range: observed value 0 with
 *     glibc (fresh zero pages) -- frozen here; histogram[max_clr] (the all-ones code) is counted but
 *     never sorted, i.e. that colour can never be picked;
 *   - scatters the picks over palette0/palette1 with the slot map of :2627-2647.
 */
#include <stdlib.h>
#include <string.h>

#include "agmv_internal.h"

void AGMV_BuildPalette(const unsigned* hist, AGMV_QUALITY quality, AGMV_OPT opt, u32 pal0[256], u32 pal1[256])
{
	u32 max_clr = quality == AGMV_MID_QUALITY ? 131071u : (quality == AGMV_LOW_QUALITY ? 65535u : (u32)AGMV_MAX_CLR);
	u32 *count = (u32*)malloc(sizeof(u32) * max_clr), *gram = (u32*)malloc(sizeof(u32) * max_clr);
	u32 pal[512], n, i, picked = 0;
	int tr = quality == AGMV_HIGH_QUALITY ? 2 : 1, tg = tr, tb = quality == AGMV_HIGH_QUALITY ? 3 : 1;

	for (i = 0; i < max_clr; i++) { count[i] = 1u + hist[i]; gram[i] = i; }   /* histogram starts at 1, :2364-2367 */
	AGMV_BubbleSort(count, gram, max_clr);                /* stable ascending by count */
	memset(pal, 0, sizeof(pal));
	memset(pal0, 0, 256 * sizeof(u32));
	memset(pal1, 0, 256 * sizeof(u32));

	for (n = max_clr; n > 0 && picked < 512; n--) {
		u32 clr = n == max_clr ? 0u : gram[n];            /* colorgram[max_clr]: frozen to 0 */
		int r = AGMV_GetQuantizedR(clr, quality), g = AGMV_GetQuantizedG(clr, quality), b = AGMV_GetQuantizedB(clr, quality);
		int skip = 0, j;
		for (j = 0; j < 512 && !skip; j++) {
			int dr = r - AGMV_GetQuantizedR(pal[j], quality), dg = g - AGMV_GetQuantizedG(pal[j], quality),
			    db = b - AGMV_GetQuantizedB(pal[j], quality);
			if (dr < 0) dr = -dr;
			if (dg < 0) dg = -dg;
			if (db < 0) db = -db;
			skip = dr <= tr && dg <= tg && db <= tb;
		}
		if (!skip) pal[picked++] = clr;
	}

	if (opt == AGMV_OPT_II || opt == AGMV_OPT_GBA_II || opt == AGMV_OPT_ANIM) {
		for (n = 0; n < 256; n++) pal0[n] = AGMV_ReverseQuantizeColor(pal[n], quality);
	} else {
		/* slot map :2627-2647: 0..125 -> p0[n]; 126..252 -> p1[n-126]; 253..381 -> p0[n-126];
		   382..510 -> p1[n-255]; p0[126] stays 0 and pick 511 is dropped */
		for (n = 0; n < 512; n++) {
			u32 c = AGMV_ReverseQuantizeColor(pal[n], quality);
			if (n < 126) pal0[n] = c;
			else if (n <= 252) pal1[n - 126] = c;
			if (n > 252 && n <= 381) pal0[n - 126] = c;
			if (n > 381 && n - 255 < 256) pal1[n - 255] = c;
		}
	}
	free(count); free(gram);
}
