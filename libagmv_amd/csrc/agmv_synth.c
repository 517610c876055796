/*
 * libagmv_amd/csrc/agmv_synth.c -- agmv_synth_v1, the canonical synthetic clip of SURVEY.md 8(d).
 * Integer-only so the host statement, the HIP kernel (k_synth) and tests/synth.py agree bit for bit.
 *   region A (x < W/4)   static: frame index 0 is used in everything below
 *   region B (y >= 3H/4) flat 32x32 tiles whose colour changes every 8 frames   -> FILL blocks
 *   elsewhere            moving gradient, 1 pixel in 16 with its low 3 bits/channel flipped -> NORMAL
 */
#include "agmv.h"

static inline unsigned long long splitmix64(unsigned long long z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

void AGMV_SynthFrame(unsigned* pix, unsigned W, unsigned H, unsigned t, unsigned long long seed)
{
	unsigned x, y;
	for (y = 0; y < H; y++)
		for (x = 0; x < W; x++) {
			const unsigned long long te = x < W / 4 ? 0 : t;
			unsigned v;
			if (y >= 3 * H / 4) {
				unsigned long long tile = ((unsigned long long)(y / 32) << 40) | ((unsigned long long)(x / 32) << 20) | (te / 8);
				v = (unsigned)(splitmix64(seed ^ tile) & 0xFFFFFFu);
			} else {
				const unsigned long long h = splitmix64(seed ^ (te * 0x9E3779B97F4A7C15ull) ^ (((unsigned long long)y << 32) | x));
				unsigned r = (unsigned)(((unsigned long long)x * 255 / (W - 1) + 2 * te) & 255);
				unsigned g = (unsigned)(((unsigned long long)y * 255 / (H - 1) + te) & 255);
				unsigned b = (unsigned)((((unsigned long long)x + y) / 2 + 3 * te) & 255);
				if ((h & 15) == 0) { r ^= (unsigned)(h >> 8) & 7; g ^= (unsigned)(h >> 16) & 7; b ^= (unsigned)(h >> 24) & 7; }
				v = r << 16 | g << 8 | b;
			}
			pix[(size_t)y * W + x] = v;
		}
}
