/*
 * oracle/agmv_oracle.c -- plain-C CPU restatement of libagmv's per-frame hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see agmv_oracle.h).  Parity: PINNED against
 * the compiled reference (oracle/_ref) and tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  The code is written from the behaviour, not copied: pixels
 * are 4-byte words, entries are packed u16, all state is explicit (the reference
 * keeps it in the AGMV object and in file-static globals).
 */
#include "agmv_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- E1: channel extraction, src/agmv_utils.c:632-642 (bits >= 24 ignored) ---- */
static inline int ch_r(uint32_t c) { return (int)((c >> 16) & 0xff); }
static inline int ch_g(uint32_t c) { return (int)((c >> 8) & 0xff); }
static inline int ch_b(uint32_t c) { return (int)(c & 0xff); }

static inline uint32_t sqdist(uint32_t a, uint32_t b)
{
	int dr = ch_r(a) - ch_r(b), dg = ch_g(a) - ch_g(b), db = ch_b(a) - ch_b(b);
	return (uint32_t)(dr * dr + dg * dg + db * db);
}

/* ---- E2: src/agmv_utils.c:785-816. strict '<' => lowest index wins ties;
        initial minimum 3*255^2+1 so entry 0 always replaces it. ---- */
uint8_t orc_find_nearest_color(const uint32_t pal[256], uint32_t color)
{
	uint32_t best = 3u * 255u * 255u + 1u;
	uint8_t idx = 0;
	for (int i = 0; i < 256; i++) {
		uint32_t d = sqdist(color, pal[i]);
		if (d < best) { best = d; idx = (uint8_t)i; }
	}
	return idx;
}

/* ---- E3: src/agmv_utils.c:851-895. '<=' => palette0 wins cross-palette ties ---- */
uint16_t orc_find_nearest_entry(const uint32_t p0[256], const uint32_t p1[256], uint32_t color)
{
	uint8_t i0 = orc_find_nearest_color(p0, color);
	uint8_t i1 = orc_find_nearest_color(p1, color);
	uint32_t d0 = sqdist(color, p0[i0]), d1 = sqdist(color, p1[i1]);
	return d0 <= d1 ? (uint16_t)i0 : (uint16_t)(0x100u | i1);
}

/* ---- E4: loop A of AGMV_EncodeFrame, src/agmv_encode.c:556-558 / :589-592 ---- */
void orc_quantise(const uint32_t p0[256], const uint32_t p1[256], int mode512,
                  const uint32_t* pix, size_t n, uint16_t* entries)
{
	for (size_t i = 0; i < n; i++)
		entries[i] = mode512 ? orc_find_nearest_entry(p0, p1, pix[i])
		                     : (uint16_t)orc_find_nearest_color(p0, pix[i]);
}

static inline uint32_t entry_colour(const uint32_t* p0, const uint32_t* p1, uint16_t e)
{
	return (e >> 8) ? p1[e & 0xff] : p0[e & 0xff];
}

static inline int within2(uint32_t a, uint32_t b)
{
	int dr = ch_r(a) - ch_r(b), dg = ch_g(a) - ch_g(b), db = ch_b(a) - ch_b(b);
	if (dr < 0) dr = -dr;
	if (dg < 0) dg = -dg;
	if (db < 0) db = -db;
	return dr <= 2 && dg <= 2 && db <= 2;
}

/* ---- E5: src/agmv_encode.c:302-352 ---- */
uint8_t orc_compare_iframe_block(const uint32_t p0[256], const uint32_t p1[256], uint32_t w,
                                 uint32_t x, uint32_t y, uint32_t color, const uint16_t* entries)
{
	uint8_t count = 0;
	for (uint32_t j = 0; j < 4; j++)
		for (uint32_t i = 0; i < 4; i++)
			count += (uint8_t)within2(color, entry_colour(p0, p1, entries[(x + i) + (y + j) * w]));
	return count;
}

/* ---- E6: src/agmv_encode.c:240-300 (compares ENTRY colours of this frame with the
        ENTRY colours of the GOP's I-frame at the same position; zero-motion only) ---- */
uint8_t orc_compare_pframe_block(const uint32_t p0[256], const uint32_t p1[256], uint32_t w,
                                 uint32_t x, uint32_t y, const uint16_t* entries,
                                 const uint16_t* iframe_entries)
{
	uint8_t count = 0;
	for (uint32_t j = 0; j < 4; j++)
		for (uint32_t i = 0; i < 4; i++) {
			size_t k = (x + i) + (size_t)(y + j) * w;
			count += (uint8_t)within2(entry_colour(p0, p1, entries[k]),
			                          entry_colour(p0, p1, iframe_entries[k]));
		}
	return count;
}

/* entry code, src/agmv_encode.c:382-388 (512 colours: 1 byte if index<127 else escape
   byte pal<<7|127 + raw index) and :419-431 (256 colours: raw index). */
static inline size_t put_code(uint8_t* out, size_t pos, uint16_t e, int mode512)
{
	uint8_t idx = (uint8_t)(e & 0xff), pal = (uint8_t)(e >> 8);
	if (!mode512) { out[pos++] = idx; return pos; }
	if (idx < 127) { out[pos++] = (uint8_t)(pal << 7 | idx); }
	else { out[pos++] = (uint8_t)(pal << 7 | 127); out[pos++] = idx; }
	return pos;
}

static size_t assemble(const uint32_t* p0, const uint32_t* p1, int mode512, uint32_t w, uint32_t h,
                       const uint16_t* entries, const uint16_t* iframe_entries, uint8_t* out)
{
	size_t pos = 0;
	for (uint32_t y = 0; y < h; y += 4) {
		for (uint32_t x = 0; x < w; x += 4) {
			uint16_t first = entries[x + (size_t)y * w];
			/* 256-colour mode always looks the block colour up in palette0
			   (src/agmv_encode.c:416,503); pal_num is 0 there anyway. */
			uint32_t color = mode512 ? entry_colour(p0, p1, first) : p0[first & 0xff];
			uint8_t count1 = orc_compare_iframe_block(p0, p1, w, x, y, color, entries);
			if (iframe_entries) {
				uint8_t count2 = orc_compare_pframe_block(p0, p1, w, x, y, entries, iframe_entries);
				if (count2 >= ORC_COPY_COUNT) { out[pos++] = ORC_COPY_FLAG; continue; }
			}
			if (count1 >= ORC_FILL_COUNT) {
				out[pos++] = ORC_FILL_FLAG;
				pos = put_code(out, pos, first, mode512);
			} else {
				out[pos++] = ORC_NORMAL_FLAG;
				for (uint32_t j = 0; j < 4; j++)
					for (uint32_t i = 0; i < 4; i++)
						pos = put_code(out, pos, entries[(x + i) + (size_t)(y + j) * w], mode512);
			}
		}
	}
	return pos;
}

/* ---- E7: src/agmv_encode.c:354-436 ---- */
size_t orc_assemble_iframe(const uint32_t p0[256], const uint32_t p1[256], int mode512,
                           uint32_t w, uint32_t h, const uint16_t* entries, uint8_t* out)
{
	return assemble(p0, p1, mode512, w, h, entries, NULL, out);
}

/* ---- E8: src/agmv_encode.c:438-527 (COPY has priority over FILL) ---- */
size_t orc_assemble_pframe(const uint32_t p0[256], const uint32_t p1[256], int mode512,
                           uint32_t w, uint32_t h, const uint16_t* entries,
                           const uint16_t* iframe_entries, uint8_t* out)
{
	return assemble(p0, p1, mode512, w, h, entries, iframe_entries, out);
}

/* ---- E9: src/agmv_encode.c:529-634 without the FILE* / LZ stage ---- */
orc_encoder* orc_encoder_new(uint32_t w, uint32_t h, int mode512, const uint32_t* p0,
                             const uint32_t* p1, uint32_t first_frame_count)
{
	orc_encoder* e = (orc_encoder*)calloc(1, sizeof(*e));
	e->w = w; e->h = h; e->mode512 = mode512; e->frame_count = first_frame_count;
	memcpy(e->p0, p0, sizeof(e->p0));
	if (p1) memcpy(e->p1, p1, sizeof(e->p1));
	e->iframe_entries = (uint16_t*)calloc((size_t)w * h, sizeof(uint16_t));
	e->scratch = (uint16_t*)calloc((size_t)w * h, sizeof(uint16_t));
	return e;
}

void orc_encoder_free(orc_encoder* e)
{
	if (!e) return;
	free(e->iframe_entries); free(e->scratch); free(e);
}

size_t orc_encode_frame(orc_encoder* e, const uint32_t* pix, uint8_t* out, uint16_t* entries_out)
{
	size_t n = (size_t)e->w * e->h, usize;
	orc_quantise(e->p0, e->p1, e->mode512, pix, n, e->scratch);
	if (e->frame_count % 4 == 0) {                     /* :560-565 */
		usize = orc_assemble_iframe(e->p0, e->p1, e->mode512, e->w, e->h, e->scratch, out);
		memcpy(e->iframe_entries, e->scratch, n * sizeof(uint16_t));   /* :626-630 */
	} else {
		usize = orc_assemble_pframe(e->p0, e->p1, e->mode512, e->w, e->h, e->scratch,
		                            e->iframe_entries, out);
	}
	if (entries_out) memcpy(entries_out, e->scratch, n * sizeof(uint16_t));
	e->frame_count++;                                   /* :633 */
	return usize;
}

/* ---- N1: LSB-first bit packer, src/agmv_utils.c:86-112. On LP64 the reference's
        'bitsin > 16' branch is arithmetically the same as shifting one byte out. ---- */
typedef struct bitwr { uint8_t* out; size_t n; uint64_t buf; unsigned bits; } bitwr;

static void bw_put(bitwr* b, uint32_t v, unsigned nb)
{
	b->buf |= (uint64_t)v << b->bits;
	b->bits += nb;
	while (b->bits >= 8) { b->out[b->n++] = (uint8_t)b->buf; b->buf >>= 8; b->bits -= 8; }
}

static void bw_flush(bitwr* b)            /* src/agmv_utils.c:106-112 */
{
	if (b->bits > 0) { b->out[b->n++] = (uint8_t)b->buf; b->buf = 0; b->bits = 0; }
}

/* longest match starting in [max(0,i-65535), i), EARLIEST start wins (strict '>'),
   src/agmv_encode.c:125-143 / :198-216 */
static void longest_match(const uint8_t* d, int i, int maxlen, int* best_len, int* best_start)
{
	int start = i - 65535, bl = 0, bs = 0;
	if (start < 0) start = 0;
	for (; start < i; start++) {
		int j;
		if (d[start] != d[i]) continue;
		for (j = 0; j < maxlen; j++)
			if (d[start + j] != d[i + j]) break;
		if (j > bl) { bl = j; bs = start; }
	}
	*best_len = bl; *best_start = bs;
}

/* src/agmv_encode.c:106-177. csize is computed IN FLOAT (:176). */
size_t orc_lzss_compress(const uint8_t* in, size_t n, uint8_t* out, uint32_t* csize_field)
{
	bitwr b = { out, 0, 0, 0 };
	int outbits = 0, pos = (int)n;
	for (int i = 0; i < pos;) {
		int maxlen = 15, bl, bs;
		if (i + maxlen > pos) maxlen = pos - i;
		longest_match(in, i, maxlen, &bl, &bs);
		if (bl < 3) {
			bw_put(&b, 1, 1); bw_put(&b, in[i], 8);
			outbits += 9; i += 1;
		} else {
			bw_put(&b, 0, 1); bw_put(&b, (uint32_t)(i - bs), 16); bw_put(&b, (uint32_t)bl, 4);
			outbits += 21; i += bl;
		}
	}
	bw_flush(&b);                                        /* :579 */
	*csize_field = (uint32_t)((float)outbits / 8.0f);
	return b.n;
}

/* src/agmv_encode.c:179-238. NOTE the reference reads in[i+bestlength], which is in[n]
   (one past the stream) when a match runs to the end: `in` must have n+1 readable bytes. */
size_t orc_lz77_compress(const uint8_t* in, size_t n, uint8_t* out, uint32_t* csize_field)
{
	size_t o = 0;
	int outbits = 0, pos = (int)n;
	for (int i = 0; i < pos;) {
		int maxlen = 255, bl, bs;
		if (i + maxlen > pos) maxlen = pos - i;
		longest_match(in, i, maxlen, &bl, &bs);
		if (bl > 0) {
			uint16_t dist = (uint16_t)(i - bs);
			out[o++] = (uint8_t)dist; out[o++] = (uint8_t)(dist >> 8);
			out[o++] = (uint8_t)bl; out[o++] = in[i + bl];
			i += bl + 1;
		} else {
			out[o++] = 0; out[o++] = 0; out[o++] = 0; out[o++] = in[i];
			i += 1;
		}
		outbits += 32;
	}
	*csize_field = (uint32_t)((float)outbits / 8.0f);
	return o;
}

/* ================================ decoder ===================================== */

orc_decoder* orc_decoder_new(uint32_t w, uint32_t h, int version, const uint32_t* p0,
                             const uint32_t* p1)
{
	orc_decoder* d = (orc_decoder*)calloc(1, sizeof(*d));
	d->w = w; d->h = h; d->version = version;
	memcpy(d->p0, p0, sizeof(d->p0));
	if (p1) memcpy(d->p1, p1, sizeof(d->p1));
	d->img = (uint32_t*)calloc((size_t)w * h, sizeof(uint32_t));
	d->iframe = (uint32_t*)calloc((size_t)w * h, sizeof(uint32_t));
	d->bitstream_cap = (size_t)w * h * 3 + 64;
	d->bitstream = (uint8_t*)calloc(d->bitstream_cap, 1);
	return d;
}

void orc_decoder_free(orc_decoder* d)
{
	if (!d) return;
	free(d->img); free(d->iframe); free(d->bitstream); free(d);
}

/* bit reader over memory, src/agmv_utils.c:38-54; fread at EOF yields 0 (:61-65) */
typedef struct bitrd { const uint8_t* p; size_t avail, pos; uint64_t buf; unsigned bits; } bitrd;

static inline uint32_t br_byte(bitrd* r)
{
	if (r->pos < r->avail) return r->p[r->pos++];
	return 0;
}

static uint32_t br_get(bitrd* r, unsigned nb)
{
	uint64_t v = r->buf >> (8 - r->bits);
	while (nb > r->bits) {
		r->buf = br_byte(r);
		v |= r->buf << r->bits;
		r->bits += 8;
	}
	r->bits -= nb;
	return (uint32_t)(v & ((1u << nb) - 1u));
}

/* D1: src/agmv_decode.c:171-222 */
size_t orc_decoder_lz(orc_decoder* d, const uint8_t* payload, size_t avail, uint32_t usize,
                      uint32_t csize)
{
	uint8_t* data = d->bitstream;
	uint64_t bpos = 0;
	bitrd r = { payload, avail, 0, 0, 0 };
	const uint64_t cap = d->bitstream_cap - 16; /* the reference has no bound; we refuse to smash the heap */

	if (d->version == 1 || d->version == 2) {
		uint64_t nbits = (uint64_t)csize * 8, bits = 0;
		while (bits < nbits && bpos < usize && bpos < cap) {
			uint32_t flag = br_get(&r, 1); bits++;
			if (flag & 1) {
				data[bpos++] = (uint8_t)br_get(&r, 8); bits += 8;
			} else {
				uint32_t offset = br_get(&r, 16);
				uint32_t len = br_get(&r, 4);
				uint64_t pos = bpos;
				bits += 20;
				for (uint32_t i = 0; i < len; i++) {
					uint64_t src = pos - offset + i;         /* unsigned wrap like the reference */
					if (src < bpos && bpos < cap) data[bpos++] = data[src];
				}
			}
		}
	} else {
		for (uint32_t i = 0; i < csize; i += 4) {
			uint32_t offset = br_byte(&r); offset |= br_byte(&r) << 8;
			uint32_t len = br_byte(&r);
			uint8_t byte = (uint8_t)br_byte(&r);
			uint64_t pos = bpos;
			for (uint32_t k = 0; k < len; k++) {
				uint64_t src = pos - offset + k;
				if (src < bpos && bpos < cap) data[bpos++] = data[src];
			}
			if (bpos < cap) data[bpos++] = byte;
		}
	}
	d->bpos = (uint32_t)bpos;
	return r.pos;
}

void orc_decoder_set_bitstream(orc_decoder* d, const uint8_t* bytes, uint32_t n)
{
	if (n > d->bitstream_cap - 16) n = (uint32_t)(d->bitstream_cap - 16);
	memcpy(d->bitstream, bytes, n);
	d->bpos = n;
}

static inline int is_flag(uint8_t b)
{
	return b == ORC_FILL_FLAG || b == ORC_NORMAL_FLAG || b == ORC_COPY_FLAG;
}

/* one block of D2 (512 colours, src/agmv_decode.c:229-319) or D3 (256 colours, :330-396).
   returns 1 when the reference raises `escape`. */
static int parse_block(orc_decoder* d, int mode512, uint32_t x, uint32_t y, uint64_t* bitpos_io)
{
	const uint8_t* data = d->bitstream;
	const uint64_t bpos = d->bpos;
	const uint32_t w = d->w, h = d->h;
	uint64_t bitpos = *bitpos_io;
	int escape = 0, invalid = 0;
	uint8_t byte;

	if (bitpos > bpos) return 1;                                   /* :229-232 */
	byte = data[bitpos++];
	while (!is_flag(byte)) {                                       /* :236-243 resync */
		byte = data[bitpos++];
		if (bitpos > bpos) { escape = 1; break; }
	}
	if (!is_flag(byte)) invalid = 1;                               /* :245-247 */

	if (byte == ORC_FILL_FLAG) {
		uint32_t color;
		uint8_t index = data[bitpos++];
		if (mode512) {
			const uint32_t* pal = (index >> 7) ? d->p1 : d->p0;
			uint8_t bot = index & 0x7f;
			if (bot < 127) color = pal[bot];
			else { index = data[bitpos++]; color = pal[index]; }
		} else {
			color = d->p0[index];
		}
		/* :264-266.  x, y are 64-bit `unsigned long` in the reference: for a frame one block wide (x == 0) the index
		   wraps to (y+1)*w - 1, the block's own pixel (3,0) as it was before this frame */
		if (x == w - 4 && y == h - 4) color = d->img[(size_t)x - 1 + (size_t)(y + 1) * w];
		if (bitpos > bpos) { escape = 1; }
		else
			for (uint32_t j = 0; j < 4; j++)
				for (uint32_t i = 0; i < 4; i++) d->img[(x + i) + (size_t)(y + j) * w] = color;
	} else if (byte == ORC_COPY_FLAG) {                            /* :281-290, no over-run check */
		for (uint32_t j = 0; j < 4; j++)
			for (uint32_t i = 0; i < 4; i++) {
				size_t k = (x + i) + (size_t)(y + j) * w;
				d->img[k] = d->iframe[k];
			}
	} else {                                                       /* NORMAL, or a non-flag after escape */
		for (uint32_t j = 0; j < 4; j++) {
			for (uint32_t i = 0; i < 4; i++) {
				uint32_t color;
				uint8_t index = data[bitpos++];
				if (mode512) {
					const uint32_t* pal = (index >> 7) ? d->p1 : d->p0;
					uint8_t bot = index & 0x7f;
					if (bot < 127) color = pal[bot];
					else { index = data[bitpos++]; color = pal[index]; }
				} else {
					color = 0;
				}
				if (bitpos > bpos || invalid) {                        /* :310-314 / :387-391 */
					escape = 1; invalid = 0;
					break;                                             /* leaves the ROW loop only */
				}
				if (!mode512) color = d->p0[index];
				d->img[(x + i) + (size_t)(y + j) * w] = color;
			}
		}
	}
	*bitpos_io = bitpos;
	return escape;
}

/* D2/D3/D4: src/agmv_decode.c:224-407 */
void orc_decoder_parse(orc_decoder* d, uint32_t* entry_offsets, uint32_t* n_entered)
{
	int mode512 = (d->version == 1 || d->version == 3);
	uint64_t bitpos = 0;
	uint32_t k = 0;
	int escape = 0;
	for (uint32_t y = 0; y < d->h && !escape; y += 4)
		for (uint32_t x = 0; x < d->w && !escape; x += 4) {
			if (entry_offsets && bitpos <= d->bpos) entry_offsets[k] = (uint32_t)bitpos;
			if (bitpos <= d->bpos) k++;
			escape = parse_block(d, mode512, x, y, &bitpos);
		}
	if (n_entered) *n_entered = k;
	if (d->frame_count % 4 == 0)                                   /* :401-405 */
		memcpy(d->iframe, d->img, (size_t)d->w * d->h * sizeof(uint32_t));
	d->frame_count++;
}

/* ---- container: header, src/agmv_decode.c:91-143; layout SURVEY Appendix A ---- */
static uint32_t rd32(const uint8_t* p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static uint32_t rd16(const uint8_t* p) { return p[0] | p[1] << 8; }

int orc_parse_header(const uint8_t* f, size_t len, orc_file_info* info, uint32_t* p0, uint32_t* p1)
{
	size_t pos = 38;
	if (len < 38) return 1;
	info->num_frames = rd32(f + 4); info->w = rd32(f + 8); info->h = rd32(f + 12);
	info->fmt = f[16]; info->version = f[17]; info->fps = rd32(f + 18);
	info->total_audio_duration = rd32(f + 22); info->sample_rate = rd32(f + 26);
	info->audio_size = rd32(f + 30); info->channels = rd16(f + 34);
	info->bits_per_sample = rd16(f + 36);
	if (memcmp(f, "AGMV", 4) != 0 || info->version < 1 || info->version > 4 ||
	    info->fps >= 200 || !(info->bits_per_sample == 16 || info->bits_per_sample == 8))
		return 1;                                         /* INVALID_HEADER_FORMATTING_ERR */
	memset(p0, 0, 256 * 4); memset(p1, 0, 256 * 4);
	for (int pal = 0; pal < ((info->version == 1 || info->version == 3) ? 2 : 1); pal++)
		for (int i = 0; i < 256; i++) {
			uint32_t r = pos < len ? f[pos] : 0, g = pos + 1 < len ? f[pos + 1] : 0,
			         b = pos + 2 < len ? f[pos + 2] : 0;
			/* AGIDL_RGB(r,g,b,fmt), extern/agidl/src/agidl_cc_manager.c:416-456: the encoder
			   always writes fmt=1 (RGB_888); fmt=2 is BGR_888 */
			uint32_t c = info->fmt == 2 ? (b << 16 | g << 8 | r) : (r << 16 | g << 8 | b);
			(pal ? p1 : p0)[i] = c;
			pos += 3;
		}
	info->first_chunk = pos;
	return 0;
}

/* src/agmv_utils.c:140-166: compare 4 bytes, step 1 byte */
size_t orc_find_next_frame_chunk(const uint8_t* f, size_t len, size_t pos)
{
	while (pos + 4 <= len) {
		if (f[pos] == 'A' && f[pos + 1] == 'G' && f[pos + 2] == 'F' && f[pos + 3] == 'C') return pos;
		pos++;
	}
	return len;
}

uint64_t orc_fnv1a64(const void* data, size_t nbytes, uint64_t seed)
{
	const uint8_t* p = (const uint8_t*)data;
	uint64_t h = seed ? seed : 0xcbf29ce484222325ull;
	for (size_t i = 0; i < nbytes; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
	return h;
}

/* ---- N2: src/agmv_utils.c:949-969; (c2-c1)>>1 is an arithmetic shift of an int ---- */
void orc_interp_frame(uint32_t* out, const uint32_t* f1, const uint32_t* f2, size_t n)
{
	for (size_t i = 0; i < n; i++) {
		int r1 = ch_r(f1[i]), g1 = ch_g(f1[i]), b1 = ch_b(f1[i]);
		int r2 = ch_r(f2[i]), g2 = ch_g(f2[i]), b2 = ch_b(f2[i]);
		int r = r1 + ((r2 - r1) >> 1), g = g1 + ((g2 - g1) >> 1), b = b1 + ((b2 - b1) >> 1);
		out[i] = (uint32_t)(r << 16 | g << 8 | b);
	}
}
