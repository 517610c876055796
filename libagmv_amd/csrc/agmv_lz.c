/*
 * libagmv_amd/csrc/agmv_lz.c -- the host entropy stage: LZSS / LZ77 as the reference defines
 * them (reference src/agmv_encode.c:106-238 encoders, src/agmv_decode.c:171-218 decoders),
 * bit-exact but not brute force.
 *
 * The reference scans the whole 65535-byte window for every token (O(n * 65535): 16 s per noisy
 * 1080p frame).  What it computes is well defined: the LONGEST match (capped at 15 for LZSS, 255
 * for LZ77) whose start lies in [i-65535, i), and among equally long ones the EARLIEST start
 * (strict '>' while scanning oldest -> newest, src/agmv_encode.c:138).  Here:
 *   LZSS  hashes of L-grams whose buckets are FIFO queues of window positions, for L = 3, 5, 8, 15: the longest of
 *         those whose gram has a verified hit bounds the match length, one walk over that gram's occurrences
 *         settles it; the queue order gives the earliest start (see lzss_run).
 *   LZ77  lengths 1..2 by the same queues, longer ones by a full walk of the 3-gram chain.
 * State is per call (the reference's is file-static), so frames compress on several host
 * threads; the FILE* entry points AGMV_LZSS/AGMV_LZ77 feed the shared AGMV_WriteBits like the
 * reference does.
 */
#include <stdlib.h>
#include <string.h>

#include "agmv.h"
#include "agmv_internal.h"

#define WIN 65535
#define RING 65536
#define HBITS 15
#define HSIZE (1u << HBITS)

/* ---- FIFO buckets of window positions for one gram length ------------------------------------
 * A bucket is a singly linked list through the ring of window slots, oldest first.  Live positions lie within RING of
 * each other, so links and bucket ends are 16-bit ring SLOTS (position & 0xFFFF) and the absolute position of a slot is
 * rebuilt from the current position: 4 bytes per slot and 8 per bucket keep the 13 queues of LZSS inside the cache
 * (3.3 MB + 3.3 MB instead of 12 MB). */
typedef struct gslot { unsigned short nxt, bkt; } gslot;       /* next newer slot of the bucket; bucket of this slot */
typedef struct gbkt { int head, tail; } gbkt;                  /* oldest / newest position + 1, 0 = empty */
typedef struct gq {
	int L;
	gbkt* b;            /* [HSIZE] */
	gslot* s;           /* [RING]  */
} gq;

static void gq_init(gq* q, int L)
{
	q->L = L;
	q->b = (gbkt*)calloc(HSIZE, sizeof(gbkt));
	q->s = (gslot*)calloc(RING, sizeof(gslot));
}

static void gq_free(gq* q) { free(q->b); free(q->s); }

/* FNV-1a over the gram, folded to HBITS: hash of length L+1 = one step from the hash of length L, so the hashes of all
   gram lengths at one position cost one pass over at most 15 bytes */
#define H_INIT 2166136261u
#define H_STEP(h, byte) (((h) ^ (byte)) * 16777619u)
#define H_FOLD(h) (((h) ^ ((h) >> 15)) & (HSIZE - 1))

static inline unsigned gram_hash(const u8* p, int L)
{
	unsigned h = H_INIT;
	int k;
	for (k = 0; k < L; k++) h = H_STEP(h, p[k]);
	return H_FOLD(h);
}

/* insert position p with its (folded) gram hash; evicts position p-RING first (it is the head of its bucket) */
static inline void gq_insert_h(gq* q, int p, unsigned h, int have_old)
{
	const unsigned slot = (unsigned)p & (RING - 1);
	gbkt* nb;
	if (have_old) {
		gbkt* ob = &q->b[q->s[slot].bkt];
		if (ob->head == p - RING + 1) {
			if (ob->tail == ob->head) ob->head = ob->tail = 0;
			else ob->head = p - RING + 1 + (int)((q->s[slot].nxt - slot) & (RING - 1));       /* next newer: rebuilt from the slot distance */
		}
	}
	nb = &q->b[h];
	q->s[slot].bkt = (unsigned short)h;
	if (nb->tail) q->s[(unsigned)(nb->tail - 1) & (RING - 1)].nxt = (unsigned short)slot;
	else nb->head = p + 1;
	nb->tail = p + 1;
}

static inline void gq_insert(gq* q, const u8* d, int p, int have_old) { gq_insert_h(q, p, gram_hash(d + p, q->L), have_old); }

/* earliest live position >= lo whose L-gram equals the one at i (hash h), or -1 */
static inline int gq_find_h(const gq* q, const u8* d, int i, unsigned h, int lo)
{
	const gbkt* b = &q->b[h];
	int p;
	if (!b->head) return -1;
	p = b->head - 1;
	for (;;) {
		if (p >= lo && memcmp(d + p, d + i, (size_t)q->L) == 0) return p;
		if (p + 1 == b->tail) return -1;
		p += (int)((q->s[(unsigned)p & (RING - 1)].nxt - ((unsigned)p & (RING - 1))) & (RING - 1));
	}
}

static inline int gq_find(const gq* q, const u8* d, int i, int lo) { return gq_find_h(q, d, i, gram_hash(d + i, q->L), lo); }

/* ---- token sinks ----------------------------------------------------------------------------*/
typedef struct sink {
	FILE* f;            /* FILE* mode: through AGMV_WriteBits / AGMV_WriteByte */
	u8* out;            /* memory mode */
	size_t n;
	unsigned long long buf;
	unsigned bits;
} sink;

static inline void put_bits(sink* s, unsigned v, unsigned nb)
{
	if (s->f) { AGMV_WriteBits(s->f, v, (u16)nb); return; }
	s->buf |= (unsigned long long)v << s->bits;
	s->bits += nb;
	while (s->bits >= 8) { s->out[s->n++] = (u8)s->buf; s->buf >>= 8; s->bits -= 8; }
}

static inline void put_byte(sink* s, unsigned v)
{
	if (s->f) AGMV_WriteByte(s->f, (u8)v);
	else s->out[s->n++] = (u8)v;
}

/* ---- LZSS (reference src/agmv_encode.c:106-177) ---------------------------------------------
 * Queues are kept for a FEW gram lengths only -- 3, 5, 8 and 15: four insertions per byte instead of thirteen, and the
 * insertions are what the time goes to.  "An L-gram has an occurrence in the window" is monotone in L, so with queue
 * lengths T[0] < T[1] < ... the longest match length L* is found from the top: the first T[k] whose gram occurs gives
 * T[k] <= L* < T[k+1], and L* with its earliest start comes from one walk over ALL the occurrences of that T[k]-gram
 * (oldest first, strictly longer wins: earliest among equals).  The 3-gram is probed first: most positions of noisy
 * data have no match at all.  Measured on pre-LZ bitstreams of the synthetic clip / a NORMAL-heavy one (ms per ~0.5 MB,
 * one core): all of 3..15: 113 / 166; {3,4,5,6,9,12,15}: 51 / 84; {3,5,8,15}: 32 / 48; {3,5,15}: 28 / 39 but 29 instead of
 * 6 on three-symbol data (long walks over frequent 5-grams); {3,15}: 35 / 47 and 190.  Same bytes in every case. */
#ifndef LZ_QSET
#define LZ_QSET {3, 5, 8, 15}
#endif
static const int LZ_Q[] = LZ_QSET;
#define LZ_NQ ((int)(sizeof(LZ_Q) / sizeof(LZ_Q[0])))

/* longest match (>= L of q, <= cap) among all live occurrences >= lo of the L-gram at i; earliest start among the
   longest; returns the length (0 = the gram does not occur) */
static inline int gq_longest(const gq* q, const u8* d, int i, unsigned h, int lo, int cap, int* start)
{
	const gbkt* b = &q->b[h];
	int p, best = 0;
	if (!b->head) return 0;
	p = b->head - 1;
	for (;;) {
		if (p >= lo && memcmp(d + p, d + i, (size_t)q->L) == 0) {
			int j = q->L;
			while (j < cap && d[p + j] == d[i + j]) j++;
			if (j > best) { best = j; *start = p; if (best == cap) return best; }
		}
		if (p + 1 == b->tail) return best;
		p += (int)((q->s[(unsigned)p & (RING - 1)].nxt - ((unsigned)p & (RING - 1))) & (RING - 1));
	}
}

static u32 lzss_run(const u8* d, int n, sink* s)
{
	gq q[16];
	unsigned char isq[16];
	int k, L, i = 0, ins = 0, outbits = 0;
	memset(isq, 0, sizeof(isq));
	for (k = 0; k < LZ_NQ; k++) { gq_init(&q[LZ_Q[k]], LZ_Q[k]); isq[LZ_Q[k]] = 1; }
	while (i < n) {
		int maxlen = n - i < 15 ? n - i : 15, lo = i - WIN, best = 0, start = 0;
		unsigned hs[16], h;
		if (lo < 0) lo = 0;
		/* make every position < i that owns an L-gram visible */
		for (; ins < i; ins++) {
			const int top = n - ins < 15 ? n - ins : 15, old = ins >= RING;
			h = H_INIT;
			for (L = 1; L <= top; L++) {
				h = H_STEP(h, d[ins + L - 1]);
				if (isq[L]) gq_insert_h(&q[L], ins, H_FOLD(h), old);
			}
		}
		h = H_INIT;
		for (L = 1; L <= maxlen; L++) { h = H_STEP(h, d[i + L - 1]); hs[L] = H_FOLD(h); }
		/* the shortest gram first (most positions of noisy data end here), then from the top */
		if (maxlen >= 3 && (best = gq_longest(&q[3], d, i, hs[3], lo, maxlen < LZ_Q[1] - 1 ? maxlen : LZ_Q[1] - 1, &start)) > 0) {
			for (k = LZ_NQ - 1; k >= 1; k--) {
				const int Lk = LZ_Q[k], cap = k + 1 < LZ_NQ ? LZ_Q[k + 1] - 1 : 15;
				int st, b;
				if (maxlen < Lk) continue;
				b = gq_longest(&q[Lk], d, i, hs[Lk], lo, maxlen < cap ? maxlen : cap, &st);
				if (b > 0) { best = b; start = st; break; }
			}
		}
		if (best < 3) {                                    /* literal: flag 1 + 8 bits */
			put_bits(s, 1, 1); put_bits(s, d[i], 8);
			outbits += 9; i += 1;
		} else {                                           /* match: flag 0 + 16-bit distance + 4-bit length */
			put_bits(s, 0, 1); put_bits(s, (unsigned)(i - start), 16); put_bits(s, (unsigned)best, 4);
			outbits += 21; i += best;
		}
	}
	for (k = 0; k < LZ_NQ; k++) gq_free(&q[LZ_Q[k]]);
	return (u32)((float)outbits / 8.0f);                   /* csize is computed in float, :176 */
}

/* ---- LZ77 (reference src/agmv_encode.c:179-238): 4-byte tokens {u16 distance, u8 length, u8 next}.
 * d must have n+1 readable bytes: a match that runs to the end emits d[n] as `next` (:222). */
static u32 lz77_run(const u8* d, int n, sink* s)
{
	gq q1, q2;
	int *chead, *cprev;                                    /* 3-gram chains, newest first */
	int i = 0, ins = 0, outbits = 0;
	gq_init(&q1, 1); gq_init(&q2, 2);
	chead = (int*)calloc(HSIZE, sizeof(int));
	cprev = (int*)calloc(RING, sizeof(int));
	while (i < n) {
		int maxlen = n - i < 255 ? n - i : 255, lo = i - WIN, best = 0, start = 0;
		if (lo < 0) lo = 0;
		for (; ins < i; ins++) {
			gq_insert(&q1, d, ins, ins >= RING);
			if (ins + 2 <= n) gq_insert(&q2, d, ins, ins >= RING);
			if (ins + 3 <= n) {
				unsigned h = gram_hash(d + ins, 3);
				cprev[(unsigned)ins & (RING - 1)] = chead[h];
				chead[h] = ins + 1;
			}
		}
		if (maxlen >= 3) {                                 /* longest >= 3, earliest among equals */
			int c = chead[gram_hash(d + i, 3)];
			while (c) {
				int p = c - 1, j;
				if (p < lo) break;                         /* chain is ordered newest -> oldest */
				for (j = 0; j < maxlen && d[p + j] == d[i + j]; j++) {}
				if (j >= 3 && j >= best) { best = j; start = p; }
				c = cprev[(unsigned)p & (RING - 1)];
				if (c && c - 1 >= p) break;                /* slot was recycled: chain left the window */
			}
		}
		if (best < 3 && maxlen >= 2) { int p = gq_find(&q2, d, i, lo); if (p >= 0) { best = 2; start = p; } }
		if (best < 2 && maxlen >= 1) { int p = gq_find(&q1, d, i, lo); if (p >= 0) { best = 1; start = p; } }
		if (best > 0) {
			unsigned dist = (unsigned)(i - start);
			put_byte(s, dist & 0xff); put_byte(s, dist >> 8); put_byte(s, (unsigned)best); put_byte(s, d[i + best]);
			i += best + 1;
		} else {
			put_byte(s, 0); put_byte(s, 0); put_byte(s, 0); put_byte(s, d[i]);
			i += 1;
		}
		outbits += 32;
	}
	gq_free(&q1); gq_free(&q2); free(chead); free(cprev);
	return (u32)((float)outbits / 8.0f);
}

/* ---- API -----------------------------------------------------------------------------------*/
u32 AGMV_LZSS(FILE* file, AGMV_BITSTREAM* in)
{
	sink s;
	memset(&s, 0, sizeof(s));
	s.f = file;
	return lzss_run(in->data, (int)in->pos, &s);
}

u32 AGMV_LZ77(FILE* file, AGMV_BITSTREAM* in)
{
	sink s;
	memset(&s, 0, sizeof(s));
	s.f = file;
	return lz77_run(in->data, (int)in->pos, &s);
}

/* memory forms used by the batch drivers: returns the bytes the reference leaves in the file for the
   payload, i.e. exactly csize bytes (the flushed partial byte is overwritten by the 0xFF guard,
   reference src/agmv_encode.c:579-585,622-624).  `out` needs 2*n+16 (LZSS) / 4*n+16 (LZ77) bytes. */
u32 agmv_lzss_mem(const u8* in, size_t n, u8* out)
{
	sink s;
	memset(&s, 0, sizeof(s));
	s.out = out;
	return lzss_run(in, (int)n, &s);
}

u32 agmv_lz77_mem(const u8* in, size_t n, u8* out)
{
	sink s;
	memset(&s, 0, sizeof(s));
	s.out = out;
	return lz77_run(in, (int)n, &s);
}

/* ---- decoders from memory (reference src/agmv_decode.c:171-218).  `data` is the persistent
 * decompression buffer (bytes beyond the new bpos keep their old content), `cap` its size.
 * Returns bpos; *consumed = payload bytes read (the bit reader runs past csize into the guard). */
typedef struct brd { const u8* p; size_t avail, pos; unsigned long long buf; unsigned bits; } brd;

static inline unsigned rd_byte(brd* r) { return r->pos < r->avail ? r->p[r->pos++] : 0u; }

static inline unsigned rd_bits(brd* r, unsigned nb)
{
	unsigned long long v = r->buf >> (8 - r->bits);
	while (nb > r->bits) { r->buf = rd_byte(r); v |= r->buf << r->bits; r->bits += 8; }
	r->bits -= nb;
	return (unsigned)(v & ((1u << nb) - 1u));
}

u32 agmv_lz_decode_mem(int version, const u8* payload, size_t avail, u32 usize, u32 csize, u8* data, size_t cap,
                       size_t* consumed)
{
	brd r;
	unsigned long long bpos = 0, lim = cap > 16 ? cap - 16 : 0;
	memset(&r, 0, sizeof(r));
	r.p = payload; r.avail = avail;
	if (version == 1 || version == 2) {
		unsigned long long nbits = (unsigned long long)csize * 8, bits = 0;
		while (bits < nbits && bpos < usize && bpos < lim) {
			unsigned flag = rd_bits(&r, 1);
			bits++;
			if (flag & 1) { data[bpos++] = (u8)rd_bits(&r, 8); bits += 8; }
			else {
				unsigned offset = rd_bits(&r, 16), len = rd_bits(&r, 4), k;
				unsigned long long pos = bpos;
				bits += 20;
				for (k = 0; k < len; k++) {
					unsigned long long src = pos - offset + k;         /* wraps like the reference's u32 */
					if (src < bpos && bpos < lim) data[bpos++] = data[src];
				}
			}
		}
	} else {
		u32 t;
		for (t = 0; t < csize; t += 4) {
			unsigned offset = rd_byte(&r), len, k;
			u8 byte;
			unsigned long long pos = bpos;
			offset |= rd_byte(&r) << 8;
			len = rd_byte(&r);
			byte = (u8)rd_byte(&r);
			for (k = 0; k < len; k++) {
				unsigned long long src = pos - offset + k;
				if (src < bpos && bpos < lim) data[bpos++] = data[src];
			}
			if (bpos < lim) data[bpos++] = byte;
		}
	}
	if (consumed) *consumed = r.pos;
	return (u32)bpos;
}
