/*
 * libagmv_amd/csrc/agmv_pipeline.c -- the pipelined sequence engine behind AGMV_EncodeAGMV / AGMV_EncodeFullAGMV /
 * AGMV_EncodeVideo and AGMV_DecodeAGMV / AGMV_DecodeVideo (reference src/agmv_encode.c:719-4407 BMP branch,
 * src/agmv_decode.c:455-647).  The reference runs load -> encode -> compress -> write one frame after the other on one
 * thread; here the stages of DIFFERENT batches overlap (SURVEY.md section 7 "pipeline shape"):
 *
 *   encode   host cores: BMP parse (+ GBA/NDS nearest scale) of batch b+1 into pinned staging
 *            GPU worker threads (two per device, devices = AGMV_DEVICES): H2D -> PDIFS midpoint -> k_encode -> D2H of
 *              batch b, each worker on its own stream and context; batches are whole GOPs, so they are independent given
 *              the palette and go round-robin over the workers / devices (multi-GPU sharding by GOP range, no exchange)
 *            host cores: exact LZSS / LZ77 of batch b-1, one task per frame
 *            calling thread: chunks written strictly in frame order
 *   decode   calling thread: chunk scan + LZ stage into ONE persistent buffer (the stale-tail semantics need it), batch
 *            slabs in pinned memory; GPU worker: H2D -> parse -> reconstruct -> D2H with the decoder state (last frame,
 *            I-frame snapshot) kept on the device; host cores: BMP export, one task per frame
 *
 * Plain C + pthreads; everything that touches the GPU goes through include/agmv_hip.h.  No CPU fallback: a GPU failure in
 * the void encoders aborts with a message, in the int-returning decoders it is returned.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "agmv_hip.h"
#include "agmv_internal.h"
#include "agmv_pipeline.h"

#include <time.h>
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
/* host allocations of the drivers: the void encoders have no error channel (reference src/agmv_encode.c:529), so a failed
   allocation ends the process with a message instead of a NULL dereference in some pool thread */
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) agmv_die("out of host memory"); return p; }
static void* xcalloc(size_t n, size_t m) { void* p = calloc(n ? n : 1, m ? m : 1); if (!p) agmv_die("out of host memory"); return p; }

static int tracing(void) { static int v = -1; if (v < 0) v = getenv("AGMV_TRACE") != NULL; return v; }
#define TRACE(...) do { if (tracing()) fprintf(stderr, "agmv trace: " __VA_ARGS__); } while (0)

/* ------------------------------------------------------------------------------------------
 * a small task pool
 * ------------------------------------------------------------------------------------------ */
typedef struct task { void (*fn)(void*); void* arg; struct task* next; } task;

struct agmv_pool {
	pthread_t th[64];
	unsigned nth;
	task *head, *tail;
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int stop;
};

static void* pool_main(void* arg)
{
	agmv_pool* p = (agmv_pool*)arg;
	for (;;) {
		task* t;
		pthread_mutex_lock(&p->mu);
		while (!p->head && !p->stop) pthread_cond_wait(&p->cv, &p->mu);
		t = p->head;
		if (t) { p->head = t->next; if (!p->head) p->tail = NULL; }
		pthread_mutex_unlock(&p->mu);
		if (!t) break;                                    /* stop and drained */
		t->fn(t->arg);
		free(t);
	}
	return NULL;
}

agmv_pool* agmv_pool_start(unsigned threads)
{
	agmv_pool* p = (agmv_pool*)xcalloc(1, sizeof(*p));
	unsigned i;
	if (threads < 1) threads = 1;
	if (threads > 64) threads = 64;
	pthread_mutex_init(&p->mu, NULL);
	pthread_cond_init(&p->cv, NULL);
	p->nth = 0;
	for (i = 0; i < threads; i++) {                            /* only threads that exist are joined later */
		if (pthread_create(&p->th[p->nth], NULL, pool_main, p) == 0) p->nth++;
	}
	if (p->nth == 0) agmv_die("cannot start a host worker thread");
	return p;
}

void agmv_pool_submit(agmv_pool* p, void (*fn)(void*), void* arg)
{
	task* t = (task*)xmalloc(sizeof(*t));
	t->fn = fn; t->arg = arg; t->next = NULL;
	pthread_mutex_lock(&p->mu);
	if (p->tail) p->tail->next = t; else p->head = t;
	p->tail = t;
	pthread_cond_signal(&p->cv);
	pthread_mutex_unlock(&p->mu);
}

void agmv_pool_stop(agmv_pool* p)
{
	unsigned i;
	if (!p) return;
	pthread_mutex_lock(&p->mu);
	p->stop = 1;
	pthread_cond_broadcast(&p->cv);
	pthread_mutex_unlock(&p->mu);
	for (i = 0; i < p->nth; i++) pthread_join(p->th[i], NULL);
	pthread_mutex_destroy(&p->mu);
	pthread_cond_destroy(&p->cv);
	free(p);
}

/* ------------------------------------------------------------------------------------------
 * source frames
 * ------------------------------------------------------------------------------------------ */
void agmv_frame_path(char* out, size_t cap, const char* dir, const char* base, long idx)
{
	if (dir[0] != 'c' || dir[1] != 'u' || dir[2] != 'r') snprintf(out, cap, "%s/%s%ld.bmp", dir, base, idx);
	else snprintf(out, cap, "%s%ld.bmp", base, idx);                /* "cur..." = current directory, src/agmv_encode.c:2373-2378 */
}

/* source frame `idx` as the encoder sees it: BMP -> 0x00RRGGBB, optional GBA/NDS nearest scale, then the first w*h pixels
   read linearly (the reference reads a 121x81 scaled image as 120x80) */
void agmv_load_source(const char* dir, const char* base, long idx, int scale_w, int scale_h, uint32_t w, uint32_t h, uint32_t* dst)
{
	char path[4096];
	uint32_t *pix = NULL, sw = 0, sh = 0;
	size_t need = (size_t)w * h, have;
	agmv_frame_path(path, sizeof(path), dir, base, idx);
	if (!scale_w) {                                        /* straight into the caller's (pinned) buffer */
		if (agmv_bmp_load_into(path, dst, need, &sw, &sh) != NO_ERR) { fprintf(stderr, "libagmv(amd): cannot read frame %s\n", path); abort(); }
		have = (size_t)sw * sh;
		if (have < need) memset(dst + have, 0, (need - have) * 4);
		return;
	}
	if (agmv_bmp_load(path, &pix, &sw, &sh) != NO_ERR) { fprintf(stderr, "libagmv(amd): cannot read frame %s\n", path); abort(); }
	{
		uint32_t nw, nh;
		uint32_t* sc = agmv_scale_nearest(pix, sw, sh, ((float)scale_w / sw) + 0.001f, ((float)scale_h / sh) + 0.001f, &nw, &nh);
		free(pix); pix = sc; sw = nw; sh = nh;
	}
	have = (size_t)sw * sh;
	memcpy(dst, pix, (have < need ? have : need) * 4);
	if (have < need) memset(dst + have, 0, (need - have) * 4);
	free(pix);
}

/* ------------------------------------------------------------------------------------------
 * encode pipeline
 * ------------------------------------------------------------------------------------------ */
typedef struct lzjob { const u8* in; uint32_t n; u8* out; u32 csize; int lz77; } lzjob;

typedef struct ebatch {
	unsigned id, n;
	long *srcA, *srcB;                     /* source frame numbers; srcB < 0: plain frame, else PDIFS midpoint of A and B */
	u32 first_fc;
	uint32_t* h_pix;                       /* pinned: frame k at [k * per], its second source at [k * per + npx] */
	unsigned loads_left, lz_left;
	int loaded, bits_ready;
	uint32_t* sizes;                       /* pinned [cap] */
	u8* h_bits; size_t h_bits_cap;         /* pinned, frames packed back to back (+1 byte each for LZ77's look past the end) */
	size_t* boff;
	lzjob* jobs;
	u8* comp; size_t comp_cap;
} ebatch;

typedef struct eworker {
	struct agmv_seq* s;
	unsigned idx;
	pthread_t th;
	agmv_hip_ctx* ctx;
	void* stream;
	uint32_t *d_frames, *d_sizes, *d_tmp[2];
	uint8_t* d_out;
	uint16_t* d_ient;
} eworker;

struct agmv_seq {
	AGMV* a;
	FILE* file;
	const char *dir, *base;
	int scale_w, scale_h, audio_chunks, mode512, lz77, use_b;
	uint32_t w, h;
	size_t npx, per, stride;
	unsigned cap, nslots, nworkers;
	ebatch* slot;
	eworker* wk;
	agmv_pool* pool;
	pthread_mutex_t mu;
	pthread_cond_t cv;
	unsigned nsubmitted, next_prep, next_write;
	int closing;
	ebatch* cur;                           /* batch being described by agmv_seq_push */
	u32 fc0, frames_written;
	u8* persist; size_t persist_len;       /* emulation of agmv->bitstream->data for LZ77's one-past-the-end read */
	char err[256];
	double t_open, t_load, t_lz, t_gpu, t_write;          /* AGMV_TRACE: summed task times */
};

typedef struct loadarg { agmv_seq* s; ebatch* b; unsigned k; int which; } loadarg;
typedef struct lzarg { agmv_seq* s; ebatch* b; unsigned k; } lzarg;

static void seq_die(agmv_seq* s, const char* what)
{
	(void)s;
	agmv_die(what);
}

static void load_task(void* p)
{
	loadarg* la = (loadarg*)p;
	agmv_seq* s = la->s;
	ebatch* b = la->b;
	const double t0 = now_s();
	agmv_load_source(s->dir, s->base, la->which ? b->srcB[la->k] : b->srcA[la->k], s->scale_w, s->scale_h, s->w, s->h,
	                 b->h_pix + (size_t)la->k * s->per + (la->which ? s->npx : 0));
	pthread_mutex_lock(&s->mu);
	s->t_load += now_s() - t0;
	if (--b->loads_left == 0) { b->loaded = 1; pthread_cond_broadcast(&s->cv); }
	pthread_mutex_unlock(&s->mu);
	free(la);
}

static void lz_task(void* p)
{
	lzarg* za = (lzarg*)p;
	agmv_seq* s = za->s;
	lzjob* j = &za->b->jobs[za->k];
	const double t0 = now_s();
	j->csize = j->lz77 ? agmv_lz77_mem(j->in, j->n, j->out) : agmv_lzss_mem(j->in, j->n, j->out);
	pthread_mutex_lock(&s->mu);
	s->t_lz += now_s() - t0;
	if (--za->b->lz_left == 0) pthread_cond_broadcast(&s->cv);
	pthread_mutex_unlock(&s->mu);
	free(za);
}

/* one GPU worker: its batches are id = idx, idx + nworkers, ... in order */
static void* eworker_main(void* p)
{
	eworker* wk = (eworker*)p;
	agmv_seq* s = wk->s;
	unsigned id;
	for (id = wk->idx;; id += s->nworkers) {
		ebatch* b;
		unsigned k;
		size_t total = 0;
		pthread_mutex_lock(&s->mu);
		while (!(id < s->nsubmitted && s->slot[id % s->nslots].id == id && s->slot[id % s->nslots].loaded) &&
		       !(s->closing && id >= s->nsubmitted))
			pthread_cond_wait(&s->cv, &s->mu);
		if (id >= s->nsubmitted) { pthread_mutex_unlock(&s->mu); break; }
		pthread_mutex_unlock(&s->mu);
		b = &s->slot[id % s->nslots];
		const double tw0 = now_s();
		for (k = 0; k < b->n; k++) {
			const uint32_t* src = b->h_pix + (size_t)k * s->per;
			uint32_t* dst = wk->d_frames + (size_t)k * s->npx;
			if (b->srcB[k] < 0) {
				if (agmv_hip_memcpy_async(wk->ctx, dst, src, s->npx * 4, 0, wk->stream)) seq_die(s, "frame upload");
			} else {                                       /* AGMV_InterpFrame on the GPU, src/agmv_utils.c:949-969 */
				if (agmv_hip_memcpy_async(wk->ctx, wk->d_tmp[0], src, s->npx * 4, 0, wk->stream) ||
				    agmv_hip_memcpy_async(wk->ctx, wk->d_tmp[1], src + s->npx, s->npx * 4, 0, wk->stream) ||
				    agmv_hip_interp_dev(wk->ctx, dst, wk->d_tmp[0], wk->d_tmp[1], s->npx, wk->stream))
					seq_die(s, "frame upload / interp");
			}
		}
		if (b->first_fc & 3u) {                            /* the batch completes a GOP the caller began: its I-frame entries */
			uint16_t* e = (uint16_t*)xmalloc(s->npx * 2);
			size_t i;
			for (i = 0; i < s->npx; i++) e[i] = (uint16_t)((s->a->iframe_entries[i].pal_num & 1u) << 8 | s->a->iframe_entries[i].index);
			if (agmv_hip_stream_sync(wk->ctx, wk->stream) || agmv_hip_memcpy_async(wk->ctx, wk->d_ient, e, s->npx * 2, 0, wk->stream) ||
			    agmv_hip_stream_sync(wk->ctx, wk->stream))
				seq_die(s, "entry plane upload");
			free(e);
		}
		if (agmv_hip_encode_frames_dev(wk->ctx, wk->d_frames, b->n, s->w, s->h, b->first_fc, wk->d_out, s->stride, wk->d_sizes, wk->d_ient,
		                               wk->stream) ||
		    agmv_hip_memcpy_async(wk->ctx, b->sizes, wk->d_sizes, 4 * (size_t)b->n, 1, wk->stream) || agmv_hip_check(wk->ctx, wk->stream))
			seq_die(s, "batch encode");
		for (k = 0; k < b->n; k++) { b->boff[k] = total; total += (size_t)b->sizes[k] + 1; }
		if (total + 16 > b->h_bits_cap) {
			agmv_hip_host_free(b->h_bits);
			b->h_bits_cap = total + total / 4 + 4096;
			b->h_bits = (u8*)agmv_hip_host_alloc(b->h_bits_cap);
			if (!b->h_bits) seq_die(s, "pinned allocation");
		}
		for (k = 0; k < b->n; k++)
			if (agmv_hip_memcpy_async(wk->ctx, b->h_bits + b->boff[k], wk->d_out + (size_t)k * s->stride, b->sizes[k], 1, wk->stream))
				seq_die(s, "bitstream download");
		if (agmv_hip_stream_sync(wk->ctx, wk->stream)) seq_die(s, "bitstream download");
		pthread_mutex_lock(&s->mu);
		s->t_gpu += now_s() - tw0;
		b->bits_ready = 1;
		pthread_cond_broadcast(&s->cv);
		pthread_mutex_unlock(&s->mu);
	}
	return NULL;
}

/* LZ stage of batch b: in frame order the byte behind each stream is set to what the reference's persistent buffer holds
   there (an earlier, longer frame's byte; LZ77 reads it, src/agmv_encode.c:222), then one task per frame */
static void prepare_batch(agmv_seq* s, ebatch* b)
{
	size_t need = 0, coff = 0;
	unsigned k;
	for (k = 0; k < b->n; k++) need += (size_t)b->sizes[k] * (s->lz77 ? 4 : 2) + 64;
	if (need > b->comp_cap) { free(b->comp); b->comp_cap = need + need / 4; b->comp = (u8*)xmalloc(b->comp_cap); }
	for (k = 0; k < b->n; k++) {
		u8* raw = b->h_bits + b->boff[k];
		const size_t n = b->sizes[k];
		raw[n] = n < s->persist_len ? s->persist[n] : 0;
		memcpy(s->persist, raw, n < s->persist_len ? n : s->persist_len);
		b->jobs[k].in = raw; b->jobs[k].n = (uint32_t)n; b->jobs[k].out = b->comp + coff; b->jobs[k].lz77 = s->lz77;
		coff += n * (s->lz77 ? 4 : 2) + 64;
	}
	pthread_mutex_lock(&s->mu);
	b->lz_left = b->n;
	pthread_mutex_unlock(&s->mu);
	for (k = 0; k < b->n; k++) {
		lzarg* za = (lzarg*)xmalloc(sizeof(*za));
		za->s = s; za->b = b; za->k = k;
		agmv_pool_submit(s->pool, lz_task, za);
	}
}

static void write_batch(agmv_seq* s, ebatch* b)
{
	unsigned k;
	const double t0 = now_s();
	for (k = 0; k < b->n; k++) {
		agmv_write_frame_chunk(s->file, s->a->frame_count + 1, b->sizes[k], b->jobs[k].csize, b->jobs[k].out);
		if (s->audio_chunks) AGMV_EncodeAudioChunk(s->file, s->a);
		s->a->frame_count++;
		s->frames_written++;
	}
	s->t_write += now_s() - t0;
}

/* the calling thread's share: prepare and write finished batches in order.  Returns when batch slot `want` is free
   (want = the id about to be described) or, with drain, when everything submitted has been written. */
static void seq_progress(agmv_seq* s, unsigned want, int drain)
{
	pthread_mutex_lock(&s->mu);
	for (;;) {
		if (s->next_prep < s->nsubmitted) {
			ebatch* b = &s->slot[s->next_prep % s->nslots];
			if (b->id == s->next_prep && b->bits_ready) {
				pthread_mutex_unlock(&s->mu);
				prepare_batch(s, b);
				pthread_mutex_lock(&s->mu);
				s->next_prep++;
				continue;
			}
		}
		if (s->next_write < s->next_prep) {
			ebatch* b = &s->slot[s->next_write % s->nslots];
			if (b->lz_left == 0) {
				pthread_mutex_unlock(&s->mu);
				write_batch(s, b);
				pthread_mutex_lock(&s->mu);
				s->next_write++;
				pthread_cond_broadcast(&s->cv);
				continue;
			}
		}
		if (drain ? s->next_write >= s->nsubmitted : want < s->next_write + s->nslots) break;
		pthread_cond_wait(&s->cv, &s->mu);
	}
	pthread_mutex_unlock(&s->mu);
}

static void begin_batch(agmv_seq* s)
{
	const unsigned id = s->nsubmitted;
	ebatch* b;
	seq_progress(s, id, 0);                                /* until the slot of batch id - nslots has been written */
	b = &s->slot[id % s->nslots];
	b->id = id; b->n = 0; b->loaded = 0; b->bits_ready = 0; b->lz_left = 0;
	b->first_fc = s->fc0 + (id ? ((s->fc0 & 3u) ? (4u - (s->fc0 & 3u)) + (id - 1) * s->cap : id * s->cap) : 0);
	s->cur = b;
}

static unsigned batch_limit(const agmv_seq* s, const ebatch* b)
{
	return (b->id == 0 && (s->fc0 & 3u)) ? 4u - (s->fc0 & 3u) : s->cap;     /* a first batch that only completes the caller's GOP */
}

static void submit_batch(agmv_seq* s)
{
	ebatch* b = s->cur;
	unsigned k, tasks = 0;
	if (!b || !b->n) return;
	for (k = 0; k < b->n; k++) tasks += b->srcB[k] < 0 ? 1 : 2;
	pthread_mutex_lock(&s->mu);
	b->loads_left = tasks;
	s->nsubmitted++;
	pthread_mutex_unlock(&s->mu);
	for (k = 0; k < b->n; k++) {
		int which;
		for (which = 0; which < (b->srcB[k] < 0 ? 1 : 2); which++) {
			loadarg* la = (loadarg*)xmalloc(sizeof(*la));
			la->s = s; la->b = b; la->k = k; la->which = which;
			agmv_pool_submit(s->pool, load_task, la);
		}
	}
	s->cur = NULL;
}

agmv_seq* agmv_seq_open(AGMV* a, FILE* file, const char* dir, const char* base, int scale_w, int scale_h, int mode512, int lz77,
                        int audio_chunks, int use_interp, unsigned cap, unsigned devices, unsigned threads, const uint32_t pal[512])
{
	agmv_seq* s = (agmv_seq*)xcalloc(1, sizeof(*s));
	unsigned i, ndev = (unsigned)agmv_hip_device_count();
	const double t0 = now_s();
	double t1;
	if (ndev < 1) agmv_die("cannot open the GPU");
	if (devices < 1) devices = 1;
	{	/* AGMV_DEVICES_OVERSUBSCRIBE=1: more "devices" than cards -- worker pair d runs on card d % ndev.  The round-robin of
		   batches over devices, the in-order chunk writer and the per-device tables are then exercised on a one-GPU box. */
		const char* ov = getenv("AGMV_DEVICES_OVERSUBSCRIBE");
		if (devices > ndev && !(ov && atoi(ov) != 0)) devices = ndev;
	}
	s->a = a; s->file = file; s->dir = dir; s->base = base; s->scale_w = scale_w; s->scale_h = scale_h;
	s->audio_chunks = audio_chunks; s->mode512 = mode512; s->lz77 = lz77; s->use_b = use_interp;
	s->w = (uint32_t)AGMV_GetWidth(a); s->h = (uint32_t)AGMV_GetHeight(a);
	s->npx = (size_t)s->w * s->h; s->per = s->npx * (use_interp ? 2 : 1); s->stride = agmv_hip_max_usize(s->w, s->h, 1);
	s->cap = (cap + 3u) & ~3u;
	s->nworkers = devices * 2;
	s->nslots = s->nworkers + 2;
	s->fc0 = (u32)a->frame_count;
	pthread_mutex_init(&s->mu, NULL);
	pthread_cond_init(&s->cv, NULL);
	s->pool = agmv_pool_start(threads);
	s->persist_len = s->stride + 64;
	s->persist = (u8*)xcalloc(s->persist_len, 1);
	s->slot = (ebatch*)xcalloc(s->nslots, sizeof(ebatch));
	for (i = 0; i < s->nslots; i++) {
		ebatch* b = &s->slot[i];
		b->id = ~0u;
		b->srcA = (long*)xmalloc(sizeof(long) * s->cap); b->srcB = (long*)xmalloc(sizeof(long) * s->cap);
		b->boff = (size_t*)xmalloc(sizeof(size_t) * s->cap);
		b->jobs = (lzjob*)xcalloc(s->cap, sizeof(lzjob));
		b->h_pix = (uint32_t*)agmv_hip_host_alloc(s->per * 4 * s->cap);
		b->sizes = (uint32_t*)agmv_hip_host_alloc(4 * (size_t)s->cap + 64);
		if (!b->h_pix || !b->sizes) agmv_die("pinned allocation");
	}
	t1 = now_s();
	TRACE("seq_open: pool + %u pinned slots of %.1f MB in %.3f s\n", s->nslots, s->per * 4.0 * s->cap / 1e6, t1 - t0);
	s->wk = (eworker*)xcalloc(s->nworkers, sizeof(eworker));
	for (i = 0; i < s->nworkers; i++) {
		eworker* wk = &s->wk[i];
		wk->s = s; wk->idx = i;
		wk->ctx = agmv_hip_create((int)((i % devices) % ndev));
		if (!wk->ctx) agmv_die("cannot open the GPU");
		if (agmv_hip_set_palette(wk->ctx, pal, pal + 256, mode512, NULL) || agmv_hip_sync()) agmv_die("palette upload");
		wk->stream = agmv_hip_stream_create(wk->ctx);
		wk->d_frames = (uint32_t*)agmv_hip_malloc_on(wk->ctx, s->npx * 4 * s->cap);
		wk->d_out = (uint8_t*)agmv_hip_malloc_on(wk->ctx, s->stride * s->cap);
		wk->d_sizes = (uint32_t*)agmv_hip_malloc_on(wk->ctx, 4 * (size_t)s->cap);
		wk->d_ient = (uint16_t*)agmv_hip_malloc_on(wk->ctx, s->npx * 2);
		wk->d_tmp[0] = use_interp ? (uint32_t*)agmv_hip_malloc_on(wk->ctx, s->npx * 4) : NULL;
		wk->d_tmp[1] = use_interp ? (uint32_t*)agmv_hip_malloc_on(wk->ctx, s->npx * 4) : NULL;
		if (!wk->stream || !wk->d_frames || !wk->d_out || !wk->d_sizes || !wk->d_ient || (use_interp && (!wk->d_tmp[0] || !wk->d_tmp[1])))
			agmv_die("device allocation");
		if (pthread_create(&wk->th, NULL, eworker_main, wk)) agmv_die("cannot start a GPU worker thread");
	}
	TRACE("seq_open: %u GPU workers (context + table + buffers) in %.3f s\n", s->nworkers, now_s() - t1);
	s->t_open = now_s();
	return s;
}

/* append one encoded frame: source `a`, or the PDIFS midpoint of sources a and b (b >= 0) */
void agmv_seq_push(agmv_seq* s, long a, long b)
{
	if (!s->cur) begin_batch(s);
	if (b >= 0 && !s->use_b) agmv_die("internal: midpoint frame on a sequence opened without interpolation");
	s->cur->srcA[s->cur->n] = a; s->cur->srcB[s->cur->n] = b;
	if (++s->cur->n == batch_limit(s, s->cur)) submit_batch(s);
}

u32 agmv_seq_close(agmv_seq* s)
{
	unsigned i;
	u32 written;
	double t0;
	submit_batch(s);
	pthread_mutex_lock(&s->mu);
	s->closing = 1;
	pthread_cond_broadcast(&s->cv);
	pthread_mutex_unlock(&s->mu);
	seq_progress(s, 0, 1);
	TRACE("pipeline: %u frames in %u batches, %.3f s from open to last chunk written; summed over the threads: BMP parse %.3f s, GPU workers "
	      "(upload + kernels + download) %.3f s, LZ %.3f s, chunk writes %.3f s\n", (unsigned)s->frames_written, s->nsubmitted, now_s() - s->t_open,
	      s->t_load, s->t_gpu, s->t_lz, s->t_write);
	t0 = now_s();
	for (i = 0; i < s->nworkers; i++) {
		eworker* wk = &s->wk[i];
		pthread_join(wk->th, NULL);
		agmv_hip_free_on(wk->ctx, wk->d_frames); agmv_hip_free_on(wk->ctx, wk->d_out); agmv_hip_free_on(wk->ctx, wk->d_sizes);
		agmv_hip_free_on(wk->ctx, wk->d_ient); agmv_hip_free_on(wk->ctx, wk->d_tmp[0]); agmv_hip_free_on(wk->ctx, wk->d_tmp[1]);
		agmv_hip_stream_destroy(wk->ctx, wk->stream);
		agmv_hip_destroy(wk->ctx);
	}
	agmv_pool_stop(s->pool);
	for (i = 0; i < s->nslots; i++) {
		ebatch* b = &s->slot[i];
		free(b->srcA); free(b->srcB); free(b->boff); free(b->jobs); free(b->comp);
		agmv_hip_host_free(b->h_pix); agmv_hip_host_free(b->sizes); agmv_hip_host_free(b->h_bits);
	}
	written = s->frames_written;
	TRACE("seq_close: teardown %.3f s\n", now_s() - t0);
	free(s->slot); free(s->wk); free(s->persist);
	pthread_mutex_destroy(&s->mu);
	pthread_cond_destroy(&s->cv);
	free(s);
	return written;
}

/* ------------------------------------------------------------------------------------------
 * pass 1 of the palette build: histogram of AGMV_QuantizeColor codes of every source frame at its ORIGINAL size (the
 * reference histograms before scaling, src/agmv_encode.c:2371-2397).  BMP parsing on the host cores, a window of frames
 * ahead of the GPU, which histograms them in order.
 * ------------------------------------------------------------------------------------------ */
typedef struct hframe { uint32_t w, h; int done; } hframe;
typedef struct hctx { pthread_mutex_t mu; pthread_cond_t cv; hframe* fr; uint32_t** ring; size_t ring_px; unsigned window; const char *dir, *base; u32 start; } hctx;
typedef struct harg { hctx* c; u32 i; } harg;

static void hist_load_task(void* p)
{
	harg* ha = (harg*)p;
	hctx* c = ha->c;
	hframe* f = &c->fr[ha->i];
	char path[4096];
	agmv_frame_path(path, sizeof(path), c->dir, c->base, (long)(c->start + ha->i));
	if (agmv_bmp_load_into(path, c->ring[ha->i % c->window], c->ring_px, &f->w, &f->h) != NO_ERR) { fprintf(stderr, "libagmv(amd): cannot read frame %s\n", path); abort(); }
	pthread_mutex_lock(&c->mu);
	f->done = 1;
	pthread_cond_broadcast(&c->cv);
	pthread_mutex_unlock(&c->mu);
	free(ha);
}

void agmv_histogram_frames(agmv_hip_ctx* ctx, const char* dir, const char* base, u32 start, u32 end, u32 size, int quality,
                           unsigned threads, uint32_t* hist /* 2^19 bins */)
{
	const u32 n = end >= start ? end - start + 1 : 0;
	const double t0 = now_s();
	uint32_t *d_hist, *d_pix;
	if (n == 0) {                                              /* the reference's `for (i = start; i <= end; i++)` does not run (src/agmv_encode.c:2371-2397) */
		memset(hist, 0, 4u << 19);
		return;
	}
	d_hist = (uint32_t*)agmv_hip_malloc_on(ctx, 4u << 19);
	void* stream = agmv_hip_stream_create(ctx);
	agmv_pool* pool = agmv_pool_start(threads);
	hctx c;
	u32 i, issued = 0;
	memset(&c, 0, sizeof(c));
	pthread_mutex_init(&c.mu, NULL);
	pthread_cond_init(&c.cv, NULL);
	c.fr = (hframe*)xcalloc(n, sizeof(hframe)); c.dir = dir; c.base = base; c.start = start;
	/* a ring of pinned frames, parsed ahead of the GPU by the host cores; only the first `size` pixels of a frame count
	   (the reference histograms img[0 .. width*height) of the size it was told, src/agmv_encode.c:2390-2394) */
	c.window = threads + 2;
	if (c.window > n) c.window = n;
	c.ring_px = size;
	c.ring = (uint32_t**)xcalloc(c.window, sizeof(uint32_t*));
	for (i = 0; i < c.window; i++) { c.ring[i] = (uint32_t*)agmv_hip_host_alloc(c.ring_px * 4); if (!c.ring[i]) agmv_die("pinned allocation"); }
	d_pix = (uint32_t*)agmv_hip_malloc_on(ctx, c.ring_px * 4);
	if (!d_hist || !d_pix || !stream) agmv_die("device allocation");
	if (agmv_hip_memset_async(ctx, d_hist, 0, 4u << 19, stream)) agmv_die("histogram");
	for (i = 0; i < n; i++) {
		hframe* f = &c.fr[i];
		size_t px;
		while (issued < n && issued < i + c.window) {      /* slot issued % window was consumed with frame issued - window */
			harg* ha = (harg*)xmalloc(sizeof(*ha));
			ha->c = &c; ha->i = issued++;
			agmv_pool_submit(pool, hist_load_task, ha);
		}
		pthread_mutex_lock(&c.mu);
		while (!f->done) pthread_cond_wait(&c.cv, &c.mu);
		pthread_mutex_unlock(&c.mu);
		px = (size_t)f->w * f->h < size ? (size_t)f->w * f->h : size;
		if (agmv_hip_memcpy_async(ctx, d_pix, c.ring[i % c.window], px * 4, 0, stream) || agmv_hip_histogram_dev(ctx, d_pix, px, quality, d_hist, stream) ||
		    agmv_hip_stream_sync(ctx, stream))
			agmv_die("histogram");
	}
	if (agmv_hip_memcpy_async(ctx, hist, d_hist, 4u << 19, 1, stream) || agmv_hip_stream_sync(ctx, stream)) agmv_die("histogram download");
	agmv_pool_stop(pool);
	for (i = 0; i < c.window; i++) agmv_hip_host_free(c.ring[i]);
	agmv_hip_free_on(ctx, d_hist); agmv_hip_free_on(ctx, d_pix);
	agmv_hip_stream_destroy(ctx, stream);
	TRACE("palette pass 1: %u frames histogrammed in %.3f s\n", (unsigned)n, now_s() - t0);
	free(c.ring); free(c.fr);
	pthread_mutex_destroy(&c.mu);
	pthread_cond_destroy(&c.cv);
}

/* ------------------------------------------------------------------------------------------
 * decode pipeline
 * ------------------------------------------------------------------------------------------ */
typedef struct dbatch {
	unsigned n;
	uint32_t first;                        /* frame_count of its first frame */
	u8* h_slab;                            /* pinned [cap][stride] */
	uint32_t *h_bpos, *h_out;              /* pinned [cap], [cap][npx] */
	unsigned long name0;                   /* quick_export_<name0 + k>.bmp */
	int filled, decoded;
	unsigned saves_left;
} dbatch;

typedef struct dpipe {
	agmv_hip_ctx* ctx;
	void* stream;
	uint32_t w, h;
	size_t npx, stride;
	unsigned cap, nslots;
	dbatch* slot;
	pthread_mutex_t mu;
	pthread_cond_t cv;
	unsigned nfilled;                      /* batches handed to the GPU worker */
	int closing, failed;
	agmv_pool* pool;
	pthread_t th;
	uint8_t* d_bits; uint32_t *d_bpos, *d_nent, *d_out[2], *d_iframe;
} dpipe;

typedef struct savearg { dpipe* d; dbatch* b; unsigned k; } savearg;

static void save_task(void* p)
{
	savearg* sa = (savearg*)p;
	dpipe* d = sa->d;
	char name[64];
	snprintf(name, sizeof(name), "quick_export_%lu.bmp", sa->b->name0 + sa->k);   /* AGIDL_QuickExport naming, agidl_img_export.c:20-41 */
	agmv_bmp_save(name, sa->b->h_out + (size_t)sa->k * d->npx, d->w, d->h);
	pthread_mutex_lock(&d->mu);
	if (--sa->b->saves_left == 0) { sa->b->filled = 0; sa->b->decoded = 0; pthread_cond_broadcast(&d->cv); }
	pthread_mutex_unlock(&d->mu);
	free(sa);
}

static void* dworker_main(void* p)
{
	dpipe* d = (dpipe*)p;
	unsigned id, prev_n = 0;
	int have_state = 0;
	for (id = 0;; id++) {
		dbatch* b = &d->slot[id % d->nslots];
		uint32_t* out = d->d_out[id & 1];
		const uint32_t* prev = have_state ? d->d_out[(id - 1) & 1] + (size_t)(prev_n - 1) * d->npx : NULL;
		unsigned k;
		int last_i = -1;
		pthread_mutex_lock(&d->mu);
		while (!(id < d->nfilled) && !d->closing) pthread_cond_wait(&d->cv, &d->mu);
		if (id >= d->nfilled) { pthread_mutex_unlock(&d->mu); break; }
		pthread_mutex_unlock(&d->mu);
		if (agmv_hip_memcpy_async(d->ctx, d->d_bits, b->h_slab, d->stride * b->n, 0, d->stream) ||
		    agmv_hip_memcpy_async(d->ctx, d->d_bpos, b->h_bpos, 4 * (size_t)b->n, 0, d->stream) ||
		    agmv_hip_decode_bitstreams_dev(d->ctx, d->d_bits, d->stride, d->d_bpos, b->n, d->w, d->h, b->first, d->d_nent, out, prev,
		                                   have_state ? d->d_iframe : NULL, d->stream))
			goto fail;
		/* decoder state for the next batch stays on the device: img_data = the last frame (read in place from this batch's
		   output), iframe = the last I-frame of the batch (src/agmv_decode.c:401-405) */
		for (k = 0; k < b->n; k++) if (((b->first + k) & 3u) == 0) last_i = (int)k;
		if (!have_state && last_i < 0 && agmv_hip_memset_async(d->ctx, d->d_iframe, 0, d->npx * 4, d->stream)) goto fail;
		if (last_i >= 0 && agmv_hip_memcpy_async(d->ctx, d->d_iframe, out + (size_t)last_i * d->npx, d->npx * 4, 2, d->stream)) goto fail;
		if (agmv_hip_memcpy_async(d->ctx, b->h_out, out, d->npx * 4 * b->n, 1, d->stream) || agmv_hip_stream_sync(d->ctx, d->stream)) goto fail;
		have_state = 1;
		prev_n = b->n;
		pthread_mutex_lock(&d->mu);
		b->decoded = 1;
		b->saves_left = b->n;
		pthread_cond_broadcast(&d->cv);
		pthread_mutex_unlock(&d->mu);
		for (k = 0; k < b->n; k++) {
			savearg* sa = (savearg*)xmalloc(sizeof(*sa));
			sa->d = d; sa->b = b; sa->k = k;
			agmv_pool_submit(d->pool, save_task, sa);
		}
		continue;
	fail:
		fprintf(stderr, "libagmv(amd): batch decode: %s (the AGMV hot path runs on the GPU only -- no CPU fallback)\n", agmv_hip_last_error());
		pthread_mutex_lock(&d->mu);
		d->failed = 1;
		pthread_cond_broadcast(&d->cv);
		pthread_mutex_unlock(&d->mu);
		break;
	}
	return NULL;
}

static size_t scan_fourcc(const u8* d, size_t len, size_t pos, const char* cc)
{
	while (pos + 4 <= len) {
		if (memcmp(d + pos, cc, 4) == 0) return pos;
		pos++;
	}
	return len;
}

/* the LZ stage of one frame on a pool thread: straight into the frame's row of the batch slab (the decoder copies only
   from bytes it has written itself, src < bpos, so the row's old content does not matter) */
typedef struct unlz { dpipe* d; int ver; const u8* payload; size_t avail, cap, used, chunk; uint32_t usize, csize, bpos; u8* row; unsigned* left; } unlz;

static void unlz_task(void* p)
{
	unlz* j = (unlz*)p;
	dpipe* d = j->d;
	j->bpos = agmv_lz_decode_mem(j->ver, j->payload, j->avail, j->usize, j->csize, j->row, j->cap, &j->used);
	pthread_mutex_lock(&d->mu);
	if (--*j->left == 0) pthread_cond_broadcast(&d->cv);
	pthread_mutex_unlock(&d->mu);
}

/* where the reference's reader stands behind a frame chunk at c whose LZ stage consumed `used` payload bytes (audio: AGMV_FindNextAudioChunk
   + skip, out of scope) */
static size_t behind_chunk(const u8* file, size_t len, size_t c, size_t used, int has_audio)
{
	size_t pos = c + 16 + used;
	if (has_audio) {
		size_t ac = scan_fourcc(file, len, pos, "AGAC");
		if (ac + 8 <= len) pos = ac + 8 + (file[ac + 4] | file[ac + 5] << 8 | file[ac + 6] << 16 | (size_t)file[ac + 7] << 24);
	}
	return pos;
}

/* the frame loop of AGMV_DecodeAGMV / AGMV_DecodeVideo on a file image: `pos` = first byte behind the header.
   The LZ stage of a batch runs on the pool, one frame per task.  Where frame k+1's chunk is depends on how many payload
   bytes the bit reader of frame k consumed (it runs past csize into the guard, src/agmv_decode.c:171-198), so the chunks of
   a batch are first located as if every reader stopped right behind its payload, and after the frames have been
   decompressed the true positions are checked in order: at the first chunk that was not where it was assumed the batch
   is cut and the next one starts from the true position.  The bytes behind bpos that the block parser may read on an
   over-run are those of the reference's ONE persistent buffer: they are taken from `persist` in frame order, which then
   receives the frame. */
int agmv_decode_stream(agmv_hip_ctx* ctx, const u8* file, size_t len, size_t pos, uint32_t w, uint32_t h, uint32_t nframes, int ver,
                       int has_audio, unsigned cap_frames, unsigned threads, unsigned long* export_count)
{
	dpipe d;
	const size_t npx = (size_t)w * h, cap = npx * 33 / 16 + 4096;
	u8* persist = (u8*)calloc(cap, 1);                     /* the reference's ONE decompression buffer, zero-initialised */
	uint32_t done = 0;
	unsigned id = 0, i;
	int rc = NO_ERR;
	memset(&d, 0, sizeof(d));
	d.ctx = ctx; d.w = w; d.h = h; d.npx = npx; d.stride = (cap + 255) & ~(size_t)255;
	d.cap = cap_frames; d.nslots = 3;
	pthread_mutex_init(&d.mu, NULL);
	pthread_cond_init(&d.cv, NULL);
	d.slot = (dbatch*)calloc(d.nslots, sizeof(dbatch));
	if (!d.slot) { d.nslots = 0; rc = MEMORY_CORRUPTION_ERR; goto out; }
	d.stream = agmv_hip_stream_create(ctx);
	d.d_bits = (uint8_t*)agmv_hip_malloc_on(ctx, d.stride * d.cap);
	d.d_bpos = (uint32_t*)agmv_hip_malloc_on(ctx, 4 * (size_t)d.cap);
	d.d_nent = (uint32_t*)agmv_hip_malloc_on(ctx, 4 * (size_t)d.cap);
	d.d_out[0] = (uint32_t*)agmv_hip_malloc_on(ctx, npx * 4 * d.cap);
	d.d_out[1] = (uint32_t*)agmv_hip_malloc_on(ctx, npx * 4 * d.cap);
	d.d_iframe = (uint32_t*)agmv_hip_malloc_on(ctx, npx * 4);
	if (!persist || !d.stream || !d.d_bits || !d.d_bpos || !d.d_nent || !d.d_out[0] || !d.d_out[1] || !d.d_iframe) { rc = MEMORY_CORRUPTION_ERR; goto out; }
	for (i = 0; i < d.nslots; i++) {
		d.slot[i].h_slab = (u8*)agmv_hip_host_alloc(d.stride * d.cap);
		d.slot[i].h_bpos = (uint32_t*)agmv_hip_host_alloc(4 * (size_t)d.cap + 64);
		d.slot[i].h_out = (uint32_t*)agmv_hip_host_alloc(npx * 4 * d.cap);
		if (!d.slot[i].h_slab || !d.slot[i].h_bpos || !d.slot[i].h_out) { rc = MEMORY_CORRUPTION_ERR; goto out; }
	}
	d.pool = agmv_pool_start(threads);
	if (pthread_create(&d.th, NULL, dworker_main, &d)) {       /* no worker: nothing to join, nothing would ever consume a batch */
		agmv_pool_stop(d.pool);
		rc = MEMORY_CORRUPTION_ERR;
		goto out;
	}
	while (done < nframes) {
		dbatch* b = &d.slot[id % d.nslots];
		unsigned n = 0;
		pthread_mutex_lock(&d.mu);
		while (b->filled && !d.failed) pthread_cond_wait(&d.cv, &d.mu);      /* the slot's frames of batch id - nslots are all exported */
		pthread_mutex_unlock(&d.mu);
		if (d.failed) break;
		{
			unlz* jobs = (unlz*)calloc(d.cap, sizeof(unlz));
			unsigned left, k;
			size_t spos = pos;
			if (!jobs) { rc = MEMORY_CORRUPTION_ERR; break; }
			while (n < d.cap && done + n < nframes) {          /* locate: every reader assumed to stop right behind its payload */
				size_t c = scan_fourcc(file, len, spos, "AGFC");
				unlz* j = &jobs[n];
				if (c + 16 > len) break;
				j->d = &d; j->ver = ver; j->chunk = c; j->left = &left;
				j->usize = file[c + 8] | file[c + 9] << 8 | file[c + 10] << 16 | (uint32_t)file[c + 11] << 24;
				j->csize = file[c + 12] | file[c + 13] << 8 | file[c + 14] << 16 | (uint32_t)file[c + 15] << 24;
				j->payload = file + c + 16; j->avail = len - (c + 16); j->cap = cap;
				j->row = b->h_slab + (size_t)n * d.stride;
				spos = behind_chunk(file, len, c, j->csize < j->avail ? j->csize : j->avail, has_audio);
				n++;
			}
			left = n;
			for (k = 0; k < n; k++) agmv_pool_submit(d.pool, unlz_task, &jobs[k]);
			pthread_mutex_lock(&d.mu);
			while (left) pthread_cond_wait(&d.cv, &d.mu);
			pthread_mutex_unlock(&d.mu);
			for (k = 0; k < n; k++) {                          /* in order: stale bytes, persistent buffer, true position of the next chunk */
				unlz* j = &jobs[k];
				const size_t bp = j->bpos, tail = bp + 16 < d.stride ? 16 : (bp < d.stride ? d.stride - bp : 0);
				if (tail) memcpy(j->row + bp, persist + bp, tail);
				memcpy(persist, j->row, bp < cap ? bp : cap);
				b->h_bpos[k] = j->bpos;
				pos = behind_chunk(file, len, j->chunk, j->used, has_audio);
				if (k + 1 < n && scan_fourcc(file, len, pos, "AGFC") != jobs[k + 1].chunk) { n = k + 1; break; }   /* the rest was decompressed from the wrong place */
			}
			free(jobs);
		}
		if (!n) break;
		b->n = n; b->first = done; b->name0 = *export_count + 1;
		*export_count += n;
		pthread_mutex_lock(&d.mu);
		b->filled = 1;
		d.nfilled = ++id;
		pthread_cond_broadcast(&d.cv);
		pthread_mutex_unlock(&d.mu);
		done += n;
	}
	pthread_mutex_lock(&d.mu);
	d.closing = 1;
	pthread_cond_broadcast(&d.cv);
	pthread_mutex_unlock(&d.mu);
	pthread_join(d.th, NULL);
	pthread_mutex_lock(&d.mu);                             /* every frame that was decoded is on disk before the call returns */
	for (i = 0; i < d.nslots; i++) while (d.slot[i].decoded && !d.failed) pthread_cond_wait(&d.cv, &d.mu);
	pthread_mutex_unlock(&d.mu);
	agmv_pool_stop(d.pool);
	if (d.failed) rc = MEMORY_CORRUPTION_ERR;
out:
	for (i = 0; i < d.nslots; i++) { agmv_hip_host_free(d.slot[i].h_slab); agmv_hip_host_free(d.slot[i].h_bpos); agmv_hip_host_free(d.slot[i].h_out); }
	agmv_hip_free_on(ctx, d.d_bits); agmv_hip_free_on(ctx, d.d_bpos); agmv_hip_free_on(ctx, d.d_nent);
	agmv_hip_free_on(ctx, d.d_out[0]); agmv_hip_free_on(ctx, d.d_out[1]); agmv_hip_free_on(ctx, d.d_iframe);
	agmv_hip_stream_destroy(ctx, d.stream);
	free(d.slot); free(persist);
	pthread_mutex_destroy(&d.mu);
	pthread_cond_destroy(&d.cv);
	return rc;
}
