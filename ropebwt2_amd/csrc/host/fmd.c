/* fmd.c -- FMD (fermi "RLD\3") writer, API of include/rb2_fmd.h.
 *
 * Byte-exact re-implementation of the encode side of /root/reference/rld0.c (rld_init(6,3),
 * rld_enc, rld_enc_finish, rld_rank_index, rld_dump) as one flat word array.  Format recap
 * (SURVEY.md section 8f-1):
 *   - a run (length l, symbol c) is Elias-delta(l) followed by 3 bits of c, written MSB first into
 *     64-bit words; adjacent runs of the same symbol are merged before encoding;
 *   - words are grouped in small blocks of 8; a run never straddles a block.  The first words of a
 *     block hold the symbol counts of the PREVIOUS block (total + 6 symbols) as 16-, 32- or 64-bit
 *     little-endian fields, the width tag sits in the two top bits of the first word;
 *   - storage is cut in chunks of 2^23 words; the last block of a chunk gives up one more word;
 *   - after the data, one frame of 7 words per 2^ibits symbols indexes (block offset, counts).
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include <pthread.h>
#include <stdio.h>
#include <errno.h>
#include <unistd.h>
#include <fcntl.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <time.h>
#include "rb2_fmd.h"
#include "rle.h"
#define RB2_THP_WHICH 2
#include "rb2_parcopy.h"

#define BLK_WORDS   8                    /* 1 << sbits, sbits = 3 */
#define SYM_BITS    3                    /* ilog2(asize) + 1 */
#define N_FIELDS    7                    /* total + 6 symbols */
#define CHUNK_WORDS (1u << 23)

struct rb2_fmd_s {
	uint64_t *w; size_t cap;             /* zero-initialised words */
	size_t head, p, tail;                /* current block: first word, write cursor, last usable word */
	int r;                               /* free bits in w[p] */
	int64_t cnt[N_FIELDS], mcnt[N_FIELDS];   /* running counts; counts at the start of the current block */
	int pend_c; int64_t pend_l;          /* run waiting to be merged with its successor */
	uint64_t n_bytes, n_frames, *frame;
	int finished;
	/* speculative encoding of one segment of the run stream (parallel writer, see rb2_fmdp_*): no chunk rule, and the
	 * stream offset of the first run of every block is recorded */
	int spec;
	uint32_t *start; uint8_t *type; size_t nblk, cap_blk;   /* start[k], type[k] of block k; block nblk is the one being filled */
	uint32_t run_pos;                                       /* offset of the run being encoded */
	int64_t pend_seg;                                       /* true orbit of the parallel writer: segment in which the pending run starts */
	/* streamed output (rb2_fmdp_set_output): words [0, out_done) are already in the file at out_base + 80; wmu keeps the
	 * writer thread's pwrite and reserve()'s realloc apart */
	int out_fd; int64_t out_base; size_t out_done; pthread_mutex_t *wmu; int out_err;
};

static const int hdr_words[3] = { 2, 4, 7 };   /* (7*16+63)/64, (7*32+63)/64, 7 */

static int ilog2_u64(uint64_t v) { return v ? 63 - __builtin_clzll(v) : -1; }   /* ilog2(0) = -1 like rld0.c:26-43 */

static void reserve(rb2_fmd_t *f, size_t n_words)
{
	if (n_words <= f->cap) return;
	size_t nc = f->cap ? f->cap : 1 << 16;
	while (nc < n_words) nc += nc >> 1;
	if (f->wmu) pthread_mutex_lock(f->wmu);
	f->w = (uint64_t*)realloc(f->w, nc * 8);
	if (f->wmu) pthread_mutex_unlock(f->wmu);
	if (!f->w) { fprintf(stderr, "[rb2_fmd] out of memory (%zu words)\n", nc); abort(); }
	{	/* tests: fresh words hold garbage, not the zeros a new mapping happens to bring (nothing may rely on them) */
		static int poison = -1;
		if (poison < 0) poison = getenv("RB2_FMD_POISON") != 0;
		if (poison) memset(f->w + f->cap, 0xA5, (nc - f->cap) * 8);
	}
	f->cap = nc;                                              /* (not zeroed: every block is zeroed when it is opened -- most blocks of the parallel writer are copied over whole) */
}

static size_t block_tail(size_t head)
{
	const size_t end = head + BLK_WORDS;                      /* one past the block */
	return end % CHUNK_WORDS == 0 ? end - 2 : end - 1;        /* last block of a chunk is one word shorter (rld0.h:75) */
}
static size_t block_tail_of(const rb2_fmd_t *f, size_t head) { return f->spec ? head + BLK_WORDS - 1 : block_tail(head); }

rb2_fmd_t *rb2_fmd_init(void)
{
	rb2_fmd_t *f = (rb2_fmd_t*)calloc(1, sizeof(rb2_fmd_t));
	reserve(f, 2 * BLK_WORDS);
	memset(f->w, 0, BLK_WORDS * 8);
	f->head = 0; f->p = hdr_words[0]; f->tail = block_tail(0); f->r = 64;
	f->pend_c = -1;
	f->out_fd = -1;
	return f;
}

/* close the current block: the next block's header receives the counts gathered in this one (rld0.c:107-135) */
static void open_next_block(rb2_fmd_t *f)
{
	int i, type;
	const int64_t tot = f->cnt[0] - f->mcnt[0];
	f->head += BLK_WORDS;
	reserve(f, f->head + 2 * BLK_WORDS);
	memset(f->w + f->head, 0, BLK_WORDS * 8);
	type = tot < 0x4000 ? 0 : tot < 0x40000000 ? 1 : 2;
	for (i = 0; i < N_FIELDS; ++i) {
		const uint64_t v = (uint64_t)(f->cnt[i] - f->mcnt[i]);
		if (type == 0)      f->w[f->head + i / 4] |= (v & 0xffffu) << (16 * (i % 4));
		else if (type == 1) f->w[f->head + i / 2] |= (v & 0xffffffffu) << (32 * (i % 2));
		else                f->w[f->head + i] = v;
	}
	f->w[f->head] |= (uint64_t)type << 62;
	f->p = f->head + hdr_words[type];
	f->tail = block_tail_of(f, f->head);
	f->r = 64;
	memcpy(f->mcnt, f->cnt, sizeof(f->cnt));
	if (f->spec) {                                            /* block nblk is complete; the new one starts with the run at run_pos */
		if (++f->nblk + 1 >= f->cap_blk) {
			f->cap_blk = f->cap_blk ? f->cap_blk * 2 : 1024;
			f->start = (uint32_t*)realloc(f->start, f->cap_blk * 4); f->type = (uint8_t*)realloc(f->type, f->cap_blk);
		}
		f->start[f->nblk] = f->run_pos; f->type[f->nblk] = (uint8_t)type;
	}
}

static void encode_run(rb2_fmd_t *f, int64_t l, int c)
{
	/* Elias delta of l: gamma(ilog2(l)+1) then the low bits of l (rld0.c:45-51) */
	const int y = ilog2_u64((uint64_t)l), z = ilog2_u64((uint64_t)y + 1);
	int w = 2 * z + 1 + y + SYM_BITS;
	const uint64_t x = ((((uint64_t)l ^ (1ull << y)) | (uint64_t)(y + 1) << y) << SYM_BITS) | (uint64_t)c;
	if (w >= f->r && f->p == f->tail) open_next_block(f);    /* note >=: an exactly fitting code still moves on (rld0.c:142) */
	if (w > f->r) {
		w -= f->r;
		f->w[f->p++] |= x >> w;
		f->r = 64 - w;
		f->w[f->p] = x << f->r;
	} else {
		f->r -= w;
		f->w[f->p] |= x << f->r;
	}
	f->cnt[0] += l; f->cnt[c + 1] += l;
}

void rb2_fmd_push(rb2_fmd_t *f, int64_t len, int sym)
{
	if (len == 0) return;
	if (sym == f->pend_c) { f->pend_l += len; return; }
	if (f->pend_l) encode_run(f, f->pend_l, f->pend_c);
	f->pend_c = sym; f->pend_l = len;
}

/* a chunk of 43+3 run bytes (any run width, rle.h:39-75): what the device exports and what rope leaves hold.  One-byte
 * runs -- all the device ever emits -- take the short path. */
void rb2_fmd_push_runs(rb2_fmd_t *f, const uint8_t *q, int64_t n)
{
	const uint8_t *end = q + n;
	while (q < end) {
		if ((*q & 0x80) == 0) { rb2_fmd_push(f, *q >> 3, *q & 7); ++q; }
		else { int c; int64_t l; q += rle_dec1_fn(q, &c, &l); rb2_fmd_push(f, l, c); }
	}
}

static void fmd_index(rb2_fmd_t *f);

void rb2_fmd_finish(rb2_fmd_t *f)
{
	if (f->finished) return;
	if (f->pend_l) encode_run(f, f->pend_l, f->pend_c);
	f->pend_l = 0;
	open_next_block(f);
	fmd_index(f);
}

/* stream length + rank frames (rld0.c:163-205); f->mcnt holds the totals.  Frame k describes the LAST block whose cumulative
 * symbol count S lies in [(k-1) 2^ibits, k 2^ibits): k(block) = (S >> ibits) + 1 is monotone, so ranges of blocks can be
 * handled independently once the counts in front of each range are known (two passes over the headers, nthr threads). */
typedef struct { rb2_fmd_t *f; uint64_t i0, i1, last; uint64_t sum[6], base[6]; int ibits, pass; } idx_job_t;

static inline void hdr_add(const uint64_t *h, uint64_t run[6])
{
	const int type = (int)(h[0] >> 62);
	int j;
	for (j = 1; j < N_FIELDS; ++j) {
		uint64_t v;
		if (type == 0)      v = (h[j / 4] >> (16 * (j % 4))) & 0xffffu;
		else if (type == 1) v = (h[j / 2] >> (32 * (j % 2))) & 0x3fffffffu;
		else                v = h[j];
		run[j - 1] += v;
	}
}

static void *idx_worker(void *arg)
{
	idx_job_t *jb = (idx_job_t*)arg;
	rb2_fmd_t *f = jb->f;
	uint64_t i, run[6], kprev = 0, iprev = 0, rprev[6];
	int j;
	if (jb->pass == 0) {
		memset(jb->sum, 0, sizeof(jb->sum));
		for (i = jb->i0; i < jb->i1; i += BLK_WORDS) hdr_add(f->w + i, jb->sum);
		return 0;
	}
	memcpy(run, jb->base, sizeof(run));
	for (i = jb->i0; i <= jb->i1 && i <= jb->last; i += BLK_WORDS) {   /* one block past the range: is my last k also the next range's first? */
		uint64_t sum = 0, k;
		hdr_add(f->w + i, run);
		for (j = 0; j < 6; ++j) sum += run[j];
		k = (sum >> jb->ibits) + 1;
		if (kprev && k != kprev && kprev < f->n_frames) {          /* block iprev was the last one with kprev */
			f->frame[kprev * N_FIELDS] = iprev;
			for (j = 0; j < 6; ++j) f->frame[kprev * N_FIELDS + 1 + j] = rprev[j];
		}
		if (i >= jb->i1) { kprev = 0; break; }                      /* the look-ahead block belongs to the next range */
		kprev = k; iprev = i; memcpy(rprev, run, sizeof(run));
	}
	if (kprev && kprev < f->n_frames) {                            /* the very last block of the stream */
		f->frame[kprev * N_FIELDS] = iprev;
		for (j = 0; j < 6; ++j) f->frame[kprev * N_FIELDS + 1 + j] = rprev[j];
	}
	return 0;
}

static void fmd_index_mt(rb2_fmd_t *f, int nthr)
{
	uint64_t n_blks, last, k, nb, per, acc[6] = { 0, 0, 0, 0, 0, 0 };
	int ibits, t, j, pass;
	idx_job_t *jobs;
	pthread_t *th;
	f->n_bytes = (uint64_t)f->p * 8;
	n_blks = f->n_bytes * 8 / 64 / BLK_WORDS + 1;
	last = (f->n_bytes >> 3) / BLK_WORDS * BLK_WORDS;          /* word offset of the last header */
	ibits = ilog2_u64((uint64_t)f->mcnt[0] / n_blks) + 4;
	f->n_frames = (((uint64_t)f->mcnt[0] + (1ull << ibits) - 1) >> ibits) + 1;
	f->frame = (uint64_t*)calloc(f->n_frames * N_FIELDS, 8);
	nb = last / BLK_WORDS;                                     /* headers at words 8, 16, ..., last */
	if (nthr < 1) nthr = 1;
	if (nb < 4096) nthr = 1;
	per = (nb + nthr - 1) / nthr;
	jobs = (idx_job_t*)calloc(nthr, sizeof(idx_job_t)); th = (pthread_t*)calloc(nthr, sizeof(pthread_t));
	for (pass = 0; pass < 2; ++pass) {
		for (t = 0; t < nthr; ++t) {
			idx_job_t *jb = &jobs[t];
			uint64_t b0 = (uint64_t)t * per, b1 = b0 + per < nb ? b0 + per : nb;
			if (b0 > nb) b0 = nb;
			jb->f = f; jb->i0 = (b0 + 1) * BLK_WORDS; jb->i1 = (b1 + 1) * BLK_WORDS; jb->last = last; jb->ibits = ibits; jb->pass = pass;
			if (nthr == 1) idx_worker(jb); else pthread_create(&th[t], 0, idx_worker, jb);
		}
		if (nthr > 1) for (t = 0; t < nthr; ++t) pthread_join(th[t], 0);
		if (pass == 0) for (t = 0; t < nthr; ++t) { memcpy(jobs[t].base, acc, sizeof(acc)); for (j = 0; j < 6; ++j) acc[j] += jobs[t].sum[j]; }
	}
	free(jobs); free(th);
	for (k = 1; k < f->n_frames; ++k)                          /* empty frames repeat their predecessor */
		if (f->frame[k * N_FIELDS] == 0)
			memcpy(&f->frame[k * N_FIELDS], &f->frame[(k - 1) * N_FIELDS], N_FIELDS * 8);
	f->finished = 1;
}

static void fmd_index(rb2_fmd_t *f) { fmd_index_mt(f, 1); }

static int pwrite_all(int fd, const void *buf, size_t n, int64_t off)
{
	const char *q = (const char*)buf;
	while (n > 0) {
		const ssize_t k = pwrite(fd, q, n > ((size_t)1 << 30) ? (size_t)1 << 30 : n, (off_t)off);
		if (k < 0) { if (errno == EINTR) continue; return -1; }
		q += k; n -= (size_t)k; off += k;
	}
	return 0;
}

/* the stream was written to out_fd while it was produced (rb2_fmdp_set_output): header, the words that became final after the
 * writer thread stopped, and the rank frames; fp (the FILE on the same descriptor) is left positioned behind the file */
static int fmd_write_rest(const rb2_fmd_t *f, FILE *fp)
{
	const uint32_t a = 6u << 16 | 3u;
	uint8_t hdr[80];
	const size_t nw = f->n_bytes / 8;
	int r = f->out_err ? -1 : 0;
	memset(hdr, 0, sizeof(hdr));
	memcpy(hdr, "RLD\3", 4); memcpy(hdr + 4, &a, 4);
	memcpy(hdr + 16, &f->n_bytes, 8); memcpy(hdr + 24, &f->n_frames, 8); memcpy(hdr + 32, f->mcnt + 1, 48);
	if (fflush(fp) != 0) r = -1;
	if (pwrite_all(f->out_fd, hdr, 80, f->out_base) != 0) r = -1;
	if (f->out_done < nw && pwrite_all(f->out_fd, f->w + f->out_done, (nw - f->out_done) * 8, f->out_base + 80 + (int64_t)f->out_done * 8) != 0) r = -1;
	if (pwrite_all(f->out_fd, f->frame, f->n_frames * 8 * N_FIELDS, f->out_base + 80 + (int64_t)nw * 8) != 0) r = -1;
	if (fseeko(fp, (off_t)(f->out_base + 80 + (int64_t)nw * 8 + (int64_t)f->n_frames * 8 * N_FIELDS), SEEK_SET) != 0) r = -1;
	return r;
}

int rb2_fmd_write(const rb2_fmd_t *f, FILE *fp)
{
	const uint32_t a = 6u << 16 | 3u;
	const uint64_t zero = 0;
	if (!f->finished) return -1;
	if (f->out_fd >= 0) return fmd_write_rest(f, fp);
	fwrite("RLD\3", 1, 4, fp);
	fwrite(&a, 4, 1, fp);
	fwrite(&zero, 8, 1, fp);
	fwrite(&f->n_bytes, 8, 1, fp);
	fwrite(&f->n_frames, 8, 1, fp);
	fwrite(f->mcnt + 1, 8, 6, fp);
	fwrite(f->w, 8, f->n_bytes / 8, fp);
	fwrite(f->frame, 8 * N_FIELDS, f->n_frames, fp);
	return ferror(fp) ? -1 : 0;
}

void rb2_fmd_counts(const rb2_fmd_t *f, int64_t c[7])
{
	memcpy(c, f->mcnt, sizeof(f->mcnt));
}

void rb2_fmd_destroy(rb2_fmd_t *f)
{
	if (!f) return;
	free(f->w); free(f->frame); free(f);
}

int rb2_fmd_write_path(const rb2_fmd_t *f, const char *path)     /* rb2_fmd_write to a named file (bindings without FILE*) */
{
	FILE *fp = fopen(path, "wb");
	int r;
	if (!fp) return -1;
	r = rb2_fmd_write(f, fp);
	if (fclose(fp) != 0) r = -1;
	return r;
}

/* ===============================================================================================
 * Parallel writer.  The greedy block packing of the format (a run goes into the current 8-word block if it fits, else opens
 * the next one, rld0.c:137-151) is a sequential automaton, but its state is short-lived: two encoders that start on the same
 * run stream at different points produce the same blocks as soon as they open a block at the same run -- which happens
 * after a few thousand blocks (the offset between them performs a random walk of a few bits per block).  So:
 *   - the run stream is cut into segments; worker threads encode every segment SPECULATIVELY, as if a block started at
 *     its first run (no chunk rule), and record the stream offset at which each of their blocks starts;
 *   - one thread follows the TRUE orbit: it encodes from where the valid output ends until it opens a block at an offset
 *     where the speculative encoding of that segment also opens one (with the same header width); from there the
 *     speculative blocks are the true ones and are copied wholesale (block contents do not depend on their position: headers
 *     hold the counts of the previous block only), up to the segment's last complete block or the next block that the
 *     chunk rule shortens (the last block of every 2^23-word chunk, rld0.h:75) -- then it encodes again, and so on.
 * The output is byte-identical to the sequential writer's (tests/test_host_layer.py); the rank frames are built afterwards
 * from the block headers as before.
 * =============================================================================================== */

typedef struct {
	uint8_t *runs; int64_t n;        /* run bytes of the segment (whole runs) */
	int prev_sym;                    /* symbol of the last run of the previous segment (-1: none) */
	int prev_sym_out;                /* symbol of this segment's last run (taken before the bytes are freed) */
	rb2_fmd_t *sp;                   /* speculative encoding */
	int64_t sum[N_FIELDS];           /* symbols in the segment (total, then per symbol) */
	int state;                       /* 0 filling, 1 queued, 2 being encoded, 3 encoded */
} fseg_t;

struct rb2_fmdp_s {
	rb2_fmd_t *f;                    /* the true stream */
	/* streamed output: the writer thread puts the words below pub_head (final: the true orbit has left those blocks) into the file */
	pthread_t writer; int writer_on, writer_quit; size_t pub_head;
	pthread_mutex_t wmu, pmu; pthread_cond_t pcv;
	fseg_t **seg; int64_t nseg, cap_seg, seg_bytes;          /* segments are allocated one by one: workers keep pointers to them */
	int64_t n_queued, next_work;     /* segments handed to the workers / next one a worker takes */
	int64_t cur_seg, cur_pos;        /* input cursor of the true orbit */
	int64_t tot[N_FIELDS];
	int64_t n_copied_blocks, n_true_blocks;
	double t_stitch_wait, t_stitch_work, t_push_wait;        /* RB2_FMD_STATS: who waited for whom */
	int stats;
	pthread_t *thr; int nthr, closing;
	pthread_t stitcher; int no_more;   /* the true orbit has a thread of its own: it copies every coupled block (6 GB at configs[1]) while the producer copies run bytes into segments */
	pthread_mutex_t mu; pthread_cond_t cv_work, cv_done, cv_space;
};

static inline int run_at(const uint8_t *q, int *c, int64_t *l)    /* one run of the 43+3 codec; returns its bytes */
{
	if ((*q & 0x80) == 0) { *c = *q & 7; *l = *q >> 3; return 1; }
	return rle_dec1_fn(q, c, l);
}

static inline uint64_t run_code(int64_t l, int c, int *w)          /* Elias delta of l + 3 bits of c (rld0.c:45-51) */
{
	const int y = ilog2_u64((uint64_t)l), z = ilog2_u64((uint64_t)y + 1);
	*w = 2 * z + 1 + y + SYM_BITS;
	return ((((uint64_t)l ^ (1ull << y)) | (uint64_t)(y + 1) << y) << SYM_BITS) | (uint64_t)c;
}

static inline void place_bits(rb2_fmd_t *f, uint64_t x, int w)
{
	if (w > f->r) {
		w -= f->r;
		f->w[f->p++] |= x >> w;
		f->r = 64 - w;
		f->w[f->p] = x << f->r;
	} else {
		f->r -= w;
		f->w[f->p] |= x << f->r;
	}
}

static void spec_encode(fseg_t *sg)
{
	rb2_fmd_t *f = (rb2_fmd_t*)calloc(1, sizeof(rb2_fmd_t));
	const uint8_t *q = sg->runs;
	int64_t i = 0, n = sg->n, pl = 0, ppos = 0, l;
	int c, pc = -1;
	f->spec = 1;
	reserve(f, (size_t)(n / 4 + 4 * BLK_WORDS));             /* ~5 bits per run byte on random reads; grows when needed */
	f->cap_blk = (size_t)(n / 256 + 1024);
	f->start = (uint32_t*)malloc(f->cap_blk * 4); f->type = (uint8_t*)malloc(f->cap_blk);
	memset(f->w, 0, BLK_WORDS * 8);                           /* (reserve does not zero: blocks are zeroed as they are opened) */
	f->head = 0; f->p = hdr_words[0]; f->tail = block_tail_of(f, 0); f->r = 64;
	memset(sg->sum, 0, sizeof(sg->sum));
	while (i < n) {                                           /* the straddling run belongs to the true orbit: skip it (but count it) */
		const int nb = run_at(q + i, &c, &l);
		if (l == 0) { i += nb; continue; }
		if (c != sg->prev_sym) break;
		sg->sum[0] += l; sg->sum[c + 1] += l; i += nb;
	}
	f->start[0] = (uint32_t)i; f->type[0] = 0;
	while (i < n) {
		const int nb = run_at(q + i, &c, &l);
		if (l == 0) { i += nb; continue; }
		sg->sum[0] += l; sg->sum[c + 1] += l;
		if (c == pc) pl += l;
		else {
			if (pl) { f->run_pos = (uint32_t)ppos; encode_run(f, pl, pc); }
			pc = c; pl = l; ppos = i;
		}
		i += nb;
	}
	/* the last maximal run may continue in the next segment, and the block being filled is incomplete: both are left to the
	 * true orbit, which resumes at start[nblk] */
	sg->sp = f;
}

static void *fmdp_worker(void *arg)
{
	rb2_fmdp_t *p = (rb2_fmdp_t*)arg;
	pthread_mutex_lock(&p->mu);
	for (;;) {
		while (p->next_work >= p->n_queued && !p->closing) pthread_cond_wait(&p->cv_work, &p->mu);
		if (p->next_work >= p->n_queued) break;
		fseg_t *sg = p->seg[p->next_work++];
		sg->state = 2;
		pthread_mutex_unlock(&p->mu);
		spec_encode(sg);
		pthread_mutex_lock(&p->mu);
		sg->state = 3;
		pthread_cond_broadcast(&p->cv_done);
	}
	pthread_mutex_unlock(&p->mu);
	return 0;
}

static void *fmdp_stitcher(void *arg);
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

rb2_fmdp_t *rb2_fmdp_init(int n_threads, int64_t seg_bytes)
{
	rb2_fmdp_t *p = (rb2_fmdp_t*)calloc(1, sizeof(rb2_fmdp_t));
	int i;
	p->f = rb2_fmd_init();
	p->seg_bytes = seg_bytes > 0 ? seg_bytes : 16 << 20;
	if (p->seg_bytes > 0x7fffffff) p->seg_bytes = 0x7fffffff;     /* offsets inside a segment are 32 bit */
	p->nthr = n_threads > 0 ? n_threads : 1;
	p->stats = getenv("RB2_FMD_STATS") != 0;
	pthread_mutex_init(&p->mu, 0); pthread_cond_init(&p->cv_work, 0); pthread_cond_init(&p->cv_done, 0);
	p->thr = (pthread_t*)calloc(p->nthr, sizeof(pthread_t));
	for (i = 0; i < p->nthr; ++i) pthread_create(&p->thr[i], 0, fmdp_worker, p);
	pthread_cond_init(&p->cv_space, 0);
	pthread_create(&p->stitcher, 0, fmdp_stitcher, p);
	return p;
}

static fseg_t *cur_fill(rb2_fmdp_t *p)                        /* the segment being filled (created on demand) */
{
	if (p->nseg == p->n_queued) {
		fseg_t *sg = (fseg_t*)calloc(1, sizeof(fseg_t));
		sg->prev_sym = -1;
		if (p->nseg > 0) {                                     /* symbol of the last run before this segment */
			const fseg_t *b = p->seg[p->nseg - 1];
			sg->prev_sym = b->prev_sym_out;
		}
		sg->runs = (uint8_t*)malloc((size_t)p->seg_bytes + 2048);
		pthread_mutex_lock(&p->mu);
		if (p->nseg == p->cap_seg) {
			p->cap_seg = p->cap_seg ? p->cap_seg * 2 : 256;
			p->seg = (fseg_t**)realloc(p->seg, p->cap_seg * sizeof(fseg_t*));
		}
		p->seg[p->nseg++] = sg;
		pthread_mutex_unlock(&p->mu);
	}
	return p->seg[p->nseg - 1];
}

static void queue_fill(rb2_fmdp_t *p)                          /* hand the segment being filled to the workers */
{
	if (p->nseg == p->n_queued) return;
	{	/* symbol of the segment's last run: its head byte is the last byte that is not a continuation byte (rle.h:39-51) */
		fseg_t *sg = p->seg[p->nseg - 1];
		int64_t k = sg->n - 1;
		sg->prev_sym_out = sg->prev_sym;
		while (k >= 0) {                                          /* ... that holds symbols: empty runs do not exist for the coder (rld_enc, rld0.c:155) */
			int c; int64_t l;
			while (k > 0 && (sg->runs[k] & 0xC0) == 0x80) --k;
			run_at(sg->runs + k, &c, &l);
			if (l > 0) { sg->prev_sym_out = c; break; }
			--k;
		}
	}
	pthread_mutex_lock(&p->mu);
	p->seg[p->nseg - 1]->state = 1;
	p->n_queued = p->nseg;
	pthread_cond_signal(&p->cv_work);
	pthread_cond_broadcast(&p->cv_done);                       /* (the stitcher also waits for segments to exist) */
	if (p->n_queued - p->cur_seg > 4 * p->nthr + 8) {          /* backlog: let workers and stitcher catch up */
		const double t0 = p->stats ? now_s() : 0;
		while (p->n_queued - p->cur_seg > 4 * p->nthr + 8) pthread_cond_wait(&p->cv_space, &p->mu);
		if (p->stats) p->t_push_wait += now_s() - t0;
	}
	pthread_mutex_unlock(&p->mu);
}

/* ---- the true orbit ---- */

static void hdr_counts(const uint64_t *h, int64_t v[N_FIELDS])   /* counts stored in a block header */
{
	const int type = (int)(h[0] >> 62);
	int j;
	for (j = 0; j < N_FIELDS; ++j) {
		if (type == 0)      v[j] = (int64_t)((h[j / 4] >> (16 * (j % 4))) & 0xffffu);
		else if (type == 1) v[j] = (int64_t)((h[j / 2] >> (32 * (j % 2))) & 0x3fffffffu);   /* every field < 2^30; the width tag shares word 0 */
		else                v[j] = (int64_t)(j == 0 ? h[0] & 0x3fffffffffffffffull : h[j]);
	}
}

/* the true orbit has just opened a block (header written) for the run at offset `pos` of segment sg: if the speculative encoding
 * of the segment opened one there too, take its blocks.  Returns the offset at which the true orbit resumes, or -1. */
static int64_t try_couple(rb2_fmdp_t *p, fseg_t *sg, int64_t pos)
{
	rb2_fmd_t *f = p->f, *sp = sg->sp;
	size_t lo = 0, hi = sp->nblk, j, e, k;
	int64_t prev[N_FIELDS];
	int i;
	if (sp->nblk == 0 || f->tail != f->head + BLK_WORDS - 1) return -1;
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if ((int64_t)sp->start[mid] < pos) lo = mid + 1; else hi = mid; }
	j = lo;
	if (j >= sp->nblk || (int64_t)sp->start[j] != pos || (int)(f->w[f->head] >> 62) != sp->type[j]) return -1;
	/* blocks j .. e-1 are copied; block e is opened by the true orbit again (it is the partial one, or the chunk rule shortens it) */
	for (e = j + 1; e < sp->nblk; ++e) {
		const size_t ah = f->head + (e - j) * BLK_WORDS;
		if (block_tail(ah) != ah + BLK_WORDS - 1) break;
	}
	reserve(f, f->head + (e - j + 2) * BLK_WORDS);
	k = hdr_words[sp->type[j]];
	memcpy(f->w + f->head + k, sp->w + j * BLK_WORDS + k, (BLK_WORDS - k) * 8);      /* body of block j behind the true header */
	if (e > j + 1) rb2_par_memcpy((uint8_t*)(f->w + f->head + BLK_WORDS), (const uint8_t*)(sp->w + (j + 1) * BLK_WORDS), (int64_t)((e - j - 1) * BLK_WORDS * 8));   /* (fresh pages: four threads fault them in) */
	p->n_copied_blocks += (int64_t)(e - j);
	/* state: block e-1 is complete; its counts are in the speculative header of block e */
	hdr_counts(sp->w + e * BLK_WORDS, prev);
	f->head += (e - j - 1) * BLK_WORDS;
	for (i = 0; i < N_FIELDS; ++i) f->cnt[i] = f->mcnt[i] + prev[i];
	f->p = f->tail = f->head + BLK_WORDS - 1; f->r = 0;       /* full: the next run opens block e through open_next_block */
	return (int64_t)sp->start[e];
}

/* advance the true orbit over everything that is encoded speculatively so far (final: over everything, waiting for the workers) */
static void stitch(rb2_fmdp_t *p)                             /* body of the stitcher thread: segment after segment until the stream ends */
{
	rb2_fmd_t *f = p->f;
	for (;;) {
		fseg_t *sg;
		const uint8_t *q;
		int64_t i, n;
		const double tw0 = p->stats ? now_s() : 0;
		double tw1;
		pthread_mutex_lock(&p->mu);
		while (p->cur_seg >= p->n_queued && !p->no_more) pthread_cond_wait(&p->cv_done, &p->mu);
		if (p->cur_seg >= p->n_queued) { pthread_mutex_unlock(&p->mu); return; }
		sg = p->seg[p->cur_seg];
		while (sg->state != 3) pthread_cond_wait(&p->cv_done, &p->mu);
		pthread_mutex_unlock(&p->mu);
		tw1 = p->stats ? now_s() : 0;
		p->t_stitch_wait += tw1 - tw0;
		q = sg->runs; n = sg->n; i = p->cur_pos;
		if (i == 0) { int k; for (k = 0; k < N_FIELDS; ++k) p->tot[k] += sg->sum[k]; }
		if (p->cur_seg == 0 && i == 0 && sg->sp->nblk > 0 && f->p == (size_t)hdr_words[0] && f->r == 64) {
			/* the very first block of the stream is a block start for both by construction */
			const int64_t r = try_couple(p, sg, (int64_t)sg->sp->start[0]);
			if (r >= 0) i = r;
		}
		while (i < n) {
			int c, w; int64_t l;
			const int nb = run_at(q + i, &c, &l);
			if (l == 0) { i += nb; continue; }                  /* (an empty run is no run: it must not separate two runs of one symbol) */
			if (c == f->pend_c) { f->pend_l += l; i += nb; continue; }
			if (f->pend_l) {                                   /* flush the pending maximal run: it started at f->run_pos of segment pend_seg */
				const uint64_t x = run_code(f->pend_l, f->pend_c, &w);
				if (w >= f->r && f->p == f->tail) {
					open_next_block(f);
					++p->n_true_blocks;
					if (f->pend_seg == p->cur_seg) {             /* the pending run started in THIS segment: its offset can be looked up */
						const int64_t r = try_couple(p, sg, (int64_t)f->run_pos);
						if (r >= 0) { f->pend_c = -1; f->pend_l = 0; i = r; continue; }
					}
				}
				place_bits(f, x, w);
				f->cnt[0] += f->pend_l; f->cnt[f->pend_c + 1] += f->pend_l;
			}
			f->pend_c = c; f->pend_l = l; f->run_pos = (uint32_t)i; f->pend_seg = p->cur_seg;
			i += nb;
		}
		if (p->stats) p->t_stitch_work += now_s() - tw1;
		free(sg->runs); sg->runs = 0;
		free(sg->sp->w); free(sg->sp->start); free(sg->sp->type); free(sg->sp); sg->sp = 0;
		if (p->writer_on) {                                   /* everything below the block being filled is final */
			pthread_mutex_lock(&p->pmu);
			p->pub_head = f->head;
			pthread_cond_signal(&p->pcv);
			pthread_mutex_unlock(&p->pmu);
		}
		pthread_mutex_lock(&p->mu);
		++p->cur_seg; p->cur_pos = 0;                        /* (the fseg_t itself stays: the next segment reads prev_sym_out) */
		pthread_cond_broadcast(&p->cv_space);
		pthread_mutex_unlock(&p->mu);
	}
}

static void *fmdp_stitcher(void *arg) { stitch((rb2_fmdp_t*)arg); return 0; }

/* The index will hold n_symbols symbols: size the output array once (a run of one symbol costs 4 bits -- Elias delta of 1 is one
 * bit -- and a block spends 2 of its 8 words on counts: at most 0.67 bytes per symbol; pages that are never written are never
 * backed) instead of growing it by half again and again, and ask for huge pages: the stitcher faults all of it in.  Call before the
 * first rb2_fmdp_push_runs; without it the array grows on demand as before. */
void rb2_fmdp_expect(rb2_fmdp_t *p, int64_t n_symbols)
{
	if (n_symbols <= 0 || p->f->head != 0) return;
	reserve(p->f, (size_t)(n_symbols / 8 * 0.7) + (1 << 16));
	rb2_hint_huge_which(p->f->w, p->f->cap * 8, 8);
}

static size_t out_step_words(void)                            /* the writer thread moves at least 32 MiB per pwrite (RB2_FMD_OUT_STEP: words, tests) */
{
	const char *e = getenv("RB2_FMD_OUT_STEP");
	const long v = e ? atol(e) : 0;
	return v > 0 ? (size_t)v : (size_t)4 << 20;
}

static void *fmdp_writer(void *arg)
{
	rb2_fmdp_t *p = (rb2_fmdp_t*)arg;
	rb2_fmd_t *f = p->f;
	const size_t OUT_STEP_WORDS = out_step_words();
	for (;;) {
		size_t upto; int quit;
		pthread_mutex_lock(&p->pmu);
		while (p->pub_head < f->out_done + OUT_STEP_WORDS && !p->writer_quit) pthread_cond_wait(&p->pcv, &p->pmu);
		upto = p->pub_head; quit = p->writer_quit;
		pthread_mutex_unlock(&p->pmu);
		if (upto > f->out_done && !f->out_err) {
			pthread_mutex_lock(&p->wmu);                       /* f->w does not move while it is being read */
			if (pwrite_all(f->out_fd, f->w + f->out_done, (upto - f->out_done) * 8, f->out_base + 80 + (int64_t)f->out_done * 8) != 0) f->out_err = 1;
			pthread_mutex_unlock(&p->wmu);
			f->out_done = upto;
		}
		if (quit) return 0;
	}
}

/* Stream the .fmd to fd while it is being encoded: the file starts at byte `offset` of fd (a regular file opened without
 * O_APPEND; returns -1 and changes nothing otherwise).  rb2_fmd_write() on the finished index then only adds the header, the
 * tail of the stream and the rank frames.  Call it before the first rb2_fmdp_push_runs. */
int rb2_fmdp_set_output(rb2_fmdp_t *p, int fd, int64_t offset)
{
	struct stat st;
	const int fl = fcntl(fd, F_GETFL);
	if (p->writer_on || fl < 0 || (fl & O_APPEND) || (fl & O_ACCMODE) == O_RDONLY) return -1;
	if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || offset < 0) return -1;
	pthread_mutex_init(&p->wmu, 0); pthread_mutex_init(&p->pmu, 0); pthread_cond_init(&p->pcv, 0);
	p->f->out_fd = fd; p->f->out_base = offset; p->f->out_done = 0; p->f->wmu = &p->wmu;
	p->writer_on = 1;
	pthread_create(&p->writer, 0, fmdp_writer, p);
	return 0;
}

void rb2_fmdp_push_runs(rb2_fmdp_t *p, const uint8_t *runs, int64_t n)
{
	while (n > 0) {
		fseg_t *sg = cur_fill(p);
		int64_t take = n;
		if (take > p->seg_bytes + 1024 - sg->n) {              /* a chunk larger than a segment: cut at a run boundary */
			take = p->seg_bytes + 1024 - sg->n;
			while (take > 0 && (runs[take] & 0xC0) == 0x80) --take;
			if (take == 0) { queue_fill(p); continue; }
		}
		rb2_par_memcpy(sg->runs + sg->n, runs, take); sg->n += take;   /* (four threads: this copy was the producer's whole time) */
		runs += take; n -= take;
		if (sg->n >= p->seg_bytes) queue_fill(p);
	}
}

rb2_fmd_t *rb2_fmdp_finish(rb2_fmdp_t *p)
{
	rb2_fmd_t *f = p->f;
	int i;
	queue_fill(p);
	pthread_mutex_lock(&p->mu);
	p->no_more = 1;
	pthread_cond_broadcast(&p->cv_done);
	pthread_mutex_unlock(&p->mu);
	pthread_join(p->stitcher, 0);
	if (p->writer_on) {                                       /* the writer drains what is published and stops; the rest goes out with rb2_fmd_write */
		pthread_mutex_lock(&p->pmu);
		p->writer_quit = 1;
		pthread_cond_signal(&p->pcv);
		pthread_mutex_unlock(&p->pmu);
		pthread_join(p->writer, 0);
		f->wmu = 0;
		pthread_mutex_destroy(&p->wmu); pthread_mutex_destroy(&p->pmu); pthread_cond_destroy(&p->pcv);
	}
	pthread_mutex_lock(&p->mu);
	p->closing = 1;
	pthread_cond_broadcast(&p->cv_work);
	pthread_mutex_unlock(&p->mu);
	for (i = 0; i < p->nthr; ++i) pthread_join(p->thr[i], 0);
	/* tail of the stream, as rb2_fmd_finish does it -- but the running counts of the true orbit are only meaningful as
	 * differences, the totals come from the segments */
	if (f->pend_l) { encode_run(f, f->pend_l, f->pend_c); f->pend_l = 0; }
	open_next_block(f);
	memcpy(f->mcnt, p->tot, sizeof(p->tot)); memcpy(f->cnt, p->tot, sizeof(p->tot));
	fmd_index_mt(f, p->nthr);
	if (p->stats) fprintf(stderr, "[rb2_fmdp] %lld segments, %lld blocks copied from the speculative encodings, %lld encoded by the true orbit; "
			"stitcher: %.3f s working, %.3f s waiting for segments; producer: %.3f s waiting for room (%d workers)\n",
			(long long)p->nseg, (long long)p->n_copied_blocks, (long long)p->n_true_blocks, p->t_stitch_work, p->t_stitch_wait, p->t_push_wait, p->nthr);
	pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->cv_work); pthread_cond_destroy(&p->cv_done); pthread_cond_destroy(&p->cv_space);
	{ int64_t k; for (k = 0; k < p->nseg; ++k) free(p->seg[k]); }
	free(p->thr); free(p->seg); free(p);
	return f;
}
