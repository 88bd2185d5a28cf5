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
#include "rb2_fmd.h"
#include "rle.h"

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
};

static const int hdr_words[3] = { 2, 4, 7 };   /* (7*16+63)/64, (7*32+63)/64, 7 */

static int ilog2_u64(uint64_t v) { return v ? 63 - __builtin_clzll(v) : -1; }   /* ilog2(0) = -1 like rld0.c:26-43 */

static void reserve(rb2_fmd_t *f, size_t n_words)
{
	if (n_words <= f->cap) return;
	size_t nc = f->cap ? f->cap : 1 << 16;
	while (nc < n_words) nc += nc >> 1;
	f->w = (uint64_t*)realloc(f->w, nc * 8);
	memset(f->w + f->cap, 0, (nc - f->cap) * 8);
	f->cap = nc;
}

static size_t block_tail(size_t head)
{
	const size_t end = head + BLK_WORDS;                      /* one past the block */
	return end % CHUNK_WORDS == 0 ? end - 2 : end - 1;        /* last block of a chunk is one word shorter (rld0.h:75) */
}

rb2_fmd_t *rb2_fmd_init(void)
{
	rb2_fmd_t *f = (rb2_fmd_t*)calloc(1, sizeof(rb2_fmd_t));
	reserve(f, 2 * BLK_WORDS);
	f->head = 0; f->p = hdr_words[0]; f->tail = block_tail(0); f->r = 64;
	f->pend_c = -1;
	return f;
}

/* close the current block: the next block's header receives the counts gathered in this one (rld0.c:107-135) */
static void open_next_block(rb2_fmd_t *f)
{
	int i, type;
	const int64_t tot = f->cnt[0] - f->mcnt[0];
	f->head += BLK_WORDS;
	reserve(f, f->head + 2 * BLK_WORDS);
	type = tot < 0x4000 ? 0 : tot < 0x40000000 ? 1 : 2;
	for (i = 0; i < N_FIELDS; ++i) {
		const uint64_t v = (uint64_t)(f->cnt[i] - f->mcnt[i]);
		if (type == 0)      f->w[f->head + i / 4] |= (v & 0xffffu) << (16 * (i % 4));
		else if (type == 1) f->w[f->head + i / 2] |= (v & 0xffffffffu) << (32 * (i % 2));
		else                f->w[f->head + i] = v;
	}
	f->w[f->head] |= (uint64_t)type << 62;
	f->p = f->head + hdr_words[type];
	f->tail = block_tail(f->head);
	f->r = 64;
	memcpy(f->mcnt, f->cnt, sizeof(f->cnt));
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

void rb2_fmd_finish(rb2_fmd_t *f)
{
	uint64_t n_blks, last, i, k, run[6] = { 0, 0, 0, 0, 0, 0 };
	int ibits, j;
	if (f->finished) return;
	if (f->pend_l) encode_run(f, f->pend_l, f->pend_c);
	f->pend_l = 0;
	open_next_block(f);
	f->n_bytes = (uint64_t)f->p * 8;
	/* rank index (rld0.c:163-205) */
	n_blks = f->n_bytes * 8 / 64 / BLK_WORDS + 1;
	last = (f->n_bytes >> 3) / BLK_WORDS * BLK_WORDS;
	ibits = ilog2_u64((uint64_t)f->mcnt[0] / n_blks) + 4;
	f->n_frames = (((uint64_t)f->mcnt[0] + (1ull << ibits) - 1) >> ibits) + 1;
	f->frame = (uint64_t*)calloc(f->n_frames * N_FIELDS, 8);
	for (i = BLK_WORDS, k = 1; i <= last; i += BLK_WORDS) {
		const uint64_t *h = f->w + i;
		const int type = (int)(h[0] >> 62);
		uint64_t sum = 0;
		for (j = 1; j < N_FIELDS; ++j) {
			uint64_t v;
			if (type == 0)      v = (h[j / 4] >> (16 * (j % 4))) & 0xffffu;
			else if (type == 1) v = (h[j / 2] >> (32 * (j % 2))) & 0x3fffffffu;
			else                v = h[j];
			run[j - 1] += v;
		}
		for (j = 0; j < 6; ++j) sum += run[j];
		while (sum >= k << ibits) ++k;
		if (k < f->n_frames) {
			f->frame[k * N_FIELDS] = i;
			for (j = 0; j < 6; ++j) f->frame[k * N_FIELDS + 1 + j] = run[j];
		}
	}
	assert(k >= f->n_frames - 1);
	for (k = 1; k < f->n_frames; ++k)                          /* empty frames repeat their predecessor */
		if (f->frame[k * N_FIELDS] == 0)
			memcpy(&f->frame[k * N_FIELDS], &f->frame[(k - 1) * N_FIELDS], N_FIELDS * 8);
	f->finished = 1;
}

int rb2_fmd_write(const rb2_fmd_t *f, FILE *fp)
{
	const uint32_t a = 6u << 16 | 3u;
	const uint64_t zero = 0;
	if (!f->finished) return -1;
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
