/* mrope.c -- multi-rope layer: the boundary between ropebwt2-style callers and the GPU engine.
 *
 * Implements include/mrope.h (the API of /root/reference/mrope.h).  mr_insert_multi -- the hot
 * path, mrope.c:258-345 in the reference -- is a thin shim over rb2_hip_insert_multi(): plain C
 * on this side, HIP on the other side of a C ABI.  There is deliberately no CPU implementation of
 * the batch insertion here: without a GPU the call aborts inside rb2_hip_create().
 *
 * State: the BWT lives either in the host ropes, in HBM, or both.
 *   host_ok  the six host ropes hold the current BWT (leaves + tree)
 *   dev_ok   the device holds the current BWT
 * r[a]->c[] (read directly by the inline helpers of mrope.h) is kept current in both states.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <assert.h>
#include <pthread.h>
#include "mrope.h"
#include "rle.h"
#include "rb2_hip.h"

typedef struct {
	mrope_t pub;            /* must stay first: callers hold mrope_t* */
	rb2_hip_t *dev;
	int host_ok, dev_ok;
	int max_nodes, block_len;
} mrx_t;

static mrx_t *X(const mrope_t *mr) { return (mrx_t*)mr; }

static int device_id(void)
{
	const char *e = getenv("RB2_HIP_DEVICE");
	return e ? atoi(e) : 0;
}

mrope_t *mr_init(int max_nodes, int block_len, int sorting_order)
{
	mrx_t *x;
	int a;
	assert(sorting_order >= 0 && sorting_order <= 2);           /* mrope.c:18 */
	x = (mrx_t*)calloc(1, sizeof(mrx_t));
	x->pub.so = (uint8_t)sorting_order;
	x->pub.thr_min = 1000;                                      /* mrope.c:21 */
	x->max_nodes = max_nodes; x->block_len = block_len;
	for (a = 0; a < 6; ++a) x->pub.r[a] = rope_init(max_nodes, block_len);
	x->host_ok = 1; x->dev_ok = 0;
	return &x->pub;
}

void mr_destroy(mrope_t *mr)
{
	int a;
	if (!mr) return;
	for (a = 0; a < 6; ++a) if (mr->r[a]) rope_destroy(mr->r[a]);   /* r[a] may be NULL after a freeing iteration */
	if (X(mr)->dev) rb2_hip_destroy(X(mr)->dev);
	free(mr);
}

int mr_thr_min(mrope_t *mr, int thr_min)
{
	if (thr_min > 0) mr->thr_min = thr_min;
	return mr->thr_min;
}

void *mr_hip_handle(mrope_t *mr) { return X(mr)->dev; }

/* ---- host <-> device ------------------------------------------------------------------------- */

/* device -> host ropes: stream each rope's run bytes off the device ONCE into a growing buffer (the size is only known
 * after the export; asking for it first would run the export kernel twice) and bulk-load a fresh B+ tree */
typedef struct { uint8_t *p; int64_t n, m; } runbuf_t;
static void runbuf_add(void *user, const uint8_t *q, int64_t n)
{
	runbuf_t *b = (runbuf_t*)user;
	if (b->n + n > b->m) { b->m = (b->n + n) + ((b->n + n) >> 1) + (1 << 20); b->p = (uint8_t*)realloc(b->p, b->m); }
	memcpy(b->p + b->n, q, n); b->n += n;
}

typedef struct { rope_t *r; runbuf_t rb; } load_job_t;
static void *load_worker(void *arg)
{
	load_job_t *j = (load_job_t*)arg;
	rope_load_runs(j->r, j->rb.p, j->rb.n);
	free(j->rb.p); j->rb.p = 0;
	return 0;
}

void mr_sync_host(mrope_t *mr)
{
	mrx_t *x = X(mr);
	load_job_t job[6];
	pthread_t th[6];
	int a;
	if (x->host_ok) return;
	assert(x->dev && x->dev_ok);
	/* the six ropes are independent trees: each is bulk-loaded by its own thread as soon as its run bytes have arrived, while
	 * the next rope is still streaming off the device (the reference has nothing to do here: its ropes were built on the host) */
	for (a = 0; a < 6; ++a) {
		memset(&job[a], 0, sizeof(job[a]));
		rb2_hip_stream_rope(x->dev, a, runbuf_add, &job[a].rb);
		if (!mr->r[a]) mr->r[a] = rope_init(x->max_nodes, x->block_len);
		job[a].r = mr->r[a];
		pthread_create(&th[a], 0, load_worker, &job[a]);
	}
	for (a = 0; a < 6; ++a) pthread_join(th[a], 0);
	x->host_ok = 1;
}

/* the BWT as run bytes, without materialising host trees when the device has it (include/mrope.h) */
void mr_stream_runs(mrope_t *mr, void (*cb)(void *user, const uint8_t *runs, int64_t n_bytes), void *user)
{
	mrx_t *x = X(mr);
	int a;
	if (!x->host_ok && x->dev && x->dev_ok) {
		for (a = 0; a < 6; ++a) rb2_hip_stream_rope(x->dev, a, cb, user);
	} else {
		mritr_t itr;
		const uint8_t *blk;
		mr_itr_first(mr, &itr, 0);
		while ((blk = mr_itr_next_block(&itr)) != 0) cb(user, blk + 2, *rle_nptr(blk));
	}
}

/* host ropes -> device */
static void sync_dev(mrope_t *mr)
{
	mrx_t *x = X(mr);
	uint8_t *rle[6]; int64_t nb[6]; int a;
	if (!x->dev) x->dev = rb2_hip_create(device_id(), mr->so);
	if (x->dev_ok) return;
	assert(x->host_ok);
	for (a = 0; a < 6; ++a) nb[a] = rope_export_runs(mr->r[a], &rle[a]);
	rb2_hip_load_ropes(x->dev, (const uint8_t *const*)rle, nb);
	for (a = 0; a < 6; ++a) free(rle[a]);
	x->dev_ok = 1;
}

/* ---- the hot path ---------------------------------------------------------------------------- */

void mr_insert_multi(mrope_t *mr, int64_t len, const uint8_t *s, int is_thr)
{
	mrx_t *x = X(mr);
	int64_t c[36];
	int a, b;
	(void)is_thr;                                               /* the reference's 5-thread switch has no meaning here */
	assert(len > 0 && s[len-1] == 0);                           /* mrope.c:268 */
	sync_dev(mr);
	rb2_hip_insert_multi(x->dev, len, s);
	rb2_hip_get_counts(x->dev, c);
	for (a = 0; a < 6; ++a)
		for (b = 0; b < 6; ++b) mr->r[a]->c[b] = c[a*6+b];      /* keep mr_get_c()/mr_get_ac() truthful */
	x->host_ok = 0;
}

/* ---- CPU-side operations (all need the host ropes) ---------------------------------------------- */

int64_t mr_insert1(mrope_t *mr, const uint8_t *str)
{	/* Algorithm 1/2 of the paper, one rank + one insert per symbol (mrope.c:42-68) */
	const int sorted = mr->so != MR_SO_IO, rc = mr->so == MR_SO_RCLO;
	int64_t l, u, tl[6], tu[6];
	const uint8_t *p;
	int a, b;
	mr_sync_host(mr);
	X(mr)->dev_ok = 0;
	for (u = 0, b = 0; b < 6; ++b) u += mr->r[b]->c[0];        /* strings so far */
	l = sorted ? 0 : u;
	for (p = str, b = 0; *p; b = *p++) {
		const int c = *p;
		int64_t before = 0;
		int bb;
		if (l != u) {                                           /* non-empty interval: place c in symbol order */
			rope_rank2a(mr->r[b], l, u, tl, tu);
			if (rc && c != 5) {                                 /* $ T G C A N */
				l += tu[0] - tl[0];
				for (a = 4; a > c; --a) l += tu[a] - tl[a];
			} else for (a = 0; a < c; ++a) l += tu[a] - tl[a];
			rope_insert_run(mr->r[b], l, c, 1, 0);
			for (bb = 0; bb < b; ++bb) before += mr->r[bb]->c[c];
			l = before + tl[c]; u = before + tu[c];
		} else {
			l = rope_insert_run(mr->r[b], l, c, 1, 0);
			for (bb = 0; bb < b; ++bb) l += mr->r[bb]->c[c];
			u = l;
		}
	}
	return rope_insert_run(mr->r[b], l, 0, 1, 0);
}

void mr_rank2a(const mrope_t *mr, int64_t x, int64_t y, int64_t *cx, int64_t *cy)
{	/* the six ropes are one sequence $,A,C,G,T,N (mrope.c:70-105).  While the BWT only lives in HBM the in-rope
	 * rank is asked from the device (rb2_hip_rank1a) instead of materialising the host trees for a few queries. */
	int a, b, pass;
	const mrx_t *xx = X(mr);
	const int on_dev = !xx->host_ok && xx->dev && xx->dev_ok;
	for (pass = 0; pass < 2; ++pass) {
		int64_t pos = pass == 0 ? x : y, *out = pass == 0 ? cx : cy, z = 0, acc[6] = { 0, 0, 0, 0, 0, 0 };
		if (pass == 1 && (cy == 0 || y < 0)) break;
		for (a = 0; a < 6; ++a) {
			const int64_t *ca = mr->r[a]->c;
			const int64_t len = ca[0] + ca[1] + ca[2] + ca[3] + ca[4] + ca[5];
			if (z + len >= pos) break;
			for (b = 0; b < 6; ++b) acc[b] += ca[b];
			z += len;
		}
		assert(a < 6);
		if (pos == z) memset(out, 0, 48);
		else if (on_dev) rb2_hip_rank1a(xx->dev, a, pos - z, out);
		else rope_rank1a(mr->r[a], pos - z, out);
		for (b = 0; b < 6; ++b) out[b] += acc[b];
	}
}

void mr_itr_first(mrope_t *mr, mritr_t *i, int to_free)
{
	mr_sync_host(mr);
	i->r = mr; i->a = 0; i->to_free = to_free;
	rope_itr_first(mr->r[0], &i->i);
}

const uint8_t *mr_itr_next_block(mritr_t *i)
{
	const uint8_t *blk;
	while (i->a < 6) {
		if ((blk = rope_itr_next_block(&i->i)) != 0) return blk;
		if (i->to_free) { rope_destroy(i->r->r[i->a]); i->r->r[i->a] = 0; }   /* mrope.c:122-125 */
		if (++i->a < 6) rope_itr_first(i->r->r[i->a], &i->i);
	}
	return 0;
}

void mr_print_tree(const mrope_t *mr)
{
	int a;
	mr_sync_host((mrope_t*)mr);
	for (a = 0; a < 6; ++a) rope_print_node(mr->r[a]->root);
	putchar('\n');
}

void mr_dump(mrope_t *mr, FILE *fp)
{
	int a;
	mr_sync_host(mr);
	fwrite("RB\2", 1, 3, fp);                                    /* magic; byte 3 = sorting order (mrope.c:139-140) */
	fwrite(&mr->so, 1, 1, fp);
	for (a = 0; a < 6; ++a) rope_dump(mr->r[a], fp);
}

mrope_t *mr_restore(FILE *fp)
{
	uint8_t magic[4];
	mrx_t *x;
	int64_t c[6];
	int a;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RB\2", 3) != 0 || magic[3] > 2) {
		fprintf(stderr, "[E::%s] not an FMR file\n", __func__);
		return 0;
	}
	x = (mrx_t*)calloc(1, sizeof(mrx_t));
	x->pub.so = magic[3];
	for (a = 0; a < 6; ++a) x->pub.r[a] = rope_restore(fp);
	x->max_nodes = x->pub.r[0]->max_nodes; x->block_len = x->pub.r[0]->block_len;
	x->host_ok = 1; x->dev_ok = 0;
	mr_get_c(&x->pub, c);
	fprintf(stderr, "[M::%s] ($, A, C, G, T, N) = (%ld, %ld, %ld, %ld, %ld, %ld)\n", __func__,
			(long)c[0], (long)c[1], (long)c[2], (long)c[3], (long)c[4], (long)c[5]);
	return &x->pub;
}
