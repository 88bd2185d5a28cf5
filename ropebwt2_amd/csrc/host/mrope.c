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
 *   raw_ok   neither yet: the run bytes of a restored .fmr wait in host arrays (mr_restore_runs) for whoever needs them first --
 *            the device (the first mr_insert_multi: decoded there) or a host operation (trees bulk-loaded by mr_sync_host)
 * r[a]->c[] (read directly by the inline helpers of mrope.h) is kept current in both states.
 */
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <stdio.h>
#include <assert.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include "mrope.h"
#include "rle.h"
#include "rb2_hip.h"
#define RB2_THP_WHICH 4
#include "rb2_parcopy.h"

typedef struct {
	mrope_t pub;            /* must stay first: callers hold mrope_t* */
	rb2_hip_t *dev;         /* one GPU ... */
	rb2_hip_multi_t *mdev;  /* ... or the index sharded over several (RB2_HIP_DEVICES=0,1,2,...): exactly one of the two is set */
	int host_ok, dev_ok;
	int counts_lazy;                                           /* r[a]->c[] moved on by text-derived deltas since the device's matrix was last read (reconcile_counts) */
	uint8_t *raw[6]; int64_t raw_n[6]; int raw_ok;          /* mr_restore_runs: the run bytes of the six ropes, no trees yet */
	int max_nodes, block_len;
	struct itr_stream_s *its;                                  /* mr_itr_first without host trees: the walk in progress (one at a time) */
	int consumed;                                              /* a freeing walk (mr_itr_first(.., 1)) has taken the ropes away: only mr_destroy is valid now (mrope.c:122-125) */
} mrx_t;

/* The reference destroys each rope behind a freeing iterator and nothing but mr_destroy may follow; it does not check.  Here more entry
 * points can reach the index afterwards (the lazy host / device copies): they stop with a message instead of reading freed state. */
static void check_live(const mrope_t *mr, const char *who)
{
	if (((const mrx_t*)mr)->consumed) { fprintf(stderr, "[E::%s] this index was consumed by a freeing iterator (mr_itr_first(mr, itr, 1)): only mr_destroy may follow\n", who); exit(1); }
}

static mrx_t *X(const mrope_t *mr) { return (mrx_t*)mr; }
static void itr_stream_drop(mrx_t *x);

static int device_id(void)
{
	const char *e = getenv("RB2_HIP_DEVICE");
	return e ? atoi(e) : 0;
}

/* ---- the device side: one engine, or N behind one handle -------------------------------------------------------------------
 * RB2_HIP_DEVICES=0,1,2,3,4,5,6,7 shards the index over the listed devices (the reference fans a round out to its worker
 * threads inside mr_insert_multi, mrope.c:287-296, 312-329: same call, same contract -- the workers are GPUs).  A device may
 * be listed several times (virtual ranks on one GPU: how the sharded path is tested on a one-GPU box).  RB2_HIP_TRANSPORT =
 * peer (default: peer access + device events inside this process) | rccl (RCCL's C API). */
static int has_dev(const mrx_t *x) { return x->dev != 0 || x->mdev != 0; }

static pthread_mutex_t g_dev_mu = PTHREAD_MUTEX_INITIALIZER;      /* mr_prefetch (reader thread) and mr_insert_multi (inserter thread) may both be first */
static void dev_create_locked(mrx_t *x);
static void dev_create(mrx_t *x)
{
	pthread_mutex_lock(&g_dev_mu);
	dev_create_locked(x);
	pthread_mutex_unlock(&g_dev_mu);
}
static void dev_create_locked(mrx_t *x)
{
	const char *e = getenv("RB2_HIP_DEVICES"), *t = getenv("RB2_HIP_TRANSPORT");
	int devs[RB2_MULTI_MAX_RANKS], n = 0;
	if (has_dev(x)) return;
	if (e && *e) {
		const char *p = e;
		while (*p && n < RB2_MULTI_MAX_RANKS) {
			char *q;
			const long v = strtol(p, &q, 10);
			if (q == p) break;
			devs[n++] = (int)v;
			p = *q == ',' ? q + 1 : q;
		}
	}
	if (n > 1) x->mdev = rb2_hip_multi_create(n, devs, x->pub.so, (t && (t[0] == 'r' || t[0] == 'R')) ? RB2_TRANSPORT_RCCL : RB2_TRANSPORT_PEER, 0);
	else x->dev = rb2_hip_create(n == 1 ? devs[0] : device_id(), x->pub.so);
}
static void dev_destroy(mrx_t *x)
{
	if (x->dev) rb2_hip_destroy(x->dev);
	if (x->mdev) rb2_hip_multi_destroy(x->mdev);
	x->dev = 0; x->mdev = 0;
}
static void dev_get_counts(mrx_t *x, int64_t c[36]) { if (x->mdev) rb2_hip_multi_get_counts(x->mdev, c); else rb2_hip_get_counts(x->dev, c); }
/* After a lazy insert mr->r[a]->c[] moved on by what the batch's text says it adds (k_pair_hist), not by the device's own counts.
 * Wherever the shim waits for the device anyway the two are reconciled: the device's matrix is the truth and overwrites the host's;
 * a difference (a defect in one of the two derivations) is reported -- fatally when RB2_HIP_DEBUG is set. */
static void reconcile_counts(mrx_t *x)
{
	int64_t c[36];
	int a, b, bad = 0;
	if (!x->counts_lazy || !has_dev(x) || !x->dev_ok) return;
	x->counts_lazy = 0;
	dev_get_counts(x, c);
	for (a = 0; a < 6; ++a)
		for (b = 0; b < 6; ++b) { if (x->pub.r[a]->c[b] != c[a*6+b]) bad = 1; x->pub.r[a]->c[b] = c[a*6+b]; }
	if (bad) {
		fprintf(stderr, "[W::%s] the count matrix derived from the batch text differs from the device's; the device's is kept\n", __func__);
		if (getenv("RB2_HIP_DEBUG")) abort();
	}
}
static int64_t dev_stream_rope(mrx_t *x, int a, rb2_hip_run_cb cb, void *user)
{
	return x->mdev ? rb2_hip_multi_stream_rope(x->mdev, a, cb, user) : rb2_hip_stream_rope(x->dev, a, cb, user);
}
static void dev_load_ropes(mrx_t *x, const uint8_t *const rle[6], const int64_t nb[6])
{
	if (x->mdev) rb2_hip_multi_load_ropes(x->mdev, rle, nb); else rb2_hip_load_ropes(x->dev, rle, nb);
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
	itr_stream_drop(X(mr));
	for (a = 0; a < 6; ++a) if (mr->r[a]) rope_destroy(mr->r[a]);   /* r[a] may be NULL after a freeing iteration */
	dev_destroy(X(mr));
	for (a = 0; a < 6; ++a) free(X(mr)->raw[a]);
	free(mr);
}

int mr_thr_min(mrope_t *mr, int thr_min)
{
	if (thr_min > 0) mr->thr_min = thr_min;
	return mr->thr_min;
}

void *mr_hip_handle(mrope_t *mr) { return X(mr)->dev; }
void mr_wait(mrope_t *mr) { if (X(mr)->dev) { rb2_hip_wait((rb2_hip_t*)X(mr)->dev); reconcile_counts(X(mr)); } }
void *mr_hip_multi_handle(mrope_t *mr) { return X(mr)->mdev; }

/* rb2 extension: what the caller knows about the job ahead (the size of one batch buffer, the symbols the finished index will
 * hold) -- the engine sizes its buffers once instead of growing them batch by batch (hipMalloc costs 25-40 ms per GB here) */
void mr_reserve(mrope_t *mr, int64_t batch_bytes, int64_t total_symbols)
{
	mrx_t *x = X(mr);
	dev_create(x);
	if (x->mdev) rb2_hip_multi_reserve(x->mdev, batch_bytes, 0, total_symbols);
	else rb2_hip_reserve(x->dev, batch_bytes, 0, total_symbols);
}

/* rb2 extension: the caller is still filling the buffer it will pass to mr_insert_multi next; bytes [0, n_final) are final
 * (rb2_hip_prefetch).  A no-op on a sharded index (every device takes its own copy inside the call). */
void mr_prefetch(mrope_t *mr, const uint8_t *s, int64_t n_final, int64_t capacity)
{
	mrx_t *x = X(mr);
	dev_create(x);
	if (x->dev) rb2_hip_prefetch(x->dev, s, n_final, capacity);
}

/* rb2 extension (`ropebwt2 -m auto`): a batch size in bytes that the device holding this index can take comfortably -- about an
 * eighth of its free memory (a batch costs ~2.2 bytes of HBM per byte: text, 100 B of state per string, merge scratch; the index
 * itself 0.76 B per symbol), between 1 and 40 GiB.  The BWT does not depend on the batch size. */
int64_t mr_auto_batch_bytes(mrope_t *mr)
{
	/* A heuristic.  Per listed device: free memory, minus what an index that is restored but not yet uploaded (-i old.fmr) will take
	 * there (0.8 bytes per symbol: two pool sides of 0.375 + directory; its share of it on a sharded index), an eighth of the rest;
	 * the smallest over the devices (every rank holds a full copy of the batch text). */
	int64_t fr = 0, tot = 0, m = -1, syms = 0;
	const char *e = getenv("RB2_HIP_DEVICES");
	int devs[RB2_MULTI_MAX_RANKS], n = 0, a, b, k;
	if (e && *e) {
		const char *p = e;
		while (*p && n < RB2_MULTI_MAX_RANKS) { char *q; const long v = strtol(p, &q, 10); if (q == p) break; devs[n++] = (int)v; p = *q == ',' ? q + 1 : q; }
	}
	if (n == 0) devs[n++] = device_id();
	if (mr && !(has_dev(X(mr)) && X(mr)->dev_ok))
		for (a = 0; a < 6; ++a) for (b = 0; b < 6; ++b) syms += mr->r[a]->c[b];
	for (k = 0; k < n; ++k) {
		int64_t avail;
		rb2_hip_mem_info(devs[k], &fr, &tot);
		avail = fr - (int64_t)((double)syms * 0.8 / n);
		if (avail < 0) avail = 0;
		if (m < 0 || avail / 8 < m) m = avail / 8;
	}
	if (m > ((int64_t)40 << 30)) m = (int64_t)40 << 30;
	if (m < ((int64_t)1 << 30)) m = (int64_t)1 << 30;
	return m;
}

/* ---- host <-> device ------------------------------------------------------------------------- */

/* device -> host ropes: stream each rope's run bytes off the device ONCE into a growing buffer (the size is only known
 * after the export; asking for it first would run the export kernel twice) and bulk-load a fresh B+ tree */
typedef struct { uint8_t *p; int64_t n, m; } runbuf_t;
static void runbuf_add(void *user, const uint8_t *q, int64_t n)
{
	runbuf_t *b = (runbuf_t*)user;
	if (b->n + n > b->m) {
		b->m = (b->n + n) + ((b->n + n) >> 1) + (1 << 20); b->p = (uint8_t*)realloc(b->p, b->m);
		if (b->p == 0) { fprintf(stderr, "[E::%s] out of memory (%lld bytes of run bytes)\n", "mr_sync_host", (long long)b->m); exit(1); }
	}
	rb2_par_memcpy(b->p + b->n, q, n); b->n += n;              /* (four threads: rb2_parcopy.h) */
}

typedef struct { rope_t *r; runbuf_t rb; } load_job_t;
static void *load_worker(void *arg)
{
	load_job_t *j = (load_job_t*)arg;
	int thr = 8;
	if (getenv("RB2_LOAD_THREADS")) thr = atoi(getenv("RB2_LOAD_THREADS"));
	else { const long nc = sysconf(_SC_NPROCESSORS_ONLN); thr = nc >= 80 ? 16 : (nc >= 40 ? 8 : 4); }   /* four big ropes are loaded side by side */
	rope_load_runs_mt(j->r, j->rb.p, j->rb.n, thr);
	free(j->rb.p); j->rb.p = 0;
	return 0;
}

void mr_sync_host(mrope_t *mr)
{
	check_live(mr, "mr_sync_host");
	mrx_t *x = X(mr);
	load_job_t job[6];
	pthread_t th[6];
	int a;
	if (x->host_ok) return;
	if (x->raw_ok && !(has_dev(x) && x->dev_ok)) {                  /* restored run bytes nobody has used yet: bulk-load the six trees */
		for (a = 0; a < 6; ++a) {
			memset(&job[a], 0, sizeof(job[a]));
			job[a].rb.p = x->raw[a]; job[a].rb.n = x->raw_n[a]; x->raw[a] = 0;
			rope_destroy(mr->r[a]); mr->r[a] = rope_init(x->max_nodes, x->block_len);
			job[a].r = mr->r[a];
			pthread_create(&th[a], 0, load_worker, &job[a]);
		}
		for (a = 0; a < 6; ++a) pthread_join(th[a], 0);
		x->raw_ok = 0; x->host_ok = 1;
		return;
	}
	assert(has_dev(x) && x->dev_ok);
	reconcile_counts(x);
	/* the six ropes are independent trees: each is bulk-loaded by its own thread as soon as its run bytes have arrived, while
	 * the next rope is still streaming off the device (the reference has nothing to do here: its ropes were built on the host) */
	{
	const int trace = getenv("RB2_SYNC_TRACE") != 0;
	struct timespec t0, t1;
	for (a = 0; a < 6; ++a) {
		memset(&job[a], 0, sizeof(job[a]));
		{	/* a run holds at least one symbol: the symbols of the rope bound its run bytes -- one allocation, no growing copies
			 * (pages that are never written are never backed) */
			int64_t c[36], ub = 1 << 20; int b;
			dev_get_counts(x, c);
			for (b = 0; b < 6; ++b) ub += c[a * 6 + b];
			job[a].rb.p = (uint8_t*)malloc((size_t)ub);
			job[a].rb.m = job[a].rb.p ? ub : 0;
			rb2_hint_huge(job[a].rb.p, (size_t)ub);
		}
		clock_gettime(CLOCK_MONOTONIC, &t0);
		dev_stream_rope(x, a, runbuf_add, &job[a].rb);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		if (trace) fprintf(stderr, "[mr_sync_host] rope %d: %.2f GB of runs off the device in %.3f s\n", a, job[a].rb.n / 1e9, (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9);
		if (!mr->r[a]) mr->r[a] = rope_init(x->max_nodes, x->block_len);
		job[a].r = mr->r[a];
		pthread_create(&th[a], 0, load_worker, &job[a]);
	}
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (a = 0; a < 6; ++a) pthread_join(th[a], 0);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (trace) fprintf(stderr, "[mr_sync_host] waited %.3f s for the tree builders\n", (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9);
	}
	x->host_ok = 1;
}

/* the BWT as run bytes, without materialising host trees when the device has it (include/mrope.h) */
void mr_stream_runs(mrope_t *mr, void (*cb)(void *user, const uint8_t *runs, int64_t n_bytes), void *user)
{
	mrx_t *x = X(mr);
	int a;
	if (!x->host_ok && has_dev(x) && x->dev_ok) {
		for (a = 0; a < 6; ++a) dev_stream_rope(x, a, cb, user);
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
	check_live(mr, "mr_insert_multi");
	mrx_t *x = X(mr);
	uint8_t *rle[6]; int64_t nb[6]; int a;
	dev_create(x);
	if (x->dev_ok) return;
	if (x->raw_ok) {                                            /* straight from the restored file to the device: no host trees */
		dev_load_ropes(x, (const uint8_t *const*)x->raw, x->raw_n);
		for (a = 0; a < 6; ++a) { free(x->raw[a]); x->raw[a] = 0; }
		x->raw_ok = 0; x->dev_ok = 1;
		return;
	}
	assert(x->host_ok);
	for (a = 0; a < 6; ++a) nb[a] = rope_export_runs(mr->r[a], &rle[a]);
	dev_load_ropes(x, (const uint8_t *const*)rle, nb);
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
	if (x->mdev) rb2_hip_multi_insert_multi(x->mdev, len, s);   /* N GPUs: the round loop, the count matrix and the exchange of the strings all run inside this call */
	else rb2_hip_insert_multi(x->dev, len, s);
	/* keep mr_get_c()/mr_get_ac() truthful.  One GPU: the call above may have returned with the rounds still running on the device
	 * (rb2_hip.h); what the batch adds to the counts follows from its text alone and is here already -- asking for the counts
	 * themselves would wait for the device, and a caller that parses its next batch now would not overlap with it */
	if (!x->mdev && rb2_hip_last_batch_counts(x->dev, c)) {
		for (a = 0; a < 6; ++a)
			for (b = 0; b < 6; ++b) mr->r[a]->c[b] += c[a*6+b];
		x->counts_lazy = 1;                                     /* reconciled with the device's own matrix at the next wait (reconcile_counts) */
	} else {
	dev_get_counts(x, c);
	for (a = 0; a < 6; ++a)
		for (b = 0; b < 6; ++b) mr->r[a]->c[b] = c[a*6+b];
	}
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
	const int on_dev = !xx->host_ok && has_dev(xx) && xx->dev_ok;
	if (!xx->host_ok && !on_dev) mr_sync_host((mrope_t*)mr);   /* restored run bytes: the trees are built on first use */
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
		else if (on_dev && xx->mdev) rb2_hip_multi_rank1a(xx->mdev, a, pos - z, out);
		else if (on_dev) rb2_hip_rank1a(xx->dev, a, pos - z, out);
		else rope_rank1a(mr->r[a], pos - z, out);
		for (b = 0; b < 6; ++b) out[b] += acc[b];
	}
}

/* The walk over the leaves of an index whose host trees do not exist (it lives on the device, or it was restored as run bytes):
 * the leaves follow from the run bytes alone -- canonical stream + leaf starts, rope_rdump_prepare: the very leaves the bulk
 * loader would put under a tree -- so rope a's run stream comes off the device once, is cut, and its leaves are handed out one
 * after the other from one block buffer, while a thread fetches and cuts rope a + 1.  No B+ trees (mr_host_resident() stays 0),
 * and never more than two ropes' run bytes on the host.  This is the path main.c:288-305 of the reference takes for `-d`. */
typedef struct itr_stream_s {
	mrx_t *x;
	const mritr_t *owner;                                      /* the iterator this walk belongs to: a second mr_itr_first ends it */
	rope_rdump_t *d; int64_t k, nl;                             /* the rope being served: next leaf, leaves (an empty rope has one empty leaf, as rope_init) */
	rope_rdump_t *next_d; int next_a, th_on; pthread_t th;     /* the look-ahead */
	int on_dev, nthr, to_free;
	uint8_t blk[2 + 65536 + 8];
} itr_stream_t;

static rope_rdump_t *itr_stream_cut(itr_stream_t *t, int a)
{
	mrx_t *x = t->x;
	rope_rdump_t *d;
	if (t->on_dev) {
		runbuf_t rb;
		int64_t c[36], ub = 1 << 20; int b;                    /* (as mr_sync_host: the symbols of the rope bound its run bytes) */
		memset(&rb, 0, sizeof(rb));
		dev_get_counts(x, c);
		for (b = 0; b < 6; ++b) ub += c[a * 6 + b];
		rb.p = (uint8_t*)malloc((size_t)ub); rb.m = rb.p ? ub : 0;
		rb2_hint_huge(rb.p, (size_t)ub);
		dev_stream_rope(x, a, runbuf_add, &rb);
		d = rope_rdump_prepare(rb.p, rb.n, x->max_nodes, x->block_len, t->nthr);
		free(rb.p);
	} else d = rope_rdump_prepare(x->raw[a], x->raw_n[a], x->max_nodes, x->block_len, t->nthr);
	return d;
}

static void *itr_stream_ahead(void *arg)
{
	itr_stream_t *t = (itr_stream_t*)arg;
	t->next_d = itr_stream_cut(t, t->next_a);
	return 0;
}

/* make rope a the one being served (it is the look-ahead's, or cut now) and start the look-ahead of rope a + 1 */
static void itr_stream_open(itr_stream_t *t, int a)
{
	if (t->th_on) { pthread_join(t->th, 0); t->th_on = 0; }
	if (t->next_d && t->next_a == a) { t->d = t->next_d; t->next_d = 0; }
	else t->d = itr_stream_cut(t, a);
	t->k = 0; t->nl = rope_rdump_nleaves(t->d) > 0 ? rope_rdump_nleaves(t->d) : 1;
	if (a + 1 < 6 && !getenv("RB2_ITR_NO_LOOKAHEAD")) {
		t->next_a = a + 1;
		if (pthread_create(&t->th, 0, itr_stream_ahead, t) == 0) t->th_on = 1;
	}
}

static void itr_stream_drop(mrx_t *x)
{
	itr_stream_t *t = x->its;
	if (t == 0) return;
	if (t->th_on) pthread_join(t->th, 0);
	rope_rdump_free(t->d); rope_rdump_free(t->next_d);
	free(t); x->its = 0;
}

int mr_host_resident(const mrope_t *mr) { return X(mr)->host_ok; }

void mr_itr_first(mrope_t *mr, mritr_t *i, int to_free)
{
	check_live(mr, "mr_itr_first");
	mrx_t *x = X(mr);
	itr_stream_drop(x);                                         /* a walk that was left half-way */
	i->r = mr; i->a = 0; i->to_free = to_free;
	if (!x->host_ok && ((has_dev(x) && x->dev_ok) || x->raw_ok) && !getenv("RB2_ITR_VIA_TREES")) {
		itr_stream_t *t = (itr_stream_t*)calloc(1, sizeof(itr_stream_t));
		if (t == 0) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(1); }
		t->x = x; t->owner = i; t->on_dev = has_dev(x) && x->dev_ok; t->to_free = to_free;
		if (getenv("RB2_LOAD_THREADS")) t->nthr = atoi(getenv("RB2_LOAD_THREADS"));
		else { const long nc = sysconf(_SC_NPROCESSORS_ONLN); t->nthr = nc >= 80 ? 16 : (nc >= 40 ? 8 : 4); }
		if (t->on_dev) reconcile_counts(x);
		x->its = t;
		memset(&i->i, 0, sizeof(i->i));                          /* rope == 0: this iterator is served by x->its */
		i->i.d = -1;
		itr_stream_open(t, 0);
		return;
	}
	mr_sync_host(mr);
	rope_itr_first(mr->r[0], &i->i);
}

static const uint8_t *itr_stream_next(mritr_t *i)
{
	mrx_t *x = X(i->r);
	itr_stream_t *t = x->its;
	if (t == 0 || t->owner != i) return 0;                      /* (the walk is over, or another mr_itr_first took its place) */
	while (i->a < 6) {
		if (t->k < t->nl) {
			if (rope_rdump_nleaves(t->d) == 0) { t->blk[0] = t->blk[1] = 0; ++t->k; }   /* the one empty leaf of an empty rope */
			else rope_rdump_block(t->d, t->k++, t->blk);
			return t->blk;
		}
		rope_rdump_free(t->d); t->d = 0;
		if (i->to_free) {                                        /* mrope.c:122-125; the run bytes a restored file left go the same way */
			rope_destroy(i->r->r[i->a]); i->r->r[i->a] = 0;
			if (!t->on_dev) { free(x->raw[i->a]); x->raw[i->a] = 0; x->raw_n[i->a] = 0; }
		}
		if (++i->a < 6) itr_stream_open(t, i->a);
	}
	if (i->to_free) { if (!t->on_dev) x->raw_ok = 0; x->consumed = 1; }   /* the run bytes / the ropes are gone: no copy of the index is whole any more (check_live) */
	itr_stream_drop(x);
	return 0;
}

const uint8_t *mr_itr_next_block(mritr_t *i)
{
	const uint8_t *blk;
	if (i->i.rope == 0) return itr_stream_next(i);
	while (i->a < 6) {
		if ((blk = rope_itr_next_block(&i->i)) != 0) return blk;
		if (i->to_free) { rope_destroy(i->r->r[i->a]); i->r->r[i->a] = 0; }   /* mrope.c:122-125 */
		if (++i->a < 6) rope_itr_first(i->r->r[i->a], &i->i);
	}
	if (i->to_free) X(i->r)->consumed = 1;                      /* (check_live) */
	return 0;
}

void mr_print_tree(const mrope_t *mr)
{
	int a;
	mr_sync_host((mrope_t*)mr);
	for (a = 0; a < 6; ++a) rope_print_node(mr->r[a]->root);
	putchar('\n');
}

typedef struct { const rope_t *r; int part; int64_t off, size; } dump_part_t;
typedef struct { dump_part_t *job; int njob, next, fd, pass, err; pthread_mutex_t mu; } dump_pool_t;

static void *dump_pool_worker(void *arg)
{
	dump_pool_t *dp = (dump_pool_t*)arg;
	for (;;) {
		int k;
		pthread_mutex_lock(&dp->mu);
		k = dp->next < dp->njob ? dp->next++ : -1;
		pthread_mutex_unlock(&dp->mu);
		if (k < 0) break;
		if (dp->pass == 0) dp->job[k].size = rope_dump_part_size(dp->job[k].r, dp->job[k].part);
		else if (rope_dump_part_at(dp->job[k].r, dp->job[k].part, dp->fd, dp->job[k].off) != 0) {
			pthread_mutex_lock(&dp->mu); dp->err = 1; pthread_mutex_unlock(&dp->mu);
		}
	}
	return 0;
}

/* The dump of an index whose host trees do not exist (it lives on the device, or it was restored as run bytes and never used
 * on the host): rope_dump's bytes follow from the run bytes alone (rope_rdump_*, rope.c), so every rope's run stream goes from
 * the device into a buffer once, a thread per rope brings it into canonical form and finds the leaf boundaries -- which gives the
 * size of its dump and so the file offset of the next rope -- and then writes its leaf records straight into the file, while the
 * next rope is streaming off the device.  No B+ trees: at configs[2] size they were 92 GB of host memory and 30 s.
 * Regular files only (pwrite); the bytes are those of the tree path (tests/test_host_layer.py, tests/test_cli_gpu.py). */
typedef struct {
	runbuf_t rb; int keep_rb;
	int max_nodes, block_len, nthr, fd, a, err;
	int64_t *off;                                              /* off[a]: start of rope a in the file, valid once off_ok > a */
	int *off_ok; pthread_mutex_t *mu; pthread_cond_t *cv;
} rdump_rope_t;

static void *rdump_rope_worker(void *arg)
{
	rdump_rope_t *j = (rdump_rope_t*)arg;
	rope_rdump_t *d = rope_rdump_prepare(j->rb.p, j->rb.n, j->max_nodes, j->block_len, j->nthr);
	int64_t at;
	if (!j->keep_rb) { free(j->rb.p); j->rb.p = 0; }
	pthread_mutex_lock(j->mu);
	while (*j->off_ok <= j->a) pthread_cond_wait(j->cv, j->mu);
	at = j->off[j->a];
	j->off[j->a + 1] = at + rope_rdump_size(d);
	*j->off_ok = j->a + 2;
	pthread_cond_broadcast(j->cv);
	pthread_mutex_unlock(j->mu);
	if (rope_rdump_write(d, j->fd, at) != 0) j->err = 1;
	return 0;
}

static int dump_without_trees(mrope_t *mr, FILE *fp)
{
	mrx_t *x = X(mr);
	rdump_rope_t job[6];
	pthread_t th[6];
	pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
	pthread_cond_t cv = PTHREAD_COND_INITIALIZER;
	int64_t off[7];
	int a, off_ok = 1, err = 0, nthr;
	const int on_dev = has_dev(x) && x->dev_ok, trace = getenv("RB2_SYNC_TRACE") != 0;
	if (getenv("RB2_LOAD_THREADS")) nthr = atoi(getenv("RB2_LOAD_THREADS"));
	else { const long nc = sysconf(_SC_NPROCESSORS_ONLN); nthr = nc >= 80 ? 16 : (nc >= 40 ? 8 : 4); }
	if (fflush(fp) != 0 || (off[0] = (int64_t)ftello(fp)) < 0) return -1;
	for (a = 0; a < 6; ++a) {
		struct timespec t0, t1;
		memset(&job[a], 0, sizeof(job[a]));
		job[a].max_nodes = x->max_nodes; job[a].block_len = x->block_len; job[a].nthr = nthr; job[a].fd = fileno(fp); job[a].a = a;
		job[a].off = off; job[a].off_ok = &off_ok; job[a].mu = &mu; job[a].cv = &cv;
		if (on_dev) {
			int64_t c[36], ub = 1 << 20; int b;                    /* (as mr_sync_host: the symbols of the rope bound its run bytes) */
			dev_get_counts(x, c);
			for (b = 0; b < 6; ++b) ub += c[a * 6 + b];
			job[a].rb.p = (uint8_t*)malloc((size_t)ub);
			job[a].rb.m = job[a].rb.p ? ub : 0;
			rb2_hint_huge(job[a].rb.p, (size_t)ub);
			clock_gettime(CLOCK_MONOTONIC, &t0);
			dev_stream_rope(x, a, runbuf_add, &job[a].rb);
			clock_gettime(CLOCK_MONOTONIC, &t1);
			if (trace) fprintf(stderr, "[mr_dump] rope %d: %.2f GB of runs off the device in %.3f s\n", a, job[a].rb.n / 1e9, (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9);
		} else { job[a].rb.p = x->raw[a]; job[a].rb.n = x->raw_n[a]; job[a].keep_rb = 1; }
		pthread_create(&th[a], 0, rdump_rope_worker, &job[a]);
	}
	for (a = 0; a < 6; ++a) { pthread_join(th[a], 0); err |= job[a].err; }
	if (err) { fprintf(stderr, "[E::%s] write error\n", "mr_dump"); exit(1); }
	if (fseeko(fp, (off_t)off[6], SEEK_SET) != 0) { fprintf(stderr, "[E::%s] cannot seek behind the dump\n", "mr_dump"); exit(1); }
	return 0;
}

void mr_dump(mrope_t *mr, FILE *fp)
{
	int a;
	struct stat st;
	mrx_t *xx = X(mr);
	fwrite("RB\2", 1, 3, fp);                                    /* magic; byte 3 = sorting order (mrope.c:139-140) */
	fwrite(&mr->so, 1, 1, fp);
	if (!xx->host_ok && ((has_dev(xx) && xx->dev_ok) || xx->raw_ok) && !getenv("RB2_DUMP_VIA_TREES")) {
		const int fl = fflush(fp) == 0 ? fcntl(fileno(fp), F_GETFL) : -1;
		if (fl >= 0 && !(fl & O_APPEND) && fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && dump_without_trees(mr, fp) == 0) return;
	}
	mr_sync_host(mr);
	/* a regular file: the dump is cut into parts -- per rope the header and one part per subtree of the root -- that are sized,
	 * given their offsets, and written with pwrite by a pool of threads (the bytes are those of rope_dump, mrope.c:141; the
	 * reference writes them one fwrite after the other).  Pipes and terminals take the sequential path. */
	/* Measured on tmpfs: at 7.5 GB (configs[1]) one thread of fwrite is the fastest (1.2 s; 4 / 16 / 32 threads: 2.3 / 2.1 / 1.7 s,
	 * the sizing pass included), at 73 GB the threads win (8.7 s): they are used for large indexes, or when RB2_DUMP_THREADS asks. */
	{
		int64_t c[6], tot = 0;
		mr_get_c(mr, c);
		for (a = 0; a < 6; ++a) tot += c[a];
		if (tot < ((int64_t)1 << 35) && !getenv("RB2_DUMP_THREADS")) st.st_mode = 0;
		else if (fflush(fp) != 0 || fstat(fileno(fp), &st) != 0) st.st_mode = 0;
		else {                                                     /* `>> out.fmr`: pwrite ignores its offset on O_APPEND descriptors (Linux) -- the parts would land in completion order */
			const int fl = fcntl(fileno(fp), F_GETFL);
			if (fl < 0 || (fl & O_APPEND)) st.st_mode = 0;
		}
	}
	if (S_ISREG(st.st_mode) && !getenv("RB2_DUMP_SEQUENTIAL")) {
		dump_pool_t dp;
		pthread_t th[32];
		int nthr = getenv("RB2_DUMP_THREADS") ? atoi(getenv("RB2_DUMP_THREADS")) : 16, k, p;
		int64_t off = (int64_t)ftello(fp);
		if (off >= 0) {
			if (nthr < 1) nthr = 1;
			if (nthr > 32) nthr = 32;
			memset(&dp, 0, sizeof(dp));
			pthread_mutex_init(&dp.mu, 0);
			for (a = 0; a < 6; ++a) dp.njob += rope_dump_nparts(mr->r[a]);
			dp.job = (dump_part_t*)calloc(dp.njob, sizeof(dump_part_t));
			if (dp.job == 0) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(1); }
			for (a = 0, k = 0; a < 6; ++a)
				for (p = 0; p < rope_dump_nparts(mr->r[a]); ++p, ++k) { dp.job[k].r = mr->r[a]; dp.job[k].part = p; }
			dp.fd = fileno(fp);
			for (dp.pass = 0; dp.pass < 2; ++dp.pass) {           /* sizes, then bytes */
				dp.next = 0;
				for (k = 0; k < nthr; ++k) pthread_create(&th[k], 0, dump_pool_worker, &dp);
				for (k = 0; k < nthr; ++k) pthread_join(th[k], 0);
				if (dp.pass == 0) for (k = 0; k < dp.njob; ++k) { dp.job[k].off = off; off += dp.job[k].size; }
			}
			free(dp.job);
			pthread_mutex_destroy(&dp.mu);
			if (dp.err) { fprintf(stderr, "[E::%s] write error\n", __func__); exit(1); }
			if (fseeko(fp, (off_t)off, SEEK_SET) != 0) { fprintf(stderr, "[E::%s] cannot seek behind the dump\n", __func__); exit(1); }
			return;
		}
	}
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

/* rb2 extension: mr_restore for an index that is going to the GPU.  The .fmr is a depth-first dump of six B+ trees
 * (rope_dump, rope.c:270-275); a device build needs none of the tree, only the run bytes of the leaves in order, so they are
 * gathered into one array per rope while the file streams by -- no nodes, no 512-byte blocks: at configs[4] size (37 GB)
 * that is the difference between ~40 s (trees + export) and a sequential read.  The marginal counts (mr_get_c) come from
 * the leaf records.  Host operations still work: mr_sync_host builds the trees from the run bytes on demand. */
typedef struct {
	FILE *fp; uint8_t *io; int64_t io_n, io_at, io_cap;         /* read window over the file */
	uint8_t *p; int64_t n, m;                                    /* run bytes of the rope being read */
	int max_nodes, block_len, err;
} rawld_t;

static const uint8_t *raw_take(rawld_t *w, int64_t k)          /* the next k bytes of the file (k <= 64 KiB), 0 at a premature end */
{
	if (w->io_n - w->io_at < k) {
		memmove(w->io, w->io + w->io_at, (size_t)(w->io_n - w->io_at));
		w->io_n -= w->io_at; w->io_at = 0;
		w->io_n += (int64_t)fread(w->io + w->io_n, 1, (size_t)(w->io_cap - w->io_n), w->fp);
		if (w->io_n < k) { w->err = 1; return 0; }
	}
	w->io_at += k;
	return w->io + w->io_at - k;
}

static void raw_bucket(rawld_t *w, int64_t c[6])
{
	const uint8_t *q;
	int i, a, n, isb;
	if (w->err || (q = raw_take(w, 3)) == 0) return;
	isb = q[0]; n = (int16_t)(q[1] | q[2] << 8);
	if (n < 1 || n > w->max_nodes) { w->err = 1; return; }
	for (i = 0; i < n && !w->err; ++i) {
		if (isb) {
			int64_t lc[6]; int nb;
			if ((q = raw_take(w, 50)) == 0) return;
			memcpy(lc, q, 48);
			nb = q[48] | q[49] << 8;
			if (nb + 2 > w->block_len || (q = raw_take(w, nb)) == 0) { w->err = 1; return; }
			if (w->n + nb > w->m) {
				w->m = (w->n + nb) + ((w->n + nb) >> 1) + (1 << 20); w->p = (uint8_t*)realloc(w->p, (size_t)w->m);
				if (w->p == 0) { fprintf(stderr, "[E::%s] out of memory (%lld bytes of run bytes)\n", "mr_restore_runs", (long long)w->m); exit(1); }
			}
			memcpy(w->p + w->n, q, (size_t)nb);
			w->n += nb;
			for (a = 0; a < 6; ++a) c[a] += lc[a];
		} else raw_bucket(w, c);
	}
}

mrope_t *mr_restore_runs(FILE *fp)
{
	uint8_t magic[4];
	mrx_t *x;
	rawld_t w;
	int64_t c[6];
	int a;
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RB\2", 3) != 0 || magic[3] > 2) {
		fprintf(stderr, "[E::%s] not an FMR file\n", __func__);
		return 0;
	}
	x = (mrx_t*)calloc(1, sizeof(mrx_t));
	x->pub.so = magic[3];
	x->pub.thr_min = 1000;
	memset(&w, 0, sizeof(w));
	w.fp = fp; w.io_cap = 64 << 20; w.io = (uint8_t*)malloc((size_t)w.io_cap);
	for (a = 0; a < 6; ++a) {
		const uint8_t *q = raw_take(&w, 8);
		int32_t hdr[2];
		if (q == 0) { fprintf(stderr, "[E::%s] not an FMR rope\n", __func__); exit(1); }
		memcpy(hdr, q, 8);
		if (hdr[0] < 2 || hdr[1] < 32) { fprintf(stderr, "[E::%s] not an FMR rope\n", __func__); exit(1); }
		w.max_nodes = hdr[0]; w.block_len = hdr[1]; w.p = 0; w.n = w.m = 0;
		x->pub.r[a] = rope_init(hdr[0], hdr[1]);
		raw_bucket(&w, x->pub.r[a]->c);
		if (w.err) { fprintf(stderr, "[E::%s] corrupt or truncated file\n", __func__); exit(1); }
		x->raw[a] = w.p; x->raw_n[a] = w.n;
		if (a == 0) { x->max_nodes = hdr[0]; x->block_len = hdr[1]; }
	}
	free(w.io);
	x->host_ok = 0; x->dev_ok = 0; x->raw_ok = 1;
	mr_get_c(&x->pub, c);
	fprintf(stderr, "[M::%s] ($, A, C, G, T, N) = (%ld, %ld, %ld, %ld, %ld, %ld)\n", "mr_restore",
			(long)c[0], (long)c[1], (long)c[2], (long)c[3], (long)c[4], (long)c[5]);
	return &x->pub;
}
