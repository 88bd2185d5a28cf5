/* rope.c -- one rope (B+ tree over run-length leaves), host side (API of include/rope.h).
 *
 * Own implementation behind the interface of /root/reference/rope.c.  Differences by design:
 *   - insertion descends first and splits on the way back up (the reference splits full nodes
 *     pre-emptively on the way down, rope.c:119-124);
 *   - rope_load_runs() bulk-loads a whole rope bottom-up from a run stream -- this is how a BWT
 *     built on the GPU (rb2_hip_download_rope) becomes a host rope;
 *   - memory comes from two append-only arenas that are dropped as a whole.
 * The node/bucket layout (rpnode_t, "first entry carries n and is_bottom") and the .fmr byte
 * format are those of the reference because callers and files depend on them.
 */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
#include <emmintrin.h>
#include <pthread.h>
#include <assert.h>
#include <stdio.h>
#include "rle.h"
#include "rope.h"
#define RB2_THP_WHICH 4
#include "rb2_parcopy.h"

/* ---------------------------------------------------------------------------------------------
 * arena: fixed-size zeroed items, never freed individually
 * ------------------------------------------------------------------------------------------- */

#define ARENA_CHUNK (1 << 20)

typedef struct arena_chunk_s { struct arena_chunk_s *next; } arena_chunk_t;
typedef struct { size_t item, per_chunk, used; arena_chunk_t *head; } arena_t;

static arena_t *arena_new(size_t item)
{
	arena_t *a = (arena_t*)calloc(1, sizeof(arena_t));
	a->item = item;
	a->per_chunk = ARENA_CHUNK / item ? ARENA_CHUNK / item : 1;
	a->used = a->per_chunk;                          /* forces a chunk on first use */
	return a;
}

static void *arena_get(arena_t *a)
{
	if (a->used == a->per_chunk) {
		arena_chunk_t *c = (arena_chunk_t*)calloc(1, sizeof(arena_chunk_t) + 16 + a->item * a->per_chunk);
		c->next = a->head; a->head = c; a->used = 0;
	}
	return (uint8_t*)(a->head + 1) + 16 - sizeof(arena_chunk_t) % 16 + a->item * a->used++;
}

static void arena_free(arena_t *a)
{
	arena_chunk_t *c, *n;
	if (!a) return;
	for (c = a->head; c; c = n) { n = c->next; free(c); }
	free(a);
}

static rpnode_t *new_bucket(rope_t *r) { return (rpnode_t*)arena_get((arena_t*)r->node); }
static uint8_t  *new_leaf(rope_t *r)   { return (uint8_t*)arena_get((arena_t*)r->leaf); }

/* ---------------------------------------------------------------------------------------------
 * construction / destruction
 * ------------------------------------------------------------------------------------------- */

static void rope_reset(rope_t *r)
{
	if (r->node) arena_free((arena_t*)r->node);
	if (r->leaf) arena_free((arena_t*)r->leaf);
	r->node = arena_new(sizeof(rpnode_t) * (r->max_nodes + 1));
	r->leaf = arena_new(r->block_len);
	memset(r->c, 0, sizeof(r->c));
	r->root = new_bucket(r);
	r->root->n = 1; r->root->is_bottom = 1;
	r->root->p = (rpnode_t*)new_leaf(r);
}

rope_t *rope_init(int max_nodes, int block_len)
{
	rope_t *r = (rope_t*)calloc(1, sizeof(rope_t));
	if (block_len < 32) block_len = 32;              /* rope.c:59 */
	r->max_nodes = (max_nodes + 1) / 2 * 2;          /* both even, rope.c:60-61 */
	if (r->max_nodes < 4) r->max_nodes = 4;
	r->block_len = (block_len + 7) / 8 * 8;
	rope_reset(r);
	return r;
}

void rope_destroy(rope_t *r)
{
	if (!r) return;
	arena_free((arena_t*)r->node);
	arena_free((arena_t*)r->leaf);
	free(r);
}

/* ---------------------------------------------------------------------------------------------
 * insertion
 * ------------------------------------------------------------------------------------------- */

static void entry_sum(const rpnode_t *bucket, int from, int to, int64_t c[6], int64_t *l)
{
	int i, a;
	memset(c, 0, 48); *l = 0;
	for (i = from; i < to; ++i) { for (a = 0; a < 6; ++a) c[a] += bucket[i].c[a]; *l += bucket[i].l; }
}

/* open a zeroed slot behind position i of a bucket */
static rpnode_t *bucket_open_slot(rpnode_t *bucket, int i)
{
	const int n = bucket->n;
	if (i + 1 < n) memmove(&bucket[i + 2], &bucket[i + 1], sizeof(rpnode_t) * (n - i - 1));
	memset(&bucket[i + 1], 0, sizeof(rpnode_t));
	bucket->n = n + 1;                               /* n lives in entry 0, which never moves */
	return &bucket[i + 1];
}

int64_t rope_insert_run(rope_t *rope, int64_t x, int a, int64_t rl, rpcache_t *cache)
{
	rpnode_t *path[ROPE_MAX_DEPTH]; int pidx[ROPE_MAX_DEPTH], depth = 0, k, j, nbytes;
	rpnode_t *u = rope->root;
	int64_t y = 0, z = 0, cnt[6];
	(void)cache;                                     /* the leaf cache is an optimisation we do not need */
	for (;;) {                                       /* a position on a boundary goes to the left child (rope.c:130) */
		int i = 0, n = u->n;
		while (i < n - 1 && y + (int64_t)u[i].l < x) { y += u[i].l; z += u[i].c[a]; ++i; }
		assert(depth < ROPE_MAX_DEPTH);
		path[depth] = u; pidx[depth] = i; ++depth;
		if (u->is_bottom) break;
		u = u[i].p;
	}
	{
		rpnode_t *e = &path[depth-1][pidx[depth-1]];
		nbytes = rle_insert((uint8_t*)e->p, x - y, a, rl, cnt, e->c);
		z += cnt[a];
	}
	for (k = 0; k < depth; ++k) { rpnode_t *e = &path[k][pidx[k]]; e->c[a] += rl; e->l += rl; }
	rope->c[a] += rl;
	if (nbytes + RLE_MIN_SPACE > rope->block_len) {  /* leaf is (nearly) full: split it, then any full bucket above */
		rpnode_t *bucket = path[depth-1], *e = &bucket[pidx[depth-1]], *ne;
		uint8_t *nl = new_leaf(rope);
		int64_t nc[6] = { 0, 0, 0, 0, 0, 0 }, nlen = 0;
		rle_split((uint8_t*)e->p, nl);
		rle_count(nl, nc);
		for (j = 0; j < 6; ++j) { e->c[j] -= nc[j]; nlen += nc[j]; }
		e->l -= nlen;
		ne = bucket_open_slot(bucket, pidx[depth-1]);
		ne->p = (rpnode_t*)nl; ne->l = nlen; memcpy(ne->c, nc, 48);
		/* a bucket is split as soon as it holds max_nodes entries.  Buckets restored from a reference-built .fmr may
		 * already hold max_nodes at rest (the reference splits those on its next descent, rope.c:120-124): they reach
		 * max_nodes + 1 here, which is why arena items have one spare entry (rope_reset / rope_restore) */
		for (k = depth - 1; k >= 0 && (int)path[k]->n >= rope->max_nodes; --k) {
			rpnode_t *full = path[k], *right = new_bucket(rope), *parent, *pe;
			const int nfull = full->n, half = nfull / 2, rest = nfull - half, isb = full->is_bottom;
			int64_t sc[6], sl;
			memcpy(right, full + half, sizeof(rpnode_t) * rest);
			right->n = rest; right->is_bottom = isb;
			full->n = half;
			if (k == 0) {                            /* grow a new root */
				parent = new_bucket(rope);
				parent->n = 1; parent->is_bottom = 0;
				parent->p = full;
				rope->root = parent;
				pe = ne = bucket_open_slot(parent, 0); pe = &parent[0];
			} else {
				parent = path[k-1];
				ne = bucket_open_slot(parent, pidx[k-1]);
				pe = &parent[pidx[k-1]];
			}
			entry_sum(full, 0, half, sc, &sl);  memcpy(pe->c, sc, 48); pe->l = sl; pe->p = full;
			entry_sum(right, 0, rest, sc, &sl); memcpy(ne->c, sc, 48); ne->l = sl; ne->p = right;
			if (k == 0) break;
		}
	}
	return z;
}

/* ---------------------------------------------------------------------------------------------
 * rank
 * ------------------------------------------------------------------------------------------- */

static void count_prefix(const rope_t *rope, int64_t x, int64_t cx[6])
{
	const rpnode_t *u = rope->root;
	int64_t y = 0;
	int a;
	memset(cx, 0, 48);
	for (;;) {
		int i = 0, n = u->n;
		while (i < n - 1 && y + (int64_t)u[i].l < x) { for (a = 0; a < 6; ++a) cx[a] += u[i].c[a]; y += u[i].l; ++i; }
		if (u->is_bottom) { rle_rank1a((const uint8_t*)u[i].p, x - y, cx, u[i].c); return; }
		u = u[i].p;
	}
}

void rope_rank2a(const rope_t *rope, int64_t x, int64_t y, int64_t *cx, int64_t *cy)
{
	count_prefix(rope, x, cx);
	if (cy && y >= x) count_prefix(rope, y, cy);
}

/* ---------------------------------------------------------------------------------------------
 * leaf iterator
 * ------------------------------------------------------------------------------------------- */

static void itr_descend(rpitr_t *it)
{
	while (!it->pa[it->d]->is_bottom) {
		const rpnode_t *child = it->pa[it->d][it->ia[it->d]].p;
		++it->d;
		it->pa[it->d] = child; it->ia[it->d] = 0;
	}
}

void rope_itr_first(const rope_t *rope, rpitr_t *it)
{
	memset(it, 0, sizeof(rpitr_t));
	it->rope = rope;
	it->pa[0] = rope->root;
	itr_descend(it);
}

const uint8_t *rope_itr_next_block(rpitr_t *it)
{
	const uint8_t *blk;
	if (it->d < 0) return 0;
	assert(it->d < ROPE_MAX_DEPTH);
	blk = (const uint8_t*)it->pa[it->d][it->ia[it->d]].p;
	while (it->d >= 0) {                             /* climb until some level has a next sibling */
		if (++it->ia[it->d] < (int)it->pa[it->d]->n) break;
		it->ia[it->d--] = 0;
	}
	if (it->d >= 0) itr_descend(it);
	return blk;
}

/* ---------------------------------------------------------------------------------------------
 * debugging / serialisation (.fmr)
 * ------------------------------------------------------------------------------------------- */

void rope_print_node(const rpnode_t *p)
{
	int i, n = p->n;
	putchar('(');
	for (i = 0; i < n; ++i) {
		if (i) putchar(',');
		if (p->is_bottom) {
			const uint8_t *q = (const uint8_t*)p[i].p + 2, *end = q + *rle_nptr(p[i].p);
			while (q < end) {
				int c; int64_t l, k;
				q += rle_dec1_fn(q, &c, &l);
				for (k = 0; k < l; ++k) putchar("$ACGTN"[c]);
			}
		} else rope_print_node(p[i].p);
	}
	putchar(')');
}

/* pre-order: u8 is_bottom, i16 n; bottom: n x (6 x i64 counts, u16 nbytes, bytes); else children */
static void dump_bucket(const rpnode_t *p, FILE *fp)
{
	const uint8_t isb = p->is_bottom;
	const int16_t n = (int16_t)p->n;
	int i;
	fwrite(&isb, 1, 1, fp);
	fwrite(&n, 2, 1, fp);
	for (i = 0; i < n; ++i) {
		if (isb) {
			fwrite(p[i].c, 8, 6, fp);
			fwrite(p[i].p, 1, 2 + *rle_nptr(p[i].p), fp);
		} else dump_bucket(p[i].p, fp);
	}
}

void rope_dump(const rope_t *r, FILE *fp)
{
	fwrite(&r->max_nodes, 4, 1, fp);
	fwrite(&r->block_len, 4, 1, fp);
	dump_bucket(r->root, fp);
}

/* The same bytes, written at a known offset of a regular file (rb2 extension, used by mr_dump to write the six ropes from six
 * threads: at configs[4] size the .fmr is 73 GB and fwrite of one small piece after the other was 90 s of the run). */
static int64_t bucket_dump_size(const rpnode_t *p)
{
	int64_t sz = 3;
	int i;
	for (i = 0; i < (int)p->n; ++i) sz += p->is_bottom ? 48 + 2 + *rle_nptr(p[i].p) : bucket_dump_size(p[i].p);
	return sz;
}

int64_t rope_dump_size(const rope_t *r) { return 8 + bucket_dump_size(r->root); }

typedef struct { int fd; int64_t off; uint8_t *buf; int64_t n, cap; int err; pthread_mutex_t *one; } dumpw_t;

/* one file in tmpfs takes 6 GB/s from ONE writing thread and less from many (tools/ubench/tmpfs_write.c: 5.4 GB/s from 16, 3.8 from
 * 32 -- the writes of one file serialise on its inode and the threads only add contention): writers that assemble their pieces in
 * parallel take turns at the file (RB2_DUMP_TURNS=0: all at once) */
static pthread_mutex_t g_dump_turn = PTHREAD_MUTEX_INITIALIZER;
static int dump_turns(void) { static int v = -1; if (v < 0) { const char *e = getenv("RB2_DUMP_TURNS"); v = e ? atoi(e) : 1; } return v; }

static void dumpw_flush(dumpw_t *w)
{
	int64_t done = 0;
	if (w->one) pthread_mutex_lock(w->one);
	while (done < w->n && !w->err) {
		const ssize_t k = pwrite(w->fd, w->buf + done, (size_t)(w->n - done), (off_t)(w->off + done));
		if (k <= 0) { w->err = 1; break; }
		done += k;
	}
	if (w->one) pthread_mutex_unlock(w->one);
	w->off += w->n; w->n = 0;
}

static inline void dumpw_put(dumpw_t *w, const void *src, int64_t len)
{
	if (w->n + len > w->cap) dumpw_flush(w);
	memcpy(w->buf + w->n, src, (size_t)len);
	w->n += len;
}

static void dump_bucket_w(const rpnode_t *p, dumpw_t *w)
{
	const uint8_t isb = p->is_bottom;
	const int16_t n = (int16_t)p->n;
	int i;
	dumpw_put(w, &isb, 1);
	dumpw_put(w, &n, 2);
	for (i = 0; i < n; ++i) {
		if (isb) {
			dumpw_put(w, p[i].c, 48);
			dumpw_put(w, p[i].p, 2 + *rle_nptr(p[i].p));
		} else dump_bucket_w(p[i].p, w);
	}
}

/* The dump in parts that can be sized and written independently: part 0 = the rope header and the root bucket's header (and,
 * when the root is a bottom bucket, its leaves); parts 1..n = the subtrees of the root's n children, in order.  Concatenated
 * they are rope_dump's bytes. */
int rope_dump_nparts(const rope_t *r) { return r->root->is_bottom ? 1 : 1 + (int)r->root->n; }

int64_t rope_dump_part_size(const rope_t *r, int part)
{
	if (part == 0) return r->root->is_bottom ? rope_dump_size(r) : 8 + 3;
	return bucket_dump_size(r->root[part - 1].p);
}

static int dumpw_finish(dumpw_t *w) { dumpw_flush(w); free(w->buf); return w->err ? -1 : 0; }

int rope_dump_part_at(const rope_t *r, int part, int fd, int64_t off)
{
	dumpw_t w;
	if (part == 0 && r->root->is_bottom) return rope_dump_at(r, fd, off);
	w.fd = fd; w.off = off; w.n = 0; w.cap = 8 << 20; w.err = 0; w.one = 0;
	w.buf = (uint8_t*)malloc((size_t)w.cap);
	if (w.buf == 0) return -1;
	if (part == 0) {
		const uint8_t isb = 0;
		const int16_t n = (int16_t)r->root->n;
		dumpw_put(&w, &r->max_nodes, 4);
		dumpw_put(&w, &r->block_len, 4);
		dumpw_put(&w, &isb, 1);
		dumpw_put(&w, &n, 2);
	} else dump_bucket_w(r->root[part - 1].p, &w);
	return dumpw_finish(&w);
}

int rope_dump_at(const rope_t *r, int fd, int64_t off)
{
	dumpw_t w;
	w.fd = fd; w.off = off; w.n = 0; w.cap = 8 << 20; w.err = 0; w.one = 0;
	w.buf = (uint8_t*)malloc((size_t)w.cap);
	if (w.buf == 0) return -1;
	dumpw_put(&w, &r->max_nodes, 4);
	dumpw_put(&w, &r->block_len, 4);
	dump_bucket_w(r->root, &w);
	dumpw_flush(&w);
	free(w.buf);
	return w.err ? -1 : 0;
}

static rpnode_t *restore_bucket(rope_t *r, FILE *fp, int64_t c[6])
{
	uint8_t isb; int16_t n; int i, a;
	rpnode_t *p = new_bucket(r);
	if (fread(&isb, 1, 1, fp) != 1 || fread(&n, 2, 1, fp) != 1) { fprintf(stderr, "[E::rope_restore] truncated file\n"); exit(1); }
	if (n < 1 || n > r->max_nodes) { fprintf(stderr, "[E::rope_restore] bucket with %d entries (max_nodes %d)\n", (int)n, r->max_nodes); exit(1); }
	p->is_bottom = isb; p->n = n;
	memset(c, 0, 48);
	for (i = 0; i < n; ++i) {
		if (isb) {
			uint8_t *blk = new_leaf(r);
			uint16_t nb;
			p[i].p = (rpnode_t*)blk;
			if (fread(p[i].c, 8, 6, fp) != 6 || fread(&nb, 2, 1, fp) != 1 || nb + 2 > r->block_len || fread(blk + 2, 1, nb, fp) != nb) {
				fprintf(stderr, "[E::rope_restore] corrupt leaf\n"); exit(1);
			}
			*rle_nptr(blk) = nb;
		} else p[i].p = restore_bucket(r, fp, p[i].c);
		p[i].l = 0;
		for (a = 0; a < 6; ++a) { p[i].l += p[i].c[a]; c[a] += p[i].c[a]; }   /* internal counts are not stored */
	}
	return p;
}

rope_t *rope_restore(FILE *fp)
{
	rope_t *r = (rope_t*)calloc(1, sizeof(rope_t));
	if (fread(&r->max_nodes, 4, 1, fp) != 1 || fread(&r->block_len, 4, 1, fp) != 1 || r->max_nodes < 2 || r->block_len < 32) {
		fprintf(stderr, "[E::rope_restore] not an FMR rope\n"); exit(1);
	}
	r->node = arena_new(sizeof(rpnode_t) * (r->max_nodes + 1));
	r->leaf = arena_new(r->block_len);
	r->root = restore_bucket(r, fp, r->c);
	return r;
}

/* ---------------------------------------------------------------------------------------------
 * bulk load / export
 * ------------------------------------------------------------------------------------------- */

typedef struct { rpnode_t *v; size_t n, m; } entvec_t;

static rpnode_t *ent_push(entvec_t *e)
{
	if (e->n == e->m) { e->m = e->m ? e->m * 2 : 1024; e->v = (rpnode_t*)realloc(e->v, e->m * sizeof(rpnode_t)); }
	memset(&e->v[e->n], 0, sizeof(rpnode_t));
	return &e->v[e->n++];
}

/* symbols of k one-byte runs (0 llll ccc, rle.h:55-57), per symbol: SSE2 (x86-64 baseline), 16 runs per step */
static void count_plain_runs(const uint8_t *b, int64_t k, int64_t c[6])
{
	const __m128i zero = _mm_setzero_si128(), m7 = _mm_set1_epi8(7), m15 = _mm_set1_epi8(15);
	__m128i acc[6];
	int64_t i = 0;
	int s;
	for (s = 0; s < 6; ++s) acc[s] = zero;
	for (; i + 16 <= k; i += 16) {
		const __m128i v = _mm_loadu_si128((const __m128i*)(b + i));
		const __m128i sym = _mm_and_si128(v, m7), len = _mm_and_si128(_mm_srli_epi16(v, 3), m15);
		for (s = 0; s < 6; ++s)
			acc[s] = _mm_add_epi64(acc[s], _mm_sad_epu8(_mm_and_si128(len, _mm_cmpeq_epi8(sym, _mm_set1_epi8((char)s))), zero));
	}
	for (s = 0; s < 6; ++s) {
		int64_t t[2];
		_mm_storeu_si128((__m128i*)t, acc[s]);
		c[s] += t[0] + t[1];
	}
	for (; i < k; ++i) if ((b[i] & 7) < 6) c[b[i] & 7] += b[i] >> 3;
}

/* first i in [q, end - 1) where byte i cannot be taken as it is: it (or its successor) is not a one-byte run of length >= 1,
 * or both carry the same symbol (they merge); end - 1 if there is none (the last byte always waits for what follows) */
static const uint8_t *plain_span_end(const uint8_t *q, const uint8_t *end)
{
	const uint8_t *i = q;
	for (; i + 17 <= end; i += 16) {                           /* 16 positions per step */
		const __m128i x = _mm_loadu_si128((const __m128i*)i), y = _mm_loadu_si128((const __m128i*)(i + 1));
		const __m128i m7 = _mm_set1_epi8(7), m78 = _mm_set1_epi8(0x78), zero = _mm_setzero_si128();
		const __m128i same = _mm_cmpeq_epi8(_mm_and_si128(_mm_xor_si128(x, y), m7), zero);
		const __m128i wide = _mm_or_si128(x, y);                /* top bit: a multi-byte run */
		const __m128i empty = _mm_or_si128(_mm_cmpeq_epi8(_mm_and_si128(x, m78), zero), _mm_cmpeq_epi8(_mm_and_si128(y, m78), zero));
		const int mask = _mm_movemask_epi8(_mm_or_si128(_mm_or_si128(same, empty), wide));
		if (mask) return i + __builtin_ctz((unsigned)mask);
	}
	for (; i + 1 < end; ++i)
		if (((i[0] | i[1]) & 0x80) || (i[0] & 0x78) == 0 || (i[1] & 0x78) == 0 || ((i[0] ^ i[1]) & 7) == 0) return i;
	return end > q ? end - 1 : q;
}

static void build_upper_levels(rope_t *rope, entvec_t *plv);

void rope_load_runs(rope_t *rope, const uint8_t *rle, int64_t n_bytes)
{
	const uint8_t *q = rle, *end = rle + (n_bytes > 0 ? n_bytes : 0);
	const int fill = rope->block_len - RLE_MIN_SPACE - 2;      /* bytes of runs per leaf: leaves room for one insertion */
	entvec_t lv = { 0, 0, 0 };
	rpnode_t *cur = 0;
	uint8_t *blk = 0;
	int pc = -1, a; int64_t pl = 0;
	rope_reset(rope);
	/* leaves: merge adjacent equal symbols, re-encode with the widest run form needed */
	for (;;) {
		int c = -1; int64_t l = 0;
		/* the stream off the device is almost only one-byte runs that neither merge nor need re-encoding: such stretches are
		 * copied into the leaves as they are and counted 16 runs at a time (one run at a time, this loop was 4.5 s per 1.7 GB
		 * rope at configs[1] size and the whole of mr_sync_host's 5.3 s) */
		if (end - q >= 64 && !(q[0] & 0x80) && (q[0] & 0x78) && (pc < 0 || (q[0] & 7) != pc)) {
			const uint8_t *m = plain_span_end(q, end);
			if (m - q >= 32) {
				if (pc >= 0) {                                     /* the pending run is final: nothing after it carries its symbol */
					uint8_t tmp[8];
					const int nb = rle_enc1(tmp, pc, pl);
					if (!cur || *rle_nptr(blk) + nb > fill) {
						cur = ent_push(&lv);
						blk = lv.n == 1 ? (uint8_t*)rope->root->p : new_leaf(rope);
						cur->p = (rpnode_t*)blk;
					}
					memcpy(blk + 2 + *rle_nptr(blk), tmp, nb);
					*rle_nptr(blk) += nb;
					cur->c[pc] += pl; cur->l += pl; rope->c[pc] += pl;
					pc = -1;
				}
				while (q < m) {
					int64_t cc[6] = { 0, 0, 0, 0, 0, 0 }, t;
					if (!cur || *rle_nptr(blk) >= fill) {
						cur = ent_push(&lv);
						blk = lv.n == 1 ? (uint8_t*)rope->root->p : new_leaf(rope);
						cur->p = (rpnode_t*)blk;
					}
					t = fill - *rle_nptr(blk);
					if (t > m - q) t = m - q;
					memcpy(blk + 2 + *rle_nptr(blk), q, (size_t)t);
					*rle_nptr(blk) += (uint16_t)t;
					count_plain_runs(q, t, cc);
					for (a = 0; a < 6; ++a) { cur->c[a] += cc[a]; cur->l += cc[a]; rope->c[a] += cc[a]; }
					q += t;
				}
				continue;
			}
		}
		if (q < end) { q += rle_dec1_fn(q, &c, &l); if (l == 0) continue; }
		if (c == pc && c >= 0) { pl += l; continue; }
		if (pc >= 0) {                                         /* flush the pending run */
			uint8_t tmp[8];
			const int nb = rle_enc1(tmp, pc, pl);
			if (!cur || *rle_nptr(blk) + nb > fill) {
				cur = ent_push(&lv);
				blk = lv.n == 1 ? (uint8_t*)rope->root->p : new_leaf(rope);   /* reuse the empty first leaf */
				cur->p = (rpnode_t*)blk;
			}
			memcpy(blk + 2 + *rle_nptr(blk), tmp, nb);
			*rle_nptr(blk) += nb;
			cur->c[pc] += pl; cur->l += pl; rope->c[pc] += pl;
		}
		if (c < 0) break;
		pc = c; pl = l;
	}
	if (lv.n == 0) { free(lv.v); return; }                     /* empty stream: the reset rope is the answer */
	build_upper_levels(rope, &lv);
}

/* levels above the leaves: pack `fan` entries per bucket until a single bucket (the root) remains; frees lv */
static void build_upper_levels(rope_t *rope, entvec_t *plv)
{
	const int fan = rope->max_nodes > 4 ? rope->max_nodes - 2 : rope->max_nodes / 2;
	entvec_t lv = *plv, up = { 0, 0, 0 };
	size_t i;
	int a;
	{
		int bottom = 1;
		for (;;) {
			const size_t nbk = (lv.n + fan - 1) / fan;
			size_t b;
			up.n = 0;
			for (b = 0; b < nbk; ++b) {
				const size_t from = b * fan, to = from + fan < lv.n ? from + fan : lv.n;
				rpnode_t *bk = nbk == 1 ? rope->root : new_bucket(rope), *pe = ent_push(&up);
				memcpy(bk, &lv.v[from], sizeof(rpnode_t) * (to - from));
				bk->n = to - from; bk->is_bottom = bottom;
				pe->p = bk;
				for (i = from; i < to; ++i) { for (a = 0; a < 6; ++a) pe->c[a] += lv.v[i].c[a]; pe->l += lv.v[i].l; }
			}
			if (nbk == 1) break;
			{ entvec_t t = lv; lv = up; up = t; }
			bottom = 0;
		}
	}
	free(lv.v); free(up.v);
}

/* ---- the same tree from several threads (rb2 extension; mr_sync_host at configs[3]/[4] size spends half a minute here) -------
 * rope_load_runs is a sequential automaton twice over: adjacent runs of one symbol merge, and a leaf takes runs until the next
 * does not fit.  Both can be cut open without changing a byte of the result:
 *   1. the stream is cut where two one-byte runs of different symbols meet (nothing merges across such a point); every segment
 *      is brought into canonical form -- merged, re-encoded, empty runs dropped -- by its own thread, into its own buffer;
 *   2. leaf k starts at the last run head <= start(k-1) + fill of the concatenated canonical stream: one pass over the leaves,
 *      a few bytes each (the codec marks continuation bytes, rle.h:39-51);
 *   3. the leaves are filled (copy + symbol counts) by the threads, each a contiguous range; the levels above as before. */
typedef struct { const uint8_t *in; int64_t n; uint8_t *out; int64_t out_n; } canon_job_t;

static int64_t canon_segment(const uint8_t *q, const uint8_t *end, uint8_t *out)
{
	uint8_t *o = out;
	int pc = -1; int64_t pl = 0;
	for (;;) {
		int c = -1; int64_t l = 0;
		if (end - q >= 64 && !(q[0] & 0x80) && (q[0] & 0x78) && (pc < 0 || (q[0] & 7) != pc)) {
			const uint8_t *m = plain_span_end(q, end);
			if (m - q >= 32) {
				if (pc >= 0) { o += rle_enc1(o, pc, pl); pc = -1; }
				memcpy(o, q, (size_t)(m - q)); o += m - q; q = m;
				continue;
			}
		}
		if (q < end) { q += rle_dec1_fn(q, &c, &l); if (l == 0) continue; }
		if (c == pc && c >= 0) { pl += l; continue; }
		if (pc >= 0) o += rle_enc1(o, pc, pl);
		if (c < 0) break;
		pc = c; pl = l;
	}
	return o - out;
}

static void *canon_worker(void *arg)
{
	canon_job_t *j = (canon_job_t*)arg;
	/* merging adjacent runs of one symbol can GROW the stream: a 2-byte run of 255 + a 1-byte run of 1 is a 4-byte run of 256, a
	 * 4-byte run + a 1-byte run that crosses 2^19 an 8-byte one -- at worst 8 bytes out for 5 in: 13/8 of the input bounds it
	 * (pages that are never written are never backed; hosts with strict overcommit count them all the same, hence not 2x). */
	const size_t cap = (size_t)j->n + (size_t)j->n / 8 * 5 + 64;
	j->out = (uint8_t*)malloc(cap);
	rb2_hint_huge(j->out, cap);
	if (j->out == 0) { fprintf(stderr, "[E::%s] out of memory (%lld bytes for the canonical run bytes of one rope segment; RB2_DUMP_VIA_TREES=1 dumps through the host trees instead)\n", __func__, (long long)cap); exit(1); }
	j->out_n = canon_segment(j->in, j->in + j->n, j->out);
	return 0;
}

typedef struct {
	const canon_job_t *seg; const int64_t *pre; int nseg;       /* the canonical stream: segments and their start offsets */
	const int64_t *start; int64_t nl, total;                    /* leaf starts */
	rpnode_t *ent; int64_t k0, k1;                              /* my leaves */
	int64_t c[6];
} fill_job_t;

static inline int vseg(const int64_t *pre, int nseg, int64_t x) { int t = 0; while (t + 1 < nseg && pre[t + 1] <= x) ++t; return t; }

static void vcopy(const fill_job_t *f, uint8_t *dst, int64_t x, int64_t n)
{
	int t = vseg(f->pre, f->nseg, x);
	while (n > 0) {
		const int64_t o = x - f->pre[t], k = n < f->seg[t].out_n - o ? n : f->seg[t].out_n - o;
		memcpy(dst, f->seg[t].out + o, (size_t)k);
		dst += k; x += k; n -= k; ++t;
	}
}

static void *fill_worker(void *arg)
{
	fill_job_t *f = (fill_job_t*)arg;
	int64_t k;
	int a;
	for (k = f->k0; k < f->k1; ++k) {
		const int64_t s = f->start[k], e = k + 1 < f->nl ? f->start[k + 1] : f->total, nb = e - s;
		rpnode_t *en = &f->ent[k];
		uint8_t *blk = (uint8_t*)en->p, *b = blk + 2;
		int64_t cc[6] = { 0, 0, 0, 0, 0, 0 }, i, tot = 0;
		int wide = 0;
		vcopy(f, b, s, nb);
		*rle_nptr(blk) = (uint16_t)nb;
		for (i = 0; i < nb; ++i) wide |= b[i];
		if (!(wide & 0x80)) count_plain_runs(b, nb, cc);
		else {
			const uint8_t *q = b, *end = b + nb;
			while (q < end) { int c; int64_t l; q += rle_dec1_fn(q, &c, &l); if (c < 6) cc[c] += l; }
		}
		for (a = 0; a < 6; ++a) { en->c[a] = cc[a]; tot += cc[a]; f->c[a] += cc[a]; }
		en->l = tot;
	}
	return 0;
}

static int canon_stream(const uint8_t *rle, int64_t n_bytes, int nthr, canon_job_t *seg, int64_t *pre);
static int64_t leaf_starts(const canon_job_t *seg, const int64_t *pre, int nseg, int fill, int64_t **pstart);

void rope_load_runs_mt(rope_t *rope, const uint8_t *rle, int64_t n_bytes, int nthr)
{
	const int fill = rope->block_len - RLE_MIN_SPACE - 2;
	canon_job_t seg[16];
	fill_job_t fj[16];
	pthread_t th[16];
	int64_t pre[17], *start = 0, nl = 0, total, pos;
	entvec_t lv = { 0, 0, 0 };
	int t, nseg = 0, a;
	if (nthr > 16) nthr = 16;
	if (nthr < 2 || n_bytes < (int64_t)nthr * (getenv("RB2_LOAD_MIN_SEG") ? atol(getenv("RB2_LOAD_MIN_SEG")) : 8 << 20)) { rope_load_runs(rope, rle, n_bytes); return; }
	nseg = canon_stream(rle, n_bytes, nthr, seg, pre);            /* 1. canonical form, segment by segment */
	total = pre[nseg];
	rope_reset(rope);
	if (total == 0) { for (t = 0; t < nseg; ++t) free(seg[t].out); return; }
	nl = leaf_starts(seg, pre, nseg, fill, &start);               /* 2. leaf starts */
	/* 3. leaves: allocated here (the arena is not shared), filled by the threads */
	lv.n = lv.m = (size_t)nl;
	lv.v = (rpnode_t*)calloc((size_t)nl, sizeof(rpnode_t));
	for (pos = 0; pos < nl; ++pos) lv.v[pos].p = (rpnode_t*)(pos == 0 ? (uint8_t*)rope->root->p : new_leaf(rope));
	for (t = 0; t < nthr; ++t) {
		memset(&fj[t], 0, sizeof(fj[t]));
		fj[t].seg = seg; fj[t].pre = pre; fj[t].nseg = nseg; fj[t].start = start; fj[t].nl = nl; fj[t].total = total;
		fj[t].ent = lv.v; fj[t].k0 = nl / nthr * t; fj[t].k1 = t == nthr - 1 ? nl : nl / nthr * (t + 1);
		pthread_create(&th[t], 0, fill_worker, &fj[t]);
	}
	for (t = 0; t < nthr; ++t) { pthread_join(th[t], 0); for (a = 0; a < 6; ++a) rope->c[a] += fj[t].c[a]; }
	for (t = 0; t < nseg; ++t) free(seg[t].out);
	free(start);
	build_upper_levels(rope, &lv);
}

/* ---- the dump of that tree without the tree (rb2 extension; mr_dump of an index that lives on the device) ---------------------
 * rope_dump's bytes (rope.c:253-275) depend only on the leaves in order: the levels above pack `fan` entries per bucket
 * (build_upper_levels), so a bucket of level v (0 = bottom) starts at every leaf k with k % fan^(v+1) == 0, and the pre-order
 * dump is the leaf records in order with the 3-byte headers of the buckets that start at a leaf in front of it, top level
 * first.  Steps 1 and 2 of rope_load_runs_mt (canonical stream, leaf starts) give the size of the dump; the leaf records
 * (48 bytes of counts, u16 length, run bytes) are then assembled by the threads, each a contiguous range of leaves, and written
 * with pwrite -- no arena, no nodes, no second pass over the leaves. */
#define RDUMP_MAX_LEVELS 40
struct rope_rdump_s {
	canon_job_t seg[16]; int64_t pre[17]; int nseg;
	int64_t *start, nl, total;
	int max_nodes, block_len, fan, nlev, nthr;
	int64_t cnt[RDUMP_MAX_LEVELS], span[RDUMP_MAX_LEVELS];      /* level v: entries of all its buckets together; leaves below one bucket */
	int64_t size;
};

static int canon_stream(const uint8_t *rle, int64_t n_bytes, int nthr, canon_job_t *seg, int64_t *pre)
{
	pthread_t th[16];
	int t, nseg = 0;
	int64_t from = 0;
	if (nthr > 16) nthr = 16;
	if (nthr < 1) nthr = 1;
	for (t = 1; t <= nthr; ++t) {                              /* cut points: the first place at or behind k * n / nthr where two one-byte runs of different symbols meet */
		int64_t cut = t == nthr ? n_bytes : n_bytes / nthr * t;
		if (t < nthr) {
			if (cut <= from) continue;
			while (cut < n_bytes && !(!(rle[cut - 1] & 0x80) && (rle[cut - 1] & 0x78) && !(rle[cut] & 0x80) && (rle[cut] & 0x78) && ((rle[cut - 1] ^ rle[cut]) & 7))) ++cut;
			if (cut >= n_bytes) continue;                         /* no such place in this stretch: it joins the next one */
		}
		seg[nseg].in = rle + from; seg[nseg].n = cut - from; seg[nseg].out = 0; seg[nseg].out_n = 0;
		++nseg; from = cut;
	}
	if (nseg == 1) canon_worker(&seg[0]);
	else {
		for (t = 0; t < nseg; ++t) pthread_create(&th[t], 0, canon_worker, &seg[t]);
		for (t = 0; t < nseg; ++t) pthread_join(th[t], 0);
	}
	for (t = 0, pre[0] = 0; t < nseg; ++t) pre[t + 1] = pre[t] + seg[t].out_n;
	return nseg;
}

static int64_t leaf_starts(const canon_job_t *seg, const int64_t *pre, int nseg, int fill, int64_t **pstart)
{
	const int64_t total = pre[nseg];
	int64_t *start = 0, nl = 0, ml = 0, pos;
	for (pos = 0; pos < total; ) {
		int64_t nxt = pos + fill;
		if (nl == ml) {
			ml = ml ? ml * 2 : 1 << 16; start = (int64_t*)realloc(start, (size_t)ml * sizeof(int64_t));
			if (start == 0) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(1); }
		}
		start[nl++] = pos;
		if (nxt >= total) break;
		for (;;) {                                             /* back to the head of the run that does not fit any more */
			const int sg = vseg(pre, nseg, nxt);
			if ((seg[sg].out[nxt - pre[sg]] & 0xC0) != 0x80) break;
			--nxt;
		}
		pos = nxt;
	}
	*pstart = start;
	return nl;
}

rope_rdump_t *rope_rdump_prepare(const uint8_t *rle, int64_t n_bytes, int max_nodes, int block_len, int nthr)
{
	rope_rdump_t *d = (rope_rdump_t*)calloc(1, sizeof(rope_rdump_t));
	int64_t c, nbuckets = 0;
	int v;
	if (d == 0) { fprintf(stderr, "[E::%s] out of memory\n", __func__); exit(1); }
	if (block_len < 32) block_len = 32;                        /* as rope_init */
	d->max_nodes = (max_nodes + 1) / 2 * 2;
	if (d->max_nodes < 4) d->max_nodes = 4;
	d->block_len = (block_len + 7) / 8 * 8;
	d->fan = d->max_nodes > 4 ? d->max_nodes - 2 : d->max_nodes / 2;
	d->nthr = nthr < 1 ? 1 : nthr > 16 ? 16 : nthr;
	if (n_bytes < 0) n_bytes = 0;
	{	/* threads that have at least 1 MiB of the stream each (RB2_LOAD_MIN_SEG: bytes, tests) */
		const int64_t min_seg = getenv("RB2_LOAD_MIN_SEG") ? atol(getenv("RB2_LOAD_MIN_SEG")) : 1 << 20;
		const int64_t fit = n_bytes / (min_seg > 0 ? min_seg : 1);
		struct timespec t0, t1, t2;
		clock_gettime(CLOCK_MONOTONIC, &t0);
		d->nseg = canon_stream(rle, n_bytes, fit < 1 ? 1 : fit < d->nthr ? (int)fit : d->nthr, d->seg, d->pre);
		clock_gettime(CLOCK_MONOTONIC, &t1);
		d->total = d->pre[d->nseg];
		d->nl = leaf_starts(d->seg, d->pre, d->nseg, d->block_len - RLE_MIN_SPACE - 2, &d->start);
		clock_gettime(CLOCK_MONOTONIC, &t2);
		if (getenv("RB2_SYNC_TRACE")) fprintf(stderr, "[rope_rdump] %.2f GB of runs: canonical form in %.3f s (%d segments), %lld leaf starts in %.3f s\n", n_bytes / 1e9,
				(t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9, d->nseg, (long long)d->nl, (t2.tv_sec - t1.tv_sec) + (t2.tv_nsec - t1.tv_nsec) * 1e-9);
	}
	/* levels, as build_upper_levels packs them (an empty stream is the reset rope: one empty leaf in a bottom root) */
	for (c = d->nl > 0 ? d->nl : 1, v = 0; ; ++v) {
		const int64_t nbk = (c + d->fan - 1) / d->fan;
		d->cnt[v] = c;
		d->span[v] = v == 0 ? d->fan : (d->span[v - 1] > INT64_MAX / d->fan ? INT64_MAX : d->span[v - 1] * d->fan);
		nbuckets += nbk;
		if (nbk == 1) break;
		c = nbk;
	}
	d->nlev = v + 1;
	d->size = 8 + 3 * nbuckets + 50 * (d->nl > 0 ? d->nl : 1) + d->total;
	return d;
}

int64_t rope_rdump_size(const rope_rdump_t *d) { return d->size; }

typedef struct { const rope_rdump_t *d; int64_t k0, k1; int fd; int64_t off; int err; } rdump_job_t;

static void *rdump_worker(void *arg)
{
	rdump_job_t *j = (rdump_job_t*)arg;
	const rope_rdump_t *d = j->d;
	fill_job_t f;
	dumpw_t w;
	uint8_t blk[2 + 65536 + 8];
	int64_t k, o = j->off + 8 + 50 * j->k0 + d->start[j->k0];
	int v;
	for (v = 0; v < d->nlev; ++v) o += 3 * ((j->k0 + d->span[v] - 1) / d->span[v]);   /* headers of the buckets that start in front of leaf k0 */
	memset(&f, 0, sizeof(f));
	f.seg = d->seg; f.pre = d->pre; f.nseg = d->nseg;
	w.fd = j->fd; w.off = o; w.n = 0; w.cap = 8 << 20; w.err = 0; w.one = dump_turns() ? &g_dump_turn : 0;
	w.buf = (uint8_t*)malloc((size_t)w.cap);
	if (w.buf == 0) { j->err = 1; return 0; }
	for (k = j->k0; k < j->k1; ++k) {
		const int64_t s = d->start[k], e = k + 1 < d->nl ? d->start[k + 1] : d->total, nb = e - s;
		int64_t cc[6] = { 0, 0, 0, 0, 0, 0 }, i;
		uint8_t *b = blk + 2;
		int wide = 0;
		for (v = d->nlev - 1; v >= 0; --v)
			if (k % d->span[v] == 0) {
				const int64_t idx = k / d->span[v], left = d->cnt[v] - idx * d->fan;
				const uint8_t isb = v == 0;
				const int16_t n = (int16_t)(left < d->fan ? left : d->fan);
				dumpw_put(&w, &isb, 1);
				dumpw_put(&w, &n, 2);
			}
		vcopy(&f, b, s, nb);
		*(uint16_t*)blk = (uint16_t)nb;
		for (i = 0; i < nb; ++i) wide |= b[i];
		if (!(wide & 0x80)) count_plain_runs(b, nb, cc);
		else {
			const uint8_t *q = b, *end = b + nb;
			while (q < end) { int c; int64_t l; q += rle_dec1_fn(q, &c, &l); if (c < 6) cc[c] += l; }
		}
		dumpw_put(&w, cc, 48);
		dumpw_put(&w, blk, 2 + nb);
	}
	if (dumpw_finish(&w) != 0) j->err = 1;
	return 0;
}

/* writes the dump at offset off of fd (a regular file) and frees d; 0 = ok */
int rope_rdump_write(rope_rdump_t *d, int fd, int64_t off)
{
	int t, err = 0;
	struct timespec t0, t1;
	clock_gettime(CLOCK_MONOTONIC, &t0);
	{
		uint8_t head[8 + 3 + 50];
		int n = 8;
		memcpy(head, &d->max_nodes, 4); memcpy(head + 4, &d->block_len, 4);
		if (d->nl == 0) { memset(head + 8, 0, 3 + 50); head[8] = 1; head[9] = 1; n = 8 + 3 + 50; }   /* bottom root, one empty leaf */
		{
			int64_t done = 0;
			while (done < n) { const ssize_t k = pwrite(fd, head + done, (size_t)(n - done), (off_t)(off + done)); if (k <= 0) { err = 1; break; } done += k; }
		}
	}
	if (d->nl > 0) {
		rdump_job_t job[16];
		pthread_t th[16];
		const int nthr = d->nl / 64 < 1 ? 1 : d->nl / 64 < d->nthr ? (int)(d->nl / 64) : d->nthr;   /* 64 leaves or more per thread */
		for (t = 0; t < nthr; ++t) {
			job[t].d = d; job[t].fd = fd; job[t].off = off; job[t].err = 0;
			job[t].k0 = d->nl / nthr * t; job[t].k1 = t == nthr - 1 ? d->nl : d->nl / nthr * (t + 1);
			if (nthr > 1) pthread_create(&th[t], 0, rdump_worker, &job[t]);
			else rdump_worker(&job[t]);
		}
		for (t = 0; t < nthr; ++t) { if (nthr > 1) pthread_join(th[t], 0); err |= job[t].err; }
	}
	clock_gettime(CLOCK_MONOTONIC, &t1);
	if (getenv("RB2_SYNC_TRACE")) fprintf(stderr, "[rope_rdump] %.2f GB of leaf records written in %.3f s\n", d->size / 1e9, (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9);
	for (t = 0; t < d->nseg; ++t) free(d->seg[t].out);
	free(d->start); free(d);
	return err ? -1 : 0;
}

/* the leaves of that tree, one at a time, still without the tree (mr_itr_next_block of an index that lives on the device) */
int64_t rope_rdump_nleaves(const rope_rdump_t *d) { return d->nl; }

/* leaf k as a leaf block (u16 byte count + run bytes, rle.h); blk holds block_len bytes or more; returns the run bytes */
int rope_rdump_block(const rope_rdump_t *d, int64_t k, uint8_t *blk)
{
	fill_job_t f;
	const int64_t s = d->start[k], e = k + 1 < d->nl ? d->start[k + 1] : d->total;
	memset(&f, 0, sizeof(f));
	f.seg = d->seg; f.pre = d->pre; f.nseg = d->nseg;
	vcopy(&f, blk + 2, s, e - s);
	*rle_nptr(blk) = (uint16_t)(e - s);
	return (int)(e - s);
}

void rope_rdump_free(rope_rdump_t *d)
{
	int t;
	if (d == 0) return;
	for (t = 0; t < d->nseg; ++t) free(d->seg[t].out);
	free(d->start); free(d);
}

int64_t rope_export_runs(const rope_t *rope, uint8_t **out)
{
	rpitr_t it;
	const uint8_t *blk;
	int64_t n = 0, m = 1 << 16;
	uint8_t *buf = (uint8_t*)malloc(m);
	rope_itr_first(rope, &it);
	while ((blk = rope_itr_next_block(&it)) != 0) {
		const int nb = *rle_nptr(blk);
		if (n + nb > m) { while (n + nb > m) m += m >> 1; buf = (uint8_t*)realloc(buf, m); }
		memcpy(buf + n, blk + 2, nb);
		n += nb;
	}
	*out = buf;
	return n;
}
