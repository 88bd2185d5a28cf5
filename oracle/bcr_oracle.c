/* oracle/bcr_oracle.c -- TEST INFRASTRUCTURE ONLY (see bcr_oracle.h).
 *
 * Restates, in plain C over flat byte arrays, what lh3/ropebwt2 computes on its hot path.
 * Every function names the reference lines it follows.  Nothing here is shipped or measured
 * as product; it exists so that tests can say "the HIP path returns exactly this".
 */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "bcr_oracle.h"

struct orc_s {
	int so;                 /* sorting order (mrope.h:6-8) */
	uint8_t *r[6];          /* rope b as a plain symbol array */
	int64_t n[6], cap[6];
	int64_t c[6][6];        /* c[b][a] = #a in rope b (rope_t.c[], rope.h:19) */
};

/* ------------------------------------------------------------------------------------------ */
/* flat "rope"                                                                                 */
/* ------------------------------------------------------------------------------------------ */

static void flat_reserve(orc_t *o, int b, int64_t need)
{
	if (need <= o->cap[b]) return;
	int64_t cap = o->cap[b] ? o->cap[b] : 1024;
	while (cap < need) cap += cap >> 1;
	o->r[b] = (uint8_t*)realloc(o->r[b], cap);
	o->cap[b] = cap;
}

static int64_t flat_rank(const orc_t *o, int b, int a, int64_t x)
{
	int64_t i, z = 0;
	const uint8_t *s = o->r[b];
	for (i = 0; i < x; ++i) z += (s[i] == a);
	return z;
}

/* rope_rank2a (rope.c:179-194): cx[] = #each symbol in [0,x), cy[] likewise for y */
static void flat_rank2a(const orc_t *o, int b, int64_t x, int64_t y, int64_t cx[6], int64_t cy[6])
{
	int64_t i;
	const uint8_t *s = o->r[b];
	memset(cx, 0, 48);
	for (i = 0; i < x; ++i) ++cx[s[i]];
	if (cy) {
		memcpy(cy, cx, 48);
		for (; i < y; ++i) ++cy[s[i]];
	}
}

/* rope_insert_run (rope.c:114-148): insert rl copies of a after x symbols; return #a in [0,x) */
static int64_t flat_insert_run(orc_t *o, int b, int64_t x, int a, int64_t rl)
{
	int64_t z = flat_rank(o, b, a, x);
	assert(x >= 0 && x <= o->n[b]);
	flat_reserve(o, b, o->n[b] + rl);
	memmove(o->r[b] + x + rl, o->r[b] + x, o->n[b] - x);
	memset(o->r[b] + x, a, rl);
	o->n[b] += rl;
	o->c[b][a] += rl;
	return z;
}

orc_t *orc_create(int so)
{
	orc_t *o = (orc_t*)calloc(1, sizeof(orc_t));
	assert(so >= 0 && so <= 2);   /* mrope.c:18 */
	o->so = so;
	return o;
}

void orc_destroy(orc_t *o)
{
	int b;
	if (!o) return;
	for (b = 0; b < 6; ++b) free(o->r[b]);
	free(o);
}

int64_t orc_rope_len(const orc_t *o, int b) { return o->n[b]; }
const uint8_t *orc_rope_ptr(const orc_t *o, int b) { return o->r[b]; }

void orc_counts(const orc_t *o, int64_t c[36])
{
	int a, b;
	for (b = 0; b < 6; ++b) for (a = 0; a < 6; ++a) c[b*6+a] = o->c[b][a];
}

int64_t orc_total(const orc_t *o)
{
	int b; int64_t t = 0;
	for (b = 0; b < 6; ++b) t += o->n[b];
	return t;
}

int64_t orc_bwt(const orc_t *o, uint8_t *out)
{
	int b; int64_t k = 0;
	for (b = 0; b < 6; ++b) { memcpy(out + k, o->r[b], o->n[b]); k += o->n[b]; }
	return k;
}

/* mr_rank2a with y<0 (mrope.c:70-105): ropes are concatenated in the order $,A,C,G,T,N */
void orc_rank1a(const orc_t *o, int64_t x, int64_t cx[6])
{
	int a, b; int64_t i;
	memset(cx, 0, 48);
	for (b = 0; b < 6 && x > 0; ++b) {
		if (x >= o->n[b]) { for (a = 0; a < 6; ++a) cx[a] += o->c[b][a]; x -= o->n[b]; }
		else { for (i = 0; i < x; ++i) ++cx[o->r[b][i]]; x = 0; }
	}
}

/* ------------------------------------------------------------------------------------------ */
/* single-string insertion: mr_insert1 (mrope.c:42-68)                                         */
/* ------------------------------------------------------------------------------------------ */

void orc_insert1(orc_t *o, const uint8_t *str)
{
	int64_t tl[6], tu[6], l, u;
	const uint8_t *p;
	int a, b, is_srt = (o->so != 0), is_comp = (o->so == 2);
	for (u = 0, b = 0; b < 6; ++b) u += o->c[b][0];         /* number of strings so far */
	l = is_srt ? 0 : u;
	for (p = str, b = 0; *p; b = *p++) {
		int c = *p, bb;
		int64_t before = 0;
		if (l != u) {
			flat_rank2a(o, b, l, u, tl, tu);
			if (is_comp && c != 5) {                        /* RCLO: $ < T < G < C < A < N */
				for (a = 4; a > c; --a) l += tu[a] - tl[a];
				l += tu[0] - tl[0];
			} else for (a = 0; a < c; ++a) l += tu[a] - tl[a];
			flat_insert_run(o, b, l, c, 1);
			for (bb = 0; bb < b; ++bb) before += o->c[bb][c];
			l = before + tl[c]; u = before + tu[c];
		} else {
			l = flat_insert_run(o, b, l, c, 1);
			for (bb = 0; bb < b; ++bb) l += o->c[bb][c];
			u = l;
		}
	}
	flat_insert_run(o, b, l, 0, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* multi-string insertion, literal: mr_insert_multi / mr_insert_multi_aux (mrope.c:184-345)    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {        /* triple64_t, mrope.c:174-178 */
	int64_t l, u;
	int c;
	const uint8_t *p;
} ostr_t;

/* one rope, one round, strictly sequential (mrope.c:184-233) */
static void aux_seq(orc_t *o, int b, int64_t m, ostr_t *a, int is_comp)
{
	int64_t k, beg, i;
	for (k = 0; k < m; ++k) a[k].c = *a[k].p++;             /* mrope.c:189-190 */
	for (k = 1, beg = 0; k <= m; ++k) {
		if (k < m && a[k].u == a[k-1].u) continue;          /* group = equal u (mrope.c:192) */
		int64_t l = a[beg].l, u = a[beg].u, tl[6], tu[6], c[6], x;
		int s, step, sym;
		if (l == u && k == beg + 1) {                       /* singleton, empty interval (195-198) */
			a[beg].l = a[beg].u = flat_insert_run(o, b, l, a[beg].c, 1);
			beg = k;
			continue;
		}
		if (l == u) { memset(tl, 0, 48); memset(tu, 0, 48); }
		else flat_rank2a(o, b, l, u, tl, tu);               /* 199-202 */
		memset(c, 0, 48);
		for (i = beg; i < k; ++i) ++c[a[i].c];
		if (c[0]) flat_insert_run(o, b, l, 0, c[0]);        /* sentinels first (206) */
		x = l + c[0] + (tu[0] - tl[0]);
		s = is_comp ? 4 : 1; step = is_comp ? -1 : 1;       /* A..T or T..A (209-210) */
		for (i = 0, sym = s; i < 4; ++i, sym += step) {
			int64_t size = tu[sym] - tl[sym];
			if (c[sym]) {
				tl[sym] = flat_insert_run(o, b, x, sym, c[sym]);
				tu[sym] = tl[sym] + size;
			}
			x += c[sym] + size;
		}
		if (c[5]) {                                         /* N last (220-224) */
			int64_t size = tu[5] - tl[5];
			tl[5] = flat_insert_run(o, b, x, 5, c[5]);
			tu[5] = tl[5] + size;
		}
		for (i = beg; i < k; ++i) a[i].l = tl[a[i].c], a[i].u = tu[a[i].c];   /* 226-229 */
		beg = k;
	}
}

typedef void (*aux_fn)(orc_t *o, int b, int64_t m, ostr_t *a, int is_comp);

/* driver shared by the literal and the bulk variant (mrope.c:258-345 without the threads) */
static void insert_multi_driver(orc_t *o, int64_t len, const uint8_t *s, aux_fn aux)
{
	int64_t k, m, n0;
	int b, is_srt = (o->so != 0), is_comp = (o->so == 2);
	ostr_t *curr, *prev, *swap;
	const uint8_t *p, *q, *end = s + len;

	assert(len > 0 && s[len-1] == 0);                       /* mrope.c:268 */
	for (p = s, m = 0; p != end; ++p) m += (*p == 0);       /* 271-272 */
	curr = (ostr_t*)malloc(m * sizeof(ostr_t));
	prev = (ostr_t*)malloc(m * sizeof(ostr_t));
	for (p = q = s, k = 0; p != end; ++p)                   /* 275-276 */
		if (*p == 0) prev[k++].p = q, q = p + 1;

	for (b = 0, n0 = 0; b < 6; ++b) n0 += o->c[b][0];       /* 279 */
	for (k = 0; k < m; ++k) {                               /* 280-284 */
		if (is_srt) prev[k].l = 0, prev[k].u = n0;
		else prev[k].l = prev[k].u = n0 + k;
		prev[k].c = 0;
	}
	aux(o, 0, m, prev, is_comp);                            /* 285 */

	n0 = 0;
	while (m) {
		int64_t c[6], ac[6];
		ostr_t *qq[6];
		memset(c, 0, 48);
		for (k = n0; k < m; ++k) ++c[prev[k].c];            /* 303 */
		for (qq[0] = curr + n0, b = 1; b < 6; ++b) qq[b] = qq[b-1] + c[b-1];
		if (n0 + c[0] < m) {
			for (k = n0; k < m; ++k) *qq[prev[k].c]++ = prev[k];    /* stable scatter (306) */
			for (b = 0; b < 6; ++b) qq[b] -= c[b];
		}
		n0 += c[0];                                         /* finished strings (310) */
		for (b = 1; b < 6; ++b)
			if (c[b]) aux(o, b, c[b], qq[b], is_comp);      /* 327-329 */
		if (n0 == m) break;
		memset(ac, 0, 48);
		for (b = 1; b < 6; ++b) {                           /* 332-340 */
			int a;
			for (a = 0; a < 6; ++a) ac[a] += o->c[b-1][a];
			for (k = 0; k < c[b]; ++k) {
				ostr_t *t = &qq[b][k];
				t->l += ac[t->c]; t->u += ac[t->c];
			}
		}
		swap = curr, curr = prev, prev = swap;
	}
	free(curr); free(prev);
}

void orc_insert_multi_seq(orc_t *o, int64_t len, const uint8_t *s)
{
	insert_multi_driver(o, len, s, aux_seq);
}

/* ------------------------------------------------------------------------------------------ */
/* multi-string insertion, one bulk merge per rope per round                                   */
/*                                                                                             */
/* Same result as aux_seq, derived from it: with F_g the index of the    */
/* first member of group g in the bucket and P_g[a] the number of members of earlier groups    */
/* that insert a, the reference's coordinates l_g,u_g (which include all earlier groups of the */
/* round) correspond to l0 = l_g - F_g, u0 = u_g - F_g on the rope as it was BEFORE the round; */
/* rank(a, l) on the live rope = rank0(a, l0) + P_g[a]; and the interval [l,u) of a sorted     */
/* build is already ordered $,A,C,G,T,N (or $,T,G,C,A,N), so the run of new a's lands right    */
/* before the existing a's of the interval.                                                    */
/* ------------------------------------------------------------------------------------------ */

#define OCC_STEP 64

static void aux_bulk(orc_t *o, int b, int64_t m, ostr_t *a, int is_comp)
{
	static const int ord_fwd[6] = {0, 1, 2, 3, 4, 5}, ord_rc[6] = {0, 4, 3, 2, 1, 5};
	const int *ord = is_comp ? ord_rc : ord_fwd;            /* insertion order of the symbols */
	const int64_t n = o->n[b];
	const uint8_t *old = o->r[b];
	int64_t k, beg, i, j, nocc = n / OCC_STEP + 1;
	int64_t (*occ)[6] = (int64_t(*)[6])malloc(nocc * 48);   /* occ[i] = counts in [0, i*OCC_STEP) */
	int64_t *ins_e = (int64_t*)malloc(m * 8);                /* pre-round position of every new symbol */
	uint8_t *ins_a = (uint8_t*)malloc(m);
	int64_t *slot_of = (int64_t*)malloc(m * 8), *pg_of = (int64_t*)malloc(m * 8), *size_of = (int64_t*)malloc(m * 8);
	int64_t run[6], P[6];
	uint8_t *neu;

	memset(run, 0, 48);
	for (i = 0, j = 0; i <= n; ++i) {
		if (i % OCC_STEP == 0) memcpy(occ[j++], run, 48);
		if (i < n) ++run[old[i]];
	}
#define RANK0(x, out) do { int64_t _x = (x), _i; memcpy(out, occ[_x / OCC_STEP], 48); \
		for (_i = _x / OCC_STEP * OCC_STEP; _i < _x; ++_i) ++(out)[old[_i]]; } while (0)

	for (k = 0; k < m; ++k) a[k].c = *a[k].p++;
	memset(P, 0, 48);                                       /* P[a] = #a inserted by earlier groups */
	for (k = 1, beg = 0; k <= m; ++k) {
		if (k < m && a[k].u == a[k-1].u) continue;
		int64_t l0 = a[beg].l - beg, u0 = a[beg].u - beg, tl[6], tu[6], c[6], off[6], pos[6], acc;
		int t;
		RANK0(l0, tl);
		if (u0 != l0) RANK0(u0, tu); else memcpy(tu, tl, 48);
		memset(c, 0, 48);
		for (i = beg; i < k; ++i) ++c[a[i].c];
		/* walk the symbols in insertion order: slot of the first new x, and where it lands */
		for (t = 0, acc = 0, j = beg; t < 6; ++t) {
			int x = ord[t];
			off[x] = j; j += c[x];                          /* slot range of the new x's */
			pos[x] = l0 + acc; acc += tu[x] - tl[x];        /* in front of the existing x's of [l0,u0) */
		}
		for (i = beg; i < k; ++i) {
			int x = a[i].c;
			int64_t sl = off[x]++;
			ins_e[sl] = pos[x]; ins_a[sl] = x;
			slot_of[i] = sl; pg_of[i] = P[x]; size_of[i] = tu[x] - tl[x];
		}
		for (t = 0; t < 6; ++t) P[t] += c[t];
		beg = k;
	}
	/* ranks on the pre-round rope, then new coordinates (reference: return value of rope_insert_run) */
	for (k = 0; k < m; ++k) {
		int64_t r[6];
		RANK0(ins_e[slot_of[k]], r);
		a[k].l = r[a[k].c] + pg_of[k];
		a[k].u = a[k].l + size_of[k];
	}
	/* one merge: new symbol q goes in front of old symbol ins_e[q] */
	neu = (uint8_t*)malloc(n + m + 1);
	for (i = 0, j = 0, k = 0; k < n + m; ++k) {
		if (j < m && ins_e[j] <= i) neu[k] = ins_a[j++];
		else neu[k] = old[i++];
	}
	assert(i == n && j == m);
	for (j = 0; j < m; ++j) ++o->c[b][ins_a[j]];
	free(o->r[b]);
	o->r[b] = neu; o->n[b] = n + m; o->cap[b] = n + m + 1;
	free(occ); free(ins_e); free(ins_a); free(slot_of); free(pg_of); free(size_of);
#undef RANK0
}

void orc_insert_multi(orc_t *o, int64_t len, const uint8_t *s)
{
	insert_multi_driver(o, len, s, aux_bulk);
}

/* ------------------------------------------------------------------------------------------ */
/* 43+3 codec (rle.h:39-75)                                                                    */
/*   1 byte : 0lllleee            l < 16                                                       */
/*   2 bytes: 110lleee 10llllll   l < 256                                                      */
/*   4 bytes: 1110leee + 3 x 10llllll   l < 2^19                                               */
/*   8 bytes: 1111leee + 7 x 10llllll   l < 2^43                                               */
/* ------------------------------------------------------------------------------------------ */

int orc_rle_enc1(uint8_t *p, int c, int64_t l)
{
	int n, i;
	if (l < 16) { p[0] = (uint8_t)(l << 3 | c); return 1; }
	n = l < 256 ? 2 : l < (1LL << 19) ? 4 : 8;
	for (i = n - 1; i >= 1; --i) { p[i] = 0x80 | (l & 0x3f); l >>= 6; }
	p[0] = (uint8_t)((n == 2 ? 0xC0 : n == 4 ? 0xE0 : 0xF0) | l << 3 | c);
	return n;
}

int orc_rle_dec1(const uint8_t *p, int *c, int64_t *l)
{
	int n, i;
	int64_t v;
	*c = p[0] & 7;
	if ((p[0] & 0x80) == 0) { *l = p[0] >> 3; return 1; }
	if ((p[0] >> 5) == 6) { n = 2; v = (p[0] >> 3) & 3; }
	else { n = (p[0] & 0x10) ? 8 : 4; v = (p[0] >> 3) & 1; }
	for (i = 1; i < n; ++i) v = v << 6 | (p[i] & 0x3f);
	*l = v;
	return n;
}
