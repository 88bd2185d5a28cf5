/* rle.c -- run-length leaf codec, host side (API of include/rle.h).
 *
 * Own implementation of the operations ropebwt2 defines in /root/reference/rle.c.  It is written
 * for clarity, not speed: every operation decodes the block front to back into a small run array,
 * edits the array and re-encodes.  The hot path of this project does not go through here -- leaf
 * edits during batch construction happen in the HIP kernels (rb2_merge.h); this file serves
 * mr_insert1, rank queries and .fmr handling on the host.
 */
#include <string.h>
#include <stdio.h>
#include <assert.h>
#include "rle.h"

/* see rle.h; indexed by bits 3..5 of the first byte of a multi-byte run */
const uint8_t rle_auxtab[8] = { 0x01, 0x11, 0x21, 0x31, 0x03, 0x13, 0x07, 0x17 };

typedef struct { int c; int64_t l; int off, nb; } run_t;   /* symbol, length, byte offset, byte size */

int rle_insert_cached(uint8_t *block, int64_t x, int a, int64_t rl, int64_t cnt[6], const int64_t ec[6], int *beg, int64_t bc[6])
{
	uint16_t *np = rle_nptr(block);
	uint8_t *body = block + 2, *end = body + *np, *p = body;
	uint8_t tmp[24];
	run_t cur = { -1, 0, 0, 0 }, nxt;
	int64_t z = 0;       /* symbols in front of p */
	int n_old, n_new, tail;
	(void)ec;
	/* the position cache is ours to define: we always restart from the front of the block */
	if (beg) *beg = 0;
	if (bc) memset(bc, 0, 48);
	memset(cnt, 0, 48);
	/* find the run that holds symbol x-1 (the last run starting before x) */
	while (p < end && z < x) {
		cur.off = (int)(p - body);
		cur.nb = rle_dec1_fn(p, &cur.c, &cur.l);
		p += cur.nb; z += cur.l; cnt[cur.c] += cur.l;
	}
	assert(z >= x);
	if (cur.c >= 0 && z > x) {                       /* x falls strictly inside cur */
		const int64_t pre = cur.l - (z - x);
		cnt[cur.c] -= z - x;
		if (cur.c == a) n_new = rle_enc1(tmp, a, cur.l + rl);
		else {
			n_new  = rle_enc1(tmp, cur.c, pre);
			n_new += rle_enc1(tmp + n_new, a, rl);
			n_new += rle_enc1(tmp + n_new, cur.c, cur.l - pre);
		}
		p = body + cur.off; n_old = cur.nb;
	} else if (cur.c == a) {                         /* right behind a run of a: grow it */
		n_new = rle_enc1(tmp, a, cur.l + rl);
		p = body + cur.off; n_old = cur.nb;
	} else if (p < end && (nxt.nb = rle_dec1_fn(p, &nxt.c, &nxt.l), nxt.c == a)) {   /* in front of a run of a */
		n_new = rle_enc1(tmp, a, nxt.l + rl);
		n_old = nxt.nb;
	} else {                                         /* between two other runs, or empty block */
		n_new = rle_enc1(tmp, a, rl);
		n_old = 0;
	}
	tail = (int)(end - (p + n_old));
	if (n_new != n_old && tail > 0) memmove(p + n_new, p + n_old, tail);
	memcpy(p, tmp, n_new);
	*np = (uint16_t)(*np + n_new - n_old);
	return *np;
}

int rle_insert(uint8_t *block, int64_t x, int a, int64_t rl, int64_t cnt[6], const int64_t end_cnt[6])
{
	int beg = 0; int64_t bc[6];
	return rle_insert_cached(block, x, a, rl, cnt, end_cnt, &beg, bc);
}

void rle_split(uint8_t *block, uint8_t *new_block)
{
	const int n = *rle_nptr(block);
	uint8_t *body = block + 2, *q = body, *end = body + n, *mid = body + n / 2;
	while (q < mid) {                                /* first run boundary at or behind the middle... */
		int c; int64_t l;
		uint8_t *nx = q + rle_dec1_fn(q, &c, &l);
		if (nx > mid) break;                         /* ...but never in front of it, like rle.c:102-103 */
		q = nx;
	}
	memcpy(new_block + 2, q, end - q);
	*rle_nptr(new_block) = (uint16_t)(end - q);
	*rle_nptr(block) = (uint16_t)(q - body);
}

void rle_count(const uint8_t *block, int64_t cnt[6])
{
	const uint8_t *q = block + 2, *end = q + *(const uint16_t*)block;
	while (q < end) { int c; int64_t l; q += rle_dec1_fn(q, &c, &l); cnt[c] += l; }
}

void rle_rank2a(const uint8_t *block, int64_t x, int64_t y, int64_t *cx, int64_t *cy, const int64_t ec[6])
{
	const uint8_t *q = block + 2, *end = q + *(const uint16_t*)block;
	int64_t z = 0, run[6] = { 0, 0, 0, 0, 0, 0 };
	int a, doney = (cy == 0), donex = 0;
	(void)ec;
	if (y < x) y = x;
	while (q < end && !(donex && doney)) {
		int c; int64_t l;
		q += rle_dec1_fn(q, &c, &l);
		if (!donex && z + l >= x) { for (a = 0; a < 6; ++a) cx[a] += run[a]; cx[c] += x - z; donex = 1; }
		if (!doney && z + l >= y) { for (a = 0; a < 6; ++a) cy[a] += run[a]; cy[c] += y - z; doney = 1; }
		z += l; run[c] += l;
	}
	if (!donex) for (a = 0; a < 6; ++a) cx[a] += run[a];
	if (!doney) for (a = 0; a < 6; ++a) cy[a] += run[a];
}

void rle_print(const uint8_t *block, int expand)
{
	const uint8_t *q = block + 2, *end = q + *(const uint16_t*)block;
	while (q < end) {
		int c; int64_t l, i;
		q += rle_dec1_fn(q, &c, &l);
		if (expand) for (i = 0; i < l; ++i) putchar("$ACGTN"[c]);
		else printf("%c%ld", "$ACGTN"[c], (long)l);
	}
	putchar('\n');
}
