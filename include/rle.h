/* rle.h -- run-length leaf codec of the rope ("43+3" codec), host side.
 *
 * Same exported names, argument meaning and block format as /root/reference/rle.h:11-75, so code
 * written against ropebwt2's rle layer links against libropebwt2.so unchanged.  A block is
 *     uint16 n_bytes | run | run | ...
 * and a run of l copies of symbol c (0..5) is 1, 2, 4 or 8 bytes:
 *     0lllleee                      l < 2^4
 *     110lleee 10llllll             l < 2^8
 *     1110leee 10llllll x3          l < 2^19
 *     1111leee 10llllll x7          l < 2^43
 * (c in the low three bits of the first byte).  The GPU engine (rb2_hip.h) emits and accepts the
 * same byte streams; its leaves only ever use the 1-byte form.
 */
#ifndef RB2_RLE_H_
#define RB2_RLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* a block must keep this many spare bytes before an insertion (reference: rle.h:35) */
#define RLE_MIN_SPACE 18
/* the byte count in front of the runs */
#define rle_nptr(block) ((uint16_t*)(block))

/* rle_auxtab[(first_byte>>3)&7] for a multi-byte run: high nibble = VALUE of the length bits held
 * by the first byte, low nibble = number of continuation bytes (reference: rle.c:7) */
extern const uint8_t rle_auxtab[8];

/* Insert rl copies of a after the first x symbols of block.  cnt[] receives the symbol counts of
 * [0,x); ec[] are the counts of the whole block (known to the caller from the parent node).
 * (*beg, bc[]) is an opaque position cache owned by this module: pass zeros to start, pass the
 * same variables again for a later insertion into the same block.  Returns the new byte count.
 * reference: rle.c:10-89 */
int  rle_insert_cached(uint8_t *block, int64_t x, int a, int64_t rl, int64_t cnt[6], const int64_t ec[6], int *beg, int64_t bc[6]);
/* same without a cache (rle.c:91-97) */
int  rle_insert(uint8_t *block, int64_t x, int a, int64_t rl, int64_t cnt[6], const int64_t end_cnt[6]);
/* move the second half of block (cut at a run boundary) to the empty new_block (rle.c:99-107) */
void rle_split(uint8_t *block, uint8_t *new_block);
/* add the symbol counts of block to cnt[] (rle.c:109-118) */
void rle_count(const uint8_t *block, int64_t cnt[6]);
/* add the counts of [0,x) to cx[] and, when cy != NULL, of [0,y) to cy[] (rle.c:134-191) */
void rle_rank2a(const uint8_t *block, int64_t x, int64_t y, int64_t *cx, int64_t *cy, const int64_t ec[6]);
#define rle_rank1a(block, x, cx, ec) rle_rank2a(block, x, -1, cx, 0, ec)
/* debugging aid (rle.c:120-132) */
void rle_print(const uint8_t *block, int expand);

/* encode one run at p; returns the number of bytes written (rle.h:53-75) */
static inline int rle_enc1(uint8_t *p, int c, int64_t l)
{
	int n, i;
	if (l < 16) { p[0] = (uint8_t)(l << 3 | c); return 1; }
	n = l < 256 ? 2 : l < (1LL << 19) ? 4 : 8;
	for (i = n - 1; i > 0; --i, l >>= 6) p[i] = (uint8_t)(0x80 | (l & 0x3f));
	p[0] = (uint8_t)((n == 2 ? 0xC0 : n == 4 ? 0xE0 : 0xF0) | l << 3 | c);
	return n;
}

/* decode the run at q into (*c,*l); returns its size in bytes */
static inline int rle_dec1_fn(const uint8_t *q, int *c, int64_t *l)
{
	const unsigned h = q[0];
	int n, i;
	int64_t v;
	*c = (int)(h & 7);
	if (h < 0x80) { *l = h >> 3; return 1; }
	n = 1 + (rle_auxtab[h >> 3 & 7] & 0xf);
	v = rle_auxtab[h >> 3 & 7] >> 4;
	for (i = 1; i < n; ++i) v = v << 6 | (q[i] & 0x3f);
	*l = v;
	return n;
}
/* reference spelling (rle.h:39-51): decode one run and advance the pointer p */
#define rle_dec1(p, c, l) do { int rb2_c_; int64_t rb2_l_; (p) += rle_dec1_fn((const uint8_t*)(p), &rb2_c_, &rb2_l_); (c) = rb2_c_; (l) = rb2_l_; } while (0)

#ifdef __cplusplus
}
#endif
#endif
