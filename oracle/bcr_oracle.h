/* oracle/bcr_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the ropebwt2 hot path (multi-string BCR insertion, mr_insert_multi,
 * /root/reference/mrope.c:258-345) over the simplest possible data structure: each of the six
 * ropes B_$,B_A,B_C,B_G,B_T,B_N is a flat byte array of nt6 symbols.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (ropebwt2_amd/) never links, imports or executes anything under oracle/.
 *
 * Parity pin: checked against the real reference (oracle/_ref, built from /root/reference by
 * oracle/Makefile) and against the golden vectors of SURVEY.md 8c in tests/test_oracle.py.
 */
#ifndef BCR_ORACLE_H_
#define BCR_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_s orc_t;

/* sorting order: 0 input order, 1 RLO, 2 RCLO (mrope.h:6-8) */
orc_t  *orc_create(int sorting_order);
void    orc_destroy(orc_t *o);

/* literal sequential restatement of mr_insert_multi + mr_insert_multi_aux (mrope.c:184-345):
 * one real insert per run, every rank taken on the rope as it is at that moment.  O(n) per
 * insert -- small inputs only. */
void    orc_insert_multi_seq(orc_t *o, int64_t len, const uint8_t *s);

/* per-round bulk restatement (SURVEY.md section 7 "data-parallel restatement"): identical
 * result, O(n) per round.  Used for inputs up to ~1e7 symbols. */
void    orc_insert_multi(orc_t *o, int64_t len, const uint8_t *s);

/* single-string insertion, restating mr_insert1 (mrope.c:42-68). str = reversed string, 0-terminated */
void    orc_insert1(orc_t *o, const uint8_t *str);

int64_t        orc_rope_len(const orc_t *o, int b);
const uint8_t *orc_rope_ptr(const orc_t *o, int b);
void           orc_counts(const orc_t *o, int64_t c[36]);   /* c[b*6+a] = #a in rope b (rope_t.c, rope.h:19) */
int64_t        orc_total(const orc_t *o);
int64_t        orc_bwt(const orc_t *o, uint8_t *out);       /* concatenation of the six ropes, nt6 codes */

/* rank of all six symbols in [0,x) of the whole BWT (mr_rank2a semantics, mrope.c:70-105) */
void    orc_rank1a(const orc_t *o, int64_t x, int64_t cx[6]);

/* 43+3 run-length codec restated from rle.h:39-75; returns bytes written / consumed */
int     orc_rle_enc1(uint8_t *p, int c, int64_t l);
int     orc_rle_dec1(const uint8_t *p, int *c, int64_t *l);

#ifdef __cplusplus
}
#endif
#endif
