/* mrope.h -- multi-rope: the BWT of a string collection as six ropes B_$,B_A,B_C,B_G,B_T,B_N.
 *
 * Drop-in replacement for /root/reference/mrope.h:6-116: same names, struct layouts, argument
 * meaning and (absent) error behaviour.  The difference is behind mr_insert_multi: the batch
 * insertion (mrope.c:258-345) runs on an MI355X through the C ABI of rb2_hip.h; nothing on that
 * path executes on the CPU, and a missing GPU aborts with a message.  The BWT lives in HBM between
 * calls; the host ropes are materialised lazily when a caller needs them (iteration, dump,
 * rank, mr_insert1), while r[a]->c[] is always current.
 */
#ifndef RB2_MROPE_H_
#define RB2_MROPE_H_

#include "rope.h"

#define MR_SO_IO    0   /* input order */
#define MR_SO_RLO   1   /* reverse-lexicographical order */
#define MR_SO_RCLO  2   /* reverse-complement lexicographical order */

typedef struct {
	uint8_t so;          /* sorting order, MR_SO_* */
	int thr_min;         /* kept for API compatibility (mrope.c:35-40); the GPU path has no thread switch */
	rope_t *r[6];
} mrope_t;

typedef struct {
	mrope_t *r;
	int a, to_free;
	rpitr_t i;
} mritr_t;

#ifdef __cplusplus
extern "C" {
#endif

mrope_t *mr_init(int max_nodes, int block_len, int sorting_order);                 /* mrope.c:14-25   */
void     mr_destroy(mrope_t *r);                                                   /* mrope.c:27-33   */
int      mr_thr_min(mrope_t *r, int thr_min);                                      /* mrope.c:35-40   */
/* insert one string; str is the REVERSE of the string, 0-terminated (CPU path)       mrope.c:42-68   */
int64_t  mr_insert1(mrope_t *r, const uint8_t *str);
/* insert all strings of s (concatenated, each reversed and 0-terminated, s[len-1]==0); runs on the
 * GPU; is_thr is accepted and ignored                                                mrope.c:258-345 */
void     mr_insert_multi(mrope_t *mr, int64_t len, const uint8_t *s, int is_thr);
void     mr_rank2a(const mrope_t *mr, int64_t x, int64_t y, int64_t *cx, int64_t *cy);   /* mrope.c:70-105 */
#define  mr_rank1a(mr, x, cx) mr_rank2a(mr, x, -1, cx, 0)
void     mr_itr_first(mrope_t *r, mritr_t *i, int to_free);                        /* mrope.c:111-115 */
const uint8_t *mr_itr_next_block(mritr_t *i);                                      /* mrope.c:117-130 */
void     mr_print_tree(const mrope_t *mr);                                         /* mrope.c:162-168 */
void     mr_dump(mrope_t *mr, FILE *fp);                                           /* mrope.c:136-143 */
mrope_t *mr_restore(FILE *fp);                                                     /* mrope.c:145-160 */
mrope_t *mr_restore_runs(FILE *fp);                                                /* rb2 extension: the same file, kept as run bytes until the device (mr_insert_multi) or a host operation needs it */

/* ---- additions ------------------------------------------------------------------------------ */
/* make the host ropes reflect the device BWT now (otherwise done on demand) */
void     mr_sync_host(mrope_t *mr);
/* 1 when the six host B+ trees hold the current BWT.  mr_itr_first / mr_itr_next_block, mr_dump and mr_stream_runs do not
 * build them for an index that lives on the device (or in restored run bytes): the leaves are cut from the run stream */
int      mr_host_resident(const mrope_t *mr);
/* the rb2_hip_t behind this mrope (NULL until the first mr_insert_multi) */
void    *mr_hip_handle(mrope_t *mr);
/* ... or the rb2_hip_multi_t, when RB2_HIP_DEVICES lists several devices: the index is then sharded over them and
 * mr_insert_multi drives all of them inside the call, as the reference drives its worker threads (mrope.c:287-296, 312-329) */
void    *mr_hip_multi_handle(mrope_t *mr);
void     mr_wait(mrope_t *mr);                         /* rb2 extension: mr_insert_multi may return while the GPU is still inserting (rb2_hip.h: rb2_hip_set_lazy); wait for it -- only needed to time it */
void     mr_prefetch(mrope_t *mr, const uint8_t *s, int64_t n_final, int64_t capacity);   /* rb2 extension: start uploading the batch that is being assembled (rb2_hip_prefetch) */
int64_t  mr_auto_batch_bytes(mrope_t *mr);                                              /* rb2 extension: batch size for `-m auto`, from the device's free memory */
void     mr_reserve(mrope_t *mr, int64_t batch_bytes, int64_t total_symbols);       /* rb2 extension: capacity hint (rb2_hip_reserve); 0 = unknown */
/* hand the whole BWT ($,A,C,G,T,N ropes in order) to cb as 43+3 run bytes, a chunk at a time, WITHOUT building the host
 * B+ trees: straight from HBM when the device holds the current BWT, else from the host leaves.  Equivalent to walking
 * mr_itr_first / mr_itr_next_block (mrope.c:107-131) and decoding every block. */
void     mr_stream_runs(mrope_t *mr, void (*cb)(void *user, const uint8_t *runs, int64_t n_bytes), void *user);

#ifdef __cplusplus
}
#endif

/* header-inlined helpers, as in the reference (mrope.h:86-116); r[a]->c[] is always current */
static inline int64_t mr_get_c(const mrope_t *mr, int64_t c[6])
{
	int a, b; int64_t tot = 0;
	for (b = 0; b < 6; ++b) c[b] = 0;
	for (a = 0; a < 6; ++a)
		for (b = 0; b < 6; ++b) { c[b] += mr->r[a]->c[b]; tot += mr->r[a]->c[b]; }
	return tot;
}
static inline int64_t mr_get_ac(const mrope_t *mr, int64_t ac[7])
{
	int a; int64_t c[6], tot = mr_get_c(mr, c);
	for (ac[0] = 0, a = 0; a < 6; ++a) ac[a+1] = ac[a] + c[a];
	return tot;
}
static inline int64_t mr_get_tot(const mrope_t *mr) { int64_t c[6]; return mr_get_c(mr, c); }

#endif
