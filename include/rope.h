/* rope.h -- one rope: a B+ tree whose leaves are run-length blocks (rle.h), host side.
 *
 * Exported names, struct layouts and semantics follow /root/reference/rope.h:8-56 so that callers
 * (and the header-inlined helpers of mrope.h) keep working.  The implementation behind it
 * (ropebwt2_amd/csrc/host/rope.c) is our own: bottom-up splitting, bulk loading from a run
 * stream (how the GPU-built BWT is materialised), arena allocation.
 */
#ifndef RB2_ROPE_H_
#define RB2_ROPE_H_

#include <stdint.h>
#include <stdio.h>

#define ROPE_MAX_DEPTH 80
#define ROPE_DEF_MAX_NODES 64
#define ROPE_DEF_BLOCK_LEN 512

/* One entry of an internal bucket (an array of max_nodes entries).  n and is_bottom are only
 * meaningful in the FIRST entry of a bucket.  At the bottom level p points to a leaf block. */
typedef struct rpnode_s {
	struct rpnode_s *p;
	uint64_t l:54, n:9, is_bottom:1;    /* l: symbols below this entry */
	int64_t c[6];                        /* per-symbol counts below this entry */
} rpnode_t;

typedef struct {
	int32_t max_nodes, block_len;        /* both even */
	int64_t c[6];                        /* marginal counts of the whole rope */
	rpnode_t *root;
	void *node, *leaf;                   /* allocators (opaque) */
} rope_t;

typedef struct {                         /* caller-allocated iterator over the leaves */
	const rope_t *rope;
	const rpnode_t *pa[ROPE_MAX_DEPTH];
	int ia[ROPE_MAX_DEPTH];
	int d;
} rpitr_t;

typedef struct {                         /* per-sweep leaf cache handed to rope_insert_run */
	int beg;
	int64_t bc[6];
	uint8_t *p;
} rpcache_t;

#ifdef __cplusplus
extern "C" {
#endif

rope_t *rope_init(int max_nodes, int block_len);                                   /* rope.c:55-69  */
void    rope_destroy(rope_t *rope);                                                /* rope.c:71-76  */
/* insert rl copies of a after x symbols; returns #a in [0,x) before the insertion    rope.c:114-148 */
int64_t rope_insert_run(rope_t *rope, int64_t x, int a, int64_t rl, rpcache_t *cache);
/* cx[] = counts of [0,x); cy[] = counts of [0,y) when cy != NULL and y >= x          rope.c:179-194 */
void    rope_rank2a(const rope_t *rope, int64_t x, int64_t y, int64_t *cx, int64_t *cy);
#define rope_rank1a(rope, x, cx) rope_rank2a(rope, x, -1, cx, 0)

void rope_itr_first(const rope_t *rope, rpitr_t *i);                               /* rope.c:200-206 */
const uint8_t *rope_itr_next_block(rpitr_t *i);                                    /* rope.c:208-219 */

void    rope_print_node(const rpnode_t *p);                                        /* rope.c:225-251 */
void    rope_dump(const rope_t *r, FILE *fp);                                      /* rope.c:270-275 */
int64_t rope_dump_size(const rope_t *r);                                           /* rb2 extension: bytes rope_dump() writes */
int     rope_dump_at(const rope_t *r, int fd, int64_t off);                         /* rb2 extension: the same bytes at offset off of a regular file (pwrite); 0 = ok */
int     rope_dump_nparts(const rope_t *r);                                          /* rb2 extension: the dump as independently sized and written parts (root header; one per child of the root) */
int64_t rope_dump_part_size(const rope_t *r, int part);
int     rope_dump_part_at(const rope_t *r, int part, int fd, int64_t off);
rope_t *rope_restore(FILE *fp);                                                    /* rope.c:308-318 */

/* ---- additions (not in the reference) ---------------------------------------------------- */
/* replace the content of an EMPTY rope by the symbols of a 43+3 run stream (bulk load) */
void    rope_load_runs(rope_t *rope, const uint8_t *rle, int64_t n_bytes);
void    rope_load_runs_mt(rope_t *rope, const uint8_t *rle, int64_t n_bytes, int n_threads);   /* rb2 extension: the same tree, byte for byte, built by several threads */
/* rb2 extension: rope_dump()'s bytes of the tree rope_load_runs_mt(max_nodes, block_len) would build from the run stream, without
 * building it: prepare (canonical stream + leaf boundaries: the size is known), then write at an offset of a regular file
 * (pwrite, n_threads; frees the handle; 0 = ok).  The stream must stay valid until prepare returns. */
typedef struct rope_rdump_s rope_rdump_t;
rope_rdump_t *rope_rdump_prepare(const uint8_t *rle, int64_t n_bytes, int max_nodes, int block_len, int n_threads);
int64_t rope_rdump_size(const rope_rdump_t *d);
int     rope_rdump_write(rope_rdump_t *d, int fd, int64_t off);
/* ... or walk its leaves (mrope.c:117-130 without the tree): leaf k as a leaf block -- u16 byte count + run bytes -- in blk
 * (block_len bytes or more; returns the byte count); rope_rdump_free() instead of rope_rdump_write() when done */
int64_t rope_rdump_nleaves(const rope_rdump_t *d);
int     rope_rdump_block(const rope_rdump_t *d, int64_t k, uint8_t *blk);
void    rope_rdump_free(rope_rdump_t *d);
/* append all run bytes of the rope to a malloc'ed buffer; returns the byte count */
int64_t rope_export_runs(const rope_t *rope, uint8_t **out);

#ifdef __cplusplus
}
#endif
#endif
