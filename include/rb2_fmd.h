/* rb2_fmd.h -- writer for fermi's FMD format (run-length delta BWT + rank frames), the artefact
 * whose bytes must match the reference (`ropebwt2 -d`).  Replaces the encode side of
 * /root/reference/rld0.c:26-244 (rld_init/rld_enc/rld_enc_finish/rld_rank_index/rld_dump) with a
 * single streaming object.  Format notes: SURVEY.md section 8f-1.
 */
#ifndef RB2_FMD_H_
#define RB2_FMD_H_

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rb2_fmd_s rb2_fmd_t;

rb2_fmd_t *rb2_fmd_init(void);                                 /* rld_init(6,3) + rld_itr_init, main.c:274-276 */
void rb2_fmd_push(rb2_fmd_t *f, int64_t len, int sym);         /* rld_enc, rld0.c:153-161 (adjacent equal symbols merge) */
void rb2_fmd_push_runs(rb2_fmd_t *f, const uint8_t *runs, int64_t n_bytes);   /* the same for a chunk of 43+3 run bytes (rle.h:39-75) */
void rb2_fmd_finish(rb2_fmd_t *f);                             /* rld_enc_finish + rld_rank_index, rld0.c:163-217 */
int  rb2_fmd_write(const rb2_fmd_t *f, FILE *fp);              /* rld_dump, rld0.c:223-244 */
int  rb2_fmd_write_path(const rb2_fmd_t *f, const char *path);
void rb2_fmd_counts(const rb2_fmd_t *f, int64_t c[7]);         /* total, $, A, C, G, T, N */
void rb2_fmd_destroy(rb2_fmd_t *f);

/* The same file from several threads: runs are pushed as they come (any chunking, whole runs per call), worker threads encode
 * segments of the stream speculatively and one thread stitches them into the exact sequential result (fmd.c).
 * rb2_fmdp_finish() joins the workers and returns a FINISHED rb2_fmd_t (rb2_fmd_write / rb2_fmd_counts / rb2_fmd_destroy). */
typedef struct rb2_fmdp_s rb2_fmdp_t;
rb2_fmdp_t *rb2_fmdp_init(int n_threads, int64_t segment_bytes /* 0: default */);
void rb2_fmdp_push_runs(rb2_fmdp_t *p, const uint8_t *runs, int64_t n_bytes);
rb2_fmd_t *rb2_fmdp_finish(rb2_fmdp_t *p);
/* Stream the encoded words to `fd` (a regular file, not O_APPEND; the .fmd starts at byte `offset`) while they are produced;
 * rb2_fmd_write() on the finished index then adds header, tail and rank frames and leaves its FILE (same descriptor) positioned
 * behind the index.  Returns -1 (nothing changes, rb2_fmd_write writes everything) when fd cannot be written at offsets.
 * Call before the first rb2_fmdp_push_runs.  The file is byte-identical either way (rld_dump, rld0.c:207-229). */
int rb2_fmdp_set_output(rb2_fmdp_t *p, int fd, int64_t offset);
/* optional: the stream will hold n_symbols symbols -- the output array is sized once (and backed by huge pages where the host has
 * them) instead of growing as it fills.  Call before the first rb2_fmdp_push_runs. */
void rb2_fmdp_expect(rb2_fmdp_t *p, int64_t n_symbols);

#ifdef __cplusplus
}
#endif
#endif
