/* rb2_hip.h -- C ABI of the MI355X (gfx950) multi-string BWT insertion engine.
 *
 * This is the drop-in boundary for ropebwt2's hot path.  The host side stays plain C; every
 * symbol below is `extern "C"`, takes plain pointers and sizes, and is what the `mrope` layer
 * (include/mrope.h, our replacement for /root/reference/mrope.h) binds to:
 *
 *   reference interface replaced                         entry point here
 *   ---------------------------------------------------  -----------------------------------------
 *   mr_insert_multi()            mrope.c:258-345         rb2_hip_insert_multi[_dev]()
 *     mr_insert_multi_aux()      mrope.c:184-233           (k_prep / k_merge / k_advance kernels)
 *     rope_insert_run()          rope.c:114-148            (k_merge: rank + positional insert)
 *     rope_rank2a()              rope.c:179-194            (k_prep: interval sizes)
 *     rle_insert_cached/rank2a   rle.c:10-89, 134-191      (k_merge: run-length leaf decode/encode)
 *   rope_t.c[6] marginal counts  rope.h:19, mrope.h:86   rb2_hip_get_counts()
 *   mr_itr_first/next_block      mrope.c:111-130         rb2_hip_rope_bytes() + rb2_hip_download_rope()
 *   mr_restore / rope_restore    mrope.c:145, rope.c:308 rb2_hip_load_ropes()
 *
 * Run-length byte streams crossing this boundary use ropebwt2's "43+3" codec (rle.h:39-75), so
 * the host can splice them straight into rope leaves.  The device itself only ever emits the
 * 1-byte form (run length < 16), which is a valid subset of that codec.
 *
 * Limits of this build (checked; a violation prints a message and abort()s):
 *   - fewer than 2^32 - 1024 strings per batch (string ids are 32 bit on the device; ropebwt2's default -m10g holds at most
 *     ~10^10 one-symbol strings, so this only excludes batches of degenerate reads);
 *   - positions inside one sub-rope below 2^48 (the sharded wire format packs l into 48 bits, rb2_device.h ShardRec);
 *   - at most 64 ranks in the sharded protocol (RB2_MULTI_MAX_RANKS; 31 sub-ropes exist; more than 16 ranks carry no additional load on DNA);
 *   - one dense merge launch covers at most 2^32 threads = 2^26 windows of 5376 symbols: about 360 G symbols per GPU and round
 *     (checked per round, with a message); larger indexes are what the sharded build is for;
 *   - in-place (sparse) rounds are used for batches of fewer than 2^27 strings (one wave per four touched leaves in one launch);
 *     larger batches simply stay on the dense path -- a performance boundary, not an error;
 *   - leaf slots of one pool are addressed with 32 bits in the sparse layout (the work orders, the split list): 2^32 slots = 4.3 T symbols;
 *   - symbols must be nt6 codes 0..5, the buffer must end with a sentinel (mrope.c:268);
 *   - the index lives in HBM: 2 x 0.38 B per symbol (dense layout) plus ~100 B per string of the batch; running out of device
 *     memory reports the size that was needed.
 *
 * Error convention: like the reference (mrope.c has none), functions do not return error codes
 * for programming errors; any HIP failure or a missing GPU prints a message to stderr and
 * abort()s -- there is no CPU fallback behind this ABI.
 */
#ifndef RB2_HIP_H_
#define RB2_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rb2_hip_s rb2_hip_t;

/* sorting orders, same values as MR_SO_IO / MR_SO_RLO / MR_SO_RCLO (mrope.h:6-8) */
#define RB2_SO_IO   0
#define RB2_SO_RLO  1
#define RB2_SO_RCLO 2

/* Fatal errors (a HIP call that fails, a missing GPU, a malformed batch, out of device memory): a message on stderr, then abort() --
 * the reference's own convention on this path (asserts and unchecked mallocs).  A host program that wants a say
 * installs a handler: it is called with the message before abort() and may log, release what it holds, or leave through longjmp /
 * exit (the handle that failed must not be used again; others may).
 * With N GPUs behind one handle (rb2_hip_multi_*) a failure may be detected on the host thread of the rank it happens on: that
 * thread records the message and ends, the threads of the other ranks leave at their next round barrier, and the handler is called
 * -- once, with the first message -- on the thread that called the API, after all rank threads are joined: the same rules as above. */
typedef void (*rb2_hip_fatal_cb)(void *user, const char *message);
void rb2_hip_set_fatal_handler(rb2_hip_fatal_cb cb, void *user);

/* number of visible HIP devices (0 when there is no GPU; never aborts) */
int rb2_hip_device_count(void);

/* create an empty six-rope BWT on `device` (mr_init, mrope.c:14-25).  abort()s without a GPU. */
rb2_hip_t *rb2_hip_create(int device, int sorting_order);
void rb2_hip_destroy(rb2_hip_t *h);
int  rb2_hip_sorting_order(const rb2_hip_t *h);
/* empty the index again (what mr_destroy + mr_init would do) but keep every buffer the handle has grown */
void rb2_hip_reset(rb2_hip_t *h);

/* mr_insert_multi (mrope.c:258): insert all strings of `s` (concatenated, each REVERSED and
 * 0-terminated, nt6 codes 0..5, s[len-1]==0).  `s` is a host buffer, borrowed for the call. */
void rb2_hip_insert_multi(rb2_hip_t *h, int64_t len, const uint8_t *s);

/* rb2_hip_insert_multi may return as soon as the batch text is on the device and its rounds are queued (default; RB2_HIP_LAZY_INSERT=0
 * or rb2_hip_set_lazy(h, 0): only when the device is done): `s` is the caller's again, and what it does next -- parse the next batch,
 * upload it (the text goes to a second device buffer on a copy stream) -- runs beside the kernels.  Every other entry point waits for
 * the queued rounds before it looks at the index; rb2_hip_wait does only that.  rb2_hip_last_batch_counts: what that batch adds to
 * the count matrix (layout of rb2_hip_get_counts), computed from its text alone and valid at once -- how mr_insert_multi keeps
 * mr_get_c() truthful (mrope.c:332-340) without waiting; returns 0 when the last insert was not a host-buffer one. */
void rb2_hip_set_lazy(rb2_hip_t *h, int on);
void rb2_hip_wait(rb2_hip_t *h);
int  rb2_hip_last_batch_counts(rb2_hip_t *h, int64_t d[36]);

/* optional: bytes [0, n_final) of the buffer a LATER rb2_hip_insert_multi(h, len >= n_final, s) will pass are final -- start
 * uploading them now (copy stream, second text buffer), concurrently with an insert running on another thread.  capacity = the
 * largest len that call may have.  `s` must stay where it is until that call; s == NULL cancels (waits for copies in flight:
 * call it before reallocating or freeing a buffer that was announced).  (The reference reads and inserts in one thread,
 * main.c:238-242; this is what lets the PCIe crossing of batch k+1 hide behind the insertion of batch k.) */
void rb2_hip_prefetch(rb2_hip_t *h, const uint8_t *s, int64_t n_final, int64_t capacity);
/* A caller that hands the same host buffers to rb2_hip_insert_multi again and again (batch after batch, as main.c:238 does with its
 * one read buffer) can page-lock them once: the batch then crosses PCIe by DMA straight from the caller's memory instead of through the
 * runtime's staging copies of pageable memory.  Thin wrappers of hipHostRegister / hipHostUnregister, so that a plain-C host needs no HIP
 * header; 0 on success.  Optional: an unregistered buffer works as before.  (No counterpart in the reference: mrope.c:258 takes a
 * pointer and reads it on the CPU.) */
int rb2_hip_host_register(void *p, int64_t nbytes);
int rb2_hip_host_unregister(void *p);

/* free / total memory of a device in bytes (0, 0 without a usable GPU; never aborts): what `ropebwt2 -m auto` sizes its batches from */
void rb2_hip_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes);

/* same, but `s_dev` already resides in this device's HBM (no PCIe transfer in the call) */
void rb2_hip_insert_multi_dev(rb2_hip_t *h, int64_t len, const uint8_t *s_dev);

/* c[b*6+a] = number of symbol a in rope b == mr->r[b]->c[a] (rope.h:19) */
void rb2_hip_get_counts(rb2_hip_t *h, int64_t c[36]);

/* rope b as a run-length byte stream (43+3 codec, 1-byte runs only).  rb2_hip_rope_bytes gives
 * the exact size; download copies it to host memory `dst` and returns the byte count. */
int64_t rb2_hip_rope_bytes(rb2_hip_t *h, int b);
int64_t rb2_hip_download_rope(rb2_hip_t *h, int b, uint8_t *dst);
/* the same stream handed to a callback in pieces of whole runs (up to 32 MiB per call; the buffer is only valid during
 * the call), without a host copy of the whole rope: what an .fmd / text writer needs (replaces the walk over mr_itr_next_block, mrope.c:114-131) */
typedef void (*rb2_hip_run_cb)(void *user, const uint8_t *runs, int64_t n_bytes);
int64_t rb2_hip_stream_rope(rb2_hip_t *h, int b, rb2_hip_run_cb cb, void *user);

/* replace all six ropes by the symbols described by six 43+3 run-length streams (any run
 * width; rle[b] may be NULL when n_bytes[b]==0).  Used to seed the device from an .fmr file /
 * host ropes (mr_restore, mrope.c:145-160; rope_restore, rope.c:308-318). */
void rb2_hip_load_ropes(rb2_hip_t *h, const uint8_t *const rle[6], const int64_t n_bytes[6]);

/* optional capacity hint (like the reference sizing its buffers from -m, main.c:136): make room for batches
 * of up to batch_bytes bytes / batch_strings strings and an index of total_symbols symbols, so that
 * rb2_hip_insert_multi* never has to grow (reallocate + copy) a buffer.  Any argument may be 0. */
void rb2_hip_reserve(rb2_hip_t *h, int64_t batch_bytes, int64_t batch_strings, int64_t total_symbols);

/* rank of all six symbols in [0,x) of rope b, computed on the device (rope_rank1a, rope.h:45) */
void rb2_hip_rank1a(rb2_hip_t *h, int b, int64_t x, int64_t cx[6]);
/* the same for n positions at once: out[i*6 + a] = number of a's in [0, x[i]) of rope b.  One wave per query (a coalesced
 * 512-byte leaf load, bit-plane popcounts per lane, DPP reduction): the query-side counterpart of rope_rank1a for workloads
 * that ask millions of ranks (x, out: host memory). */
void rb2_hip_rank_batch(rb2_hip_t *h, int b, int64_t n, const int64_t *x, int64_t *out);

/* 64-bit checksum of rope b computed on the device (position-weighted sum over the packed words of its pieces): equal for
 * equal symbol sequences however the index was built (one engine, N ranks, an .fmr loaded back), sensitive to order.  Lets
 * tests compare indexes of 10^11 symbols without moving them off the device (the reference has no counterpart; its ropes are
 * compared through their .fmd) */
uint64_t rb2_hip_rope_hash(rb2_hip_t *h, int b);

/* ---- rope sharding across GPUs ---------------------------------------------------------------
 * The unit of ownership is a SUB-ROPE: rope b is kept as six independent pieces (b,x), x = the symbol that follows b in the
 * row's suffix (piece (b,x) holds exactly the b-symbols of rope x; rope $ is one piece; NR = 31 pieces, see rb2_device.h).
 * owner[r] = rank that holds piece r and processes its bucket.  Inside a round the pieces are independent (the reference runs
 * the ropes on separate threads, mrope.c:312-329); between rounds every rank needs the NR x 6 count matrix (the master reads
 * r[b]->c[] of all ropes, mrope.c:332-340) and strings move from piece (b,x) to the owner of piece (a,b) for the symbol a they
 * just inserted (mrope.c:303-309).  Both exchanges happen inside the library: rb2_hip_multi_* below.  (Rounds 2-3 also exported
 * the phases of a round one by one, rb2_hip_shard_*, for a Python driver that issued the collectives itself; that second
 * protocol was retired in round 4.) */
int     rb2_hip_num_subropes(void);                    /* NR = 31: rope $ + pieces (b,x), index 1+(b-1)*6+x */
void    rb2_hip_memcpy(rb2_hip_t *h, void *dst, const void *src, int64_t bytes, int kind);   /* on the handle's stream; kind 0 host->device, 1 device->host, 2 device->device */
/* run on the caller's stream (e.g. torch.cuda.current_stream().cuda_stream): the caller's own work on that stream and the
 * engine's kernels then need no host synchronisation between them */
void    rb2_hip_use_stream(rb2_hip_t *h, void *hip_stream);


/* ---- N GPUs behind one handle (round 3) ------------------------------------------------------------------
 * The reference fans a round out to its workers and joins them INSIDE mr_insert_multi (mrope.c:287-296, 312-329) and reads
 * every rope's counts after the barrier (mrope.c:332-340); callers just call mr_insert_multi (main.c:240, 248).  This is the
 * same contract for N engines: one call inserts a batch into ONE index whose 31 sub-ropes are dealt out over the ranks
 * (owner map: owner[r] = rank of sub-rope r), the round loop -- count matrix, merge, exchange of the string records, unpack --
 * runs inside the library, one host thread per local rank, no host <-> device synchronisation between the rounds of a
 * batch on the PEER transport.  Two transports behind the same loop:
 *   RB2_TRANSPORT_PEER  one process, every rank an engine of its own on devices[i] (the same device may be listed several
 *                       times: N "virtual" ranks on one GPU -- how the whole path is tested on a one-GPU box).  The count
 *                       matrices are summed by a kernel that reads the peers' rows, the string records are fetched by the
 *                       RECEIVER's unpack kernel straight from the senders' buffers (peer access over xGMI; device events
 *                       order the streams), so an exchange costs no launch of its own and no host round trip.
 *   RB2_TRANSPORT_RCCL  RCCL's C API (librccl, loaded on first use): ncclAllReduce of the 31 x 6 matrix in place, grouped
 *                       ncclSend / ncclRecv of the records.  Works across processes (rb2_hip_multi_create_rank: one process
 *                       per GPU, the launch contract of bench.py) and inside one (rb2_hip_multi_create on distinct devices).
 *                       The host reads the reduced matrix from pinned memory behind an event that fires BEFORE the merge
 *                       kernels of the round run -- it sizes the sends while the GPU merges; the stream never drains.
 */
typedef struct rb2_hip_multi_s rb2_hip_multi_t;
#define RB2_TRANSPORT_PEER 0
#define RB2_TRANSPORT_RCCL 1
#define RB2_MULTI_MAX_RANKS 64
/* one process drives n ranks; devices[i] = HIP device of rank i; owner == NULL: the default owner map (rb2_hip_default_owners) */
/* Start-up checks (round 4), so that the first run on real multi-GPU hardware fails loudly or not at all:
 *   - PEER transport asked for, but some pair of the listed devices has no peer access: the handle is created on the RCCL
 *     transport instead (message on stderr; never a CPU path).  Impossible (a device listed twice AND a pair without peer access): fatal.
 *   - self-test: whenever the ranks sit on more than one physical device (or RB2_MULTI_SELFTEST=1; =0 turns it off) a job of
 *     2000 short reads is built across the ranks and, sub-rope by sub-rope, compared -- device-side checksums and the count matrix
 *     -- with the same job on one engine; a mismatch is fatal.  The handle is reset afterwards.  Under rb2_hip_multi_create_rank (one
 *     process per GPU) the self-test is a COLLECTIVE: every process of the group must take the same decision, i.e. RB2_MULTI_SELFTEST
 *     must be set to the same value (or left unset) in all of them -- a rank that skips it alone leaves the others waiting in RCCL. */
rb2_hip_multi_t *rb2_hip_multi_create(int n, const int *devices, int sorting_order, int transport, const int *owner /* [NR] or NULL */);
/* the transport the handle really uses (RB2_TRANSPORT_*) */
int rb2_hip_multi_transport(const rb2_hip_multi_t *m);
/* one rank of a group of `nranks` processes (RCCL): `nccl_id` = the 128 bytes rb2_hip_multi_unique_id() produced on rank 0 */
void rb2_hip_multi_unique_id(void *id128);
rb2_hip_multi_t *rb2_hip_multi_create_rank(int device, int rank, int nranks, const void *nccl_id, int sorting_order, const int *owner);
void rb2_hip_multi_destroy(rb2_hip_multi_t *m);
void rb2_hip_default_owners(int nranks, int owner[] /* NR */);
int  rb2_hip_multi_nranks(const rb2_hip_multi_t *m);            /* ranks of the whole group */
int  rb2_hip_multi_nlocal(const rb2_hip_multi_t *m);            /* ... driven by this process */
rb2_hip_t *rb2_hip_multi_engine(rb2_hip_multi_t *m, int local_rank);   /* the engine of a local rank (profiling, rb2_hip_dev_alloc, ...) */
/* mr_insert_multi (mrope.c:258) on the sharded index: `s` host memory, borrowed for the call (uploaded once per device) */
void rb2_hip_multi_insert_multi(rb2_hip_multi_t *m, int64_t len, const uint8_t *s);
/* the batch already sits on the devices: s_dev[i] = the (whole) batch text in the memory of local rank i's device, 16-byte
 * aligned; ranks on one device may share a buffer */
void rb2_hip_multi_insert_multi_dev(rb2_hip_multi_t *m, int64_t len, const uint8_t *const *s_dev);
void rb2_hip_multi_get_counts(rb2_hip_multi_t *m, int64_t c[36]);
/* rope b as run bytes: its pieces (b,x), x = $ACGTN, each from its owner, in order.  A multi-process handle only holds the
 * pieces of its own rank: the functions then return those (in order) -- gathering across processes is the caller's job */
int64_t rb2_hip_multi_rope_bytes(rb2_hip_multi_t *m, int b);
int64_t rb2_hip_multi_download_rope(rb2_hip_multi_t *m, int b, uint8_t *dst);
int64_t rb2_hip_multi_stream_rope(rb2_hip_multi_t *m, int b, rb2_hip_run_cb cb, void *user);
void rb2_hip_multi_load_ropes(rb2_hip_multi_t *m, const uint8_t *const rle[6], const int64_t n_bytes[6]);
void rb2_hip_multi_reserve(rb2_hip_multi_t *m, int64_t batch_bytes, int64_t batch_strings, int64_t total_symbols);
void rb2_hip_multi_reset(rb2_hip_multi_t *m);
void rb2_hip_multi_sync(rb2_hip_multi_t *m);
void rb2_hip_multi_rank1a(rb2_hip_multi_t *m, int b, int64_t x, int64_t cx[6]);
/* out[0] host <-> device synchronisations inside the round loops so far (PEER: none in dense rounds; an in-place round reads a
 * one-word verdict before its exchange; RCCL: one event wait per round, behind the reduce only), out[1] rounds, out[2] batches,
 * out[3] in-place (sparse) rounds summed over the ranks, out[4] void sparse rounds, out[5] re-layouts */
void rb2_hip_multi_stats(rb2_hip_multi_t *m, int64_t out[6]);
/* bytes of device memory local rank k holds for the text of the last batch given to rb2_hip_multi_insert_multi (host buffer): with
 * ranks on several devices of one process (PEER) the text is ONE copy, shared piece by piece between the devices' memories and
 * mapped for all of them -- about len / n per rank, each rank uploads its own share --, otherwise a copy of the whole batch per
 * device, held by the first rank on it (0 for the others); -1: no such rank.  RB2_MULTI_TEXT=shard / =copy forces either. */
int64_t rb2_hip_multi_text_bytes(const rb2_hip_multi_t *m, int k);
uint64_t rb2_hip_multi_rope_hash(rb2_hip_multi_t *m, int b);
/* the device-side exchange plan (k_mround) evaluated on the host, for tests: see csrc/rb2_multi.h */
int rb2_hip_multi_plan_host(const int *owner /* [NR] */, int nranks, const int64_t *g /* [NR*6] */, int me, int64_t *sdest /* [NR*6] */, int64_t (*pieces)[5] /* [NR*6] */, int64_t *total);   /* == rb2_hip_rope_hash of the same rope on one engine */

/* ---- measurement helpers (bench.py; not part of the reference API) ------------------------ */

/* allocate / free raw device memory */
void *rb2_hip_dev_alloc(rb2_hip_t *h, int64_t bytes);
void  rb2_hip_dev_free(rb2_hip_t *h, void *p);

/* fill dst_dev (n_reads*(read_len+1) bytes) with synthetic reads first_read..first_read+n_reads
 * of the SURVEY.md 8c generator (splitmix64 counter stream), already nt6-encoded, reversed and
 * 0-terminated -- i.e. exactly what main.c:177-237 would put in the batch buffer for `-R`.
 * strand: 0 = forward strand only; 1 = forward followed by reverse complement (buffer doubles). */
void rb2_hip_synth_reads(rb2_hip_t *h, uint8_t *dst_dev, int64_t first_read, int64_t n_reads,
                         int read_len, uint64_t seed, int strand);
/* the same, but the reads are windows of one random genome of genome_len bases (uniform start positions): overlapping
 * reads as from sequencing at coverage n_reads*read_len/genome_len -- large groups, non-empty intervals in every round */
void rb2_hip_synth_reads_cov(rb2_hip_t *h, uint8_t *dst_dev, int64_t first_read, int64_t n_reads,
                             int read_len, uint64_t seed, int strand, int64_t genome_len);
/* ... with a composition knob: skew != 0 draws 85 % A and 5 % each of C, G, T from the same random words (tools/synth_reads.c,
 * seventh argument): sub-rope (A,A) then holds 72 % of the index and passes 2^32 symbols at 59 M x 101 bp */
void rb2_hip_synth_reads_skew(rb2_hip_t *h, uint8_t *dst_dev, int64_t first_read, int64_t n_reads,
                              int read_len, uint64_t seed, int strand, int64_t genome_len, int skew);

void rb2_hip_sync(rb2_hip_t *h);

/* leaf-layout statistics since rb2_hip_create: out[0] re-layouts (dense <-> sparse), out[1] void sparse rounds (a leaf ran out
 * of slack; the round was redone densely), out[2] rounds inserted in place, out[3] 1 when the index currently has the sparse layout */
void rb2_hip_sparse_stats(rb2_hip_t *h, int64_t out[4]);
/* the same four, then out[4] re-spreads among the re-layouts (sparse -> sparse: a superblock had no free slot for a split),
 * out[5] leaves split in place by k_split (the leaf split of rope.c:143-146); out[6] device buffers that had to grow while the rounds
 * of a dense batch were being queued (each one is a device-wide wait in the middle of the batch: 0 unless a sizing rule is missing);
 * out[7] reserved */
void rb2_hip_layout_stats(rb2_hip_t *h, int64_t out[8]);
/* window formats of the dense layout (a window = 4 leaves = 4096 symbols; csrc/rb2_merge.h): out[0..3] = windows the dense merge wrote
 * plain (three bit planes) / compact with no, one, two lines of exception positions -- counted on the device only when the handle was
 * created with RB2_COMPACT_STATS=1 in the environment (zeros otherwise) --, out[4] = dense rounds that were allowed to write compact
 * windows (all intervals empty, not the last round of a batch, no re-layout ahead), out[5] = 1 when the per-format counts are on.
 * RB2_COMPACT=0 keeps every window plain.  The reference has no counterpart: its leaves are always run-length coded (rle.h:39-75). */
void rb2_hip_window_stats(rb2_hip_t *h, int64_t out[6]);

/* per-kernel timing, measured with hipEvents on the engine's own stream when enabled */
#define RB2_K_SYM      0
#define RB2_K_TSCAN    1
#define RB2_K_PREP     2
#define RB2_K_PART     3
#define RB2_K_MERGE    4
#define RB2_K_META     5
#define RB2_K_ADVANCE  6
#define RB2_K_INIT     7
#define RB2_K_RELAYOUT 8   /* change between the dense and the sparse (slack) leaf layout */
#define RB2_K_SPLIT    9   /* leaf splits at the end of an in-place round */
#define RB2_K_COUNT    10
void rb2_hip_profile(rb2_hip_t *h, int enable);   /* 0: off; 1: every kernel group of a round (sixteen events per round); 2: the merge launches only (two) */
/* launches[k], ms[k] (summed), units[k] (strings processed, summed) since the last reset */
void rb2_hip_profile_get(rb2_hip_t *h, int64_t launches[RB2_K_COUNT], double ms[RB2_K_COUNT],
                         int64_t units[RB2_K_COUNT], int reset);
const char *rb2_hip_kernel_name(int k);

/* device layout constants: symbols per leaf, leaves per merge tile, strings per string tile */
void rb2_hip_layout(int *leaf_syms, int *tile_leaves, int *string_tile);

#ifdef __cplusplus
}
#endif
#endif
