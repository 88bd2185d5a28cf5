// rb2_device.h -- device data layout + wave/block primitives for the gfx950 BWT insertion engine.
//
// Layout in HBM (all sizes are compile-time constants below):
//
//   sub-rope      rope b is kept as six independent pieces (b,x), x = the symbol following b in the row's suffix; NR = 31 pieces
//                 (see below).  A piece is a flat array of symbols; pieces lie back to back in the pool, each on a superblock boundary.
//   group         GSYM = 64 consecutive symbols as three 64-bit BIT PLANES: bit i of plane k = bit k of symbol i.  Counting symbols is
//                 popcounts of dense words, a symbol compare is two ANDs of XORed planes, every position -> (word, bit) is a shift.
//   leaf          LEAFG = 16 groups = LEAF = 1024 symbols = LEAFW = 48 words = LEAFB = 384 bytes = three 128-byte lines, PLANE-MAJOR:
//                 word pl * 16 + g holds plane pl of group g.  A leaf is one DPP row of 16 lanes (lane = group, the planes in registers).
//                 Dense layout: every leaf of a piece holds exactly LEAF symbols except the last, "which leaf holds position p" is
//                 p >> 10 -- no B+ tree descent (the reference walks rpnode_t buckets, rope.c:119-134).
//   window        WPL = 4 consecutive leaves = 4096 symbols = twelve lines: the unit of the dense merge (one wave, lane = group) and of the
//                 COMPACT format (rb2_merge.h "window formats"): between two rewrites that only the merge reads, a window keeps planes 0, 1
//                 and, in place of its four plane-2 lines, up to two lines of exception positions ($ / N) -- or nothing at all.
//                 The format of a window = LeafMeta::npre of its first leaf in own[] (0 = plain, what everything but k_merge expects).
//   own[]         16 B per leaf (LeafMeta): its six own counts + fill, written by whoever writes the leaf (k_merge, k_relayout, the loader).
//   meta[]        dense layout: 16 B per leaf, the counts of the preceding leaves of the same superblock (k_meta_sb, from own[]);
//                 sparse layout: the same 512 bytes per superblock are eight rows of 32 u16 -- row 0 fills (+ bit 15: the leaf's third plane line is valid, see "two-plane leaves"), rows 1-6 own counts, row 7 the
//                 claim word of k_split -- that an in-place insert updates by atomics on its own entries (dir_row / dir_commit).
//   sparse layout the same arrays, but leaves carry SLACK (SP_FILL symbols after a re-layout, SP_USED of a superblock's 32 slots in use):
//                 rounds that touch few leaves insert IN PLACE (k_merge_leaf), position -> leaf is a search (locate()), a leaf that fills
//                 up is split into a free slot of its superblock (k_split: split_node, rope.c:78-112).
//   superblock    SB = 32 leaf slots.  Pool-wide prefix in two levels: SbRec (32 B per superblock: 32-bit prefixes of the six counts and of
//                 the position inside its chunk of SCHUNK = 1024 superblocks) + SbBase (64-bit, per chunk); SbTot: the superblock's own
//                 totals, six 16-bit fields.  rank(a, p) = sbbase + sbrec + meta prefix + in-leaf popcounts (rope_rank2a, rope.c:179-194).
//   pool          two sides (ping-pong in dense rounds, one side in place in sparse ones); arrays grow behind reserved address ranges.
//   strings       SoA per-string state (reference triple64_t, mrope.c:174-178), ping-pong: L, U (the interval in the reference's own
//                 coordinates, piece-relative, 32-bit storage while no piece can hold 2^32 symbols), W (the next CUR_SYMS = 9 symbols,
//                 3 bits each, + the text position of the symbols behind them), A (this round's symbol + group-head / non-empty flags).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rb2 {

constexpr int SBITS  = 3;              // bit planes per symbol ($ACGTN = 0..5)
constexpr int GSYM   = 64;             // symbols per group (one 64-bit word per plane)
constexpr int LEAFG  = 16;             // groups per leaf: one per lane of a DPP row
constexpr int LEAFW  = SBITS * LEAFG;  // 48 words per leaf, plane-major: word pl * LEAFG + g
constexpr int LEAF   = GSYM * LEAFG;   // 1024 symbols per leaf
constexpr int LEAF_SH = 10;            // log2(LEAF)
constexpr int SB     = 32;            // leaves per superblock
constexpr int LEAFB  = LEAFW * 8;      // 384 bytes per leaf
#ifndef RB2_GPL
#define RB2_GPL 1
#endif
constexpr int GPL    = RB2_GPL;        // groups per lane in k_merge: one wave rewrites a window of 64 * GPL groups
constexpr int WPL    = 64 * GPL / LEAFG;   // leaves per window
constexpr int WIN    = WPL * LEAF;     // symbols per window
constexpr int STILE  = 512;           // strings per string tile
constexpr int MW     = 4;             // waves per block in the one-wave-per-leaf / per-window kernels
constexpr int SCHUNK = 1024;          // items per block in the 3-kernel scans
constexpr int ZBLOCK = 16384;         // bytes per block when locating sentinels

// Sub-ropes.  Rope b (the rows that start with symbol b) is kept as six independent pieces (b,x),
// x = the symbol that FOLLOWS b in the row's suffix; piece (b,x) holds exactly the b-symbols of rope x,
// in order (LF-mapping), so the pieces of rope b in the order x = $,A,C,G,T,N concatenate to rope b.
// A string that sits in (b,x) and inserts a moves to (a,b).  Rope $ is one piece.  Pieces never
// interact inside a round, which is what lets more than four GPUs share the work.
constexpr int NR = 31;
__host__ __device__ inline int rope_sym(int r)  { return r == 0 ? 0 : 1 + (r - 1) / 6; }   // b
__host__ __device__ inline int rope_prev(int r) { return r == 0 ? 0 : (r - 1) % 6; }       // x
__host__ __device__ inline int rope_of(int a, int b) { return a == 0 ? 0 : 1 + (a - 1) * 6 + b; }

// The per-string positions (L, U, INS_E, SIZE) are piece-relative: while no sub-rope holds 2^32 symbols they are STORED as 32-bit values
// (template parameter P of the string kernels: uint32_t or uint64_t) -- 24 bytes less per string and round through HBM than with
// 64-bit storage (configs[1]: +5.6 % on one box).  The engine starts a batch in the narrow mode when it can and widens the arrays the
// round before a piece could reach 2^32 symbols (k_setup reports the largest piece to pinned memory; rb2_engine.hip maybe_widen).

struct LeafMeta { uint16_t c[6]; uint16_t npre; uint16_t n; };   // meta[]: prefixes inside the superblock + own fill; own[]: own counts + own fill
#ifndef RB2_SP_FILL
#define RB2_SP_FILL 768
#endif
#ifndef RB2_SP_USED
#define RB2_SP_USED 24
#endif
#ifndef RB2_SP_MARGIN
#define RB2_SP_MARGIN 64
#endif
constexpr int SP_FILL = RB2_SP_FILL;           // sparse layout: symbols per leaf after a re-layout (75 % of LEAF: room for 256 inserts)
constexpr int SP_USED = RB2_SP_USED;            // ... leaf slots in use per superblock; the other 8 are the superblock's own reserve: a leaf that comes within
                                       // SP_MARGIN symbols of LEAF is split into one of them at the end of the round (k_split: the counterpart of the
                                       // reference's leaf split, rope.c:143-146 / split_node rope.c:78-112, one level of its B+ tree)
constexpr int SP_MARGIN = RB2_SP_MARGIN;          // a leaf is split when its fill exceeds LEAF - SP_MARGIN: whatever a round brings, a leaf takes 64 more symbols
struct Cnt6 { uint64_t v[6]; };
struct SbTot { uint32_t p01, p23, p45, pad; };             // symbol counts of one superblock, six 16-bit fields (<= SB * LEAF each)

struct RopeDesc {
	uint64_t n;         // symbols
	uint64_t leaf0;     // first leaf (multiple of SB)
	uint64_t nleaves;   // ceil(n / LEAF)
	uint64_t sb0;       // leaf0 / SB
	uint64_t cnt[6];    // marginal counts (rope_t.c, rope.h:19)
};

struct SegDesc {        // where the strings of bucket r (= sub-rope r) live in the current SoA arrays
	uint64_t start[NR], cnt[NR];
	uint32_t tile0[NR + 3];  // first string tile of each segment; [NR] = total
};

struct Ctl {
	RopeDesc rope[2][NR];
	SegDesc  seg[2];
	uint64_t wf0[NR + 3];   // first output window (WPL consecutive leaves, one merge wave) per sub-rope this round; [NR] = total
	uint64_t ac[NR][6];     // ac[r][a] = #a in the pieces of the same rope in front of piece r, after this round (mrope.c:332-336)
	uint64_t dest[NR][6];   // where members of bucket r inserting a go in the next arrays
	uint64_t count[NR][6];  // count[r][a] = members of bucket r inserting a this round
	uint64_t nsb_total;     // superblocks in use on the new side
	uint64_t n_strings;     // strings in this batch
	uint64_t max_len;       // longest string (without sentinel)
	uint64_t n0;            // strings already in the index (#'$' in the BWT, mrope.c:279)
	uint64_t len;           // batch bytes
	// ne[p] != 0: some string has a NON-EMPTY interval in the arrays read by rounds of parity p.  Zero (always in input
	// order; on random reads from round ~14 of a batch on) selects the kernel variants that never touch U / SIZE.
	uint32_t ne[2];
	// ---- sparse (in-place) rounds
	uint32_t wstride;       // work orders of a sparse round: WLC lists of at most wstride entries each, list c at LD[c * wstride ..) (see wcnt)
	uint32_t overflow;      // some touched leaf cannot take its inserts: the round is void (every later kernel returns) and the host redoes it densely
	uint32_t sbfull;        // (same 8-byte verdict word) k_split found a leaf to split in a superblock without a free slot: the host re-spreads the index before the next round
	uint32_t nsplit2[2];    // [round & 1]: leaves that came within SP_MARGIN of LEAF in that round (k_part_sparse appends them, the splits -- queued with the NEXT round's counting
	                        // phase, beside its k_setup, which clears the counter of ITS round -- read them: two counters, no race)
	uint64_t nsplit_total;  // leaves split since the handle was created (statistics)
	RopeDesc relay_old[NR]; // k_relayout: the layout being read while rope[side] already describes the one being written
	// ---- rope sharding across GPUs (single GPU: own[] all 1, sdest unused)
	uint32_t own[NR + 1];   // own[r] != 0: this rank holds sub-rope r and processes bucket r
	uint64_t sdest[NR][6];  // sharded mode: record offset in the send buffer for members of bucket r inserting a (RCCL transport) ...
	uint64_t pdst[NR][6];   // ... or where they start in the NEXT arrays of the rank that owns piece (a, b) (PEER transport: k_advance writes them there itself)
	uint32_t pdev[NR][6];   // ... and which rank that is
	// The work list of a sparse round (touched leaves, appended by k_part_sparse, read by k_merge_leaf) is WLC lists, the c-th 64th of the
	// string tiles appends to list c: ONE counter took a returning atomic from every string tile, and atomics on one address are served
	// one after the other (~12 ns each on MI355X: 25 us for 2048 tiles, tools/ubench/hot_atomic.hip; counters inside one 128-byte line
	// are no better than one counter); 16 counters on lines of their own cost < 1 us.  Consecutive tiles share a list; a wave of
	// k_merge_leaf works on one list.  wcnt[c * WLS] = entries of list c.
	uint32_t wcnt[16 * 32];
	unsigned long long wfmt[4]; // windows k_merge wrote in each format (rb2_merge.h), counted only when asked for (RB2_COMPACT_STATS=1: one atomic per window)
};
constexpr int GCN = NR * 6 + 2;         // words of the per-round count matrix buffers: NR x 6 counts + [NR * 6] = "some string of this rank has a non-empty
                                        // interval this round" (summed over the ranks of a sharded index like the counts: zero = every rank may launch the
                                        // all-empty kernel variants only, from this round on) + one word of padding
constexpr int WLC = 16;                 // work lists of a sparse round
constexpr int WLS = 32;                 // their counters sit 128 bytes apart

// one string's state on the wire (24 B): a = l (48 bits) | size[15:0] << 48;  b = id | size[47:16] << 32;  w = the symbol cursor.
// (Rounds 1-2 sent 16 bytes and rebuilt the cursor on arrival from the batch text every rank holds: a 20-byte random gather per
// string and round on the receiver, ~100 B of HBM traffic to save 8 B on the wire -- and pure loss between ranks of one device.)
struct ShardRec { uint64_t a, b, w; };
// PEER transport: the next-round string arrays and the control block of every rank of the handle, as this device sees them (its own
// memory or a peer mapping): k_advance stores a string that changes owner straight into the owner's arrays (posted writes over xGMI)
// instead of leaving a record for the owner to fetch.  64 = RB2_MULTI_MAX_RANKS (include/rb2_hip.h).
struct PushTab { uint64_t *L2[64], *U2[64], *W2[64]; uint8_t *A2[64]; struct Ctl *ctl[64]; };
__host__ __device__ inline ShardRec shard_pack(uint64_t l, uint64_t size, uint32_t id, uint64_t w)
{
	ShardRec r; r.a = (l & 0xffffffffffffull) | (size & 0xffffull) << 48; r.b = (uint64_t)id | (size >> 16) << 32; r.w = w; return r;
}

struct LeafDesc {           // work order of one output window (WPL leaves), written by k_part, read by k_merge (32 B)
	uint64_t i0;            // position (in the old sub-rope) of the first old symbol the window consumes = j*WIN - q0
	uint64_t ins0;          // index of its first new symbol in INS_E / INS_A / RKREL
	uint64_t gl;            // first leaf slot on the new pool side
	uint32_t oleaf0;        // first leaf slot of the sub-rope on the old pool side
	uint16_t ni, nvalid;    // new symbols / symbols in the window (low 14 bits); bits 14-15: the formats of the first / second old window it draws from (rb2_merge.h)
};

struct SpOrd {              // work order of one touched leaf in a sparse round, written by k_part_sparse, read by k_merge_leaf (16 B, in the LD buffer)
	uint32_t gl;            // leaf slot (32 bits in the sparse layout)
	uint32_t ins0;          // index of its first new symbol in INS_E / INS_A / RKREL (a batch has < 2^32 strings)
	uint32_t i0;            // piece position of the leaf's first symbol, low half (positions inside a leaf need no more); ni == 1: the insert itself -- its place inside the leaf (bits 0-11) and its symbol (12-14)
	uint16_t ni, nvalid;    // new symbols / symbols in the leaf after the round (bits 0-10); bit 15 of nvalid: the leaf has a plane-2 line (FILL_P2)
};

// The pool-wide prefix over the superblocks, two levels: what lies in front of superblock sb = base of its chunk of SCHUNK superblocks
// (64-bit, a few thousand records: cache-resident) + the prefix inside the chunk (32-bit: a chunk holds < 2^25 symbols).  One 32-byte
// record per superblock instead of the 48 + 8 bytes of 64-bit values of rounds 1-3: the scan that rebuilds them every round -- the one part
// of an in-place round that reads every superblock -- moves 48 bytes per superblock instead of 72, in whole lines.
constexpr int SCHUNK_SH = 10;          // log2(SCHUNK)
struct SbRec { uint32_t cum[6]; uint32_t pos; uint32_t tot; };     // inside the chunk: symbol counts / symbols in front of the superblock; tot: the symbols it holds itself (its second
                                                                   // 16 bytes -- cum[4], cum[5], pos, tot -- are all one probe of an in-place round's descent needs: k_part_sparse)
struct SbBase { uint64_t cum[6]; uint64_t pos; uint64_t pad; };    // in front of the chunk
struct PoolView { uint8_t *data; LeafMeta *meta; SbRec *sbrec; LeafMeta *own; SbBase *sbbase; uint8_t *xh; };   // xh: one byte per window (4 leaf slots), the format k_merge wrote it in (rb2_merge.h)
__device__ __forceinline__ uint64_t sb_pos(const PoolView &pv, uint64_t sb) { return pv.sbbase[sb >> SCHUNK_SH].pos + pv.sbrec[sb].pos; }   // symbols in front of superblock sb (pool-wide)
__device__ __forceinline__ uint64_t sb_cum(const PoolView &pv, uint64_t sb, int a) { return pv.sbbase[sb >> SCHUNK_SH].cum[a] + pv.sbrec[sb].cum[a]; }

struct TileRec {            // per string tile, written by k_sym (80 bytes of it: the buffer is sized in these; layout: TileRecs below)
	uint32_t hist[6];
	uint32_t lhpre[6];      // symbol counts in the tile before its last group head
	uint32_t fhpre[6];      // ... before its first group head
	int32_t  lh, fh;        // in-tile index of last / first head, -1 when the tile has none
};

// ... as the kernels see it: a structure of arrays (20 arrays of `cap` words, cap % 4 == 0) -- a scan reads one column of consecutive
// tiles with coalesced (16-byte) loads; records of 80 bytes made every lane of the single-block scan touch lines of its own
struct TileRecs {
	uint32_t *p; uint32_t cap;
	__device__ __forceinline__ uint32_t &hist(int s, uint32_t t) const { return p[(uint64_t)s * cap + t]; }
	__device__ __forceinline__ uint32_t &lhpre(int s, uint32_t t) const { return p[(uint64_t)(6 + s) * cap + t]; }
	__device__ __forceinline__ uint32_t &fhpre(int s, uint32_t t) const { return p[(uint64_t)(12 + s) * cap + t]; }
	__device__ __forceinline__ int32_t &lh(uint32_t t) const { return ((int32_t*)p)[(uint64_t)18 * cap + t]; }
	__device__ __forceinline__ int32_t &fh(uint32_t t) const { return ((int32_t*)p)[(uint64_t)19 * cap + t]; }
};

struct TileScan {           // per string tile (+1), written by the tile scan
	uint32_t pre[6];        // exclusive prefix of hist over ALL tiles (subtract the segment's first)
	int32_t  lht;           // last tile before this one that contains a head (-1 if none)
	int32_t  nht;           // first tile after this one that contains a head (INT_MAX if none)
};

struct TileFix {            // per string tile, written by k_tfix: what group_setup needs, relative to the tile's segment
	uint32_t tpre[6];       // members of the segment in front of this tile that insert s
	uint32_t popen[6];      // the same count in front of the group that is open at the start of the tile
	uint32_t pnext[6];      // ... in front of the first group head after the tile (segment total if none)
	uint32_t fopen;         // first member (segment-relative index) of the group open at the start of the tile
	uint32_t b, lt, nexthead;   // bucket (sub-rope) of the tile, its number inside the bucket; nexthead bit 0: the string behind the tile starts a group (or the bucket ends there), bit 1: k_sym did k_prep's work for this tile
	uint64_t segstart, segend;   // the bucket's range in the string arrays: k_prep / k_advance get their tile context
	                             // from this one record instead of chasing tile0[] -> start[] / cnt[]
};
constexpr int TILEFIX_LDS_WORDS = 22;   // tpre, popen, pnext, fopen, b, lt, nexthead: what group_setup keeps in LDS

struct ChunkPart { uint32_t sum[6]; int32_t mx, mn; };

// ---------------------------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes on gfx950)
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Block b of a launch is observed to run on XCD b % 8 (MI355X_MICROARCH.md), and every XCD has an L2 of its own.  Neighbouring work items
// of the tile and window kernels share 128-byte lines -- the partial lines at the ends of the chunks a string tile scatters, the old
// lines two output windows both draw from -- and with blocks dealt round-robin the two halves of such a line meet in two different
// L2s: the line is fetched twice, a partial write is merged at the memory side.  xcd_item() hands every XCD runs of XCD_RUN consecutive
// items instead, the runs dealt round-robin (a permutation of the block numbers when the grid is a multiple of 8 * XCD_RUN -- the host
// rounds its grids up -- else the identity).  Runs, not one contiguous eighth per XCD: work per item is not uniform (round 0 of a batch
// puts every string into the first windows of the pool), and an eighth of the GRID is an eighth of the CHIP.  A speed choice only:
// nothing depends on where a block runs.
#ifndef RB2_XCD
#define RB2_XCD 1
#endif
constexpr uint32_t XCD_RUN = 16;
__device__ __forceinline__ uint32_t xcd_item()
{
	if (!RB2_XCD || (gridDim.x & (8u * XCD_RUN - 1u))) return blockIdx.x;
	const uint32_t x = blockIdx.x & 7u, i = blockIdx.x >> 3;    // my XCD, my number among its blocks
	return (i / XCD_RUN) * (8u * XCD_RUN) + x * XCD_RUN + (i % XCD_RUN);
}

// segment lookup in a monotone table tab[0..NR] for a WAVE-UNIFORM value v: the s with tab[s] <= v < tab[s+1].
// All 64 lanes must call it together.
template <class T> __device__ __forceinline__ int seg_of(const T *tab, uint64_t v)
{
	const int l = threadIdx.x & 63;
	const T e = tab[l < NR ? l + 1 : NR];
	return __popcll(__ballot(l < NR && v >= (uint64_t)e));
}
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

template <typename T> __device__ __forceinline__ T wave_incl_add(T v)
{
	const int l = lane_id();
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { T t = __shfl_up(v, d); if (l >= d) v += t; }
	return v;
}
__device__ __forceinline__ int wave_incl_max(int v)
{
	const int l = lane_id();
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(v, d); if (l >= d) v = max(v, t); }
	return v;
}
__device__ __forceinline__ int wave_incl_min_down(int v)     // suffix minimum
{
	const int l = lane_id();
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { int t = __shfl_down(v, d); if (l + d < 64) v = min(v, t); }
	return v;
}
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
	return v;
}

// exclusive prefix sum over the block; s_w needs blockDim/64 (+0) entries; *total gets the block sum
template <typename T> __device__ __forceinline__ T block_excl_add(T v, T *s_w, T *total)
{
	const int l = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
	T inc = wave_incl_add(v);
	if (l == 63) s_w[w] = inc;
	__syncthreads();
	T off = 0, tot = 0;
	for (int i = 0; i < nw; ++i) { T x = s_w[i]; if (i < w) off += x; tot += x; }
	__syncthreads();
	if (total) *total = tot;
	return off + inc - v;
}
__device__ __forceinline__ int block_excl_max(int v, int *s_w, int ident)    // max over earlier threads
{
	const int l = lane_id(), w = wave_id();
	int inc = wave_incl_max(v);
	if (l == 63) s_w[w] = inc;
	__syncthreads();
	int off = ident;
	for (int i = 0; i < w; ++i) off = max(off, s_w[i]);
	__syncthreads();
	int prev = __shfl_up(inc, 1);
	return l == 0 ? off : max(off, prev);
}
__device__ __forceinline__ int block_excl_min_down(int v, int *s_w, int ident)   // min over later threads
{
	const int l = lane_id(), w = wave_id(), nw = blockDim.x >> 6;
	int inc = wave_incl_min_down(v);
	if (l == 0) s_w[w] = inc;
	__syncthreads();
	int off = ident;
	for (int i = w + 1; i < nw; ++i) off = min(off, s_w[i]);
	__syncthreads();
	int nxt = __shfl_down(inc, 1);
	return l == 63 ? off : min(off, nxt);
}

__device__ __forceinline__ uint64_t lt_mask(int lane) { return lane ? (~0ull >> (64 - lane)) : 0ull; }   // bits < lane

// insertion order of the symbols inside one suffix-array interval (mrope.c:206-224):
// RLO / input order: $ A C G T N;  RCLO: $ T G C A N
__device__ __forceinline__ int sym_ord(int a, int is_comp) { return (is_comp && a >= 1 && a <= 4) ? 5 - a : a; }

// symbol counts from bit planes.  Valid symbols are 0..5 = 000..101, so with the planes b0,b1,b2 (one bit per symbol, already
// masked to the symbols to count): #3 = |b0&b1|, #2 = |b1|-#3, #5 = |b0&b2|, #4 = |b2|-#5, #1 = |b0|-#3-#5.
struct PlAcc { uint32_t p0 = 0, p1 = 0, p2 = 0, p01 = 0, p02 = 0; };
__device__ __forceinline__ void pl_acc(PlAcc &A, uint64_t b0, uint64_t b1, uint64_t b2, uint64_t M /* the symbols to count */)
{
	b0 &= M; b1 &= M; b2 &= M;
	A.p0 += (uint32_t)__popcll(b0); A.p1 += (uint32_t)__popcll(b1); A.p2 += (uint32_t)__popcll(b2);
	A.p01 += (uint32_t)__popcll(b0 & b1); A.p02 += (uint32_t)__popcll(b0 & b2);
}
__device__ __forceinline__ void pl_finish(const PlAcc &A, uint32_t n, uint32_t c[6])
{
	c[3] = A.p01; c[2] = A.p1 - A.p01; c[5] = A.p02; c[4] = A.p2 - A.p02; c[1] = A.p0 - A.p01 - A.p02;
	c[0] = n - (c[1] + c[2] + c[3] + c[4] + c[5]);
}
// bit i set: symbol i of the group equals a
__device__ __forceinline__ uint64_t pl_eq(uint64_t b0, uint64_t b1, uint64_t b2, uint32_t a)
{
	const uint64_t m0 = 0ull - (uint64_t)(~a & 1u), m1 = 0ull - (uint64_t)(~(a >> 1) & 1u), m2 = 0ull - (uint64_t)(~(a >> 2) & 1u);   // all ones where the bit of a is CLEAR
	return (b0 ^ m0) & (b1 ^ m1) & (b2 ^ m2);
}

// the first n positions of a group, n <= 64
__device__ __forceinline__ uint64_t bits_below(uint32_t n) { return n >= 64u ? ~0ull : (1ull << n) - 1ull; }
// word pl * LEAFG + g of leaf slot gl
__device__ __forceinline__ const uint64_t *leaf_words(const uint8_t *data, uint64_t gl) { return (const uint64_t*)data + gl * LEAFW; }
// accumulate the symbols at [from, to) of one leaf, by one thread: the groups the interval touches, three words each
// p2: the leaf has a plane-2 line (always, in the dense layout; sparse layout: Loc::p2)
__device__ inline void leaf_count(const uint64_t *lw, uint32_t from, uint32_t to, PlAcc &A, bool p2 = true)
{
	if (from >= to) return;
	const uint32_t g0 = from >> 6, g1 = (to - 1) >> 6;
	for (uint32_t g = g0; g <= g1; ++g) {
		const uint32_t base = g << 6;
		const uint32_t lo = from > base ? from - base : 0u, hi = min(to - base, 64u);
		const uint64_t b0 = lw[g], b1 = lw[LEAFG + g];
		pl_acc(A, b0, b1, p2 ? lw[2 * LEAFG + g] : ~(b0 | b1), bits_below(hi) & ~bits_below(lo));
	}
}

// ---- sparse layout: the directory of a superblock ---------------------------------------------
// In the sparse layout the 512 bytes that hold the 32 LeafMeta prefixes of a superblock in the dense one (PoolView::meta) are
// eight rows of 32 16-bit values, one per leaf slot: row 0 = fill, rows 1-6 = OWN counts of the six symbols (row 7 unused).
// An in-place insert changes the entries of its own leaf and nothing else -- there is no prefix behind it to move, which is
// what keeps a round's cost proportional to the leaves it touches (the reference updates the counts along one root-to-leaf
// path, rope.c:139-146); a query sums the row in front of its slot, one 48-byte read per symbol.
// TWO-PLANE LEAVES.  With $ACGTN = 000 .. 101 a leaf that holds neither `$` nor `N` -- nine leaves in ten of a long-read index -- is told by
// planes 0 and 1 alone: plane 2 marks T, and T is "neither bit set" (~(p0 | p1) on the valid positions).  Such a leaf's third line is
// neither read nor written by an in-place insert (k_merge_leaf: two lines in, two lines out instead of three and three; the memory side
// moves whole lines and a skipped line costs nothing, tools/ubench/leaf_rw2.hip: 140 -> 103 us per million leaves) and holds NOTHING that
// may be looked at.  Bit 15 of the leaf's fill entry (row 0; a fill is at most LEAF = 2^10) says that the plane-2 line is valid: set by
// whoever writes a leaf that holds a `$` or an `N` (k_meta_sb after a re-layout, k_split for both halves of such a leaf) and by the insert
// that brings the first one (k_merge_leaf adds the bit with the fill, in the same atomic).  Every reader of sparse-layout leaf words goes
// through leaf_p2() below; the dense layout always has three planes (and, between two merges, its own compact windows: rb2_merge.h).
// What rle_insert_cached gains by run-length coding -- fewer bytes moved per insert than a fixed-width field costs (rle.c:63-86) -- is
// gained here by not moving the plane that carries no information.
constexpr uint32_t FILL_MASK = 0x7ffu, FILL_P2 = 0x8000u;
__device__ __forceinline__ uint64_t leaf_p2(bool p2, uint64_t b0, uint64_t b1, uint64_t w2) { return p2 ? w2 : ~(b0 | b1); }   // (positions behind the fill: masked by whoever counts)
constexpr int DIRW = 8 * SB;            // 16-bit values per superblock
__device__ __forceinline__ uint16_t *dir_row(const PoolView &pv, uint64_t sb, int row) { return (uint16_t*)pv.meta + sb * DIRW + row * SB; }
__device__ __forceinline__ uint32_t dir_prefix(const PoolView &pv, uint64_t sb, int row, uint32_t k)   // sum of slots [0, k) of a row, k <= SB
{
	const uint4 *q = (const uint4*)dir_row(pv, sb, row);
	uint32_t acc = 0;                                          // two 16-bit sums side by side (<= 16 * LEAF each)
	auto add8 = [&](int i) {
		const uint4 v = q[i];
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const uint32_t s0 = (uint32_t)(8 * i + 2 * j);      // the word holds slots s0, s0 + 1
			acc += w[j] & (k > s0 + 1 ? 0xffffffffu : (k == s0 + 1 ? 0xffffu : 0u));
		}
	};
#pragma unroll
	for (int i = 0; i <= SP_USED / 8; ++i) if (k > (uint32_t)(8 * i)) add8(i);   // only the 16-byte pieces in front of slot k (all in one 64-byte line)
	return (acc & 0xffffu) + (acc >> 16);
}

// position -> leaf when leaves carry slack (sparse layout; also valid on the dense one): the last leaf of the piece that
// starts at or before p -- superblock by binary search over sbpos, leaf by binary search over the in-superblock prefixes.
// A position on a leaf boundary goes to the RIGHT leaf, p == n to the last leaf in use (the reference sends boundaries to
// the left child, rope.c:130; the BWT does not depend on it).  This is the descent of rope.c:119-134.
struct Loc { uint64_t gl, s; uint32_t n, p2; };   // leaf slot (pool-wide), piece position of its first symbol, its fill, "its plane-2 line is valid" (two-plane leaves)
__device__ inline Loc locate(const PoolView &pv, const RopeDesc &rp, uint64_t p)
{
	Loc r; r.gl = rp.leaf0; r.s = 0; r.n = 0; r.p2 = 0;        // (an empty slot has no plane-2 line either: the insert that brings a `$` / `N` says so)
	if (rp.nleaves == 0) return r;
	const uint64_t nsb = (rp.nleaves + SB - 1) / SB, base = sb_pos(pv, rp.sb0);
	// superblocks hold about the same number of symbols each (a re-layout fills them evenly, inserts land at random), so start
	// from the interpolated guess and gallop: a handful of probes instead of log2(nsb); still O(log distance) on skewed pieces
	uint64_t lo, hi;
	{
		uint64_t g = rp.n ? (uint64_t)((double)p / (double)rp.n * (double)nsb) : 0;
		if (g >= nsb) g = nsb - 1;
		if (sb_pos(pv, rp.sb0 + g) - base <= p) {                  // answer in [g, nsb): gallop up
			uint64_t step = 1; lo = g;
			while (lo + step < nsb && sb_pos(pv, rp.sb0 + lo + step) - base <= p) { lo += step; step <<= 1; }
			hi = min(lo + step, nsb) - 1;                          // sbpos[lo] <= p; everything above hi is > p
		} else {                                                   // answer in [0, g): gallop down
			uint64_t step = 1; hi = g - 1;                         // g > 0 here: sbpos[sb0] - base == 0 <= p
			while (hi >= step && sb_pos(pv, rp.sb0 + hi - step + 1) - base > p) { hi -= step; step <<= 1; }
			lo = hi >= step ? hi - step + 1 : 0;                   // sbpos[lo] <= p (lo == 0 at worst); everything above hi is > p
		}
	}
	while (lo < hi) {
		const uint64_t mid = (lo + hi + 1) >> 1;
		if (sb_pos(pv, rp.sb0 + mid) - base <= p) lo = mid; else hi = mid - 1;
	}
	const uint64_t sbs = sb_pos(pv, rp.sb0 + lo) - base, l0 = (rp.sb0 + lo) * SB;
	const uint32_t rel = (uint32_t)(p - sbs);                  // < 2^16: a superblock holds at most SB * LEAF symbols
	// inside the superblock: the fills of its slots are one 48-byte read (dir_row 0); unused slots (n == 0) trail the used ones
	const uint4 *q = (const uint4*)dir_row(pv, rp.sb0 + lo, 0);
	uint32_t run = 0, klo = 0, pre = 0, nk = 0;
	auto scan8 = [&](int i) {
		const uint4 v = q[i];
		const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const uint32_t e = (j & 1) ? w[j >> 1] >> 16 : w[j >> 1] & 0xffffu, n = e & FILL_MASK;   // (bit 15: the leaf has a plane-2 line)
			if (n > 0 && run <= rel) { klo = (uint32_t)(8 * i + j); pre = run; nk = e; }
			run += n;
		}
	};
#pragma unroll
	for (int i = 0; i < SP_USED / 8; ++i) scan8(i);             // the slots a re-layout fills: three 16-byte loads, issued together
	if (run <= rel) scan8(SP_USED / 8);                        // the reserve slots (leaf splits): only when the position lies behind the first 24
	r.gl = l0 + klo; r.s = sbs + pre; r.n = nk & FILL_MASK; r.p2 = nk >> 15;
	return r;
}

// the dense layout needs no search: every leaf of a piece but the last is full
__device__ __forceinline__ Loc locate_dense(const RopeDesc &rp, uint64_t p)
{
	Loc r;
	const uint64_t lf = rp.nleaves ? min(p >> LEAF_SH, rp.nleaves - 1) : 0;
	r.gl = rp.leaf0 + lf; r.s = lf << LEAF_SH; r.n = (uint32_t)min((uint64_t)LEAF, rp.n - r.s); r.p2 = 1;
	return r;
}

// counts of all six symbols in [0,p) of a sub-rope on pool side `pv` (rope_rank1a, rope.h:45):
// superblock prefix + leaf-relative prefix + popcounts over the groups of the leaf up to p
// (the reference walks the runs of one leaf, rle.c:147-158).  SPARSE: leaves carry slack, the leaf is found by locate().
template <bool SPARSE = false, typename Q = uint64_t> __device__ inline void rank_all(const PoolView &pv, const RopeDesc &rp, uint64_t p, Q out[6])
{
	if (p >= rp.n) {
#pragma unroll
		for (int s = 0; s < 6; ++s) out[s] = (Q)rp.cnt[s];
		return;
	}
	uint64_t gl; uint32_t off; bool p2 = true;
	if (SPARSE) { const Loc lc = locate(pv, rp, p); gl = lc.gl; off = (uint32_t)(p - lc.s); p2 = lc.p2 != 0; }
	else { gl = rp.leaf0 + (p >> LEAF_SH); off = (uint32_t)(p & (LEAF - 1)); }
	uint32_t pc[6];                                            // symbols of the superblock in front of the leaf
	if (SPARSE) {
#pragma unroll
		for (int s = 0; s < 6; ++s) pc[s] = dir_prefix(pv, gl / SB, 1 + s, (uint32_t)(gl % SB));
	} else {
		const LeafMeta m = pv.meta[gl];
#pragma unroll
		for (int s = 0; s < 6; ++s) pc[s] = m.c[s];
	}
	PlAcc A;
	leaf_count(leaf_words(pv.data, gl), 0, off, A, p2);
	uint32_t c[6];
	pl_finish(A, off, c);
#pragma unroll
	for (int s = 0; s < 6; ++s) out[s] = (Q)(sb_cum(pv, gl / SB, s) - sb_cum(pv, rp.sb0, s) + pc[s] + c[s]);
}

// occurrences of the six symbols inside [l, u), l < u: what mr_insert_multi_aux needs from rope_rank2a (tu[] - tl[],
// mrope.c:202-224).  When the interval lies in one leaf -- the common case: intervals are short -- this is a scan of the
// interval itself, no directory and no prefix (rle_rank2a counts the same way, rle.c:134-191); else two full ranks.
// Q: the width the counts are wanted in (the difference of two ranks is exact modulo 2^32 when it is below 2^32: the string kernels ask for
// uint32_t while positions are stored in 32 bits and keep half the registers)
template <bool SPARSE = false, typename Q = uint64_t> __device__ inline void range_counts(const PoolView &pv, const RopeDesc &rp, uint64_t l, uint64_t u, Q d[6])
{
	uint64_t gl; uint32_t ol; bool one, p2 = true;
	if (SPARSE) { const Loc lc = locate(pv, rp, l); gl = lc.gl; ol = (uint32_t)(l - lc.s); one = u - lc.s <= lc.n; p2 = lc.p2 != 0; }
	else { const uint64_t lf = l >> LEAF_SH; gl = rp.leaf0 + lf; ol = (uint32_t)(l & (LEAF - 1)); one = ((u - 1) >> LEAF_SH) == lf; }
	if (one) {
		PlAcc A;
		const uint32_t ou = ol + (uint32_t)(u - l);
		leaf_count(leaf_words(pv.data, gl), ol, ou, A, p2);
		uint32_t c[6];
		pl_finish(A, ou - ol, c);
#pragma unroll
		for (int s = 0; s < 6; ++s) d[s] = c[s];
	} else {
		Q cl[6], cu[6];
		rank_all<SPARSE, Q>(pv, rp, l, cl);
		rank_all<SPARSE, Q>(pv, rp, u, cu);
#pragma unroll
		for (int s = 0; s < 6; ++s) d[s] = (Q)(cu[s] - cl[s]);
	}
}

} // namespace rb2

// ---------------------------------------------------------------------------------------------
// DPP wave primitives (gfx9/CDNA data-parallel-primitive controls; VALU only, no LDS round trip)
// ---------------------------------------------------------------------------------------------
namespace rb2 {

template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);   // lanes without a source read 0
}
// inclusive prefix sum over the 64 lanes: row_shr 1,2,4,8 then row_bcast15 / row_bcast31
__device__ __forceinline__ uint32_t dpp_incl_add(uint32_t v)
{
	v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
	v += dpp0<0x142, 0xa>(v); v += dpp0<0x143, 0xc>(v);
	return v;
}
// the same for 64-bit values: both halves travel by the same DPP move, one 64-bit add per step (the shuffle-based wave_incl_add<uint64_t>
// is twelve dependent LDS-crossbar round trips; k_setup runs ten such scans back to back on its one wave)
__device__ __forceinline__ uint64_t dpp_incl_add64(uint64_t v)
{
#define RB2_DPP64(CTRL, MASK) v += (uint64_t)dpp0<CTRL, MASK>((uint32_t)(v >> 32)) << 32 | dpp0<CTRL, MASK>((uint32_t)v)
	RB2_DPP64(0x111, 0xf); RB2_DPP64(0x112, 0xf); RB2_DPP64(0x114, 0xf); RB2_DPP64(0x118, 0xf); RB2_DPP64(0x142, 0xa); RB2_DPP64(0x143, 0xc);
#undef RB2_DPP64
	return v;
}
__device__ __forceinline__ uint32_t dpp_incl_max(uint32_t v)     // unsigned max, identity 0
{
	v = max(v, dpp0<0x111, 0xf>(v)); v = max(v, dpp0<0x112, 0xf>(v)); v = max(v, dpp0<0x114, 0xf>(v)); v = max(v, dpp0<0x118, 0xf>(v));
	v = max(v, dpp0<0x142, 0xa>(v)); v = max(v, dpp0<0x143, 0xc>(v));
	return v;
}
// a wave-uniform 64-bit value, in scalar registers from here on
__device__ __forceinline__ uint64_t uniform64(uint64_t v) { return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32 | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); }
// a value that is only looked at under the condition it was loaded under starts out as whatever its register holds: a zero is an instruction
// (__builtin_nondeterministic_value becomes a zero as well.  Each statement carries a number of its own: identical ones are merged into one register
// that is then copied.  Not volatile: a volatile statement counts as a write to memory, and uniform loads behind it turn into vector loads.)
#define RB2_UNDEFV(x) asm("" : "=v"(x) : "n"(__COUNTER__))
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }   // (__ballot takes an int: a flag kept in scalar registers is written to a vector register and compared again)
__device__ __forceinline__ uint32_t dpp_prev_lane(uint32_t v) { return dpp0<0x138, 0xf>(v); }   // wave_shr:1, lane 0 reads 0
__device__ __forceinline__ uint32_t dpp_next_lane(uint32_t v) { return dpp0<0x130, 0xf>(v); }   // wave_shl:1, lane 63 reads 0
__device__ __forceinline__ uint32_t lane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }


// ---------------------------------------------------------------------------------------------
// wave-cooperative rank: a leaf is 16 groups = one per lane of the first DPP row.  Three coalesced 128-byte loads, per-lane
// popcounts of the lane's share of [from, to), three packed DPP reductions -> the six counts in every lane.  Cost is independent
// of the interval length (the single-thread scan of leaf_count is linear in it): this is what serves long intervals and intervals
// that span leaves in k_prep, and every query of k_rank_batch.  (rle_rank2a, rle.c:134-191; rope_rank2a, rope.c:179-194.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_leaf_counts(const uint64_t *leaf, uint32_t from, uint32_t to, uint32_t c[6], bool p2 = true)
{
	const uint32_t g = (uint32_t)lane_id(), b = g << 6;
	PlAcc A;
	if (g < (uint32_t)LEAFG) {
		const uint32_t lo = from > b ? min(from - b, 64u) : 0u, hi = to > b ? min(to - b, 64u) : 0u;   // my symbols [lo, hi)
		const uint64_t b0 = leaf[g], b1 = leaf[LEAFG + g];
		pl_acc(A, b0, b1, p2 ? leaf[2 * LEAFG + g] : ~(b0 | b1), bits_below(hi) & ~bits_below(lo));
	}
	const uint32_t r0 = lane63(dpp_incl_add(A.p0 | A.p1 << 16)), r1 = lane63(dpp_incl_add(A.p2 | A.p01 << 16)), r2 = lane63(dpp_incl_add(A.p02));
	PlAcc T;
	T.p0 = r0 & 0xffffu; T.p1 = r0 >> 16; T.p2 = r1 & 0xffffu; T.p01 = r1 >> 16; T.p02 = r2;
	pl_finish(T, to > from ? to - from : 0u, c);
}

// all 64 lanes call it with the same p
template <bool SPARSE, typename Q = uint64_t> __device__ __forceinline__ void wave_rank_all(const PoolView &pv, const RopeDesc &rp, uint64_t p, Q out[6])
{
	if (p >= rp.n) {
#pragma unroll
		for (int s = 0; s < 6; ++s) out[s] = (Q)rp.cnt[s];
		return;
	}
	uint64_t gl; uint32_t off; bool p2 = true;
	if (SPARSE) { const Loc lc = locate(pv, rp, p); gl = lc.gl; off = (uint32_t)(p - lc.s); p2 = lc.p2 != 0; }
	else { gl = rp.leaf0 + (p >> LEAF_SH); off = (uint32_t)(p & (LEAF - 1)); }
	uint32_t pc[6];
	if (SPARSE) {                                              // lane j < k holds the counts of slot j: three packed wave sums
		const uint32_t k = (uint32_t)(gl % SB), ln = (uint32_t)lane_id();
		uint32_t v[6];
#pragma unroll
		for (int s = 0; s < 6; ++s) v[s] = ln < k ? dir_row(pv, gl / SB, 1 + s)[ln] : 0u;
		const uint32_t t01 = lane63(dpp_incl_add(v[0] | v[1] << 16)), t23 = lane63(dpp_incl_add(v[2] | v[3] << 16)), t45 = lane63(dpp_incl_add(v[4] | v[5] << 16));
		pc[0] = t01 & 0xffffu; pc[1] = t01 >> 16; pc[2] = t23 & 0xffffu; pc[3] = t23 >> 16; pc[4] = t45 & 0xffffu; pc[5] = t45 >> 16;
	} else {
		const LeafMeta m = pv.meta[gl];
#pragma unroll
		for (int s = 0; s < 6; ++s) pc[s] = m.c[s];
	}
	uint32_t c[6];
	wave_leaf_counts(leaf_words(pv.data, gl), 0, off, c, p2);
#pragma unroll
	for (int s = 0; s < 6; ++s) out[s] = (Q)(sb_cum(pv, gl / SB, s) - sb_cum(pv, rp.sb0, s) + pc[s] + c[s]);
}

// occurrences of the six symbols inside [l, u), l < u, by one wave (what range_counts does with one thread)
template <bool SPARSE, typename Q = uint64_t> __device__ __forceinline__ void wave_range_counts(const PoolView &pv, const RopeDesc &rp, uint64_t l, uint64_t u, Q d[6])
{
	uint64_t gl; uint32_t ol; bool one, p2 = true;
	if (SPARSE) { const Loc lc = locate(pv, rp, l); gl = lc.gl; ol = (uint32_t)(l - lc.s); one = u - lc.s <= lc.n; p2 = lc.p2 != 0; }
	else { const uint64_t lf = l >> LEAF_SH; gl = rp.leaf0 + lf; ol = (uint32_t)(l & (LEAF - 1)); one = ((u - 1) >> LEAF_SH) == lf; }
	if (one) {
		uint32_t c[6];
		wave_leaf_counts(leaf_words(pv.data, gl), ol, ol + (uint32_t)(u - l), c, p2);
#pragma unroll
		for (int s = 0; s < 6; ++s) d[s] = c[s];
	} else {
		Q cl[6], cu[6];
		wave_rank_all<SPARSE, Q>(pv, rp, l, cl);
		wave_rank_all<SPARSE, Q>(pv, rp, u, cu);
#pragma unroll
		for (int s = 0; s < 6; ++s) d[s] = (Q)(cu[s] - cl[s]);
	}
}

} // namespace rb2
