// rb2_kernels.h -- the per-round kernels of the gfx950 BCR insertion engine.
//
// One round = one string position (mrope.c:299-342).  The reference walks each bucket
// sequentially (mr_insert_multi_aux, mrope.c:184-233); here the same quantities are obtained from
// prefix sums:
//
//   k_sym      next symbol of every active string + group heads (mrope.c:189-192)
//   k_tscan*   prefix of the per-tile symbol histograms, nearest group head left/right of a tile
//   k_tfix     the same, folded into one record per tile for k_prep / k_advance; + the rows of the count matrix seen here
//   k_setup    NR x 6 count matrix -> new sub-rope sizes, AC offsets (mrope.c:332-336), next buckets
//   k_prep     per string: pre-round position of its new symbol, slot in the sorted insert list,
//              interval sizes via rank on non-empty intervals (mrope.c:199-224)
//   k_part     merge-path split: which inserts land in which output window -> LeafDesc work orders
//   k_merge    rank + positional insert, one wave per output window of WPL leaves (rb2_merge.h)
//   k_meta*    rank directory of the new side (replaces the rpnode_t counts, rope.h:11-15)
//   k_advance  new (l,u) per string + stable 6-way partition into next round's buckets
//              (mrope.c:226-229, 303-309, 332-340)
#pragma once
#include "rb2_device.h"

namespace rb2 {

// ---------------------------------------------------------------------------------------------
// batch set-up: find the sentinels, string starts, initial state (mrope.c:269-284)
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)   // number of 0x00 bytes in x (exact)
{
	const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;
	return __popc(~t & 0x80808080u);
}

// each thread owns 64 consecutive bytes; returns them in w[16] (zero padded with 0xff beyond len)
__device__ __forceinline__ void load64(const uint8_t *s, uint64_t len, uint64_t base, uint32_t w[16])
{
	if (base + 64 <= len) {
		const uint4 *p = (const uint4*)(s + base);
#pragma unroll
		for (int i = 0; i < 4; ++i) { uint4 v = p[i]; w[4*i] = v.x; w[4*i+1] = v.y; w[4*i+2] = v.z; w[4*i+3] = v.w; }
	} else {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			uint32_t x = 0;
			for (int k = 0; k < 4; ++k) {
				const uint64_t p = base + 4*i + k;
				x |= (uint32_t)(p < len ? s[p] : 0xffu) << (8*k);
			}
			w[i] = x;
		}
	}
}

// also validates the input: *bad_flag is set when a byte is not an nt6 code 0..5 (the reference indexes arrays with
// these bytes unchecked, mrope.c:204; here a stray value would corrupt the bucket bookkeeping silently)
__global__ __launch_bounds__(256) void k_count_zeros(const uint8_t *s, uint64_t len, uint64_t *blk, uint64_t *bad_flag)
{
	__shared__ uint32_t s_w[4];
	uint32_t w[16], c = 0, bad = 0;
	const uint64_t base = (uint64_t)blockIdx.x * ZBLOCK + threadIdx.x * 64;
	load64(s, len, base, w);
#pragma unroll
	for (int i = 0; i < 16; ++i) c += zero_bytes(w[i]);
	if (base + 64 <= len) {
#pragma unroll
		for (int i = 0; i < 16; ++i) bad |= (w[i] & 0xf8f8f8f8u) | ((w[i] >> 1) & (w[i] >> 2) & 0x01010101u);   // > 7, or 6/7
	} else for (uint64_t p = base; p < len; ++p) bad |= s[p] > 5;
	uint32_t tot;
	block_excl_add<uint32_t>(c, s_w, &tot);
	if (threadIdx.x == 0) blk[blockIdx.x] = tot;
	if (bad) *bad_flag = 1;
}

// What a batch adds to the count matrix, known before a single round has run: symbol t[i] of the batch text goes into the rope of
// the symbol inserted before it -- t[i-1], and rope $ for the first symbol of a string, which follows the sentinel 0 of the string
// in front of it (mrope.c:299-342) -- so c[a][b] grows by the number of adjacent pairs (a, b) of the text, with t[-1] = 0.  A host
// that only wants mr_get_c() / mr_get_ac() to be truthful after mr_insert_multi need not wait for the rounds.  out[a*6+b].
__global__ __launch_bounds__(256) void k_pair_hist(const uint8_t *s /* 16-byte aligned */, uint64_t len, unsigned long long *out)
{
	__shared__ uint32_t h[36];
	if (threadIdx.x < 36) h[threadIdx.x] = 0;
	__syncthreads();
	uint32_t loc[36];
#pragma unroll
	for (int i = 0; i < 36; ++i) loc[i] = 0;
	const uint64_t K = 0x0101010101010101ull, L7 = 0x7f7f7f7f7f7f7f7full;
	auto eq = [&](uint64_t x, uint32_t v) -> uint64_t { const uint64_t t = x ^ (K * v); return ~(((t & L7) + L7) | t | L7); };   // 0x80 in every byte of x that equals v
	for (uint64_t p = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; p < len; p += (uint64_t)gridDim.x * 256 * 16) {
		uint64_t w[2];
		if (p + 16 <= len) { const uint4 v = *(const uint4*)(s + p); w[0] = v.x | (uint64_t)v.y << 32; w[1] = v.z | (uint64_t)v.w << 32; }
		else { w[0] = w[1] = 0x0707070707070707ull; for (uint64_t q = p; q < len; ++q) { uint64_t &d = w[(q - p) >> 3]; const int sh = 8 * (int)((q - p) & 7); d = (d & ~(0xffull << sh)) | (uint64_t)s[q] << sh; } }   // 7: no symbol
		uint64_t prev = p ? s[p - 1] : 0ull;
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const uint64_t cur = w[k], pw = cur << 8 | prev;      // byte i of pw = the byte in front of byte i of cur
			prev = cur >> 56;
			uint64_t mp[6], mc[6];
#pragma unroll
			for (int v = 0; v < 6; ++v) { mp[v] = eq(pw, v); mc[v] = eq(cur, v); }
#pragma unroll
			for (int a = 0; a < 6; ++a)
#pragma unroll
				for (int b = 0; b < 6; ++b) loc[a * 6 + b] += (uint32_t)__popcll(mp[a] & mc[b]);
		}
	}
#pragma unroll
	for (int i = 0; i < 36; ++i) {                             // wave sum, one LDS atomic per wave and counter
		uint32_t v = loc[i];
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor((int)v, o);
		if (lane_id() == 0 && v) atomicAdd(&h[i], v);
	}
	__syncthreads();
	if (threadIdx.x < 36 && h[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// in-place exclusive scan of the per-block sentinel counts (one block); blk[n] = total = number of strings
__global__ __launch_bounds__(SCHUNK) void k_zscan(uint64_t *blk, uint32_t n)
{
	__shared__ uint64_t s_w[16];
	uint64_t run = 0;
	for (uint32_t i0 = 0; i0 < n; i0 += SCHUNK) {
		const uint32_t i = i0 + threadIdx.x;
		const uint64_t v = i < n ? blk[i] : 0ull;
		uint64_t tot;
		const uint64_t ex = block_excl_add<uint64_t>(v, s_w, &tot);
		if (i < n) blk[i] = run + ex;
		run += tot;
	}
	if (threadIdx.x == 0) blk[n] = run;
}

// START[id+1] = position after the id-th sentinel; START[0] = 0
__global__ __launch_bounds__(256) void k_write_starts(const uint8_t *s, uint64_t len, const uint64_t *blkoff, uint64_t *START)
{
	__shared__ uint32_t s_w[4];
	uint32_t w[16], c = 0;
	const uint64_t base = (uint64_t)blockIdx.x * ZBLOCK + threadIdx.x * 64;
	load64(s, len, base, w);
	if (blockIdx.x == 0 && threadIdx.x == 0) START[0] = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) c += zero_bytes(w[i]);
	uint64_t id = blkoff[blockIdx.x] + block_excl_add<uint32_t>(c, s_w, (uint32_t*)0);
	if (c == 0) return;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		if (zero_bytes(w[i]) == 0) continue;
		for (int k = 0; k < 4; ++k)
			if (((w[i] >> (8*k)) & 0xff) == 0) START[++id] = base + 4*i + k + 1;
	}
}

// The per-string cursor word W that rides through the partition of every round: the next symbols of the string -- up to CUR_SYMS of
// them, 3 bits each, stored as code + 1 so that an empty slot is 0 (low 27 bits) -- and, above them, the position in the batch text
// where the symbols behind those start (36 bits: a batch has < 64 GiB).  A string whose cursor runs empty refills it by ONE gather
// from the text at that position.  The strings of a batch start with 1 .. CUR_SYMS symbols in turn, so in every round one string in
// CUR_SYMS refills: the random reads are spread over all rounds and run beside the streaming traffic of the same launch (while all
// strings refilled in the same rounds, every ninth k_advance took 3.8 times as long as the others).  (Rounds 1-3 carried the string
// id instead and looked the position up in START[id]: a second random 8-byte read per refill, 64 bytes of traffic for it.)
// The symbol a string inserts NEXT round also travels as a byte of its own (array A, written by k_advance at the string's new place):
// k_sym reads one byte per string instead of the 8-byte word.
constexpr int CUR_SYMS = 9;
constexpr int CUR_BITS = 3 * CUR_SYMS;
constexpr uint64_t CUR_MASK = (1ull << CUR_BITS) - 1ull;
__device__ __forceinline__ uint32_t tri4(uint32_t x)      // (low 3 bits of 4 bytes) + 1 each -> 12 bits
{
	x = (x & 0x07070707u) + 0x01010101u;
	x = (x | x >> 5) & 0x003f003fu;
	return (x | x >> 10) & 0xfffu;
}
__device__ __forceinline__ uint32_t pack9(const uint8_t *s, uint64_t len, uint64_t p)   // s[p .. p+9) as 27 bits, code + 1 each (bytes past the end read as 0)
{
	if (p + 16 <= len) {
		const uint32_t *q = (const uint32_t*)(s + (p & ~3ull));
		const uint32_t sh = (uint32_t)(p & 3) * 8;
		const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
		const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sh), w1 = __builtin_amdgcn_alignbit(d2, d1, sh), w2 = __builtin_amdgcn_alignbit(d3, d2, sh);
		return tri4(w0) | tri4(w1) << 12 | (tri4(w2) & 0x7u) << 24;
	}
	uint32_t w = 0;
	for (int i = 0; i < CUR_SYMS; ++i) {
		const uint64_t q = p + i;
		w |= (uint32_t)((q < len ? (s[q] & 7) : 0) + 1) << (3 * i);
	}
	return w;
}
// ... from the four dwords at s + (p & ~3) (the fast path of pack9, split so that the load can be issued long before its value is needed)
__device__ __forceinline__ uint32_t pack9_words(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3, uint32_t sh /* 8 * (p & 3) */)
{
	const uint32_t w0 = __builtin_amdgcn_alignbit(d1, d0, sh), w1 = __builtin_amdgcn_alignbit(d2, d1, sh), w2 = __builtin_amdgcn_alignbit(d3, d2, sh);
	return tri4(w0) | tri4(w1) << 12 | (tri4(w2) & 0x7u) << 24;
}
__device__ __forceinline__ uint64_t cur_make(uint64_t next_pos, uint32_t syms) { return next_pos << CUR_BITS | syms; }
__device__ __forceinline__ uint64_t cur_pos(uint64_t w) { return w >> CUR_BITS; }
__device__ __forceinline__ bool cur_empty(uint64_t w) { return (w & 7) == 0; }
__device__ __forceinline__ int cur_sym(uint64_t w) { return (int)(w & 7) - 1; }
__device__ __forceinline__ uint64_t cur_next(uint64_t w) { return (w & ~CUR_MASK) | ((w & CUR_MASK) >> 3); }   // one symbol consumed
__device__ __forceinline__ uint64_t cur_refill(const uint8_t *s, uint64_t len, uint64_t w) { const uint64_t p = cur_pos(w); return cur_make(p + CUR_SYMS, pack9(s, len, p)); }

template <typename P = uint64_t> __global__ __launch_bounds__(256) void k_init_strings(Ctl *ctl, int is_srt, const uint8_t *s, const uint64_t *START,
		P *L, P *U, uint64_t *W, uint8_t *A)
{
	__shared__ int s_wm[4];
	const uint64_t m = ctl->n_strings, k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	uint64_t ln = 0;
	if (k < m) {
		const uint64_t st = START[k], n0 = ctl->n0;
		ln = START[k+1] - 1 - st;
		L[k] = is_srt ? 0 : n0 + k;                 // mrope.c:280-283
		U[k] = is_srt ? n0 : n0 + k;
		const uint32_t n1 = 1 + (uint32_t)(k % CUR_SYMS);       // the refills of the batch take turns (see above)
		const uint32_t c9 = pack9(s, ctl->len, st) & ((1u << 3 * n1) - 1u);
		W[k] = cur_make(st + n1, c9);
		A[k] = (uint8_t)cur_sym(c9);
	}
	// block max of the lengths -> ctl->max_len
	unsigned long long v = ln;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { unsigned long long t = __shfl_xor(v, d); v = t > v ? t : v; }
	(void)s_wm;
	// one atomic per wave only while the maximum still grows (40 M strings of equal length: a handful in total)
	if (lane_id() == 0 && v > *(volatile uint64_t*)&ctl->max_len) atomicMax((unsigned long long*)&ctl->max_len, v);
}

// 32-bit positions -> 64-bit (the engine leaves the narrow storage mode: some piece may reach 2^32 symbols next round)
__global__ __launch_bounds__(256) void k_widen(const uint32_t *src, uint64_t *dst, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n) dst[i] = src[i];
}

// round 0: every string sits in "bucket 0" and inserts its last symbol into rope $ (mrope.c:285)
__global__ void k_batch_setup(Ctl *ctl, int side, uint64_t m, uint64_t len, int is_srt)
{
	if (threadIdx.x || blockIdx.x) return;
	SegDesc &sg = ctl->seg[side];
	for (int b = 0; b < NR; ++b) { sg.start[b] = 0; sg.cnt[b] = 0; }
	sg.cnt[0] = ctl->own[0] ? m : 0;                        // sharded: only the owner of rope $ starts with the strings
	uint32_t t = 0;
	for (int b = 0; b < NR; ++b) { sg.tile0[b] = t; t += (uint32_t)((sg.cnt[b] + STILE - 1) / STILE); }
	sg.tile0[NR] = t; sg.tile0[NR + 1] = t;
	uint64_t n0 = 0;
	for (int b = 0; b < NR; ++b) n0 += ctl->rope[side][b].cnt[0];
	ctl->n0 = n0; ctl->n_strings = m; ctl->max_len = 0; ctl->len = len;
	ctl->ne[0] = (is_srt && n0) ? 1u : 0u;          // round 0: [0, n0) for every string in the sorted modes (mrope.c:280-283)
	ctl->ne[1] = 0;
}

// ---------------------------------------------------------------------------------------------
// k_sym: a[k].c = *a[k].p++ (mrope.c:189-190) + group heads (mrope.c:192) + tile summaries
// ---------------------------------------------------------------------------------------------

struct TileCtx { int b; uint64_t lt, base, segstart, segend; };

__device__ __forceinline__ bool tile_ctx(const SegDesc &sg, uint32_t tile, TileCtx &t)
{
	if (tile >= sg.tile0[NR]) return false;
	const int b = seg_of(sg.tile0, tile);                  // wave-uniform: one load + ballot instead of a walk over NR entries
	t.b = b; t.lt = tile - sg.tile0[b];
	t.segstart = sg.start[b]; t.segend = sg.start[b] + sg.cnt[b];
	t.base = t.segstart + t.lt * STILE;
	return true;
}

// When every interval of the round is empty (u == l, ctl->ne[par] == 0) U is dead and L is read in its place.
// Tile kernels (k_sym, k_prep, k_part_sparse, k_advance) walk the string tiles with a GRID STRIDE: on one GPU the grid covers every
// tile the round can have and a block runs one; a rank of a sharded index launches about twice its fair share of the batch's
// tiles (the host cannot know how many strings the rank holds this round without asking the device) and a block takes more
// tiles when the rank holds more -- an upper-bound grid of N times the work would cost more in empty workgroups than the work itself.
// STRIDE = false is the one-GPU kernel, one tile per block and no loop (the loop costs registers: k_advance 87 -> 112 VGPRs, k_prep
// with interval counts 121 -> 254); the host launches STRIDE = true only on a rank of a sharded index.
// SPLIT: the launch that follows an in-place round on one GPU also does that round's leaf splits (k_split), in nsplitb blocks of its own
// in front of the tile blocks: k_sym reads string arrays only, the splits touch the pool only -- one launch instead of two, the two running
// side by side; the verdict of the round reaches the host behind this launch.
struct SplitArgs { Ctl *ctl; PoolView pool; const uint32_t *SPL; uint32_t spl_cap, epoch; volatile uint32_t *hv; uint32_t nsplitb;
	uint32_t round1;    // the in-place round whose splits these are, + 1: reported to the host (hv[2]) -- it queues in-place rounds without waiting for them, but only a few ahead
	SbBase *scan2; };   // scan2 != null: one more block, behind the split blocks, turns the chunk totals k_advance's scan blocks left into chunk bases (sbscan2_body; "the directory rides along", below)
template <int NT, int CT> __device__ __forceinline__ void sbscan2_lean(const Ctl *ctl, SbBase *base, uint64_t *s_w /* NT / 64 words */);
template <int NT, int CT> __device__ __forceinline__ void sbscan2_col(const Ctl *ctl, SbBase *base, uint64_t *s_w, const int col);
__device__ __forceinline__ void split_body(Ctl *ctl, const PoolView &pool, const uint32_t *SPL, uint32_t spl_cap, uint32_t epoch, volatile uint32_t *hv,
		const uint32_t bidx, const uint32_t nblk, uint16_t (*s_row)[7][SB], uint32_t round1);
// Fused k_prep: in a round whose intervals are all empty (ctl->ne[par] == 0) a tile in which every string is a group of its own -- the
// rule from round ~14 of a batch on -- needs nothing from the tile scans to place its new symbols: slot = the string's index in its
// bucket, e = l - slot (prep_tile's all-single path).  k_sym has l in hand (it reads L in U's place) and writes INS_E / INS_A itself;
// the tile is marked done (bit TILE_DONE of its `lh` record -> bit 1 of TileFix::nexthead) and k_prep<AE> returns at once for it.
// (r04: k_prep<AE> was a second pass over the same strings at 2.3 TB/s: 0.18 ms per round of 42 M strings.)
constexpr int32_t TILE_DONE = 0x10000;       // in TileRecs::lh (a head index is < STILE)
constexpr int32_t TILE_SINGLE = 0x20000;     // ... every string of the tile is a group of its own and so is the string behind it (-> TileFix::nexthead bit 2: k_advance asks for its gathers before the barriers only then)
template <bool STRIDE, typename P = uint64_t, bool SPLIT = false> __global__ __launch_bounds__(256) void k_sym(const Ctl *ctl, int side, int par, const P *L, const P *UU,
		uint8_t *A /* in: the symbol every string inserts this round (k_init_strings / k_advance); out: + the group-head flag */, TileRecs trec, SplitArgs sp,
		P *INS_E, uint8_t *INS_A)
{
	__shared__ uint64_t s_bal[8][6], s_head[8];
	__shared__ __align__(16) uint32_t s_ok[4];                 // per wave: every string of the wave is a group of its own
	if (SPLIT) {
		__shared__ uint16_t s_row[MW][7][SB];
		// the FIRST blocks of the grid: the splits' registers (99 VGPRs) cap the launch at five workgroups per CU, the tile blocks take two
		// turns -- behind them the split blocks started when the first turn was over (16.9 us for the launch; 6.2 + 9.3 apart)
		if (blockIdx.x < sp.nsplitb) { split_body(sp.ctl, sp.pool, sp.SPL, sp.spl_cap, sp.epoch, sp.hv, blockIdx.x, sp.nsplitb, s_row, sp.round1); return; }
		if (sp.scan2 && blockIdx.x == sp.nsplitb) { __shared__ uint64_t s_w2[4]; sbscan2_lean<256, 2>(ctl, sp.scan2, s_w2); return; }
	}
	const bool ae = ctl->ne[par] == 0;
	const P *U = ae ? L : UU;
	for (uint32_t tile = (STRIDE || SPLIT) ? blockIdx.x - (SPLIT ? sp.nsplitb + (sp.scan2 ? 1u : 0u) : 0u) : xcd_item(); ; tile += gridDim.x) {     // the first tile as ever (its loads issue at once); the bound ends the walk
	if (STRIDE && tile != blockIdx.x) __syncthreads();          // the LDS tables of the previous tile are done with (STRIDE and SPLIT never come together)
	TileCtx t;
	if (!tile_ctx(ctl->seg[side], tile, t)) return;
	const int ln = lane_id(), w = wave_id();
	// all loads of the thread's two strings first: A is a byte array (it may alias anything as far as the compiler knows), so a load
	// written behind the store of the first string's flag would wait for it
	// every access of the tile is a wave-uniform base + a 32-bit offset (string x of the tile = string t.base + x of the arrays; its slot in the
	// bucket's insert list, lt * STILE + x, is the same place in INS_E / INS_A): no 64-bit address arithmetic per lane
	const uint32_t nval = (uint32_t)min((uint64_t)STILE, t.segend - t.base);   // strings in this tile
	const uint8_t *Ab = A + t.base; const P *Ub = U + t.base;
	const bool first_tile = t.base == t.segstart;
	uint32_t av[2]; P uv[2], up[2], un = 0;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint32_t x = (uint32_t)(h * 256) + threadIdx.x;
		RB2_UNDEFV(av[h]); RB2_UNDEFV(uv[h]); RB2_UNDEFV(up[h]);   // (only looked at where x < nval)
		if (x < nval) { av[h] = Ab[x]; uv[h] = Ub[x]; up[h] = (x > 0 || !first_tile) ? Ub[(int32_t)x - 1] : (P)0; }
	}
	const bool last_thread = threadIdx.x == 255;
	const bool has_next = last_thread && t.base + STILE < t.segend;   // the string behind the tile (same bucket): does it start a group?
	if (has_next) un = Ub[STILE];
	bool single = true;                                       // my strings are groups of their own (and, last thread: so is the tile's end)
	int sym2[2];
	uint8_t *Aw = A + t.base;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint32_t x = (uint32_t)(h * 256) + threadIdx.x;
		int sym = 7; bool head = false;
		if (x < nval) {
			sym = (int)(av[h] & 7);
			head = (x == 0 && first_tile) || (uv[h] != up[h]);
			Aw[x] = (uint8_t)(sym | (head ? 0x80 : 0));
			single = single && head;
		}
		sym2[h] = sym;
		const int c = h * 4 + w;
		uint64_t bm[6];                                             // all ballots, then one lane-0 block that stores them (group_setup)
#pragma unroll
		for (int s = 0; s < 6; ++s) bm[s] = ballot64(sym == s);
		const uint64_t hm = ballot64(head);
		if (ln == 0) {
#pragma unroll
			for (int s = 0; s < 6; ++s) s_bal[c][s] = bm[s];
			s_head[c] = hm;
		}
	}
	if (has_next) single = single && un != uv[1];
	// "every string of the tile is a group of its own": a scalar comparison per wave, a flag per wave, ONE barrier (the one the tile summaries need
	// anyway) -- __syncthreads_and was a DPP reduction, an LDS atomic and three barriers
	{ const uint64_t sm = ballot64(single); if (ln == 0) s_ok[w] = sm == ~0ull ? 1u : 0u; }
	__syncthreads();
	const uint4 okv = *(const uint4*)s_ok;
	const bool allsingle = (okv.x & okv.y & okv.z & okv.w) != 0;
	const bool fused = ae && allsingle;
	if (fused) {
		P *Eb = INS_E + t.base; uint8_t *Ib = INS_A + t.base;     // slot + segstart = t.base + x
		const P slot0 = (P)(t.lt * STILE);
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const uint32_t x = (uint32_t)(h * 256) + threadIdx.x;
			if (x >= nval) continue;
			Eb[x] = (P)(uv[h] - (slot0 + (P)x));                   // empty interval: the new symbol goes to l - F (pre-round coordinates), F = slot = lt * STILE + x
			Ib[x] = (uint8_t)sym2[h];
		}
	}
	if (threadIdx.x < 6) {
		const int s = threadIdx.x;
		uint32_t run = 0, fhpre = 0, lhpre = 0; int fh = -1, lh = -1;
		if (allsingle) {                                        // (block-uniform; the rule from round ~14 on) every string is a head: the first one is string 0, the last one string nval - 1
#pragma unroll
			for (int c = 0; c < 8; ++c) run += __popcll(s_bal[c][s]);
			if (nval) { fh = 0; lh = (int)nval - 1; lhpre = run - (uint32_t)((s_bal[lh >> 6][s] >> (lh & 63)) & 1ull); }
		} else
		for (int c = 0; c < 8; ++c) {
			const uint64_t hm = s_head[c], bm = s_bal[c][s];
			if (hm) {
				const int f = __builtin_ctzll(hm), l = 63 - __builtin_clzll(hm);
				if (fh < 0) { fh = c * 64 + f; fhpre = run + __popcll(bm & lt_mask(f)); }
				lh = c * 64 + l; lhpre = run + __popcll(bm & lt_mask(l));
			}
			run += __popcll(bm);
		}
		trec.hist(s, tile) = run; trec.fhpre(s, tile) = fhpre; trec.lhpre(s, tile) = lhpre;
		if (s == 0) { trec.fh(tile) = fh; trec.lh(tile) = lh | (fused ? TILE_DONE : 0) | (allsingle ? TILE_SINGLE : 0); }   // (all-single: every string is a head, lh >= 0)
	}
	if (!STRIDE) return;
	}
}

// ---------------------------------------------------------------------------------------------
// tile scan (3 kernels): exclusive add-scan of hist, nearest head-bearing tile to the left/right
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(SCHUNK) void k_tscan1(const Ctl *ctl, int side, const TileRecs trec, ChunkPart *part)
{
	__shared__ uint32_t s_w[16]; __shared__ int s_wi[16];
	const uint32_t nt = ctl->seg[side].tile0[NR];
	const uint32_t t = blockIdx.x * SCHUNK + threadIdx.x;
	if (blockIdx.x * SCHUNK >= nt) return;
	const bool ok = t < nt;
	uint32_t tot;
	ChunkPart p;
	for (int s = 0; s < 6; ++s) { block_excl_add<uint32_t>(ok ? trec.hist(s, t) : 0u, s_w, &tot); p.sum[s] = tot; }
	const bool hh = ok && trec.fh(t) >= 0;
	int v = hh ? (int)t : -1;
	v = wave_incl_max(v);
	if (lane_id() == 63) s_wi[wave_id()] = v;
	__syncthreads();
	int mx = -1; for (int i = 0; i < SCHUNK / 64; ++i) mx = max(mx, s_wi[i]);
	__syncthreads();
	v = hh ? (int)t : INT_MAX;
	v = wave_incl_min_down(v);
	if (lane_id() == 0) s_wi[wave_id()] = v;
	__syncthreads();
	int mn = INT_MAX; for (int i = 0; i < SCHUNK / 64; ++i) mn = min(mn, s_wi[i]);
	p.mx = mx; p.mn = mn;
	if (threadIdx.x == 0) part[blockIdx.x] = p;
}

__device__ __forceinline__ int block_all_max(int v, int *s_w)      // maximum over the block, returned to every thread
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
	if (lane_id() == 0) s_w[wave_id()] = v;
	__syncthreads();
	int m = s_w[0];
	for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = max(m, s_w[i]);
	__syncthreads();
	return m;
}

// one block, any number of chunks: SCHUNK at a time with running carries (sums and the last head run forwards, the
// next head runs backwards)
__global__ __launch_bounds__(SCHUNK) void k_tscan2(const Ctl *ctl, int side, ChunkPart *part)
{
	__shared__ uint32_t s_w[16]; __shared__ int s_wi[16];
	const uint32_t nt = ctl->seg[side].tile0[NR];
	const uint32_t nc = (nt + SCHUNK - 1) / SCHUNK;
	uint32_t run[6] = {0, 0, 0, 0, 0, 0};
	int run_mx = -1;
	for (uint32_t i0 = 0; i0 < nc; i0 += SCHUNK) {
		const uint32_t i = i0 + threadIdx.x;
		const bool ok = i < nc;
		ChunkPart p;
		if (ok) p = part[i];
		else { for (int s = 0; s < 6; ++s) p.sum[s] = 0; p.mx = -1; p.mn = INT_MAX; }
		uint32_t o[6];
		for (int s = 0; s < 6; ++s) { uint32_t tot; o[s] = run[s] + block_excl_add<uint32_t>(p.sum[s], s_w, &tot); run[s] += tot; }
		const int omx = max(run_mx, block_excl_max(p.mx, s_wi, -1));
		run_mx = max(run_mx, block_all_max(p.mx, s_wi));
		if (ok) { for (int s = 0; s < 6; ++s) part[i].sum[s] = o[s]; part[i].mx = omx; }   // mn is still the chunk's own
	}
	int run_mn = INT_MAX;
	for (uint32_t k = (nc + SCHUNK - 1) / SCHUNK; k-- > 0; ) {
		const uint32_t i = k * SCHUNK + threadIdx.x;
		const bool ok = i < nc;
		const int mn = ok ? part[i].mn : INT_MAX;
		const int omn = min(run_mn, block_excl_min_down(mn, s_wi, INT_MAX));
		run_mn = min(run_mn, -block_all_max(mn == INT_MAX ? INT_MIN + 1 : -mn, s_wi));
		if (ok) part[i].mn = omn;
	}
}

// FOLD: at most SCHUNK chunks (2^20 tiles: every batch of up to 2^29 strings) -- no k_tscan2 launch between k_tscan1 and this kernel: `part`
// holds the chunks' own totals and every block sums the ones in front of its chunk itself (a few KB of cache-resident reads against a
// single-block launch that the whole round waits for)
template <bool FOLD> __global__ __launch_bounds__(SCHUNK) void k_tscan3(const Ctl *ctl, int side, const TileRecs trec, const ChunkPart *part, TileScan *tsc)
{
	__shared__ uint32_t s_w[16]; __shared__ int s_wi[16];
	const uint32_t nt = ctl->seg[side].tile0[NR];
	const uint32_t t = blockIdx.x * SCHUNK + threadIdx.x;
	if (blockIdx.x * SCHUNK >= nt) return;
	const bool ok = t < nt;
	ChunkPart cp;
	if (FOLD) {
		const uint32_t nc = (nt + SCHUNK - 1) / SCHUNK, i = threadIdx.x;
		ChunkPart p;
		for (int s = 0; s < 6; ++s) p.sum[s] = 0;
		p.mx = -1; p.mn = INT_MAX;
		if (i < nc) p = part[i];
		for (int s = 0; s < 6; ++s) { uint32_t tot; block_excl_add<uint32_t>(i < blockIdx.x ? p.sum[s] : 0u, s_w, &tot); cp.sum[s] = tot; }
		cp.mx = block_all_max(i < blockIdx.x ? p.mx : -1, s_wi);
		const int nmn = block_all_max((i > blockIdx.x && p.mn != INT_MAX) ? -p.mn : INT_MIN + 1, s_wi);
		cp.mn = nmn == INT_MIN + 1 ? INT_MAX : -nmn;
	} else cp = part[blockIdx.x];
	TileScan o;
	uint32_t h[6];
	for (int s = 0; s < 6; ++s) {
		h[s] = ok ? trec.hist(s, t) : 0u;
		o.pre[s] = cp.sum[s] + block_excl_add<uint32_t>(h[s], s_w, (uint32_t*)0);
	}
	const bool hh = ok && trec.fh(t) >= 0;
	o.lht = max(cp.mx, block_excl_max(hh ? (int)t : -1, s_wi, -1));
	o.nht = min(cp.mn, block_excl_min_down(hh ? (int)t : INT_MAX, s_wi, INT_MAX));
	if (ok) tsc[t] = o;
	if (ok && t == nt - 1) {                                  // grand total at index nt
		TileScan e;
		for (int s = 0; s < 6; ++s) e.pre[s] = o.pre[s] + h[s];
		e.lht = -1; e.nht = INT_MAX;
		tsc[nt] = e;
	}
}

// ---------------------------------------------------------------------------------------------
// k_setup: everything the rest of the round needs that depends on the NR x 6 count matrix
// ---------------------------------------------------------------------------------------------

// one thread per string tile: fold the tile scans into the numbers group_setup needs, so that k_prep and
// k_advance read one 76-byte record per block instead of chasing tsc[lt] / trec[lt] / tsc[nt] / trec[nt]
// Block 0 also writes the rows of the count matrix this rank can see -- gcnt[r * 6 + a] = members of (local) bucket r inserting a --
// and word NR * 6 of the buffer (GCN, rb2_device.h), and, on one GPU (do_setup), runs k_setup of the round on its first wave: the local
// matrix IS the global one.  (Two launches of their own before; nothing k_setup writes is read by the other blocks of this kernel.)
template <bool SPARSE> __device__ __forceinline__ void setup_body(Ctl *ctl, int side, const uint64_t *gcnt, int par, uint32_t round, volatile unsigned long long *hmax, bool keep_ne = false);
// spec: as in k_tscan_setup below -- queued before the host saw the verdict of the in-place round in front of it; a void round (ctl->overflow) is redone
// from its own counting phase, and k_setup of the NEXT round must not have replaced the descriptors it starts from.
__global__ __launch_bounds__(256) void k_tfix(Ctl *ctl, int side, int par, const TileRecs trec, const TileScan *tsc, TileFix *tf, uint64_t *gcnt, int do_setup, int sparse,
		uint32_t round, volatile unsigned long long *hmax, int spec)
{
	__shared__ uint32_t s_t0[NR + 1];
	__shared__ uint64_t s_g[NR * 6];
	if (spec && ctl->overflow) return;
	const SegDesc &sg = ctl->seg[side];
	if (threadIdx.x <= NR) s_t0[threadIdx.x] = sg.tile0[threadIdx.x];
	__syncthreads();
	if (blockIdx.x == 0) {
		const int i = threadIdx.x;
		if (i < NR * 6) { const int b = i / 6, a = i % 6; const uint64_t v = s_t0[NR] ? (uint64_t)(tsc[s_t0[b + 1]].pre[a] - tsc[s_t0[b]].pre[a]) : 0ull; gcnt[i] = v; s_g[i] = v; }
		if (i == NR * 6) gcnt[NR * 6] = ctl->ne[par];
		if (do_setup) {                                          // (block-uniform)
			__syncthreads();
			if (i < 64) { if (sparse) setup_body<true>(ctl, side, s_g, par, round, hmax); else setup_body<false>(ctl, side, s_g, par, round, hmax); }
		}
	}
	const uint32_t tile = blockIdx.x * 256 + threadIdx.x;
	if (tile >= s_t0[NR]) return;
	int b = 0;
	while (tile >= s_t0[b+1]) ++b;
	const uint32_t t0 = s_t0[b], t1 = s_t0[b+1];
	const TileScan me = tsc[tile], first = tsc[t0];
	const int lt = me.lht, nt = me.nht;
	TileFix f;
	TileScan sl, sn;
	const bool hl = lt >= (int)t0, hn = nt < (int)t1;
	if (hl) sl = tsc[lt];
	if (hn) sn = tsc[nt]; else sn = tsc[t1];
	for (int s = 0; s < 6; ++s) {
		f.tpre[s] = me.pre[s] - first.pre[s];
		f.popen[s] = hl ? sl.pre[s] - first.pre[s] + trec.lhpre(s, (uint32_t)lt) : 0u;
		f.pnext[s] = sn.pre[s] - first.pre[s] + (hn ? trec.fhpre(s, (uint32_t)nt) : 0u);
	}
	f.fopen = hl ? (uint32_t)((lt - t0) * STILE + (trec.lh((uint32_t)lt) & (TILE_DONE - 1))) : 0u;
	const int32_t mylh = trec.lh(tile);
	f.b = (uint32_t)b; f.lt = tile - t0; f.nexthead = ((tile + 1 >= t1 || trec.fh(tile + 1) == 0) ? 1u : 0u) | ((mylh >= 0 && (mylh & TILE_DONE)) ? 2u : 0u) | ((mylh >= 0 && (mylh & TILE_SINGLE)) ? 4u : 0u);
	f.segstart = sg.start[b]; f.segend = sg.start[b] + sg.cnt[b];
	tf[tile] = f;
}

// gcnt = the GLOBAL NR x 6 count matrix of the round (== the local one on a single GPU; the sum
// over ranks when sub-ropes are sharded).  One wave, lane r = sub-rope r; the running sums of the
// sequential formulation (mrope.c:332-340) are wave scans.
// SPARSE: the round inserts in place -- every piece keeps its slots (leaf0, nleaves, sb0), only n and the counts move.
// hmax (pinned host memory, may be null): the round and the size of the largest piece after it -- what the host needs to know
// to keep the per-string positions in 32-bit storage for as long as they fit (rb2_device.h "P"; one 8-byte store, no copy command)
// keep_ne: the next round's "some interval is non-empty" flag is not cleared here (PEER transport of a sharded index: the other ranks'
// k_advance set it, and they may run ahead of this rank's k_mround; the host clears it before the round's counting phase instead)
template <bool SPARSE> __device__ __forceinline__ void setup_body(Ctl *ctl, int side, const uint64_t *gcnt /* global or LDS */, int par, uint32_t round, volatile unsigned long long *hmax, bool keep_ne)
{
	const int r = lane_id();
	if (r == 0) { if (!keep_ne) ctl->ne[par ^ 1] = 0; ctl->overflow = 0; ctl->sbfull = 0; ctl->nsplit2[round & 1u] = 0; }   // ne: k_advance / k_munpack of this round count into it
	if (r < WLC) ctl->wcnt[r * WLS] = 0;                       // the work lists of a sparse round
	if (r == 0) ctl->wstride = max(1u, (ctl->seg[side].tile0[NR] + WLC - 1) / WLC) * STILE;   // wstride / STILE consecutive tiles share a list, a tile appends at most STILE orders
	const bool ok = r < NR;
	const int rr = ok ? r : 0;
	const SegDesc &sg = ctl->seg[side];
	SegDesc &ng = ctl->seg[side ^ 1];
	const RopeDesc o = ctl->rope[side][rr];
	// new sub-ropes (side^1): sizes, layout in the leaf pool, output windows.  Pieces held by other ranks
	// keep n = 0 here but their symbol counts are tracked (needed for AC and for n0).
	RopeDesc n;
	n.n = ok ? o.n + sg.cnt[rr] : 0;
	if (hmax) {
		unsigned long long mx = n.n;
#pragma unroll
		for (int dd = 32; dd >= 1; dd >>= 1) { const unsigned long long t = __shfl_xor(mx, dd); mx = t > mx ? t : mx; }
		if (r == 0) *hmax = (unsigned long long)round << 40 | (mx < (1ull << 40) ? mx : (1ull << 40) - 1ull);
	}
	const int first = rope_of(rope_sym(rr), 0);               // first piece of my rope
	for (int a = 0; a < 6; ++a) {
		const uint64_t c = ok ? gcnt[rr * 6 + a] : 0ull;
		n.cnt[a] = ok ? o.cnt[a] + c : 0ull;
		const uint64_t inc = dpp_incl_add64(n.cnt[a]);
		const uint64_t ex = inc - n.cnt[a];
		const uint64_t exf = __shfl(ex, first);
		if (ok) { ctl->count[r][a] = c; ctl->ac[r][a] = ex - exf; }   // #a in this rope in front of piece r, after the round (mrope.c:332-336)
	}
	if (SPARSE) {
		n.nleaves = o.nleaves; n.leaf0 = o.leaf0; n.sb0 = o.sb0;
		if (ok) ctl->rope[side ^ 1][r] = n;
	} else {
		n.nleaves = (n.n + LEAF - 1) / LEAF;
		const uint64_t padded = (n.nleaves + SB - 1) / SB * SB, nwin = (n.nleaves + WPL - 1) / WPL;
		const uint64_t pinc = dpp_incl_add64(padded), winc = dpp_incl_add64(nwin);
		n.leaf0 = pinc - padded; n.sb0 = n.leaf0 / SB;
		if (ok) { ctl->rope[side ^ 1][r] = n; ctl->wf0[r] = winc - nwin; }
		if (r == 63) { ctl->wf0[NR] = winc; ctl->wf0[NR + 1] = winc; ctl->nsb_total = pinc / SB; }
	}
	// next round's buckets: bucket (a,b) = strings that sat in a piece of rope b and inserted a, in
	// (piece, order) order -- the stable scatter of mrope.c:303-309; only buckets of pieces held here
	// are laid out locally; strings that inserted $ are dropped (mrope.c:310)
	uint64_t c2 = 0;
	if (ok && r != 0 && ctl->own[r]) {
		const int a = rope_sym(r), b = rope_prev(r);
		if (b == 0) c2 = gcnt[a];
		else for (int x = 0; x < 6; ++x) c2 += gcnt[rope_of(b, x) * 6 + a];
	}
	const uint64_t cinc = dpp_incl_add64(c2), st = cinc - c2;
	const uint32_t nt = (uint32_t)((c2 + STILE - 1) / STILE);
	const uint32_t tinc = dpp_incl_add(nt);
	if (ok) { ng.start[r] = st; ng.cnt[r] = c2; ng.tile0[r] = tinc - nt; }
	if (r == 63) { ng.tile0[NR] = tinc; ng.tile0[NR + 1] = tinc; }
	// dest[r][a]: where the members of bucket r = (b,x) that insert a start inside bucket (a,b)
	for (int a = 1; a < 6; ++a) {
		const int r2 = rope_of(a, rope_sym(rr));
		const uint64_t st2 = __shfl(st, r2);
		uint64_t before = 0;
		if (rope_sym(rr) != 0) for (int x = 0; x < rope_prev(rr); ++x) before += gcnt[rope_of(rope_sym(rr), x) * 6 + a];
		if (ok) ctl->dest[r][a] = st2 + before;
	}
	if (ok) ctl->dest[r][0] = 0;
}
template <bool SPARSE> __global__ __launch_bounds__(64) void k_setup(Ctl *ctl, int side, const uint64_t *gcnt, int par, uint32_t round, volatile unsigned long long *hmax, int keep_ne)
{
	if (blockIdx.x) return;
	setup_body<SPARSE>(ctl, side, gcnt, par, round, hmax, keep_ne != 0);
}

// ---------------------------------------------------------------------------------------------
// k_tscan_setup: the whole counting tail of a round with few string tiles (long reads: 10^4 rounds of 10^3 tiles) in ONE single-block
// launch -- k_tscan1-3 + k_tfix (with the count matrix) + k_setup.  Same results, same formulas; what differs is where the numbers live: the
// exclusive prefixes of the six histogram columns stay in LDS (96 KB for 4096 tiles), the "nearest head-bearing tile" on either
// side comes from a bitmap of the tiles that have a head (highest set bit below / lowest above), and the count matrix reaches
// k_setup's wave through LDS.  Global memory is read twice in a row (the tile records; the records of the neighbouring head tiles)
// instead of eight times (round 3: 25.6 us for 2000 tiles + 7.9 us for k_setup behind it).
// do_setup = 0: a rank of a sharded index -- the count matrix is summed over the ranks before k_setup may run.
// spec: the launch was queued before the host saw the verdict of the in-place round in front of it; if that round was void
// (ctl->overflow) nothing may be overwritten -- the host redoes the round from its own counting phase.
// ---------------------------------------------------------------------------------------------
constexpr int TS_MAX = 4 * SCHUNK;          // string tiles the single-block path takes
constexpr int TFW = 26;                     // dwords of a TileFix
static_assert(sizeof(TileFix) == TFW * 4, "TileFix is written out as 26 dwords");
// More than one block (round 6).  The launch sat on ONE compute unit for 19.5 us of every long-read round.  Now TSB blocks each run the scan (the tile
// records are 160 KB, read from L2) and write a share of the TileFix records -- the part that took most of the time: three passes of 960 tiles for
// one block, one pass for four --; block 0 alone publishes the count matrix and runs k_setup.  (Tried: the leaf splits of the round before and the
// directory's chunk bases as further blocks of this launch instead of the k_sym launch's -- 1024-thread blocks cap the kernel at 128 VGPRs, the
// splits keep a superblock's 32 leaves in 64 of them: 215 spilled registers, the job 8 % slower.)
constexpr int TSB = 4;                      // blocks of k_tscan_setup
template <bool SPARSE> __global__ __launch_bounds__(SCHUNK) void k_tscan_setup(Ctl *ctl, int side, int par, const TileRecs trec, TileFix *tf, uint64_t *gcnt, int do_setup, int spec,
		uint32_t round, volatile unsigned long long *hmax, SbBase *scan2)
{
	__shared__ uint32_t s_pre[6][TS_MAX + 4];                    // exclusive prefix of hist over all tiles; [.][nt] = total
	__shared__ uint32_t s_out[SCHUNK / 64][32 * TFW];            // per wave: 32 TileFix records on their way out (coalesced stores)
	__shared__ unsigned long long s_head[TS_MAX / 64];           // bit t: tile t holds a group head
	__shared__ uint32_t s_p[6][16];
	__shared__ uint32_t s_t0[NR + 1];
	__shared__ uint64_t s_g[NR * 6];
	// behind an in-place round: one more block turns the chunk totals the k_advance launch left into the directory's chunk bases ("the directory rides
	// along", k_advance) -- here and not in the k_sym launch in front of this one, whose two thousand tile blocks its registers would hold back
	// (seven blocks, one column each -- the six symbols' counts and the positions, whose chunk totals k_sbscan3 leaves side by side --, 1024 threads x 5 chunks: one pass up to
	// 5120 chunks = 94 G symbols of the in-place layout.  One block that took the columns in turn needed a round trip to the freshly written totals and two barriers per
	// column and pass: 27-32 us of this launch at 90 G symbols)
	if (scan2 && blockIdx.x + 7 >= gridDim.x) { sbscan2_col<SCHUNK, 5>(ctl, scan2, (uint64_t*)s_out, (int)(blockIdx.x + 7 - gridDim.x)); return; }   // the last seven blocks: one column each
	if (spec && ctl->overflow) return;
	const SegDesc &sg = ctl->seg[side];
	const int ln = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	if (threadIdx.x <= NR) s_t0[threadIdx.x] = sg.tile0[threadIdx.x];
	if (threadIdx.x < TS_MAX / 64) s_head[threadIdx.x] = 0;
	// thread i owns tiles 4i .. 4i + 3: one 16-byte load per column
	const uint32_t nt_g = sg.tile0[NR];                         // (uniform scalar load; s_t0 is not visible yet)
	const uint32_t t_first = threadIdx.x * 4;
	uint32_t hs[4][6], tot[6] = {0, 0, 0, 0, 0, 0}, hb = 0;
	{
		const bool any = t_first < nt_g;
#pragma unroll
		for (int s = 0; s < 6; ++s) {
			uint4 v = make_uint4(0, 0, 0, 0);
			if (any) v = *(const uint4*)&trec.hist(s, t_first);
			hs[0][s] = v.x; hs[1][s] = v.y; hs[2][s] = v.z; hs[3][s] = v.w;
		}
		int4 f = make_int4(-1, -1, -1, -1);
		if (any) f = *(const int4*)&trec.fh(t_first);
		const int fv[4] = { f.x, f.y, f.z, f.w };
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool ok = t_first + k < nt_g;
			if (!ok) { for (int s = 0; s < 6; ++s) hs[k][s] = 0; }
			if (ok && fv[k] >= 0) hb |= 1u << k;
		}
	}
#pragma unroll
	for (int k = 0; k < 4; ++k)
#pragma unroll
		for (int s = 0; s < 6; ++s) { const uint32_t v = hs[k][s]; hs[k][s] = tot[s]; tot[s] += v; }   // exclusive inside the thread
	uint32_t inc[6];
#pragma unroll
	for (int s = 0; s < 6; ++s) { inc[s] = dpp_incl_add(tot[s]); if (ln == 63) s_p[s][wv] = inc[s]; }
	__syncthreads();
	if (hb) atomicOr(&s_head[t_first >> 6], (unsigned long long)hb << (t_first & 63));
#pragma unroll
	for (int s = 0; s < 6; ++s) {
		const uint32_t p = ln < 16 ? s_p[s][ln] : 0u;
		const uint32_t pin = dpp_incl_add(p);
		const uint32_t off = wv ? (uint32_t)__builtin_amdgcn_readlane((int)pin, wv - 1) : 0u;
		const uint32_t base = off + inc[s] - tot[s];
#pragma unroll
		for (int k = 0; k < 4; ++k) if (t_first + k <= nt_g) s_pre[s][t_first + k] = base + hs[k][s];
	}
	__syncthreads();
	const uint32_t nt = s_t0[NR];
	const bool lead = blockIdx.x == 0;                          // the block that publishes the count matrix and runs k_setup
	if (threadIdx.x < NR * 6) {                                 // the rows of the count matrix seen here (k_tfix's block 0 in the many-tiles path)
		const int b = threadIdx.x / 6, a = threadIdx.x % 6;
		const uint64_t v = nt ? (uint64_t)(s_pre[a][s_t0[b + 1]] - s_pre[a][s_t0[b]]) : 0ull;
		s_g[threadIdx.x] = v; if (lead) gcnt[threadIdx.x] = v;
	}
	if (lead && threadIdx.x == NR * 6) gcnt[NR * 6] = ctl->ne[par];     // (GCN, rb2_device.h)
	__syncthreads();
	// k_setup of the round (one GPU) runs on the LAST wave of block 0 while the others write the tile records: it needs the count matrix only,
	// and its ten 64-bit wave scans in a row were 4 us at the end of the kernel with fifteen waves waiting
	constexpr int NWV = SCHUNK / 64;
	const int tw0 = (do_setup ? NWV - 1 : NWV);                 // waves of block 0 that write tile records
	if (lead && do_setup && wv == NWV - 1) { setup_body<SPARSE>(ctl, side, s_g, par, round, hmax); return; }
	const uint32_t nscanb = gridDim.x - (scan2 ? 7u : 0u);      // (a launch of one block: everything here)
	const uint32_t wid = lead ? (uint32_t)wv : (uint32_t)tw0 + (blockIdx.x - 1u) * NWV + (uint32_t)wv, tw = (uint32_t)tw0 + (nscanb - 1u) * NWV;   // my number among the writing waves / how many there are
	uint32_t *so = s_out[wv];
	for (uint32_t tb = wid * 64; tb < nt; tb += tw * 64) {      // k_tfix: a wave takes 64 consecutive tiles
		const uint32_t tile = tb + (uint32_t)ln;
		const bool live = tile < nt;
		uint32_t f[TFW];
#pragma unroll
		for (int i = 0; i < TFW; ++i) f[i] = 0;
		if (live) {
			int b = 0;
			while (tile >= s_t0[b + 1]) ++b;
			const uint32_t t0 = s_t0[b], t1 = s_t0[b + 1];
			int lt = -1, nx = INT_MAX;
			{
				int w = (int)(tile >> 6);
				unsigned long long m = s_head[w] & lt_mask((int)(tile & 63));
				while (m == 0 && w > 0) m = s_head[--w];
				if (m) lt = w * 64 + 63 - __builtin_clzll(m);
				w = (int)(tile >> 6);
				const int wl = (int)((nt - 1) >> 6);
				m = (tile & 63) == 63 ? 0ull : s_head[w] & (~0ull << ((tile & 63) + 1));
				while (m == 0 && w < wl) m = s_head[++w];
				if (m) nx = w * 64 + __builtin_ctzll(m);
			}
			const bool hl = lt >= (int)t0, hn = nx < (int)t1;
			const uint32_t pn = hn ? (uint32_t)nx : t1;
#pragma unroll
			for (int s = 0; s < 6; ++s) {
				const uint32_t first = s_pre[s][t0];
				f[s] = s_pre[s][tile] - first;                                                        // tpre
				f[6 + s] = hl ? s_pre[s][lt] - first + trec.lhpre(s, (uint32_t)lt) : 0u;              // popen
				f[12 + s] = s_pre[s][pn] - first + (hn ? trec.fhpre(s, (uint32_t)nx) : 0u);           // pnext
			}
			f[18] = hl ? (uint32_t)((lt - (int)t0) * STILE + (trec.lh((uint32_t)lt) & (TILE_DONE - 1))) : 0u;          // fopen
			const int32_t mylh = trec.lh(tile);
			f[19] = (uint32_t)b; f[20] = tile - t0; f[21] = ((tile + 1 >= t1 || trec.fh(tile + 1) == 0) ? 1u : 0u) | ((mylh >= 0 && (mylh & TILE_DONE)) ? 2u : 0u) | ((mylh >= 0 && (mylh & TILE_SINGLE)) ? 4u : 0u);   // nexthead | done | single
			const uint64_t ss = sg.start[b], se = ss + sg.cnt[b];
			f[22] = (uint32_t)ss; f[23] = (uint32_t)(ss >> 32); f[24] = (uint32_t)se; f[25] = (uint32_t)(se >> 32);
		}
		// out through LDS, 32 records at a time: 13 store instructions of 64 consecutive dwords each instead of 26 that hit 64 lines each
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			if ((ln >> 5) == half) {
#pragma unroll
				for (int i = 0; i < TFW; ++i) so[(ln & 31) * TFW + i] = f[i];
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
			const uint32_t r0 = tb + 32u * half;                        // first record of this half
			const uint32_t nrec = r0 < nt ? min(32u, nt - r0) : 0u;
			uint32_t *dst = (uint32_t*)(tf + r0);
#pragma unroll
			for (int i = 0; i < 32 * TFW / 64; ++i) { const uint32_t d = (uint32_t)(i * 64 + ln); if (d < nrec * TFW) dst[d] = so[d]; }
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
		}
	}
}

// ---------------------------------------------------------------------------------------------
// k_prep
// ---------------------------------------------------------------------------------------------

// Who is where inside a bucket: shared by k_prep (which needs the slot of every new symbol) and
// k_advance (which recomputes the same numbers instead of reading them back from HBM).
struct GroupLds {
	uint64_t bal[8][6], head[8];                // per wave-chunk of the tile: lanes with symbol s / group heads
	uint32_t cpre[9][6];
	TileFix fix;                                // tpre, popen, pnext, fopen of this tile (k_tfix)
	uint32_t allsingle;                         // every string of the tile is a group of its own (the rule once intervals are narrow): group_member takes the short way
	uint32_t okc[8];                            // ... per wave-chunk: every valid lane is a head (a scalar comparison of two ballots, by the wave that has them)
};

// fills G for the string tile of this block; sym2[h] = symbol of string t.base + h*256 + threadIdx.x (7: none).
// araw[h] = the string's byte of A (symbol + flags), fixw = word threadIdx.x of the tile's TileFix: both LOADED BY THE CALLER together with
// its own loads of the tile (r04 loaded them in here, one after the other, each waited for at once: the tile kernels walked five
// dependent round trips to memory -- tile record, strings, A of the first string, A of the second, gathers -- where three do).
__device__ __forceinline__ uint32_t tilefix_word(const TileFix *tf, uint32_t tile) { return threadIdx.x < TILEFIX_LDS_WORDS ? ((const uint32_t*)&tf[tile])[threadIdx.x] : 0u; }
__device__ __forceinline__ void group_setup(GroupLds &G, const TileCtx &t, const uint32_t araw[2], const uint32_t fixw, int sym2[2], int flag2[2])
{
	const int ln = lane_id(), w = wave_id();
	if (threadIdx.x < TILEFIX_LDS_WORDS) ((uint32_t*)&G.fix)[threadIdx.x] = fixw;
	const uint32_t nv = (uint32_t)min((uint64_t)STILE, t.segend - t.base);
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		int sym = 7, fl = 0; bool head = false;
		if ((uint32_t)(h * 256) + threadIdx.x < nv) { const uint32_t a = araw[h]; sym = (int)(a & 7u); head = (a & 0x80u) != 0; fl = (int)(a & 0x40u); }
		sym2[h] = sym; flag2[h] = fl;
		const int c = h * 4 + w;
		// all ballots first, then ONE lane-0 block that stores them (a store per ballot was a branch per ballot: seven exec round trips)
		uint64_t bm[6];
#pragma unroll
		for (int s = 0; s < 6; ++s) bm[s] = ballot64(sym == s);
		const uint64_t hm = ballot64(head), vm = ballot64((uint32_t)(h * 256) + threadIdx.x < nv);
		if (ln == 0) {
#pragma unroll
			for (int s = 0; s < 6; ++s) G.bal[c][s] = bm[s];
			G.head[c] = hm; G.okc[c] = hm == vm ? 1u : 0u;
		}
	}
	__syncthreads();
	if (threadIdx.x < 6) {
		const int s = threadIdx.x;
		uint32_t run = 0;
		for (int c = 0; c < 8; ++c) { G.cpre[c][s] = run; run += __popcll(G.bal[c][s]); }
		G.cpre[8][s] = run;
	}
	if (threadIdx.x == 6) {                                     // all heads, and the string behind the tile starts a group too?
		uint32_t all = G.fix.nexthead & 1u;
#pragma unroll
		for (int c = 0; c < 8; ++c) all &= G.okc[c];
		G.allsingle = all;
	}
	__syncthreads();
}

// tile context from the TileFix record (one uniform load; the caller checks blockIdx.x against the tile count)
__device__ __forceinline__ void tile_ctx_fix(const TileFix &f, TileCtx &t)
{
	t.b = (int)f.b; t.lt = f.lt; t.segstart = f.segstart; t.segend = f.segend;
	t.base = t.segstart + t.lt * STILE;
}

struct Member { uint32_t pa, pga, slot, F; int lead; };   // lead: first member of my group inside this tile (F, like slot, is an index inside the bucket: a batch has < 2^32 strings)

// string x of the tile, inserting a: pa = members of the bucket in front of it inserting a, pga = the same count
// in front of its group, F = first member of its group, slot = its place in the bucket's insert list
// (group, symbol order, array order)
__device__ __forceinline__ Member group_member(const GroupLds &G, const TileCtx &t, int x, int a, const int orda[6])
{
	const int c = x >> 6, l6 = x & 63;
	auto before = [&](int y, int s) -> uint32_t { return G.cpre[y >> 6][s] + __popcll(G.bal[y >> 6][s] & lt_mask(y & 63)); };
	Member m;
	if (G.allsingle) {                                         // (block-uniform) a tile of one-member groups -- asked FIRST: behind the search for the group's ends below, every string of such a tile (the rule from round ~14 on) still searched
		m.pa = G.fix.tpre[a] + before(x, a);
		m.pga = m.pa; m.F = (uint32_t)t.lt * STILE + (uint32_t)x; m.slot = m.F; m.lead = x;
		return m;
	}
	const uint64_t le = lt_mask(l6) | (1ull << l6);
	int hpos = -1, npos = -1;
	{
		uint64_t hm = G.head[c] & le;
		if (hm) hpos = c * 64 + 63 - __builtin_clzll(hm);
		else for (int cc = c - 1; cc >= 0; --cc) if (G.head[cc]) { hpos = cc * 64 + 63 - __builtin_clzll(G.head[cc]); break; }
		uint64_t nm = G.head[c] & ~le;
		if (nm) npos = c * 64 + __builtin_ctzll(nm);
		else for (int cc = c + 1; cc < 8; ++cc) if (G.head[cc]) { npos = cc * 64 + __builtin_ctzll(G.head[cc]); break; }
	}
	m.pa = G.fix.tpre[a] + before(x, a);
	if (hpos == x && npos == x + 1) {                      // a group of one (the common case once intervals are narrow)
		m.pga = m.pa; m.F = (uint32_t)t.lt * STILE + (uint32_t)x; m.slot = m.F; m.lead = x;
		return m;
	}
	m.pga = hpos >= 0 ? G.fix.tpre[a] + before(hpos, a) : G.fix.popen[a];
	m.F = hpos >= 0 ? (uint32_t)t.lt * STILE + (uint32_t)hpos : G.fix.fopen;
	uint32_t bef = 0;                                      // members of my group inserting a smaller symbol
	const int oa = orda[a];
	for (int s = 0; s < 6; ++s) {
		if (orda[s] >= oa) continue;
		const uint32_t pg = hpos >= 0 ? G.fix.tpre[s] + before(hpos, s) : G.fix.popen[s];
		const uint32_t pn = npos >= 0 ? G.fix.tpre[s] + before(npos, s) : G.fix.pnext[s];
		bef += pn - pg;
	}
	m.slot = m.F + bef + (m.pa - m.pga);
	m.lead = hpos >= 0 ? hpos : 0;
	return m;
}

template <bool AE, bool SPARSE, typename P> __device__ __forceinline__ bool prep_tile(const uint32_t tile, const Ctl *ctl, int side, int par, int is_comp, const PoolView &oldp,
		const P *L, const P *U, uint8_t *A, const TileFix *tf,
		P *INS_E, uint8_t *INS_A, P *SIZE);                              // false: nothing (more) to do for this block
constexpr int PREP_PT = 8;                  // string tiles per block of k_prep<AE> on one engine

template <bool AE, bool SPARSE = false, bool STRIDE = false, typename P = uint64_t> __global__ __launch_bounds__(256) void k_prep(const Ctl *ctl, int side, int par, int is_comp, PoolView oldp,
		const P *L, const P *U, uint8_t *A, const TileFix *tf,
		P *INS_E, uint8_t *INS_A, P *SIZE)
{
	if (AE && !STRIDE) {
		// all-empty rounds, one engine: k_sym has placed the new symbols of nearly every tile itself (TILE_DONE), and a launch of one block
		// per tile was 22 us of blocks that read one word and left (82 K of them, 40 turns of the chip).  A block takes PREP_PT consecutive
		// tiles: one vector load says which of them are still to do (none, as a rule); those go one after the other.
		const uint32_t t0 = xcd_item() * PREP_PT, nt = ctl->seg[side].tile0[NR];
		if (ctl->ne[par] != 0) return;
		if (SPARSE && ctl->overflow) return;                        // (queued behind a void in-place round: its tile records are not this round's -- k_part_sparse, "a void round is sticky")
		const uint32_t ln = (uint32_t)lane_id();
		const bool mine = ln < (uint32_t)PREP_PT && t0 + ln < nt;
		uint32_t nh = 2u;
		if (mine) nh = tf[t0 + ln].nexthead;
		uint64_t todo = __ballot(mine && !(nh & 2u));               // (the same in all four waves)
		while (todo) {
			const uint32_t k = (uint32_t)__builtin_ctzll(todo);
			todo &= todo - 1;
			prep_tile<AE, SPARSE, P>(t0 + k, ctl, side, par, is_comp, oldp, L, U, A, tf, INS_E, INS_A, SIZE);
			if (todo) __syncthreads();
		}
		return;
	}
	// the first tile exactly as a one-tile-per-block kernel would run it (its loads are issued before anything is waited for);
	// further tiles only when the grid is smaller than the number of tiles (grid stride: see k_sym)
	for (uint32_t tile = STRIDE ? blockIdx.x : xcd_item(); ; ) {
		if (!prep_tile<AE, SPARSE, P>(tile, ctl, side, par, is_comp, oldp, L, U, A, tf, INS_E, INS_A, SIZE)) return;
		if (!STRIDE) return;
		tile += gridDim.x;
		if (tile >= ctl->seg[side].tile0[NR]) return;
		__syncthreads();
	}
}

template <bool AE, bool SPARSE, typename P> __device__ __forceinline__ bool prep_tile(const uint32_t tile, const Ctl *ctl, int side, int par, int is_comp, const PoolView &oldp,
		const P *L, const P *U, uint8_t *A, const TileFix *tf,
		P *INS_E, uint8_t *INS_A, P *SIZE)
{
	__shared__ GroupLds G;
	const TileFix &tfx = tf[tile];                              // issued together with the mode and tile-count loads
	const SegDesc &sg = ctl->seg[side];
	if ((ctl->ne[par] == 0) != AE) return false;
	if (SPARSE && ctl->overflow) return false;                 // (queued behind a void in-place round: its tile records are not this round's)
	if (tile >= sg.tile0[NR]) return false;
	if (AE && (tfx.nexthead & 2u)) return true;                // k_sym placed this tile's new symbols itself (all-single tile)
	TileCtx t;
	tile_ctx_fix(tfx, t);
	int sym2[2], flag2[2];
	P l2[2], u2[2];                                            // issued before the barriers of group_setup (in the batch's storage width: half the registers while positions fit 32 bits)
	uint32_t araw[2];
	const uint32_t fixw = tilefix_word(tf, tile);
	const uint32_t nval = (uint32_t)min((uint64_t)STILE, t.segend - t.base);   // strings in this tile; every access below: uniform base + 32-bit offset
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint32_t x = (uint32_t)(h * 256) + threadIdx.x;
		l2[h] = u2[h] = 0; araw[h] = 7;
		if (x < nval) { l2[h] = (L + t.base)[x]; u2[h] = AE ? l2[h] : (U + t.base)[x]; araw[h] = (A + t.base)[x]; }
	}
	group_setup(G, t, araw, fixw, sym2, flag2);
	const RopeDesc &rp = ctl->rope[side][t.b];
	const int orda[6] = { sym_ord(0, is_comp), sym_ord(1, is_comp), sym_ord(2, is_comp), sym_ord(3, is_comp), sym_ord(4, is_comp), sym_ord(5, is_comp) };
	if (AE) {
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int x = h * 256 + threadIdx.x;
			if ((uint32_t)x >= nval) continue;
			const int a = sym2[h];
			const Member m = group_member(G, t, x, a, orda);
			(INS_E + t.segstart)[m.slot] = (P)(l2[h] - (P)m.F);   // empty interval: the new symbol goes to l (pre-round coordinates)
			(INS_A + t.segstart)[m.slot] = (uint8_t)a;
		}
		return true;
	}
	// Non-empty intervals: the members of a group share [l, u) (mrope.c:192-202), so rope_rank2a is evaluated once per
	// group and tile -- by the group's first member inside the tile -- and handed to the others through LDS.
	// The lead threads only POST their query (l, u) in the group's LDS slot and append the slot to one of two lists; then
	//   - short intervals inside one leaf (the bulk: a handful of rows) are counted by consecutive threads, one query each,
	//     scanning only the interval (range_counts) -- the leads are scattered over all waves of the tile, the list is dense,
	//     so one or two waves run the scan instead of all of them;
	//   - long intervals and those that span leaves are served by whole WAVES (wave_range_counts: one coalesced 512-byte load
	//     per leaf, bit-plane popcounts per lane, packed DPP reductions): their cost does not grow with the interval.
	__shared__ P s_d[AE ? 1 : STILE][6];                       // in: [0] = l, [1] = u; out: #s in [l, u) of the group led by string x of the tile (a count is at most u - l: it fits the positions' width)
	__shared__ uint16_t s_qs[AE ? 1 : STILE], s_qw[AE ? 1 : STILE];
	__shared__ uint32_t s_ns, s_nw;
	if (threadIdx.x == 0) { s_ns = 0; s_nw = 0; }
	__syncthreads();
	Member mm[2];
	P l0[2], u0[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int x = h * 256 + threadIdx.x;
		l0[h] = u0[h] = 0; mm[h].lead = -1;
		if ((uint32_t)x >= nval) continue;
		mm[h] = group_member(G, t, x, sym2[h], orda);
		l0[h] = (P)(l2[h] - (P)mm[h].F); u0[h] = (P)(u2[h] - (P)mm[h].F);   // coordinates on the pre-round rope
		if (mm[h].lead == x && u0[h] != l0[h]) {               // rope_rank2a (mrope.c:202)
			const bool small = u0[h] - l0[h] <= 3 * GSYM && (SPARSE || ((u0[h] - 1) >> LEAF_SH) == (l0[h] >> LEAF_SH));
			s_d[x][0] = l0[h]; s_d[x][1] = u0[h];
			if (small) s_qs[atomicAdd(&s_ns, 1u)] = (uint16_t)x; else s_qw[atomicAdd(&s_nw, 1u)] = (uint16_t)x;
		}
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < s_ns; i += 256) {
		const int x = s_qs[i];
		P d[6];
		range_counts<SPARSE, P>(oldp, rp, s_d[x][0], s_d[x][1], d);
		for (int s = 0; s < 6; ++s) s_d[x][s] = d[s];
	}
	for (uint32_t q = wave_id(); q < s_nw; q += 4) {           // wave-uniform loop
		const int x = s_qw[q];
		P d[6];
		wave_range_counts<SPARSE, P>(oldp, rp, s_d[x][0], s_d[x][1], d);
		__builtin_amdgcn_wave_barrier();
		if (lane_id() == 0) for (int s = 0; s < 6; ++s) s_d[x][s] = d[s];
	}
	__syncthreads();
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int x = h * 256 + threadIdx.x;
		if ((uint32_t)x >= nval) continue;
		const int a = sym2[h];
		P e = l0[h];
		if (u0[h] != l0[h]) {
			const int oa = orda[a];
			P size = 0;
			for (int s = 0; s < 6; ++s) {
				const P d = s_d[mm[h].lead][s];
				if (orda[s] < oa) e += d;
				if (s == a) size = d;
			}
			(SIZE + t.base)[x] = (P)size;                                 // only non-empty intervals have one (flag 0x40 in A)
			(A + t.base)[x] = (uint8_t)(a | 0x40 | (G.head[x >> 6] >> (x & 63) & 1 ? 0x80 : 0));
		}
		(INS_E + t.segstart)[mm[h].slot] = (P)e;
		(INS_A + t.segstart)[mm[h].slot] = (uint8_t)a;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// k_part: for every output window boundary o = j*WIN of sub-rope b, the number of inserts that land
// before it = smallest q with E[q] + q >= o (the final position of insert q is E[q] + q).  Two
// neighbouring boundaries make the work order of one output window (LeafDesc): everything k_merge
// needs, in one 32-byte record, so that the merge has no dependent descriptor loads.
// A block covers 256 boundaries = 255 windows (boundaries overlap by one between blocks).
// ---------------------------------------------------------------------------------------------

template <bool STRIDE, typename P = uint64_t> __global__ __launch_bounds__(256) void k_part(const Ctl *ctl, int side, const P *INS_E, LeafDesc *LD, const uint8_t *old_xh)
{
	__shared__ uint64_t s_wf0[NR + 1];
	__shared__ uint32_t s_q[256];
	if (threadIdx.x <= NR) s_wf0[threadIdx.x] = ctl->wf0[threadIdx.x];
	__syncthreads();
	for (uint64_t blk = blockIdx.x; blk * 255 < s_wf0[NR] + NR; blk += gridDim.x) {   // grid stride (a sharded rank launches fewer blocks than its upper bound)
	if (STRIDE && blk != blockIdx.x) __syncthreads();           // s_q of the previous chunk is done with
	const uint64_t gid = blk * 255 + threadIdx.x;               // boundary number: window (b,j) <-> wf0[b] + b + j
	const bool ok = gid < s_wf0[NR] + NR;
	int b = 0; uint64_t j = 0; uint32_t q = 0;
	if (ok) {
		while (gid >= s_wf0[b+1] + b + 1) ++b;
		j = gid - s_wf0[b] - b;
		const SegDesc &sg = ctl->seg[side];
		const P *E = INS_E + sg.start[b];
		const uint64_t o = j * WIN;
		uint64_t lo = 0, hi = sg.cnt[b];
		while (lo < hi) {
			const uint64_t mid = (lo + hi) >> 1;
			if (E[mid] + mid >= o) hi = mid; else lo = mid + 1;
		}
		q = (uint32_t)lo;
	}
	s_q[threadIdx.x] = q;
	__syncthreads();
	if (!ok || threadIdx.x == 255 || j >= s_wf0[b+1] - s_wf0[b]) { if (!STRIDE) return; continue; }   // the closing boundary of a piece is no window
	const RopeDesc &orp = ctl->rope[side][b], &nrp = ctl->rope[side ^ 1][b];
	const uint32_t q1 = s_q[threadIdx.x + 1];
	const uint64_t o0 = j * WIN, i0 = o0 - q;
	LeafDesc d;
	d.i0 = i0;
	d.ins0 = ctl->seg[side].start[b] + q;
	d.gl = nrp.leaf0 + j * WPL;
	d.oleaf0 = (uint32_t)orp.leaf0;
	// the formats of the (at most two) old windows the merge will draw from (rb2_merge.h "window formats"): one byte per window, written by the
	// merge that wrote the old side (old_xh == null: that side holds plain windows only -- a re-layout, the loader, or a round that was
	// not allowed to write compact ones); a window that does not exist reads as "compact, no exceptions" (nothing to fetch)
	const uint64_t ow = i0 >> 12, onw = (orp.nleaves + WPL - 1) / WPL;
	static_assert(WPL * LEAF == 4096, "a window is 4096 symbols");
	const uint8_t *xw = old_xh ? old_xh + orp.leaf0 / WPL : nullptr;
	const uint32_t h0 = ow < onw ? (xw ? (uint32_t)xw[ow] & 3u : 0u) : 1u;
	const uint32_t h1 = ow + 1 < onw ? (xw ? (uint32_t)xw[ow + 1] & 3u : 0u) : 1u;
	d.ni = (uint16_t)((uint32_t)(q1 - q) | h0 << 14);
	d.nvalid = (uint16_t)((uint32_t)min((uint64_t)WIN, nrp.n - o0) | h1 << 14);
	LD[s_wf0[b] + j] = d;
	if (!STRIDE) return;
	}
}

// ---------------------------------------------------------------------------------------------
// k_part_sparse: the same split seen from the inserts' side, for rounds that touch few leaves (sparse layout).  Every
// insert finds its leaf (locate() = the descent of rope_insert_run, rope.c:119-134); the first insert of a leaf -- its
// slot neighbour sits in another leaf -- gallops forward over E to count the leaf's inserts and appends one work order
// for k_merge_leaf.  Cost is proportional to the inserts, not to the index.  A leaf that cannot take its inserts
// (fill + ni > LEAF) voids the round: ctl->overflow, the host falls back to the dense rewrite.
// One block per string tile (slots and strings of a bucket share the index range).
// ---------------------------------------------------------------------------------------------

template <typename P> __device__ __forceinline__ bool part_sparse_tile(const uint32_t tile, Ctl *ctl, int side, const PoolView &oldp, const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, const TileFix *tf, SpOrd *LD, uint32_t *SPL, uint32_t spl_cap, P *RKOLD, uint32_t round);

// A VOID ROUND IS STICKY.  ctl->overflow = (the round that could not be done in place) + 1 stays set until the HOST has dealt with it: every
// kernel of an in-place round returns at once while it is set (this one, k_merge_leaf, k_advance<SPARSE>, the leaf splits; the counting
// phases queued behind it: k_tscan_setup / k_tfix with spec), so the device state stays what it was in front of the void round however
// many rounds the host has queued behind it.  That is what lets one engine queue in-place rounds WITHOUT reading a verdict per round
// (rounds 2-5: an event and a host round trip in every round, 14 us of an otherwise 230 us round with nothing on the device): the host
// polls the verdict word in pinned memory when it queues a round, and finds out at the latest at the end of the batch (insert_dev).
template <bool STRIDE, typename P = uint64_t> __global__ __launch_bounds__(256) void k_part_sparse(Ctl *ctl, int side, PoolView oldp, const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, const TileFix *tf, SpOrd *LD, uint32_t *SPL, uint32_t spl_cap, P *RKOLD, uint32_t round)
{
	if (ctl->overflow) return;                                  // (an earlier round is void -- or a block of this launch just found this one to be)
	for (uint32_t tile = blockIdx.x; ; ) {                      // (first tile as ever, then a grid stride: see k_prep)
		if (!part_sparse_tile<P>(tile, ctl, side, oldp, INS_E, INS_A, tf, LD, SPL, spl_cap, RKOLD, round)) return;
		if (!STRIDE) return;
		tile += gridDim.x;
		if (tile >= ctl->seg[side].tile0[NR]) return;
		__syncthreads();
	}
}

template <typename P> __device__ __forceinline__ bool part_sparse_tile(const uint32_t tile, Ctl *ctl, int side, const PoolView &oldp, const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, const TileFix *tf, SpOrd *LD, uint32_t *SPL, uint32_t spl_cap, P *RKOLD, uint32_t round)
{
	const TileFix &tfx = tf[tile];
	if (tile >= ctl->seg[side].tile0[NR]) return false;
	TileCtx t;
	tile_ctx_fix(tfx, t);
	const RopeDesc &rp = ctl->rope[side][t.b];
	const P *E = INS_E;
	__shared__ uint32_t s_w[4], s_base;
	// The descent (locate(), rb2_device.h) of the thread's TWO inserts, level by level, the loads of a level issued together: a thread
	// that ran one locate() after the other (and thread 0 a third one for the tile's left neighbour) had eight to twelve dependent
	// round trips to memory in a row.
	const uint64_t nsb = (rp.nleaves + SB - 1) / SB;
	const uint64_t base = nsb ? sb_pos(oldp, rp.sb0) : 0;
	// The rank of every insert's symbol on the rope AS IT IS (before the round) is taken here, on the way down -- the prefix in front of its
	// superblock sits in the record the descent reads anyway, the row of own counts in the directory block whose fills it reads -- and
	// handed to k_advance (RKOLD; k_merge_leaf adds the part inside the leaf: RKREL).  k_advance then needs nothing of the directory
	// after the merge: no leaf slot per symbol, no gathers behind it, and the prefix over the superblock totals (k_sbscan*) has until
	// the NEXT round's descent to be rebuilt (rope_insert_run returns the rank on its way down as well, rope.c:132-147).
	__shared__ P s_cb[6];                                       // symbol counts in front of the piece (pool-wide prefix of its first superblock)
	if (threadIdx.x < 6) s_cb[threadIdx.x] = nsb ? (P)sb_cum(oldp, rp.sb0, (int)threadIdx.x) : (P)0;
	bool ok[2];
	uint64_t p[2], pprev[2], sbi[2], sbs[2];
	uint32_t aq[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const uint64_t g = t.base + h * 256 + threadIdx.x;
		ok[h] = g < t.segend;
		p[h] = ok[h] ? (uint64_t)E[g] : 0;
		aq[h] = ok[h] ? (uint32_t)INS_A[g] & 7u : 0u;
		if (aq[h] > 5u) aq[h] = 0;
		pprev[h] = (ok[h] && g > t.segstart) ? (uint64_t)E[g - 1] : ~0ull;   // ~0: no insert in front of mine in this piece
	}
	// Level 1: the interpolated superblock.  ONE 16-byte load of its record says where it starts and how many symbols it holds (SbRec: cum[4],
	// cum[5], pos, tot), a 4-byte one my symbol's count in front of it.  Superblocks fill evenly: seven times in ten that is the superblock,
	// else the neighbour on the side the position lies (one more probe), else -- a piece that fills unevenly -- the search (locate()).
	// (Every one of these gathers costs what its sectors cost at the CU's address unit -- one 32-byte sector per clock, tools/ubench/gather_rate.hip --,
	// whatever it hits: r05 probed three records side by side and read both 64-byte rows whole, twelve such loads per insert; now seven to eight.)
	bool slow[2];
	uint64_t cum[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) { slow[h] = false; sbs[h] = 0; cum[h] = 0; sbi[h] = 0; }
	{
		uint4 hi[2]; uint32_t cg[2]; uint64_t bp[2], bc[2];
		// The guess, in two steps: the piece's superblocks as a whole, then -- with the chunk bases (SbBase: one 64-byte record per 1024 superblocks, a few
		// thousand of them, cache-resident, neighbouring lanes read the same ones) -- the part of the guessed chunk that belongs to the piece.  What a
		// straight line over the WHOLE piece misses grows with its length (inserts land at random: the position of superblock g strays from g / nsb of
		// the piece like sqrt(g)): at 90 G symbols, 160 k superblocks per piece, it was off by one more often than not and off by two often enough that
		// the fallback search took most of the kernel (38 -> 114 us per round from the first to the ninth batch of configs[3]).
		uint64_t gq[2], c0p[2], c1p[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			uint64_t g = (rp.n && nsb) ? (uint64_t)((double)p[h] / (double)rp.n * (double)nsb) : 0;
			if (nsb && g >= nsb) g = nsb - 1;
			gq[h] = g;
			const bool on = ok[h] && nsb;
			const uint64_t c = on ? (rp.sb0 + g) >> SCHUNK_SH : 0;
			c0p[h] = oldp.sbbase[c].pos; c1p[h] = oldp.sbbase[c + 1].pos;   // (the bases reach one entry past the last chunk in use: k_sbscan2)
		}
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			uint64_t g = gq[h];
			const bool on = ok[h] && nsb;
			if (on) {
				const uint64_t c = (rp.sb0 + g) >> SCHUNK_SH, pa = base + p[h];
				const uint64_t s_lo = max(c << SCHUNK_SH, rp.sb0), s_hi = min((c + 1) << SCHUNK_SH, rp.sb0 + nsb);   // the chunk's superblocks of this piece
				const uint64_t p_lo = (c << SCHUNK_SH) >= rp.sb0 ? c0p[h] : base, p_hi = ((c + 1) << SCHUNK_SH) <= rp.sb0 + nsb ? c1p[h] : base + rp.n;
				if (pa >= p_lo && pa < p_hi && p_hi > p_lo) {          // (else: the straight line's chunk was wrong -- keep its guess, the probes below sort it out)
					uint64_t g2 = s_lo + (uint64_t)((double)(pa - p_lo) / (double)(p_hi - p_lo) * (double)(s_hi - s_lo));
					if (g2 >= s_hi) g2 = s_hi - 1;
					g = g2 - rp.sb0;
				}
			}
			sbi[h] = g;
			const SbRec *rec = oldp.sbrec + (on ? rp.sb0 + g : 0);
			const SbBase *bb = oldp.sbbase + (on ? (rp.sb0 + g) >> SCHUNK_SH : 0);
			hi[h] = *(const uint4*)&rec->cum[4];
			cg[h] = rec->cum[aq[h] & 3u];
			bp[h] = bb->pos; bc[h] = bb->cum[aq[h]];
		}
		uint64_t g2[2]; bool again[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const bool on = ok[h] && nsb;
			const uint64_t P0 = bp[h] + hi[h].z - base;
			const bool in = P0 <= p[h] && (p[h] < P0 + hi[h].w || sbi[h] + 1 >= nsb);   // (the piece's last superblock takes everything behind it: p == n)
			again[h] = on && !in;
			g2[h] = p[h] < P0 ? sbi[h] - 1 : sbi[h] + 1;           // (in range: the first superblock starts at 0, the last one takes the rest)
			sbs[h] = P0;
			cum[h] = bc[h] + (aq[h] < 4u ? cg[h] : (aq[h] == 4u ? hi[h].x : hi[h].y));
		}
		if (again[0] || again[1]) {                                // the neighbour
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const SbRec *rec = oldp.sbrec + (again[h] ? rp.sb0 + g2[h] : 0);
				const SbBase *bb = oldp.sbbase + (again[h] ? (rp.sb0 + g2[h]) >> SCHUNK_SH : 0);
				if (again[h]) { hi[h] = *(const uint4*)&rec->cum[4]; cg[h] = rec->cum[aq[h] & 3u]; bp[h] = bb->pos; bc[h] = bb->cum[aq[h]]; }
			}
#pragma unroll
			for (int h = 0; h < 2; ++h) if (again[h]) {
				const uint64_t P0 = bp[h] + hi[h].z - base;
				const bool in = P0 <= p[h] && (p[h] < P0 + hi[h].w || g2[h] + 1 >= nsb);
				slow[h] = !in;
				sbi[h] = g2[h]; sbs[h] = P0;
				cum[h] = bc[h] + (aq[h] < 4u ? cg[h] : (aq[h] == 4u ? hi[h].x : hi[h].y));
			}
		}
	}
	// Level 2: the fills of the superblock's slots (row 0 of its directory block) and the own counts of my symbol (row 1 + a): the 48 bytes of the 24 slots a
	// re-layout fills, each; the reserve slots (leaf splits) only when the position lies behind the first 24
	Loc lc[2];
	uint4 fr[2][3], cr[2][3];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const bool ld = ok[h] && nsb && !slow[h];
		const uint4 *q = (const uint4*)dir_row(oldp, ld ? rp.sb0 + sbi[h] : 0, 0), *qc = (const uint4*)dir_row(oldp, ld ? rp.sb0 + sbi[h] : 0, 1 + (int)aq[h]);
#pragma unroll
		for (int i = 0; i < 3; ++i) { fr[h][i] = q[i]; cr[h][i] = qc[i]; }
	}
	static_assert(SP_USED == 24, "three 16-byte pieces of a directory row are the slots a re-layout fills");
	P rko[2] = { 0, 0 };                                      // my symbol in front of my leaf (without the piece's base: s_cb)
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		lc[h].gl = rp.leaf0; lc[h].s = 0; lc[h].n = 0; lc[h].p2 = 0;
		if (!ok[h] || !nsb) continue;
		if (slow[h]) {                                             // (rare: a piece that fills unevenly)
			lc[h] = locate(oldp, rp, p[h]);
			rko[h] = (P)(sb_cum(oldp, lc[h].gl / SB, (int)aq[h]) + dir_prefix(oldp, lc[h].gl / SB, 1 + (int)aq[h], (uint32_t)(lc[h].gl % SB)));
			continue;
		}
		const uint32_t rel = (uint32_t)(p[h] - sbs[h]);
		uint32_t run = 0, klo = 0, pre = 0, nk = 0;
		auto scan8 = [&](const uint4 &v, int i) {
			const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int j = 0; j < 8; ++j) {
				const uint32_t e = (j & 1) ? w[j >> 1] >> 16 : w[j >> 1] & 0xffffu, n = e & FILL_MASK;   // (bit 15: the leaf has a plane-2 line)
				if (n > 0 && run <= rel) { klo = (uint32_t)(8 * i + j); pre = run; nk = e; }
				run += n;
			}
		};
#pragma unroll
		for (int i = 0; i < 3; ++i) scan8(fr[h][i], i);
		uint32_t acc = 0;                                          // own counts of the slots in front of klo: two 16-bit sums side by side (dir_prefix)
		auto add8 = [&](const uint4 &v, int i) {
			const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t s0 = (uint32_t)(8 * i + 2 * j);
				acc += w[j] & (klo > s0 + 1 ? 0xffffffffu : (klo == s0 + 1 ? 0xffffu : 0u));
			}
		};
		if (run <= rel) {                                          // behind the 24 slots a re-layout fills: the reserve slots
			scan8(((const uint4*)dir_row(oldp, rp.sb0 + sbi[h], 0))[3], 3);
			if (klo > 24u) add8(((const uint4*)dir_row(oldp, rp.sb0 + sbi[h], 1 + (int)aq[h]))[3], 3);
		}
#pragma unroll
		for (int i = 0; i < 3; ++i) add8(cr[h][i], i);
		lc[h].gl = (rp.sb0 + sbi[h]) * SB + klo; lc[h].s = sbs[h] + pre; lc[h].n = nk & FILL_MASK; lc[h].p2 = nk >> 15;
		rko[h] = (P)(cum[h] + (acc & 0xffffu) + (acc >> 16));
	}
	// first insert of its leaf: the insert in front of mine (ascending positions) lies in front of my leaf's first symbol -- a leaf is
	// the LAST one in use that starts at or before the position, so two inserts share it exactly when the earlier one is not in front of it
	bool head[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) head[h] = ok[h] && (pprev[h] == ~0ull || pprev[h] < lc[h].s);
	// one slot in the work list per head: block-aggregated, one atomic per tile
	uint32_t tot;
	const uint32_t mine = (uint32_t)head[0] + (uint32_t)head[1];
	uint32_t off = block_excl_add<uint32_t>(mine, s_w, &tot);
	if (threadIdx.x == 0) { const uint32_t ws = ctl->wstride, c = tile / (ws / STILE); s_base = c * ws + (tot ? atomicAdd(&ctl->wcnt[c * WLS], tot) : 0u); }   // my list (Ctl::wcnt)
	__syncthreads();
	off += s_base;
#pragma unroll
	for (int h = 0; h < 2; ++h) if (ok[h]) RKOLD[t.base + h * 256 + threadIdx.x] = (P)(rko[h] - s_cb[aq[h]]);
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int x = h * 256 + threadIdx.x;
		const uint64_t g = t.base + x;
		if (!head[h]) continue;
		const uint64_t send = lc[h].s + lc[h].n;                // piece position one past the leaf
		uint64_t q1 = t.segend;
		if (send < rp.n) {                                     // (the last leaf in use takes everything up to the end)
			uint64_t step = 1, lo = g + 1;                     // gallop: first q > g with E[q] >= send
			while (lo + step - 1 < t.segend && E[lo + step - 1] < send) { lo += step; step <<= 1; }
			uint64_t hi = min(lo + step - 1, t.segend);        // E[hi] >= send or hi == segend; everything below lo is < send
			while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (E[mid] >= send) hi = mid; else lo = mid + 1; }
			q1 = lo;
		}
		const uint64_t ni = q1 - g;
		SpOrd d;
		d.i0 = (uint32_t)lc[h].s; d.ins0 = (uint32_t)g; d.gl = (uint32_t)lc[h].gl;
		if (ni == 1) d.i0 = (uint32_t)(p[h] - lc[h].s) | aq[h] << 12;   // the one insert of the leaf rides in its order (nineteen leaves in twenty of a long-read round): place inside the leaf + symbol, no gathers in k_merge_leaf
		d.ni = (uint16_t)min(ni, (uint64_t)LEAF); d.nvalid = (uint16_t)(min(lc[h].n + ni, (uint64_t)LEAF) | (lc[h].p2 ? FILL_P2 : 0u));
		if (lc[h].n + ni > (uint64_t)LEAF) ctl->overflow = round + 1u;  // the leaf cannot take them: void round (every block that finds one writes the same value)
		else if (lc[h].n + ni > (uint64_t)(LEAF - SP_MARGIN)) {  // close to full after this round: k_split gives it a second slot (rare: one atomic each)
			const uint32_t e = atomicAdd(&ctl->nsplit2[round & 1u], 1u);
			if (e < spl_cap) SPL[e] = (uint32_t)lc[h].gl;
		}
		LD[off++] = d;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// re-layout: copy every piece from one pool side to the other with a new slack policy.  Logical leaf t of a piece holds
// its symbols [t*F, (t+1)*F); a superblock uses its first K slots.  F = LEAF, K = SB is the dense layout (flat array, no
// slack: what k_merge streams); F = SP_FILL, K = SP_USED the sparse one.  Not on the per-round path: it runs when the
// engine changes between the two regimes (touched leaves << leaves, or back) and when slack runs out.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_relayout_setup(Ctl *ctl, int side, uint32_t F, uint32_t K)
{
	if (blockIdx.x) return;
	const int r = threadIdx.x;
	const bool ok = r < NR;
	RopeDesc o = ctl->rope[side][ok ? r : 0];
	if (ok) ctl->relay_old[r] = o;
	const bool dense = F == (uint32_t)LEAF && K == (uint32_t)SB;
	uint64_t slots = 0;
	if (ok) {
		if (dense) { o.nleaves = (o.n + LEAF - 1) / LEAF; slots = (o.nleaves + SB - 1) / SB * SB; }
		else { const uint64_t per = (uint64_t)F * K, nsb = max((uint64_t)1, (o.n + per - 1) / per); slots = nsb * SB; o.nleaves = slots; }
	}
	const uint64_t inc = wave_incl_add<uint64_t>(slots);
	o.leaf0 = inc - slots; o.sb0 = o.leaf0 / SB;
	if (ok) ctl->rope[side][r] = o;
	if (r == 63) ctl->nsb_total = inc / SB;
}

// one wave per output leaf slot: gathers its symbols from the (one to three) old leaves that hold them -- lane = old group, its bits
// ORed into place with LDS atomics, plane by plane --, writes the leaf and its own counts.  Slots that stay empty get zeroed counts.
__global__ __launch_bounds__(256) void k_relayout(const Ctl *ctl, int side, PoolView oldp, PoolView newp, uint32_t F, uint32_t K, int old_sparse)
{
	__shared__ __align__(16) uint64_t lds[MW][3 * (LEAFG + 1)];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint64_t *LX = lds[wv];                                     // plane pl of output group g at pl * (LEAFG + 1) + g (+ one spill word per plane)
	const int ln = lane_id();
	// grid-stride over the output slots: a launch is capped at 2^32 threads, i.e. 2^26 one-wave slots
	// (found at full configs[3] size: the capped launch was refused and the re-layout silently did nothing)
	for (uint64_t gl = (uint64_t)blockIdx.x * MW + wv; gl < ctl->nsb_total * SB; gl += (uint64_t)gridDim.x * MW) {
	// the piece that owns the slot: last one with leaf0 <= gl among those that have slots
	const uint64_t l0 = ctl->rope[side][ln < NR ? ln : NR - 1].leaf0;
	const int r = max(0, (int)__popcll(__ballot(ln < NR && l0 <= gl)) - 1);
	const RopeDesc &nrp = ctl->rope[side][r], &orp = ctl->relay_old[r];
	const uint64_t rel = gl - nrp.leaf0, sbi = rel / SB, k = rel % SB;
	const uint64_t t = sbi * K + k, p0 = t * F;                 // logical leaf, its first symbol
	const bool used = k < K && p0 < nrp.n;
	const uint32_t nvalid = used ? (uint32_t)min((uint64_t)F, nrp.n - p0) : 0u;
	if (ln < 3 * (LEAFG + 1)) LX[ln] = 0;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	if (nvalid) {
		Loc lc = old_sparse ? locate(oldp, orp, p0) : locate_dense(orp, p0);   // wave-uniform
		uint32_t off = (uint32_t)(p0 - lc.s), taken = 0;
		while (taken < nvalid) {
			const uint32_t take = min(lc.n - off, nvalid - taken);
			if (take && ln < LEAFG) {
				const uint64_t *lw = leaf_words(oldp.data, lc.gl) + ln;
				const uint32_t a = max(off, (uint32_t)(ln * GSYM)), b = min(off + take, (uint32_t)((ln + 1) * GSYM));   // my symbols [a, b) of the old leaf
				if (a < b) {
					const uint32_t op = taken + (a - off), og = op >> 6, sh = op & 63;
					const uint64_t m = bits_below(b - a);
					uint64_t ow[3];
					ow[0] = lw[0]; ow[1] = lw[LEAFG]; ow[2] = lc.p2 ? lw[2 * LEAFG] : ~(ow[0] | ow[1]);   // (a two-plane leaf of the sparse layout: rb2_device.h)
#pragma unroll
					for (int pl = 0; pl < 3; ++pl) {
						const uint64_t bits = (ow[pl] >> (a & 63)) & m;
						const uint64_t lo = bits << sh, hi = sh ? bits >> (64 - sh) : 0ull;
						if (lo) atomicOr((unsigned long long*)&LX[pl * (LEAFG + 1) + og], (unsigned long long)lo);
						if (hi) atomicOr((unsigned long long*)&LX[pl * (LEAFG + 1) + og + 1], (unsigned long long)hi);
					}
				}
			}
			taken += take; off = 0;
			if (taken < nvalid) {                                // next old leaf in use: the next slot, or the first of the next superblock
				uint64_t g2 = lc.gl + 1;
				uint32_t n2 = 0;
				if (!old_sparse) n2 = g2 < orp.leaf0 + orp.nleaves ? (uint32_t)min((uint64_t)LEAF, orp.n - (g2 - orp.leaf0) * LEAF) : 0u;
				else {
					if (g2 % SB != 0) n2 = dir_row(oldp, g2 / SB, 0)[g2 % SB];
					if ((n2 & FILL_MASK) == 0) { g2 = (lc.gl / SB + 1) * SB; n2 = g2 < orp.leaf0 + orp.nleaves ? dir_row(oldp, g2 / SB, 0)[0] : 0u; }
					lc.p2 = n2 >> 15; n2 &= FILL_MASK;
				}
				lc.gl = g2; lc.n = n2;
				if (n2 == 0) break;                              // cannot happen on a consistent directory
			}
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	PlAcc A;
	if (ln < LEAFG) {
		const uint64_t b0 = LX[ln], b1 = LX[(LEAFG + 1) + ln], b2 = LX[2 * (LEAFG + 1) + ln];
		const int v = min(GSYM, max(0, (int)nvalid - ln * GSYM));
		pl_acc(A, b0, b1, b2, bits_below((uint32_t)v));
		uint64_t *dst = (uint64_t*)newp.data + gl * LEAFW + ln;
		dst[0] = b0; dst[LEAFG] = b1; dst[2 * LEAFG] = b2;
	}
	const uint32_t r0 = wave_sum<uint32_t>(A.p0 | A.p1 << 16), r1 = wave_sum<uint32_t>(A.p2 | A.p01 << 16), r2 = wave_sum<uint32_t>(A.p02);
	if (ln == 0) {
		PlAcc T;
		T.p0 = r0 & 0xffffu; T.p1 = r0 >> 16; T.p2 = r1 & 0xffffu; T.p01 = r1 >> 16; T.p02 = r2;
		uint32_t c[6];
		pl_finish(T, nvalid, c);
		LeafMeta m;
		for (int sy = 0; sy < 6; ++sy) m.c[sy] = (uint16_t)c[sy];
		m.npre = 0; m.n = (uint16_t)nvalid;
		newp.own[gl] = m;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();   // LX is reused by the next slot
	}
}


// ---------------------------------------------------------------------------------------------
// k_split: the leaf split of a B+ tree (split_node, rope.c:78-112: "move the upper half of a full leaf to a new sibling, shift the
// parent's entries"), for the sparse layout.  Runs as the LAST kernel of an in-place round -- nothing else touches the pool then,
// and every number the round handed out (RKOLD, RKREL, the work orders) has been consumed.  k_part_sparse listed the leaves
// whose fill exceeds LEAF - SP_MARGIN; one wave per listed leaf, but only the first wave to CLAIM the leaf's superblock acts (an
// atomic exchange of this round's epoch into the spare directory row 7), and does all of that superblock's splits: slots behind a split leaf move up by one (from the last one down), the split leaf
// keeps its first h symbols (h = a whole number of words, about half) and hands the rest to the slot behind it, and the directory
// rows (fills + own counts) are rewritten to match.  The superblock's total does not change: sbcum / sbpos stay valid.
// A marked leaf in a superblock without a free slot sets ctl->sbfull: the host re-spreads the index before the next round.
// ---------------------------------------------------------------------------------------------
// The verdict of the round -- hv[0]: void round, hv[1]: a superblock ran out of slots -- goes straight into pinned host memory (the host
// zeroes both words before it queues the round): no copy command behind the last kernel.
// (the body: k_split proper, and the last blocks of the k_sym launch that follows an in-place round -- k_sym<.., SPLIT> -- run it: bidx of nblk)
__device__ __forceinline__ void split_body(Ctl *ctl, const PoolView &pool, const uint32_t *SPL, uint32_t spl_cap, uint32_t epoch /* != 0, never repeats */, volatile uint32_t *hv,
		const uint32_t bidx, const uint32_t nblk, uint16_t (*s_row)[7][SB], uint32_t round1)
{
	const uint32_t tid = threadIdx.x & 255u;                    // (a 1024-thread block of k_tscan_setup is four of these "blocks": bidx says which)
	if (bidx == 0 && tid == 0) hv[2] = round1;                  // how far the device has come (the host stays a few rounds ahead of this: insert_dev)
	if (ctl->overflow) { if (bidx == 0 && tid == 0) hv[0] = ctl->overflow; return; }   // void round (this one or one in front of it): nothing was inserted; the host learns which
	const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int ln = lane_id();
	const uint32_t nsp_raw = ctl->nsplit2[(round1 - 1u) & 1u];  // (the counter of the round whose splits these are)
	const uint32_t nsp_all = min(nsp_raw, spl_cap);
	if (nsp_raw > spl_cap && bidx == 0 && tid == 0) { ctl->sbfull = 1; hv[1] = 1; }   // list overflow (never in practice): re-spread
	for (uint32_t e = bidx * MW + wv; e < nsp_all; e += nblk * MW) {
		const uint64_t gl = SPL[e], sb = gl / SB;
		uint32_t mine = 0;
		if (ln == 0) mine = atomicExch((uint32_t*)dir_row(pool, sb, 7), epoch) != epoch;
		if (!__shfl((int)mine, 0)) continue;                     // a wave that came earlier does (or did) this superblock
		uint16_t v[7];
#pragma unroll
		for (int r = 0; r < 7; ++r) v[r] = ln < SB ? dir_row(pool, sb, r)[ln] : (uint16_t)0;
		const uint32_t used = (uint32_t)__popcll(__ballot(ln < SB && (v[0] & FILL_MASK) > 0));
		uint32_t marked = (uint32_t)__ballot(ln < SB && (v[0] & FILL_MASK) > (uint32_t)(LEAF - SP_MARGIN));
		if (marked == 0) continue;
		const uint32_t room = SB - used;
		if ((uint32_t)__popc(marked) > room) {
			if (ln == 0) { ctl->sbfull = 1; hv[1] = 1; }
			while ((uint32_t)__popc(marked) > room) marked &= ~(1u << (31 - __builtin_clz(marked)));   // split the lowest ones that fit
			if (marked == 0) continue;
		}
		const uint32_t first = (uint32_t)__builtin_ctz(marked);
		uint16_t (*R)[SB] = s_row[wv];
		// directory entries of the slots that only move (or stay): lane j = old slot j
		if (ln < (int)used && !((marked >> ln) & 1u)) {
			const uint32_t nj = (uint32_t)ln + (uint32_t)__popc(marked & ((1u << ln) - 1u));
#pragma unroll
			for (int r = 0; r < 7; ++r) R[r][nj] = v[r];
		}
		uint64_t *leaves = (uint64_t*)pool.data + sb * SB * LEAFW;
		// every leaf from the first split one on moves (or splits).  Eight slots at a time, FROM THE TOP DOWN: the eight are loaded back to back
		// (lane = word: plane ln >> 4 of group ln & 15), then stored at their new slots -- a slot moves up, never down, so nothing a later
		// (lower) eight still has to read is overwritten, and no two slots share a destination: at most four round trips to memory per
		// superblock (a wave that moved slot after slot spent 60 us on a full one).  (Rounds 4-5 loaded all 32 slots at once into 64 registers
		// per lane: the 99 VGPRs of these few blocks held the k_sym launch they ride in -- two thousand tile blocks of 16 VGPRs -- to five
		// workgroups per CU.)
		const bool wl = ln < LEAFW;                              // 48 words per leaf
		const uint32_t gq = (uint32_t)(ln & 15);                 // my group
		for (int c0 = SB - 8; c0 >= 0; c0 -= 8) {
		if ((uint32_t)(c0 + 8) <= first || (uint32_t)c0 >= used) continue;   // (wave-uniform)
		uint64_t W[8];
#pragma unroll
		for (int j = 0; j < 8; ++j) { const uint32_t k = (uint32_t)(c0 + j); W[j] = (wl && k >= first && k < used) ? leaves[(uint64_t)k * LEAFW + ln] : 0ull; }
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int k = c0 + j;
			if ((uint32_t)k < first || (uint32_t)k >= used) continue;   // wave-uniform
			const uint32_t nk = (uint32_t)k + (uint32_t)__popc(marked & ((1u << k) - 1u));
			const uint64_t w = W[j];
			if (!((marked >> k) & 1u)) { if (nk != (uint32_t)k && wl) leaves[(uint64_t)nk * LEAFW + ln] = w; continue; }
			const uint32_t nraw = (uint32_t)__builtin_amdgcn_readlane((int)v[0], k), n = nraw & FILL_MASK, p2f = nraw & FILL_P2;   // (both halves of a leaf with a plane-2 line keep one)
			uint32_t ck[6];
#pragma unroll
			for (int s = 0; s < 6; ++s) ck[s] = (uint32_t)__builtin_amdgcn_readlane((int)v[1 + s], k);
			const uint32_t hg = (n / 2) >> 6, h = hg << 6;           // the first hg groups stay
			// counts of the part that stays: the three planes of a group sit 16 lanes apart
			const uint64_t w1 = (uint64_t)__shfl((unsigned long long)w, (ln + 16) & 63), w2 = (uint64_t)__shfl((unsigned long long)w, (ln + 32) & 63);
			PlAcc A;
			pl_acc(A, w, w1, leaf_p2(p2f != 0, w, w1, w2), (ln < LEAFG && gq < hg) ? ~0ull : 0ull);
			const uint32_t r0 = lane63(dpp_incl_add(A.p0 | A.p1 << 16)), r1 = lane63(dpp_incl_add(A.p2 | A.p01 << 16)), r2 = lane63(dpp_incl_add(A.p02));
			PlAcc T;
			T.p0 = r0 & 0xffffu; T.p1 = r0 >> 16; T.p2 = r1 & 0xffffu; T.p01 = r1 >> 16; T.p02 = r2;
			uint32_t c1[6];
			pl_finish(T, h, c1);
			const uint64_t up = (uint64_t)__shfl((unsigned long long)w, (ln + (int)hg) & 63);   // group g + hg of my plane
			if (wl) {
				leaves[(uint64_t)(nk + 1) * LEAFW + ln] = gq + hg < (uint32_t)LEAFG ? up : 0ull;
				leaves[(uint64_t)nk * LEAFW + ln] = gq < hg ? w : 0ull;
			}
			if (ln == 0) {
				R[0][nk] = (uint16_t)(h | p2f); R[0][nk + 1] = (uint16_t)((n - h) | p2f);
#pragma unroll
				for (int s = 0; s < 6; ++s) { R[1 + s][nk] = (uint16_t)c1[s]; R[1 + s][nk + 1] = (uint16_t)(ck[s] - c1[s]); }
			}
		}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
		const uint32_t nused = used + (uint32_t)__popc(marked);
		if (ln == 0) atomicAdd((unsigned long long*)&ctl->nsplit_total, (unsigned long long)__popc(marked));
		if ((uint32_t)ln >= first && (uint32_t)ln < nused) {
#pragma unroll
			for (int r = 0; r < 7; ++r) dir_row(pool, sb, r)[ln] = R[r][ln];
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();   // R is reused by the wave's next entry
	}
}
__global__ __launch_bounds__(256) void k_split(Ctl *ctl, PoolView pool, const uint32_t *SPL, uint32_t spl_cap, uint32_t epoch, volatile uint32_t *hv, SbBase *scan2, uint32_t round1)
{
	__shared__ uint16_t s_row[MW][7][SB];
	const uint32_t nb = gridDim.x - (scan2 ? 1u : 0u);         // (the last block: the chunk bases of the directory, see SplitArgs::scan2)
	if (scan2 && blockIdx.x == nb) { __shared__ uint64_t s_w2[4]; sbscan2_lean<256, 2>(ctl, scan2, s_w2); return; }
	split_body(ctl, pool, SPL, spl_cap, epoch, hv, blockIdx.x, nb, s_row, round1);
}

} // namespace rb2
#include "rb2_merge.h"
namespace rb2 {

// ---------------------------------------------------------------------------------------------
// rank directory of the new side
// ---------------------------------------------------------------------------------------------

// one wave per TWO superblocks (lanes 0-31 / 32-63): per-leaf counts -> exclusive prefix inside the superblock;
// superblock totals.  Counts are <= LEAF per leaf, prefixes < 2^16: two symbols per packed DPP scan.
// Runs after every kernel that rewrites whole pieces (k_merge, k_relayout, the loader).  sparse: the pool is in the sparse layout --
// its directory holds own counts by rows (dir_row, rb2_device.h), so this is a transposition of own[]; the in-place rounds
// that follow keep rows and totals current themselves (dir_put, rb2_merge.h).
template <bool STRIDE> __global__ __launch_bounds__(256) void k_meta_sb(const Ctl *ctl, int nside, PoolView newp, SbTot *sbtot, int sparse)
{
	const int ln = lane_id();
	const uint64_t nsb = ctl->nsb_total;
	for (uint64_t sbw = ((uint64_t)blockIdx.x * 4 + wave_id()) * 2; sbw < nsb; sbw += (uint64_t)gridDim.x * 8) {   // grid stride (a sharded rank launches fewer waves than its upper bound)
	const uint64_t sb = sbw + (ln >> 5);
	const bool live = sb < nsb;
	const uint64_t gl = sb * SB + (ln & 31);
	// sub-ropes start on superblock boundaries, in ascending order: the one that owns a superblock is the
	// last with sb0 <= sb (one strided load + ballot per half, see seg_of); its tail leaves may be padding
	const uint64_t sb0 = ctl->rope[nside][ln < NR ? ln : NR - 1].sb0;
	const uint64_t sbA = sb - (ln >> 5);
	const int rA = max(0, (int)__popcll(__ballot(ln < NR && sb0 <= sbA)) - 1);
	const int rB = max(0, (int)__popcll(__ballot(ln < NR && sb0 <= sbA + 1)) - 1);
	const RopeDesc &rp = ctl->rope[nside][(ln >> 5) ? rB : rA];
	const bool ok = live && gl >= rp.leaf0 && gl < rp.leaf0 + rp.nleaves;
	LeafMeta m;
	for (int s = 0; s < 6; ++s) m.c[s] = 0;
	m.npre = 0; m.n = 0;
	if (ok) m = newp.own[gl];                                  // own counts + fill, written by the merge kernels / k_relayout / the loader
	const uint32_t e01 = m.c[0] | (uint32_t)m.c[1] << 16, e23 = m.c[2] | (uint32_t)m.c[3] << 16, e45 = m.c[4] | (uint32_t)m.c[5] << 16;
	uint32_t s01 = dpp_incl_add(e01), s23 = dpp_incl_add(e23), s45 = dpp_incl_add(e45);
	const uint32_t h01 = (uint32_t)__builtin_amdgcn_readlane((int)s01, 31), h23 = (uint32_t)__builtin_amdgcn_readlane((int)s23, 31), h45 = (uint32_t)__builtin_amdgcn_readlane((int)s45, 31);
	if (ln >> 5) { s01 -= h01; s23 -= h23; s45 -= h45; }      // second superblock: prefix relative to its own first leaf
	const uint32_t x01 = s01 - e01, x23 = s23 - e23, x45 = s45 - e45;
	if (sparse) {
		if (live) {                                            // all 32 slots: an unused one must read as empty
			uint16_t *dr = dir_row(newp, sb, 0) + (ln & 31);
			dr[0] = (uint16_t)(m.n | ((m.c[0] | m.c[5]) ? FILL_P2 : 0u));   // (a leaf without `$` and `N` is told by two planes: its third line is dead from here on)
#pragma unroll
			for (int s = 0; s < 6; ++s) dr[(1 + s) * SB] = m.c[s];
			dr[7 * SB] = 0;                                        // spare row: the claim word of k_split
		}
	} else if (ok) {
		m.c[0] = (uint16_t)x01; m.c[1] = (uint16_t)(x01 >> 16); m.c[2] = (uint16_t)x23; m.c[3] = (uint16_t)(x23 >> 16);
		m.c[4] = (uint16_t)x45; m.c[5] = (uint16_t)(x45 >> 16);
		m.npre = (uint16_t)((x01 & 0xffffu) + (x01 >> 16) + (x23 & 0xffffu) + (x23 >> 16) + (x45 & 0xffffu) + (x45 >> 16));   // <= SB * LEAF < 2^16
		newp.meta[gl] = m;
	}
	if ((ln & 31) == 31 && live) {                             // inclusive prefix of the last leaf = superblock total (<= 43008 per symbol)
		SbTot c;
		c.p01 = s01; c.p23 = s23; c.p45 = s45; c.pad = 0;
		sbtot[sb] = c;
	}
	if (!STRIDE) return;
	}
}

// the two-kernel prefix over the superblock totals (k_sbscan3: prefixes inside every chunk + the chunk totals; k_sbscan2: the chunk bases).  A total is six 16-bit counts (<= SB * LEAF = 32768 each) in 16 bytes; inside
// a chunk of 1024 superblocks sums stay below 2^25, so the chunk-level work is 32-bit DPP scans with one LDS exchange and what is
// stored per superblock is a 32-byte record of 32-bit prefixes (SbRec); only the chunk bases (SbBase) are 64 bit.
// (blocks of 256 threads, four consecutive superblocks per thread -- 64 bytes in flight per lane: with 1024-thread blocks of one
// superblock per thread the two streaming kernels ran at half the rate, each block waiting out its one round trip to memory)
constexpr int SBT = 4;                      // superblocks per thread in k_sbscan3
// exclusive prefix over the chunk totals, in place (one block): a thread takes eight consecutive chunks, the wave and block levels are
// shuffles and one LDS exchange -- one pass for up to 8192 chunks (270 G symbols); more: with a running total
constexpr int SB2T = 512;                   // threads of k_sbscan2 (its one block)
template <int NT, int CT = 8> __device__ __forceinline__ void sbscan2_body(const Ctl *ctl, SbBase *base, uint64_t (*s_w)[NT / 64])
{
	constexpr int SB2T = NT;                                    // CT: chunks per thread -- 4096 per pass of k_sbscan2's 512 threads, all six columns in flight at once
	const uint64_t nc = (ctl->nsb_total + SCHUNK - 1) / SCHUNK;
	const int ln = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint64_t run[6] = {0, 0, 0, 0, 0, 0};
	for (uint64_t i0 = 0; i0 < nc; i0 += CT * SB2T) {
		const uint64_t j0 = i0 + (uint64_t)threadIdx.x * CT;
		uint64_t v[CT][6], tot[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < CT; ++k)
#pragma unroll
			for (int s = 0; s < 6; ++s) v[k][s] = j0 + k < nc ? base[j0 + k].cum[s] : 0ull;
#pragma unroll
		for (int k = 0; k < CT; ++k)
#pragma unroll
			for (int s = 0; s < 6; ++s) { const uint64_t x = v[k][s]; v[k][s] = tot[s]; tot[s] += x; }
		uint64_t inc[6];
#pragma unroll
		for (int s = 0; s < 6; ++s) { inc[s] = wave_incl_add<uint64_t>(tot[s]); if (ln == 63) s_w[s][wv] = inc[s]; }
		__syncthreads();
		uint64_t b0[6];
#pragma unroll
		for (int s = 0; s < 6; ++s) {
			uint64_t off = 0, all = 0;
			for (int w = 0; w < SB2T / 64; ++w) { const uint64_t x = s_w[s][w]; if (w < wv) off += x; all += x; }
			b0[s] = run[s] + off + inc[s] - tot[s];
			run[s] += all;
		}
#pragma unroll
		for (int k = 0; k < CT; ++k) if (j0 + k < nc) {
			SbBase o;
#pragma unroll
			for (int s = 0; s < 6; ++s) o.cum[s] = b0[s] + v[k][s];
			o.pos = o.cum[0] + o.cum[1] + o.cum[2] + o.cum[3] + o.cum[4] + o.cum[5]; o.pad = 0;
			base[j0 + k] = o;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {                                     // one entry past the last chunk: the pool's totals (the descent of an in-place round reads base[c + 1] as the end of chunk c)
		SbBase o;
#pragma unroll
		for (int s = 0; s < 6; ++s) o.cum[s] = run[s];
		o.pos = run[0] + run[1] + run[2] + run[3] + run[4] + run[5]; o.pad = 0;
		base[nc] = o;
	}
}
// the same, for a block that rides in another kernel's launch ("the directory rides along", k_advance): one column at a time, two chunks per thread -- two dozen
// registers instead of a hundred (all six columns of eight chunks in flight, 64-bit values: what the riding block needs, its host kernel is compiled for);
// the chunk totals are a few thousand records, cache-resident, and nothing waits for this block but the launch it rides in
// one column of the chunk bases (0-5: a symbol's counts, 6: the positions) by one block: exclusive prefix in place, the total one entry past the last chunk
template <int NT, int CT> __device__ __forceinline__ void sbscan2_col(const Ctl *ctl, SbBase *base, uint64_t *s_w, const int col)
{
	const uint64_t nc = (ctl->nsb_total + SCHUNK - 1) / SCHUNK;
	const int ln = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint64_t *colp = col < 6 ? &base[0].cum[col] : &base[0].pos;  // (records of eight 64-bit words)
	uint64_t run = 0;
#pragma unroll 1
	for (uint64_t i0 = 0; i0 < nc; i0 += (uint64_t)CT * NT) {       // CT consecutive chunks per thread and pass
		const uint64_t j0 = i0 + (uint64_t)threadIdx.x * CT;
		uint64_t v[CT], tot = 0;
#pragma unroll
		for (int k = 0; k < CT; ++k) v[k] = j0 + k < nc ? colp[(j0 + k) * 8] : 0ull;
#pragma unroll
		for (int k = 0; k < CT; ++k) { const uint64_t x = v[k]; v[k] = tot; tot += x; }
		const uint64_t inc = dpp_incl_add64(tot);
		if (ln == 63) s_w[wv] = inc;
		__syncthreads();
		uint64_t off = 0, all = 0;
		for (int w = 0; w < NT / 64; ++w) { const uint64_t x = s_w[w]; if (w < wv) off += x; all += x; }
		const uint64_t b0 = run + off + inc - tot;
#pragma unroll
		for (int k = 0; k < CT; ++k) if (j0 + k < nc) colp[(j0 + k) * 8] = b0 + v[k];
		run += all;
		__syncthreads();
	}
	if (threadIdx.x == 0) { colp[nc * 8] = run; if (col == 6) base[nc].pad = 0; }   // one entry past the last chunk: the pool's totals
}
static_assert(sizeof(SbBase) == 64, "sbscan2_col strides over the records in 64-bit words");
// all seven, one after the other (a 256-thread block that rides in k_sym<SPLIT> / k_split)
template <int NT, int CT> __device__ __forceinline__ void sbscan2_lean(const Ctl *ctl, SbBase *base, uint64_t *s_w)
{
#pragma unroll 1
	for (int col = 0; col < 7; ++col) sbscan2_col<NT, CT>(ctl, base, s_w, col);
}
__global__ __launch_bounds__(SB2T) void k_sbscan2(const Ctl *ctl, SbBase *base)
{
	__shared__ uint64_t s_w[6][SB2T / 64];
	sbscan2_body<SB2T>(ctl, base, s_w);
}
// (the body: k_sbscan3 proper, and blocks of their own in the k_advance launch of an in-place round -- "the directory rides along", k_advance)
__device__ __forceinline__ void sbscan3_body(const Ctl *ctl, const SbTot *sbtot, const PoolView &newp, const uint32_t blk, uint32_t (*s_p)[4])
{
	const uint64_t n = ctl->nsb_total, i0 = (uint64_t)blk * SCHUNK + (uint64_t)threadIdx.x * SBT;
	if ((uint64_t)blk * SCHUNK >= n) return;
	const int ln = lane_id(), wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint32_t e[SBT][6], tot[6] = {0, 0, 0, 0, 0, 0}, own[SBT];  // exclusive inside the thread, the thread's totals; the symbols of each superblock
#pragma unroll
	for (int k = 0; k < SBT; ++k) {
		SbTot t; t.p01 = t.p23 = t.p45 = t.pad = 0;
		if (i0 + k < n) t = sbtot[i0 + k];
		const uint32_t v[6] = { t.p01 & 0xffffu, t.p01 >> 16, t.p23 & 0xffffu, t.p23 >> 16, t.p45 & 0xffffu, t.p45 >> 16 };
		own[k] = v[0] + v[1] + v[2] + v[3] + v[4] + v[5];
#pragma unroll
		for (int s = 0; s < 6; ++s) { e[k][s] = tot[s]; tot[s] += v[s]; }
	}
	uint32_t inc[6];
#pragma unroll
	for (int s = 0; s < 6; ++s) { inc[s] = dpp_incl_add(tot[s]); if (ln == 63) s_p[s][wv] = inc[s]; }
	__syncthreads();
	uint32_t b0[6];
#pragma unroll
	for (int s = 0; s < 6; ++s) {
		uint32_t off = 0;
#pragma unroll
		for (int w = 0; w < 3; ++w) if (w < wv) off += s_p[s][w];
		b0[s] = off + inc[s] - tot[s];
	}
	// the chunk's totals, for k_sbscan2 behind this kernel (a kernel of its own read every total a second time for them)
	if (threadIdx.x < 6) newp.sbbase[blk].cum[threadIdx.x] = (uint64_t)s_p[threadIdx.x][0] + s_p[threadIdx.x][1] + s_p[threadIdx.x][2] + s_p[threadIdx.x][3];
	if (threadIdx.x == 6) {                                      // ... and its symbols: the position column is scanned like the six others (sbscan2_col), by whoever gets to it
		uint64_t p = 0;
#pragma unroll
		for (int s = 0; s < 6; ++s) p += (uint64_t)s_p[s][0] + s_p[s][1] + s_p[s][2] + s_p[s][3];
		newp.sbbase[blk].pos = p;
	}
#pragma unroll
	for (int k = 0; k < SBT; ++k) if (i0 + k < n) {             // one 32-byte record per superblock: 128 contiguous bytes per lane
		uint32_t o[6];
#pragma unroll
		for (int s = 0; s < 6; ++s) o[s] = b0[s] + e[k][s];
		uint4 *q = (uint4*)&newp.sbrec[i0 + k];
		q[0] = make_uint4(o[0], o[1], o[2], o[3]);
		q[1] = make_uint4(o[4], o[5], o[0] + o[1] + o[2] + o[3] + o[4] + o[5], own[k]);   // (pos, tot: what one probe of the descent reads: k_part_sparse)
	}
}
__global__ __launch_bounds__(SCHUNK / SBT) void k_sbscan3(const Ctl *ctl, const SbTot *sbtot, PoolView newp)
{
	__shared__ uint32_t s_p[6][4];
	sbscan3_body(ctl, sbtot, newp, blockIdx.x, s_p);
}

// ---------------------------------------------------------------------------------------------
// k_advance: new interval of every string (mrope.c:226-229 + 332-340), consume one symbol, and
// the stable 6-way partition into next round's buckets (mrope.c:303-309)
// ---------------------------------------------------------------------------------------------

template <bool AE, bool SPARSE, typename P> __device__ __forceinline__ bool advance_tile(const uint32_t tile, const Ctl *ctl, int side, int is_comp, uint32_t round, const uint8_t *s, const PoolView &newp,
		uint8_t *A2, const uint8_t *A, const TileFix *tf,
		const P *SIZE, const P *INS_E, const uint16_t *RKREL, const P *L, const uint64_t *W,
		P *L2, P *U2, uint64_t *W2, ShardRec *send, const P *RKOLD, const PushTab *push);
struct ScanRide { const SbTot *sbtot; uint32_t nscan; };     // in-place rounds: the first nscan blocks of the k_advance launch do k_sbscan3's work (see k_advance)

template <bool AE, bool SPARSE = false, bool STRIDE = false, typename P = uint64_t> __global__ __launch_bounds__(256) void k_advance(const Ctl *ctl, int side, int is_comp, uint32_t round, const uint8_t *s, PoolView newp,
		uint8_t *A2, const uint8_t *A, const TileFix *tf,
		const P *SIZE, const P *INS_E, const uint16_t *RKREL, const P *L, const uint64_t *W,
		P *L2, P *U2, uint64_t *W2, ShardRec *send, const P *RKOLD, const PushTab *push, ScanRide sr)
{
	// THE DIRECTORY RIDES ALONG.  An in-place round leaves the superblock totals current (k_merge_leaf); what is left is the prefix over them
	// (k_sbscan3 + k_sbscan2: two launches, 12 us per round at 10 G symbols, ~100 us at 90 G -- the one part of the round that reads every
	// superblock).  Nothing in this kernel reads the directory in an in-place round any more (ranks come from before the merge: RKOLD), and
	// the next reader is the next round's descent -- so the first sr.nscan blocks of this launch ARE k_sbscan3 (one chunk of superblocks
	// each), beside the tile blocks, and the chunk bases follow in one block of the next launch (k_sym<.., SPLIT> / k_split: SplitArgs::scan2).
	// No launch, no event, no second stream; the latency-bound tile blocks leave the memory system to the scan.
	uint32_t nsc = 0;
	if (SPARSE) {
		nsc = sr.nscan;
		if (blockIdx.x < nsc) { __shared__ uint32_t s_p[6][4]; sbscan3_body(ctl, sr.sbtot, newp, blockIdx.x, s_p); return; }
	}
	for (uint32_t tile = (SPARSE && nsc) ? blockIdx.x - nsc : (STRIDE ? blockIdx.x : xcd_item()); ; ) {   // first tile as a one-tile-per-block kernel would run it, then a grid stride (see k_prep)
		if (!advance_tile<AE, SPARSE, P>(tile, ctl, side, is_comp, round, s, newp, A2, A, tf, SIZE, INS_E, RKREL, L, W, L2, U2, W2, send, RKOLD, push)) return;
		if (!STRIDE) return;
		tile += gridDim.x - nsc;
		if (tile >= ctl->seg[side].tile0[NR]) return;
		__syncthreads();
	}
}

template <bool AE, bool SPARSE, typename P> __device__ __forceinline__ bool advance_tile(const uint32_t tile, const Ctl *ctl, int side, int is_comp, uint32_t round, const uint8_t *s, const PoolView &newp,
		uint8_t *A2, const uint8_t *A, const TileFix *tf,
		const P *SIZE, const P *INS_E, const uint16_t *RKREL, const P *L, const uint64_t *W,
		P *L2, P *U2, uint64_t *W2, ShardRec *send, const P *RKOLD, const PushTab *push)
{
	__shared__ GroupLds G;
	const TileFix &tfx = tf[tile];                              // issued together with the mode and tile-count loads
	const SegDesc &sg = ctl->seg[side];
	if ((ctl->ne[round & 1] == 0) != AE) return false;
	if (SPARSE && ctl->overflow) return false;                 // void round: the host redoes it on the dense layout
	if (tile >= sg.tile0[NR]) return false;
	TileCtx t;
	tile_ctx_fix(tfx, t);
	int sym2[2], flag2[2];
	uint64_t w2[2]; P l2[2];                                   // issued before the barriers of group_setup
	uint32_t araw[2];
	const uint32_t fixw = tilefix_word(tf, tile);
	// Every access of the tile is a wave-uniform base + a 32-bit offset, and every position is computed in the batch's storage width P
	// (while positions are stored in 32 bits the sums below are exact modulo 2^32 and their results are below 2^32: half the integer
	// instructions and registers of the 64-bit form -- the kernel issued 352 VALU per wave of 128 strings)
	const uint32_t nval = (uint32_t)min((uint64_t)STILE, t.segend - t.base);   // strings in this tile
	{
		const uint64_t *Wb = W + t.base; const P *Lb = L + t.base; const uint8_t *Ab = A + t.base;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const uint32_t x = (uint32_t)(h * 256) + threadIdx.x;
			w2[h] = 0; l2[h] = 0; araw[h] = 7;
			if (x < nval) { w2[h] = Wb[x]; l2[h] = Lb[x]; araw[h] = Ab[x]; }
		}
	}
	// per tile and symbol: AC offset minus the directory prefix in front of the piece, and where the bucket's members that insert the
	// symbol go -- six values each, looked up in LDS by every string instead of rebuilt from two directory loads and two table loads
	__shared__ P s_acb[6]; __shared__ uint32_t s_dst[6];       // (a batch has < 2^32 strings: places in the string arrays fit 32 bits)
	const RopeDesc &nrp = ctl->rope[side ^ 1][t.b];
	__shared__ uint32_t s_pr[6];                               // PEER transport: the rank a member that inserts the symbol moves to
	if (threadIdx.x < 6) {
		const int a6 = threadIdx.x;
		s_acb[a6] = SPARSE ? (P)ctl->ac[t.b][a6] : (P)(ctl->ac[t.b][a6] - sb_cum(newp, nrp.sb0, a6));   // (in-place rounds: ranks come piece-relative, see rank_issue)
		s_dst[a6] = (uint32_t)(push ? ctl->pdst[t.b][a6] : ctl->dest[t.b][a6]);
		s_pr[a6] = push ? ctl->pdev[t.b][a6] : 0u;
	}
	// The gathers of the round -- the directory entries in front of my new symbol, its rank inside its leaf, the text of a cursor that ran
	// empty -- are issued NOW, on the assumption that every string of the tile is a group of its own (slot = index in the bucket; the rule
	// from round ~14 of a batch on): they are in flight while group_setup runs its ballots and barriers.  A tile with a larger group
	// (block-uniform: G.allsingle) asks again with its real slots.  (r04: gathers behind the barriers -- one more round trip per tile.)
	const int orda[6] = { sym_ord(0, is_comp), sym_ord(1, is_comp), sym_ord(2, is_comp), sym_ord(3, is_comp), sym_ord(4, is_comp), sym_ord(5, is_comp) };
	// rank of a in front of my new symbol on the NEW rope = directory prefix of its leaf + count inside
	// the leaf (k_merge); minus the PA new a's in front of it = rank on the old rope = what
	// rope_insert_run returns (rope.c:147) before the a's of earlier groups (PGA) are added back
	P rk[2], sz[2]; uint64_t wv[2];
	bool act0[2];
	// the four (dense layout) loads of one rank, issued together and only added up when their sum is needed
	struct RankRaw { P sbb; uint32_t sbr, meta, rkrel; };
	// the piece's first leaf / superblock (a piece starts on a superblock boundary) -- as scalars whatever branch the compiler first loads them in (a value merged behind a divergent branch
	// lives in vector registers, and so does every address built on it: seven 64-bit vector additions per string)
	const uint64_t sb0u = uniform64(nrp.sb0), leaf0u = uniform64(nrp.leaf0);
	const LeafMeta *metab = newp.meta + leaf0u; const SbRec *sbrb = newp.sbrec + sb0u;
	const uint16_t *RKb = RKREL + t.segstart; const P *Eb = INS_E + t.segstart; const P *ROb = SPARSE ? RKOLD + t.segstart : nullptr;
	auto rank_issue = [&](int h, int a, uint32_t slot, P F, bool flag, RankRaw &q) {
		if (SPARSE) {                                          // in-place rounds: the rank on the rope as it was BEFORE the round, taken on the way down: in front of the leaf (k_part_sparse) + inside it (k_merge_leaf)
			q.sbb = ROb[slot]; q.sbr = 0; q.meta = 0; q.rkrel = RKb[slot];
			return;
		}
		const P f = (P)(((!AE && flag) ? Eb[slot] : (P)(l2[h] - F)) + (P)slot);   // where my symbol went: e + slot; empty interval: e = l - F (k_prep)
		const uint32_t lf = (uint32_t)(f >> LEAF_SH), sbq = (uint32_t)sb0u + lf / SB;         // its leaf of the piece (a piece has < 2^32 leaves), its superblock of the pool
		if (sizeof(P) == 4) {                                  // positions fit 32 bits: so does every byte offset below (< 2^22 leaves) -- a scalar base + one 32-bit vector offset per gather
			const uint32_t ua = (uint32_t)a;
			q.sbb = *(const P*)((const char*)newp.sbbase + ((sbq >> SCHUNK_SH) * (uint32_t)sizeof(SbBase) + ua * 8u));   // (little endian: the low half of the 64-bit sum)
			q.sbr = *(const uint32_t*)((const char*)sbrb + ((lf / SB) * (uint32_t)sizeof(SbRec) + ua * 4u));
			q.meta = *(const uint16_t*)((const char*)metab + (lf * (uint32_t)sizeof(LeafMeta) + ua * 2u));
			q.rkrel = *(const uint16_t*)((const char*)RKb + slot * 2u);
			return;
		}
		const P *bb = (const P*)&newp.sbbase[sbq >> SCHUNK_SH].cum[a];                       // (little endian: the low half when positions are stored in 32 bits)
		q.sbb = *bb; q.sbr = sbrb[lf / SB].cum[a]; q.meta = metab[lf].c[a]; q.rkrel = RKb[slot];
	};
	auto rank_sum = [](const RankRaw &q) -> P { return (P)(q.sbb + q.sbr + q.meta + q.rkrel); };
	RankRaw rq[2];
	const bool spec = (tfx.nexthead & 4u) != 0;                // k_sym saw a tile of one-member groups (tile-uniform): ask now; else after the groups are known
#pragma unroll
	for (int h = 0; h < 2; ++h) {                              // speculative: slot = F = my index in the bucket
		const uint32_t x = (uint32_t)(h * 256) + threadIdx.x;
		const int a = (int)(araw[h] & 7u);
		act0[h] = (uint32_t)(a - 1) < 6u;                        // a symbol, not the sentinel (0), not "no string here" (7: araw of x >= nval) -- one comparison, one branch
		RB2_UNDEFV(rq[h].sbb); RB2_UNDEFV(rq[h].sbr); RB2_UNDEFV(rq[h].meta); RB2_UNDEFV(rq[h].rkrel);   // (only looked at by strings that asked: no zeros to write)
		const uint32_t slot = (uint32_t)t.lt * STILE + x;
		if (spec && act0[h]) rank_issue(h, a, slot, (P)slot, (araw[h] & 0x40u) != 0, rq[h]);
	}
	// a cursor that ran empty (one string in CUR_SYMS per round) is refilled from the batch text: the 16 bytes are asked for here, with the
	// gathers -- by EVERY lane, the ones that need nothing read the first bytes of the text (one hot line): a load inside a branch of its
	// own was moved behind the barriers by the compiler, where every wave then waited a whole round trip for it
	uint32_t rt[2][4]; bool need[2], fastr[2];
	const uint64_t tlen = ctl->len;
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		sz[h] = 0;
		wv[h] = cur_next(w2[h]);
		need[h] = act0[h] && cur_empty(wv[h]);
		const uint64_t tp = cur_pos(wv[h]);
		fastr[h] = need[h] && tp + 16 <= tlen;
		const uint32_t *q = fastr[h] ? (const uint32_t*)(s + (tp & ~3ull)) : (const uint32_t*)ctl;   // (16 readable bytes whatever the batch)
		rt[h][0] = q[0]; rt[h][1] = q[1]; rt[h][2] = q[2]; rt[h][3] = q[3];
	}
	group_setup(G, t, araw, fixw, sym2, flag2);                 // (its barriers also cover the two tables)
	uint32_t nz = 0;
	// The two strings of a thread go through the kernel level by level -- group bookkeeping (LDS), then every gather of both, then the
	// stores: written string after string, the second one's loads sat behind the first one's stores (the arrays may alias as far as the
	// compiler knows) and a thread walked two chains of dependent round trips one after the other.
	bool act[2];
	Member mem[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int x = h * 256 + threadIdx.x;
		act[h] = (uint32_t)x < nval && sym2[h] != 0;            // sentinel inserted: string is done (mrope.c:310)
		if (act[h]) mem[h] = group_member(G, t, x, sym2[h], orda);
	}
	if (!spec || !G.allsingle) {                               // (block-uniform) some group of the tile has more than one member: the real slots
#pragma unroll
		for (int h = 0; h < 2; ++h) if (act[h]) rank_issue(h, sym2[h], mem[h].slot, (P)mem[h].F, flag2[h] != 0, rq[h]);
	}
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		rk[h] = rank_sum(rq[h]);
		if (need[h]) {
			const uint64_t tp = cur_pos(wv[h]);
			wv[h] = fastr[h] ? cur_make(tp + CUR_SYMS, pack9_words(rt[h][0], rt[h][1], rt[h][2], rt[h][3], (uint32_t)(tp & 3) * 8)) : cur_refill(s, tlen, wv[h]);   // (slow way: the last bytes of the text)
		}
	}
	if (!AE) {
#pragma unroll
		for (int h = 0; h < 2; ++h) if (act[h] && flag2[h]) sz[h] = (SIZE + t.base)[(uint32_t)(h * 256) + threadIdx.x];
	}
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		if (!act[h]) continue;
		const int a = sym2[h];
		const Member &m = mem[h];
		const P l = SPARSE ? (P)(s_acb[a] + rk[h] + (P)m.pga)    // old rank + the a's of earlier groups (mrope.c:226-229)
		                   : (P)(s_acb[a] + rk[h] - (P)m.pa + (P)m.pga);   // dense rounds: the rank was taken on the NEW rope, the pa new a's in front of mine included
		const P u = (P)(l + sz[h]);
		const uint32_t d = s_dst[a] + m.pa;
		if (push) {                                            // sharded, PEER transport: straight into the next arrays of the owner of piece (a, b) (mrope.c:303-309: the scatter is a write)
			const uint32_t pr = s_pr[a];
			push->L2[pr][d] = l; push->W2[pr][d] = wv[h]; push->A2[pr][d] = (uint8_t)cur_sym(wv[h]);
			push->U2[pr][d] = u;                                 // always: the owner may hold non-empty intervals of other senders next round and then reads U of every string
			if (!AE && u != l) push->ctl[pr]->ne[(round & 1) ^ 1] = 1;   // (the owner's flag: its next round sees a non-empty interval)
		} else if (send) {                                     // sharded, RCCL transport: the string travels as a record, cursor and all
			send[ctl->sdest[t.b][a] + m.pa] = shard_pack((uint64_t)l, (uint64_t)(u - l), 0u, wv[h]);
		} else {
			L2[d] = l; W2[d] = wv[h]; A2[d] = (uint8_t)cur_sym(wv[h]);
			if (!AE) { U2[d] = u; nz += (u != l); }            // AE: u == l for every string of the batch from here on; U is dead
		}
	}
	if (!AE && !send && !push) {                               // does the next round see a non-empty interval?  (a flag: plain store, no atomic)
		if (__any(nz != 0) && lane_id() == 0) ((Ctl*)ctl)->ne[(round & 1) ^ 1] = 1;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// misc: synthetic reads, single rank query
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t splitmix(uint64_t seed, uint64_t k)
{
	uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}

// 16 output bytes per thread (one 16-byte store; HIP caps a launch at 2^32 threads, a -m10g batch has 10^10 bytes);
// strand 0: [rev(read) 0]; strand 1: [rev(read) 0 comp(read) 0] (main.c:200-237)
// genome_len == 0: i.i.d. bases.  genome_len > 0: read i is a window of a random "genome" of that many
// bases (base p = splitmix(seed, p) >> 62), starting at splitmix(seed ^ COV_SALT, i) % (genome_len - L + 1): reads overlap
// like sequencing reads at coverage n*L/genome_len, so suffix-array intervals stay non-empty and groups stay large.
constexpr uint64_t COV_SALT = 0x5bd1e995c0f3a1d7ull;
__device__ __forceinline__ uint64_t synth_base_index(uint64_t i, uint32_t j, uint32_t L, uint64_t seed, uint64_t genome_len)
{
	return genome_len ? splitmix(seed ^ COV_SALT, i) % (genome_len - L + 1) + j : i * L + j;
}
// skew != 0: skewed composition (85 % A, 5 % each C, G, T) from the top byte of the same random word (tools/synth_reads.c)
__device__ __forceinline__ uint32_t synth_base(uint64_t z, int skew)
{
	if (!skew) return (uint32_t)(z >> 62);
	const uint32_t u = (uint32_t)(z >> 56);
	return u < 218u ? 0u : 1u + (u - 218u) % 3u;
}
__device__ __forceinline__ uint8_t synth_byte(uint64_t g, uint64_t first, uint64_t per, uint32_t L, uint64_t seed, uint64_t genome_len, int skew)
{
	const uint64_t r = g / per; uint32_t off = (uint32_t)(g % per);
	const uint64_t i = first + r;
	if (off < L) return (uint8_t)(1 + synth_base(splitmix(seed, synth_base_index(i, L - 1 - off, L, seed, genome_len)), skew));   // reversed forward strand
	if (off == L) return 0;
	off -= L + 1;
	return off < L ? (uint8_t)(4 - synth_base(splitmix(seed, synth_base_index(i, off, L, seed, genome_len)), skew)) : 0;       // complement, original order
}
__global__ __launch_bounds__(256) void k_synth(uint8_t *dst, uint64_t first, uint64_t n_reads, uint32_t L, uint64_t seed, int strand, uint64_t genome_len, int skew)
{
	const uint64_t per = (uint64_t)(L + 1) * (strand ? 2 : 1), total = n_reads * per;
	const uint64_t g0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
	if (g0 >= total) return;
	if (g0 + 16 <= total) {
		uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < 16; ++k) w[k >> 2] |= (uint32_t)synth_byte(g0 + k, first, per, L, seed, genome_len, skew) << ((k & 3) * 8);
		*(uint4*)(dst + g0) = make_uint4(w[0], w[1], w[2], w[3]);
	} else for (uint64_t g = g0; g < total; ++g) dst[g] = synth_byte(g, first, per, L, seed, genome_len, skew);
}

// one wave per query: counts of the six symbols in [0, x[i]) of ROPE b (its pieces (b,$), (b,A), ... in order)
__global__ __launch_bounds__(256) void k_rank_batch(const Ctl *ctl, int side, PoolView pv, int b, const uint64_t *x, uint64_t n, uint64_t *out, int sparse)
{
	const uint64_t i = (uint64_t)blockIdx.x * MW + (threadIdx.x >> 6);
	if (i >= n) return;
	uint64_t p = x[i], acc[6] = {0, 0, 0, 0, 0, 0};
	for (int r = 0; r < NR && p > 0; ++r) {
		if (rope_sym(r) != b) continue;
		const RopeDesc &d = ctl->rope[side][r];
		uint64_t c[6];
		if (p >= d.n) { for (int s = 0; s < 6; ++s) acc[s] += d.cnt[s]; p -= d.n; continue; }
		if (sparse) wave_rank_all<true>(pv, d, p, c); else wave_rank_all<false>(pv, d, p, c);
		for (int s = 0; s < 6; ++s) acc[s] += c[s];
		p = 0;
	}
	if (lane_id() < 6) { uint64_t v = acc[0]; for (int s = 1; s < 6; ++s) if (lane_id() == s) v = acc[s]; out[i * 6 + lane_id()] = v; }
}

// position-weighted checksum of the symbols of piece r (dense layout: a flat array of groups, three plane words each): sum over its
// plane words of word * (2 * (3 * group + plane) + 1) mod 2^64 -- equal for equal symbol sequences whatever built them (one engine,
// N ranks, a loaded .fmr), sensitive to order.  Bits behind the last symbol are masked.  *out must be zeroed by the caller.
__global__ __launch_bounds__(256) void k_piece_hash(const Ctl *ctl, int side, PoolView pv, int r, unsigned long long *out)
{
	__shared__ uint64_t s_w[4];
	const RopeDesc &d = ctl->rope[side][r];
	const uint64_t ng = (d.n + GSYM - 1) / GSYM;                // groups in use
	const uint64_t *w = (const uint64_t*)pv.data + d.leaf0 * LEAFW;
	uint64_t acc = 0;
	for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < 3 * ng; j += (uint64_t)gridDim.x * 256) {
		const uint64_t G = j / 3; const uint32_t pl = (uint32_t)(j - 3 * G);
		uint64_t v = w[(G >> 4) * LEAFW + pl * LEAFG + (G & 15)];
		if (G == ng - 1) v &= bits_below((uint32_t)(d.n - G * GSYM));
		acc += v * (2 * j + 1);
	}
	uint64_t tot;
	block_excl_add<uint64_t>(acc, s_w, &tot);
	if (threadIdx.x == 0 && tot) atomicAdd(out, (unsigned long long)tot);
}

// one wave: counts of the six symbols in [0, p) of PIECE r (a sharded index answers a rope query piece by piece, each from its owner)
__global__ __launch_bounds__(64) void k_rank_piece(const Ctl *ctl, int side, PoolView pv, int r, uint64_t p, uint64_t *out, int sparse)
{
	const RopeDesc &d = ctl->rope[side][r];
	uint64_t c[6];
	if (sparse) wave_rank_all<true>(pv, d, p, c); else wave_rank_all<false>(pv, d, p, c);
	if (lane_id() < 6) { uint64_t v = c[0]; for (int s = 1; s < 6; ++s) if (lane_id() == s) v = c[s]; out[lane_id()] = v; }
}

// ---------------------------------------------------------------------------------------------
// loader: ropebwt2's run-length bytes (43+3 codec, rle.h:39-75) -> packed leaves of the dense layout, for an index that
// arrives from the host (mr_restore / -i old.fmr, rope_load_runs' counterpart).  The codec resynchronises on any byte --
// continuation bytes are 10xxxxxx, everything else starts a run -- so the byte stream of a rope is cut into blocks of LDB
// bytes that are decoded independently:
//   k_ld_count   symbols of the runs that start in each block (+ the rope's six symbol totals)
//   (host)       exclusive prefix over the blocks = the rope position of each block's first run
//   k_ld_expand  every thread re-decodes its 32 bytes, a block scan gives it the rope position of its first run, and it ORs
//                the symbols into the (zeroed) pieces of the rope: a piece is a flat array of 21-symbol words, consecutive
//                runs share words, so a thread collects one word's bits and issues one atomicOr per word it touches;
//                the symbol counts per piece (RopeDesc::cnt) are tallied in LDS and flushed once per block
//   k_ld_long    runs of more than LD_LONG symbols inside one piece, one block each
//   k_ld_own     own counts + fill of every leaf (k_meta_sb and the scans follow, as after any rewrite)
// ---------------------------------------------------------------------------------------------
constexpr int LDT = 32;                     // bytes per thread
constexpr int LDB = 256 * LDT;              // bytes per block
constexpr uint32_t LD_LONG = 4096;          // longer (parts of) runs are left to k_ld_long
struct LdPieces {                           // the pieces of one rope, in rope order
	uint64_t q[7];                          // rope position of the first symbol of piece i (q[np] = symbols of the rope)
	uint64_t word0[6];                      // first 64-bit word of the piece in the pool
	uint32_t keep[6];                       // 0: a piece held by another rank -- counted, not stored
	int32_t  r[6];                          // sub-rope index
	int32_t  np, pad;
};
struct LdLong { uint64_t word0, o, n; uint32_t c, pad; };   // n symbols c at symbol offset o of the piece that starts at word0

__device__ __forceinline__ int ld_dec43(const uint8_t *p, const uint8_t *end, uint32_t &c, uint64_t &l)   // bytes of the run, 0: truncated
{
	const uint32_t b0 = p[0];
	c = b0 & 7u;
	if ((b0 & 0x80u) == 0) { l = b0 >> 3; return 1; }
	if ((b0 >> 5) == 6u) { if (p + 2 > end) return 0; l = ((uint64_t)(b0 & 0x18u) << 3) | (p[1] & 0x3fu); return 2; }
	const int nb = (b0 & 0x10u) ? 8 : 4;
	if (p + nb > end) return 0;
	uint64_t v = (b0 >> 3) & 1u;
	for (int i = 1; i < nb; ++i) v = (v << 6) | (p[i] & 0x3fu);
	l = v;
	return nb;
}
// the runs that start in [o0, o0 + LDT): f(c, l) for each, in order
template <typename F> __device__ __forceinline__ void ld_runs(const uint8_t *rle, uint64_t nbytes, uint64_t o0, uint32_t *bad, F f)
{
	const uint64_t o1 = min(o0 + LDT, nbytes);
	for (uint64_t i = o0; i < o1; ++i) {
		if ((rle[i] & 0xc0u) == 0x80u) continue;
		uint32_t c; uint64_t l;
		const int nb = ld_dec43(rle + i, rle + nbytes, c, l);
		if (nb == 0 || c > 5u) { *bad = 1; continue; }
		f(c, l);
	}
}

__global__ __launch_bounds__(256) void k_ld_count(const uint8_t *rle, uint64_t nbytes, uint64_t *blk_n, unsigned long long *tot6, uint32_t *bad)
{
	__shared__ uint64_t s_w[4];
	__shared__ unsigned long long s_c[6];
	if (threadIdx.x < 6) s_c[threadIdx.x] = 0;
	__syncthreads();
	uint64_t n = 0, cnt[6] = {0, 0, 0, 0, 0, 0};
	ld_runs(rle, nbytes, ((uint64_t)blockIdx.x * 256 + threadIdx.x) * LDT, bad, [&](uint32_t c, uint64_t l) {
		n += l;
#pragma unroll
		for (int s = 0; s < 6; ++s) cnt[s] += c == (uint32_t)s ? l : 0ull;     // (no dynamic register indexing)
	});
#pragma unroll
	for (int c = 0; c < 6; ++c) { const uint64_t w = wave_sum<uint64_t>(cnt[c]); if (lane_id() == 0 && w) atomicAdd(&s_c[c], (unsigned long long)w); }
	uint64_t tot;
	block_excl_add<uint64_t>(n, s_w, &tot);                    // (its barriers also cover s_c)
	if (threadIdx.x == 0) blk_n[blockIdx.x] = tot;
	if (threadIdx.x < 6 && s_c[threadIdx.x]) atomicAdd(&tot6[threadIdx.x], s_c[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_ld_expand(const uint8_t *rle, uint64_t nbytes, const uint64_t *blk_off, LdPieces tab_arg, uint64_t *data,
		unsigned long long *pcnt /* [NR][6] */, LdLong *longs, uint32_t *nlong, uint32_t long_cap, uint32_t *bad)
{
	__shared__ uint64_t s_w[4];
	__shared__ unsigned long long s_pc[6][6];
	__shared__ LdPieces tab;                                   // indexed by the thread's piece cursor
	if (threadIdx.x == 0) tab = tab_arg;
	if (threadIdx.x < 36) s_pc[threadIdx.x / 6][threadIdx.x % 6] = 0;
	const uint64_t o0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * LDT;
	uint64_t n = 0;
	ld_runs(rle, nbytes, o0, bad, [&](uint32_t, uint64_t l) { n += l; });
	uint64_t S = blk_off[blockIdx.x] + block_excl_add<uint64_t>(n, s_w, (uint64_t*)0);   // rope position of my first run
	int x = 0;                                                 // piece of S (monotone)
	uint64_t lc[6] = {0, 0, 0, 0, 0, 0};                       // symbols this thread put into piece x so far
	auto tally_out = [&]() {
#pragma unroll
		for (int s = 0; s < 6; ++s) { if (lc[s]) atomicAdd(&s_pc[x][s], (unsigned long long)lc[s]); lc[s] = 0; }
	};
	uint64_t cur_g = ~0ull, cur_w0 = 0, cur_v[3] = {0, 0, 0};  // the group being collected: its number inside the piece, the piece's first word, its plane bits
	auto flush = [&]() {
		if (cur_g != ~0ull) {
			unsigned long long *q = (unsigned long long*)&data[cur_w0 + (cur_g >> 4) * LEAFW + (cur_g & 15)];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) if (cur_v[pl]) atomicOr(q + pl * LEAFG, (unsigned long long)cur_v[pl]);
		}
		cur_v[0] = cur_v[1] = cur_v[2] = 0;
	};
	ld_runs(rle, nbytes, o0, bad, [&](uint32_t c, uint64_t l) {
		while (l > 0) {
			if (x < tab.np && S >= tab.q[x + 1]) { tally_out(); while (x < tab.np && S >= tab.q[x + 1]) ++x; }
			if (x >= tab.np) { *bad = 2; return; }              // the rope is longer than the symbol counts of the other ropes imply
			const uint64_t part = min(l, tab.q[x + 1] - S), o = S - tab.q[x];
#pragma unroll
			for (int s = 0; s < 6; ++s) lc[s] += c == (uint32_t)s ? part : 0ull;
			if (tab.keep[x] && c != 0) {                       // $ = 0: the pool is zeroed
				if (part > LD_LONG) {
					const uint32_t k = atomicAdd(nlong, 1u);
					if (k < long_cap) { LdLong e; e.word0 = tab.word0[x]; e.o = o; e.n = part; e.c = c; e.pad = 0; longs[k] = e; }
				} else {
					uint64_t gd = o >> 6; uint32_t off = (uint32_t)(o & 63), t = (uint32_t)part;
					while (t > 0) {
						const uint32_t k = min(t, (uint32_t)GSYM - off);
						const uint64_t field = bits_below(k) << off;
						if (gd != cur_g || tab.word0[x] != cur_w0) { flush(); cur_g = gd; cur_w0 = tab.word0[x]; }
						if (c & 1u) cur_v[0] |= field;
						if (c & 2u) cur_v[1] |= field;
						if (c & 4u) cur_v[2] |= field;
						t -= k; off = 0; ++gd;
					}
				}
			}
			S += part; l -= part;
		}
	});
	flush();
	if (x < tab.np) tally_out();
	__syncthreads();
	if (threadIdx.x < 36) {
		const int px = threadIdx.x / 6, c = threadIdx.x % 6;
		if (px < tab.np && s_pc[px][c]) atomicAdd(&pcnt[tab.r[px] * 6 + c], s_pc[px][c]);
	}
}

__global__ __launch_bounds__(256) void k_ld_long(const LdLong *longs, const uint32_t *nlong, uint32_t long_cap, uint64_t *data)
{
	const uint32_t n = min(*nlong, long_cap);
	for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
		const LdLong L = longs[e];
		const uint64_t g0 = L.o >> 6, g1 = (L.o + L.n - 1) >> 6;        // groups [g0, g1] of the piece
		for (uint64_t gd = g0 + threadIdx.x; gd <= g1; gd += 256) {
			const uint64_t lo = max(L.o, gd << 6), hi = min(L.o + L.n, (gd + 1) << 6);   // symbols [lo, hi) of this group
			const uint64_t field = bits_below((uint32_t)(hi - lo)) << (uint32_t)(lo & 63);
			unsigned long long *q = (unsigned long long*)&data[L.word0 + (gd >> 4) * LEAFW + (gd & 15)];
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) if ((L.c >> pl) & 1u) atomicOr(q + pl * LEAFG, (unsigned long long)field);
		}
	}
}

// one wave per leaf of the piece [leaf0, leaf0 + nleaves) holding n symbols: own counts + fill
__global__ __launch_bounds__(256) void k_ld_own(PoolView pv, uint64_t leaf0, uint64_t nleaves, uint64_t n)
{
	const uint64_t i = (uint64_t)blockIdx.x * MW + wave_id();
	if (i >= nleaves) return;
	const uint32_t fill = (uint32_t)min((uint64_t)LEAF, n - i * LEAF);
	uint32_t c[6];
	wave_leaf_counts(leaf_words(pv.data, leaf0 + i), 0, fill, c);
	if (lane_id() == 0) {
		LeafMeta m;
		for (int s = 0; s < 6; ++s) m.c[s] = (uint16_t)c[s];
		m.npre = 0; m.n = (uint16_t)fill;
		pv.own[leaf0 + i] = m;
	}
}

} // namespace rb2
