// rb2_merge.h -- k_merge: rank + positional insert over the bit-plane leaves of every sub-rope;
//                k_merge_leaf: the same for the few leaves a sparse round touches, in place, one leaf per DPP row;
//                k_export: leaves -> ropebwt2's run-length bytes (only when the BWT leaves the GPU).
//
// Reference semantics: rope_insert_run (rope.c:114-148) -> rle_insert_cached (rle.c:10-89): put
// rl copies of symbol a in front of position x and return the number of a's before x.  The
// reference does this one run at a time through a B+ tree of run-length leaves; here one launch
// rewrites every sub-rope side -> side^1 as a merge of two sorted sequences (old symbols, new
// symbols).  In HBM a sub-rope is a flat array of symbols held as bit planes (a group = 64 symbols =
// three 64-bit words, a leaf = 16 groups, plane-major: rb2_device.h), so the merge is a pure stream:
// no run decoding, no re-encoding, no length-dependent paths, no divisions.  Run-length coding is
// applied once, by k_export, when the host asks for the ropes (mr_sync_host -> .fmd/.fmr writers).
//
// k_merge work decomposition: ONE WAVE PER OUTPUT WINDOW of 64 * GPL groups (WPL leaves), four
// independent waves per block, no block-level barrier anywhere.  Lane l owns GPL consecutive groups:
//   1. the new symbols of the window are OR-ed into position-indexed LDS words -- one flag word and
//      three plane words per group -- one 64-bit LDS atomic per set bit; the old groups the window
//      draws from are loaded at the same time (whole 128-byte lines) and staged in LDS
//   2. one packed wave prefix sum (not-new count | new count) -> first old symbol each lane
//      consumes; the old bits of each of its groups are an unaligned 64-bit window of each plane
//   3. expand: open one 1-bit gap per new symbol in each plane (wave-uniform loop, 1-2 trips in
//      steady state); the new symbols are already in place
//   4. symbol counts per lane: five popcounts of dense words per group, three packed scans -> new
//      LeafMeta of each leaf of the window
//   5. RKREL: every new symbol gets the number of equal symbols before it INSIDE its leaf, one new
//      symbol per lane (prefix of the owning lane + a masked plane compare of its group, both read back
//      from LDS); k_advance adds the directory prefix of the new sub-rope to obtain the reference's
//      return value of rope_insert_run.
#pragma once
#include <type_traits>
#include <utility>
#include "rb2_device.h"

namespace rb2 {

// ---- DPP row primitives: a row = 16 lanes = the 16 groups of one leaf
template <int J> __device__ __forceinline__ uint32_t row_share(uint32_t v) { return dpp0<0x150 + J, 0xf>(v); }   // lane J of my row, to every lane of the row
__device__ __forceinline__ uint32_t row_prev(uint32_t v) { return dpp0<0x111, 0xf>(v); }                          // lane - 1 of my row; its lane 0 reads 0
__device__ __forceinline__ uint32_t row_incl_add(uint32_t v)
{
	v += dpp0<0x111, 0xf>(v); v += dpp0<0x112, 0xf>(v); v += dpp0<0x114, 0xf>(v); v += dpp0<0x118, 0xf>(v);
	return v;
}
template <class F, int... Js> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Js...>) { (f(std::integral_constant<int, Js>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// open a one-bit gap at every set bit of f (ascending), in all three planes: the low bits of x[] move up past the gaps
// (a software bit deposit; f has few bits in steady state -- one new symbol per ~170 old ones at configs[1])
__device__ __forceinline__ void open_gaps(uint64_t x[3], uint64_t f)
{
	{	// first new symbol of the group, branch-free (most groups have none or one)
		const uint64_t lm = f ? (1ull << __builtin_ctzll(f)) - 1ull : ~0ull;   // no new symbol: everything stays
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) x[pl] = (x[pl] & lm) | ((x[pl] & ~lm) << 1);
		f &= f - 1;
	}
	while (__any(f != 0)) {
		if (f) {
			const uint64_t lm = (1ull << __builtin_ctzll(f)) - 1ull;   // bits below the new symbol
			f &= f - 1;
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) x[pl] = (x[pl] & lm) | ((x[pl] & ~lm) << 1);
		}
	}
}

// the rewritten index is read once and written once per round: nontemporal loads and stores of the leaf words (RB2_NT=0: plain).
// A/B on one box, r04: k_merge 0.900 -> 0.863 ms per launch, configs[1] 17.56 -> 17.9 Gsym/s.  The same hint on the per-string
// arrays (L, W, A, INS_E) made the job SLOWER (17.85 -> 17.5): they are written by one kernel and read by the next.
#ifndef RB2_NT
#define RB2_NT 1
#endif
#if RB2_NT
#define RB2_LDNT(p) __builtin_nontemporal_load(p)
#define RB2_STNT(v, p) __builtin_nontemporal_store(v, p)
#else
#define RB2_LDNT(p) (*(p))
#define RB2_STNT(v, p) (*(p) = (v))
#endif

// LDS words one wave of merge_window<.., GPL_, ..> needs: flags + three planes of new symbols + three planes of staged old groups
template <int GPL_> struct MergeLds { static constexpr int WG = 64 * GPL_, WORDS = WG + 3 * WG + 3 * (WG + 2) + 2; };

// FULL: the window holds WIN symbols (all but the last window of a piece) -- every position is valid.
// GPL_ = groups per lane: the window is 64 * GPL_ groups (dense merge: GPL; in-place leaf merge: 1, the leaf in the first row).
// INPLACE: the window IS one leaf with slack, rewritten where it lies (sparse rounds, leaves that receive many symbols): its old
// symbols are its own first groups, d.i0 is the piece position of its first symbol, every new symbol also gets its leaf slot (RKLEAF).
template <bool FULL, int GPL_, bool INPLACE, typename P> __device__ __forceinline__ void merge_window(const LeafDesc &d, uint64_t *lds, const int ln,
		const PoolView &oldp, const PoolView &newp, const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, uint16_t *RKREL, uint32_t *RKLEAF)
{
	constexpr int WG = 64 * GPL_, LPL = LEAFG / GPL_, WINS = WG * GSYM;      // groups per window, lanes per leaf, symbols per window
	// LDS layout.  A lane owns GPL_ consecutive groups of the window; arrays indexed by group are kept LANE-major -- group G lives at
	// (G % GPL_) * 64 + G / GPL_, so lane ln's w-th group is at w * 64 + ln and the lanes of a wave hit 64 different banks pairs.
	uint64_t *LF = lds, *LX = lds + WG, *LO = lds + 4 * WG;                // flags; planes of the new symbols (later: of the output); staged old groups, plane pl at LO + pl * (WG + 2)
	auto SX = [](uint32_t G) -> uint32_t { return GPL_ == 1 ? G : (G % GPL_) * 64u + G / GPL_; };
	auto LM = [](int w, int lane) -> int { return 64 * w + lane; };         // index of lane's w-th group
	const int nvalid = FULL ? WINS : d.nvalid, ni = d.ni;
	const uint32_t nold = (uint32_t)(nvalid - ni);              // old symbols consumed by this window
	const uint64_t G0 = INPLACE ? 0 : d.i0 >> 6;                // old group that holds the first of them
	const uint32_t sh0 = INPLACE ? 0u : (uint32_t)(d.i0 & 63);  // ... and its place in that group
	const uint32_t nwg = (sh0 + nold + 63) >> 6;                // groups of the old side they live in (<= WG + 1)

	// ---- 1. new symbols of this window, by output position (planes and "new here" flag); the old groups it draws from
#pragma unroll
	for (int w = 0; w < GPL_; ++w) {
		LF[ln + 64 * w] = 0;
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) LX[pl * WG + ln + 64 * w] = 0;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	const uint64_t *ob = (const uint64_t*)oldp.data + (INPLACE ? d.gl : (uint64_t)d.oleaf0) * LEAFW;
	uint64_t wa[GPL_][3], wt[3] = {0, 0, 0};
#pragma unroll
	for (int w = 0; w < GPL_; ++w) {
		const uint32_t k = (uint32_t)(ln + 64 * w);
		const uint64_t og = G0 + k;
		const uint64_t *q = ob + (og >> 4) * LEAFW + (og & 15);
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) { wa[w][pl] = 0; if (k < nwg) wa[w][pl] = RB2_LDNT(&q[pl * LEAFG]); }
	}
	if (ln == 0 && (uint32_t)WG < nwg) {
		const uint64_t og = G0 + WG;
		const uint64_t *q = ob + (og >> 4) * LEAFW + (og & 15);
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) wt[pl] = q[pl * LEAFG];
	}
	uint32_t p_first = 0, a_first = 0;                          // my first new symbol, kept for step 5
	for (int jj = ln; jj < ni; jj += 64) {
		const uint64_t e = INS_E[d.ins0 + jj];
		const uint32_t a = INS_A[d.ins0 + jj];
		const uint32_t p = (uint32_t)(e - d.i0) + (uint32_t)jj;   // final position E[q] + q, relative to the window
		if (jj == ln) { p_first = p; a_first = a; }
		const uint32_t ix = SX(p >> 6);
		const unsigned long long bit = 1ull << (p & 63);
		atomicOr((unsigned long long*)&LF[ix], bit);
		if (a & 1u) atomicOr((unsigned long long*)&LX[ix], bit);
		if (a & 2u) atomicOr((unsigned long long*)&LX[WG + ix], bit);
		if (a & 4u) atomicOr((unsigned long long*)&LX[2 * WG + ix], bit);
	}
#pragma unroll
	for (int w = 0; w < GPL_; ++w)
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) LO[pl * (WG + 2) + ln + 64 * w] = wa[w][pl];
	if (ln == 0) {
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) LO[pl * (WG + 2) + WG] = wt[pl];
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();

	// ---- 2. what does each lane consume
	uint64_t F[GPL_], VM[GPL_], out[GPL_][3];
	uint32_t non[GPL_], ntot = 0, ktot = 0, vtot = 0;
	const int p0 = ln * GSYM * GPL_;
#pragma unroll
	for (int w = 0; w < GPL_; ++w) {
		F[w] = LF[LM(w, ln)];                                   // bit i: position i holds a new symbol
		const int v = FULL ? GSYM : min(GSYM, max(0, nvalid - p0 - GSYM * w));
		VM[w] = FULL ? ~0ull : bits_below((uint32_t)v);         // the valid positions
		const uint32_t kin = (uint32_t)__popcll(F[w]);
		non[w] = (uint32_t)v - kin;
		ntot += non[w]; ktot += kin; vtot += (uint32_t)v;
	}
	const uint32_t sc2 = dpp_incl_add(ntot | ktot << 16);       // both prefix sums in one scan (each <= WINS < 2^16)
	{
		uint32_t op = sh0 + ((sc2 & 0xffffu) - ntot);             // first old symbol of this lane, in symbols of the stage
#pragma unroll
		for (int w = 0; w < GPL_; ++w) {
			const uint32_t k = op >> 6, sh = op & 63;
#pragma unroll
			for (int pl = 0; pl < 3; ++pl) {
				const uint64_t w0 = LO[pl * (WG + 2) + k], w1 = LO[pl * (WG + 2) + k + 1];   // k + 1 <= WG + 1
				out[w][pl] = (w0 >> sh) | ((w1 << 1) << (63 - sh));   // bits [op, op + 64) of the plane; what lies behind my non[w] bits is shifted out or masked below
			}
			op += non[w];
		}
	}

	// ---- 3. deal the old symbols to the not-new positions, add the new ones
#pragma unroll
	for (int w = 0; w < GPL_; ++w) {
		if (__any(F[w] != 0)) {
			if (__all(F[w] == VM[w])) { out[w][0] = out[w][1] = out[w][2] = 0; }   // nothing old in any lane's group (the first rounds on an empty index)
			else open_gaps(out[w], F[w]);
		}
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) out[w][pl] = (out[w][pl] & VM[w]) | LX[pl * WG + LM(w, ln)];   // (the gaps hold zeros)
	}

	// ---- 4. counts per lane -> prefix over the window -> LeafMeta of its leaves
	uint32_t c[6];
	{
		PlAcc A;
#pragma unroll
		for (int w = 0; w < GPL_; ++w) pl_acc(A, out[w][0], out[w][1], out[w][2], VM[w]);
		pl_finish(A, vtot, c);
	}
	const uint32_t e01 = c[0] | c[1] << 16, e23 = c[2] | c[3] << 16, e45 = c[4] | c[5] << 16;
	const uint32_t s01 = dpp_incl_add(e01), s23 = dpp_incl_add(e23), s45 = dpp_incl_add(e45);
	// publish my groups and my exclusive prefixes (the old-group stage is dead by now)
	uint32_t *LP = (uint32_t*)LO;
#pragma unroll
	for (int w = 0; w < GPL_; ++w)
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) LX[pl * WG + LM(w, ln)] = out[w][pl];
	LP[ln] = s01 - e01; LP[64 + ln] = s23 - e23; LP[128 + ln] = s45 - e45;   // LP[q * 64 + lane]
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();

	// ---- 5. leaf-relative rank of every new symbol, one per lane
	for (int jj = ln; jj < ni; jj += 64) {
		uint32_t p = p_first, a = a_first;
		if (jj != ln) {                                        // more than 64 new symbols in the window: read them again
			a = INS_A[d.ins0 + jj];
			p = (uint32_t)(INS_E[d.ins0 + jj] - d.i0) + (uint32_t)jj;
		}
		const uint32_t G = p >> 6, lo = G / GPL_, wi = G - lo * GPL_;
		const uint32_t bl = (p >> LEAF_SH) * LPL;                // first lane of its leaf
		const uint32_t sh = (a & 1) * 16;
		uint32_t r = ((LP[64 * (a >> 1) + lo] >> sh) & 0xffffu) - ((LP[64 * (a >> 1) + bl] >> sh) & 0xffffu);
#pragma unroll
		for (int w = 0; w < GPL_; ++w) {
			const uint64_t m = (uint32_t)w < wi ? ~0ull : ((uint32_t)w == wi ? (1ull << (p & 63)) - 1ull : 0ull);
			const int ix = LM(w, (int)lo);
			r += (uint32_t)__popcll(pl_eq(LX[ix], LX[WG + ix], LX[2 * WG + ix], a) & m);
		}
		RKREL[d.ins0 + jj] = (uint16_t)r;
		if (INPLACE) RKLEAF[d.ins0 + jj] = (uint32_t)d.gl;
	}
	if (!INPLACE && (ln % LPL) == LPL - 1 && (ln / LPL) * LEAF < nvalid) {   // last lane of a leaf that exists (in place: the directory is kept by dir_add)
		const uint32_t bl = (uint32_t)(ln / LPL) * LPL;
		const uint32_t t01 = s01 - LP[bl], t23 = s23 - LP[64 + bl], t45 = s45 - LP[128 + bl];
		LeafMeta m;
		m.c[0] = (uint16_t)t01; m.c[1] = (uint16_t)(t01 >> 16); m.c[2] = (uint16_t)t23; m.c[3] = (uint16_t)(t23 >> 16);
		m.c[4] = (uint16_t)t45; m.c[5] = (uint16_t)(t45 >> 16);
		m.npre = 0;
		m.n = (uint16_t)((t01 & 0xffffu) + (t01 >> 16) + (t23 & 0xffffu) + (t23 >> 16) + (t45 & 0xffffu) + (t45 >> 16));
		newp.own[d.gl + ln / LPL] = m;                          // own counts + fill; k_meta_sb turns them into prefixes
	}
#pragma unroll
	for (int w = 0; w < GPL_; ++w) {
		const uint32_t G = (uint32_t)(ln * GPL_ + w);
		if (INPLACE && G >= (uint32_t)LEAFG) continue;          // in place the window is ONE leaf
		uint64_t *dst = (uint64_t*)newp.data + (d.gl + (G >> 4)) * LEAFW + (G & 15);
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) RB2_STNT(out[w][pl], &dst[pl * LEAFG]);   // leaves past the end of the piece are padding slots of the same piece
	}
}

template <bool STRIDE, typename P = uint64_t> __global__ __launch_bounds__(256) void k_merge(const Ctl *ctl, const LeafDesc *__restrict__ LD, PoolView oldp, PoolView newp,
		const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, uint16_t *RKREL)
{
	__shared__ __align__(16) uint64_t lds[MW][MergeLds<GPL>::WORDS];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ln = lane_id();
	// one window per wave; a rank of a sharded index launches fewer waves than the upper bound of its windows (the host does not
	// know the rank's share of the batch) and a wave then takes more than one: grid stride over the windows.  The first window's
	// work order is loaded together with the window count (LD holds an entry for every window a grid can name).
	uint64_t gw = (uint64_t)blockIdx.x * MW + wv;
	LeafDesc d = LD[gw];
	const uint64_t nwin = ctl->wf0[NR];
	for (; gw < nwin; gw += (uint64_t)gridDim.x * MW, d = LD[gw < nwin ? gw : 0]) {
		if (d.nvalid == WIN) merge_window<true, GPL, false, P>(d, lds[wv], ln, oldp, newp, INS_E, INS_A, RKREL, nullptr);
		else merge_window<false, GPL, false, P>(d, lds[wv], ln, oldp, newp, INS_E, INS_A, RKREL, nullptr);
		if (!STRIDE) return;                                    // (one GPU: the grid covers every window; no loop, no extra registers)
		if (gw + (uint64_t)gridDim.x * MW < nwin) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // the wave's LDS arrays are reused
	}
}

// Sparse rounds keep the rank directory current themselves (the dense rounds rebuild it, k_meta_sb): the directory of a
// superblock holds OWN counts by rows (dir_row, rb2_device.h), so a leaf that received symbols adds them to its own entries and
// to the superblock total, and nothing behind it moves.  What a round costs is then proportional to the leaves it touches; only
// the prefix over the superblock totals (k_sbscan*) still reads every superblock.  (rope.c:139-146: the counts along the path.)
// Atomics although a row entry has one writer (the total has several): a 2-byte store is a partial write the memory side has to
// merge, and measured slower.  They are issued as ONE instruction, and before the wave starts shifting words.
// Roles (the lane's place in its row): 0 = the fill, 1-6 = the own count of symbol role - 1, 7-9 = the three packed words of the
// superblock total (its own small array: 16 bytes per superblock stay in cache, the directory blocks do not).  cnt = what the lane adds.
__device__ __forceinline__ void dir_commit(const PoolView &pool, SbTot *sbtot, uint64_t gl, int role, uint32_t cnt)
{
	const uint32_t k = (uint32_t)(gl % SB);
	uint32_t *ptr = (uint32_t*)dir_row(pool, gl / SB, 0) + (uint32_t)role * (SB / 2) + (k >> 1);
	if (role >= 7) ptr = (uint32_t*)&sbtot[gl / SB] + (role - 7);
	const uint32_t val = role < 7 ? cnt << ((k & 1) * 16) : cnt;
	if (role < 10 && val) atomicAdd(ptr, val);
}
// ... any number of new symbols, already counted: d01 | d23 | d45 packed like LeafMeta::c, same values in all lanes that act
__device__ __forceinline__ void dir_add_packed(const PoolView &pool, SbTot *sbtot, uint64_t gl, int role, uint32_t d01, uint32_t d23, uint32_t d45)
{
	const int s = role - 1;
	const uint32_t dw = s < 2 ? d01 : (s < 4 ? d23 : d45);
	uint32_t cnt = (dw >> ((uint32_t)(s & 1) * 16)) & 0xffffu;
	if (role == 0) cnt = (d01 & 0xffffu) + (d01 >> 16) + (d23 & 0xffffu) + (d23 >> 16) + (d45 & 0xffffu) + (d45 >> 16);
	if (role >= 7) cnt = role == 7 ? d01 : (role == 8 ? d23 : d45);
	dir_commit(pool, sbtot, gl, role, cnt);
}

// ---------------------------------------------------------------------------------------------
// k_merge_leaf: sparse rounds.  ONE LEAF PER DPP ROW -- a wave inserts into four touched leaves at once (work orders appended by
// k_part_sparse, any order), each rewritten in place: rope_insert_run's descent ends here (rope.c:136-141) and this is
// rle_insert_cached (rle.c:10-89) for all the inserts a leaf receives this round at once.  Untouched leaves keep their bytes.
// No LDS: lane g of the row keeps the three plane words of group g in registers; per new symbol (ascending position, so earlier
// ones are already in place) one plane compare + row sum gives its rank, one shift with a DPP carry from the lane below opens the
// gap; only the groups from the first changed one on are stored.  The row's j-th insert reaches its lanes through a DPP row
// broadcast; a leaf that receives more than LTURN symbols (hot spots: the normal case of a sparse round is one or two) takes them
// in turns of LTURN.  The kernel is persistent and software-pipelined: while a wave shifts the words of one quad of leaves, the
// words and insert records of its next quad are in flight and the work orders of the quad after that are being fetched.
// (Rounds 2-3: 512-byte leaves of 3-bit fields, one WAVE per leaf: 372 VALU per four leaves and 0.95 KB per insert.)
// A round that set ctl->overflow is void.
// ---------------------------------------------------------------------------------------------
constexpr int LROWS = 4;                    // leaves per wave step
constexpr int LTURN = 8;                    // inserts a row takes per turn (lanes 0 .. LTURN - 1 of the row hold them)
struct RowOrd { uint32_t gl, ins0, i0, ni; };                               // the row's work order (SpOrd, rb2_device.h), ni = 0: none
struct RowJob { uint64_t w[3]; uint32_t pj, aj; };

__device__ __forceinline__ void row_ord_load(const SpOrd *LD, uint64_t g, uint32_t nwork, int ln, RowOrd &o)
{
	const uint64_t q = g + (uint32_t)(ln >> 4);
	const bool ok = q < nwork;
	const uint4 a = *(const uint4*)(LD + (ok ? q : g));         // behind the end: a duplicate of the quad's first order, loaded but never run
	o.gl = a.x; o.ins0 = a.y; o.i0 = a.z;
	o.ni = ok ? (a.w & 0xffffu) : 0u;
}
template <typename P> __device__ __forceinline__ void row_job_load(const RowOrd &o, const int g, const PoolView &pool, const P *INS_E, const uint8_t *INS_A, RowJob &J)
{
	const uint64_t *lw = (const uint64_t*)pool.data + (uint64_t)o.gl * LEAFW + g;
#pragma unroll
	for (int pl = 0; pl < 3; ++pl) J.w[pl] = RB2_LDNT(&lw[pl * LEAFG]);   // (nontemporal, like the stores: see the end of the loop)
	// no branch and no use of a loaded value in here: the loads of the quad are to be in flight together (lanes >= ni load the
	// row's last insert again; they never use it)
	const uint64_t q = (uint64_t)o.ins0 + (uint32_t)min(g, max((int)o.ni, 1) - 1);
	J.aj = INS_A[q]; J.pj = sizeof(P) == 4 ? ((const uint32_t*)INS_E)[q] : ((const uint32_t*)INS_E)[2 * q];   // low half: positions inside a leaf need no more
}

template <typename P = uint64_t> __global__ __launch_bounds__(256, 7) void k_merge_leaf(const Ctl *ctl, const SpOrd *__restrict__ LD, PoolView pool,
		const P *INS_E, const uint8_t *INS_A /* not __restrict__: the loads are to stay where they are issued */, uint16_t *RKREL, uint32_t *RKLEAF, SbTot *sbtot)
{
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ln = lane_id(), g = ln & 15;
	// the round's work orders are WLC lists (Ctl::wcnt): wave gw takes list gw % WLC and walks it with the stride of the waves that
	// share it -- one counter to read per wave, and no search for "the q-th order of the round"
	const uint32_t gw = blockIdx.x * MW + (uint32_t)wv, per = gridDim.x * MW / WLC;   // (the host launches at least WLC waves)
	const uint32_t lc = gw % WLC, wi = gw / WLC;
	const uint64_t stride = (uint64_t)per * LROWS;
	uint64_t g0 = (uint64_t)wi * LROWS;
	const uint32_t nwork = ctl->wcnt[lc * WLS];
	LD += (uint64_t)lc * ctl->wstride;
	if (ctl->overflow || wi >= per || g0 >= nwork) return;
	RowOrd o, on, onn;
	RowJob J, Jn;
	row_ord_load(LD, g0, nwork, ln, on);
	row_ord_load(LD, g0 + stride < nwork ? g0 + stride : g0, nwork, ln, onn);
	row_job_load<P>(on, g, pool, INS_E, INS_A, Jn);
	for (;;) {
		o = on; J = Jn; on = onn;
		const uint64_t g1 = g0 + stride, g2 = g1 + stride;
		const bool more = g1 < nwork;
		if (more) {
			row_job_load<P>(on, g, pool, INS_E, INS_A, Jn);       // next quad: in flight while this one is worked on
			row_ord_load(LD, g2 < nwork ? g2 : g1, nwork, ln, onn);
		}
		asm volatile("" ::: "memory");
		uint64_t w0 = J.w[0], w1 = J.w[1], w2 = J.w[2];
		uint32_t pjr = J.pj, aj = J.aj;
		const uint32_t nimax = max(max((uint32_t)__builtin_amdgcn_readlane((int)o.ni, 0), (uint32_t)__builtin_amdgcn_readlane((int)o.ni, 16)),
				max((uint32_t)__builtin_amdgcn_readlane((int)o.ni, 32), (uint32_t)__builtin_amdgcn_readlane((int)o.ni, 48)));
		uint32_t pg0 = 0;                                         // first group of the row that changes
		for (uint32_t c0 = 0; c0 < nimax; c0 += LTURN) {            // turns of LTURN inserts per row (one turn, normally)
			if (c0) {                                               // (rare) the row's next inserts
				const uint64_t q = (uint64_t)o.ins0 + min(c0 + (uint32_t)g, max(o.ni, 1u) - 1u);
				aj = INS_A[q]; pjr = sizeof(P) == 4 ? ((const uint32_t*)INS_E)[q] : ((const uint32_t*)INS_E)[2 * q];
			}
			const uint32_t nic = o.ni > c0 ? min(o.ni - c0, (uint32_t)LTURN) : 0u;   // what my row inserts in this turn (row-uniform)
			const uint32_t pj = pjr - o.i0 + c0 + (uint32_t)g;      // final position E[q] + q inside the leaf (lanes >= nic: unused)
			const uint32_t ncmax = min(nimax - c0, (uint32_t)LTURN);
			const bool mine = (uint32_t)g < nic;
			// what a leaf receives is known before the first symbol is placed: the directory atomics go out first and are under way
			// while the wave shifts words
			{
				const uint32_t rsh = (uint32_t)(ln & 48);
				uint32_t cs[6];
#pragma unroll
				for (int s = 0; s < 6; ++s) cs[s] = (uint32_t)__popc((uint32_t)(__ballot(mine && aj == (uint32_t)s) >> rsh) & 0xffffu);
				dir_add_packed(pool, sbtot, o.gl, nic ? g : 16, cs[0] | cs[1] << 16, cs[2] | cs[3] << 16, cs[4] | cs[5] << 16);
				if (mine) RKLEAF[(uint64_t)o.ins0 + c0 + g] = o.gl;
			}
			if (c0 == 0) pg0 = row_share<0>(pj) >> 6;
			uint32_t myrank = 0;
			static_for<LTURN>([&](auto jc) {
				constexpr int j = decltype(jc)::value;
				if ((uint32_t)j >= ncmax) return;                     // wave-uniform
				const uint32_t p = row_share<j>(pj), a = row_share<j>(aj);
				const bool act = (uint32_t)j < nic;
				const uint32_t pg = p >> 6, pb = p & 63;
				const uint64_t below = (1ull << pb) - 1ull;
				const uint64_t msk = (uint32_t)g < pg ? ~0ull : ((uint32_t)g == pg ? below : 0ull);
				const uint32_t inc = row_incl_add((uint32_t)__popcll(pl_eq(w0, w1, w2, a) & msk));
				const uint32_t r = row_share<15>(inc);                // a's in front of p, leaf as it is now
				if (act && g == j) myrank = r;
				const uint32_t top = (uint32_t)(w0 >> 63) | (uint32_t)(w1 >> 63) << 1 | (uint32_t)(w2 >> 63) << 2;
				const uint32_t cprev = row_prev(top);                 // top symbol of the group below moves up
				if (act) {
					if ((uint32_t)g > pg) {
						w0 = (w0 << 1) | (cprev & 1u); w1 = (w1 << 1) | ((cprev >> 1) & 1u); w2 = (w2 << 1) | (cprev >> 2);
					} else if ((uint32_t)g == pg) {
						w0 = (w0 & below) | (uint64_t)(a & 1u) << pb | ((w0 & ~below) << 1);
						w1 = (w1 & below) | (uint64_t)((a >> 1) & 1u) << pb | ((w1 & ~below) << 1);
						w2 = (w2 & below) | (uint64_t)(a >> 2) << pb | ((w2 & ~below) << 1);
					}
				}
			});
			if (mine) RKREL[(uint64_t)o.ins0 + c0 + g] = (uint16_t)myrank;
		}
		// The WHOLE leaf goes back, three full 128-byte lines, with nontemporal stores behind nontemporal loads: the leaf streams through
		// once and nothing of it has to wait in L2 for a partial line to be merged.  (Rounds 2-4 stored only the groups from the first
		// changed one on, counting on the lines the load had left in L2: 1 M x 10 kbp 3.09-3.19 s, this way 2.72-2.75 s on one box.  Each
		// half alone is no gain: whole lines with plain accesses 3.09, nontemporal accesses with partial lines 3.31-3.39.)
		(void)pg0;
		if (o.ni) {
			uint64_t *lw = (uint64_t*)pool.data + (uint64_t)o.gl * LEAFW + g;
			RB2_STNT(w0, &lw[0]); RB2_STNT(w1, &lw[LEAFG]); RB2_STNT(w2, &lw[2 * LEAFG]);
		}
		if (!more) return;
		g0 = g1;
	}
}

// ---------------------------------------------------------------------------------------------
// k_export: chunks [c0, c0+nc) of XCHUNK symbols of one sub-rope -> run-length bytes of ropebwt2's 43+3 codec, one
// byte per run of <= 15 symbols (rle_enc1's 1-byte form, rle.h:55-57), runs cut at chunk ends.  A sub-rope in the dense layout is
// a flat array of symbols (every leaf but the last is full): chunk c of the piece is its leaf c.
// Output: slot i of `dst` (stride XCHUNK) holds nb[i] bytes.  Not on the hot path.
// ---------------------------------------------------------------------------------------------

constexpr int XCHUNK = LEAF;                // symbols per export chunk: 16 per lane
struct ExportLds { uint8_t outb[XCHUNK + 16]; };

__device__ __forceinline__ uint32_t byte_of(const uint32_t w[4], int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }
__device__ __forceinline__ uint32_t bits4_to_bytes(uint32_t t)     // four bits -> bit 0 of four bytes
{
	return ((t & 0xfu) * 0x00204081u) & 0x01010101u;
}

__global__ __launch_bounds__(256) void k_export(PoolView pv, uint64_t leaf0, uint64_t n_syms, uint64_t c0, uint32_t nc, uint8_t *dst, uint16_t *nb)
{
	__shared__ __align__(16) ExportLds lds[MW];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	ExportLds &L = lds[wv];
	const int ln = lane_id();
	const uint32_t li = blockIdx.x * MW + wv;
	if (li >= nc) return;
	const uint64_t s0 = (c0 + li) * (uint64_t)XCHUNK;           // first symbol of the chunk
	const int nvalid = (int)min((uint64_t)XCHUNK, n_syms - s0);
	const int p0 = ln * 16;
	const int myvalid = min(16, max(0, nvalid - p0));
	const uint32_t vmask = (1u << myvalid) - 1u;
	uint32_t b0 = 0, b1 = 0, b2 = 0;                            // my 16 symbols, one bit per symbol and plane
	if (myvalid) {
		const uint64_t *lw = leaf_words(pv.data, leaf0 + c0 + li) + (ln >> 2);
		const uint32_t sh = (uint32_t)(ln & 3) * 16;
		b0 = (uint32_t)(lw[0] >> sh) & 0xffffu; b1 = (uint32_t)(lw[LEAFG] >> sh) & 0xffffu; b2 = (uint32_t)(lw[2 * LEAFG] >> sh) & 0xffffu;
	}
	uint32_t pw[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) pw[k] = bits4_to_bytes(b0 >> (4 * k)) | bits4_to_bytes(b1 >> (4 * k)) << 1 | bits4_to_bytes(b2 >> (4 * k)) << 2;
#pragma unroll
	for (int i = 0; i < 16; ++i) if (i >= myvalid) pw[i >> 2] |= 0xffu << ((i & 3) * 8);      // past the end: never equal to a symbol
	((uint4*)L.outb)[ln] = make_uint4(0, 0, 0, 0);
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	// run heads by packed neighbour compare
	uint32_t hm = 0;                                           // bit i: a run starts at my position i
	uint32_t prevw = dpp_prev_lane(pw[3]);                     // NB: cross-lane reads stay outside of lane-dependent conditionals
	if (ln == 0) prevw = 0xff000000u;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const uint32_t ps = __builtin_amdgcn_alignbyte(pw[k], k == 0 ? prevw : pw[k - 1], 3);   // my symbols shifted by one position
		const uint32_t x = pw[k] ^ ps;
		const uint32_t nz = ((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7) & 0x01010101u;       // 1 per differing byte
		hm |= ((nz * 0x01020408u) >> 24) << (4 * k);
	}
	hm &= vmask;
	const uint32_t pm = dpp_prev_lane(hm);
	uint32_t cov = pm | (hm << 16);
	cov |= cov << 1; cov |= cov << 2; cov |= cov << 4; cov |= cov << 7;   // bit set: a run start within the 14 positions before
	const bool short_runs = ((cov >> 16) & vmask) == vmask;
	uint32_t nbytes;
	if (__all(short_runs)) {
		const uint32_t nh = __popc(hm);
		const uint32_t hinc = dpp_incl_add(nh);
		const uint32_t hb = hinc - nh;
		nbytes = lane63(hinc);
		const uint32_t nvnext = dpp_next_lane((uint32_t)myvalid), hmnext = dpp_next_lane(hm);
		const uint32_t tail = nvnext ? (hmnext ? (uint32_t)__builtin_ctz(hmnext) : nvnext) : 0u;   // symbols of my last run living in the next lane
		const uint32_t lastlen = (uint32_t)myvalid + tail;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const uint32_t t = hm >> (i + 1);
			const uint32_t len = t ? (uint32_t)__builtin_ctz(t) + 1u : lastlen - (uint32_t)i;
			const uint32_t idx = (hm >> i & 1u) ? hb + __popc(hm & ((1u << i) - 1u)) : (uint32_t)XCHUNK;   // non-heads go to the dump slot
			L.outb[idx] = (uint8_t)(len << 3 | byte_of(pw, i));
		}
	} else {
		// runs longer than 15 symbols: a byte boundary every 15 symbols of a run (plain, shuffle-based form)
		const uint32_t prevsym = prevw >> 24;
		int lastnat = -1;
		for (int i = 0; i < 16; ++i) if (hm >> i & 1u) lastnat = p0 + i;
		const int incmax = wave_incl_max(lastnat);
		int rs = __shfl_up(incmax, 1);                             // start of the run open at p0-1
		if (ln == 0) rs = 0;
		int lh = ln == 0 ? 0 : rs + (p0 - 1 - rs) / 15 * 15;       // last byte boundary before p0
		int hc = 0;
		{
			int r = rs;
			for (int i = 0; i < 16; ++i) if (i < myvalid) {
				const int p = p0 + i;
				if (hm >> i & 1u) r = p;
				hc += ((hm >> i & 1u) || (p - r) % 15 == 0);
			}
		}
		const int hinc = wave_incl_add(hc);
		const int hb = hinc - hc;
		nbytes = (uint32_t)__shfl(hinc, 63);
		{
			uint32_t pv2 = prevsym; int r = rs, seen = 0;
			for (int i = 0; i < 16; ++i) if (i < myvalid) {
				const int p = p0 + i;
				if (hm >> i & 1u) r = p;
				if ((hm >> i & 1u) || (p - r) % 15 == 0) {
					if (p != 0) L.outb[hb + seen - 1] = (uint8_t)((p - lh) << 3 | pv2);
					lh = p; ++seen;
				}
				pv2 = byte_of(pw, i);
			}
			if (myvalid > 0 && p0 + myvalid == nvalid) L.outb[nbytes - 1] = (uint8_t)((nvalid - lh) << 3 | pv2);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	if (ln == 0) nb[li] = (uint16_t)nbytes;
	((uint4*)(dst + (uint64_t)li * XCHUNK))[ln] = ((const uint4*)L.outb)[ln];
}

// exclusive prefix of the chunks' byte counts (one block); off[n] = total
__global__ __launch_bounds__(SCHUNK) void k_xscan(const uint16_t *nb, uint32_t n, uint64_t *off)
{
	__shared__ uint64_t s_w[16];
	uint64_t run = 0;
	for (uint32_t i0 = 0; i0 < n; i0 += SCHUNK) {
		const uint32_t i = i0 + threadIdx.x;
		uint64_t tot;
		const uint64_t ex = block_excl_add<uint64_t>(i < n ? (uint64_t)nb[i] : 0ull, s_w, &tot);
		if (i < n) off[i] = run + ex;
		run += tot;
	}
	if (threadIdx.x == 0) off[n] = run;
}

// pack the chunks' run bytes back to back (one wave per chunk, byte-wise: the destinations are unaligned)
__global__ __launch_bounds__(256) void k_xcompact(const uint8_t *stage, const uint16_t *nb, const uint64_t *off, uint32_t nc, uint8_t *dst)
{
	const uint32_t li = blockIdx.x * MW + (threadIdx.x >> 6);
	if (li >= nc) return;
	const uint32_t n = nb[li];
	const uint8_t *src = stage + (uint64_t)li * XCHUNK;
	uint8_t *o = dst + off[li];
	for (uint32_t i = lane_id(); i < n; i += 64) o[i] = src[i];
}

} // namespace rb2
