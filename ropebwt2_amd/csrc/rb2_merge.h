// rb2_merge.h -- k_merge: rank + positional insert over the packed leaves of every sub-rope;
//                k_export: packed leaves -> ropebwt2's run-length bytes (only when the BWT leaves the GPU).
//
// Reference semantics: rope_insert_run (rope.c:114-148) -> rle_insert_cached (rle.c:10-89): put
// rl copies of symbol a in front of position x and return the number of a's before x.  The
// reference does this one run at a time through a B+ tree of run-length leaves; here one launch
// rewrites every sub-rope side -> side^1 as a merge of two sorted sequences (old symbols, new
// symbols).  In HBM a sub-rope is a flat array of 3-bit symbols (a leaf = LEAF symbols = LEAFB
// bytes, symbol i in bits 3(i%21).. of 64-bit word i/21), so the merge is a pure stream: no run
// decoding, no re-encoding, no length-dependent paths.  Run-length coding is applied once, by
// k_export, when the host asks for the ropes (mr_sync_host -> .fmd/.fmr writers).
//
// k_merge work decomposition: ONE WAVE PER OUTPUT WINDOW of WPL consecutive leaves, four
// independent waves per block, no block-level barrier anywhere.  Lane l owns output positions
// [SPW*WPL*l, SPW*WPL*(l+1)) = WPL consecutive 64-bit words (64/WPL lanes per leaf):
//   1. the new symbols of the window are OR-ed into a position-indexed nibble array in LDS as 8|a
//      (bit 3 doubles as the "this position is new" flag), one LDS atomic per new symbol; the old
//      words the window draws from are loaded at the same time and staged in LDS
//   2. one packed wave prefix sum (not-new count | new count) -> first old symbol each lane
//      consumes; the old symbols of each of its words are an unaligned 64-bit window of the stage
//   3. expand: open one 3-bit gap per new symbol (wave-uniform loop, 1-2 trips in steady state;
//      16-step branch-free deal when some lane has many); the new symbols are already in place
//   4. symbol counts per lane from three bit planes + five popcounts, three packed scans -> new
//      LeafMeta of each leaf of the window
//   5. RKREL: every new symbol gets the number of equal symbols before it INSIDE its leaf, one new
//      symbol per lane (prefix of the owning lane + a masked compare of its words, both read back
//      from LDS); k_advance adds the directory prefix of the new sub-rope to obtain the reference's
//      return value of rope_insert_run.
#pragma once
#include "rb2_device.h"

namespace rb2 {

#ifndef RB2_KMAX
#define RB2_KMAX 5
#endif
#ifndef RB2_LO_PAD
#define RB2_LO_PAD 0            // LDS layout experiments of merge_window (see there)
#endif
#ifndef RB2_LX_LANEMAJOR
#define RB2_LX_LANEMAJOR 1
#endif
constexpr int NXW = 64 * WPL;               // words per window (dense merge)

__device__ __forceinline__ uint64_t nib_eq(uint64_t w, uint32_t a)   // bit 3i set: symbol i of w == a
{
	const uint64_t x = w ^ (a * MLOW);
	return ~(x | x >> 1 | x >> 2) & MLOW;                       // the three bits of a field, not a bit of its neighbour
}

// FULL: the window holds WIN symbols (all but the last window of a piece) -- every position is valid.
// WPL_ = words per lane: the window is WPL_ consecutive leaves (dense merge: WPL; in-place leaf merge: 1).
// INPLACE: the window IS one leaf with slack, rewritten where it lies (sparse rounds): its old symbols are its own
// first words, d.i0 is the piece position of its first symbol, every new symbol also gets its leaf slot (RKLEAF).
template <bool FULL, int WPL_, bool INPLACE> __device__ __forceinline__ void merge_window(const LeafDesc &d, uint64_t *LX, uint32_t *LF, uint64_t *LO, const int ln,
		const PoolView &oldp, const PoolView &newp, const uint64_t *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, uint16_t *RKREL, uint32_t *RKLEAF)
{
	constexpr int WPL = WPL_, NXW = 64 * WPL_, LPW = 64 / WPL_, WIN = WPL_ * LEAF;   // shadow the dense constants
	// LDS layout.  A lane owns WPL consecutive words of the window; laid out position-major (word pw at index pw) the lanes of a
	// wave touch addresses 8 * WPL bytes apart: with WPL = 4 only 8 of the 64 banks, a 4-way conflict on every access (PMC, round 3:
	// SQ_LDS_BANK_CONFLICT = 60 % of SQ_LDS_IDX_ACTIVE in k_merge).  LX, LF and the prefix table LP are therefore kept LANE-major --
	// word pw lives at (pw % WPL) * 64 + pw / WPL, so lane ln's w-th word is at w * 64 + ln --, the stage of old words LO, which
	// is read at data-dependent offsets of about WPL * ln, gets one pad word per 32.  WPL = 1 (in-place leaf merge): both are the identity.
	auto SX = [](uint32_t pw) -> uint32_t { return (WPL == 1 || !RB2_LX_LANEMAJOR) ? pw : (pw % WPL) * 64u + pw / WPL; };
	auto LM = [](int w, int lane) -> int { return (WPL == 1 || !RB2_LX_LANEMAJOR) ? WPL * lane + w : 64 * w + lane; };   // index of lane's w-th word
	auto SO = [](uint32_t i) -> uint32_t { return (WPL == 1 || !RB2_LO_PAD) ? i : i + (i >> 5); };
	const int nvalid = FULL ? WIN : d.nvalid, ni = d.ni;
	const uint32_t nold = (uint32_t)(nvalid - ni);              // old symbols consumed by this window
	const uint64_t w0i = INPLACE ? 0 : d.i0 / SPW;              // old word that holds the first of them
	const uint32_t sh0 = INPLACE ? 0u : (uint32_t)(d.i0 - w0i * SPW);   // ... and its place in that word
	const uint32_t nw = (sh0 + nold + SPW - 1) / SPW;           // words of the old side they live in (<= NXW + 1)

	// ---- 1. new symbols of this window, by output position (symbol and "new here" flag); the old words it draws from
#pragma unroll
	for (int w = 0; w < WPL; ++w) { LX[ln + 64 * w] = 0; LF[ln + 64 * w] = 0; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	const uint64_t *ob = (const uint64_t*)oldp.data + (INPLACE ? d.gl * LEAFW : (uint64_t)d.oleaf0 * LEAFW + w0i);
	uint64_t wa[WPL], wt = 0;
#pragma unroll
	for (int w = 0; w < WPL; ++w) { wa[w] = 0; if ((uint32_t)(ln + 64 * w) < nw) wa[w] = ob[ln + 64 * w]; }
	if (ln < 2 && (uint32_t)(NXW + ln) < nw) wt = ob[NXW + ln];
	uint32_t p_first = 0, a_first = 0;                          // my first new symbol, kept for step 5
	for (int jj = ln; jj < ni; jj += 64) {
		const uint64_t e = INS_E[d.ins0 + jj];
		const uint64_t a = INS_A[d.ins0 + jj];
		const uint32_t p = (uint32_t)(e - d.i0) + (uint32_t)jj;   // final position E[q] + q, relative to the window
		if (jj == ln) { p_first = p; a_first = (uint32_t)a; }
		const uint32_t pw = p / SPW, ps = (p - pw * SPW) * SBITS;
		const uint64_t sv = a << ps;                              // 32-bit LDS atomics: a 3-bit field may straddle bit 32
		uint32_t *x32 = (uint32_t*)LX + 2 * SX(pw);
		if ((uint32_t)sv) atomicOr(x32, (uint32_t)sv);
		if ((uint32_t)(sv >> 32)) atomicOr(x32 + 1, (uint32_t)(sv >> 32));
		atomicOr(LF + SX(pw), 1u << (p - pw * SPW));               // flags: one bit per position, 21 per word
	}
#pragma unroll
	for (int w = 0; w < WPL; ++w) LO[SO((uint32_t)(ln + 64 * w))] = wa[w];
	if (ln < 2) LO[SO((uint32_t)(NXW + ln))] = wt;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();

	// ---- 2. what does each lane consume
	uint64_t X[WPL], VM[WPL];
	uint32_t F[WPL];
	uint32_t kin[WPL], non[WPL], ntot = 0, ktot = 0, vtot = 0;
	const int p0 = ln * SPW * WPL;
#pragma unroll
	for (int w = 0; w < WPL; ++w) {
		X[w] = LX[LM(w, ln)];                                   // = SX(WPL * ln + w)
		F[w] = LF[LM(w, ln)];                                   // bit i: position i holds a new symbol
		const int v = FULL ? SPW : min(SPW, max(0, nvalid - p0 - SPW * w));
		VM[w] = FULL ? MALL : nib_below((uint32_t)v);           // all bits of the valid positions
		kin[w] = (uint32_t)__popc(F[w]);
		non[w] = (uint32_t)v - kin[w];
		ntot += non[w]; ktot += kin[w]; vtot += (uint32_t)v;
	}
	const uint32_t sc2 = dpp_incl_add(ntot | ktot << 16);       // both prefix sums in one scan (each <= WIN < 2^16)
	uint64_t out[WPL];
	{
		const uint32_t op = sh0 + ((sc2 & 0xffffu) - ntot);       // first old symbol of this lane, in symbols of LO[]
		uint32_t k = op / SPW, sh = (op - k * SPW) * SBITS;
#pragma unroll
		for (int w = 0; w < WPL; ++w) {
			const uint64_t w0 = LO[SO(k)], w1 = LO[SO(k + 1)];   // k + 1 <= NXW + 1
			out[w] = ((w0 >> sh) | (w1 << (63 - sh))) & MALL;     // 63 payload bits per word; sh == 0: the second term lands on bit 63
			sh += SBITS * non[w];                                  // non <= SPW: at most one word further
			if (sh >= 63) { sh -= 63; ++k; }
		}
	}

	// ---- 3. deal the old symbols to the not-new positions
	uint32_t kmax = kin[0];
#pragma unroll
	for (int w = 1; w < WPL; ++w) kmax = max(kmax, kin[w]);
	if (!__any(kmax > RB2_KMAX)) {
		// steady state: few new symbols per word.  Open one gap per new symbol, in ascending position.
		// One loop per word index: its trip count is the largest number of new symbols any lane has in THAT word.
#pragma unroll
		for (int w = 0; w < WPL; ++w) {
			uint32_t f = F[w];
			{	// first new symbol of the word, branch-free (most words have none or one)
				const uint64_t lm = (1ull << (f ? SBITS * __builtin_ctz(f) : 63)) - 1ull;   // no new symbol: all 63 payload bits stay
				out[w] = (out[w] & lm) | ((out[w] & ~lm) << SBITS);
				f &= f - 1;
			}
			while (__any(f != 0)) {
				if (f) {
					const uint64_t lm = (1ull << (SBITS * __builtin_ctz(f))) - 1ull;   // bits below the new symbol
					f &= f - 1;
					out[w] = (out[w] & lm) | ((out[w] & ~lm) << SBITS);
				}
			}
		}
#pragma unroll
		for (int w = 0; w < WPL; ++w) out[w] = (out[w] & VM[w]) | X[w];
	} else {
#pragma unroll
		for (int w = 0; w < WPL; ++w) {
			const int v = FULL ? SPW : min(SPW, max(0, nvalid - p0 - SPW * w));
			const uint32_t G = ~F[w] & ((1u << v) - 1u);            // bit i: position i takes an old symbol
			uint64_t o = 0, old = out[w];
#pragma unroll
			for (int i = 0; i < SPW; ++i) {
				const uint64_t nm = 0ull - (uint64_t)((G >> i) & 1u);    // all ones: old symbol here
				o |= (old & nm & 7ull) << (SBITS * i);
				old >>= (nm & SBITS);
			}
			out[w] = o | X[w];
		}
	}

	// ---- 4. counts per lane -> prefix over the window -> LeafMeta of its leaves
	uint32_t c[6];
	{
		NibAcc A;
#pragma unroll
		for (int w = 0; w < WPL; ++w) nib_acc(A, out[w], VM[w] & MLOW);
		nib_finish(A, vtot, c);
	}
	const uint32_t e01 = c[0] | c[1] << 16, e23 = c[2] | c[3] << 16, e45 = c[4] | c[5] << 16;
	const uint32_t s01 = dpp_incl_add(e01), s23 = dpp_incl_add(e23), s45 = dpp_incl_add(e45);
	// publish my words and my exclusive prefixes (the old-word stage is dead by now)
	uint32_t *LP = (uint32_t*)LO;
#pragma unroll
	for (int w = 0; w < WPL; ++w) LX[LM(w, ln)] = out[w];
	LP[ln] = s01 - e01; LP[64 + ln] = s23 - e23; LP[128 + ln] = s45 - e45;   // lane-major too: LP[q * 64 + lane]
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();

	// ---- 5. leaf-relative rank of every new symbol, one per lane
	for (int jj = ln; jj < ni; jj += 64) {
		uint32_t p = p_first, a = a_first;
		if (jj != ln) {                                        // more than 64 new symbols in the window: read them again
			a = INS_A[d.ins0 + jj];
			p = (uint32_t)(INS_E[d.ins0 + jj] - d.i0) + (uint32_t)jj;
		}
		const uint32_t pw = p / SPW, lo = pw / WPL, wi = pw - lo * WPL, below = (p - pw * SPW) * SBITS;
		const uint32_t bl = (p / LEAF) * LPW;                    // first lane of its leaf
		const uint32_t sh = (a & 1) * 16;
		uint32_t r = ((LP[64 * (a >> 1) + lo] >> sh) & 0xffffu) - ((LP[64 * (a >> 1) + bl] >> sh) & 0xffffu);
#pragma unroll
		for (int w = 0; w < WPL; ++w) {
			const uint64_t m = (uint32_t)w < wi ? ~0ull : ((uint32_t)w == wi ? (1ull << below) - 1ull : 0ull);
			r += (uint32_t)__popcll(nib_eq(LX[LM(w, (int)lo)], a) & m);
		}
		RKREL[d.ins0 + jj] = (uint16_t)r;
		if (INPLACE) RKLEAF[d.ins0 + jj] = (uint32_t)d.gl;
	}
	if (!INPLACE && (ln % LPW) == LPW - 1 && (ln / LPW) * LEAF < nvalid) {   // last lane of a leaf that exists (in place: the directory is kept by dir_add)
		const uint32_t bl = (uint32_t)(ln / LPW) * LPW;
		const uint32_t t01 = s01 - LP[bl], t23 = s23 - LP[64 + bl], t45 = s45 - LP[128 + bl];
		LeafMeta m;
		m.c[0] = (uint16_t)t01; m.c[1] = (uint16_t)(t01 >> 16); m.c[2] = (uint16_t)t23; m.c[3] = (uint16_t)(t23 >> 16);
		m.c[4] = (uint16_t)t45; m.c[5] = (uint16_t)(t45 >> 16);
		m.npre = 0;
		m.n = (uint16_t)((t01 & 0xffffu) + (t01 >> 16) + (t23 & 0xffffu) + (t23 >> 16) + (t45 & 0xffffu) + (t45 >> 16));
		newp.own[d.gl + ln / LPW] = m;                          // own counts + fill; k_meta_sb turns them into prefixes
	}
	{
		uint64_t *dst = (uint64_t*)(newp.data + d.gl * (uint64_t)LEAFB) + WPL * ln;
#pragma unroll
		for (int w = 0; w < WPL; ++w) dst[w] = out[w];           // leaves past the end of the piece are padding slots of the same piece
	}
}

template <bool STRIDE> __global__ __launch_bounds__(256) void k_merge(const Ctl *ctl, const LeafDesc *__restrict__ LD, PoolView oldp, PoolView newp,
		const uint64_t *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, uint16_t *RKREL)
{
	__shared__ __align__(16) uint64_t lds[MW][2 * NXW + 16 + NXW / 2];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	uint64_t *LX = lds[wv], *LO = lds[wv] + NXW;                // LO: NXW + 2 words + one pad word per 32 (SO)
	uint32_t *LF = (uint32_t*)(lds[wv] + 2 * NXW + 16);         // NXW flag words of 32 bits
	const int ln = lane_id();
	// one window per wave; a rank of a sharded index launches fewer waves than the upper bound of its windows (the host does not
	// know the rank's share of the batch) and a wave then takes more than one: grid stride over the windows.  The first window's
	// work order is loaded together with the window count (LD holds an entry for every window a grid can name).
	uint64_t gw = (uint64_t)blockIdx.x * MW + wv;
	LeafDesc d = LD[gw];
	const uint64_t nwin = ctl->wf0[NR];
	for (; gw < nwin; gw += (uint64_t)gridDim.x * MW, d = LD[gw < nwin ? gw : 0]) {
		if (d.nvalid == WIN) merge_window<true, WPL, false>(d, LX, LF, LO, ln, oldp, newp, INS_E, INS_A, RKREL, nullptr);
		else merge_window<false, WPL, false>(d, LX, LF, LO, ln, oldp, newp, INS_E, INS_A, RKREL, nullptr);
		if (!STRIDE) return;                                    // (one GPU: the grid covers every window; no loop, no extra registers)
		if (gw + (uint64_t)gridDim.x * MW < nwin) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // the wave's LDS arrays are reused
	}
}

// Sparse rounds keep the rank directory current themselves (the dense rounds rebuild it, k_meta_sb): the directory of a
// superblock holds OWN counts by rows (dir_row, rb2_device.h), so a leaf that received symbols adds them to its own entries and
// to the superblock total, and nothing behind it moves.  What a round costs is then proportional to the leaves it touches; only
// the prefix over the superblock totals (k_sbscan*) still reads every superblock.  (rope.c:139-146: the counts along the path.)
// Atomics although a row entry has one writer (the total has several): a 2-byte store is a partial write the memory side has to
// merge, and measured slower (1 M touched leaves per round: k_merge_leaf 0.42 ms with stores, 0.32 ms with atomics).  They are
// issued as ONE instruction, and before the wave starts shifting words: at its end they cost 0.05 ms more, as three
// instructions another 0.04.
// Lane roles: 0 = the fill, 1-6 = the own count of symbol ln - 1, 7-9 = the three packed words of the superblock total (its own
// small array: 16 bytes per superblock stay in cache, the directory blocks do not -- with the total in the spare row of the block
// k_merge_leaf took 0.06 ms longer and the scan kernels twice as long).  cnt = what the lane adds.
__device__ __forceinline__ void dir_commit(const PoolView &pool, SbTot *sbtot, uint64_t gl, int ln, uint32_t cnt)
{
	const uint32_t k = (uint32_t)(gl % SB);
	uint32_t *ptr = (uint32_t*)dir_row(pool, gl / SB, 0) + (uint32_t)ln * (SB / 2) + (k >> 1);
	if (ln >= 7) ptr = (uint32_t*)&sbtot[gl / SB] + (ln - 7);
	const uint32_t val = ln < 7 ? cnt << ((k & 1) * 16) : cnt;
	if (ln < 10 && val) atomicAdd(ptr, val);
}
// a leaf that receives ni <= 64 symbols, lane j holding the j-th (light path: ni <= LIGHT_NI, a handful of instructions per symbol)
__device__ __forceinline__ void dir_add(const PoolView &pool, SbTot *sbtot, uint64_t gl, int ln, int ni, uint32_t aj)
{
	const uint32_t keyA = (uint32_t)ln - 1u, keyW = (uint32_t)ln - 7u;   // lanes without that role never match: a <= 5, a >> 1 <= 2
	uint32_t cnt = ln == 0 ? (uint32_t)ni : 0u;
	auto tally = [&](int j) {
		const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)aj, j);
		cnt += (a == keyA ? 1u : 0u) + ((a >> 1) == keyW ? 1u << (16 * (a & 1)) : 0u);
	};
	// first symbol outside the loop (ni >= 1): in straight-line code the compiler waits for exactly the load that brings aj; at a
	// loop head it waits for every load in flight -- also the other leaf's words, whose latency this leaf's work is meant to hide
	tally(0);
	for (int j = 1; j < ni; ++j) tally(j);
	dir_commit(pool, sbtot, gl, ln, cnt);
}
// ... any number, already counted: d01 | d23 | d45 packed like LeafMeta::c, same values in all lanes
__device__ __forceinline__ void dir_add_packed(const PoolView &pool, SbTot *sbtot, uint64_t gl, int ln, uint32_t d01, uint32_t d23, uint32_t d45)
{
	const int s = ln - 1;
	const uint32_t dw = s < 2 ? d01 : (s < 4 ? d23 : d45);
	uint32_t cnt = (dw >> ((uint32_t)(s & 1) * 16)) & 0xffffu;
	if (ln == 0) cnt = (d01 & 0xffffu) + (d01 >> 16) + (d23 & 0xffffu) + (d23 >> 16) + (d45 & 0xffffu) + (d45 >> 16);
	if (ln >= 7) cnt = ln == 7 ? d01 : (ln == 8 ? d23 : d45);
	dir_commit(pool, sbtot, gl, ln, cnt);
}

// A leaf that receives only a few symbols (the normal case of a sparse round: one or two) does not need the window machinery:
// every lane keeps its word in a register; per new symbol (ascending position, so earlier ones are already in place) one
// masked compare + wave sum gives its rank, one shift with a DPP carry from the lane below opens the gap.  No LDS, and only
// the words from the first changed one on are stored.  (rle_insert_cached, rle.c:10-89, for <= LIGHT_NI inserts.)
constexpr int LIGHT_NI = 8;
constexpr int LPWV = 4;                     // touched leaves per wave in k_merge_leaf.  The kernel is bound by instruction issue (PMC: 225
                                            // VALU + 184 SALU per two leaves at first) and by the latency of each wave's loads: all loads of
                                            // the wave's leaves are issued first, back to back (see the barrier in k_merge_leaf; while the
                                            // compiler still waited inside leaf_job_load, or sank loads into the branches, two leaves per
                                            // wave were the optimum and the kernel took 0.31 ms; now 0.26 with two, 0.24 with four)
struct LeafJob { uint64_t w; uint32_t pj, aj; };

__device__ __forceinline__ void leaf_job_load(const LeafDesc &d, const int ln, const PoolView &pool,
		const uint64_t *INS_E, const uint8_t *INS_A, LeafJob &J)
{
	J.w = ((const uint64_t*)pool.data)[d.gl * LEAFW + ln];
	// No branch and no use of a loaded value in here: the loads of all the wave's leaves are to be in flight together, and the
	// compiler's wait counts stay exact only in straight-line code (lanes >= ni load the last insert again; they never use it).
	const uint32_t q = (uint32_t)d.ins0 + (uint32_t)min(ln, (int)d.ni - 1);
	J.aj = INS_A[q]; J.pj = ((const uint32_t*)INS_E)[2 * (uint64_t)q];   // low half: positions inside a leaf need no more
}

__device__ __forceinline__ void leaf_job_run(const LeafDesc &d, const int ln, const PoolView &pool, LeafJob &J, uint16_t *RKREL, uint32_t *RKLEAF, SbTot *sbtot)
{
	uint64_t *leaf = (uint64_t*)pool.data + d.gl * LEAFW;
	const int ni = d.ni;
	uint64_t w = J.w;
	uint32_t myrank = 0;
	J.pj = J.pj - (uint32_t)d.i0 + (uint32_t)ln;                 // final position E[q] + q inside the leaf (lanes >= ni: unused)
	const uint32_t pwl = J.pj / SPW, pol = (J.pj - pwl * SPW) * SBITS;   // word and bit offset of every insert, by its lane, once (a scalar
	                                                                     // division by 21 per insert in the loop is eight SALU instructions)
	const uint32_t pw0 = (uint32_t)__builtin_amdgcn_readlane((int)pwl, 0);   // first word that changes
	// what the leaf receives is known before the first symbol is placed: the directory atomics go out first and are under way
	// while the wave shifts words
	dir_add(pool, sbtot, d.gl, ln, ni, J.aj);
	if (ln < ni) RKLEAF[d.ins0 + ln] = (uint32_t)d.gl;
	for (int j = 0; j < ni; ++j) {
		const uint32_t pw = (uint32_t)__builtin_amdgcn_readlane((int)pwl, j), po = (uint32_t)__builtin_amdgcn_readlane((int)pol, j);
		const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)J.aj, j);
		const uint64_t below = (1ull << po) - 1ull;
		const uint64_t msk = (uint32_t)ln < pw ? MLOW : ((uint32_t)ln == pw ? (MLOW & below) : 0ull);
		const uint32_t r = lane63(dpp_incl_add((uint32_t)__popcll(nib_eq(w, a) & msk)));   // a's in front of p, leaf as it is now
		if (ln == j) myrank = r;
		const uint32_t carry = dpp_prev_lane((uint32_t)(w >> (SBITS * (SPW - 1))) & 7u);    // top symbol of the lane below moves up
		if ((uint32_t)ln > pw) w = ((w << SBITS) & MALL) | carry;
		else if ((uint32_t)ln == pw) w = (w & below) | ((uint64_t)a << po) | (((w & ~below) << SBITS) & MALL);
	}
	if ((uint32_t)ln >= pw0) leaf[ln] = w;                     // (the line is in L2: the leaf was just read)
	if (ln < ni) RKREL[d.ins0 + ln] = (uint16_t)myrank;
}

// sparse rounds: one wave per LPWV TOUCHED leaves (work orders appended by k_part_sparse, any order), rewritten in place --
// rope_insert_run's descent ends here (rope.c:136-141) and this is rle_insert_cached (rle.c:10-89) for all the inserts a
// leaf receives this round at once.  Untouched leaves keep their bytes.  A round that set ctl->overflow is void.
template <bool STRIDE> __global__ __launch_bounds__(256) void k_merge_leaf(const Ctl *ctl, const LeafDesc *__restrict__ LD, PoolView pool,
		const uint64_t *INS_E, const uint8_t *INS_A /* not __restrict__: see the barrier below */, uint16_t *RKREL, uint32_t *RKLEAF, SbTot *sbtot)
{
	__shared__ __align__(16) uint64_t lds[MW][64 + 136 + 32];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ln = lane_id();
	const uint64_t stride = (uint64_t)gridDim.x * MW * LPWV;
	for (uint64_t g0 = ((uint64_t)blockIdx.x * MW + wv) * LPWV; ; g0 += stride) {   // grid stride: a sharded rank launches waves for about twice its fair share of the batch
	LeafDesc d[LPWV];
	LeafJob J[LPWV];
#pragma unroll
	for (int k = 0; k < LPWV; ++k) d[k] = LD[g0 + k];          // LD has a slot for every order the grid could run: issued together with the counters
	const uint32_t nwork = ctl->nwork;                        // (the kernel is bound by the latency of its chain of loads: every link counts)
	if (ctl->overflow || g0 >= nwork) return;
#pragma unroll
	for (int k = 1; k < LPWV; ++k) if (g0 + k >= nwork) d[k] = d[0];   // (a duplicate of the first order is loaded but never run)
#pragma unroll
	for (int k = 0; k < LPWV; ++k) leaf_job_load(d[k], ln, pool, INS_E, INS_A, J[k]);
	asm volatile("" ::: "memory");                            // the loads stay here, all of them, in this order: the compiler would sink each into the branch that uses
	                                                          // it (and does, across this barrier, for pointers it knows to be read-only and unaliased)
#pragma unroll
	for (int k = 0; k < LPWV; ++k) {
		if (g0 + k >= nwork) break;
		if (d[k].ni <= LIGHT_NI) { leaf_job_run(d[k], ln, pool, J[k], RKREL, RKLEAF, sbtot); continue; }
		uint64_t *LX = lds[wv], *LO = lds[wv] + 64;             // LO: 64 + 2 old words, later the 64 x 4 packed prefixes (128 words)
		uint32_t *LF = (uint32_t*)(lds[wv] + 64 + 136);         // 64 flag words of 32 bits
		uint32_t dd[3] = {0, 0, 0};                              // what the leaf receives, per symbol
		for (int j0 = 0; j0 < (int)d[k].ni; j0 += 64) {
			const uint32_t a = j0 + ln < (int)d[k].ni ? (uint32_t)INS_A[d[k].ins0 + j0 + ln] : 7u;
#pragma unroll
			for (int sy = 0; sy < 6; ++sy) dd[sy >> 1] += (uint32_t)__popcll(__ballot(a == (uint32_t)sy)) << (16 * (sy & 1));
		}
		dir_add_packed(pool, sbtot, d[k].gl, ln, dd[0], dd[1], dd[2]);
		merge_window<false, 1, true>(d[k], LX, LF, LO, ln, pool, pool, INS_E, INS_A, RKREL, RKLEAF);
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();   // the LDS arrays are reused by the next order
	}
	if (!STRIDE) return;
	}
}

// The same work as a PERSISTENT, software-pipelined kernel: a wave walks the work list with a grid stride and keeps three groups
// of LPWP orders going at once -- the group it is inserting into, the group whose leaf words and insert records are in flight,
// and the group whose work orders (scalar loads) are in flight.  k_merge_leaf above starts a wave per group: every wave pays the
// chain order -> leaf + inserts -> stores once, and what hides it is only the other waves of the SIMD (8 at most).  Here the
// chain of the NEXT group runs beside the shifts of the current one.
constexpr int LPWP = 2;
__global__ __launch_bounds__(256) void k_merge_leaf_pipe(const Ctl *ctl, const LeafDesc *__restrict__ LD, PoolView pool,
		const uint64_t *INS_E, const uint8_t *INS_A, uint16_t *RKREL, uint32_t *RKLEAF, SbTot *sbtot)
{
	__shared__ __align__(16) uint64_t lds[MW][64 + 136 + 32];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ln = lane_id();
	const uint64_t stride = (uint64_t)gridDim.x * MW * LPWP;
	uint64_t g0 = ((uint64_t)blockIdx.x * MW + wv) * LPWP;
	const uint32_t nwork = ctl->nwork;
	if (ctl->overflow || g0 >= nwork) return;
	auto ld_desc = [&](uint64_t g, LeafDesc *d) {                // orders g .. g + LPWP - 1 (behind the end: a duplicate of the group's first, loaded but never run)
#pragma unroll
		for (int k = 0; k < LPWP; ++k) d[k] = LD[g + k < nwork ? g + k : g];
	};
	LeafDesc d[LPWP], dn[LPWP], dnn[LPWP];
	LeafJob J[LPWP], Jn[LPWP];
	ld_desc(g0, dn);
	if (g0 + stride < nwork) ld_desc(g0 + stride, dnn); else ld_desc(g0, dnn);
#pragma unroll
	for (int k = 0; k < LPWP; ++k) leaf_job_load(dn[k], ln, pool, INS_E, INS_A, Jn[k]);
	for (;;) {
#pragma unroll
		for (int k = 0; k < LPWP; ++k) { d[k] = dn[k]; J[k] = Jn[k]; dn[k] = dnn[k]; }
		const uint64_t g1 = g0 + stride, g2 = g1 + stride;
		const bool more = g1 < nwork;
		if (more) {
#pragma unroll
			for (int k = 0; k < LPWP; ++k) leaf_job_load(dn[k], ln, pool, INS_E, INS_A, Jn[k]);   // next group: in flight while this one is worked on
			ld_desc(g2 < nwork ? g2 : g1, dnn);
		}
		asm volatile("" ::: "memory");
#pragma unroll
		for (int k = 0; k < LPWP; ++k) {
			if (g0 + k >= nwork) break;
			if (d[k].ni <= LIGHT_NI) { leaf_job_run(d[k], ln, pool, J[k], RKREL, RKLEAF, sbtot); continue; }
			uint64_t *LX = lds[wv], *LO = lds[wv] + 64;
			uint32_t *LF = (uint32_t*)(lds[wv] + 64 + 136);
			uint32_t dd[3] = {0, 0, 0};
			for (int j0 = 0; j0 < (int)d[k].ni; j0 += 64) {
				const uint32_t a = j0 + ln < (int)d[k].ni ? (uint32_t)INS_A[d[k].ins0 + j0 + ln] : 7u;
#pragma unroll
				for (int sy = 0; sy < 6; ++sy) dd[sy >> 1] += (uint32_t)__popcll(__ballot(a == (uint32_t)sy)) << (16 * (sy & 1));
			}
			dir_add_packed(pool, sbtot, d[k].gl, ln, dd[0], dd[1], dd[2]);
			merge_window<false, 1, true>(d[k], LX, LF, LO, ln, pool, pool, INS_E, INS_A, RKREL, RKLEAF);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
		}
		if (!more) return;
		g0 = g1;
	}
}

// ---------------------------------------------------------------------------------------------
// k_export: chunks [c0, c0+nc) of XCHUNK symbols of one sub-rope -> run-length bytes of ropebwt2's 43+3 codec, one
// byte per run of <= 15 symbols (rle_enc1's 1-byte form, rle.h:55-57), runs cut at chunk ends.  A sub-rope is a flat
// array of 3-bit symbols (every leaf but the last is full), so chunk boundaries need not respect leaves.
// Output: slot i of `dst` (stride XCHUNK) holds nb[i] bytes.  Not on the hot path.
// ---------------------------------------------------------------------------------------------

constexpr int XCHUNK = 1024;                // symbols per export chunk: 16 per lane
struct ExportLds { uint8_t outb[XCHUNK + 16]; };

__device__ __forceinline__ uint32_t byte_of(const uint32_t w[4], int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }
__device__ __forceinline__ uint32_t tri4_to_bytes(uint32_t t)      // four 3-bit symbols (12 bits) -> 4 bytes
{
	return (t & 7u) | (t & 0x38u) << 5 | (t & 0x1c0u) << 10 | (t & 0xe00u) << 15;
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
	uint64_t bits = 0;                                          // my 16 symbols, 3 bits each
	if (myvalid) {
		const uint64_t s = s0 + p0, wi = s / SPW;
		const uint32_t sh = (uint32_t)(s - wi * SPW) * SBITS;
		const uint64_t *W = (const uint64_t*)(pv.data + leaf0 * (uint64_t)LEAFB) + wi;
		bits = W[0] >> sh;
		if (sh + 16 * SBITS > 63) bits |= W[1] << (63 - sh);      // the next word belongs to the piece's slots (padding at worst)
	}
	uint32_t pw[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) pw[k] = tri4_to_bytes((uint32_t)(bits >> (12 * k)) & 0xfffu);
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
