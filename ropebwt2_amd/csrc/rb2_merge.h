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
// What rle_insert_cached gains by coding (fewer bytes per symbol than a fixed-width field, rle.h:53-75) is gained here by not
// storing the third plane of a window where it almost never differs from what the other two imply (round 5: window formats).
//
// k_merge work decomposition: ONE WAVE PER OUTPUT WINDOW of 64 groups (WPL = 4 leaves), lane = group, four independent waves
// per block, no block-level barrier anywhere:
//   1. the new symbols of the window are OR-ed into position-indexed LDS words -- one flag word and three plane words per group
//      -- one 64-bit LDS atomic per set bit; the old groups the window draws from are loaded (whole 128-byte lines) and staged in
//      LDS: three plane words of a PLAIN old window, two of a COMPACT one, whose third plane is OR-ed together from its list of
//      exception positions ("window formats" below)
//   2. one wave prefix sum of the not-new counts -> first old symbol each lane consumes; the old bits of its group are an
//      unaligned 64-bit window of each staged plane (three dword reads, two funnel shifts)
//   3. expand: open one 1-bit gap per new symbol in each plane (open_gaps: a three-input bit operation per half and an addition per
//      plane and gap; 2-3 trips in steady state -- as many as the fullest group of the window takes new symbols); the new symbols are
//      already in place
//   4. symbol counts per lane: five popcounts of dense words, three packed scans inside the DPP row (= leaf) -> LeafMeta of each
//      leaf, the format of the new window (its $ + N counts are its exceptions) and, for a compact one, its list
//   5. RKREL: every new symbol gets the number of equal symbols before it INSIDE its leaf, one new symbol per lane (row prefix of
//      the owning lane + a masked plane compare of its group, both read back from LDS); k_advance adds the directory prefix of the
//      new sub-rope to obtain the reference's return value of rope_insert_run.
// The kernel is bound by VALU issue, by the length of a wave's own chain at eight waves per SIMD and by HBM bytes, in that order of what was
// found (DESIGN.md sections 6 and 10): every step above is written for instruction count -- 216 vector instructions per window as executed.
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

// open a one-bit gap at every set bit of f (ascending), in all three planes: the low bits of x[] move up past the gaps (a software bit
// deposit; f has few bits in steady state -- one new symbol per ~170 old ones at configs[1]).  Per gap: the bits below the lowest set bit
// of f are lm = (f - 1) & ~f -- all ones when f is empty, which makes the step a no-op for a lane that is done, no branch --, and moving
// the part of x above them up by one is an ADD: x + (x & ~lm).  13 VALU per trip for the three planes (r04: masks from a count of
// trailing zeros, shift and two ORs per plane: 28; r05: 17, the mask built first).  The trip count is the largest number of new symbols in one group of the window.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define RB2_UNDEF(x) RB2_UNDEFV(x)                            // (rb2_device.h)
__device__ __forceinline__ void open_gaps(uint64_t x[3], uint64_t f)
{
	do {
		const uint64_t t = f - 1ull;
		const uint32_t fl = (uint32_t)f, fh = (uint32_t)(f >> 32), tl = (uint32_t)t, th = (uint32_t)(t >> 32);
#pragma unroll
		for (int pl = 0; pl < 3; ++pl)                              // x & ~lm = x & (f | ~t): one v_bitop3_b32 per half, no mask to build (left to itself the compiler builds f | -f first: 4 more)
		{
			const u32x2 m = { (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)x[pl], fl, tl, 0xd0), (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)(x[pl] >> 32), fh, th, 0xd0) };
			x[pl] += __builtin_bit_cast(uint64_t, m);               // (a register pair as it stands: written as hi << 32 | lo it became two 64-bit additions)
		}
		f &= t;
	} while (__any(f != 0));
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

// ---- window formats of the dense layout ------------------------------------------------------------------------------------
// A window = WPL = 4 consecutive leaves = 4096 symbols = twelve 128-byte lines (leaf-major, three plane lines per leaf).  DNA needs two
// planes: with $ACGTN = 000 001 010 011 100 101, planes 0 and 1 tell A, C, G, T apart, and plane 2 is 1 for T -- a quarter of all symbols.
// Swap the codes of $ and T (plane 2' = plane 2 ^ ~(plane 0 | plane 1)) and plane 2' is set for `$` and `N` only: about 1 % of the
// symbols of a read set (one sentinel per read), none at all while the first batch of a job is being inserted.  A COMPACT window keeps
// planes 0 and 1 where they always are and, instead of the four plane-2 lines, the POSITIONS of its plane-2' bits as 16-bit entries
// ("exceptions": entry 0 = their number, entries 1 .. n = positions 0 .. 4095) in the plane-2 line of its first leaf (n <= 63) and of its
// second leaf (n <= 127); the other plane-2 lines are not touched at all.  A window with more exceptions (runs of N, very short reads)
// stays PLAIN: three planes, original codes.  The memory side moves whole lines, and lines that are skipped cost nothing
// (tools/ubench/window_stream.hip: 10 of 12 lines of every window at the same 5.9-6.0 TB/s as 12 of 12, 8 of 12 at 6.6) -- so the
// stride, the leaf addresses and the directory stay what they are, and a round reads and writes 8 to 10 lines per window instead of 12.
// What rle_insert_cached gains by coding runs (rle.c:63-86, rle.h:53-75) -- fewer bytes per symbol than a fixed-width field -- is gained
// here by not storing the plane that almost never differs.
// The format of a window is the `npre` field of its first leaf's entry in own[] (written by the merge with the leaf's counts; k_relayout
// and the loader write 0 = plain), mirrored in one byte per window (PoolView::xh) that is valid on a pool side this kernel wrote.  k_part
// hands the formats of the (at most two) old windows an output window draws from to the merge in its work order.  Only k_merge reads and writes compact windows: the host asks for them (compact_out) only when every reader of
// the pool until the next rewrite is k_merge again -- all intervals empty, not the last round of the batch, no change of layout ahead.
constexpr uint32_t WF_PLAIN = 0, WF_C0 = 1, WF_C1 = 2, WF_C2 = 3;   // compact: 0 / 1 / 2 exception lines
constexpr uint32_t XCAP1 = 63, XCAP2 = 127;                          // exceptions one / two lines hold
static_assert(WIN < (1 << 14), "LeafDesc packs the old windows' formats above ni / nvalid");

// LDS words one wave of merge_window needs: flags + three planes of new symbols + three planes of staged old groups
template <int GPL_> struct MergeLds { static constexpr int WG = 64 * GPL_, WORDS = WG + 3 * WG + 3 * (WG + 2) + 2; };

// bits [op, op + 64) of a staged plane (dwords `pl` ... in LDS), op = 32 * dk + sh: three dword reads and two funnel shifts
__device__ __forceinline__ uint64_t stage_bits(const uint32_t *pl, uint32_t dk, uint32_t sh)
{
	const uint32_t d0 = pl[dk], d1 = pl[dk + 1], d2 = pl[dk + 2];
	return (uint64_t)__builtin_amdgcn_alignbit(d2, d1, sh) << 32 | __builtin_amdgcn_alignbit(d1, d0, sh);
}

// One output window of 64 groups, lane = group.  FULL: the window holds WIN symbols (all but the last window of a piece) -- every
// position is valid.  Inside, symbols are in the swapped coding (plane 2' above).
// compact_out: bit 0 -- new windows may be written compact (else plain); bit 1 -- count the windows per format in Ctl::wfmt.
// The kernel is bound by VALU issue as much as by HBM (r04: 272 VALU per window at 75 % of the issue rate while moving 5 GB at 6 TB/s;
// r05 first cut of the compact format: 25 % fewer bytes, 323 VALU, 11 % SLOWER), so every step is written for instruction count: lane
// offsets in 32 bits on uniform bases, funnel shifts of dwords for the unaligned plane windows, gaps opened by additions, scans inside
// the DPP row (= leaf) only, exception bookkeeping from numbers the counts already provide.
template <bool FULL, int GPL_, typename P> __device__ __forceinline__ void merge_window(const LeafDesc &d, uint64_t *lds, const int ln,
		const PoolView &oldp, const PoolView &newp, const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, uint16_t *RKREL, const int compact_out, unsigned long long *wfmt)
{
	static_assert(GPL_ == 1, "one group per lane");
	constexpr int WG = 64, WINS = WG * GSYM;
	uint64_t *LF = lds, *LX = lds + WG, *LO = lds + 4 * WG;                  // flags; planes of the new symbols (later: of the output); staged old groups, plane pl at LO + pl * (WG + 2)
	uint64_t *LO2 = LO + 2 * (WG + 2);
	const uint32_t nvalid = FULL ? (uint32_t)WINS : (uint32_t)(d.nvalid & 0x3fffu), ni = d.ni & 0x3fffu;
	const uint32_t h0 = d.ni >> 14, h1 = d.nvalid >> 14;        // formats of the first / second old window this one draws from
	const uint32_t nold = nvalid - ni;                          // old symbols consumed by this window
	const uint32_t sh0 = (uint32_t)d.i0 & 63u;                  // the first of them: bit sh0 of old group G0 = i0 >> 6 ...
	const uint32_t g0 = (uint32_t)(d.i0 >> 6) & 63u;            // ... which is group g0 of old window ow (of the piece)
	const uint64_t ow = d.i0 >> 12;
	const uint32_t nwg = nold ? (sh0 + nold + 63) >> 6 : 0u;    // old groups they live in (<= WG + 1); nothing old (the first rounds on an empty index): nothing is staged and the stage reads zeros
	const uint64_t *obw = (const uint64_t*)oldp.data + ((uint64_t)d.oleaf0 + ow * WPL) * LEAFW;   // window ow; staged group k is its group g0 + k
	const uint32_t ln32 = (uint32_t)ln;

	// ---- 1. new symbols of this window, by output position (planes and "new here" flag); the old groups it draws from
	LF[ln] = 0; LX[ln] = 0; LX[WG + ln] = 0; LX[2 * WG + ln] = 0;
	LO2[ln] = 0; LO2[ln + 1] = 0;                               // plane 2' of the old groups of compact windows is OR-ed together from their exception lists (WG + 1 words: one two-word store, no lane picked out)
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	// Loads in the order their values are needed (the wave waits for its loads in issue order): my first new symbol, the exception
	// entries, then the plane words -- the LDS atomics of the first two run while the planes are still on their way.
	const P *E0 = INS_E + d.ins0; const uint8_t *A0 = INS_A + d.ins0;
	const uint32_t i0lo = (uint32_t)d.i0;
	uint32_t e_first, a_first; RB2_UNDEF(e_first); RB2_UNDEF(a_first);
	if (ln32 < ni) { e_first = (uint32_t)E0[ln32]; a_first = A0[ln32]; }   // (positions inside a window: the low half decides)
	// the exception lists of the old windows: lane ln reads entry ln of a line (entry 0 of the first line = the number of exceptions)
	const bool two = g0 + nwg > 64u;                            // some staged group lies in window ow + 1
	const uint32_t hx0 = nwg > 0 ? h0 : 0u, hx1 = two ? h1 : 0u; // the formats as far as their lists are read (numbers, compared where they are used: a flag kept across the branches below is rebuilt from a vector register)
#define xa (hx0 >= WF_C1)
#define xb (hx1 >= WF_C1)
	uint32_t xe[4]; RB2_UNDEF(xe[0]); RB2_UNDEF(xe[1]); RB2_UNDEF(xe[2]); RB2_UNDEF(xe[3]);
	{
		const uint16_t *x0 = (const uint16_t*)(obw + 2 * LEAFG);   // plane-2 line of the first leaf of window ow
		if (xa) xe[0] = x0[ln32];
		if (hx0 == WF_C2) xe[1] = x0[LEAFW * 4 + ln32];             // ... of its second leaf
		if (xb) xe[2] = x0[WPL * LEAFW * 4 + ln32];
		if (hx1 == WF_C2) xe[3] = x0[WPL * LEAFW * 4 + LEAFW * 4 + ln32];
	}
	const uint32_t t = g0 + ln32;                               // my staged group as a group of window ow (or, from 64 on, of ow + 1)
	const uint32_t woff = (t >> 4) * LEAFW + (t & 15u);         // its plane-0 word
	const bool have = ln32 < nwg;
	const bool pl0 = h0 == WF_PLAIN, pl1 = h1 == WF_PLAIN;
	const bool plain_k = have && ((t < 64u && pl0) || (t >= 64u && pl1));
	uint64_t wa0 = 0, wa1 = 0;                                  // (zero: a window with nothing old reads its stage, see nwg)
	uint64_t wa2, wt0, wt1, wt2; RB2_UNDEF(wa2); RB2_UNDEF(wt0); RB2_UNDEF(wt1); RB2_UNDEF(wt2);   // (group 64 of the stage is only read when it was loaded: tail)
	if (have) { wa0 = RB2_LDNT(&obw[woff]); wa1 = RB2_LDNT(&obw[woff + LEAFG]); }
	if (plain_k) wa2 = RB2_LDNT(&obw[woff + 2 * LEAFG]);
	const bool tail = ln == 0 && (uint32_t)WG < nwg;            // (group g0 + 64 lies in window ow + 1)
	if (tail) {
		const uint32_t t2 = g0 + 64u, w2 = (t2 >> 4) * LEAFW + (t2 & 15u);
		wt0 = obw[w2]; wt1 = obw[w2 + LEAFG];
		if (h1 == WF_PLAIN) wt2 = obw[w2 + 2 * LEAFG];
	}
	auto put_new = [&](uint32_t p, uint32_t ix /* p >> 6 */, uint32_t a) {
		const unsigned long long bit = 1ull << (p & 63);
		atomicOr((unsigned long long*)&LF[ix], bit);
		if (a & 1u) atomicOr((unsigned long long*)&LX[ix], bit);
		if (a & 2u) atomicOr((unsigned long long*)&LX[WG + ix], bit);
		if ((0x21u >> a) & 1u) atomicOr((unsigned long long*)&LX[2 * WG + ix], bit);   // plane 2': $ and N
	};
	const uint32_t p_first = e_first - i0lo + ln32;             // final position E[q] + q, relative to the window (kept for step 5)
	uint32_t g_first = p_first >> 6;                            // its group (kept as well; opaque, or the index below becomes (p >> 3) & ~7: three instructions for two)
	asm volatile("" : "+v"(g_first));
	if (ln32 < ni) put_new(p_first, g_first, a_first);
	if (ni > 64u)                                               // (more than 64 new symbols: rare in steady state -- a scalar test in front of the loop's vector one)
		for (uint32_t jj = ln32 + 64; jj < ni; jj += 64) { const uint32_t p = (uint32_t)E0[jj] - i0lo + jj; put_new(p, p >> 6, A0[jj]); }
	{	// exceptions -> plane 2' bits of the staged groups: entry e of window ow + wi is bit e & 63 of its group e >> 6 = staged group (e >> 6) + 64 wi - g0
		auto x_in = [&](uint32_t e, uint32_t idx_m1 /* entry number - 1 */, uint32_t cnt, uint32_t delta) {
			const uint32_t k = (e >> 6) + delta;
			if (idx_m1 < cnt && k <= (uint32_t)WG) atomicOr((unsigned long long*)&LO2[k], 1ull << (e & 63));
		};
		if (xa) {
			const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)xe[0]);
			x_in(xe[0], ln32 - 1u, cnt, 0u - g0);
			if (hx0 == WF_C2) x_in(xe[1], ln32 + 63u, cnt, 0u - g0);
		}
		if (xb) {
			const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)xe[2]);
			x_in(xe[2], ln32 - 1u, cnt, 64u - g0);
			if (hx1 == WF_C2) x_in(xe[3], ln32 + 63u, cnt, 64u - g0);
		}
	}
#undef xa
#undef xb
	asm volatile("" : "+v"(wa0), "+v"(wa1), "+v"(wa2), "+v"(wt0), "+v"(wt1), "+v"(wt2));   // nothing of the plane words is looked at before this point (the first look waits for them)
	LO[ln] = wa0; LO[(WG + 2) + ln] = wa1;
	if (plain_k) LO2[ln] = wa2 ^ ~(wa0 | wa1);                  // plain window: to the swapped coding (a word no exception list writes to)
	if (ln == 0) {
		LO[WG] = wt0; LO[(WG + 2) + WG] = wt1;
		if (tail && h1 == WF_PLAIN) LO2[WG] = wt2 ^ ~(wt0 | wt1);
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();

	// ---- 2. what does each lane consume
	uint64_t F = LF[ln];                                        // bit i: position i holds a new symbol
	const uint32_t v = FULL ? (uint32_t)GSYM : (uint32_t)min((int)GSYM, max(0, (int)nvalid - ln * GSYM));
	const uint64_t VM = FULL ? ~0ull : bits_below(v);           // the valid positions
	const uint32_t non = v - (uint32_t)__popcll(F);
	uint64_t out[3];
	{
		const uint32_t op = sh0 + (dpp_incl_add(non) - non);      // first old symbol of this lane, in symbols of the stage
		const uint32_t dk = op >> 5, sh = op & 31u;
		const uint32_t *S = (const uint32_t*)LO;
#pragma unroll
		for (int pl = 0; pl < 3; ++pl) out[pl] = stage_bits(S + pl * 2 * (WG + 2), dk, sh);   // what lies behind my `non` bits is shifted out or masked below
	}

	// ---- 3. deal the old symbols to the not-new positions, add the new ones
	if (nold != 0 && ni != 0) open_gaps(out, F);                // (nothing old: out[] is zero, see nwg -- and every position is a gap)
	out[0] = (out[0] & VM) | LX[ln]; out[1] = (out[1] & VM) | LX[WG + ln]; out[2] = (out[2] & VM) | LX[2 * WG + ln];   // (the gaps hold zeros)

	// ---- 4. counts per lane -> prefix inside the leaf (= DPP row) -> LeafMeta of the leaves, rank bases of the new symbols
	uint32_t c[6];
	{
		PlAcc A;
		pl_acc(A, out[0], out[1], out[2], VM);
		pl_finish(A, v, c);
		const uint32_t t4 = c[0]; c[0] = c[4]; c[4] = t4;         // (the planes are in the swapped coding)
	}
	const uint32_t e01 = c[0] | c[1] << 16, e23 = c[2] | c[3] << 16, e45 = c[4] | c[5] << 16;
	const uint32_t s01 = row_incl_add(e01), s23 = row_incl_add(e23), s45 = row_incl_add(e45);
	// the format of the new window: its exceptions are its $ and N symbols, already counted and scanned
	const uint32_t nx = c[0] + c[5], xs = (s01 & 0xffffu) + (s45 >> 16);   // mine / inclusive prefix inside my row
	uint32_t fmt = WF_PLAIN, xw = xs, xt = 0;
	if (compact_out & 1) {
		xw += dpp0<0x142, 0xa>(xw); xw += dpp0<0x143, 0xc>(xw);  // the row prefixes carried over the wave (row_bcast 15 / 31): two instructions, and the list below starts where it ends
		xt = lane63(xw);
		fmt = xt == 0 ? WF_C0 : (xt <= XCAP1 ? WF_C1 : (xt <= XCAP2 ? WF_C2 : WF_PLAIN));
	}
	if ((compact_out & 2) && ln == 0) atomicAdd(wfmt + fmt, 1ull);   // statistics for the tests (RB2_COMPACT_STATS=1)
	// publish my group and my exclusive prefixes inside the leaf (the old-group stage is dead by now)
	uint16_t *LP = (uint16_t*)LO;                               // LP[8 * lane + symbol]: sixteen bytes per lane, so that a new symbol finds its entry with two instructions and reads it as it is
	LX[ln] = out[0]; LX[WG + ln] = out[1]; LX[2 * WG + ln] = out[2];
	{ const u32x2 w = { s01 - e01, s23 - e23 }; *(u32x2*)(LP + 8 * ln) = w; *(uint32_t*)(LP + 8 * ln + 4) = s45 - e45; }
	uint16_t *XL = (uint16_t*)LF;                               // the exception list on its way out (the flags are dead)
	if (fmt >= WF_C1) {                                          // (wave-uniform)
		uint64_t x = out[2];
		uint32_t at = 1u + xw - nx;
		if (ln == 0) XL[0] = (uint16_t)xt;
		while (x) { XL[at++] = (uint16_t)((ln32 << 6) + (uint32_t)__builtin_ctzll(x)); x &= x - 1; }
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();

	// ---- 5. leaf-relative rank of every new symbol, one per lane
	uint16_t *RK0 = RKREL + d.ins0;
	for (uint32_t jj = ln32; jj < ni; jj += 64) {
		uint32_t p = p_first, a = a_first, lo = g_first;
		if (jj != ln32) { a = A0[jj]; p = (uint32_t)E0[jj] - i0lo + jj; lo = p >> 6; }   // more than 64 new symbols in the window: read them again
		uint32_t r = LP[8 * lo + a];                               // equal symbols in the groups of its leaf in front of its group
		// all-ones where its code in the planes ($ <-> T) has the bit set: bit a of the set of symbols whose code has it (plane 0: A T N, plane 1: C T... as
		// numbers: 1 3 5 / 2 3 / 0 5) -- one signed bit-field extract per plane; then, per half, three three-input bit operations (v_bitop3_b32)
		// AND together "plane bit equals code bit" and "below my position": 18 VALU per new symbol (r05: 34)
		const uint32_t n0 = (uint32_t)__builtin_amdgcn_sbfe(0x2a, a, 1), n1 = (uint32_t)__builtin_amdgcn_sbfe(0x0c, a, 1), n2 = (uint32_t)__builtin_amdgcn_sbfe(0x21, a, 1);
		const uint64_t q0 = LX[lo], q1 = LX[WG + lo], q2 = LX[2 * WG + lo];
		const uint64_t nb = ~0ull << (p & 63);                     // my position and above
		uint32_t el = (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)q0, n0, (uint32_t)nb, 0x41);            // ~(q ^ n) & ~nb
		uint32_t eh = (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)(q0 >> 32), n0, (uint32_t)(nb >> 32), 0x41);
		el = (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)q1, n1, el, 0x82); eh = (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)(q1 >> 32), n1, eh, 0x82);   // ~(q ^ n) & e
		el = (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)q2, n2, el, 0x82); eh = (uint32_t)__builtin_amdgcn_bitop3_b32((uint32_t)(q2 >> 32), n2, eh, 0x82);
		r += (uint32_t)__popc(el) + (uint32_t)__popc(eh);
		RK0[jj] = (uint16_t)r;
	}
	const uint32_t lf = ln32 >> 4;                              // my leaf of the window
	if ((ln32 & 15u) == 15u && lf * (uint32_t)LEAF < nvalid) {  // last lane of a leaf that exists: the inclusive row prefixes are the leaf's counts
		LeafMeta m;
		m.c[0] = (uint16_t)s01; m.c[1] = (uint16_t)(s01 >> 16); m.c[2] = (uint16_t)s23; m.c[3] = (uint16_t)(s23 >> 16);
		m.c[4] = (uint16_t)s45; m.c[5] = (uint16_t)(s45 >> 16);
		m.npre = (uint16_t)(lf == 0 ? fmt : 0u);                // the window's format rides in its first leaf's entry
		m.n = (uint16_t)(FULL ? (uint32_t)LEAF : min((uint32_t)LEAF, nvalid - lf * (uint32_t)LEAF));
		newp.own[d.gl + lf] = m;                                // own counts + fill; k_meta_sb turns them into prefixes
		if (lf == 0) newp.xh[d.gl / WPL] = (uint8_t)fmt;       // ... and in the byte per window k_part reads next round (16 bytes apart in own[], a line per 128 windows here)
	}
	uint64_t *dstw = (uint64_t*)newp.data + d.gl * LEAFW;       // the window; lane's group: leaf lf, group ln & 15
	const uint32_t doff = lf * LEAFW + (ln32 & 15u);
	RB2_STNT(out[0], &dstw[doff]); RB2_STNT(out[1], &dstw[doff + LEAFG]);   // leaves past the end of the piece are padding slots of the same piece
	if (fmt == WF_PLAIN) RB2_STNT((out[2] ^ ~(out[0] | out[1])) & VM, &dstw[doff + 2 * LEAFG]);   // back to the original codes
	else if (fmt >= WF_C1 && ln32 < (fmt == WF_C2 ? 32u : 16u)) RB2_STNT(((const uint64_t*)XL)[ln], &dstw[doff + 2 * LEAFG]);   // one or two whole lines of 64 entries
}

#ifndef RB2_MMW
#define RB2_MMW 4                            // waves (= windows) per block of k_merge
#endif
constexpr int MMW = RB2_MMW;
template <bool STRIDE, typename P = uint64_t> __global__ __launch_bounds__(64 * MMW) void k_merge(const Ctl *ctl, const LeafDesc *__restrict__ LD, PoolView oldp, PoolView newp,
		const P *__restrict__ INS_E, const uint8_t *__restrict__ INS_A, uint16_t *RKREL, int compact_out, int par)
{
	__shared__ __align__(16) uint64_t lds[MMW][MergeLds<GPL>::WORDS];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ln = lane_id();
	// Compact windows only when the device itself knows that no string has a non-empty interval any more (ctl->ne of this round: once zero
	// it stays zero for the rest of the batch, so the next round's k_prep<false> -- the one reader of leaf words besides this kernel --
	// returns at once even if the host, which learns it rounds later, still launches it).  The host's consent (bit 0) covers what the
	// device cannot know: the last round of a batch and a pending change of layout.
	if (ctl->ne[par] != 0) compact_out &= ~1;
	// one window per wave; a rank of a sharded index launches fewer waves than the upper bound of its windows (the host does not
	// know the rank's share of the batch) and a wave then takes more than one: grid stride over the windows.  The first window's
	// work order is loaded together with the window count (LD holds an entry for every window a grid can name).
	uint64_t gw = (uint64_t)(STRIDE ? blockIdx.x : xcd_item()) * MMW + wv;
	LeafDesc d = LD[gw];
	const uint64_t nwin = ctl->wf0[NR];
	for (; gw < nwin; gw += (uint64_t)gridDim.x * MMW, d = LD[gw < nwin ? gw : 0]) {
		if ((d.nvalid & 0x3fffu) == WIN) merge_window<true, GPL, P>(d, lds[wv], ln, oldp, newp, INS_E, INS_A, RKREL, compact_out, (unsigned long long*)ctl->wfmt);
		else merge_window<false, GPL, P>(d, lds[wv], ln, oldp, newp, INS_E, INS_A, RKREL, compact_out, (unsigned long long*)ctl->wfmt);
		if (!STRIDE) return;                                    // (one GPU: the grid covers every window; no loop, no extra registers)
		if (gw + (uint64_t)gridDim.x * MMW < nwin) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }   // the wave's LDS arrays are reused
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
// fill_flag: FILL_P2 when this insert brings the leaf's first `$` / `N` -- the leaf has a plane-2 line from now on (rb2_device.h "two-plane
// leaves"); the bit travels with the fill, in the same atomic (the entry has one writer per round, and a fill never carries into bit 15)
__device__ __forceinline__ void dir_add_packed(const PoolView &pool, SbTot *sbtot, uint64_t gl, int role, uint32_t d01, uint32_t d23, uint32_t d45, uint32_t fill_flag = 0)
{
	const int s = role - 1;
	const uint32_t dw = s < 2 ? d01 : (s < 4 ? d23 : d45);
	uint32_t cnt = (dw >> ((uint32_t)(s & 1) * 16)) & 0xffffu;
	if (role == 0) cnt = ((d01 & 0xffffu) + (d01 >> 16) + (d23 & 0xffffu) + (d23 >> 16) + (d45 & 0xffffu) + (d45 >> 16)) | fill_flag;
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
struct RowOrd { uint32_t gl, ins0, i0, nn; };                               // the row's work order (SpOrd, rb2_device.h); nn = ni | nvalid << 16 (bit 31: the leaf has a plane-2 line), 0: none
struct RowJob { uint64_t w[3]; uint32_t pj, aj; };
__device__ __forceinline__ uint32_t ord_ni(const RowOrd &o) { return o.nn & 0xffffu; }

// (no use of the loaded words in here -- not even a select: the wave would wait for them, and with them for every load issued before, on the spot.
// r05 and before: `ni = ok ? a.w : 0` behind the load made every step of k_merge_leaf wait for the leaf lines it had just asked for -- one
// quad of leaves in flight per wave, nothing under way while it worked.  The caller clears nn of a row without an order when it takes the
// order into use, a step later.)
__device__ __forceinline__ bool row_ord_load(const SpOrd *LD, uint64_t g, uint32_t nwork, int ln, RowOrd &o)
{
	const uint64_t q = g + (uint32_t)(ln >> 4);
	const bool ok = q < nwork;
	const uint4 a = *(const uint4*)(LD + (ok ? q : 0));         // behind the end: a duplicate of the list's first order, loaded but never run
	o.gl = a.x; o.ins0 = a.y; o.i0 = a.z; o.nn = a.w;
	return ok;
}
template <typename P> __device__ __forceinline__ void row_job_load(const RowOrd &o, const int g, const PoolView &pool, const P *INS_E, const uint8_t *INS_A, RowJob &J)
{
	const uint64_t *lw = (const uint64_t*)pool.data + (uint64_t)o.gl * LEAFW + g;
	J.w[0] = RB2_LDNT(&lw[0]); J.w[1] = RB2_LDNT(&lw[LEAFG]);   // (nontemporal, like the stores: see the end of the loop)
	J.w[2] = 0;
	if (o.nn >> 31) J.w[2] = RB2_LDNT(&lw[2 * LEAFG]);         // the third line only of a leaf that holds a `$` or an `N` (two-plane leaves, rb2_device.h)
	// no use of a loaded value in here: the loads of the quad are to be in flight together (lanes >= ni load the
	// row's last insert again; they never use it)
	// (a leaf that takes ONE symbol has it in its order: SpOrd::i0 -- no gather, no sector of INS_E / INS_A moved for it)
	const uint64_t q = (uint64_t)o.ins0 + (uint32_t)min(g, max((int)ord_ni(o), 1) - 1);
	J.aj = (o.i0 >> 12) & 7u; J.pj = o.i0 & 0xfffu;
	if (ord_ni(o) > 1u) { J.aj = INS_A[q]; J.pj = (sizeof(P) == 4 ? ((const uint32_t*)INS_E)[q] : ((const uint32_t*)INS_E)[2 * q]) - o.i0 + (uint32_t)g; }   // low half: positions inside a leaf need no more; pj = E[q] + q - (leaf start + first slot)
}

#ifndef RB2_LQ
#define RB2_LQ 2
#endif
constexpr int LQ = RB2_LQ;                  // quads (of LROWS leaves) a wave has in flight per step
#ifndef RB2_LEAF_WAVES
#define RB2_LEAF_WAVES 6
#endif
// What bounds the kernel is how many leaf lines a wave keeps IN FLIGHT, not its instructions (77 VALU per quad) and not yet the bytes: on
// gfx9 one counter (vmcnt) covers loads AND stores, loads return in order but stores and atomics complete out of order with them, so every
// wait for a load is a wait for everything outstanding (s_waitcnt vmcnt(0)) -- a software pipeline that keeps the NEXT quad's loads in
// flight across this quad's stores (rounds 4-5) ends every step waiting for its own stores to be acknowledged, about as long as a load
// takes: one quad per memory round trip and wave, 127 us per million leaves.  A step now asks for LQ quads at once -- their leaf lines,
// their insert records, the work orders of the step after -- waits ONCE, and works the quads off one after the other; the stores of a
// step drain while the next step's loads are under way.  Per quad: a round trip / LQ + the work.
template <typename P = uint64_t> __global__ __launch_bounds__(256, RB2_LEAF_WAVES) void k_merge_leaf(const Ctl *ctl, const SpOrd *__restrict__ LD, PoolView pool,
		const P *INS_E, const uint8_t *INS_A /* not __restrict__: the loads are to stay where they are issued */, uint16_t *RKREL, SbTot *sbtot)
{
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int ln = lane_id(), g = ln & 15;
	// the round's work orders are WLC lists (Ctl::wcnt): wave gw takes list gw % WLC and walks it with the stride of the waves that
	// share it -- one counter to read per wave, and no search for "the q-th order of the round"
	const uint32_t gw = blockIdx.x * MW + (uint32_t)wv, per = gridDim.x * MW / WLC;   // (the host launches at least WLC waves)
	const uint32_t lc = gw % WLC, wi = gw / WLC;
	const uint64_t stride = (uint64_t)per * (LROWS * LQ);       // a wave takes LROWS * LQ consecutive orders of its list per step
	uint64_t g0 = (uint64_t)wi * (LROWS * LQ);
	const uint32_t nwork = ctl->wcnt[lc * WLS];
	LD += (uint64_t)lc * ctl->wstride;
	if (ctl->overflow || wi >= per || g0 >= nwork) return;
	RowOrd o[LQ], on[LQ];
	RowJob J[LQ];
	bool on_ok[LQ];
#pragma unroll
	for (int k = 0; k < LQ; ++k) on_ok[k] = row_ord_load(LD, g0 + (uint64_t)(k * LROWS), nwork, ln, on[k]);
	// (waited for here: left pending into the loop, the compiler guards the loop's first use of on[] with a wait that -- coming round again -- is a wait for
	// the stores of the step before)
#pragma unroll
	for (int k = 0; k < LQ; ++k) asm volatile("" : "+v"(on[k].gl), "+v"(on[k].ins0), "+v"(on[k].i0), "+v"(on[k].nn));
	for (;;) {
		// the orders of this step arrived with the loads of the step before (or in front of the loop: the first use waits for them)
#pragma unroll
		for (int k = 0; k < LQ; ++k) { o[k] = on[k]; if (!on_ok[k]) o[k].nn = 0; }   // (nn = 0: a row behind the end of the list)
		const uint64_t g1 = g0 + stride;
		const bool more = g1 < nwork;
#pragma unroll
		for (int k = 0; k < LQ; ++k) row_job_load<P>(o[k], g, pool, INS_E, INS_A, J[k]);
		if (more) {
#pragma unroll
			for (int k = 0; k < LQ; ++k) on_ok[k] = row_ord_load(LD, g1 + (uint64_t)(k * LROWS), nwork, ln, on[k]);
		}
		// ONE wait for everything asked for above (the first look at any loaded word waits for all of them: see the head of the kernel)
#pragma unroll
		for (int k = 0; k < LQ; ++k) asm volatile("" : "+v"(J[k].w[0]), "+v"(J[k].w[1]), "+v"(J[k].w[2]), "+v"(J[k].pj), "+v"(J[k].aj));
#pragma unroll
		for (int k = 0; k < LQ; ++k) {
		const RowOrd &oo = o[k];
		uint64_t w0 = J[k].w[0], w1 = J[k].w[1], w2 = J[k].w[2];
		uint32_t pjr = J[k].pj, aj = J[k].aj;
		const uint32_t oni = ord_ni(oo);
		bool p2 = (oo.nn >> 31) != 0;                             // the leaf has a plane-2 line (row-uniform)
		if (!p2) {                                                // two planes tell its symbols: T is "neither bit set", on the positions in use
			const int nold = (int)((oo.nn >> 16) & FILL_MASK) - (int)oni;
			w2 = ~(w0 | w1) & bits_below((uint32_t)min(GSYM, max(0, nold - g * GSYM)));
		}
		const uint32_t nimax = max(max((uint32_t)__builtin_amdgcn_readlane((int)oni, 0), (uint32_t)__builtin_amdgcn_readlane((int)oni, 16)),
				max((uint32_t)__builtin_amdgcn_readlane((int)oni, 32), (uint32_t)__builtin_amdgcn_readlane((int)oni, 48)));
		for (uint32_t c0 = 0; c0 < nimax; c0 += LTURN) {            // turns of LTURN inserts per row (one turn, normally)
			if (c0) {                                               // (rare) the row's next inserts
				const uint64_t q = (uint64_t)oo.ins0 + min(c0 + (uint32_t)g, max(oni, 1u) - 1u);
				aj = INS_A[q]; pjr = sizeof(P) == 4 ? ((const uint32_t*)INS_E)[q] : ((const uint32_t*)INS_E)[2 * q];
				asm volatile("" : "+v"(aj), "+v"(pjr));             // waited for HERE, on the rare path: a wait at the first use, behind the join, would make the common path wait as well
				pjr = pjr - oo.i0 + c0 + (uint32_t)g;
			}
			const uint32_t nic = oni > c0 ? min(oni - c0, (uint32_t)LTURN) : 0u;   // what my row inserts in this turn (row-uniform)
			const uint32_t pj = pjr;                                // final position E[q] + q inside the leaf (lanes >= nic: unused)
			const uint32_t ncmax = min(nimax - c0, (uint32_t)LTURN);
			const bool mine = (uint32_t)g < nic;
			// what a leaf receives is known before the first symbol is placed: the directory atomics go out first and are under way
			// while the wave shifts words
			{
				const uint32_t rsh = (uint32_t)(ln & 48);
				uint32_t cs[6];
#pragma unroll
				for (int s = 0; s < 6; ++s) cs[s] = (uint32_t)__popc((uint32_t)(__ballot(mine && aj == (uint32_t)s) >> rsh) & 0xffffu);
				const bool first_x = !p2 && (cs[0] | cs[5]) != 0;       // the leaf's first `$` / `N`: it has a plane-2 line from here on
				dir_add_packed(pool, sbtot, oo.gl, nic ? g : 16, cs[0] | cs[1] << 16, cs[2] | cs[3] << 16, cs[4] | cs[5] << 16, first_x ? FILL_P2 : 0u);
				p2 = p2 || first_x;
			}
			// RKREL = the rank of the symbol inside the leaf AS IT WAS before the round (what rope_insert_run's descent ends with: rle_insert_cached's
			// count, rle.c:86-88; k_part_sparse took the part in front of the leaf from the directory before the round: RKOLD) = the rank in the
			// leaf as it is when the symbol goes in, minus the inserts of the same symbol the leaf took before it this round: the ones of this
			// turn are counted off as they go by, the ones of earlier turns (rare) are read again
			uint32_t myrank = 0;
			if (c0) {
				const uint32_t rsh = (uint32_t)(ln & 48);
				uint32_t cnt = 0;
				for (uint32_t t0 = 0; t0 < c0; t0 += 16) {
					const uint32_t idx = t0 + (uint32_t)g;
					uint32_t av = 7u;
					if (idx < c0 && idx < oni) { av = (uint32_t)INS_A[(uint64_t)oo.ins0 + idx]; asm volatile("" : "+v"(av)); }
#pragma unroll
					for (int s = 0; s < 6; ++s) { const uint32_t bm = (uint32_t)(__ballot(av == (uint32_t)s) >> rsh) & 0xffffu; if (aj == (uint32_t)s) cnt += (uint32_t)__popc(bm); }
				}
				myrank = 0u - cnt;
			}
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
				if (act && g == j) myrank += r;
				if (mine && g > j && aj == a) --myrank;               // (an insert of my symbol in front of mine)
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
			if (mine) RKREL[(uint64_t)oo.ins0 + c0 + g] = (uint16_t)myrank;
		}
		// The WHOLE leaf goes back, full 128-byte lines -- two, or three when it holds a `$` or an `N` --, with nontemporal stores behind nontemporal
		// loads: the leaf streams through once and nothing of it has to wait in L2 for a partial line to be merged.  (Rounds 2-4 stored only the
		// groups from the first changed one on, counting on the lines the load had left in L2: 1 M x 10 kbp 3.09-3.19 s, this way 2.72-2.75 s
		// on one box.  Each half alone is no gain: whole lines with plain accesses 3.09, nontemporal accesses with partial lines 3.31-3.39.)
		if (oni) {
			uint64_t *lw = (uint64_t*)pool.data + (uint64_t)oo.gl * LEAFW + g;
			RB2_STNT(w0, &lw[0]); RB2_STNT(w1, &lw[LEAFG]);
			if (p2) RB2_STNT(w2, &lw[2 * LEAFG]);
		}
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
