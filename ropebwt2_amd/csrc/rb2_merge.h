// rb2_merge.h -- k_merge: rank + positional insert over the run-length leaves of one rope.
//
// Reference semantics: rope_insert_run (rope.c:114-148) -> rle_insert_cached (rle.c:10-89): put
// rl copies of symbol a in front of position x and return the number of a's before x.  The
// reference does this one run at a time through a B+ tree; here one launch rewrites the whole
// rope side -> side^1 as a merge of two sorted sequences (old symbols, new symbols).
//
// Work decomposition: ONE WAVE PER OUTPUT LEAF (LEAF symbols), four independent waves per block,
// no block-level barrier anywhere.  Lane l owns output positions [16l, 16l+16) of the leaf:
//   1. the <= 2 old leaves that feed this leaf are loaded 16 B per lane and staged in LDS as one
//      contiguous byte stream; SWAR + wave scan give the first symbol of every 16-byte chunk
//   2. the new symbols of the leaf are scattered into a position-indexed LDS array + bit flags
//   3. wave prefix sum of the non-insert counts -> first old symbol each lane consumes; a 7-step
//      search finds the chunk, a register walk the byte and the offset inside its run
//   4. 16 output symbols per lane are produced from a 16-byte shift register of run bytes
//      (rle_dec1, rle.h:39-51) with the flagged positions taken from the insert array
//   5. re-encode: run heads by neighbour compare, wave scan for byte offsets, one byte per run
//      (rle_enc1's 1-byte form, rle.h:55-57); runs longer than 15 take a slower exact path
//   6. symbol counts of the leaf (new LeafMeta) and, for every new symbol, the number of equal
//      symbols before it INSIDE the leaf (RKREL); k_advance adds the directory prefix of the new
//      rope to obtain the reference's return value of rope_insert_run.
#pragma once
#include "rb2_device.h"

namespace rb2 {

constexpr int MW = 4;                       // waves (= output leaves) per block

struct WaveLds {
	uint8_t  raw[2 * LEAF + 32];            // old run bytes: leaf A then leaf B, contiguous
	uint8_t  ins[LEAF];                     // new symbol at output position p (where flagged)
	uint8_t  outb[LEAF];                    // encoded output leaf
	uint16_t starts[128];                   // first symbol (in A|B coordinates) of each 16-byte chunk
	uint32_t flags[LEAF / 32];              // bit p: output position p is a new symbol
};

// packed per-symbol counters: symbols 0..4 in 12-bit fields; N (5) is derived from the position
__device__ __forceinline__ uint64_t pk_add(uint64_t acc, uint32_t sym, uint32_t len) { return sym < 5 ? acc + ((uint64_t)len << (12 * sym)) : acc; }
__device__ __forceinline__ uint32_t pk_get(uint64_t acc, uint32_t sym) { return (uint32_t)(acc >> (12 * sym)) & 0xfffu; }
__device__ __forceinline__ uint32_t pk_sum5(uint64_t acc) { return pk_get(acc, 0) + pk_get(acc, 1) + pk_get(acc, 2) + pk_get(acc, 3) + pk_get(acc, 4); }

__device__ __forceinline__ uint32_t byte_of(const uint32_t w[4], int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }

// sum of the run lengths of the first nv bytes of a 16-byte chunk (SWAR)
__device__ __forceinline__ uint32_t chunk_len_sum(const uint32_t w[4], int nv)
{
	uint32_t s = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int v = min(4, max(0, nv - 4 * k));
		const uint32_t m = v >= 4 ? 0xffffffffu : ((1u << (8 * v)) - 1u);
		const uint32_t l4 = ((w[k] & m) >> 3) & 0x1f1f1f1fu;
		s += (l4 * 0x01010101u) >> 24;
	}
	return s;
}

__global__ __launch_bounds__(256) void k_merge(const Ctl *ctl, int side, PoolView oldp, PoolView newp,
		const uint64_t *INS_E, const uint8_t *INS_A, uint16_t *RKREL, const uint32_t *TQ)
{
	__shared__ __align__(16) WaveLds lds[MW];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	WaveLds &L = lds[wv];
	const int ln = lane_id();
	const uint64_t gleaf = (uint64_t)blockIdx.x * MW + wv;
	if (gleaf >= ctl->lf0[6]) return;
	int b = 0;
	while (gleaf >= ctl->lf0[b+1]) ++b;
	const uint64_t j = gleaf - ctl->lf0[b];
	const RopeDesc &orp = ctl->rope[side][b], &nrp = ctl->rope[side ^ 1][b];
	const uint64_t segs = ctl->seg[side].start[b];
	const uint32_t q0 = TQ[gleaf + b], q1 = TQ[gleaf + b + 1];
	const uint64_t o0 = j * LEAF;
	const int nvalid = (int)min((uint64_t)LEAF, nrp.n - o0);
	const int ni = (int)(q1 - q0);
	const int nold = nvalid - ni;                              // old symbols consumed by this leaf
	const uint64_t i0 = o0 - q0;
	const uint64_t A = i0 / LEAF;
	const int x0 = (int)(i0 % LEAF);
	const bool haveA = nold > 0;
	const bool needB = haveA && (x0 + nold > LEAF);

	// ---- 1. old leaves -> registers -> LDS byte stream; chunk starts
	uint32_t wa[4] = {0, 0, 0, 0}, wb[4] = {0, 0, 0, 0};
	int nbA = 0, nva = 0, nvb = 0;
	if (haveA) {
		const uint64_t gl = orp.leaf0 + A;
		nbA = oldp.meta[gl].nbytes;
		nva = min(16, max(0, nbA - ln * 16));
		if (nva > 0) { const uint4 v = ((const uint4*)(oldp.data + gl * (uint64_t)LEAF))[ln]; wa[0] = v.x; wa[1] = v.y; wa[2] = v.z; wa[3] = v.w; }
	}
	if (needB) {
		const uint64_t gl = orp.leaf0 + A + 1;
		const int nbB = oldp.meta[gl].nbytes;
		nvb = min(16, max(0, nbB - ln * 16));
		if (nvb > 0) { const uint4 v = ((const uint4*)(oldp.data + gl * (uint64_t)LEAF))[ln]; wb[0] = v.x; wb[1] = v.y; wb[2] = v.z; wb[3] = v.w; }
	}
	if (ln < LEAF / 32) L.flags[ln] = 0;
	((uint4*)L.raw)[ln] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
	{
		uint8_t *d = L.raw + nbA + ln * 16;                    // B right behind the used bytes of A
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < nvb) d[i] = (uint8_t)byte_of(wb, i);
	}
	{
		const uint32_t sa = chunk_len_sum(wa, nva), sb = chunk_len_sum(wb, nvb);
		const uint32_t ia = wave_incl_add(sa), ib = wave_incl_add(sb);
		const uint32_t totA = __shfl(ia, 63);
		L.starts[ln] = haveA ? (uint16_t)(ia - sa) : (uint16_t)0xffffu;
		L.starts[64 + ln] = needB ? (uint16_t)(totA + ib - sb) : (uint16_t)0xffffu;
	}
	// ---- 2. new symbols of this leaf
	for (int jj = ln; jj < ni; jj += 64) {
		const uint64_t e = INS_E[segs + q0 + jj];
		const uint32_t a = INS_A[segs + q0 + jj];
		const uint32_t p = (uint32_t)(e + q0 + jj - o0);
		L.ins[p] = (uint8_t)a;
		atomicOr(&L.flags[p >> 5], 1u << (p & 31));
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	// ---- 3. what does each lane consume
	const int p0 = ln * 16;
	const int myvalid = min(16, max(0, nvalid - p0));
	const uint32_t flags = ((const uint16_t*)L.flags)[ln] & ((1u << myvalid) - 1u);
	const int kins = __popc(flags);
	const int nonins = myvalid - kins;
	const int oinc = wave_incl_add(nonins);
	const int iinc = wave_incl_add(kins);
	const int oldpos = x0 + oinc - nonins;                     // first old symbol of this lane, A|B coordinates
	uint32_t sr[4] = {0, 0, 0, 0};                             // shift register of upcoming run bytes
	int rem = 0; uint32_t cs = 0;
	if (nonins > 0) {
		int c = 0;
#pragma unroll
		for (int st = 64; st >= 1; st >>= 1) if (L.starts[c + st] <= (uint16_t)oldpos) c += st;   // starts[] is non-decreasing
		const int off = oldpos - L.starts[c];
		const int g0 = c < 64 ? c * 16 : nbA + (c - 64) * 16;
		uint32_t cw[4];
		__builtin_memcpy(cw, L.raw + g0, 16);
		int bi = 0, acc = 0, dd = 0; bool found = false;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int len = (int)(byte_of(cw, i) >> 3);
			if (!found && acc + len > off) { bi = i; dd = off - acc; found = true; }
			acc += len;
		}
		__builtin_memcpy(sr, L.raw + g0 + bi, 16);
		rem = (int)((sr[0] & 0xff) >> 3) - dd; cs = sr[0] & 7;
	}
	// ---- 4. my 16 output symbols
	uint32_t sy[16];
	{
		const uint8_t *ip = L.ins + p0;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			uint32_t v = 0xff;
			if (i < myvalid) {
				if (flags >> i & 1) v = ip[i];
				else {
					if (rem == 0) {                                // next run byte
						sr[0] = __builtin_amdgcn_alignbyte(sr[1], sr[0], 1); sr[1] = __builtin_amdgcn_alignbyte(sr[2], sr[1], 1);
						sr[2] = __builtin_amdgcn_alignbyte(sr[3], sr[2], 1); sr[3] >>= 8;
						rem = (int)((sr[0] & 0xff) >> 3); cs = sr[0] & 7;
					}
					v = cs; --rem;
				}
			}
			sy[i] = v;
		}
	}
	// ---- 5. re-encode
	const uint32_t prevsym = __shfl_up(sy[15], 1);
	uint32_t hm = 0;                                               // bit i: a run starts at my position i
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const uint32_t pv = i == 0 ? (ln == 0 ? 0xffu : prevsym) : sy[i - 1];
		if (i < myvalid && sy[i] != pv) hm |= 1u << i;
	}
	const uint32_t vmask = (1u << myvalid) - 1u;
	uint32_t pm = __shfl_up(hm, 1);
	if (ln == 0) pm = 0;
	uint32_t cov = pm | (hm << 16);
	cov |= cov << 1; cov |= cov << 2; cov |= cov << 4; cov |= cov << 7;   // bit set: a run start within the 14 positions before
	const bool short_runs = ((cov >> 16) & vmask) == vmask;
	int nbytes;
	if (__builtin_expect(__all(short_runs), 1)) {
		const int nh = __popc(hm);
		const int hinc = wave_incl_add(nh);
		int hb = hinc - nh;
		nbytes = __shfl(hinc, 63);
		const int nvnext = __shfl_down(myvalid, 1), hmnext = __shfl_down((int)hm, 1);
		int tail = 0;                                              // symbols of my last run that live in the next lane
		if (ln < 63 && nvnext > 0) tail = hmnext ? __builtin_ctz(hmnext) : nvnext;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			if (hm >> i & 1) {
				const uint32_t t = hm >> (i + 1);
				const int len = t ? __builtin_ctz(t) + 1 : myvalid - i + tail;
				L.outb[hb++] = (uint8_t)(len << 3 | sy[i]);
			}
		}
	} else {
		// exact path for runs longer than 15 symbols: a byte boundary every 15 symbols of a run
		int lastnat = -1;
#pragma unroll
		for (int i = 0; i < 16; ++i) if (hm >> i & 1) lastnat = p0 + i;
		const int incmax = wave_incl_max(lastnat);
		int rs = __shfl_up(incmax, 1);                             // start of the run open at p0-1
		if (ln == 0) rs = 0;
		int lh = ln == 0 ? 0 : rs + (p0 - 1 - rs) / 15 * 15;       // last byte boundary before p0
		int hc = 0;
		{
			int since = ln == 0 ? 0 : (p0 - 1 - rs) % 15 + 1;      // symbols since the last boundary, at p0
#pragma unroll
			for (int i = 0; i < 16; ++i) if (i < myvalid) {
				const bool nat = hm >> i & 1;
				if (nat || since == 15) { ++hc; since = 0; }
				++since;
			}
		}
		const int hinc = wave_incl_add(hc);
		const int hb = hinc - hc;
		nbytes = __shfl(hinc, 63);
		{
			int since = ln == 0 ? 0 : (p0 - 1 - rs) % 15 + 1, seen = 0;
			uint32_t pv = ln == 0 ? 0xffu : prevsym;
#pragma unroll
			for (int i = 0; i < 16; ++i) if (i < myvalid) {
				const int p = p0 + i;
				const bool nat = hm >> i & 1;
				if (nat || since == 15) {
					if (p != 0) L.outb[hb + seen - 1] = (uint8_t)((p - lh) << 3 | pv);
					lh = p; ++seen; since = 0;
				}
				++since; pv = sy[i];
			}
			if (myvalid > 0 && p0 + myvalid == nvalid) L.outb[nbytes - 1] = (uint8_t)((nvalid - lh) << 3 | pv);
		}
	}
	// ---- 6. counts of the leaf, leaf-relative ranks of the new symbols
	uint64_t tot = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) if (i < myvalid) tot = pk_add(tot, sy[i], 1);
	const uint64_t tinc = wave_incl_add(tot);
	if (kins > 0) {
		uint64_t rc = tinc - tot;
		uint16_t *dst = RKREL + segs + q0 + (iinc - kins);
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < myvalid) {
			if (flags >> i & 1) *dst++ = (uint16_t)(sy[i] < 5 ? pk_get(rc, sy[i]) : (uint32_t)(p0 + i) - pk_sum5(rc));
			rc = pk_add(rc, sy[i], 1);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	{
		const uint64_t gl = nrp.leaf0 + j;
		if (ln == 63) {
			LeafMeta m;
#pragma unroll
			for (int s = 0; s < 5; ++s) m.c[s] = (uint16_t)pk_get(tinc, s);
			m.c[5] = (uint16_t)((uint32_t)nvalid - pk_sum5(tinc));
			m.nbytes = (uint16_t)nbytes; m.pad = 0;
			newp.meta[gl] = m;                                     // own counts; k_meta_sb turns them into prefixes
		}
		if (ln * 16 < nbytes) ((uint4*)(newp.data + gl * (uint64_t)LEAF))[ln] = ((const uint4*)L.outb)[ln];
	}
}

} // namespace rb2
