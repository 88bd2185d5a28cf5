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
//      contiguous byte stream; SWAR + DPP wave scan give the first symbol of every 16-byte chunk
//      (leaf slots are zero padded to 16 bytes, and a zero byte is a run of length 0)
//   2. the new symbols of the leaf are scattered into a position-indexed LDS array + bit flags
//   3. wave prefix sum of the non-insert counts -> first old symbol each lane consumes; a 7-step
//      search finds the chunk, SWAR prefix sums the byte and the offset inside its run
//   4. 16 output symbols per lane are produced from a 16-byte shift register of run bytes
//      (rle_dec1, rle.h:39-51) with the flagged positions taken from the insert array
//   5. re-encode: run heads by packed neighbour compare, wave scan for byte offsets, one byte per
//      run (rle_enc1's 1-byte form, rle.h:55-57); runs longer than 15 take a slower exact path
//   6. symbol counts of the leaf (new LeafMeta) and, for every new symbol, the number of equal
//      symbols before it INSIDE the leaf (RKREL); k_advance adds the directory prefix of the new
//      rope to obtain the reference's return value of rope_insert_run.
// The code is written branch-free on purpose: per-position conditions become selects, so the
// compiler does not fragment the 16-fold unrolled loops into exec-mask regions.
#pragma once
#include "rb2_device.h"

namespace rb2 {

constexpr int MW = 4;                       // waves (= output leaves) per block

struct WaveLds {
	uint8_t  raw[2 * LEAF + 64];            // old run bytes: leaf A then leaf B, contiguous
	uint8_t  ins[LEAF];                     // new symbol at output position p (where flagged)
	uint8_t  outb[LEAF + 16];               // encoded output leaf (+ one dump slot)
	uint16_t starts[128];                   // first symbol (in A|B coordinates) of each 16-byte chunk
	uint32_t flags[LEAF / 32];              // bit p: output position p is a new symbol
};

__device__ __forceinline__ uint32_t byte_of(const uint32_t w[4], int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }
__device__ __forceinline__ uint32_t sel4(const uint32_t w[4], uint32_t k) { return k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3]; }

// sum of the run lengths of a 16-byte chunk (SWAR; zero bytes count 0)
__device__ __forceinline__ uint32_t chunk_len_sum(const uint32_t w[4])
{
	uint32_t s = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) s += ((((w[k] >> 3) & 0x1f1f1f1fu) * 0x01010101u) >> 24);
	return s;
}

// 16 bytes from an arbitrary LDS byte address: five aligned dwords + funnel shifts
__device__ __forceinline__ void lds_read16(const uint8_t *p, uint32_t out[4])
{
	const uint32_t a = (uint32_t)(uintptr_t)p;
	const uint32_t *q = (const uint32_t*)(p - (a & 3));
	const uint32_t sh = (a & 3) * 8;
	const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
	out[0] = __builtin_amdgcn_alignbit(d1, d0, sh); out[1] = __builtin_amdgcn_alignbit(d2, d1, sh);
	out[2] = __builtin_amdgcn_alignbit(d3, d2, sh); out[3] = __builtin_amdgcn_alignbit(d4, d3, sh);
}

// per-lane symbol counters: six 5-bit fields (a lane holds <= 16 symbols)
__device__ __forceinline__ uint32_t c5_one(uint32_t sym) { return 1u << (sym * 5u); }
__device__ __forceinline__ uint32_t c5_get(uint32_t acc, uint32_t sym) { return (acc >> (sym * 5u)) & 31u; }

__global__ __launch_bounds__(256) void k_merge(const Ctl *ctl, int side, PoolView oldp, PoolView newp,
		const uint64_t *INS_E, const uint8_t *INS_A, uint16_t *RKREL, const uint32_t *TQ, int dbg)
{
	__shared__ __align__(16) WaveLds lds[MW];
	const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	WaveLds &L = lds[wv];
	const int ln = lane_id();
	const uint64_t gleaf = (uint64_t)blockIdx.x * MW + wv;
	if (gleaf >= ctl->lf0[NR]) return;
	const int b = seg_of(ctl->lf0, gleaf);
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
	int nbA = 0;
	if (haveA) {
		const uint64_t gl = orp.leaf0 + A;
		nbA = oldp.meta[gl].nbytes;
		if (ln * 16 < nbA) { const uint4 v = ((const uint4*)(oldp.data + gl * (uint64_t)LEAF))[ln]; wa[0] = v.x; wa[1] = v.y; wa[2] = v.z; wa[3] = v.w; }
	}
	if (needB) {
		const uint64_t gl = orp.leaf0 + A + 1;
		const int nbB = oldp.meta[gl].nbytes;
		if (ln * 16 < nbB) { const uint4 v = ((const uint4*)(oldp.data + gl * (uint64_t)LEAF))[ln]; wb[0] = v.x; wb[1] = v.y; wb[2] = v.z; wb[3] = v.w; }
	}
	if (ln < LEAF / 32) L.flags[ln] = 0;
	((uint4*)L.outb)[ln] = make_uint4(0, 0, 0, 0);             // the stored leaf is zero padded to 16 bytes
	((uint4*)L.raw)[ln] = make_uint4(wa[0], wa[1], wa[2], wa[3]);
	{
		uint8_t *d = L.raw + nbA + ln * 16;                    // B right behind the used bytes of A
#pragma unroll
		for (int i = 0; i < 16; ++i) d[i] = (uint8_t)byte_of(wb, i);
	}
	{
		const uint32_t sa = chunk_len_sum(wa), sb = chunk_len_sum(wb);
		const uint32_t ia = dpp_incl_add(sa), ib = dpp_incl_add(sb);
		const uint32_t totA = lane63(ia);
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
	const uint32_t vmask = (1u << myvalid) - 1u;
	const uint32_t flags = ((const uint16_t*)L.flags)[ln] & vmask;
	const uint32_t kins = __popc(flags);
	const uint32_t nonins = (uint32_t)myvalid - kins;
	const uint32_t oinc = dpp_incl_add(nonins);
	const uint32_t iinc = dpp_incl_add(kins);
	const uint32_t oldpos = (uint32_t)x0 + oinc - nonins;      // first old symbol of this lane, A|B coordinates
	uint32_t sr[4];                                            // shift register of upcoming run bytes
	int rem; uint32_t cs;
	{
		uint32_t c = 0;
#pragma unroll
		for (uint32_t st = 64; st >= 1; st >>= 1) c += (L.starts[c + st] <= (uint16_t)oldpos) ? st : 0u;   // starts[] is non-decreasing
		const uint32_t off = oldpos - L.starts[c];
		const uint32_t g0 = c < 64 ? c * 16 : (uint32_t)nbA + (c - 64) * 16;
		uint32_t cw[4], pre[4];
		lds_read16(L.raw + g0, cw);
#pragma unroll
		for (int k = 0; k < 4; ++k) pre[k] = ((cw[k] >> 3) & 0x1f1f1f1fu) * 0x01010101u;   // in-dword inclusive prefix of the run lengths
		const uint32_t t0 = pre[0] >> 24, t1 = t0 + (pre[1] >> 24), t2 = t1 + (pre[2] >> 24);
		const uint32_t wsel = (off >= t0) + (off >= t1) + (off >= t2);                       // dword that holds symbol `off`
		const uint32_t wbase = wsel == 0 ? 0u : wsel == 1 ? t0 : wsel == 2 ? t1 : t2;
		const uint32_t offw = off - wbase, P = sel4(pre, wsel);
		const uint32_t K = (255u - offw) * 0x00010001u;                                      // per byte: P > offw ?
		const uint32_t ge = (((P & 0x00ff00ffu) + K) >> 8) & 0x00010001u, go = ((((P >> 8) & 0x00ff00ffu) + K) >> 8) & 0x00010001u;
		const uint32_t ngt = ((ge + go) & 0xffffu) + ((ge + go) >> 16);
		const uint32_t bw = 4u - ngt;                                                        // byte inside the dword
		const uint32_t bi = wsel * 4u + bw;
		const uint32_t exb = ((P << 8) >> (bw * 8u)) & 0xffu;                                // symbols of the dword before that byte
		lds_read16(L.raw + g0 + bi, sr);
		rem = (int)((sr[0] & 0xffu) >> 3) - (int)(offw - exb);
		cs = sr[0] & 7u;
		rem = nonins ? rem : 16;                               // lanes that consume nothing never advance
	}
	// ---- 4. my 16 output symbols, packed 4 per dword
	uint32_t pw[4] = {0, 0, 0, 0};
	{
		uint32_t iw[4];
		const uint4 v = ((const uint4*)L.ins)[ln];
		iw[0] = v.x; iw[1] = v.y; iw[2] = v.z; iw[3] = v.w;
		const uint32_t fx = flags | ~vmask;                    // positions past the end behave like inserts of 0xff
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const uint32_t fl = (fx >> i) & 1u;
			const uint32_t insb = i < myvalid ? byte_of(iw, i) : 0xffu;
			const bool need = !fl && rem == 0;                 // next run byte
			const uint32_t sh = need ? 8u : 0u;
			sr[0] = __builtin_amdgcn_alignbit(sr[1], sr[0], sh); sr[1] = __builtin_amdgcn_alignbit(sr[2], sr[1], sh);
			sr[2] = __builtin_amdgcn_alignbit(sr[3], sr[2], sh); sr[3] >>= sh;
			rem = need ? (int)((sr[0] & 0xffu) >> 3) : rem;
			cs = need ? (sr[0] & 7u) : cs;
			const uint32_t sym = fl ? insb : cs;
			rem -= (int)(fl ^ 1u);
			pw[i >> 2] |= sym << ((i & 3) * 8);
		}
	}
	// ---- 5. re-encode
	uint32_t hm = 0;                                           // bit i: a run starts at my position i
	uint32_t prevw = dpp_prev_lane(pw[3]);                     // NB: cross-lane reads stay outside of lane-dependent conditionals
	if (ln == 0) prevw = 0xff000000u;
	{
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const uint32_t ps = __builtin_amdgcn_alignbyte(pw[k], k == 0 ? prevw : pw[k - 1], 3);   // my symbols shifted by one position
			const uint32_t x = pw[k] ^ ps;
			const uint32_t nz = ((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7) & 0x01010101u;       // 1 per differing byte
			hm |= ((nz * 0x01020408u) >> 24) << (4 * k);
		}
		hm &= vmask;
	}
	uint32_t pm = dpp_prev_lane(hm);
	uint32_t cov = pm | (hm << 16);
	cov |= cov << 1; cov |= cov << 2; cov |= cov << 4; cov |= cov << 7;   // bit set: a run start within the 14 positions before
	const bool short_runs = ((cov >> 16) & vmask) == vmask;
	uint32_t nbytes;
	if (__builtin_expect(__all(short_runs) && !(dbg & 1), 1)) {
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
			const uint32_t idx = (hm >> i & 1u) ? hb + __popc(hm & ((1u << i) - 1u)) : (uint32_t)LEAF;   // non-heads go to the dump slot
			L.outb[idx] = (uint8_t)(len << 3 | byte_of(pw, i));
		}
	} else {
		// exact path for runs longer than 15 symbols: a byte boundary every 15 symbols of a run.
		// Rare on reads without long homopolymers; kept in the plain (branchy, shuffle-based) form.
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
			uint32_t pv = prevsym; int r = rs, seen = 0;
			for (int i = 0; i < 16; ++i) if (i < myvalid) {
				const int p = p0 + i;
				if (hm >> i & 1u) r = p;
				if ((hm >> i & 1u) || (p - r) % 15 == 0) {
					if (p != 0) L.outb[hb + seen - 1] = (uint8_t)((p - lh) << 3 | pv);
					lh = p; ++seen;
				}
				pv = byte_of(pw, i);
			}
			if (myvalid > 0 && p0 + myvalid == nvalid) L.outb[nbytes - 1] = (uint8_t)((nvalid - lh) << 3 | pv);
		}
	}
	// ---- 6. counts of the leaf, leaf-relative ranks of the new symbols
	uint32_t c5 = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) c5 += i < myvalid ? c5_one(byte_of(pw, i)) : 0u;
	// widen to 16-bit fields for the wave scan: (sym0,sym1) (sym2,sym3) (sym4,sym5)
	const uint32_t e01 = (c5 & 31u) | ((c5 >> 5 & 31u) << 16), e23 = (c5 >> 10 & 31u) | ((c5 >> 15 & 31u) << 16), e45 = (c5 >> 20 & 31u) | ((c5 >> 25 & 31u) << 16);
	const uint32_t s01 = dpp_incl_add(e01), s23 = dpp_incl_add(e23), s45 = dpp_incl_add(e45);
	if (__any(kins != 0)) {
		uint32_t f = flags, n = 0;
		uint16_t *dst = RKREL + segs + q0 + (iinc - kins);
		while (f) {                                            // few iterations: new symbols are sparse in steady state
			const uint32_t i = (uint32_t)__builtin_ctz(f);
			f &= f - 1;
			const uint32_t a = (sel4(pw, i >> 2) >> ((i & 3) * 8)) & 0xffu;
			// equal symbols in front of position i inside my 16: packed compare + mask
			uint32_t cnt = 0;
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t x = pw[k] ^ (a * 0x01010101u);
				const uint32_t eq = ~((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x)) & 0x80808080u;       // 0x80 per equal byte
				const int nb = min(4, max(0, (int)i - 4 * k));                                       // bytes of this dword in front of i
				cnt += __popc(eq & (nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u)));
			}
			const uint32_t w2 = a < 2 ? s01 - e01 : a < 4 ? s23 - e23 : s45 - e45;
			dst[n++] = (uint16_t)(((w2 >> ((a & 1) * 16)) & 0xffffu) + cnt);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
	{
		const uint64_t gl = nrp.leaf0 + j;
		if (ln == 63) {
			LeafMeta m;
			m.c[0] = (uint16_t)s01; m.c[1] = (uint16_t)(s01 >> 16); m.c[2] = (uint16_t)s23; m.c[3] = (uint16_t)(s23 >> 16);
			m.c[4] = (uint16_t)s45; m.c[5] = (uint16_t)(s45 >> 16);
			m.nbytes = (uint16_t)nbytes; m.pad = 0;
			newp.meta[gl] = m;                                 // own counts; k_meta_sb turns them into prefixes
		}
		if ((uint32_t)(ln * 16) < nbytes) ((uint4*)(newp.data + gl * (uint64_t)LEAF))[ln] = ((const uint4*)L.outb)[ln];
	}
}

} // namespace rb2
