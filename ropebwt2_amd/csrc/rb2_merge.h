// rb2_merge.h -- k_merge: rank + positional insert over the run-length leaves of one rope.
//
// Reference semantics: rope_insert_run (rope.c:114-148) -> rle_insert_cached (rle.c:10-89): put
// rl copies of symbol a in front of position x, return the number of a's before x.  The
// reference does this one run at a time through a B+ tree; here one launch rewrites the whole
// rope side -> side^1 as a merge of two sorted sequences (old symbols, new symbols), one block
// per MT output symbols (TL leaves).
//
// Two code paths inside the block, chosen per tile:
//   sparse  (<= NI_SPARSE new symbols in the tile, the steady state of every batch but the
//           first rounds of the first): BYTE level.  Old run bytes are copied verbatim; only the
//           few runs hit by an insert or by a new leaf boundary are re-cut by one lane each.
//   dense   symbol level: decode to one symbol per byte in LDS, splice, re-encode canonically.
#pragma once
#include "rb2_device.h"

namespace rb2 {

constexpr int NI_SPARSE = 58;               // max inserts for the sparse path (events fit one wave)
constexpr int PCMAX     = 10;               // max output bytes one re-cut run may produce
constexpr int NOL       = TL + 1;           // old leaves a tile can touch
constexpr uint16_t GK_TAIL = 0xFFFF;        // "byte" after the last loaded old byte

struct DenseLds {
	uint8_t  old[NOL * LEAF];
	uint8_t  out[MT];
	uint8_t  bytes[MT];
	uint32_t flag[MT / 32];
	uint64_t cplo[256], cphi[256];
	uint64_t w64[4];
	uint32_t w32[4];
	uint64_t base[6];
};

struct SparseLds {
	uint8_t  raw[NOL * LEAF];               // old leaf bytes, leaf li at li*LEAF
	uint8_t  outb[TL * LEAF];               // new leaf bytes, leaf l at l*LEAF
	uint64_t lcnt[NOL][64];                 // per 16-byte lane chunk: packed symbol counts before it (in leaf)
	uint64_t base[NOL][6];                  // rope-cumulative symbol counts at the start of old leaf li
	uint64_t cutC[TL + 1][6];               // rope-cumulative counts of OLD symbols before each new leaf boundary
	uint32_t insC[TL][6];                   // inserted symbols per new leaf
	uint16_t lstart[NOL][64];               // per lane chunk: symbols before it (in leaf)
	uint16_t nb[NOL + 1], boff[NOL + 1], ltot[NOL];
	uint16_t ev_x[64];                      // event: old position relative to the first loaded leaf
	uint16_t ev_gk[64];                     // event: index of the old byte it falls into (li*LEAF+k) or GK_TAIL
	uint8_t  ev_t[64], ev_d[64];            // event: type (0..5 insert of that symbol, 8+c cut c), offset inside the run
	uint16_t dl_gk[64], dl_S[64], dl_npx[65];   // dirty (re-cut) bytes: byte index, stream index, exclusive prefix of piece counts
	uint8_t  pc[64][PCMAX];                 // their replacement bytes
	uint8_t  dl_np[64];
	uint16_t lbeg[TL + 1];                  // output stream index where new leaf l starts
	uint8_t  cut_u[TL + 1], cut_pi[TL + 1]; // per cut: owner index in dl_*, piece index where the new leaf starts
	int      nd, fallback;
};

union MergeLds { DenseLds d; SparseLds s; };

struct MergeTile {
	int b;
	uint64_t j, segs, o0, i0, i1, fl;
	uint32_t q0, q1, nvalid, dlen;
	int nl;
	bool have;
};

__device__ __forceinline__ uint32_t get_byte(const uint32_t w[4], int i) { return (w[i >> 2] >> ((i & 3) * 8)) & 0xffu; }

// packed per-symbol counters: symbols 0..4 in 12-bit fields; N (5) is derived from the position
__device__ __forceinline__ uint64_t pk_add(uint64_t acc, uint32_t sym, uint32_t len) { return sym < 5 ? acc + ((uint64_t)len << (12 * sym)) : acc; }
__device__ __forceinline__ uint32_t pk_get(uint64_t acc, int sym) { return (uint32_t)(acc >> (12 * sym)) & 0xfffu; }

// =============================================================================================
// dense path
// =============================================================================================

__device__ void merge_dense(DenseLds &L, const MergeTile &T, const RopeDesc &orp, const RopeDesc &nrp,
		PoolView oldp, PoolView newp, const uint64_t *INS_E, const uint8_t *INS_A, uint64_t *RK)
{
	const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
	if (tid < MT / 32) L.flag[tid] = 0;
	if (tid < 6) {
		uint64_t v;
		if (T.have) {
			const uint64_t gl = orp.leaf0 + T.fl;
			v = oldp.sbcum[gl / SB].v[tid] - oldp.sbcum[orp.sb0].v[tid] + oldp.meta[gl].c[tid];
		} else v = orp.cnt[tid];
		L.base[tid] = v;
	}
	// ---- decode the old leaves into one symbol per byte (rle_dec1, rle.h:39-51; 1-byte runs only)
	for (int li = w; li < T.nl; li += 4) {
		const uint64_t gl = orp.leaf0 + T.fl + li;
		const int nb = oldp.meta[gl].nbytes;
		const uint4 *src = (const uint4*)(oldp.data + gl * (uint64_t)LEAF);
		uint32_t wd[4] = {0, 0, 0, 0};
		const int nv = min(16, max(0, nb - ln * 16));
		if (nv > 0) { const uint4 v = src[ln]; wd[0] = v.x; wd[1] = v.y; wd[2] = v.z; wd[3] = v.w; }
		uint32_t mysum = 0;
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < nv) mysum += get_byte(wd, i) >> 3;
		const uint32_t start = wave_incl_add(mysum) - mysum;
		uint8_t *dst = L.old + li * LEAF + start;
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			if (i >= nv) break;
			const uint32_t byte = get_byte(wd, i), len = byte >> 3, s = byte & 7;
			for (uint32_t x = 0; x < len; ++x) *dst++ = (uint8_t)s;
		}
	}
	__syncthreads();
	// ---- per-32-symbol prefix counts over the decoded region (for the ranks of the inserts)
	{
		uint64_t lo = 0, hi = 0;
		const uint32_t cb = tid * 32;
		if (cb < T.dlen) {
			const uint32_t ce = min(cb + 32u, T.dlen);
			for (uint32_t y = cb; y < ce; ++y) {
				const uint32_t s = L.old[y];
				if (s < 4) lo += 1ull << (16 * s); else hi += 1ull << (16 * (s - 4));
			}
		}
		L.cplo[tid] = block_excl_add<uint64_t>(lo, L.w64, (uint64_t*)0);
		L.cphi[tid] = block_excl_add<uint64_t>(hi, L.w64, (uint64_t*)0);
	}
	__syncthreads();
	// ---- inserts: rank on the old rope (return value of rope_insert_run, rope.c:147) and placement
	for (uint32_t q = T.q0 + tid; q < T.q1; q += 256) {
		const uint64_t e = INS_E[T.segs + q];
		const uint32_t a = INS_A[T.segs + q];
		const uint32_t x = (uint32_t)(e - T.fl * LEAF), c = x >> 5;
		uint32_t cnt = (uint32_t)(((a < 4 ? L.cplo[c] >> (16 * a) : L.cphi[c] >> (16 * (a - 4)))) & 0xffffu);
		for (uint32_t y = c * 32; y < x; ++y) cnt += (L.old[y] == a);
		RK[T.segs + q] = L.base[a] + cnt;
		const uint32_t p = (uint32_t)(e + q - T.o0);
		L.out[p] = (uint8_t)a;
		atomicOr(&L.flag[p >> 5], 1u << (p & 31));
	}
	__syncthreads();
	// ---- assemble my 16 output symbols
	const uint32_t nvalid = T.nvalid;
	const uint32_t p0 = tid * 16;
	const int myvalid = (int)min(16u, nvalid > p0 ? nvalid - p0 : 0u);
	const uint32_t flags = (L.flag[tid >> 1] >> ((tid & 1) * 16)) & 0xffffu & ((1u << myvalid) - 1u);
	const uint32_t nonins = myvalid - __popc(flags);
	uint32_t oldoff = (uint32_t)(T.i0 - T.fl * LEAF) + block_excl_add<uint32_t>(nonins, L.w32, (uint32_t*)0);
	uint32_t sy[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		uint32_t v = 0xff;
		if (i < myvalid) v = (flags >> i & 1) ? L.out[p0 + i] : L.old[oldoff++];
		sy[i] = v;
	}
	__syncthreads();
	{
		uint4 v;
		v.x = sy[0] | sy[1] << 8 | sy[2] << 16 | sy[3] << 24;
		v.y = sy[4] | sy[5] << 8 | sy[6] << 16 | sy[7] << 24;
		v.z = sy[8] | sy[9] << 8 | sy[10] << 16 | sy[11] << 24;
		v.w = sy[12] | sy[13] << 8 | sy[14] << 16 | sy[15] << 24;
		*(uint4*)(L.out + p0) = v;
	}
	__syncthreads();
	// ---- re-encode: one wave per output leaf (16 symbols per lane x 64 lanes = LEAF)
	const uint32_t prev0 = ln == 0 ? 0xffu : L.out[p0 - 1];
	const int lp0 = ln * 16;
	const int lv = (int)min((uint32_t)LEAF, nvalid > (uint32_t)(w * LEAF) ? nvalid - w * LEAF : 0u);
	int lastnat = -1;
	{
		uint32_t pv = prev0;
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < myvalid) { if (sy[i] != pv) lastnat = lp0 + i; pv = sy[i]; }
	}
	const int incmax = wave_incl_max(lastnat);
	int rs = __shfl_up(incmax, 1);
	if (ln == 0) rs = 0;
	int lh = ln == 0 ? 0 : rs + (lp0 - 1 - rs) / 15 * 15;
	int hc = 0;
	{
		uint32_t pv = prev0; int r = rs;
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < myvalid) {
			const int p = lp0 + i;
			const bool nat = sy[i] != pv;
			if (nat) r = p;
			hc += (nat || (p - r) % 15 == 0);
			pv = sy[i];
		}
	}
	const int hinc = wave_incl_add(hc);
	const int hb = hinc - hc;
	const int nbytes = __shfl(hinc, 63);
	{
		uint32_t pv = prev0; int r = rs, seen = 0;
		uint8_t *ob = L.bytes + w * LEAF;
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < myvalid) {
			const int p = lp0 + i;
			const bool nat = sy[i] != pv;
			if (nat) r = p;
			if (nat || (p - r) % 15 == 0) {
				if (p != 0) ob[hb + seen - 1] = (uint8_t)((p - lh) << 3 | pv);
				lh = p; ++seen;
			}
			pv = sy[i];
		}
		if (myvalid > 0 && lp0 + myvalid == lv) ob[nbytes - 1] = (uint8_t)((lv - lh) << 3 | pv);
	}
	uint64_t clo = 0; uint32_t chi = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) if (i < myvalid) {
		const uint32_t s = sy[i];
		if (s < 4) clo += 1ull << (16 * s); else chi += 1u << (16 * (s - 4));
	}
	clo = wave_sum(clo); chi = wave_sum(chi);
	__syncthreads();
	if (lv > 0) {
		const uint64_t gl = nrp.leaf0 + T.j * TL + w;
		if (ln == 0) {
			LeafMeta m;
			m.c[0] = (uint16_t)clo; m.c[1] = (uint16_t)(clo >> 16); m.c[2] = (uint16_t)(clo >> 32); m.c[3] = (uint16_t)(clo >> 48);
			m.c[4] = (uint16_t)chi; m.c[5] = (uint16_t)(chi >> 16);
			m.nbytes = (uint16_t)nbytes; m.pad = 0;
			newp.meta[gl] = m;
		}
		if (ln * 16 < nbytes) ((uint4*)(newp.data + gl * (uint64_t)LEAF))[ln] = ((const uint4*)(L.bytes + w * LEAF))[ln];
	}
}

// =============================================================================================
// sparse path
// =============================================================================================

// tiny run writer used when one old run is re-cut (the moral equivalent of rle.c:73-86)
struct PieceOut {
	uint8_t *dst; int n, cap; int cs, cl; bool on; bool ovf;
	__device__ void flush() {
		while (cl > 0) {
			const int t = cl > 15 ? 15 : cl;
			if (on) { if (n < cap) dst[n] = (uint8_t)(t << 3 | cs); else ovf = true; ++n; }
			cl -= t;
		}
	}
	__device__ void add(int s, int len) { if (len <= 0) return; if (s == cs) cl += len; else { flush(); cs = s; cl = len; } }
};

__device__ bool merge_sparse(SparseLds &L, const MergeTile &T, const RopeDesc &orp, const RopeDesc &nrp,
		PoolView oldp, PoolView newp, const uint64_t *INS_E, const uint8_t *INS_A, uint64_t *RK)
{
	const int tid = threadIdx.x, ln = lane_id(), w = wave_id();
	const int nl = T.nl;
	const int ni = (int)(T.q1 - T.q0);
	const int nlo = (int)((T.nvalid + LEAF - 1) / LEAF);       // new leaves produced by this tile
	const int ne = ni + nlo + 1;
	const uint32_t x0 = (uint32_t)(T.i0 - T.fl * LEAF);
	if (tid == 0) { L.fallback = 0; L.nd = 0; }
	if (tid < NOL * 6) {
		const int li = tid / 6, s = tid % 6;
		uint64_t v = 0;
		if (li < nl) {
			const uint64_t gl = orp.leaf0 + T.fl + li;
			v = oldp.sbcum[gl / SB].v[s] - oldp.sbcum[orp.sb0].v[s] + oldp.meta[gl].c[s];
		}
		L.base[li][s] = v;
	}
	// ---- step 0: load the old leaves; per 16-byte chunk: symbols before it, packed counts before it
	for (int li = w; li < nl; li += 4) {
		const uint64_t gl = orp.leaf0 + T.fl + li;
		const int nb = oldp.meta[gl].nbytes;
		const uint4 *src = (const uint4*)(oldp.data + gl * (uint64_t)LEAF);
		uint32_t wd[4] = {0, 0, 0, 0};
		const int nv = min(16, max(0, nb - ln * 16));
		if (nv > 0) { const uint4 v = src[ln]; wd[0] = v.x; wd[1] = v.y; wd[2] = v.z; wd[3] = v.w; }
		((uint4*)(L.raw + li * LEAF))[ln] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
		uint32_t sum = 0; uint64_t pk = 0;
#pragma unroll
		for (int i = 0; i < 16; ++i) if (i < nv) { const uint32_t bt = get_byte(wd, i); sum += bt >> 3; pk = pk_add(pk, bt & 7, bt >> 3); }
		const uint32_t inc = wave_incl_add(sum);
		const uint64_t pinc = wave_incl_add(pk);
		L.lstart[li][ln] = (uint16_t)(inc - sum);
		L.lcnt[li][ln] = pinc - pk;
		if (ln == 63) { L.ltot[li] = (uint16_t)inc; L.nb[li] = (uint16_t)nb; }
	}
	__syncthreads();
	// ---- step 1..3 on wave 0: one lane per event (inserts + new leaf boundaries), sorted by output position
	uint64_t my_rk = 0; bool my_is_ins = false; uint32_t my_q = 0;
	if (w == 0) {
		if (ln == 0) { uint16_t o = 0; for (int li = 0; li < nl; ++li) { L.boff[li] = o; o += L.nb[li]; } L.boff[nl] = o; }
		// inserts -> event slots
		uint32_t f = 0xffffffffu, xe = 0, a = 0;
		if (ln < ni) {
			const uint64_t e = INS_E[T.segs + T.q0 + ln];
			a = INS_A[T.segs + T.q0 + ln];
			f = (uint32_t)(e + T.q0 + ln - T.o0);
			xe = (uint32_t)(e - T.fl * LEAF);
			const int idx = ln + min((int)(f / LEAF) + 1, nlo);
			L.ev_x[idx] = (uint16_t)xe; L.ev_t[idx] = (uint8_t)a;
		}
		int cut_idx[TL + 1];
#pragma unroll
		for (int c = 0; c <= TL; ++c) {
			cut_idx[c] = -1;
			if (c > nlo) continue;
			const uint32_t P = c < nlo ? (uint32_t)c * LEAF : T.nvalid;
			const int before = __popcll(__ballot(ln < ni && f < P));      // inserts in front of this boundary
			cut_idx[c] = c + before;
			if (ln == 0) { L.ev_x[c + before] = (uint16_t)(x0 + P - before); L.ev_t[c + before] = (uint8_t)(8 + c); }
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
		// my event
		const bool act = ln < ne;
		const uint32_t ex = act ? L.ev_x[ln] : 0; const uint32_t et = act ? L.ev_t[ln] : 0;
		uint16_t gk = GK_TAIL; uint32_t d = 0;
		uint64_t C[6] = {0, 0, 0, 0, 0, 0};
		if (act) {
			const int li = ex / LEAF; const uint32_t xin = ex % LEAF;
			if (li < nl && xin < L.ltot[li]) {
				int lo = 0, hi = 63;                                      // last lane chunk starting at or before xin
				while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (L.lstart[li][mid] <= xin) lo = mid; else hi = mid - 1; }
				uint32_t pos = L.lstart[li][lo]; uint64_t pk = L.lcnt[li][lo];
				const uint8_t *bp = L.raw + li * LEAF + lo * 16;
				int i = 0; uint32_t bt = bp[0];
				while (pos + (bt >> 3) <= xin) { pk = pk_add(pk, bt & 7, bt >> 3); pos += bt >> 3; ++i; bt = bp[i]; }
				d = xin - pos; gk = (uint16_t)(li * LEAF + lo * 16 + i);
				pk = pk_add(pk, bt & 7, d);
				uint32_t sum5 = 0;
#pragma unroll
				for (int s = 0; s < 5; ++s) { const uint32_t v = pk_get(pk, s); C[s] = L.base[li][s] + v; sum5 += v; }
				C[5] = L.base[li][5] + (xin - sum5);
			} else if (nl > 0) {                                          // at the end of the loaded old symbols ("tail")
				const int ll = nl - 1;                                    // counts = everything up to the end of the last loaded leaf
				uint64_t pk = L.lcnt[ll][63]; uint32_t pos = L.lstart[ll][63];
				const uint8_t *bp = L.raw + ll * LEAF + 63 * 16;
				const int nvl = min(16, max(0, (int)L.nb[ll] - 63 * 16));
				for (int i = 0; i < nvl; ++i) { const uint32_t bt = bp[i]; pk = pk_add(pk, bt & 7, bt >> 3); pos += bt >> 3; }
				uint32_t sum5 = 0;
#pragma unroll
				for (int s = 0; s < 5; ++s) { const uint32_t v = pk_get(pk, s); C[s] = L.base[ll][s] + v; sum5 += v; }
				C[5] = L.base[ll][5] + (pos - sum5);
			}
			L.ev_gk[ln] = gk; L.ev_d[ln] = (uint8_t)d;
			my_is_ins = et < 8;
		}
		// which insert am I?  event index -> insert index: subtract the cuts in front of me
		if (act && et < 8) {
			int cuts_before = 0;
#pragma unroll
			for (int c = 0; c <= TL; ++c) cuts_before += (cut_idx[c] >= 0 && cut_idx[c] < ln);
			my_q = T.q0 + (uint32_t)(ln - cuts_before);
			uint64_t r = 0;
#pragma unroll
			for (int s = 0; s < 6; ++s) if ((int)et == s) r = C[s];
			my_rk = r;
		}
		if (act && et >= 8) {
#pragma unroll
			for (int s = 0; s < 6; ++s) L.cutC[et - 8][s] = C[s];
		}
		// inserted symbols per new leaf
		{
#pragma unroll
			for (int s = 0; s < 6; ++s) {
				const uint64_t ms = __ballot(act && (int)et == s);
#pragma unroll
				for (int l = 0; l < TL; ++l) {
					if (l >= nlo) continue;
					const uint64_t lo_m = lt_mask(cut_idx[l]) | (1ull << cut_idx[l]);       // events <= cut l
					const uint64_t hi_m = lt_mask(cut_idx[l + 1]);                           // events <  cut l+1
					if (ln == 0) L.insC[l][s] = __popcll(ms & hi_m & ~lo_m);
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
		// owners: first event on each old byte re-cuts that byte for all events on it
		const uint16_t gk_prev = __shfl_up((int)gk, 1);
		const bool owner = act && (ln == 0 || gk_prev != gk);
		const uint64_t om = __ballot(owner);
		const int u = __popcll(om & lt_mask(ln));
		int np = 0;
		if (owner) {
			uint32_t len = 0, s = 7; uint16_t S;
			if (gk != GK_TAIL) { const uint32_t bt = L.raw[gk]; len = bt >> 3; s = bt & 7; S = (uint16_t)(L.boff[gk / LEAF] + gk % LEAF); }
			else S = L.boff[nl];
			PieceOut po; po.dst = L.pc[u]; po.n = 0; po.cap = PCMAX; po.cs = -1; po.cl = 0; po.on = (u != 0); po.ovf = false;
			uint32_t pd = 0;
			for (int t = ln; t < ne && L.ev_gk[t] == gk; ++t) {
				const uint32_t dd = L.ev_d[t], ty = L.ev_t[t];
				po.add((int)s, (int)(dd - pd)); pd = dd;
				if (ty >= 8) {                                             // new leaf boundary
					po.flush(); po.cs = -1;
					const int c = ty - 8;
					if (c == 0) { po.n = 0; po.on = true; }                // everything before belongs to the previous tile
					L.cut_u[c] = (uint8_t)u; L.cut_pi[c] = (uint8_t)po.n;
					if (c == nlo) po.on = false;                           // everything after belongs to the next tile
				} else po.add((int)ty, 1);
			}
			po.add((int)s, (int)(len - pd));
			po.flush();
			np = po.n;
			if (po.ovf || np > PCMAX) L.fallback = 1;
			L.dl_gk[u] = gk; L.dl_S[u] = S; L.dl_np[u] = (uint8_t)min(np, PCMAX);
		}
		{
			const int inc = wave_incl_add(owner ? np : 0);
			if (owner) L.dl_npx[u] = (uint16_t)(inc - np);
			const int nd = __popcll(om);
			if (ln == 63) { L.dl_npx[nd] = (uint16_t)inc; L.nd = nd; }
		}
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
		if (ln <= nlo) {
			const int uu = L.cut_u[ln];
			L.lbeg[ln] = (uint16_t)(L.dl_S[uu] - L.dl_S[0] - uu + L.dl_npx[uu] + L.cut_pi[ln]);
		}
	}
	__syncthreads();
	if (L.fallback) return false;
	if (my_is_ins) RK[T.segs + my_q] = my_rk;
	// ---- step 4: copy.  Clean bytes move as they are; their destination only depends on how many
	// re-cut bytes precede them.
	const int nd = L.nd;
	const int S0 = L.dl_S[0];
	const int lb1 = nlo > 1 ? L.lbeg[1] : 0x7fff, lb2 = nlo > 2 ? L.lbeg[2] : 0x7fff, lb3 = nlo > 3 ? L.lbeg[3] : 0x7fff;
	auto put = [&](int D, uint8_t v) {
		const int l = (D >= lb1) + (D >= lb2) + (D >= lb3);
		L.outb[l * LEAF + D - L.lbeg[l]] = v;
	};
	for (int ch = tid; ch < nl * 64; ch += 256) {
		const int li = ch >> 6, c16 = ch & 63;
		const int nv = min(16, max(0, (int)L.nb[li] - c16 * 16));
		if (nv <= 0) continue;
		const int gk0 = li * LEAF + c16 * 16;
		int lo = 0, hi = nd;                                              // t = #dirty bytes with gk < gk0
		while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int)L.dl_gk[mid] < gk0) lo = mid + 1; else hi = mid; }
		int t = lo;
		const int Sb = L.boff[li] + c16 * 16;
		const bool inside_dirty = t < nd && (int)L.dl_gk[t] < gk0 + nv;
		if (!inside_dirty) {
			if (t == 0 || t == nd) continue;                              // before the first / after the last boundary
			const int D0 = Sb - S0 - t + L.dl_npx[t];
			const int l = (D0 >= lb1) + (D0 >= lb2) + (D0 >= lb3);
			uint8_t *dst = L.outb + l * LEAF + D0 - L.lbeg[l];
			const uint8_t *sp = L.raw + gk0;
			for (int i = 0; i < nv; ++i) dst[i] = sp[i];
		} else {
			for (int i = 0; i < nv; ++i) {
				const int g = gk0 + i;
				while (t < nd && (int)L.dl_gk[t] < g) ++t;
				if (t < nd && (int)L.dl_gk[t] == g) continue;             // re-cut byte: written by its owner
				if (t == 0 || t == nd) continue;
				put(Sb + i - S0 - t + L.dl_npx[t], L.raw[g]);
			}
		}
	}
	for (int uu = tid; uu < nd; uu += 256) {
		const int D0 = L.dl_S[uu] - S0 - uu + L.dl_npx[uu];
		const int np = L.dl_np[uu];
		for (int i = 0; i < np; ++i) put(D0 + i, L.pc[uu][i]);
	}
	__syncthreads();
	// ---- step 5: store the new leaves + their metadata
	if (w < nlo) {
		const uint64_t gl = nrp.leaf0 + T.j * TL + w;
		const int nbytes = L.lbeg[w + 1] - L.lbeg[w];
		if (ln == 0) {
			LeafMeta m;
#pragma unroll
			for (int s = 0; s < 6; ++s) m.c[s] = (uint16_t)(L.cutC[w + 1][s] - L.cutC[w][s] + L.insC[w][s]);
			m.nbytes = (uint16_t)nbytes; m.pad = 0;
			newp.meta[gl] = m;
		}
		if (ln * 16 < nbytes) ((uint4*)(newp.data + gl * (uint64_t)LEAF))[ln] = ((const uint4*)(L.outb + w * LEAF))[ln];
	}
	return true;
}

// =============================================================================================
// kernel
// =============================================================================================

__global__ __launch_bounds__(256) void k_merge(const Ctl *ctl, int side, int force_dense, PoolView oldp, PoolView newp,
		const uint64_t *INS_E, const uint8_t *INS_A, uint64_t *RK, const uint32_t *TQ, unsigned long long *stats)
{
	__shared__ __align__(16) MergeLds lds;
	const uint64_t tile = blockIdx.x;
	if (tile >= ctl->mt0[6]) return;
	MergeTile T;
	int b = 0;
	while (tile >= ctl->mt0[b+1]) ++b;
	T.b = b;
	T.j = tile - ctl->mt0[b];
	const RopeDesc &orp = ctl->rope[side][b], &nrp = ctl->rope[side ^ 1][b];
	T.segs = ctl->seg[side].start[b];
	T.q0 = TQ[tile + b]; T.q1 = TQ[tile + b + 1];
	T.o0 = T.j * MT;
	const uint64_t o1 = min(T.o0 + (uint64_t)MT, nrp.n);
	T.nvalid = (uint32_t)(o1 - T.o0);
	T.i0 = T.o0 - T.q0; T.i1 = o1 - T.q1;                      // old symbols [i0,i1) belong to this tile
	T.fl = T.i0 / LEAF;                                        // first old leaf touched
	T.have = T.fl < orp.nleaves;
	T.dlen = T.have ? (uint32_t)(T.i1 - T.fl * LEAF) : 0u;     // decoded region = [fl*LEAF, i1)
	T.nl = (T.dlen + LEAF - 1) / LEAF;
	bool done = false;
	if (!force_dense && T.have && T.nl > 0 && (T.q1 - T.q0) <= (uint32_t)NI_SPARSE) {
		// the sparse path also wants the leaf that holds old symbol i1 (an insert or the closing
		// boundary may sit exactly at a leaf start): take one more leaf when i1 is leaf aligned
		MergeTile S = T;
		if (T.i1 % LEAF == 0 && T.fl + T.nl < orp.nleaves && T.nl < NOL) S.nl = T.nl + 1;
		done = merge_sparse(lds.s, S, orp, nrp, oldp, newp, INS_E, INS_A, RK);
		if (!done) __syncthreads();
		if (stats && threadIdx.x == 0) atomicAdd(&stats[done ? 1 : 2], 1ull);
	} else if (stats && threadIdx.x == 0) atomicAdd(&stats[0], 1ull);
	if (!done) merge_dense(lds.d, T, orp, nrp, oldp, newp, INS_E, INS_A, RK);
}

} // namespace rb2
