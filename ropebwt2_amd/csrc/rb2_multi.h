// rb2_multi.h -- N engines behind one handle: the sharded build driven from INSIDE the library (include/rb2_hip.h,
// "N GPUs behind one handle").  Included at the end of rb2_engine.hip (one translation unit: it drives the engine's phases
// batch_begin / round_counts / round_merge / batch_end directly).
//
// What the reference does (mrope.c:287-296, 312-340): mr_insert_multi hands the buckets of a round to its worker threads,
// waits for them, reads every rope's counts, moves the strings to their next buckets -- all inside the call.  Here a "worker"
// is a GPU (or a virtual rank on one), the buckets are the 31 sub-ropes dealt out by the owner map, and the two places where
// the reference's master touches all ropes become the two exchange steps of a round:
//
//   counts     every rank's rows of the 31 x 6 matrix "members of bucket r that insert a"  ->  their sum on every rank
//              PEER: k_mround reads the peers' rows (peer access), RCCL: ncclAllReduce in place
//   layout     k_mround (one block, on every rank, from the summed matrix and the owner map): where this rank's k_advance
//              writes the records it sends (Ctl::sdest) and the list of pieces k_munpack fetches (MTab) -- the host never
//              needs the matrix for that
//   records    24-byte ShardRec {l, size, id, symbol cursor} per surviving string, from the owner of piece (b,x) to the owner of (a,b)
//              PEER: the receiver's k_munpack reads them from the senders' buffers, RCCL: grouped ncclSend / ncclRecv
//
// Ordering between ranks on the PEER transport is by device events (hipStreamWaitEvent), two per round and rank; the host
// threads only meet at a spin barrier so that nobody waits on an event that has not been recorded yet.  No host <-> device
// synchronisation inside a batch.  Buffers a peer reads are never rewritten before that peer is done with them:
//   gcnt (local rows)  written by round_counts(r), read by the peers' k_mround(r); rewritten by round_counts(r + 1), which
//                      is queued behind this rank's k_munpack(r), which waited for every peer's evB(r), recorded behind
//                      that peer's k_mround(r);
//   send[r & 1]        written by k_advance(r), read by the peers' k_munpack(r); rewritten by k_advance(r + 2), queued
//                      behind this rank's k_munpack(r + 1), which waited for every peer's evB(r + 1), recorded behind that
//                      peer's k_munpack(r).
#pragma once
#include <atomic>
#include <thread>
#include <dlfcn.h>
// librccl is dlopen'ed on first use of the RCCL transport; its header only supplies the handful of types and enums the calls need.
// A ROCm install without the RCCL development files still builds the library: the same declarations, spelled out (they are ABI).
#if !defined(RB2_NO_RCCL_HEADER) && defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define RB2_HAVE_RCCL_HEADER 1
#endif
#endif
#ifndef RB2_HAVE_RCCL_HEADER
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
#endif

namespace rb2 {

struct MPiece { uint64_t vsrc, dst, cnt; const ShardRec *src; };   // cnt records at src[0..) (virtual receive index vsrc..) go to the next arrays at dst
constexpr int MPIECES = NR * 6;
struct MTab { uint64_t total; uint32_t npieces, pad; MPiece pc[MPIECES]; };
struct MPtrs { const void *p[RB2_MULTI_MAX_RANKS]; };
struct MOwner { uint8_t o[32]; };

// The exchange plan of a round, on the device, from the global count matrix g and the owner map.  An ENTRY is (r, a):
// the members of bucket r that insert a (a = 1..5; strings that insert $ retire, mrope.c:310); it travels from
// s = owner[r] to d = owner[(a, rope_sym(r))].  One thread per entry; every quantity is an exclusive sum over the entries
// in some order:
//   off   place in s's send buffer: entries of the same source in (d, r2, r) order            (== shard_layout())
//   vsrc  place in d's receive order: entries of the same destination in (s, r2, r) order      (the RCCL receive buffer)
//   dst   place in d's next string arrays: entries of the same destination in (r2, r) order    (== k_setup's seg.start + dest)
struct MPlan { uint64_t off, vsrc, dst, cnt; uint32_t pidx; int s, d, r, a; };
// entry t = (r, a) of the plan; g = the global count matrix, own = the owner map.  Host and device run the same code
// (rb2_hip_multi_plan_host lets the CPU tests check it against a simulation of the exchange).
__host__ __device__ inline MPlan mplan_entry(const uint64_t *g, const uint8_t *own, int t)
{
	MPlan P;
	const int r = t / 5, a = 1 + t % 5;
	const int r2 = rope_of(a, rope_sym(r)), s = own[r], d = own[r2];
	uint64_t off = 0, vsrc = 0, dst = 0; uint32_t pidx = 0;
	for (int t2 = 0; t2 < NR * 5; ++t2) {
		const int rr = t2 / 5, aa = 1 + t2 % 5;
		const int rr2 = rope_of(aa, rope_sym(rr)), ss = own[rr], dd = own[rr2];
		const uint64_t v = g[rr * 6 + aa];
		const bool lt = rr2 < r2 || (rr2 == r2 && rr < r);
		if (ss == s && (dd < d || (dd == d && lt))) off += v;
		if (dd == d) {
			if (ss < s || (ss == s && lt)) { vsrc += v; pidx += v != 0; }
			if (lt) dst += v;
		}
	}
	P.off = off; P.vsrc = vsrc; P.dst = dst; P.cnt = g[r * 6 + a]; P.pidx = pidx; P.s = s; P.d = d; P.r = r; P.a = a;
	return P;
}

// One single-block launch per round and rank for everything that sits between the counting phase and the merge phase of a sharded
// round: the sum of the count rows over the peers (PEER), the exchange plan (mplan_entry) and k_setup of the round -- three
// launches in a row on every rank's stream otherwise, and launches are what N rank threads of one process queue up behind each other
// for (profiles/r04d_vranks8_one_gpu.txt: 8 ranks on one device keep it busy 40 % of the time).
// npeer == 0: g_inout already holds the sum (RCCL: ncclAllReduce in place).
template <bool SPARSE> __global__ __launch_bounds__(256) void k_mround(Ctl *ctl, MPtrs rows, int npeer, uint64_t *g_inout, MOwner ow, int me, int peer, MPtrs srcs, const ShardRec *recv, MTab *tab,
		int side, int par, uint32_t round, volatile unsigned long long *hmax, volatile unsigned long long *hne)
{
	__shared__ uint64_t g[NR * 6];
	__shared__ unsigned long long s_tot;
	__shared__ uint32_t s_np;
	for (int i = threadIdx.x; i < NR * 6; i += 256) {
		uint64_t v = 0;
		if (npeer) { for (int p = 0; p < npeer; ++p) v += ((const uint64_t*)rows.p[p])[i]; g_inout[i] = v; }   // (a void in-place round is redone from the summed matrix: k_setup<false>)
		else v = g_inout[i];
		g[i] = v;
	}
	if (threadIdx.x == 0) { s_tot = 0; s_np = 0; }
	if (threadIdx.x == 255) {                                   // does ANY rank hold a string with a non-empty interval this round?  (word NR * 6 of the rows, GCN)
		uint64_t v = 0;
		if (npeer) { for (int p = 0; p < npeer; ++p) v += ((const uint64_t*)rows.p[p])[NR * 6]; g_inout[NR * 6] = v; }
		else v = g_inout[NR * 6];
		*hne = (unsigned long long)round << 32 | (v != 0 ? 1ull : 0ull);   // pinned host memory: the host reads it when it gets there, no synchronisation
	}
	__syncthreads();
	const int t = threadIdx.x;
	if (t < NR * 5) {
		const MPlan P = mplan_entry(g, ow.o, t);
		if (P.s == me) { ctl->sdest[P.r][P.a] = P.off; ctl->pdst[P.r][P.a] = P.dst; ctl->pdev[P.r][P.a] = (uint32_t)P.d; }   // RCCL: place in my send buffer; PEER: place in the owner's next arrays
		if (P.d == me && P.cnt) {
			MPiece p;
			p.vsrc = P.vsrc; p.dst = P.dst; p.cnt = P.cnt;
			p.src = peer ? (const ShardRec*)srcs.p[P.s] + P.off : recv + P.vsrc;
			tab->pc[P.pidx] = p;
			atomicAdd(&s_tot, (unsigned long long)P.cnt); atomicAdd(&s_np, 1u);
		}
	}
	__syncthreads();
	if (t == 0) { tab->total = s_tot; tab->npieces = s_np; tab->pad = 0; }
	if (t < 64) setup_body<SPARSE>(ctl, side, g, par, round, hmax, peer != 0);
}

// records -> next round's SoA arrays in bucket order, fetched from wherever k_mround says they are: the
// senders' buffers (PEER: loads over xGMI, 24 bytes per lane, consecutive per piece) or the local receive buffer (RCCL).
// The host does not know how many strings arrive: the grid covers about twice the rank's fair share, with a grid stride behind it.
__global__ __launch_bounds__(256) void k_munpack(const Ctl *ctl, const MTab *tab, const uint8_t *s, uint8_t *A2, uint32_t round,
		uint64_t *L2, uint64_t *U2, uint64_t *W2)
{
	const uint64_t total = tab->total;
	bool nonempty = false;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {   // grid stride: the host sizes the grid for about twice the rank's fair share
		int lo = 0, hi = (int)tab->npieces - 1;                  // last piece with vsrc <= i (pieces tile the receive order)
		while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab->pc[mid].vsrc <= i) lo = mid; else hi = mid - 1; }
		const MPiece &pc = tab->pc[lo];
		const ShardRec r = pc.src[i - pc.vsrc];
		const uint64_t d = pc.dst + (i - pc.vsrc);
		const uint64_t l = r.a & 0xffffffffffffull, size = r.a >> 48 | (r.b >> 32) << 16;
		L2[d] = l; U2[d] = l + size;
		W2[d] = r.w; A2[d] = (uint8_t)cur_sym(r.w);
		nonempty |= size != 0;
	}
	if (__any(nonempty) && lane_id() == 0) ((Ctl*)ctl)->ne[(round & 1) ^ 1] = 1;   // see Ctl::ne
}

} // namespace rb2

namespace {

// ---- librccl, loaded when the first RCCL handle is created (the one-GPU product path never maps its 570 MB) ----
struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

RcclApi &rccl()
{
	if (g_rccl.lib) return g_rccl;
	// a process that already has an RCCL mapped (torch brings its own) must use THAT one: two copies would each think they own the
	// devices' communication resources
	void *lib = nullptr;
	const char *cand[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
	if (dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) lib = dlopen(nullptr, RTLD_NOW);
	for (int i = 0; !lib && i < 4; ++i) lib = dlopen(cand[i], RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
	for (int i = 0; !lib && i < 4; ++i) lib = dlopen(cand[i], RTLD_NOW | RTLD_GLOBAL);
	if (!lib) { rb2_fatal("[rb2_hip] the RCCL transport was asked for but librccl cannot be loaded (%s)\n", dlerror()); }
	RcclApi &R = g_rccl;
	R.lib = lib;
#define RB2_NCCL_SYM(field, name) do { *(void**)&R.field = dlsym(lib, name); if (!R.field) { rb2_fatal("[rb2_hip] librccl has no %s\n", name); } } while (0)
	RB2_NCCL_SYM(GetUniqueId, "ncclGetUniqueId"); RB2_NCCL_SYM(CommInitRank, "ncclCommInitRank"); RB2_NCCL_SYM(CommInitAll, "ncclCommInitAll");
	RB2_NCCL_SYM(CommDestroy, "ncclCommDestroy"); RB2_NCCL_SYM(AllReduce, "ncclAllReduce"); RB2_NCCL_SYM(Send, "ncclSend"); RB2_NCCL_SYM(Recv, "ncclRecv");
	RB2_NCCL_SYM(GroupStart, "ncclGroupStart"); RB2_NCCL_SYM(GroupEnd, "ncclGroupEnd"); RB2_NCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RB2_NCCL_SYM
	return R;
}
#define NCCLCHK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
	rb2_fatal("[rb2_hip] %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, rccl().GetErrorString(r_)); } } while (0)

// the host threads of the local ranks meet here (never the devices): sense-reversing, spins briefly, then yields
struct SpinBarrier {
	std::atomic<int> count{0}, gen{0};
	int n = 1;
	RankFail *fail = nullptr;                                   // the owning handle's failure record
	void wait()
	{
		if (n <= 1) return;
		const int g = gen.load(std::memory_order_acquire);
		if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == n) { count.store(0, std::memory_order_relaxed); gen.fetch_add(1, std::memory_order_release); return; }
		for (int spins = 0; gen.load(std::memory_order_acquire) == g; ++spins) {
			if (fail && fail->failed.load(std::memory_order_relaxed)) throw RankAbort();   // a rank of THIS handle gave up (rb2_fatal on its thread): nobody waits for it
			if (spins > 4000) std::this_thread::yield();
		}
	}
};

// f(local rank) on a thread per local rank; a fatal error on one of them is reported here, on the calling thread (rb2_fatal)
template <class F> void rank_threads(RankFail *rf, int n, F f)
{
	std::vector<std::thread> th;
	for (int k = 0; k < n; ++k) th.emplace_back([&f, k, rf]() { t_rank_fail = rf; try { f(k); } catch (const RankAbort &) {} });
	for (auto &t : th) t.join();
	if (rf->failed.load()) {                                    // (cleared first: the handler may leave by longjmp)
		char msg[1024];
		memcpy(msg, rf->msg, sizeof(msg)); msg[sizeof(msg) - 1] = 0;
		rf->failed.store(0);
		rb2_fatal("%s", msg);
	}
}

struct MRank {
	rb2_hip_t *h = nullptr;
	int dev = 0, grank = 0;
	ShardRec *send[2] = {nullptr, nullptr}; size_t send_cap = 0;
	ShardRec *recv = nullptr; size_t recv_cap = 0;
	uint64_t *gred = nullptr;               // PEER: the summed matrix (h->gcnt keeps this rank's rows for the peers to read)
	uint64_t *gloc = nullptr;               // = h->gcnt as created
	MTab *tab = nullptr;
	PushTab *push[2] = {nullptr, nullptr};  // PEER: every rank's string arrays of either side as this rank's device sees them (k_advance stores into them)
	hipEvent_t evA = nullptr, evB = nullptr, evG = nullptr;
	uint64_t *pin_g = nullptr;              // RCCL: 2 x GCN pinned, the reduced matrix of the round (host sizes the sends from it)
	ncclComm_t comm = nullptr;
	DevBuf<uint8_t> text;                   // the batch on this rank's device (host-buffer entry point)
	const uint8_t *s_dev = nullptr;
	BatchState B;
};

} // namespace

// The batch text of a host-buffer insert, ONE copy for all local ranks: a reserved address range with the text in it, piece by piece in
// the memory of the ranks' devices (rank k's device holds the k-th share of the pieces) and mapped for every device of the handle -- each
// rank uploads its own share (1 / n of the batch over its own PCIe link), every kernel of every rank sees the same linear text, and
// what a rank reads outside its share (the cursor refills: 16 bytes per string every nine rounds; the batch set-up on the rank that holds
// rope $) travels over xGMI.  Used when the ranks sit on more than one physical device (RB2_MULTI_TEXT=shard forces it on one device:
// the tests; =copy switches it off); any failing call means one copy of the whole text per device, as before.
struct ShardedText {
	void *base = nullptr; size_t range = 0, piece = 0, mapped = 0;
	std::vector<hipMemGenericAllocationHandle_t> h;
	std::vector<int> owner;                                     // local rank that holds piece i
	std::vector<int64_t> bytes;                                 // per local rank: bytes of text memory it holds
	void release()
	{
		for (size_t i = 0; i < h.size(); ++i) { (void)hipMemUnmap((char*)base + i * piece, piece); (void)hipMemRelease(h[i]); }
		if (base) (void)hipMemAddressFree(base, range);
		h.clear(); owner.clear(); base = nullptr; range = mapped = piece = 0;
	}
};

struct rb2_hip_multi_s {
	RankFail fail;                          // a fatal error on one of this handle's rank threads (reported on the calling thread)
	ShardedText stext;
	int n = 0, world = 0, rank0 = 0, transport = 0, so = 0;
	int owner[NR];
	std::vector<MRank> rk;
	SpinBarrier bar;
	int64_t n_sync = 0, n_rounds = 0, n_batches = 0;
	int active = 1;                         // ranks that own one of the 16 heavy pieces
	int trace = 0;
	int rccl_self = 0;                      // RB2_RCCL_SELF=1: a rank's own block travels through ncclSend / ncclRecv too (exercises librccl on a one-GPU box)
};

namespace {

void multi_ensure_exchange(rb2_hip_multi_t *m, MRank &R, uint64_t records)
{
	if (m->transport == RB2_TRANSPORT_PEER) return;            // strings are stored straight into their next owner's arrays: no records, no staging
	if (records <= R.send_cap) return;
	const size_t cap = records + records / 8 + 1024;
	for (int i = 0; i < 2; ++i) { if (R.send[i]) HIPCHK(hipFree(R.send[i])); HIPCHK(hipMalloc((void**)&R.send[i], cap * sizeof(ShardRec))); }
	if (m->transport == RB2_TRANSPORT_RCCL) { if (R.recv) HIPCHK(hipFree(R.recv)); HIPCHK(hipMalloc((void**)&R.recv, cap * sizeof(ShardRec))); R.recv_cap = cap; }
	R.send_cap = cap;
}

// one batch on one local rank (its own host thread); every rank runs the same number of rounds
void multi_rank_batch(rb2_hip_multi_t *m, int k, int64_t len)
{
	MRank &R = m->rk[k];
	rb2_hip_t *h = R.h;
	HIPCHK(hipSetDevice(R.dev));
	hipStream_t st = h->st;
	const bool peer = m->transport == RB2_TRANSPORT_PEER;
	BatchState &B = R.B;
	B = BatchState();
	batch_begin(h, B, len, R.s_dev);
	// the layout of this rank's slice follows ITS regime: strings it expects per round against the leaves it holds (choose_layout);
	// the first rounds of a batch are hot spots by construction (see insert_dev)
	if (h->sparse && h->sp_backoff < h->sp_head) h->sp_backoff = h->sp_head;
	const uint64_t m_eff = std::max<uint64_t>(1, B.m / (uint64_t)std::max(1, m->active) + B.m / (uint64_t)(4 * std::max(1, m->active)));
	const int64_t sp0 = h->n_sparse_rounds + h->n_void;
	// exchange buffers: a rank never holds (or receives) more strings than the batch has.  They are (re)allocated between
	// batches only, and the peers learn the new addresses behind the barrier below.
	multi_ensure_exchange(m, R, B.m);
	m->bar.wait();
	if (peer) {                                                // (behind the barrier: every rank's arrays have their size for this batch)
		for (int c = 0; c < 2; ++c) {
			PushTab T;
			memset(&T, 0, sizeof(T));
			for (int p = 0; p < m->n; ++p) { rb2_hip_t *q = m->rk[p].h; T.L2[p] = q->L[c].p; T.U2[p] = q->U[c].p; T.W2[p] = q->W[c].p; T.A2[p] = q->A[c].p; T.ctl[p] = q->ctl; }
			HIPCHK(hipMemcpyAsync(R.push[c], &T, sizeof(T), hipMemcpyHostToDevice, st));
			h->push[c] = R.push[c];
		}
		HIPCHK(hipStreamSynchronize(st));                          // (T lives on this stack)
	}
	MOwner ow;
	memset(&ow, 0, sizeof(ow));
	for (int r = 0; r < NR; ++r) ow.o[r] = (uint8_t)m->owner[r];
	MPtrs rows, sends[2];
	memset(&rows, 0, sizeof(rows)); memset(sends, 0, sizeof(sends));
	if (peer) for (int p = 0; p < m->n; ++p) { rows.p[p] = m->rk[p].gloc; sends[0].p[p] = m->rk[p].send[0]; sends[1].p[p] = m->rk[p].send[1]; }
	const unsigned grid_m = cdiv(rank_share(h, B.m), 256);
	((volatile unsigned long long*)(h->h_flag + 8))[0] = ~0ull;   // nothing reported yet
	for (uint64_t r = 0; r <= B.max_len; ++r) {                 // one round per string position (mrope.c:299-342)
		// Once NO rank holds a string with a non-empty interval none ever will again: k_mround reports the sum of the ranks' flags of
		// every round to pinned memory; a report that has landed and reads zero lets this rank launch only the all-empty variants of
		// k_prep / k_advance from here on (before: both variants every round of every batch on a loaded index, one returning at once)
		if (!B.known_ae) {
			const unsigned long long v = ((volatile unsigned long long*)(h->h_flag + 8))[0];
			if (v != ~0ull && (v & 1ull) == 0 && (v >> 32) < r) B.known_ae = true;
		}
		h->gcnt = R.gloc;
		if (peer) HIPCHK(hipMemsetAsync(&h->ctl->ne[(r & 1) ^ 1], 0, 4, st));   // next round's flag: the peers' k_advance of THIS round set it (ordered behind this by evA)
		round_counts(h, B, r);
		// the layout of the round (a host decision; a change re-lays the slice out on this stream before anything of the round reads it)
		choose_layout(h, B, r, m_eff);
		if (peer) {
			HIPCHK(hipEventRecord(R.evA, st));
			m->bar.wait();                                        // every evA of this round is recorded
			for (int p = 0; p < m->n; ++p) if (p != k) HIPCHK(hipStreamWaitEvent(st, m->rk[p].evA, 0));
			h->gcnt = R.gred;
		} else {
			if (R.comm) NCCLCHK(rccl().AllReduce(h->gcnt, h->gcnt, NR * 6 + 1, ncclUint64, ncclSum, R.comm, st));
			HIPCHK(hipMemcpyAsync(R.pin_g + (r & 1) * GCN, h->gcnt, NR * 6 * 8, hipMemcpyDeviceToHost, st));
			HIPCHK(hipEventRecord(R.evG, st));
		}
		// sum of the count rows (PEER) + exchange plan + k_setup of the round: one launch (k_mround)
		{
			volatile unsigned long long *hmax = h->pos32 ? (volatile unsigned long long*)(h->d_flag + 4) : (volatile unsigned long long*)nullptr;
			volatile unsigned long long *hne = (volatile unsigned long long*)(h->d_flag + 8);   // (round, any non-empty interval on any rank) as k_mround last saw it
			if (h->sparse) hipLaunchKernelGGL(k_mround<true>, dim3(1), dim3(256), 0, st, h->ctl, rows, peer ? m->n : 0, h->gcnt, ow, R.grank, (int)peer, sends[r & 1], (const ShardRec*)R.recv, R.tab, h->side, (int)(r & 1), (uint32_t)r, hmax, hne);
			else hipLaunchKernelGGL(k_mround<false>, dim3(1), dim3(256), 0, st, h->ctl, rows, peer ? m->n : 0, h->gcnt, ow, R.grank, (int)peer, sends[r & 1], (const ShardRec*)R.recv, R.tab, h->side, (int)(r & 1), (uint32_t)r, hmax, hne);
			B.setup_round = r; B.setup_sparse = h->sparse; B.setup_epoch = h->layout_epoch;
		}
		// dense round: the slice is rewritten pool -> pool; in-place round: only the touched leaves, and the host reads a one-word
		// verdict before the exchange may go ahead (a void round is redone densely: its records do not exist yet)
		round_merge_any(h, B, r, peer ? (ShardRec*)nullptr : R.send[r & 1], false);   // flips side / cur: B.cur now names next round's arrays (PEER: k_advance stores into the owners' arrays, h->push)
		if (peer) {
			HIPCHK(hipEventRecord(R.evB, st));
			m->bar.wait();                                        // every evB of this round is recorded
			for (int p = 0; p < m->n; ++p) if (p != k) HIPCHK(hipStreamWaitEvent(st, m->rk[p].evB, 0));
		} else {
			// the reduced matrix reached pinned memory before the merge kernels of this round started: waiting for it does not
			// drain the stream -- the host sizes the sends while the GPU merges
			HIPCHK(hipEventSynchronize(R.evG));
			if (k == 0) ++m->n_sync;
			const int64_t *g = (const int64_t*)(R.pin_g + (r & 1) * GCN);
			int64_t off[NR][6], per[RB2_MULTI_MAX_RANKS], start[RB2_MULTI_MAX_RANKS];
			memset(off, 0, sizeof(off));
			shard_layout(m->owner, m->world, R.grank, g, off, per, start);
			int64_t rbase = 0;
			RcclApi &N = rccl();
			NCCLCHK(N.GroupStart());
			for (int p = 0; p < m->world; ++p) {
				int64_t off2[NR][6], per2[RB2_MULTI_MAX_RANKS], start2[RB2_MULTI_MAX_RANKS];
				memset(off2, 0, sizeof(off2));
				shard_layout(m->owner, m->world, p, g, off2, per2, start2);
				const int64_t nrecv = per2[R.grank], nsend = per[p];
				if (p == R.grank && !(m->rccl_self && R.comm)) {     // own block: a device copy, not a message (RB2_RCCL_SELF=1: through RCCL all the same)
					if (nsend) HIPCHK(hipMemcpyAsync(R.recv + rbase, R.send[r & 1] + start[p], (size_t)nsend * sizeof(ShardRec), hipMemcpyDeviceToDevice, st));
				} else {
					if (nsend) NCCLCHK(N.Send(R.send[r & 1] + start[p], (size_t)nsend * (sizeof(ShardRec) / 8), ncclUint64, p, R.comm, st));
					if (nrecv) NCCLCHK(N.Recv(R.recv + rbase, (size_t)nrecv * (sizeof(ShardRec) / 8), ncclUint64, p, R.comm, st));
				}
				rbase += nrecv;
			}
			NCCLCHK(N.GroupEnd());
		}
		if (!peer) {                                             // RCCL: the records that arrived -> next round's arrays
			const int cur = B.cur;
			hipLaunchKernelGGL(k_munpack, dim3(grid_m), dim3(256), 0, st, (const Ctl*)h->ctl, (const MTab*)R.tab, B.s, h->A[cur].p, (uint32_t)r,
					h->L[cur].p, h->U[cur].p, h->W[cur].p);
		}
		HIPCHK(hipGetLastError());
	}
	h->gcnt = R.gloc;
	batch_end(h);                                                // the one synchronisation of the batch (+ two in batch_begin)
	if (k == 0) { m->n_rounds += (int64_t)B.max_len + 1; ++m->n_batches; m->n_sync += h->n_sparse_rounds + h->n_void - sp0; }
	m->bar.wait();
}

void multi_run(rb2_hip_multi_t *m, int64_t len)
{
	if (m->n == 1) { multi_rank_batch(m, 0, len); return; }
	rank_threads(&m->fail, m->n, [&](int k) { multi_rank_batch(m, k, len); });
}

rb2_hip_multi_t *multi_new(int n, const int *devices, int world, int rank0, int so, int transport, const int *owner)
{
	if (n < 1 || world < n || world > RB2_MULTI_MAX_RANKS) { rb2_fatal("[rb2_hip] multi: bad number of ranks (%d local of %d, at most %d)\n", n, world, RB2_MULTI_MAX_RANKS); }
	if (transport != RB2_TRANSPORT_PEER && transport != RB2_TRANSPORT_RCCL) { rb2_fatal("[rb2_hip] multi: unknown transport %d\n", transport); }
	rb2_hip_multi_t *m = new rb2_hip_multi_s();
	m->n = n; m->world = world; m->rank0 = rank0; m->transport = transport; m->so = so;
	m->trace = getenv("RB2_HIP_TRACE") ? atoi(getenv("RB2_HIP_TRACE")) : 0;
	m->rccl_self = getenv("RB2_RCCL_SELF") ? atoi(getenv("RB2_RCCL_SELF")) : 0;
	if (owner) { for (int r = 0; r < NR; ++r) m->owner[r] = owner[r]; } else rb2_hip_default_owners(world, m->owner);
	for (int r = 0; r < NR; ++r) if (m->owner[r] < 0 || m->owner[r] >= world) { rb2_fatal("[rb2_hip] multi: bad owner of sub-rope %d\n", r); }
	{ bool seen[RB2_MULTI_MAX_RANKS] = {false}; m->active = 0; for (int r = 1; r < NR; ++r) if (!seen[m->owner[r]]) { seen[m->owner[r]] = true; ++m->active; } }
	m->bar.n = n; m->bar.fail = &m->fail;
	m->rk.resize(n);
	for (int k = 0; k < n; ++k) {
		MRank &R = m->rk[k];
		R.dev = devices[k]; R.grank = rank0 + k;
		R.h = rb2_hip_create(R.dev, so);
		engine_set_shard(R.h, R.grank, world, m->owner);
		HIPCHK(hipSetDevice(R.dev));
		R.gloc = R.h->gcnt;
		HIPCHK(hipMalloc((void**)&R.gred, GCN * 8));
		HIPCHK(hipMalloc((void**)&R.tab, sizeof(MTab)));
		HIPCHK(hipMemset(R.tab, 0, sizeof(MTab)));
		for (int c = 0; c < 2; ++c) HIPCHK(hipMalloc((void**)&R.push[c], sizeof(PushTab)));
		HIPCHK(hipEventCreateWithFlags(&R.evA, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&R.evB, hipEventDisableTiming));
		HIPCHK(hipEventCreateWithFlags(&R.evG, hipEventDisableTiming));
		HIPCHK(hipHostMalloc((void**)&R.pin_g, 2 * GCN * 8, hipHostMallocDefault));
	}
	if (transport == RB2_TRANSPORT_PEER) {
		// the receiver's kernels read the senders' memory: every pair of distinct devices needs peer access
		for (int k = 0; k < n; ++k) for (int p = 0; p < n; ++p) {
			if (m->rk[k].dev == m->rk[p].dev) continue;
			int can = 0;
			HIPCHK(hipDeviceCanAccessPeer(&can, m->rk[k].dev, m->rk[p].dev));
			if (!can) { rb2_fatal("[rb2_hip] multi: device %d cannot access device %d (no peer access): use the RCCL transport\n", m->rk[k].dev, m->rk[p].dev); }
			HIPCHK(hipSetDevice(m->rk[k].dev));
			const hipError_t e = hipDeviceEnablePeerAccess(m->rk[p].dev, 0);
			if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIPCHK(e);
			(void)hipGetLastError();
		}
	}
	return m;
}

// run f(local rank) on every local rank, each on a thread of its own
template <class F> void multi_each(rb2_hip_multi_t *m, F f)
{
	if (m->n == 1) { f(0); return; }
	rank_threads(&m->fail, m->n, f);
}

} // namespace

namespace {

// see ShardedText.  true: every rank's s_dev names the shared range and the text is in it.
bool multi_text_sharded(rb2_hip_multi_t *m, int64_t len, const uint8_t *s)
{
	ShardedText &T = m->stext;
	T.bytes.assign(m->n, 0);
	const char *e = getenv("RB2_MULTI_TEXT");
	if (e && !strcmp(e, "copy")) return false;
	int distinct = 0;
	std::vector<int> devs;
	for (int k = 0; k < m->n; ++k) { bool seen = false; for (int d : devs) seen |= d == m->rk[k].dev; if (!seen) { devs.push_back(m->rk[k].dev); ++distinct; } }
	if (m->n < 2 || m->transport != RB2_TRANSPORT_PEER || !vmm_enabled()) return false;
	if (distinct < 2 && !(e && !strcmp(e, "shard"))) return false;
	// pieces of one size, a power of two (mixed or odd sizes made hipMemSetAccess fail on this stack: DevBuf::vm_grow), about a rank's share
	const size_t need = (size_t)len + 64;
	size_t piece = 64ull << 20;
	while (piece < (2ull << 30) && piece * (size_t)m->n < need) piece <<= 1;
	const size_t npieces = (need + piece - 1) / piece;
	if (T.base && (T.piece != piece || npieces * piece > T.range)) { for (auto &R : m->rk) { HIPCHK(hipSetDevice(R.dev)); HIPCHK(hipStreamSynchronize(R.h->st)); } T.release(); }
	if (!T.base) {
		const size_t range = std::max<size_t>(npieces * piece, 4 * piece);
		void *base = nullptr;
		if (hipMemAddressReserve(&base, range, piece, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
		T.base = base; T.range = range; T.piece = piece;
	}
	std::vector<hipMemAccessDesc> acc(devs.size());
	for (size_t i = 0; i < devs.size(); ++i) { acc[i] = hipMemAccessDesc(); acc[i].location.type = hipMemLocationTypeDevice; acc[i].location.id = devs[i]; acc[i].flags = hipMemAccessFlagsProtReadWrite; }
	while (T.h.size() < npieces) {                              // (pieces are kept from batch to batch; a batch needs at most as many as the largest before it, or more)
		const size_t i = T.h.size();
		const int k = (int)(i * (size_t)m->n / npieces);          // contiguous shares: rank k holds pieces [k * np / n, (k + 1) * np / n)
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = m->rk[k].dev;
		hipMemGenericAllocationHandle_t hd;
		bool ok = hipMemCreate(&hd, piece, &prop, 0) == hipSuccess;
		if (ok && hipMemMap((char*)T.base + i * piece, piece, 0, hd, 0) != hipSuccess) { (void)hipMemRelease(hd); ok = false; }
		if (ok && hipMemSetAccess((char*)T.base + i * piece, piece, acc.data(), acc.size()) != hipSuccess) { (void)hipMemUnmap((char*)T.base + i * piece, piece); (void)hipMemRelease(hd); ok = false; }
		if (!ok) {
			fprintf(stderr, "[rb2_hip] multi: the batch text cannot be shared between the devices (%s): every device gets a copy of its own\n", hipGetErrorString(hipGetLastError()));
			(void)hipGetLastError();
			T.release();
			return false;
		}
		T.h.push_back(hd); T.owner.push_back(k);
	}
	// every rank uploads the pieces it holds, on its own stream; the kernels of the other ranks read them: all uploads are waited for
	multi_each(m, [&](int k) {
		MRank &R = m->rk[k];
		HIPCHK(hipSetDevice(R.dev));
		for (size_t i = 0; i < npieces; ++i) {
			if (T.owner[i] != k) continue;
			T.bytes[k] += (int64_t)piece;
			const size_t o = i * piece, nb = std::min(piece, (size_t)len > o ? (size_t)len - o : 0);
			if (nb) HIPCHK(hipMemcpyAsync((char*)T.base + o, s + o, nb, hipMemcpyHostToDevice, R.h->st));
		}
		HIPCHK(hipStreamSynchronize(R.h->st));
	});
	for (int k = 0; k < m->n; ++k) m->rk[k].s_dev = (const uint8_t*)T.base;
	return true;
}

} // namespace

extern "C" {

/* piece -> rank.  On DNA the 16 pieces (b,x), b,x in ACGT, carry ~1/16 of the rows each; they are dealt out in contiguous
 * blocks, so up to 16 ranks get load.  The light pieces ((b,$): one row per read; everything with N) ride with a neighbour,
 * rope $ with rank 0.  (tests/helpers.py restates it in Python.) */
void rb2_hip_default_owners(int nranks, int owner[])
{
	for (int r = 0; r < NR; ++r) owner[r] = 0;
	if (nranks <= 1) return;
	for (int b = 1; b < 6; ++b) for (int x = 0; x < 6; ++x) {
		const int k = (std::min(b, 4) - 1) * 4 + (std::min(std::max(x, 1), 4) - 1);
		owner[rope_of(b, x)] = std::min(k * nranks / 16, nranks - 1);
	}
}

/* Build a small job across the ranks of m and compare what every LOCAL rank holds with the same job on one engine: sub-rope by
 * sub-rope, device-side checksums of the symbols + the count matrix.  Two batches: the second one goes on top of an index that holds
 * reads, in whatever order the handle sorts -- rounds with non-empty intervals (both kernel variants, U and SIZE travelling with the
 * strings), the next-round flag set by the senders, copies of reads (groups of more than one string).  Collective over the ranks of the
 * group (every process of an RCCL group runs it at the same point).  soft = false: fatal on a mismatch -- a first run on hardware this
 * build has not seen either works or says where it does not; soft = true: says so and returns false (rb2_hip_multi_create then falls
 * back from the PEER transport to RCCL).  Leaves the handle empty. */
static bool multi_selftest(rb2_hip_multi_t *m, bool soft = false)
{
	const int n_reads = 2000, L = 48;
	char why[256] = "";
	rb2_hip_t *one = rb2_hip_create(m->rk[0].dev, m->so);
	uint64_t x = 0x9E3779B97F4A7C15ull;
	for (int batch = 0; batch < 2 && !why[0]; ++batch) {
		std::vector<uint8_t> s;
		s.reserve((size_t)n_reads * (L + 1));
		for (int i = 0; i < n_reads; ++i) {                      // reversed nt6 strings, 0-terminated (mrope.h:46-54); every 7th a copy of its neighbour
			const size_t at = s.size();
			for (int j = 0; j < L; ++j) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; s.push_back((uint8_t)(1 + (x >> 60) % 4)); }
			if (i % 7 == 6) memcpy(&s[at], &s[at - (L + 1)], L);
			s.push_back(0);
		}
		const int64_t len = (int64_t)s.size();
		rb2_hip_multi_insert_multi(m, len, s.data());
		rb2_hip_insert_multi(one, len, s.data());
		rb2_hip_wait(one);
		int64_t cm[36], c1[36];
		rb2_hip_multi_get_counts(m, cm); rb2_hip_get_counts(one, c1);
		if (memcmp(cm, c1, sizeof(cm)) != 0) snprintf(why, sizeof(why), "the count matrix of %d ranks differs from one engine's after batch %d", m->world, batch);
		for (int r = 0; r < NR && !why[0]; ++r) {
			const int o = m->owner[r] - m->rank0;
			if (o < 0 || o >= m->n) continue;                    /* another process holds (and checks) it */
			const uint64_t a = piece_hash(m->rk[o].h, r), b = piece_hash(one, r);
			if (a != b || m->rk[o].h->h_rope[r].n != one->h_rope[r].n)
				snprintf(why, sizeof(why), "sub-rope %d held by rank %d on device %d differs from the one-engine build after batch %d (%llu vs %llu symbols)", r, m->owner[r], m->rk[o].dev, batch,
						(unsigned long long)m->rk[o].h->h_rope[r].n, (unsigned long long)one->h_rope[r].n);
		}
	}
	rb2_hip_destroy(one);
	const char *tn = m->transport == RB2_TRANSPORT_PEER ? "PEER" : "RCCL";
	if (why[0]) {
		if (!soft) rb2_fatal("[rb2_hip] multi self-test (%s transport): %s\n", tn, why);
		fprintf(stderr, "[rb2_hip] multi self-test (%s transport): %s\n", tn, why);
		return false;
	}
	rb2_hip_multi_reset(m);
	m->n_sync = m->n_rounds = m->n_batches = 0;
	for (auto &R : m->rk) { R.h->n_sparse_rounds = R.h->n_void = R.h->n_relayout = R.h->n_respread = 0; }
	if (m->trace) fprintf(stderr, "[rb2_hip] multi self-test passed: %d ranks, %s transport\n", m->world, tn);
	return true;
}
static bool multi_want_selftest(int distinct_devices, int world)
{
	const char *e = getenv("RB2_MULTI_SELFTEST");
	if (e) return atoi(e) != 0;
	(void)world;
	return distinct_devices > 1;                                /* ranks on more than one physical device: hardware this build may never have seen */
}

int rb2_hip_multi_transport(const rb2_hip_multi_t *m) { return m->transport; }

rb2_hip_multi_t *rb2_hip_multi_create(int n, const int *devices, int sorting_order, int transport, const int *owner)
{
	int distinct = 0;
	for (int k = 0; k < n; ++k) { bool seen = false; for (int p = 0; p < k; ++p) seen |= devices[p] == devices[k]; distinct += !seen; }
	if (transport == RB2_TRANSPORT_PEER && distinct > 1) {      // every pair of distinct devices must see each other's memory
		bool ok = true; int a = -1, b = -1;
		for (int k = 0; k < n && ok; ++k) for (int p = 0; p < n && ok; ++p) {
			if (devices[k] == devices[p]) continue;
			int can = 0;
			if (hipDeviceCanAccessPeer(&can, devices[k], devices[p]) != hipSuccess || !can) { ok = false; a = devices[k]; b = devices[p]; (void)hipGetLastError(); }
		}
		if (!ok) {
			if (distinct < n) rb2_fatal("[rb2_hip] multi: device %d cannot access device %d (no peer access) and a device is listed more than once: neither transport can serve this list\n", a, b);
			fprintf(stderr, "[rb2_hip] multi: device %d cannot access device %d (no peer access): using the RCCL transport instead of PEER\n", a, b);
			transport = RB2_TRANSPORT_RCCL;
		}
	}
	auto make = [&](int tr) {
		rb2_hip_multi_t *m = multi_new(n, devices, n, 0, sorting_order, tr, owner);
		if (tr == RB2_TRANSPORT_RCCL && n > 1) {
			for (int k = 0; k < n; ++k) for (int p = 0; p < k; ++p)
				if (devices[k] == devices[p]) rb2_fatal("[rb2_hip] multi: RCCL needs one device per rank (device %d is listed twice); several ranks on one device run on the PEER transport\n", devices[k]);
			std::vector<ncclComm_t> comms(n);
			NCCLCHK(rccl().CommInitAll(comms.data(), n, devices));
			for (int k = 0; k < n; ++k) m->rk[k].comm = comms[k];
		} else if (tr == RB2_TRANSPORT_RCCL) {                    // a group of one: the same calls on a one-rank communicator
			ncclUniqueId id;
			HIPCHK(hipSetDevice(devices[0]));
			NCCLCHK(rccl().GetUniqueId(&id));
			NCCLCHK(rccl().CommInitRank(&m->rk[0].comm, 1, id, 0));
		}
		return m;
	};
	rb2_hip_multi_t *m = make(transport);
	if (multi_want_selftest(distinct, n)) {
		// PEER between PHYSICAL devices (strings stored straight into the owner's arrays over xGMI, ordered by events only) has never run on the
		// hardware this was built on -- every box had one GPU: if the job it builds there differs from one engine's, the records + RCCL
		// transport (the owner fetches, unpacks in a pass of its own: k_munpack) takes over, and has to pass the same test
		const bool fallback = transport == RB2_TRANSPORT_PEER && distinct == n && n > 1;
		if (!multi_selftest(m, fallback)) {
			fprintf(stderr, "[rb2_hip] multi: the PEER transport failed its start-up test on these devices: using the RCCL transport\n");
			rb2_hip_multi_destroy(m);
			m = make(RB2_TRANSPORT_RCCL);
			multi_selftest(m, false);
		}
	}
	return m;
}

void rb2_hip_multi_unique_id(void *id128)
{
	ncclUniqueId id;
	NCCLCHK(rccl().GetUniqueId(&id));
	static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
	memcpy(id128, &id, sizeof(id));
}

rb2_hip_multi_t *rb2_hip_multi_create_rank(int device, int rank, int nranks, const void *nccl_id, int sorting_order, const int *owner)
{
	if (rank < 0 || rank >= nranks) { rb2_fatal("[rb2_hip] multi: bad rank %d of %d\n", rank, nranks); }
	rb2_hip_multi_t *m = multi_new(1, &device, nranks, rank, sorting_order, RB2_TRANSPORT_RCCL, owner);
	ncclUniqueId id;
	if (nccl_id) memcpy(&id, nccl_id, sizeof(id));
	else if (nranks == 1) NCCLCHK(rccl().GetUniqueId(&id));
	else { rb2_fatal("[rb2_hip] multi: a group of %d ranks needs the id of rb2_hip_multi_unique_id() from rank 0\n", nranks); }
	HIPCHK(hipSetDevice(device));
	NCCLCHK(rccl().CommInitRank(&m->rk[0].comm, nranks, id, rank));
	if (multi_want_selftest(nranks, nranks)) multi_selftest(m);  /* collective: every process of the group is here */
	return m;
}

void rb2_hip_multi_destroy(rb2_hip_multi_t *m)
{
	if (!m) return;
	for (auto &R : m->rk) {
		HIPCHK(hipSetDevice(R.dev));
		HIPCHK(hipStreamSynchronize(R.h->st));
		if (R.comm) NCCLCHK(rccl().CommDestroy(R.comm));
		for (int i = 0; i < 2; ++i) if (R.send[i]) HIPCHK(hipFree(R.send[i]));
		if (R.recv) HIPCHK(hipFree(R.recv));
		HIPCHK(hipFree(R.gred)); HIPCHK(hipFree(R.tab)); for (int c = 0; c < 2; ++c) if (R.push[c]) HIPCHK(hipFree(R.push[c]));
		HIPCHK(hipEventDestroy(R.evA)); HIPCHK(hipEventDestroy(R.evB)); HIPCHK(hipEventDestroy(R.evG));
		HIPCHK(hipHostFree(R.pin_g));
		R.text.release();
		R.h->gcnt = R.gloc;
		rb2_hip_destroy(R.h);
	}
	m->stext.release();
	delete m;
}

int rb2_hip_multi_nranks(const rb2_hip_multi_t *m) { return m->world; }
int rb2_hip_multi_nlocal(const rb2_hip_multi_t *m) { return m->n; }
rb2_hip_t *rb2_hip_multi_engine(rb2_hip_multi_t *m, int k) { return (k >= 0 && k < m->n) ? m->rk[k].h : nullptr; }

void rb2_hip_multi_insert_multi_dev(rb2_hip_multi_t *m, int64_t len, const uint8_t *const *s_dev)
{
	if (len <= 0) { rb2_fatal("[rb2_hip] insert_multi: len must be > 0\n"); }   // mrope.c:268
	for (int k = 0; k < m->n; ++k) {
		if ((uintptr_t)s_dev[k] & 15) { rb2_fatal("[rb2_hip] multi insert: the device buffers must be 16-byte aligned\n"); }
		m->rk[k].s_dev = s_dev[k];
	}
	HIPCHK(hipSetDevice(m->rk[0].dev));
	check_last_byte(m->rk[0].h, len, s_dev[0]);
	multi_run(m, len);
}

void rb2_hip_multi_insert_multi(rb2_hip_multi_t *m, int64_t len, const uint8_t *s)
{
	if (len <= 0 || s[len - 1] != 0) { rb2_fatal("[rb2_hip] insert_multi: buffer must be non-empty and end with a sentinel\n"); }   // mrope.c:268
	if (multi_text_sharded(m, len, s)) { multi_run(m, len); return; }
	// one copy of the batch text per DEVICE (every rank reads all of it: the cursor refills every 10 rounds), uploaded by the first rank on it
	for (auto &b : m->stext.bytes) b = 0;
	std::vector<const uint8_t*> ptr(m->n, nullptr);
	multi_each(m, [&](int k) {
		MRank &R = m->rk[k];
		for (int p = 0; p < k; ++p) if (m->rk[p].dev == R.dev) return;
		HIPCHK(hipSetDevice(R.dev));
		R.text.ensure((size_t)len + 64);
		if ((size_t)k < m->stext.bytes.size()) m->stext.bytes[k] = (int64_t)R.text.cap;
		HIPCHK(hipMemcpyAsync(R.text.p, s, (size_t)len, hipMemcpyHostToDevice, R.h->st));
		HIPCHK(hipStreamSynchronize(R.h->st));                  // ranks on other streams read it
		ptr[k] = R.text.p;
	});
	for (int k = 0; k < m->n; ++k) if (!ptr[k]) for (int p = 0; p < k; ++p) if (m->rk[p].dev == m->rk[k].dev) { ptr[k] = ptr[p]; break; }
	for (int k = 0; k < m->n; ++k) m->rk[k].s_dev = ptr[k];
	multi_run(m, len);
}

void rb2_hip_multi_get_counts(rb2_hip_multi_t *m, int64_t c[36]) { rb2_hip_get_counts(m->rk[0].h, c); }   /* every rank tracks the counts of all pieces */

static int64_t multi_export(rb2_hip_multi_t *m, int b, uint8_t *dst, rb2_hip_run_cb cb, void *user)
{
	int64_t k = 0;
	for (int r = 0; r < NR; ++r) {
		if (rope_sym(r) != b) continue;
		const int o = m->owner[r] - m->rank0;
		if (o < 0 || o >= m->n) continue;                        /* another process holds it */
		rb2_hip_t *h = m->rk[o].h;
		HIPCHK(hipSetDevice(h->dev));
		ensure_dense(h);
		k += export_piece(h, r, dst ? dst + k : nullptr, cb, user);
	}
	return k;
}
int64_t rb2_hip_multi_rope_bytes(rb2_hip_multi_t *m, int b) { return multi_export(m, b, nullptr, nullptr, nullptr); }
int64_t rb2_hip_multi_download_rope(rb2_hip_multi_t *m, int b, uint8_t *dst) { return multi_export(m, b, dst, nullptr, nullptr); }
int64_t rb2_hip_multi_stream_rope(rb2_hip_multi_t *m, int b, rb2_hip_run_cb cb, void *user) { return multi_export(m, b, nullptr, cb, user); }

void rb2_hip_multi_load_ropes(rb2_hip_multi_t *m, const uint8_t *const rle[6], const int64_t n_bytes[6])
{
	multi_each(m, [&](int k) { rb2_hip_load_ropes(m->rk[k].h, rle, n_bytes); });   /* every rank decodes the stream, keeps its pieces, counts the others */
}

void rb2_hip_multi_reserve(rb2_hip_multi_t *m, int64_t batch_bytes, int64_t batch_strings, int64_t total_symbols)
{
	const int64_t share = total_symbols > 0 ? (int64_t)((double)total_symbols * 1.25 / std::max(1, m->active)) : 0;
	multi_each(m, [&](int k) { rb2_hip_reserve(m->rk[k].h, batch_bytes, batch_strings, share); });
}

void rb2_hip_multi_reset(rb2_hip_multi_t *m) { for (auto &R : m->rk) rb2_hip_reset(R.h); }
void rb2_hip_multi_sync(rb2_hip_multi_t *m) { for (auto &R : m->rk) rb2_hip_sync(R.h); }

/* rank of all six symbols in [0,x) of rope b: the pieces in front of the one that holds x are known by their counts (every
 * rank tracks them), the rest is a rank query inside one piece, answered by its owner */
void rb2_hip_multi_rank1a(rb2_hip_multi_t *m, int b, int64_t x, int64_t cx[6])
{
	for (int a = 0; a < 6; ++a) cx[a] = 0;
	if (x <= 0 || b < 0 || b > 5) return;
	const RopeDesc *hr = m->rk[0].h->h_rope;
	uint64_t p = (uint64_t)x;
	for (int r = 0; r < NR && p > 0; ++r) {
		if (rope_sym(r) != b) continue;
		uint64_t n = 0;
		for (int a = 0; a < 6; ++a) n += hr[r].cnt[a];
		if (p >= n) { for (int a = 0; a < 6; ++a) cx[a] += (int64_t)hr[r].cnt[a]; p -= n; continue; }
		const int o = m->owner[r] - m->rank0;
		if (o < 0 || o >= m->n) { rb2_fatal("[rb2_hip] multi rank: piece %d lives in another process\n", r); }
		int64_t c[6];
		rank_piece(m->rk[o].h, r, (int64_t)p, c);
		for (int a = 0; a < 6; ++a) cx[a] += c[a];
		p = 0;
	}
}

/* the same value rb2_hip_rope_hash gives for the same rope on one engine: every piece hashed by its owner */
uint64_t rb2_hip_multi_rope_hash(rb2_hip_multi_t *m, int b)
{
	uint64_t acc = 0;
	for (int r = 0; r < NR; ++r) {
		if (rope_sym(r) != b) continue;
		const int o = m->owner[r] - m->rank0;
		if (o < 0 || o >= m->n) { rb2_fatal("[rb2_hip] multi rope_hash: piece %d lives in another process\n", r); }
		acc = hash_mix(acc, piece_hash(m->rk[o].h, r), m->rk[o].h->h_rope[r].n);
	}
	return acc;
}

/* the exchange plan of one round as rank `me` sees it, computed on the HOST by the very function k_mround runs per entry
 * (mplan_entry): sdest[r*6+a] = where `me` writes the records of (r,a) in its send buffer (-1: not its entry); pieces[i] =
 * {source rank, offset in that rank's send buffer, offset in me's receive order, offset in me's next string arrays, records}
 * for the i-th piece `me` receives; returns their number, *total = records received.  No device needed (CPU tests). */
int rb2_hip_multi_plan_host(const int *owner, int nranks, const int64_t *g, int me, int64_t *sdest, int64_t (*pieces)[5], int64_t *total)
{
	uint8_t own[32];
	int np = 0;
	(void)nranks;
	for (int r = 0; r < NR; ++r) own[r] = (uint8_t)owner[r];
	for (int i = 0; i < NR * 6; ++i) sdest[i] = -1;
	*total = 0;
	for (int t = 0; t < NR * 5; ++t) {
		const MPlan P = mplan_entry((const uint64_t*)g, own, t);
		if (P.s == me) sdest[P.r * 6 + P.a] = (int64_t)P.off;
		if (P.d == me && P.cnt) {
			int64_t *q = pieces[P.pidx];
			q[0] = P.s; q[1] = (int64_t)P.off; q[2] = (int64_t)P.vsrc; q[3] = (int64_t)P.dst; q[4] = (int64_t)P.cnt;
			*total += (int64_t)P.cnt; ++np;
		}
	}
	return np;
}

int64_t rb2_hip_multi_text_bytes(const rb2_hip_multi_t *m, int k) { return (k >= 0 && (size_t)k < m->stext.bytes.size()) ? m->stext.bytes[k] : -1; }

void rb2_hip_multi_stats(rb2_hip_multi_t *m, int64_t out[6])
{
	out[0] = m->n_sync; out[1] = m->n_rounds; out[2] = m->n_batches; out[3] = out[4] = out[5] = 0;
	for (auto &R : m->rk) { out[3] += R.h->n_sparse_rounds; out[4] += R.h->n_void; out[5] += R.h->n_relayout; }
}

} // extern "C"
