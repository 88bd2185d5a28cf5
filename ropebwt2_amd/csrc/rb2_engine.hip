// rb2_engine.hip -- host orchestration + C ABI (include/rb2_hip.h) of the gfx950 BCR engine.
//
// Replaces the body of mr_insert_multi (/root/reference/mrope.c:258-345).  One process drives
// one GPU; everything is enqueued on a private HIP stream and the host synchronises twice per
// batch (string count, end of batch).  No CPU fallback: a missing device aborts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <vector>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <thread>
#include <condition_variable>
#include "rb2_hip.h"
#include "rb2_kernels.h"

using namespace rb2;

// Fatal errors.  The reference has no error convention on this path (asserts, unchecked mallocs: SURVEY.md 8b); the engine keeps that
// -- a message on stderr, then abort() -- but a host program can ask to be told first (rb2_hip_set_fatal_handler): it may log, clean up,
// or leave by longjmp / exit; if the handler returns, abort() follows.
// With N ranks behind one handle the failure may happen on a rank's own host thread (rb2_multi.h).  A handler that leaves by longjmp
// must not run there, and exit() there would run the atexit handlers while the other ranks still spin at their barrier: the first
// failure is recorded, the thread unwinds (RankAbort), the other rank threads leave at their next barrier, and the thread that called
// the API reports it -- message, handler, abort() -- once all of them are joined.
static rb2_hip_fatal_cb g_fatal_cb = nullptr;
static void *g_fatal_user = nullptr;
#include <cstdarg>
#include <atomic>
#include <mutex>
// (the failure record belongs to the HANDLE whose rank thread failed: the rank threads of another, healthy rb2_hip_multi_t in the same
// process neither see the flag at their barriers nor consume the message)
struct RankFail { std::atomic<int> failed{0}; std::mutex mu; char msg[1024]; };
static thread_local RankFail *t_rank_fail = nullptr;       // set on the host threads of a handle's ranks (rank_threads, rb2_multi.h)
struct RankAbort {};
[[noreturn]] static void rb2_fatal(const char *fmt, ...)
{
	char msg[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(msg, sizeof(msg), fmt, ap);
	va_end(ap);
	if (t_rank_fail) {
		{ std::lock_guard<std::mutex> lk(t_rank_fail->mu); if (!t_rank_fail->failed.load()) { memcpy(t_rank_fail->msg, msg, sizeof(msg)); t_rank_fail->failed.store(1); } }
		throw RankAbort();
	}
	fputs(msg, stderr);
	if (g_fatal_cb) g_fatal_cb(g_fatal_user, msg);
	abort();
}
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	rb2_fatal("[rb2_hip] %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__, hipGetErrorString(e_)); } } while (0)

namespace {

// Large, growing arrays (the leaf pools) live behind a reserved address range and grow by mapping more physical memory at their
// end (hipMemAddressReserve / hipMemCreate / hipMemMap): what is there stays where it is -- no second allocation, no device copy,
// no hipFree -- and a growth costs what the added memory costs.  (hipMalloc of tens of GB takes 25-40 ms per GB on some boxes of
// this pool: growing a 60 GB pool the classic way stalled full configs[3] twice for 1.6 s.)  RB2_NO_VMM=1, or any failing call,
// falls back to hipMalloc + copy.
static int g_vmm_ok = -1;
static size_t g_vmm_va = 0;
static bool vmm_enabled()
{
	if (g_vmm_ok < 0) {
		size_t fr = 0, tot = 0;
		g_vmm_ok = getenv("RB2_NO_VMM") ? 0 : 1;
		if (g_vmm_ok && hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0) g_vmm_va = (tot + (1ull << 30)) & ~((1ull << 30) - 1);   // no array outgrows the device
		else g_vmm_ok = 0;
	}
	return g_vmm_ok == 1;
}

// set while insert_dev queues the rounds of a batch: DevBuf::ensure counts the buffers that have to grow there.  Growing one means a
// hipFree (or new mappings) in the middle of the queued rounds -- the runtime waits for the whole device, the host thread with it
// (r05: the superblock totals grew round by round; a host-buffer job lost 8 % to it).  Everything a batch needs is sized in batch_begin.
static thread_local int64_t *t_grow_in_rounds = nullptr;
template <typename T> struct DevBuf {
	T *p = nullptr; size_t cap = 0;
	bool vm = false;                                            // set by the owner before first use: grow in place
	size_t vm_div = 1;                                         // the reserved range is 1 / vm_div of the device's memory (+ 1 GiB): small arrays need less address space
	size_t vm_bytes = 0;                                       // mapped so far
	bool vm_res = false;                                       // p is a reserved address range (not a hipMalloc pointer), whatever is mapped behind it
	size_t vm_res_bytes = 0;                                   // ... of this size (vm_range() at the time of the reservation, on the device it was made for)
	size_t vm_piece = 0;                                       // every mapping of this array has this size, a power of two between 64 MiB and 2 GiB chosen at the first
	                                                           // request (pieces of mixed or odd sizes made hipMemSetAccess fail with 'invalid argument' on this stack)
	std::vector<hipMemGenericAllocationHandle_t> vm_h; std::vector<size_t> vm_sz;
	size_t vm_range() const { return (g_vmm_va / vm_div + (4ull << 30)) & ~((2ull << 30) - 1); }   // a multiple of every piece size
	bool vm_grow(size_t n)                                     // false: nothing changed (the caller falls back)
	{
		const size_t CH = 64ull << 20;                         // mapping granule of ours (a multiple of the device's)
		if (!vmm_enabled() || (p && !vm_res)) return false;      // (a classic allocation stays classic)
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess) return false;
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
		if (!p) {
			void *base = nullptr;
			const hipError_t er = hipMemAddressReserve(&base, vm_range(), CH, nullptr, 0);
			if (er != hipSuccess) {
				if (getenv("RB2_HIP_TRACE")) fprintf(stderr, "[rb2_hip] hipMemAddressReserve(%.1f GB) failed (%s): pools grow by hipMalloc + copy\n", vm_range() / 1e9, hipGetErrorString(er));
				(void)hipGetLastError(); g_vmm_ok = 0; return false;
			}
			p = (T*)base; vm_res = true; vm_res_bytes = vm_range();
		}
		if (!vm_piece) { vm_piece = CH; while (vm_piece < (2ull << 30) && vm_piece * 8 < n * sizeof(T)) vm_piece <<= 1; }
		const size_t want = (n * sizeof(T) + vm_piece - 1) / vm_piece * vm_piece;
		if (want > vm_res_bytes) {                              // does not fit the range: back to a classic allocation (ensure() copies what is mapped, then releases the range)
			if (vm_bytes == 0) vm_release();
			return false;
		}
		hipMemAccessDesc acc = {};
		acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
		while (vm_bytes < want) {                               // one handle per piece
			const size_t add = vm_piece;
			hipMemGenericAllocationHandle_t h;
			bool ok = hipMemCreate(&h, add, &prop, 0) == hipSuccess;
			if (ok && (hipMemMap((char*)p + vm_bytes, add, 0, h, 0) != hipSuccess || hipMemSetAccess((char*)p + vm_bytes, add, &acc, 1) != hipSuccess)) {
				(void)hipMemUnmap((char*)p + vm_bytes, add); (void)hipMemRelease(h); ok = false;
			}
			if (!ok) {                                          // keep what is mapped (cap says how much); the caller falls back to a fresh allocation
				if (getenv("RB2_HIP_TRACE")) fprintf(stderr, "[rb2_hip] mapping %.2f GB behind %.2f GB failed (%s)\n", add / 1e9, vm_bytes / 1e9, hipGetErrorString(hipGetLastError()));
				(void)hipGetLastError();
				if (vm_bytes == 0) vm_release();
				return false;
			}
			vm_h.push_back(h); vm_sz.push_back(add);
			vm_bytes += add; cap = vm_bytes / sizeof(T);
		}
		return true;
	}
	void vm_release()
	{
		size_t off = 0;
		for (size_t i = 0; i < vm_h.size(); ++i) { (void)hipMemUnmap((char*)p + off, vm_sz[i]); (void)hipMemRelease(vm_h[i]); off += vm_sz[i]; }
		if (vm_res) (void)hipMemAddressFree(p, vm_res_bytes);
		vm_h.clear(); vm_sz.clear(); vm_bytes = 0; vm_res = false; vm_res_bytes = 0; p = nullptr; cap = 0;
	}
	void ensure(size_t n, bool keep = false, hipStream_t st = 0) {
		if (n <= cap) return;
		if (t_grow_in_rounds) ++*t_grow_in_rounds;              // (a buffer that grows while a batch's rounds are being queued: see rb2_hip_layout_stats out[6])
		if (vm && vm_grow(n)) return;
		size_t ncap = std::max(n, cap + cap / 2);
		T *q = nullptr;
		hipError_t e_ = hipMalloc((void**)&q, ncap * sizeof(T));
		if (e_ != hipSuccess && ncap > n) { ncap = n; (void)hipGetLastError(); e_ = hipMalloc((void**)&q, ncap * sizeof(T)); }   // no room for the growth margin: take what is needed
		if (e_ != hipSuccess) {                                    // tell the caller how much was needed (a 288 GB part can run out: both strands of 1.2 B reads)
			size_t fr = 0, tot = 0;
			(void)hipMemGetInfo(&fr, &tot);
			rb2_fatal("[rb2_hip] out of device memory: a buffer of %.2f GB was needed (it held %.2f GB before), %.2f of %.2f GB are free; "
					"use smaller batches (-m) or shard the index over more GPUs\n", ncap * sizeof(T) / 1e9, cap * sizeof(T) / 1e9, fr / 1e9, tot / 1e9);
		}
		if (keep && p && cap) { HIPCHK(hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, st)); HIPCHK(hipStreamSynchronize(st)); }
		if (p) { if (vm_res) vm_release(); else HIPCHK(hipFree(p)); }
		p = q; cap = ncap;
	}
	void release() { if (p) { if (vm_res) vm_release(); else HIPCHK(hipFree(p)); } p = nullptr; cap = 0; }
};

struct Pool {
	DevBuf<uint8_t> data, xh; DevBuf<LeafMeta> meta, own; DevBuf<SbRec> sbrec; DevBuf<SbBase> sbbase;
	uint64_t cap_leaves = 0;
	void ensure(uint64_t leaves, bool keep, hipStream_t st) {
		if (leaves <= cap_leaves) return;
		data.vm = meta.vm = own.vm = sbrec.vm = true;   // grow in place (DevBuf::vm_grow); the chunk bases are a few thousand records
		meta.vm_div = own.vm_div = 16; sbrec.vm_div = 256;   // (16 of 384 bytes per leaf, 32 of 12288 per superblock)
		uint64_t nl = std::max<uint64_t>(leaves, cap_leaves + cap_leaves / 4);
		nl = (nl + SB - 1) / SB * SB;
		data.ensure(nl * LEAFB, keep, st); meta.ensure(nl, keep, st); own.ensure(nl, keep, st);
		sbrec.ensure(nl / SB + 1, keep, st); sbbase.ensure(nl / SB / SCHUNK + 2, keep, st); xh.ensure(nl / WPL + 1024, keep, st);
		cap_leaves = nl;
	}
	PoolView view() const { return PoolView{data.p, meta.p, sbrec.p, own.p, sbbase.p, xh.p}; }
	void release() { data.release(); meta.release(); own.release(); sbrec.release(); sbbase.release(); xh.release(); cap_leaves = 0; }
};

struct ProfRec { int k; hipEvent_t a, b; int64_t units; int round; };
constexpr uint64_t POS32_LIMIT = (1ull << 32) - (1ull << 24);   // narrow position storage only while every position stays below this

} // namespace

struct rb2_hip_s {
	int dev = 0, so = 0;
	hipStream_t st = 0;
	Pool pool[2];
	int side = 0;                       // descriptor parity: ctl->rope[side] / ctl->seg[side] describe the current BWT (flips every round)
	int pside = 0;                      // pool that physically holds it (flips with `side` in dense rounds, stays in sparse ones)
	bool sparse = false;                // the pool is in the sparse layout (leaves with slack, in-place rounds)
	double sp_lambda = 0.6;             // go sparse when (strings per round) / (leaves of the index) falls below this
	int sp_backoff = 0, sp_penalty = 0; // after a void sparse round: dense rounds to run before trying again / its growth
	int sp_head = 8;                    // dense rounds at the start of a batch while the index is sparse (see insert_dev)
	int sp_maxpen = 6;                  // at most 64 dense rounds between two attempts (a failed attempt costs about four dense rounds)
	int leaf_pipe = 8192;               // in-place rounds use the software-pipelined k_merge_leaf_pipe with at most this many workgroups (RB2_LEAF_PIPE; 0: one wave per
	                                    // four work orders, k_merge_leaf).  1 M inserts per round: 232 us with k_merge_leaf, 259 / 228 / 212 / 207 / 212 us with 1024 / 1536 / 4096 / 8192 / 16384
	uint32_t *h_flag = nullptr;         // pinned: verdict of a sparse round (written by k_split through d_flag); [16..16+2*NE_RING): ring of ctl->ne snapshots, one per round
	uint32_t *d_flag = nullptr;         // ... its device address
	uint64_t layout_epoch = 0;          // counts the re-layouts: what k_setup derived from the piece descriptors before one is stale after it
	static constexpr int NE_RING = 32;
	hipEvent_t ev_flag = nullptr;
	// in-place rounds: the prefix over the superblock totals (k_sbscan*) is needed by the NEXT round's descent only (k_advance takes its ranks
	// from before the merge: RKOLD), so it rides in blocks of their own of the launches that follow the merge anyway (k_advance; k_sym / k_split)
	int dir_ride = 1;                   // RB2_DIR_RIDE=0: two launches of its own behind k_merge_leaf, as in rounds 2-5
	int ts_blocks = TSB;                // RB2_TS_BLOCKS=1: the single-launch counting tail on one block, as in rounds 4-5 (A/B)
	// in-place rounds of one engine are queued without waiting for their verdict (a void round is sticky on the device: rb2_kernels.h k_part_sparse);
	// spec_rounds = in-place rounds queued since the host last looked with the stream drained (verdict_check)
	int lazy_verdict = 1;               // RB2_LAZY_VERDICT=0: an event and a wait per in-place round, as in rounds 2-5
	uint64_t spec_rounds = 0;
	int run_ahead = 3;                  // ... and at most this many in-place rounds ahead of what the device reports done (h_flag[2]): a re-spread asked for by round r takes place this
	                                    // many rounds late at worst (RB2_RUN_AHEAD)
	unsigned long long *pair_d = nullptr, *pair_h = nullptr;   // k_pair_hist: what the batch just uploaded adds to the count matrix (device, pinned host)
	bool pair_valid = false;
	bool pending_end = false;           // rb2_hip_insert_multi returned with the batch queued but not awaited (finish_pending); lazy_insert: it may
	double tl_base = 0, tl_last = 0;
	int64_t n_grow_in_rounds = 0;       // device buffers that had to grow while the rounds of a dense batch were being queued (should stay 0)
	int timeline = 0;                   // RB2_HIP_TIMELINE=1: host timestamps of the phases of a host-buffer insert on stderr (changes nothing else)
	int lazy_insert = 1;                // RB2_HIP_LAZY_INSERT=0 / rb2_hip_set_lazy(h, 0): every insert returns only when the device is done
	int64_t n_relayout = 0, n_void = 0, n_sparse_rounds = 0;
	Ctl *ctl = nullptr;                 // device
	RopeDesc h_rope[NR];                // host mirror of ctl->rope[side] (sub-ropes)
	// per-string state
	DevBuf<uint64_t> L[2], U[2], W[2], START, SIZE, INS_E, zblk;   // L, U, SIZE, INS_E: sized for 64-bit values, read and written as 32-bit ones while pos32
	bool pos32 = false, want_pos32 = false;   // positions in 32-bit storage this batch (single-engine inserts only; maybe_widen)
	int pos_mode = getenv("RB2_POS") ? atoi(getenv("RB2_POS")) : 0;   // 0 auto, 64: never narrow; RB2_POS_WIDEN_AT=r: leave the narrow mode before round r (tests)
	uint64_t pos_m0 = 0;                // largest piece when the batch began
	DevBuf<uint16_t> RKREL;
	DevBuf<uint64_t> RKOLD;             // sparse rounds: rank of every new symbol on the rope as it was, in front of its leaf (k_part_sparse -> k_advance; stored in the positions' width)
	DevBuf<uint32_t> SPL;               // sparse rounds: leaves to split at the end of the round (k_part_sparse -> k_split)
	uint32_t split_epoch = 0;           // claim word of k_split: one value per in-place round, never repeated (rb2_kernels.h)
	bool want_respread = false;         // k_split met a superblock without a free slot: re-spread (sparse -> sparse) before the next round
	int64_t n_respread = 0;
	DevBuf<uint64_t> qbuf;              // rank queries and their answers
	uint64_t sp_nsb = 0;                // superblocks of the sparse pool (upper bound)
	DevBuf<LeafDesc> LD;
	DevBuf<uint8_t> A[2], INS_A, sbuf;   // A: symbol (+ flags) of every string this round; the other side receives next round's from k_advance
	// rb2_hip_prefetch: the NEXT batch travels to the device (second text buffer, copy stream) while the current one is inserted
	DevBuf<uint8_t> sbuf2; hipStream_t st_copy = nullptr; const uint8_t *pf_host = nullptr; size_t pf_done = 0; std::mutex pf_mu;
	bool pf_busy = false; std::condition_variable pf_cv;
	// pageable host memory -> device through pinned staging buffers filled by several host threads (par_upload)
	static constexpr int UP_T = 6, UP_NB = 2; static constexpr size_t UP_CH = 8u << 20;
	uint8_t *up_pin[UP_T][UP_NB] = {}; hipStream_t up_st[UP_T] = {}; hipEvent_t up_ev[UP_T][UP_NB] = {}; bool up_init = false;        // a prefetch copy is running (outside the lock: an insert of ANOTHER buffer must not wait for it)
	DevBuf<TileRec> trec; DevBuf<TileScan> tsc; DevBuf<TileFix> tfix; DevBuf<ChunkPart> cpart;
	DevBuf<SbTot> sbtot;
	uint64_t *d_tmp = nullptr;          // small scratch (8 x u64)
	// profiling
	int prof = 0;
	std::vector<ProfRec> recs;
	std::vector<hipEvent_t> evpool;
	int64_t p_launch[RB2_K_COUNT]; double p_ms[RB2_K_COUNT]; int64_t p_units[RB2_K_COUNT];
	int debug = 0;
	int cur_round = -1;
	uint64_t *gcnt = nullptr;           // device: NR x 6 count matrix of the current round
	bool own_stream = true;
	int rank = 0, nranks = 1; int owner[NR] = {0};
	PushTab *push[2] = {nullptr, nullptr};   // PEER transport of a sharded index: device tables of every rank's next arrays, [side of the arrays being written] (rb2_multi.h); else null
	int nactive = 1;                    // ranks that own a sub-rope other than rope $ (sizes the grids of a sharded rank: tile_grid)
	DevBuf<uint8_t> xstage, xpack; DevBuf<uint16_t> xnb; DevBuf<uint64_t> xoff;   // k_export staging, packed bytes, chunk offsets
	uint8_t *xhost[2] = {nullptr, nullptr}; uint64_t *xtot[2] = {nullptr, nullptr}; hipEvent_t xev[2] = {nullptr, nullptr};   // pinned double buffer
	int compact_ok = 1;                 // RB2_COMPACT=0: dense rounds always write plain (three-plane) windows
	int compact_stats = 0;              // RB2_COMPACT_STATS=1: k_merge counts the windows it writes per format (rb2_hip_window_stats)
	int64_t n_compact_rounds = 0;       // dense rounds that were allowed to write compact windows
	int64_t n_plain_handover = 0;       // dense rounds that wrote plain windows over compact ones because a re-layout was pending (plain_next): rb2_hip_layout_stats out[7]
	bool pool_compact = false;          // the last dense round wrote compact windows (only k_merge may read the pool now)
	bool plain_next = false;            // choose_layout wants to leave the dense layout: this round writes plain windows
	int ts_fold = SCHUNK;               // up to this many chunks of string tiles k_tscan3 scans the chunk totals itself (no k_tscan2 launch); RB2_TS_FOLD lowers it (tests)
	int ts_max = TS_MAX;                // batches with fewer string tiles run their counting tail in one single-block launch (k_tscan_setup); RB2_TS_MAX lowers it (tests)
	int trace = 0;                      // RB2_HIP_TRACE=1: per-round kernel times + merge path statistics on stderr
};

namespace {

hipEvent_t get_event(rb2_hip_t *h)
{
	if (!h->evpool.empty()) { hipEvent_t e = h->evpool.back(); h->evpool.pop_back(); return e; }
	hipEvent_t e; HIPCHK(hipEventCreate(&e)); return e;
}

struct Scope {          // times everything enqueued between construction and destruction
	rb2_hip_t *h; int k; ProfRec r; bool on;
	Scope(rb2_hip_t *h_, int k_, int64_t units) : h(h_), k(k_) {
		on = h->prof == 1 || (h->prof == 2 && k == RB2_K_MERGE);   // (2: the merge launches only -- two events per round instead of sixteen; rb2_hip_profile)
		if (!on) return;
		r.k = k; r.units = units; r.round = h->cur_round; r.a = get_event(h); r.b = get_event(h);
		HIPCHK(hipEventRecord(r.a, h->st));
	}
	~Scope() {
		if (h->debug) { hipError_t e = hipStreamSynchronize(h->st); if (e != hipSuccess) { rb2_fatal("[rb2_hip] kernel group %s failed: %s\n", rb2_hip_kernel_name(k), hipGetErrorString(e)); } }
		if (!on) return;
		HIPCHK(hipEventRecord(r.b, h->st));
		h->recs.push_back(r);
	}
};

void drain_profile(rb2_hip_t *h)
{
	for (auto &r : h->recs) {
		float ms = 0;
		HIPCHK(hipEventElapsedTime(&ms, r.a, r.b));
		h->p_launch[r.k] += 1; h->p_ms[r.k] += ms; h->p_units[r.k] += r.units;
		if (h->trace && r.round >= 0 && (r.round < 20 || r.round % 10 == 0)) fprintf(stderr, "[rb2_hip] round %3d %-10s %8.3f ms\n", r.round, rb2_hip_kernel_name(r.k), ms);
		h->evpool.push_back(r.a); h->evpool.push_back(r.b);
	}
	h->recs.clear();
}

inline unsigned cdiv(uint64_t a, uint64_t b) { return (unsigned)((a + b - 1) / b); }
inline unsigned grid8(unsigned g) { return (g + 8u * XCD_RUN - 1u) & ~(8u * XCD_RUN - 1u); }   // grids of the tile / window kernels: a multiple of the XCD count (xcd_item(), rb2_device.h); the extra blocks find nothing to do

// Grids of a rank of a sharded index.  The host knows the batch (m strings, len symbols), not the rank's share of it this round --
// asking would cost a synchronisation per round.  The upper bound "all of it" makes every launch N times too large: at N = 8
// (weak scaling, 327 M strings) 640 k workgroups per tile kernel of which 80 k find a tile -- empty workgroups alone then cost
// more than a millisecond per round.  The kernels walk their work with a grid stride, so the grid is a hint: twice the fair
// share (+ slack); a rank that holds more than that loops.  One GPU: the exact upper bound, one tile per block as before.
// launch KERNEL<..., STRIDE> with STRIDE = (this handle is a rank of a sharded index)
#define RB2_LAUNCH_STRIDE(h, KT, KF, ...) do { if ((h)->nranks > 1) hipLaunchKernelGGL(KT, __VA_ARGS__); else hipLaunchKernelGGL(KF, __VA_ARGS__); } while (0)
inline uint64_t rank_share(const rb2_hip_t *h, uint64_t whole)
{
	if (h->nranks <= 1) return whole;
	return std::min<uint64_t>(whole, 2 * whole / (uint64_t)std::max(1, h->nactive) + 64);
}

// recompute the rank directory (meta prefixes + superblock prefix) of pool side `sd`
void build_directory(rb2_hip_t *h, int sd /* descriptors */, int pool /* arrays */, uint64_t nsb_ub, bool sparse = false, bool leaves_done = false, uint64_t nsb_grid = 0)
{
	if (nsb_ub == 0) return;
	const hipStream_t st_ = h->st;
	if (nsb_grid == 0 || nsb_grid > nsb_ub) nsb_grid = nsb_ub;    // k_meta_sb walks the superblocks with a grid stride (rank_share); the scans need the true bound
	const unsigned nchunk = cdiv(nsb_ub, SCHUNK);
	h->sbtot.ensure(nsb_ub);
	PoolView pv = h->pool[pool].view();
	// leaves_done: an in-place round -- k_merge_leaf updated the entries of the leaves it rewrote and the superblock totals
	// itself (h->sbtot lives on between rounds); what is left is the prefix over the totals
	if (!leaves_done) RB2_LAUNCH_STRIDE(h, k_meta_sb<true>, k_meta_sb<false>, dim3(cdiv(nsb_grid, 8)), dim3(256), 0, st_, h->ctl, sd, pv, h->sbtot.p, (int)sparse);
	// in-chunk prefixes (SbRec) + chunk totals in one pass over the totals, then the chunk bases (the in-chunk prefixes need no base: queries add it)
	hipLaunchKernelGGL(k_sbscan3, dim3(nchunk), dim3(SCHUNK / SBT), 0, st_, (const Ctl*)h->ctl, (const SbTot*)h->sbtot.p, pv);
	hipLaunchKernelGGL(k_sbscan2, dim3(1), dim3(SB2T), 0, st_, (const Ctl*)h->ctl, pv.sbbase);
}

void fetch_ropes(rb2_hip_t *h)
{
	HIPCHK(hipMemcpyAsync(h->h_rope, &h->ctl->rope[h->side][0], sizeof(RopeDesc) * NR, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
}

// ---- one batch, in phases (the single-GPU path runs them back to back; the rope-sharded path
// ---- interleaves them with the exchanges of the round, rb2_multi.h) --------------------------------

struct BatchState {
	const uint8_t *s = nullptr; uint64_t len = 0, m = 0, max_len = 0, n_tot = 0, nsb_ub = 0;
	unsigned nst_ub = 0, nsc = 0;
	int cur = 0;                            // string array side
	bool known_ae = false;                  // the host can tell that every interval of the batch is empty: input order, or an empty index
	bool known_ae0 = false;                 // ... from the start of the batch (what a rollback falls back to: insert_dev)
	uint64_t counted = (uint64_t)-1;        // round whose counting phase (round_counts) is already queued
	uint64_t setup_round = (uint64_t)-1;    // round whose k_setup ran inside its counting phase (k_tscan_setup) ...
	bool setup_sparse = false; uint64_t setup_epoch = 0;   // ... for this layout, at this layout epoch (a re-layout in between: k_setup runs again)
};

// per-string arrays + tile tables for batches of up to m strings
void ensure_strings(rb2_hip_t *h, uint64_t m)
{
	for (int i = 0; i < 2; ++i) { h->L[i].ensure(m); h->U[i].ensure(m); h->W[i].ensure(m); }
	h->SIZE.ensure(m); h->INS_E.ensure(m); h->RKREL.ensure(m); h->RKOLD.ensure(m); h->SPL.ensure(m + 64);
	h->A[0].ensure(m); h->A[1].ensure(m); h->INS_A.ensure(m); h->START.ensure(m + 1);
	const uint64_t nst = cdiv(m, STILE) + NR;
	h->trec.ensure(nst + 8 * XCD_RUN + 8); h->tsc.ensure(nst + 8 * XCD_RUN + 8); h->tfix.ensure(nst + 8 * XCD_RUN + 8); h->cpart.ensure(cdiv(nst, SCHUNK) + 1);
}

// A batch may hold at most this many strings (32-bit slots, tile numbers and work orders).  The reference takes any count
// (mrope.c:269-277); the single-engine entry points cut a batch that holds more into two and insert them one after the other
// (insert_dev) -- the BWT does not depend on how a read set is cut into batches, which is what `-m` does to every input anyway.
// RB2_MAX_BATCH_STRINGS lowers the limit (tests).
static uint64_t max_batch_strings()
{
	const char *e = getenv("RB2_MAX_BATCH_STRINGS");
	uint64_t v = (1ull << 32) - 2 * STILE;
	if (e && atoll(e) > 0) v = std::min<uint64_t>(v, (uint64_t)atoll(e));
	return v;
}

// split into strings (mrope.c:269-277), size the buffers, initial state (mrope.c:279-284).  false: more strings than a batch may
// hold (B.m says how many) -- nothing was changed, the caller cuts the batch (may_split) or gives up.
bool batch_begin(rb2_hip_t *h, BatchState &B, int64_t len64, const uint8_t *s, bool may_split = false)
{
	const uint64_t len = (uint64_t)len64;
	hipStream_t st = h->st;
	const int is_srt = h->so != RB2_SO_IO;
	const unsigned nzb = cdiv(len, ZBLOCK);
	uint64_t m = 0;
	{
		Scope sc(h, RB2_K_INIT, 0);
		h->zblk.ensure(nzb + 2);
		uint64_t res[2] = {0, 0};                                // [0] number of strings, [1] input-validation flag
		HIPCHK(hipMemsetAsync(h->zblk.p + nzb + 1, 0, 8, st));
		hipLaunchKernelGGL(k_count_zeros, dim3(nzb), dim3(256), 0, st, s, len, h->zblk.p, h->zblk.p + nzb + 1);
		hipLaunchKernelGGL(k_zscan, dim3(1), dim3(SCHUNK), 0, st, h->zblk.p, nzb);
		HIPCHK(hipMemcpyAsync(res, h->zblk.p + nzb, 16, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		m = res[0];
		if (res[1]) { rb2_fatal("[rb2_hip] the batch contains bytes that are not nt6 codes 0..5 ($ACGTN)\n"); }
		if (m >= max_batch_strings() && may_split) { B.m = m; return false; }
		if (m == 0 || m >= max_batch_strings()) { rb2_fatal("[rb2_hip] unsupported number of strings in one batch: %llu\n", (unsigned long long)m); }
		h->START.ensure(m + 1);
		hipLaunchKernelGGL(k_write_starts, dim3(nzb), dim3(256), 0, st, s, len, h->zblk.p, h->START.p);
	}
	ensure_strings(h, m);
	B.nst_ub = cdiv(m, STILE) + NR;                           // string tiles, upper bound for every round
	B.nsc = cdiv(B.nst_ub, SCHUNK);
	uint64_t n_tot = 0;
	for (int b = 0; b < NR; ++b) n_tot += h->h_rope[b].n;     // symbols held by THIS rank
	const uint64_t leaves_ub = (n_tot + len) / LEAF + NR * (SB + 1);
	if (!h->sparse) {                                          // (the sparse layout sizes its pools in relayout())
		h->pool[h->pside].ensure(leaves_ub, true, st);
		h->pool[h->pside ^ 1].ensure(leaves_ub, false, st);
	}
	if (!h->sparse) h->sbtot.ensure(leaves_ub / SB + 1);          // what build_directory will ask for by the batch's last round: growing it there, round by round, is a
	                                                           // hipFree in the middle of the queued rounds -- a device-wide wait (r05: 2 x 30-100 ms of a host-buffer batch)
	h->LD.ensure(std::max<uint64_t>(leaves_ub + NR + 16, m + 64 + (uint64_t)(NR + WLC + 1) * STILE / 2));   // dense: one work order per output window; sparse: at most one per string (+ the slots the last workgroup of k_merge_leaf reads past them)
	B.s = s; B.len = len; B.m = m; B.n_tot = n_tot; B.nsb_ub = leaves_ub / SB + 1; B.cur = 0;
	{ uint64_t n0 = 0; for (int b = 0; b < NR; ++b) n0 += h->h_rope[b].cnt[0]; B.known_ae = B.known_ae0 = !is_srt || n0 == 0; }
	{	// storage width of the positions: narrow while no piece can hold 2^32 symbols (checked again every round: maybe_widen)
		uint64_t mx = 0;
		for (int b = 0; b < NR; ++b) mx = std::max<uint64_t>(mx, h->h_rope[b].n);
		h->pos_m0 = mx;
		h->pos32 = h->want_pos32 && h->pos_mode != 64 && mx + 2 * m < POS32_LIMIT;
		h->want_pos32 = false;
		((volatile unsigned long long*)(h->h_flag + 4))[0] = 0;  // (round, largest piece) as k_setup last reported it: nothing yet
		((volatile uint32_t*)h->h_flag)[2] = 0;                   // (in-place rounds the device has come through: none of this batch)
	}
	{
		Scope sc(h, RB2_K_INIT, 0);
		hipLaunchKernelGGL(k_batch_setup, dim3(1), dim3(1), 0, st, h->ctl, h->side, m, len, is_srt);
		if (h->pos32) hipLaunchKernelGGL(k_init_strings<uint32_t>, dim3(cdiv(m, 256)), dim3(256), 0, st, h->ctl, is_srt, s, h->START.p,
				(uint32_t*)h->L[0].p, (uint32_t*)h->U[0].p, h->W[0].p, h->A[0].p);
		else hipLaunchKernelGGL(k_init_strings<uint64_t>, dim3(cdiv(m, 256)), dim3(256), 0, st, h->ctl, is_srt, s, h->START.p,
				h->L[0].p, h->U[0].p, h->W[0].p, h->A[0].p);
	}
	HIPCHK(hipMemcpyAsync(&B.max_len, &h->ctl->max_len, 8, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	return true;
}

// Once no string of the batch has a non-empty interval, none ever will again (u' = l' + 0): ctl->ne is monotone inside a batch.
// Every round leaves a snapshot of it in a pinned ring WITHOUT synchronising; a snapshot that has landed and reads "the next
// round sees only empty intervals" lets the host launch only the all-empty variants of k_prep / k_advance from then on
// (a stale snapshot only delays that).  Before: both variants every round, the wrong one returning at once (2 x ~18 us).
void ne_snapshot(rb2_hip_t *h, uint64_t r)
{
	uint32_t *slot = h->h_flag + 16 + 2 * (r % rb2_hip_s::NE_RING);
	slot[0] = slot[1] = 0xffffffffu;                           // "not landed yet"
	HIPCHK(hipMemcpyAsync(slot, &h->ctl->ne[0], 8, hipMemcpyDeviceToHost, h->st));
}
bool ne_all_empty_from(rb2_hip_t *h, uint64_t r)               // may round r (and all later ones) skip the non-AE variants?
{
	for (uint64_t q = r; q-- > 0 && r - q < (uint64_t)rb2_hip_s::NE_RING; ) {
		const volatile uint32_t *slot = h->h_flag + 16 + 2 * (q % rb2_hip_s::NE_RING);
		const uint32_t v = slot[(q & 1) ^ 1];                  // after round q: ne[(q&1)^1] is the flag of round q + 1
		if (v == 0) return true;
		if (v != 0xffffffffu) return false;                    // the newest snapshot that landed says "still non-empty intervals"
	}
	return false;
}

// launch F with P = the storage type of this batch's positions; PP(x) = x's array seen as P*
template <class F> inline void with_pos(rb2_hip_t *h, F f) { if (h->pos32) f((uint32_t*)nullptr); else f((uint64_t*)nullptr); }
#define RB2_P(x) ((P*)(x))

uint32_t verdict_check(rb2_hip_t *h, bool drain);
// Leave the narrow storage mode when a piece could reach POS32_LIMIT symbols in round r.  What the host knows: the largest piece as
// k_setup last reported it to pinned memory, for some round q < r (lock-free, possibly many rounds stale -- the host queues rounds
// ahead of the device), and that a piece grows by at most the m strings of the batch per round.
void maybe_widen(rb2_hip_t *h, BatchState &B, uint64_t r)
{
	if (!h->pos32) return;
	const unsigned long long v = ((volatile unsigned long long*)(h->h_flag + 4))[0];
	uint64_t known = h->pos_m0, since = r + 1;                   // rounds the bound must cover
	if (v != 0) { known = v & ((1ull << 40) - 1); const uint64_t q = v >> 40; since = r >= q ? r - q : r + 1; }   // (the report of round q is the size AFTER round q)
	const char *e = getenv("RB2_POS_WIDEN_AT");
	if (known + (since + 1) * B.m < POS32_LIMIT && !(e && (uint64_t)atoll(e) == r)) return;
	if (verdict_check(h, true)) return;                        // (an in-place round queued behind is void: the caller rolls back first and comes here again)
	// widen L and U of the current side (what the next kernels read); scratch arrays are per round
	hipStream_t st = h->st;
	const int cur = B.cur;
	for (DevBuf<uint64_t> *a : { &h->L[cur], &h->U[cur] }) {
		HIPCHK(hipMemcpyAsync(h->INS_E.p, a->p, B.m * 4, hipMemcpyDeviceToDevice, st));
		hipLaunchKernelGGL(k_widen, dim3(cdiv(B.m, 256)), dim3(256), 0, st, (const uint32_t*)h->INS_E.p, a->p, B.m);
	}
	h->pos32 = false;
	B.counted = (uint64_t)-1;                                  // a counting phase queued ahead wrote INS_E in the narrow form (k_sym's fused k_prep): count again
	if (h->trace) fprintf(stderr, "[rb2_hip] round %llu: positions widened to 64 bits (largest piece known: %llu symbols, %llu rounds ago)\n", (unsigned long long)r, (unsigned long long)known, (unsigned long long)since);
}

// phase 1 of a round: next symbols, group heads, tile scans, the rows of the count matrix seen here
// spec: queued while the verdict of the in-place round in front of it is still on its way (round_merge_sparse)
// with_split: the k_sym launch also does the leaf splits of the in-place round in front of it and the verdict event follows it (round_merge_sparse)
static inline void tl_slow(rb2_hip_t *h, const char *what)    // RB2_HIP_TIMELINE=2: which host call of a round took more than half a millisecond
{
	if (h->timeline < 2) return;
	const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	if (h->tl_last != 0 && now - h->tl_last > 0.5) fprintf(stderr, "[rb2_hip] t = %8.3f ms  round %d: %.3f ms on the host up to and including %s\n", now - h->tl_base, h->cur_round, now - h->tl_last, what);
	h->tl_last = now;
}
void round_counts(rb2_hip_t *h, BatchState &B, uint64_t r, bool spec = false, bool with_split = false, bool with_event = true)
{
	hipStream_t st = h->st;
	const int sd = h->side, cur = B.cur;
	const int64_t units = (int64_t)B.m;
	h->cur_round = (int)r;
	const TileRecs trs = { (uint32_t*)h->trec.p, (uint32_t)(h->trec.cap & ~(size_t)3) };   // (20 columns of cap words in the 80-byte records' space)
	tl_slow(h, "start of round_counts");
	const bool one_launch_tail = B.nst_ub < (unsigned)h->ts_max;  // few tiles (long reads): the counting tail is one launch (k_tscan_setup)
	SplitArgs sp; memset(&sp, 0, sizeof(sp));
	if (with_split) {
	  sp.ctl = h->ctl; sp.pool = h->pool[h->pside].view(); sp.SPL = h->SPL.p; sp.spl_cap = (uint32_t)std::min<uint64_t>(h->SPL.cap, 0xffffffffu); sp.epoch = h->split_epoch;
	  sp.hv = (volatile uint32_t*)h->d_flag; sp.nsplitb = 64;
	  sp.round1 = (uint32_t)r;                                   // (the splits of round r - 1)
	  sp.scan2 = (h->dir_ride && !one_launch_tail) ? sp.pool.sbbase : (SbBase*)nullptr;   // the chunk bases of the directory the k_advance launch in front of this one left half-built
	                                                             // (in a block of this launch -- or, when the counting tail is one launch, of that one: see below)
	}
	SbBase *scan2_tail = (with_split && h->dir_ride && one_launch_tail) ? h->pool[h->pside].view().sbbase : (SbBase*)nullptr;
	{ Scope sc(h, RB2_K_SYM, units);
	  with_pos(h, [&](auto *tg_) { using P = std::remove_pointer_t<decltype(tg_)>;
	    if (with_split) {
	      hipLaunchKernelGGL((k_sym<false, P, true>), dim3((unsigned)rank_share(h, B.nst_ub) + sp.nsplitb + (sp.scan2 ? 1u : 0u)), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), RB2_P(h->L[cur].p), RB2_P(h->U[cur].p), h->A[cur].p, trs, sp, RB2_P(h->INS_E.p), h->INS_A.p);
	    } else
	    RB2_LAUNCH_STRIDE(h, (k_sym<true, P>), (k_sym<false, P>), dim3(grid8((unsigned)rank_share(h, B.nst_ub))), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), RB2_P(h->L[cur].p), RB2_P(h->U[cur].p), h->A[cur].p, trs, sp, RB2_P(h->INS_E.p), h->INS_A.p); }); }
	tl_slow(h, "k_sym");
	if (with_split && with_event) HIPCHK(hipEventRecord(h->ev_flag, st));   // (the splits left the verdict in pinned memory)
	if (one_launch_tail) {                                      // one launch instead of six, k_setup included (one GPU)
	  Scope sc(h, RB2_K_TSCAN, units);
	  const int do_setup = h->nranks == 1;
	  const unsigned grid = (unsigned)std::max(1, std::min<int>(TSB, h->ts_blocks)) + (scan2_tail ? 7u : 0u);   // (+ one block per column of the chunk bases)
	  if (h->sparse) hipLaunchKernelGGL(k_tscan_setup<true>, dim3(grid), dim3(SCHUNK), 0, st, h->ctl, sd, (int)(r & 1), trs, h->tfix.p, h->gcnt, do_setup, (int)spec, (uint32_t)r, h->pos32 ? (volatile unsigned long long*)(h->d_flag + 4) : (volatile unsigned long long*)nullptr, scan2_tail);
	  else hipLaunchKernelGGL(k_tscan_setup<false>, dim3(grid), dim3(SCHUNK), 0, st, h->ctl, sd, (int)(r & 1), trs, h->tfix.p, h->gcnt, do_setup, (int)spec, (uint32_t)r, h->pos32 ? (volatile unsigned long long*)(h->d_flag + 4) : (volatile unsigned long long*)nullptr, scan2_tail);
	  if (do_setup) { B.setup_round = r; B.setup_sparse = h->sparse; B.setup_epoch = h->layout_epoch; }
	} else
	{ Scope sc(h, RB2_K_TSCAN, units);
	  hipLaunchKernelGGL(k_tscan1, dim3(B.nsc), dim3(SCHUNK), 0, st, h->ctl, sd, trs, h->cpart.p);
	  if (B.nsc <= (unsigned)h->ts_fold) hipLaunchKernelGGL(k_tscan3<true>, dim3(B.nsc), dim3(SCHUNK), 0, st, h->ctl, sd, trs, h->cpart.p, h->tsc.p);   // (the scan over the chunks folded in)
	  else {
	    hipLaunchKernelGGL(k_tscan2, dim3(1), dim3(SCHUNK), 0, st, h->ctl, sd, h->cpart.p);
	    hipLaunchKernelGGL(k_tscan3<false>, dim3(B.nsc), dim3(SCHUNK), 0, st, h->ctl, sd, trs, h->cpart.p, h->tsc.p);
	  }
	  const int do_setup = h->nranks == 1;                     // one GPU: k_setup of the round rides on block 0 of k_tfix (the local count matrix is the global one)
	  hipLaunchKernelGGL(k_tfix, dim3(std::max<unsigned>(1u, cdiv(B.nst_ub, 256))), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), trs, h->tsc.p, h->tfix.p, h->gcnt, do_setup, (int)h->sparse, (uint32_t)r,
	                     h->pos32 ? (volatile unsigned long long*)(h->d_flag + 4) : (volatile unsigned long long*)nullptr, (int)spec);
	  if (do_setup) { B.setup_round = r; B.setup_sparse = h->sparse; B.setup_epoch = h->layout_epoch; } }
	tl_slow(h, "tile scans");
}

// phase 2: with the global count matrix in h->gcnt: layout, ranks, merge, directory, new intervals.
// send == nullptr: strings go straight to the next-round arrays; else they are written as ShardRec.
// Dense round: every piece is rewritten pool[pside] -> pool[pside ^ 1] (k_merge).
// compact_out: the new windows may be written in the compact format (rb2_merge.h): only k_merge reads them again before the next rewrite
void round_merge(rb2_hip_t *h, BatchState &B, uint64_t r, ShardRec *send, bool compact_out = false)
{
	hipStream_t st = h->st;
	const int sd = h->side, cur = B.cur, is_comp = h->so == RB2_SO_RCLO;
	const int64_t units = (int64_t)B.m;
	PoolView oldp = h->pool[h->pside].view(), newp = h->pool[h->pside ^ 1].view();
	const uint64_t n_new_ub = B.n_tot + std::min<uint64_t>(B.len, (r + 1) * B.m);
	const unsigned nlf = cdiv(n_new_ub, WIN) + NR;            // output windows, upper bound
	const unsigned tg = grid8((unsigned)rank_share(h, B.nst_ub));   // string tiles / output windows this handle launches blocks for (rank_share)
	const unsigned wg = cdiv(B.n_tot + rank_share(h, std::min<uint64_t>(B.len, (r + 1) * B.m)), WIN) + NR;
	if ((uint64_t)nlf * 64 >= (1ull << 32)) { rb2_fatal("[rb2_hip] the index is too large for one k_merge launch (%llu symbols: a launch is capped at 2^32 threads)\n", (unsigned long long)n_new_ub); }
	if (!(B.setup_round == r && !B.setup_sparse && B.setup_epoch == h->layout_epoch))
	{ Scope sc(h, RB2_K_TSCAN, 0);
	  hipLaunchKernelGGL(k_setup<false>, dim3(1), dim3(64), 0, st, h->ctl, sd, h->gcnt, (int)(r & 1), (uint32_t)r, h->pos32 ? (volatile unsigned long long*)(h->d_flag + 4) : (volatile unsigned long long*)nullptr, (int)(h->push[0] != nullptr)); }
	with_pos(h, [&](auto *tg_) { using P = std::remove_pointer_t<decltype(tg_)>;
	{ Scope sc(h, RB2_K_PREP, units);
	  if (!B.known_ae) RB2_LAUNCH_STRIDE(h, (k_prep<false, false, true, P>), (k_prep<false, false, false, P>), dim3(tg), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), is_comp, oldp, RB2_P(h->L[cur].p), RB2_P(h->U[cur].p), h->A[cur].p,
			h->tfix.p, RB2_P(h->INS_E.p), h->INS_A.p, RB2_P(h->SIZE.p));
	  RB2_LAUNCH_STRIDE(h, (k_prep<true, false, true, P>), (k_prep<true, false, false, P>), dim3(h->nranks > 1 ? tg : grid8(cdiv(tg, PREP_PT))), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), is_comp, oldp, RB2_P(h->L[cur].p), RB2_P(h->U[cur].p), h->A[cur].p,
			h->tfix.p, RB2_P(h->INS_E.p), h->INS_A.p, RB2_P(h->SIZE.p)); }
	tl_slow(h, "k_prep");
	{ Scope sc(h, RB2_K_PART, units);
	  RB2_LAUNCH_STRIDE(h, (k_part<true, P>), (k_part<false, P>), dim3(cdiv(wg + NR, 255)), dim3(256), 0, st, h->ctl, sd, RB2_P(h->INS_E.p), h->LD.p, h->pool_compact ? (const uint8_t*)oldp.xh : (const uint8_t*)nullptr); }   // (the formats of the old windows: only a pool side the compact-capable merge wrote has any but plain)
	tl_slow(h, "k_part");
	{ Scope sc(h, RB2_K_MERGE, units);
	  RB2_LAUNCH_STRIDE(h, (k_merge<true, P>), (k_merge<false, P>), dim3(grid8(cdiv(wg, MMW))), dim3(64 * MMW), 0, st, h->ctl, h->LD.p, oldp, newp, RB2_P(h->INS_E.p), h->INS_A.p, h->RKREL.p, (int)compact_out | (h->compact_stats ? 2 : 0), (int)(r & 1)); }
	});
	tl_slow(h, "k_merge");
	{ Scope sc(h, RB2_K_META, units);
	  build_directory(h, sd ^ 1, h->pside ^ 1, std::min<uint64_t>(B.nsb_ub, n_new_ub / (LEAF * SB) + NR + 1), false, false, (uint64_t)wg * WPL / SB + NR + 1); }
	with_pos(h, [&](auto *tg_) { using P = std::remove_pointer_t<decltype(tg_)>;
	tl_slow(h, "directory");
	{ Scope sc(h, RB2_K_ADVANCE, units);
	  if (!B.known_ae) RB2_LAUNCH_STRIDE(h, (k_advance<false, false, true, P>), (k_advance<false, false, false, P>), dim3(tg), dim3(256), 0, st, h->ctl, sd, is_comp, (uint32_t)r, B.s, newp, h->A[cur ^ 1].p, h->A[cur].p, h->tfix.p,
			RB2_P(h->SIZE.p), RB2_P(h->INS_E.p), h->RKREL.p, RB2_P(h->L[cur].p), h->W[cur].p, RB2_P(h->L[cur ^ 1].p), RB2_P(h->U[cur ^ 1].p), h->W[cur ^ 1].p, send, (const P*)nullptr, (const PushTab*)h->push[cur ^ 1], ScanRide{ nullptr, 0u });
	  RB2_LAUNCH_STRIDE(h, (k_advance<true, false, true, P>), (k_advance<true, false, false, P>), dim3(tg), dim3(256), 0, st, h->ctl, sd, is_comp, (uint32_t)r, B.s, newp, h->A[cur ^ 1].p, h->A[cur].p, h->tfix.p,
			RB2_P(h->SIZE.p), RB2_P(h->INS_E.p), h->RKREL.p, RB2_P(h->L[cur].p), h->W[cur].p, RB2_P(h->L[cur ^ 1].p), RB2_P(h->U[cur ^ 1].p), h->W[cur ^ 1].p, send, (const P*)nullptr, (const PushTab*)h->push[cur ^ 1], ScanRide{ nullptr, 0u }); }
	});
	tl_slow(h, "k_advance");
	if (!B.known_ae && !send && h->nranks == 1) ne_snapshot(h, r);
	HIPCHK(hipGetLastError());                                  // a refused launch (grid limits) must not go unnoticed until the end of the batch
	tl_slow(h, "ne snapshot");
	h->side ^= 1; h->pside ^= 1; B.cur ^= 1;
	h->pool_compact = compact_out;
	if (compact_out) ++h->n_compact_rounds;
}

// leaf slots the pools must hold for an index of n symbols in the given layout
uint64_t slots_for(uint64_t n, bool sparse)
{
	if (!sparse) return n / LEAF + NR * (SB + 1);
	return (n / ((uint64_t)SP_FILL * SP_USED) + NR + 1) * SB;
}

// Compact windows (rb2_merge.h "window formats") may only be met by k_merge: whoever else is about to read leaf words of the dense layout -- a
// re-layout, the export, rank queries, the checksums -- says so here.  The round loop keeps the promise (the last round of a batch and the round in
// front of a re-layout write plain windows: round_merge_any, choose_layout); this makes a new reader that forgets about it fail loudly.
void require_plain(rb2_hip_t *h, const char *who)
{
	if (!h->sparse && h->pool_compact) { rb2_fatal("[rb2_hip] internal: %s would read a pool that holds compact windows\n", who); }
}

// copy the index pool[pside] -> pool[pside ^ 1] in the other layout (or the same one, re-spread); descriptors of
// ctl->rope[side] are rewritten in place.  n_ub: upper bound of the symbols held now.
// n_grow: what the index may grow to while it stays in the new layout's pools (dense rounds ping-pong between both pools
// and grow them; sparse rounds never leave their slots).
void relayout(rb2_hip_t *h, bool to_sparse, uint64_t n_ub, uint64_t n_grow)
{
	hipStream_t st = h->st;
	require_plain(h, "a re-layout");
	struct NoWatch { int64_t *keep = t_grow_in_rounds; NoWatch() { t_grow_in_rounds = nullptr; } ~NoWatch() { t_grow_in_rounds = keep; } } nw;   // (a re-layout sizes its target pool here and waits for the device anyway)
	const auto t_host0 = std::chrono::steady_clock::now();
	const uint64_t cap_before[2] = { h->pool[0].cap_leaves, h->pool[1].cap_leaves };
	const uint32_t F = to_sparse ? SP_FILL : LEAF, K = to_sparse ? SP_USED : SB;
	const uint64_t slots = slots_for(n_ub, to_sparse), cap = to_sparse ? slots : slots_for(std::max(n_ub, n_grow), false);
	if (h->pool[h->pside ^ 1].cap_leaves < cap) {              // kernels of earlier rounds may still read the buffers about to be replaced
		HIPCHK(hipStreamSynchronize(st));
		const auto tg0 = std::chrono::steady_clock::now();
		h->pool[h->pside ^ 1].ensure(cap, false, st);
		if (h->trace) fprintf(stderr, "[rb2_hip] pool grown to %.1f M leaf slots in %.3f s\n", h->pool[h->pside ^ 1].cap_leaves / 1e6, std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count());
	}
	{
		Scope sc(h, RB2_K_RELAYOUT, 0);
		hipLaunchKernelGGL(k_relayout_setup, dim3(1), dim3(64), 0, st, h->ctl, h->side, F, K);
		hipLaunchKernelGGL(k_relayout, dim3(std::min<uint64_t>(cdiv(slots, MW), 1u << 20)), dim3(256), 0, st, (const Ctl*)h->ctl, h->side, h->pool[h->pside].view(), h->pool[h->pside ^ 1].view(), F, K, (int)h->sparse);
		build_directory(h, h->side, h->pside ^ 1, slots / SB + 1, to_sparse);
		HIPCHK(hipGetLastError());                              // a refused launch here would leave descriptors without data
	}
	h->pside ^= 1; h->sparse = to_sparse; ++h->n_relayout; ++h->layout_epoch; h->pool_compact = false;   // (k_relayout writes plain windows)
	if (to_sparse) h->sp_nsb = slots / SB + 1;
	if (!to_sparse && h->pool[h->pside ^ 1].cap_leaves < cap) {   // the pool just left becomes the target of the next dense round
		HIPCHK(hipStreamSynchronize(st));
		h->pool[h->pside ^ 1].ensure(cap, false, st);
	}
	if (h->trace) {
		HIPCHK(hipStreamSynchronize(st));
		fprintf(stderr, "[rb2_hip] re-layout to %s: %.1f G symbols, %.3f s on the host clock, pools %.1f / %.1f -> %.1f / %.1f M leaf slots\n", to_sparse ? "sparse" : "dense", n_ub / 1e9,
				std::chrono::duration<double>(std::chrono::steady_clock::now() - t_host0).count(), cap_before[0] / 1e6, cap_before[1] / 1e6, h->pool[0].cap_leaves / 1e6, h->pool[1].cap_leaves / 1e6);
	}
}

// Sparse round: the strings of the batch touch few leaves, each touched leaf is rewritten where it lies (k_merge_leaf),
// the rest of the index keeps its bytes.  Returns false when some leaf could not take its inserts: nothing was
// changed (every kernel behind k_part_sparse saw ctl->overflow and returned) and the caller redoes the round densely.
// send: sharded build -- the surviving strings are written as ShardRec for the exchange (round_merge).
// spec: queue the counting phase of round r + 1 while the verdict of this round travels to the host (single-GPU path; a sharded
// round cannot: its next counting phase needs the strings the other ranks send).
bool round_merge_sparse(rb2_hip_t *h, BatchState &B, uint64_t r, ShardRec *send = nullptr, bool spec = true)
{
	hipStream_t st = h->st;
	const int sd = h->side, cur = B.cur, is_comp = h->so == RB2_SO_RCLO;
	const int64_t units = (int64_t)B.m;
	PoolView pv = h->pool[h->pside].view();
	const unsigned tg = grid8((unsigned)rank_share(h, B.nst_ub));
	if (++h->split_epoch == 0) ++h->split_epoch;
	if (!(B.setup_round == r && B.setup_sparse && B.setup_epoch == h->layout_epoch))
	{ Scope sc(h, RB2_K_TSCAN, 0);
	  hipLaunchKernelGGL(k_setup<true>, dim3(1), dim3(64), 0, st, h->ctl, sd, h->gcnt, (int)(r & 1), (uint32_t)r, h->pos32 ? (volatile unsigned long long*)(h->d_flag + 4) : (volatile unsigned long long*)nullptr, (int)(h->push[0] != nullptr)); }
	const bool lazy = spec && h->lazy_verdict && h->nranks == 1;   // no verdict read here: the caller polls (insert_dev)
	if (!lazy) h->h_flag[0] = h->h_flag[1] = 0;                // the verdict words k_split writes
	else if (h->spec_rounds > (uint64_t)h->run_ahead) {          // not too far ahead of the device: it reports the round whose splits it has reached (split_body -> h_flag[2] = round + 1)
		const volatile uint32_t *prog = (const volatile uint32_t*)h->h_flag + 2;
		const uint32_t want = (uint32_t)r - (uint32_t)h->run_ahead;   // round `want - 1` at least must be over
		const auto t_spin = std::chrono::steady_clock::now();
		for (uint32_t spins = 0; (int32_t)(*prog - want) < 0 && !((const volatile uint32_t*)h->h_flag)[0]; ++spins) {
			if ((spins & 1023u) == 1023u) {
				if (hipStreamQuery(st) == hipSuccess) break;         // (everything queued is done: nothing more will be reported)
				if (std::chrono::steady_clock::now() - t_spin > std::chrono::seconds(2)) { HIPCHK(hipStreamSynchronize(st)); break; }   // (a report that does not come: wait the plain way)
			}
			__builtin_ia32_pause();
		}
	}
	with_pos(h, [&](auto *tg_) { using P = std::remove_pointer_t<decltype(tg_)>;
	{ Scope sc(h, RB2_K_PREP, units);
	  if (!B.known_ae) RB2_LAUNCH_STRIDE(h, (k_prep<false, true, true, P>), (k_prep<false, true, false, P>), dim3(tg), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), is_comp, pv, RB2_P(h->L[cur].p), RB2_P(h->U[cur].p), h->A[cur].p,
			h->tfix.p, RB2_P(h->INS_E.p), h->INS_A.p, RB2_P(h->SIZE.p));
	  RB2_LAUNCH_STRIDE(h, (k_prep<true, true, true, P>), (k_prep<true, true, false, P>), dim3(h->nranks > 1 ? tg : grid8(cdiv(tg, PREP_PT))), dim3(256), 0, st, h->ctl, sd, (int)(r & 1), is_comp, pv, RB2_P(h->L[cur].p), RB2_P(h->U[cur].p), h->A[cur].p,
			h->tfix.p, RB2_P(h->INS_E.p), h->INS_A.p, RB2_P(h->SIZE.p)); }
	{ Scope sc(h, RB2_K_PART, units);
	  RB2_LAUNCH_STRIDE(h, (k_part_sparse<true, P>), (k_part_sparse<false, P>), dim3(tg), dim3(256), 0, st, h->ctl, sd, pv, (const P*)h->INS_E.p, (const uint8_t*)h->INS_A.p, h->tfix.p, (SpOrd*)h->LD.p, h->SPL.p, (uint32_t)std::min<uint64_t>(h->SPL.cap, 0xffffffffu), RB2_P(h->RKOLD.p), (uint32_t)r); }
	{ Scope sc(h, RB2_K_MERGE, units);
	  const unsigned quads = cdiv(rank_share(h, B.m), MW * LROWS * LQ);   // a wave takes LQ x four work orders per step (one leaf per DPP row) and walks the list with a grid stride
	  hipLaunchKernelGGL(k_merge_leaf<P>, dim3(std::max<unsigned>(WLC / MW, (h->leaf_pipe > 0 ? std::min<unsigned>(quads, (unsigned)h->leaf_pipe) : quads) / (WLC / MW) * (WLC / MW))), dim3(256), 0, st, (const Ctl*)h->ctl, (const SpOrd*)h->LD.p, pv, (const P*)h->INS_E.p, (const uint8_t*)h->INS_A.p, h->RKREL.p, h->sbtot.p); }
	});
	// the prefix over the superblock totals: in blocks of their own of the k_advance launch and of the launch behind it (rb2_kernels.h "the directory rides along")
	const bool ride = h->dir_ride != 0;
	const ScanRide sr = { (const SbTot*)h->sbtot.p, ride ? cdiv(h->sp_nsb, SCHUNK) : 0u };
	const ScanRide sr0 = { nullptr, 0u };
	if (!ride)
	{ Scope sc(h, RB2_K_META, units);
	  build_directory(h, sd ^ 1, h->pside, h->sp_nsb, true, true); }
	with_pos(h, [&](auto *tg_) { using P = std::remove_pointer_t<decltype(tg_)>;
	{ Scope sc(h, RB2_K_ADVANCE, units);
	  if (!B.known_ae) RB2_LAUNCH_STRIDE(h, (k_advance<false, true, true, P>), (k_advance<false, true, false, P>), dim3(tg), dim3(256), 0, st, h->ctl, sd, is_comp, (uint32_t)r, B.s, pv, h->A[cur ^ 1].p, h->A[cur].p, h->tfix.p,
			RB2_P(h->SIZE.p), RB2_P(h->INS_E.p), h->RKREL.p, RB2_P(h->L[cur].p), h->W[cur].p, RB2_P(h->L[cur ^ 1].p), RB2_P(h->U[cur ^ 1].p), h->W[cur ^ 1].p, send, (const P*)h->RKOLD.p, (const PushTab*)h->push[cur ^ 1], sr0);
	  RB2_LAUNCH_STRIDE(h, (k_advance<true, true, true, P>), (k_advance<true, true, false, P>), dim3(tg + sr.nscan), dim3(256), 0, st, h->ctl, sd, is_comp, (uint32_t)r, B.s, pv, h->A[cur ^ 1].p, h->A[cur].p, h->tfix.p,
			RB2_P(h->SIZE.p), RB2_P(h->INS_E.p), h->RKREL.p, RB2_P(h->L[cur].p), h->W[cur].p, RB2_P(h->L[cur ^ 1].p), RB2_P(h->U[cur ^ 1].p), h->W[cur ^ 1].p, send, (const P*)h->RKOLD.p, (const PushTab*)h->push[cur ^ 1], sr); }   // (this launch always runs: the scan blocks ride here)
	});
	// leaves that came close to full get a second slot of their superblock now: the last kernel of the round (k_split, rb2_kernels.h) --
	// or, when the counting phase of round r + 1 is queued at once (spec), blocks of their own in its first launch (k_sym<.., SPLIT>)
	const bool sp = spec && r + 1 <= B.max_len;
	if (!sp)
	{ Scope sc(h, RB2_K_SPLIT, 0);
	  hipLaunchKernelGGL(k_split, dim3(256 + (ride ? 1 : 0)), dim3(256), 0, st, h->ctl, pv, (const uint32_t*)h->SPL.p, (uint32_t)std::min<uint64_t>(h->SPL.cap, 0xffffffffu), h->split_epoch, (volatile uint32_t*)h->d_flag,
	                     ride ? pv.sbbase : (SbBase*)nullptr, (uint32_t)r + 1u); }
	// The verdict of the round (did every leaf fit?  did every split find a slot?) travels to pinned host memory behind the last
	// kernel.  While it is on its way the host already queues the counting phase of round r + 1 -- it only writes per-round scratch,
	// and a void round r is redone from its own counting phase anyway -- so the GPU has work while the host waits and then queues
	// the next merge.
	if (!B.known_ae && !send && h->nranks == 1) ne_snapshot(h, r);
	HIPCHK(hipGetLastError());
	if (!sp && !lazy) HIPCHK(hipEventRecord(h->ev_flag, st));  // (k_split left the verdict in pinned memory)
	h->side ^= 1; B.cur ^= 1;
	if (sp) round_counts(h, B, r + 1, true, true, !lazy);      // (records the event behind its first launch)
	if (lazy) {                                                // one engine: nobody waits; a void round makes every kernel queued behind it return (sticky: k_part_sparse)
		B.counted = sp ? r + 1 : (uint64_t)-1;
		++h->n_sparse_rounds; ++h->spec_rounds;
		return true;
	}
	HIPCHK(hipEventSynchronize(h->ev_flag));
	if (h->h_flag[0]) {                                         // void: nothing was changed; the flag comes down here, with nothing queued behind the round
		h->side ^= 1; B.cur ^= 1; B.counted = (uint64_t)-1;
		HIPCHK(hipMemsetAsync(&h->ctl->overflow, 0, 4, st));
		return false;
	}
	if (h->h_flag[1]) h->want_respread = true;
	B.counted = sp ? r + 1 : (uint64_t)-1;
	++h->n_sparse_rounds;
	return true;
}

// Has an in-place round queued since the last look been void (returns round + 1, else 0)?  drain: wait for everything queued first -- the
// answer is then final; else whatever has landed in pinned memory so far.  A superblock that ran out of slots asks for a re-spread.
uint32_t verdict_check(rb2_hip_t *h, bool drain)
{
	if (!h->spec_rounds) return 0;
	if (drain) HIPCHK(hipStreamSynchronize(h->st));
	const uint32_t v = ((volatile uint32_t*)h->h_flag)[0];
	if (((volatile uint32_t*)h->h_flag)[1]) { h->want_respread = true; if (drain) h->h_flag[1] = 0; }
	if (drain && !v) h->spec_rounds = 0;
	return v;
}

void batch_end(rb2_hip_t *h)
{
	h->cur_round = -1;
	HIPCHK(hipGetLastError());
	fetch_ropes(h);
	drain_profile(h);
}

// Which regime is a round in?  lambda = strings of the round / leaves of the index: the dense rewrite costs ~ the index, the
// in-place round ~ the touched leaves (+ a search per string).  The host only knows upper bounds of both; that is enough.
// m_eff: the strings this handle expects per round (the batch on one GPU; its share of it on a rank of a sharded index).
// Changes the layout when the regime changes (dense <-> sparse) and re-spreads a sparse index whose superblocks ran out of slots.
// false: an in-place round queued earlier turned out void (found when the stream was drained in front of a re-layout): nothing was done, the caller rolls back
bool choose_layout(rb2_hip_t *h, BatchState &B, uint64_t r, uint64_t m_eff)
{
	const uint64_t n_ub = B.n_tot + std::min<uint64_t>(B.len, r * B.m);            // symbols in the index before this round
	const double lambda = (double)m_eff / ((double)n_ub / LEAF + 1.0);
	bool want = h->sp_lambda > 0 && lambda < h->sp_lambda && h->sp_backoff == 0 && B.m < (1ull << 27);   // (k_merge_leaf: one wave per LROWS work orders, 2^32 threads per launch)
	if (h->sparse && !want && lambda < 2 * h->sp_lambda && h->sp_backoff == 0) want = true;   // hysteresis
	if (h->sp_backoff > 0) --h->sp_backoff;
	h->plain_next = false;
	if (want && !h->sparse && h->pool_compact) { want = false; h->plain_next = true; ++h->n_plain_handover; }   // the re-layout reads plain windows: this round writes them, the next one switches
	if (want && !h->sparse) {                              // the sparse pool is 1.8x one dense side and lives next to both: only if it fits
		const uint64_t need = slots_for(n_ub, true);        // both pools take turns as the target of a re-layout: both must be able to grow
		const int grow = (need > h->pool[0].cap_leaves) + (need > h->pool[1].cap_leaves);
		size_t fr = 0, tot = 0;
		double leaves_more = 0;                                // pools that grow in place need what is added (+ the 25 % margin of Pool::ensure); others a whole new buffer beside the old one
		for (int k = 0; k < 2; ++k)
			if (need > h->pool[k].cap_leaves) leaves_more += h->pool[k].data.vm_res ? 1.25 * (double)need - (double)h->pool[k].cap_leaves : (double)need * 1.25;
		const double bytes = leaves_more * (LEAFB + 2 * sizeof(LeafMeta) + 2.0) * 1.02 + (256u << 20);
		if (grow && hipMemGetInfo(&fr, &tot) == hipSuccess && bytes > (double)fr) {
			want = false; h->sp_backoff = 1 << 30;             // not again in this handle's lifetime
			if (h->trace) fprintf(stderr, "[rb2_hip] not enough free device memory for the sparse layout (%.1f GB needed, %.1f GB free): staying dense\n", bytes / 1e9, fr / 1e9);
		}
	}
	if (want != h->sparse || (h->sparse && h->want_respread)) {
		if (verdict_check(h, true)) return false;              // (a re-layout copies the index as it is: every in-place round queued so far must have taken place)
	}
	if (want != h->sparse) { relayout(h, want, n_ub, B.n_tot + B.len); h->want_respread = false; }
	else if (h->sparse && h->want_respread) {               // a superblock ran out of slots: spread the index over fresh superblocks (sparse -> sparse)
		if (h->trace) fprintf(stderr, "[rb2_hip] round %llu: re-spread (a superblock has no free slot left)\n", (unsigned long long)r);
		relayout(h, true, n_ub, B.n_tot + B.len);
		h->want_respread = false; ++h->n_respread;
	}
	return true;
}

// the merge phase of round r in whatever layout the index has (its counting phase is queued); a void in-place round is redone densely
// round r could not be done in place (nothing was changed): back to the dense layout, stay dense for a while (doubling: hot spots tend to persist)
void void_to_dense(rb2_hip_t *h, BatchState &B, uint64_t r)
{
	++h->n_void;
	if (h->trace) fprintf(stderr, "[rb2_hip] void round %llu (penalty %d)\n", (unsigned long long)r, h->sp_penalty);
	const uint64_t n_ub = B.n_tot + std::min<uint64_t>(B.len, r * B.m);
	relayout(h, false, n_ub, B.n_tot + B.len);
	h->sp_penalty = std::min(h->sp_penalty + 1, h->sp_maxpen);
	h->sp_backoff = 1 << h->sp_penalty;
}

void round_merge_any(rb2_hip_t *h, BatchState &B, uint64_t r, ShardRec *send, bool spec)
{
	if (h->sparse) {
		if (round_merge_sparse(h, B, r, send, spec)) { if (h->sp_penalty > 0 && (h->n_sparse_rounds & 63) == 0) --h->sp_penalty; return; }
		// void round: redone on the dense layout
		void_to_dense(h, B, r);
		if (spec) round_counts(h, B, r);                       // the scratch of this round's counting phase was reused by the look-ahead
	}
	// compact windows (rb2_merge.h) while k_merge is the only reader of the pool until the next rewrite: every interval of the batch is
	// empty from here on (k_prep<AE> reads no leaf), the batch goes on (its last round leaves plain windows to whoever comes next:
	// export, rank queries, the next batch's first rounds), and no re-layout is pending
	// (one engine: the kernel itself checks that every interval is empty -- the host learns that ~20 rounds late, from the ne snapshots; a
	// rank of a sharded index may receive a string with a non-empty interval from another rank: there the global flag decides)
	round_merge(h, B, r, send, h->compact_ok && (B.known_ae || h->nranks == 1) && r < B.max_len && !h->plain_next);
}

void batch_trace(rb2_hip_t *h)
{
	if (h->trace) fprintf(stderr, "[rb2_hip] batch done: layout %s, relayouts %lld (of them re-spreads %lld), void sparse rounds %lld, sparse rounds %lld\n", h->sparse ? "sparse" : "dense",
			(long long)h->n_relayout, (long long)h->n_respread, (long long)h->n_void, (long long)h->n_sparse_rounds);
}

// A host-buffer insert may return as soon as its text is on the device and its rounds are queued (`lazy`): the caller gets its buffer
// back, and what it does next -- parse the next batch, upload it -- runs beside the kernels.  Every entry point that looks at the
// handle's state waits for them first (finish_pending).
void finish_pending(rb2_hip_t *h)
{
	if (!h || !h->pending_end) return;
	h->pending_end = false;
	HIPCHK(hipSetDevice(h->dev));
	batch_end(h);
	batch_trace(h);
}

// the byte behind a sentinel near the middle of a device text of more than one string: where a batch of too many strings is cut
static int64_t split_point(rb2_hip_t *h, const uint8_t *s, int64_t len)
{
	const int64_t W = 1 << 20;
	std::vector<uint8_t> win((size_t)W);
	for (int64_t a = len / 2; a < len - 1; a += W) {            // forwards from the middle ...
		const int64_t n = std::min(W, len - 1 - a);
		HIPCHK(hipMemcpyAsync(win.data(), s + a, (size_t)n, hipMemcpyDeviceToHost, h->st)); HIPCHK(hipStreamSynchronize(h->st));
		const void *z = memchr(win.data(), 0, (size_t)n);
		if (z) return a + ((const uint8_t*)z - win.data()) + 1;
	}
	for (int64_t b = len / 2; b > 0; b -= W) {                   // ... else backwards
		const int64_t a = std::max<int64_t>(0, b - W), n = b - a;
		HIPCHK(hipMemcpyAsync(win.data(), s + a, (size_t)n, hipMemcpyDeviceToHost, h->st)); HIPCHK(hipStreamSynchronize(h->st));
		for (int64_t i = n - 1; i >= 0; --i) if (win[(size_t)i] == 0) return a + i + 1;
	}
	rb2_fatal("[rb2_hip] insert_multi: cannot cut a batch of one string\n");
}

void insert_dev(rb2_hip_t *h, int64_t len64, const uint8_t *s, bool lazy = false)
{
	if (h->nranks > 1) { rb2_fatal("[rb2_hip] this handle is one rank of a rope-sharded index: insert through its rb2_hip_multi_t\n"); }
	BatchState B;
	h->want_pos32 = true;                                       // (one engine, whole index: the narrow position storage may be used)
	if (!batch_begin(h, B, len64, s, true)) {
		// more strings than one batch may hold: two batches, one after the other (the second half moves to a 16-byte aligned place)
		const int64_t p = split_point(h, s, len64);
		if (h->trace) fprintf(stderr, "[rb2_hip] batch of %llu strings cut at byte %lld of %lld\n", (unsigned long long)B.m, (long long)p, (long long)len64);
		insert_dev(h, p, s, false);
		DevBuf<uint8_t> half;
		half.ensure((size_t)(len64 - p) + 64);
		HIPCHK(hipMemcpyAsync(half.p, s + p, (size_t)(len64 - p), hipMemcpyDeviceToDevice, h->st));
		insert_dev(h, len64 - p, half.p, false);
		HIPCHK(hipStreamSynchronize(h->st));
		half.release();
		return;
	}
	// The first rounds of a batch are hot spots by construction: round 0 puts every string into rope $ (at its end in input order,
	// at a handful of positions in the sorted orders), round k touches ~4^k places.  A sparse index would void each of them (a
	// re-layout there and back per round); one dense phase of eight rounds costs two re-layouts for all of them.
	if (h->sparse && h->sp_backoff < h->sp_head) h->sp_backoff = h->sp_head;
	struct GrowWatch { GrowWatch(int64_t *c) { t_grow_in_rounds = c; } ~GrowWatch() { t_grow_in_rounds = nullptr; } } gw(&h->n_grow_in_rounds);
	for (uint64_t r = 0; ; ) {                                 // one round per string position, last symbol first (mrope.c:285, 299-342)
		// In-place rounds are queued without a look at their verdict.  The host looks here -- at what has landed in pinned memory when it queues
		// the next round, with the stream drained at the end of the batch and in front of anything that is not an in-place round (a re-layout,
		// a widening of the positions: they come back with rv set) -- and takes a void round back: every kernel queued behind it returned at
		// once (rb2_kernels.h k_part_sparse), so the device is where it was in front of that round and only the host's own bookkeeping rewinds.
		uint32_t rv = verdict_check(h, r > B.max_len);
		if (!rv && r > B.max_len) break;
		if (!rv) {
			if (h->timeline > 1 && (r < 4 || r % 10 == 0)) fprintf(stderr, "[rb2_hip] t = %8.3f ms  queueing round %llu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - h->tl_base, (unsigned long long)r);
			if (!B.known_ae && r > 0 && ne_all_empty_from(h, r)) B.known_ae = true;
			maybe_widen(h, B, r);
			if (!(rv = verdict_check(h, false)) && !choose_layout(h, B, r, B.m)) rv = verdict_check(h, true);
		}
		if (rv) {
			rv = verdict_check(h, true);                           // (final: everything queued has run -- or returned)
			const uint64_t r_void = rv - 1;
			if (r_void >= r || r - r_void > h->spec_rounds) { rb2_fatal("[rb2_hip] internal: void round %llu reported while queueing round %llu (%llu in flight)\n", (unsigned long long)r_void, (unsigned long long)r, (unsigned long long)h->spec_rounds); }
			if ((r - r_void) & 1) { h->side ^= 1; B.cur ^= 1; }     // every in-place round queued from r_void on flipped the descriptor and array sides once
			h->n_sparse_rounds -= (int64_t)(r - r_void);
			h->spec_rounds = 0; h->h_flag[0] = 0;
			HIPCHK(hipMemsetAsync(&h->ctl->overflow, 0, 4, h->st));
			B.counted = (uint64_t)-1; B.setup_round = (uint64_t)-1;
			// the "every interval is empty from here on" snapshots of the void round and of the rounds behind it are not what those rounds leave
			// when they really run (the void round's k_advance never set the flag): forget them, and what the host concluded from them
			B.known_ae = B.known_ae0;
			for (int q = 0; q < 2 * rb2_hip_s::NE_RING; ++q) h->h_flag[16 + q] = 0xffffffffu;
			r = r_void;
			void_to_dense(h, B, r);
			round_counts(h, B, r);
			round_merge_any(h, B, r, nullptr, true);
			++r;
			continue;
		}
		if (B.counted != r) round_counts(h, B, r);
		round_merge_any(h, B, r, nullptr, true);
		++r;
	}
	if (lazy && h->lazy_insert && !h->prof && !h->debug) { h->cur_round = -1; HIPCHK(hipGetLastError()); h->pending_end = true; return; }
	batch_end(h);
	batch_trace(h);
}

// the batch must end with a sentinel (mrope.c:268): bytes after the last 0 would be sized for but never inserted
void check_last_byte(rb2_hip_t *h, int64_t len, const uint8_t *s_dev)
{
	uint8_t last = 1;
	HIPCHK(hipMemcpyAsync(&last, s_dev + len - 1, 1, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	if (last != 0) { rb2_fatal("[rb2_hip] insert_multi: buffer must be non-empty and end with a sentinel\n"); }
}

} // namespace

// =============================================================================================
// C ABI
// =============================================================================================

extern "C" {

void rb2_hip_set_fatal_handler(rb2_hip_fatal_cb cb, void *user) { g_fatal_cb = cb; g_fatal_user = user; }

int rb2_hip_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

rb2_hip_t *rb2_hip_create(int device, int sorting_order)
{
	int n = rb2_hip_device_count();
	if (n <= 0 || device < 0 || device >= n) {
		rb2_fatal("[rb2_hip] no usable HIP device (requested %d, visible %d); this engine has no CPU fallback\n", device, n);
	}
	if (sorting_order < 0 || sorting_order > 2) { rb2_fatal("[rb2_hip] bad sorting order %d\n", sorting_order); }   // mrope.c:18
	HIPCHK(hipSetDevice(device));
	rb2_hip_t *h = new rb2_hip_s();
	h->dev = device; h->so = sorting_order;
	h->debug = getenv("RB2_HIP_DEBUG") ? atoi(getenv("RB2_HIP_DEBUG")) : 0;
	h->trace = getenv("RB2_HIP_TRACE") ? atoi(getenv("RB2_HIP_TRACE")) : 0;
	if (getenv("RB2_SPARSE_LAMBDA")) h->sp_lambda = atof(getenv("RB2_SPARSE_LAMBDA"));   // 0: never leave the dense layout
	if (getenv("RB2_SPARSE_HEAD")) h->sp_head = atoi(getenv("RB2_SPARSE_HEAD"));
	if (getenv("RB2_LEAF_PIPE")) h->leaf_pipe = atoi(getenv("RB2_LEAF_PIPE"));
	if (getenv("RB2_COMPACT")) h->compact_ok = atoi(getenv("RB2_COMPACT"));
	if (getenv("RB2_COMPACT_STATS")) h->compact_stats = atoi(getenv("RB2_COMPACT_STATS"));
	if (getenv("RB2_TS_MAX")) h->ts_max = std::max(0, std::min((int)TS_MAX, atoi(getenv("RB2_TS_MAX"))));
	if (getenv("RB2_TS_FOLD")) h->ts_fold = std::max(0, std::min((int)SCHUNK, atoi(getenv("RB2_TS_FOLD"))));   // tests: 0 = the scan over the chunks always as a launch of its own (k_tscan2)
	if (getenv("RB2_HIP_LAZY_INSERT")) h->lazy_insert = atoi(getenv("RB2_HIP_LAZY_INSERT"));
	if (getenv("RB2_HIP_TIMELINE")) h->timeline = atoi(getenv("RB2_HIP_TIMELINE"));
	if (getenv("RB2_SPARSE_MAXPEN")) h->sp_maxpen = atoi(getenv("RB2_SPARSE_MAXPEN"));     // tests: 0 = retry the sparse layout after every dense fallback round
	if (h->trace) h->prof = 1;
	HIPCHK(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
	HIPCHK(hipMalloc((void**)&h->ctl, sizeof(Ctl)));
	HIPCHK(hipMalloc((void**)&h->d_tmp, 256));
	HIPCHK(hipMalloc((void**)&h->gcnt, GCN * 8));
	HIPCHK(hipHostMalloc((void**)&h->h_flag, 64 + 8 * rb2_hip_s::NE_RING, hipHostMallocDefault));
	memset(h->h_flag, 0, 64 + 8 * rb2_hip_s::NE_RING);
	HIPCHK(hipHostGetDevicePointer((void**)&h->d_flag, h->h_flag, 0));
	HIPCHK(hipEventCreateWithFlags(&h->ev_flag, hipEventDisableTiming));
	if (getenv("RB2_DIR_RIDE")) h->dir_ride = atoi(getenv("RB2_DIR_RIDE"));
	if (getenv("RB2_TS_BLOCKS")) h->ts_blocks = atoi(getenv("RB2_TS_BLOCKS"));
	if (getenv("RB2_LAZY_VERDICT")) h->lazy_verdict = atoi(getenv("RB2_LAZY_VERDICT"));
	if (getenv("RB2_RUN_AHEAD")) h->run_ahead = std::max(1, atoi(getenv("RB2_RUN_AHEAD")));
	{ Ctl *hc = (Ctl*)calloc(1, sizeof(Ctl)); for (int b = 0; b < NR; ++b) hc->own[b] = 1; HIPCHK(hipMemcpy(h->ctl, hc, sizeof(Ctl), hipMemcpyHostToDevice)); free(hc); }
	HIPCHK(hipMemsetAsync(h->d_tmp, 0, 256, h->st));
	memset(h->h_rope, 0, sizeof(h->h_rope));
	memset(h->p_launch, 0, sizeof(h->p_launch)); memset(h->p_ms, 0, sizeof(h->p_ms)); memset(h->p_units, 0, sizeof(h->p_units));
	h->pool[0].ensure(SB * 8, false, h->st); h->pool[1].ensure(SB * 8, false, h->st);
	HIPCHK(hipStreamSynchronize(h->st));
	return h;
}

void rb2_hip_destroy(rb2_hip_t *h)
{ finish_pending(h);
	if (!h) return;
	HIPCHK(hipSetDevice(h->dev));
	if (h->own_stream) HIPCHK(hipStreamSynchronize(h->st)); else HIPCHK(hipDeviceSynchronize());   /* a caller's stream (rb2_hip_use_stream) may be gone already */
	for (int i = 0; i < 2; ++i) { h->pool[i].release(); h->L[i].release(); h->U[i].release(); h->W[i].release(); }
	h->START.release(); h->SIZE.release(); h->INS_E.release(); h->RKREL.release(); h->RKOLD.release(); h->SPL.release(); h->qbuf.release(); h->zblk.release();
	h->LD.release(); h->A[0].release(); h->A[1].release(); h->INS_A.release(); h->sbuf.release(); h->sbuf2.release();
	if (h->st_copy) HIPCHK(hipStreamDestroy(h->st_copy));
	h->trec.release(); h->tsc.release(); h->tfix.release(); h->cpart.release(); h->sbtot.release();
	for (auto e : h->evpool) hipEventDestroy(e);
	HIPCHK(hipHostFree(h->h_flag)); HIPCHK(hipEventDestroy(h->ev_flag));

	if (h->pair_d) { HIPCHK(hipFree(h->pair_d)); HIPCHK(hipHostFree(h->pair_h)); }
	HIPCHK(hipFree(h->ctl)); HIPCHK(hipFree(h->d_tmp)); HIPCHK(hipFree(h->gcnt)); h->xstage.release(); h->xnb.release(); h->xpack.release(); h->xoff.release();
	for (int i = 0; i < 2; ++i) if (h->xhost[i]) { HIPCHK(hipHostFree(h->xhost[i])); HIPCHK(hipHostFree(h->xtot[i])); HIPCHK(hipEventDestroy(h->xev[i])); }
	if (h->own_stream) HIPCHK(hipStreamDestroy(h->st));
	delete h;
}

int rb2_hip_sorting_order(const rb2_hip_t *h) { return h->so; }

/* back to the empty index of rb2_hip_create; buffers, shard ownership and profiling state are kept */
void rb2_hip_reset(rb2_hip_t *h)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	memset(h->h_rope, 0, sizeof(h->h_rope));
	h->sparse = false; h->sp_backoff = h->sp_penalty = 0;
	HIPCHK(hipMemsetAsync(&h->ctl->rope[0][0], 0, sizeof(RopeDesc) * 2 * NR, h->st));
	HIPCHK(hipMemsetAsync(&h->ctl->nsb_total, 0, 8, h->st));
}

void rb2_hip_insert_multi_dev(rb2_hip_t *h, int64_t len, const uint8_t *s_dev)
{ finish_pending(h); h->pair_valid = false;
	HIPCHK(hipSetDevice(h->dev));
	if (len <= 0) { rb2_fatal("[rb2_hip] insert_multi: len must be > 0\n"); }   // mrope.c:268
	check_last_byte(h, len, s_dev);
	if (((uintptr_t)s_dev & 15) != 0) {              // kernels use 16-byte loads
		h->sbuf.ensure((size_t)len + 64);
		HIPCHK(hipMemcpyAsync(h->sbuf.p, s_dev, (size_t)len, hipMemcpyDeviceToDevice, h->st));
		s_dev = h->sbuf.p;
	}
	insert_dev(h, len, s_dev);
}

/* The caller is still assembling the batch it will pass to rb2_hip_insert_multi next: bytes [0, n_final) of `s` are final.
 * Upload what is new since the last call on a copy stream, into a second text buffer -- concurrently with whatever the handle is
 * inserting on another thread (the CLI's reader thread calls this while its inserter thread is inside mr_insert_multi for the
 * previous batch): when the batch is handed over, only its tail still has to cross PCIe.  `capacity` = the largest size the
 * batch may reach (the device buffer must not move once bytes are in it).  The buffer `s` must not be reallocated or freed
 * between the first prefetch and the insert.  Entirely optional: an insert of a buffer that was not prefetched uploads all of it. */
void rb2_hip_prefetch(rb2_hip_t *h, const uint8_t *s, int64_t n_final, int64_t capacity)
{
	std::unique_lock<std::mutex> lk(h->pf_mu);
	HIPCHK(hipSetDevice(h->dev));
	h->pf_cv.wait(lk, [h] { return !h->pf_busy; });             // one copy at a time
	if (!h->st_copy) HIPCHK(hipStreamCreateWithFlags(&h->st_copy, hipStreamNonBlocking));
	if (s == nullptr) {                                          // cancel: the caller is about to move or free the buffer -- no copy may still read it
		HIPCHK(hipStreamSynchronize(h->st_copy));
		h->pf_host = nullptr; h->pf_done = 0;
		return;
	}
	if (h->pf_host != s) { h->pf_host = s; h->pf_done = 0; }
	if (n_final <= (int64_t)h->pf_done) return;
	const size_t need = (size_t)std::max(n_final, capacity) + 64;
	const size_t from = need > h->sbuf2.cap ? 0 : h->pf_done;  // (first call of a batch, normally; a buffer that has to grow starts over)
	h->pf_busy = true;
	lk.unlock();                                                 // the copy keeps this thread busy for its whole duration (pageable memory): not under the lock
	if (need > h->sbuf2.cap) { HIPCHK(hipStreamSynchronize(h->st_copy)); h->sbuf2.ensure(need); }
	HIPCHK(hipMemcpyAsync(h->sbuf2.p + from, s + from, (size_t)n_final - from, hipMemcpyHostToDevice, h->st_copy));
	lk.lock();
	h->pf_done = (size_t)n_final;
	h->pf_busy = false;
	h->pf_cv.notify_all();
}

int rb2_hip_host_register(void *p, int64_t nbytes)
{
	if (!p || nbytes <= 0) return -1;
	if (hipHostRegister(p, (size_t)nbytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return -1; }
	return 0;
}
int rb2_hip_host_unregister(void *p)
{
	if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return -1; }
	return 0;
}

void rb2_hip_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes)
{
	size_t fr = 0, tot = 0;
	if (rb2_hip_device_count() <= device || hipSetDevice(device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { fr = tot = 0; (void)hipGetLastError(); }
	if (free_bytes) *free_bytes = (int64_t)fr;
	if (total_bytes) *total_bytes = (int64_t)tot;
}

void rb2_hip_insert_multi(rb2_hip_t *h, int64_t len, const uint8_t *s)
{
	HIPCHK(hipSetDevice(h->dev));
	if (len <= 0 || s[len - 1] != 0) { rb2_fatal("[rb2_hip] insert_multi: buffer must be non-empty and end with a sentinel\n"); }   // mrope.c:268
	static const double tl0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	h->tl_base = tl0;
	auto tl = [&](const char *what) { if (h->timeline) fprintf(stderr, "[rb2_hip] t = %8.3f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - tl0, what); };
	tl("insert_multi: enter");
	// The text goes into the SECOND text buffer on the copy stream -- the first may still be read by the rounds of the previous
	// insert, which was allowed to return before they were done (finish_pending): a caller that inserts batch after batch has batch
	// k + 1 cross PCIe while batch k is being inserted.  Whatever rb2_hip_prefetch has brought over already is not sent again.
	{
		std::unique_lock<std::mutex> lk(h->pf_mu);
		if (!h->st_copy) HIPCHK(hipStreamCreateWithFlags(&h->st_copy, hipStreamNonBlocking));
		h->pf_cv.wait(lk, [h] { return !h->pf_busy; });          // a prefetch copy is running (of this batch, or a stale one): it owns sbuf2 until it is done
		size_t have = 0;
		if (h->pf_host == s && h->pf_done > 0 && (int64_t)h->pf_done <= len && h->sbuf2.cap >= (size_t)len + 64) have = h->pf_done;
		h->pf_host = nullptr; h->pf_done = 0;
		h->pf_busy = true;                                       // (keeps a concurrent rb2_hip_prefetch of the NEXT batch out of sbuf2 until the buffers are swapped)
		lk.unlock();
		if (h->sbuf2.cap < (size_t)len + 64) { HIPCHK(hipStreamSynchronize(h->st_copy)); h->sbuf2.ensure((size_t)len + 64); have = 0; }
		if ((size_t)len > have) HIPCHK(hipMemcpyAsync(h->sbuf2.p + have, s + have, (size_t)len - have, hipMemcpyHostToDevice, h->st_copy));   // (six host threads staging through pinned
		                                                           // buffers of their own were SLOWER than the runtime's pageable path here: 0.90-1.06 s against 0.77 s per configs[1] job, r04)
		if (!h->pair_d) { HIPCHK(hipMalloc((void**)&h->pair_d, 36 * 8)); HIPCHK(hipHostMalloc((void**)&h->pair_h, 36 * 8, hipHostMallocDefault)); }
		HIPCHK(hipMemsetAsync(h->pair_d, 0, 36 * 8, h->st_copy));  // the count matrix of the batch, from its text alone (rb2_hip_last_batch_counts)
		hipLaunchKernelGGL(k_pair_hist, dim3((unsigned)std::min<uint64_t>(((uint64_t)len + 4095) / 4096, 8192)), dim3(256), 0, h->st_copy, (const uint8_t*)h->sbuf2.p, (uint64_t)len, h->pair_d);
		HIPCHK(hipMemcpyAsync(h->pair_h, h->pair_d, 36 * 8, hipMemcpyDeviceToHost, h->st_copy));
		tl("upload queued");
		HIPCHK(hipStreamSynchronize(h->st_copy));
		h->pair_valid = true;
		tl("upload done");
		finish_pending(h);                                       // the previous batch is done with sbuf: it becomes the target of the next upload
		lk.lock();
		std::swap(h->sbuf, h->sbuf2);
		h->pf_busy = false;
		h->pf_cv.notify_all();
	}
	tl("previous batch done");
	insert_dev(h, len, h->sbuf.p, true);
	tl("rounds queued");
}

/* what the last rb2_hip_insert_multi adds to the count matrix (d[b*6+a], layout of rb2_hip_get_counts), computed from the text of the
 * batch alone: available as soon as that call returns, without waiting for its rounds.  0 if the last insert was not a host-buffer one. */
int rb2_hip_last_batch_counts(rb2_hip_t *h, int64_t d[36])
{
	if (!h->pair_valid) return 0;
	for (int i = 0; i < 36; ++i) d[i] = (int64_t)h->pair_h[i];
	return 1;
}

void rb2_hip_set_lazy(rb2_hip_t *h, int on) { finish_pending(h); h->lazy_insert = on; }
void rb2_hip_wait(rb2_hip_t *h) { finish_pending(h); }

void rb2_hip_get_counts(rb2_hip_t *h, int64_t c[36])
{ finish_pending(h);
	for (int i = 0; i < 36; ++i) c[i] = 0;
	for (int r = 0; r < NR; ++r) for (int a = 0; a < 6; ++a) c[rope_sym(r) * 6 + a] += (int64_t)h->h_rope[r].cnt[a];   // rope b = its pieces (b,x)
}

/* run-length export of sub-rope r in chunks of CH leaves: k_export writes one 43+3 byte per run into a staging
 * buffer (slot stride LEAF) + the byte count of every leaf; dst == NULL only counts.  Returns the bytes. */
static void ensure_dense(rb2_hip_t *h)          /* k_export streams flat pieces: leave the sparse layout first */
{
	if (!h->sparse) return;
	uint64_t n = 0;
	for (int r = 0; r < NR; ++r) n += h->h_rope[r].n;
	relayout(h, false, n, n);
	fetch_ropes(h);
}

/* run-length export of sub-rope r: k_export writes one 43+3 byte per run into a staging buffer (slot stride XCHUNK) and the
 * byte count of every chunk, k_xcompact packs the slots back to back, and the packed bytes travel to one of two pinned
 * host buffers -- while the host consumes one (dst copy / one callback per staging round), the device fills the other. */
static int64_t export_piece(rb2_hip_t *h, int r, uint8_t *dst, rb2_hip_run_cb cb = nullptr, void *user = nullptr)
{
	require_plain(h, "the export");
	const uint64_t CH = 32768;                       // export chunks (XCHUNK symbols each) per staging round: at most 32 MiB of run bytes
	const RopeDesc &d = h->h_rope[r];
	const uint64_t nchunks = (d.n + XCHUNK - 1) / XCHUNK;
	if (nchunks == 0 || d.nleaves == 0) return 0;
	const uint64_t ch = std::min<uint64_t>(CH, nchunks);
	const bool want = dst || cb;
	h->xstage.ensure(ch * XCHUNK); h->xnb.ensure(ch); h->xpack.ensure(ch * XCHUNK + 16); h->xoff.ensure(ch + 1);
	if (!h->xhost[0]) {
		for (int i = 0; i < 2; ++i) {
			HIPCHK(hipHostMalloc((void**)&h->xhost[i], CH * XCHUNK + 64, hipHostMallocDefault));
			HIPCHK(hipHostMalloc((void**)&h->xtot[i], 64, hipHostMallocDefault));
			HIPCHK(hipEventCreateWithFlags(&h->xev[i], hipEventDisableTiming));
		}
	}
	auto launch = [&](uint64_t c0, int buf) {
		const uint64_t nc = std::min<uint64_t>(CH, nchunks - c0);
		hipLaunchKernelGGL(k_export, dim3(cdiv(nc, MW)), dim3(256), 0, h->st, h->pool[h->pside].view(), d.leaf0, d.n, c0, (uint32_t)nc, h->xstage.p, h->xnb.p);
		hipLaunchKernelGGL(k_xscan, dim3(1), dim3(SCHUNK), 0, h->st, (const uint16_t*)h->xnb.p, (uint32_t)nc, h->xoff.p);
		HIPCHK(hipMemcpyAsync(h->xtot[buf], h->xoff.p + nc, 8, hipMemcpyDeviceToHost, h->st));
		if (want) {
			hipLaunchKernelGGL(k_xcompact, dim3(cdiv(nc, MW)), dim3(256), 0, h->st, (const uint8_t*)h->xstage.p, (const uint16_t*)h->xnb.p, (const uint64_t*)h->xoff.p, (uint32_t)nc, h->xpack.p);
			// the packed size is only known on the device: copy the upper bound the slots could hold, in one piece
			HIPCHK(hipMemcpyAsync(h->xhost[buf], h->xpack.p, nc * XCHUNK, hipMemcpyDeviceToHost, h->st));
		}
		HIPCHK(hipGetLastError());
		HIPCHK(hipEventRecord(h->xev[buf], h->st));
	};
	int64_t k = 0;
	int buf = 0;
	launch(0, 0);
	for (uint64_t c0 = 0; c0 < nchunks; c0 += CH, buf ^= 1) {
		if (c0 + CH < nchunks) launch(c0 + CH, buf ^ 1);
		HIPCHK(hipEventSynchronize(h->xev[buf]));
		const int64_t n = (int64_t)*h->xtot[buf];
		if (dst) memcpy(dst + k, h->xhost[buf], (size_t)n);
		if (cb && n) cb(user, h->xhost[buf], n);
		k += n;
	}
	return k;
}

int64_t rb2_hip_stream_rope(rb2_hip_t *h, int b, rb2_hip_run_cb cb, void *user)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	ensure_dense(h);
	int64_t k = 0;
	for (int r = 0; r < NR; ++r) if (rope_sym(r) == b) k += export_piece(h, r, nullptr, cb, user);
	return k;
}

/* rope b = its pieces (b,x) in the order x = $,A,C,G,T,N (rb2_device.h) */
int64_t rb2_hip_rope_bytes(rb2_hip_t *h, int b)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	ensure_dense(h);
	int64_t t = 0;
	for (int r = 0; r < NR; ++r) if (rope_sym(r) == b) t += export_piece(h, r, nullptr);
	return t;
}

int64_t rb2_hip_download_rope(rb2_hip_t *h, int b, uint8_t *dst)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	ensure_dense(h);
	int64_t k = 0;
	for (int r = 0; r < NR; ++r) if (rope_sym(r) == b) k += export_piece(h, r, dst + k);
	return k;
}

void rb2_hip_load_ropes(rb2_hip_t *h, const uint8_t *const rle[6], const int64_t n_bytes[6])
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	hipStream_t st = h->st;
	// The run bytes go to the device as they are and are decoded there (k_ld_*, rb2_kernels.h): at configs[4] size -- an index of
	// 500 M reads, 35 GB of runs -- a host loop over the runs is minutes of one core (the first version: 17 s for 5 G symbols).
	DevBuf<uint8_t> dr[6]; DevBuf<uint64_t> blk[6];
	DevBuf<unsigned long long> cnt;                            // [0, 36): symbol totals of the six ropes; [36, 36 + NR * 6): per piece
	DevBuf<uint32_t> flag;                                     // [0] bad input, [1] long runs
	const uint32_t long_cap = 1u << 20;
	DevBuf<LdLong> longs;
	cnt.ensure(36 + NR * 6); flag.ensure(2); longs.ensure(long_cap);
	HIPCHK(hipMemsetAsync(cnt.p, 0, (36 + NR * 6) * 8, st));
	HIPCHK(hipMemsetAsync(flag.p, 0, 8, st));
	uint64_t nblk[6];
	for (int b = 0; b < 6; ++b) {
		const uint64_t nb = rle[b] && n_bytes[b] > 0 ? (uint64_t)n_bytes[b] : 0;
		nblk[b] = cdiv(nb, LDB);
		if (nb == 0) continue;
		dr[b].ensure(nb + 16); blk[b].ensure(nblk[b] + 1);
		HIPCHK(hipMemcpyAsync(dr[b].p, rle[b], nb, hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_ld_count, dim3((unsigned)nblk[b]), dim3(256), 0, st, (const uint8_t*)dr[b].p, nb, blk[b].p, cnt.p + b * 6, flag.p);
	}
	unsigned long long tot_h[36]; uint32_t flag_h[2];
	HIPCHK(hipMemcpyAsync(tot_h, cnt.p, sizeof(tot_h), hipMemcpyDeviceToHost, st));
	HIPCHK(hipMemcpyAsync(flag_h, flag.p, 8, hipMemcpyDeviceToHost, st));
	HIPCHK(hipStreamSynchronize(st));
	if (flag_h[0]) { rb2_fatal("[rb2_hip] load_ropes: not run-length bytes of ropebwt2's codec (bad symbol or truncated run)\n"); }
	// piece (b,x) of rope b has as many rows as rope x has b's; rope $ is one piece
	uint64_t tot[6][6];
	for (int b = 0; b < 6; ++b) for (int a = 0; a < 6; ++a) tot[b][a] = tot_h[b * 6 + a];
	RopeDesc rp[NR];
	LdPieces tab[6];
	uint64_t leaf = 0;
	for (int r = 0; r < NR; ++r) memset(&rp[r], 0, sizeof(RopeDesc));
	for (int b = 0; b < 6; ++b) {
		LdPieces &t = tab[b];
		memset(&t, 0, sizeof(t));
		t.np = b == 0 ? 1 : 6;
		uint64_t have = 0, want = 0;
		for (int a = 0; a < 6; ++a) have += tot[b][a];
		for (int x = 0; x < t.np; ++x) {
			const int r = b == 0 ? 0 : rope_of(b, x);
			uint64_t quota = 0;
			if (b == 0) quota = have; else quota = tot[x][b];
			const bool keep = h->nranks == 1 || h->owner[r] == h->rank;   // sharded: other ranks' pieces are only counted
			RopeDesc &d = rp[r];
			d.leaf0 = leaf; d.sb0 = leaf / SB;
			d.n = keep ? quota : 0;
			d.nleaves = keep ? (quota + LEAF - 1) / LEAF : 0;
			t.q[x] = want; t.word0[x] = leaf * LEAFW; t.keep[x] = keep; t.r[x] = r;
			want += quota;
			leaf += (d.nleaves + SB - 1) / SB * SB;
		}
		t.q[t.np] = want;
		if (have < want) { rb2_fatal("[rb2_hip] load_ropes: rope %d is shorter than the symbol counts of the other ropes imply\n", b); }
		if (have > want) { rb2_fatal("[rb2_hip] load_ropes: rope %d is longer than the symbol counts of the other ropes imply (not a BWT of complete strings?)\n", b); }
	}
	const int sd = h->side, ps = h->pside;
	h->sparse = false; h->sp_backoff = h->sp_penalty = 0;      /* what is loaded is the dense layout */
	h->pool[ps].ensure(leaf + SB, false, st);
	PoolView pv = h->pool[ps].view();
	if (leaf) {
		HIPCHK(hipMemsetAsync(pv.data, 0, leaf * (uint64_t)LEAFB, st));
		HIPCHK(hipMemsetAsync(pv.own, 0, leaf * sizeof(LeafMeta), st));
	}
	std::vector<uint64_t> off;
	for (int b = 0; b < 6; ++b) {
		if (nblk[b] == 0) continue;
		off.resize(nblk[b] + 1);
		HIPCHK(hipMemcpyAsync(off.data(), blk[b].p, nblk[b] * 8, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
		uint64_t run = 0;
		for (uint64_t i = 0; i < nblk[b]; ++i) { const uint64_t v = off[i]; off[i] = run; run += v; }
		HIPCHK(hipMemcpyAsync(blk[b].p, off.data(), nblk[b] * 8, hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_ld_expand, dim3((unsigned)nblk[b]), dim3(256), 0, st, (const uint8_t*)dr[b].p, (uint64_t)n_bytes[b], (const uint64_t*)blk[b].p, tab[b],
				(uint64_t*)pv.data, cnt.p + 36, longs.p, flag.p + 1, long_cap, flag.p);
		hipLaunchKernelGGL(k_ld_long, dim3(1024), dim3(256), 0, st, (const LdLong*)longs.p, (const uint32_t*)(flag.p + 1), long_cap, (uint64_t*)pv.data);
		HIPCHK(hipMemcpyAsync(flag_h, flag.p, 8, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));                      // (off is reused; the long-run list is per rope)
		if (flag_h[0]) { rb2_fatal("[rb2_hip] load_ropes: rope %d does not match the symbol counts of the other ropes\n", b); }
		if (flag_h[1] > long_cap) { rb2_fatal("[rb2_hip] load_ropes: more than %u runs longer than %u symbols in rope %d\n", long_cap, LD_LONG, b); }
		HIPCHK(hipMemsetAsync(flag.p + 1, 0, 4, st));
	}
	unsigned long long pc_h[NR * 6];
	HIPCHK(hipMemcpyAsync(pc_h, cnt.p + 36, sizeof(pc_h), hipMemcpyDeviceToHost, st));
	for (int r = 0; r < NR; ++r)
		if (rp[r].nleaves) hipLaunchKernelGGL(k_ld_own, dim3((unsigned)cdiv(rp[r].nleaves, MW)), dim3(256), 0, st, pv, rp[r].leaf0, rp[r].nleaves, rp[r].n);
	HIPCHK(hipStreamSynchronize(st));
	for (int r = 0; r < NR; ++r) for (int a = 0; a < 6; ++a) rp[r].cnt[a] = pc_h[r * 6 + a];
	HIPCHK(hipMemcpyAsync(&h->ctl->rope[sd][0], rp, sizeof(rp), hipMemcpyHostToDevice, st));
	const uint64_t nsb = leaf / SB;
	HIPCHK(hipMemcpyAsync(&h->ctl->nsb_total, &nsb, 8, hipMemcpyHostToDevice, st));
	build_directory(h, sd, ps, nsb);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(st));
	memcpy(h->h_rope, rp, sizeof(rp));
	for (int b = 0; b < 6; ++b) { dr[b].release(); blk[b].release(); }
	cnt.release(); flag.release(); longs.release();
}

/* ---- rope sharding across GPUs ------------------------------------------- */

// this engine becomes rank `rank` of `nranks`: it holds (and launches blocks for) the sub-ropes owner[] gives it (rb2_multi.h)
void engine_set_shard(rb2_hip_t *h, int rank, int nranks, const int *owner)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	if (nranks < 1 || rank < 0 || rank >= nranks) { rb2_fatal("[rb2_hip] bad shard rank %d/%d\n", rank, nranks); }
	ensure_dense(h);
	uint32_t own[NR + 1];
	memset(own, 0, sizeof(own));
	for (int r = 0; r < NR; ++r) {
		if (owner[r] < 0 || owner[r] >= nranks) { rb2_fatal("[rb2_hip] bad owner of sub-rope %d\n", r); }
		h->owner[r] = owner[r]; own[r] = owner[r] == rank;
	}
	h->rank = rank; h->nranks = nranks;
	{ bool seen[64] = {false}; h->nactive = 0; for (int r = 1; r < NR; ++r) if (owner[r] < 64 && !seen[owner[r]]) { seen[owner[r]] = true; ++h->nactive; } if (h->nactive < 1) h->nactive = 1; }
	HIPCHK(hipMemcpyAsync(&h->ctl->own[0], own, sizeof(own), hipMemcpyHostToDevice, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
}

int rb2_hip_num_subropes(void) { return NR; }

/* run everything on the caller's stream (e.g. torch's current stream), so that the caller's own work on that stream and the
 * engine's kernels need no host synchronisation between them */
void rb2_hip_use_stream(rb2_hip_t *h, void *hip_stream)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	if (!h->own_stream && h->st == (hipStream_t)hip_stream) return;            /* bound already */
	if (h->own_stream) { HIPCHK(hipStreamSynchronize(h->st)); HIPCHK(hipStreamDestroy(h->st)); h->own_stream = false; }
	else HIPCHK(hipDeviceSynchronize());                                       /* the previous foreign stream may be gone */
	h->st = (hipStream_t)hip_stream;
}

/* where the members of (piece r -> symbol a) sit in the send buffer of rank `src`: per destination rank d,
 * for the pieces r2 = (a,b) owned by d (ascending), for the pieces r of rope b owned by src (ascending):
 * g[r][a] records.  Evaluated for another source rank it gives the layout of what arrives from it. */
static void shard_layout(const int owner[NR], int nranks, int src, const int64_t *g, int64_t off[NR][6], int64_t per_rank[], int64_t start_rank[])
{
	int64_t run = 0;
	for (int d = 0; d < nranks; ++d) {
		start_rank[d] = run;
		for (int r2 = 1; r2 < NR; ++r2) {
			if (owner[r2] != d) continue;
			const int a = rope_sym(r2), b = rope_prev(r2);
			for (int r = 0; r < NR; ++r) {
				if (rope_sym(r) != b || owner[r] != src) continue;
				off[r][a] = run; run += g[r * 6 + a];
			}
		}
		per_rank[d] = run - start_rank[d];
	}
}

/* plain copies on the handle's stream: kind 0 host->device, 1 device->host, 2 device->device */
void rb2_hip_memcpy(rb2_hip_t *h, void *dst, const void *src, int64_t bytes, int kind)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	if (bytes <= 0) return;
	HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
}

/* n rank queries against rope b in one launch: one wave per query (k_rank_batch / wave_rank_all) */
void rb2_hip_rank_batch(rb2_hip_t *h, int b, int64_t n, const int64_t *x, int64_t *out)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	if (h->nranks > 1) rb2_fatal("[rb2_hip] rank: this handle holds only its own sub-ropes of a sharded index; ask the owner of the piece\n");
	if (n <= 0) return;
	if (b < 0 || b > 5) { rb2_fatal("[rb2_hip] rank: bad rope %d\n", b); }
	const int64_t CH = 1 << 24;                                  // queries per launch (one wave each; a launch is capped at 2^32 threads)
	h->qbuf.ensure((size_t)std::min(n, CH) * 7);
	for (int64_t i0 = 0; i0 < n; i0 += CH) {
		const int64_t nc = std::min(CH, n - i0);
		HIPCHK(hipMemcpyAsync(h->qbuf.p, x + i0, (size_t)nc * 8, hipMemcpyHostToDevice, h->st));
		require_plain(h, "a rank query");
		hipLaunchKernelGGL(k_rank_batch, dim3(cdiv((uint64_t)nc, MW)), dim3(256), 0, h->st, (const Ctl*)h->ctl, h->side, h->pool[h->pside].view(), b,
				(const uint64_t*)h->qbuf.p, (uint64_t)nc, h->qbuf.p + nc, (int)h->sparse);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(out + i0 * 6, h->qbuf.p + nc, (size_t)nc * 48, hipMemcpyDeviceToHost, h->st));
		HIPCHK(hipStreamSynchronize(h->st));
	}
}

void rb2_hip_rank1a(rb2_hip_t *h, int b, int64_t x, int64_t cx[6])
{ finish_pending(h);
	if (x < 0) x = 0;
	rb2_hip_rank_batch(h, b, 1, &x, cx);
}

/* checksum of sub-rope r (k_piece_hash); the handle must hold the piece in the dense layout */
static uint64_t piece_hash(rb2_hip_t *h, int r)
{
	HIPCHK(hipSetDevice(h->dev));
	ensure_dense(h);
	if (h->h_rope[r].n == 0) return 0;
	h->qbuf.ensure(8);
	HIPCHK(hipMemsetAsync(h->qbuf.p, 0, 8, h->st));
	require_plain(h, "a checksum");
	hipLaunchKernelGGL(k_piece_hash, dim3(2048), dim3(256), 0, h->st, (const Ctl*)h->ctl, h->side, h->pool[h->pside].view(), r, (unsigned long long*)h->qbuf.p);
	HIPCHK(hipGetLastError());
	uint64_t v = 0;
	HIPCHK(hipMemcpyAsync(&v, h->qbuf.p, 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	return v;
}
static uint64_t hash_mix(uint64_t acc, uint64_t piece, uint64_t n)      /* pieces of a rope, in order */
{
	acc = (acc ^ piece) * 0x9E3779B97F4A7C15ull; acc ^= acc >> 29;
	return (acc ^ n) * 0xBF58476D1CE4E5B9ull;
}
uint64_t rb2_hip_rope_hash(rb2_hip_t *h, int b)
{ finish_pending(h);
	uint64_t acc = 0;
	if (h->nranks > 1) { rb2_fatal("[rb2_hip] rope_hash: this handle holds only its own sub-ropes of a sharded index (use rb2_hip_multi_rope_hash)\n"); }
	for (int r = 0; r < NR; ++r) if (rope_sym(r) == b) acc = hash_mix(acc, piece_hash(h, r), h->h_rope[r].n);
	return acc;
}

/* rank inside one sub-rope (rb2_multi.h: the pieces of a sharded rope live on different handles) */
static void rank_piece(rb2_hip_t *h, int r, int64_t p, int64_t out[6])
{
	HIPCHK(hipSetDevice(h->dev));
	h->qbuf.ensure(8);
	require_plain(h, "a rank query");
	hipLaunchKernelGGL(k_rank_piece, dim3(1), dim3(64), 0, h->st, (const Ctl*)h->ctl, h->side, h->pool[h->pside].view(), r, (uint64_t)p, h->qbuf.p, (int)h->sparse);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(out, h->qbuf.p, 48, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
}

void *rb2_hip_dev_alloc(rb2_hip_t *h, int64_t bytes)
{
	HIPCHK(hipSetDevice(h->dev));
	void *p = nullptr;
	HIPCHK(hipMalloc(&p, (size_t)bytes));
	return p;
}

void rb2_hip_dev_free(rb2_hip_t *h, void *p) { finish_pending(h); HIPCHK(hipSetDevice(h->dev)); HIPCHK(hipFree(p)); }

void rb2_hip_synth_reads_skew(rb2_hip_t *h, uint8_t *dst_dev, int64_t first_read, int64_t n_reads, int read_len, uint64_t seed, int strand, int64_t genome_len, int skew)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	const uint64_t total = (uint64_t)n_reads * (read_len + 1) * (strand ? 2 : 1);
	if (total == 0) return;
	if ((uintptr_t)dst_dev & 15) { rb2_fatal("[rb2_hip] synth_reads: destination must be 16-byte aligned\n"); }
	if (genome_len != 0 && genome_len < read_len) { rb2_fatal("[rb2_hip] synth_reads: genome shorter than a read\n"); }
	hipLaunchKernelGGL(k_synth, dim3(cdiv(total, 256 * 16)), dim3(256), 0, h->st, dst_dev, (uint64_t)first_read, (uint64_t)n_reads, (uint32_t)read_len, seed, strand, (uint64_t)genome_len, skew);
	HIPCHK(hipGetLastError());
}

void rb2_hip_synth_reads_cov(rb2_hip_t *h, uint8_t *dst_dev, int64_t first_read, int64_t n_reads, int read_len, uint64_t seed, int strand, int64_t genome_len)
{
	rb2_hip_synth_reads_skew(h, dst_dev, first_read, n_reads, read_len, seed, strand, genome_len, 0);
}

void rb2_hip_synth_reads(rb2_hip_t *h, uint8_t *dst_dev, int64_t first_read, int64_t n_reads, int read_len, uint64_t seed, int strand)
{
	rb2_hip_synth_reads_cov(h, dst_dev, first_read, n_reads, read_len, seed, strand, 0);
}

void rb2_hip_reserve(rb2_hip_t *h, int64_t batch_bytes, int64_t batch_strings, int64_t total_symbols)
{ finish_pending(h);
	HIPCHK(hipSetDevice(h->dev));
	if (batch_bytes > 0) h->zblk.ensure(cdiv((uint64_t)batch_bytes, ZBLOCK) + 2);
	if (batch_strings > 0) ensure_strings(h, (uint64_t)batch_strings);
	if (total_symbols > 0) {
		const uint64_t leaves = (uint64_t)total_symbols / LEAF + NR * (SB + 1);
		size_t fr = 0, tot = 0;                                    // a hint must not be what runs the device out of memory
		if (hipMemGetInfo(&fr, &tot) == hipSuccess && 2.0 * leaves * (LEAFB + 2.0 * sizeof(LeafMeta) + 2.0) > 0.6 * (double)fr) return;
		h->pool[h->pside].ensure(leaves, true, h->st);
		h->pool[h->pside ^ 1].ensure(leaves, false, h->st);
		h->LD.ensure(leaves + NR + 16);
		if (!h->sparse) h->sbtot.ensure(leaves / SB + 1);
	}
}

void rb2_hip_sparse_stats(rb2_hip_t *h, int64_t out[4])
{ finish_pending(h);
	out[0] = h->n_relayout; out[1] = h->n_void; out[2] = h->n_sparse_rounds; out[3] = h->sparse ? 1 : 0;
}

void rb2_hip_layout_stats(rb2_hip_t *h, int64_t out[8])
{ finish_pending(h);
	uint64_t ns = 0;
	HIPCHK(hipSetDevice(h->dev));
	HIPCHK(hipMemcpyAsync(&ns, &h->ctl->nsplit_total, 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	out[0] = h->n_relayout; out[1] = h->n_void; out[2] = h->n_sparse_rounds; out[3] = h->sparse ? 1 : 0;
	out[4] = h->n_respread; out[5] = (int64_t)ns; out[6] = h->n_grow_in_rounds; out[7] = h->n_plain_handover;
}

void rb2_hip_window_stats(rb2_hip_t *h, int64_t out[6])
{ finish_pending(h);
	unsigned long long w[4] = {0, 0, 0, 0};
	HIPCHK(hipSetDevice(h->dev));
	HIPCHK(hipMemcpyAsync(w, &h->ctl->wfmt[0], sizeof(w), hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	for (int i = 0; i < 4; ++i) out[i] = (int64_t)w[i];
	out[4] = h->n_compact_rounds; out[5] = h->compact_stats ? 1 : 0;
}

void rb2_hip_sync(rb2_hip_t *h) { finish_pending(h); HIPCHK(hipSetDevice(h->dev)); HIPCHK(hipStreamSynchronize(h->st)); }

void rb2_hip_profile(rb2_hip_t *h, int enable) { finish_pending(h); h->prof = enable; }

void rb2_hip_profile_get(rb2_hip_t *h, int64_t launches[RB2_K_COUNT], double ms[RB2_K_COUNT], int64_t units[RB2_K_COUNT], int reset)
{ finish_pending(h);
	for (int k = 0; k < RB2_K_COUNT; ++k) { launches[k] = h->p_launch[k]; ms[k] = h->p_ms[k]; units[k] = h->p_units[k]; }
	if (reset) { memset(h->p_launch, 0, sizeof(h->p_launch)); memset(h->p_ms, 0, sizeof(h->p_ms)); memset(h->p_units, 0, sizeof(h->p_units)); }
}

const char *rb2_hip_kernel_name(int k)
{
	static const char *nm[RB2_K_COUNT] = {"k_sym", "k_tscan", "k_prep", "k_part", "k_merge", "k_meta", "k_advance", "k_init", "k_relayout", "k_split"};
	return (k >= 0 && k < RB2_K_COUNT) ? nm[k] : "?";
}

void rb2_hip_layout(int *leaf_syms, int *tile_leaves, int *string_tile)
{
	if (leaf_syms) *leaf_syms = LEAF;
	if (tile_leaves) *tile_leaves = WPL;          /* leaves per merge window */
	if (string_tile) *string_tile = STILE;
}

} // extern "C"

#include "rb2_multi.h"
