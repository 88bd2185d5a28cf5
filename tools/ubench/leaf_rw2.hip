// leaf_rw2.hip -- what an in-place round's k_merge_leaf can cost at best, and what a deeper rank directory would add to it:
// N random 384-byte leaf slots are streamed through one DPP row each (nontemporal whole-line loads and stores, like the kernel),
//   PL = 3: all three plane lines in and out      PL = 2: planes 0 and 1 only (a leaf without $ / N needs no third plane)
// with the directory atomics of one inserted symbol:
//   ATOM = 0: none    3: fill + own count (leaf rows, u16 pairs) + superblock total (today)
//   4: fill + own count + the superblock's entry in two rows (position, count) of its hyperblock of 32 superblocks (u32)
//   6: ... + the hyperblock's entry in two rows of its block of 32 hyperblocks
//   hipcc --offload-arch=gfx950 -O3 -o leaf_rw2 leaf_rw2.hip && ./leaf_rw2 [pool GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int PL, int ATOM> __global__ __launch_bounds__(256, 7) void k(uint64_t *pool, const uint32_t *idx, uint32_t n, uint32_t *dir, uint32_t *tot, uint32_t *sbrow, uint32_t *hbrow)
{
	const uint32_t lane = threadIdx.x & 63, g = lane & 15;
	const uint32_t nw = gridDim.x * 4;
	for (uint32_t q = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + lane / 16; q < n; q += nw * 4) {
		const uint32_t leaf = idx[q];
		uint64_t *p = pool + (uint64_t)leaf * 48 + g;
		uint64_t a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 16), c = PL == 3 ? __builtin_nontemporal_load(p + 32) : 0;
		const uint32_t sym = leaf % 5, sb = leaf / 32;
		if (ATOM >= 3) {
			if (g == 0) atomicAdd(&dir[(uint64_t)sb * 128 + (leaf & 31) / 2], 1u);
			if (g == 1) atomicAdd(&dir[(uint64_t)sb * 128 + (1 + sym) * 16 + (leaf & 31) / 2], 1u);
		}
		if (ATOM == 3) { if (g == 2) atomicAdd(&tot[(uint64_t)sb * 4 + sym / 2], 1u); }
		if (ATOM >= 4) {
			if (g == 2) atomicAdd(&sbrow[(uint64_t)(sb / 32) * 256 + (sb & 31)], 1u);
			if (g == 3) atomicAdd(&sbrow[(uint64_t)(sb / 32) * 256 + (1 + sym) * 32 + (sb & 31)], 1u);
		}
		if (ATOM >= 6) {
			const uint32_t hb = sb / 32;
			if (g == 4) atomicAdd(&hbrow[(uint64_t)(hb / 32) * 256 + (hb & 31)], 1u);
			if (g == 5) atomicAdd(&hbrow[(uint64_t)(hb / 32) * 256 + (1 + sym) * 32 + (hb & 31)], 1u);
		}
		a = (a << 1) | (b >> 63); b = (b << 1) | (c >> 63); c = (c << 1) ^ a;
		__builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 16);
		if (PL == 3) __builtin_nontemporal_store(c, p + 32);
	}
}

static uint64_t *pool; static uint32_t *idx, *dir, *tot, *sbrow, *hbrow; static uint32_t m;
template <int PL, int ATOM> void run()
{
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int grid : {2048, 8192}) {
		float best = 1e9;
		for (int it = 0; it < 5; ++it) {
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL((k<PL, ATOM>), dim3(grid), dim3(256), 0, 0, pool, idx, m, dir, tot, sbrow, hbrow);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
		}
		printf("planes=%d atom=%d grid=%5d: %u leaves  %.1f us  %.2f TB/s of leaf lines\n", PL, ATOM, grid, m, best * 1e3, m * 256.0 * PL / (best * 1e-3) / 1e12);
	}
}

int main(int argc, char **argv)
{
	const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 6;
	const size_t pool_bytes = gib << 30;                        // 6 GiB: 12 G symbols at 768 per slot (configs[3] at a tenth); 48 GiB: the full size
	CHK(hipMalloc(&pool, pool_bytes)); CHK(hipMemset(pool, 1, pool_bytes));
	const uint32_t n = 1000000;
	const uint32_t nleaves = (uint32_t)(pool_bytes / 384);
	const size_t nsb = nleaves / 32 + 1;
	CHK(hipMalloc(&idx, n * 4));
	CHK(hipMalloc(&dir, nsb * 512)); CHK(hipMemset(dir, 0, nsb * 512));
	CHK(hipMalloc(&tot, nsb * 16)); CHK(hipMemset(tot, 0, nsb * 16));
	CHK(hipMalloc(&sbrow, (nsb / 32 + 1) * 1024)); CHK(hipMemset(sbrow, 0, (nsb / 32 + 1) * 1024));
	CHK(hipMalloc(&hbrow, (nsb / 1024 + 1) * 1024)); CHK(hipMemset(hbrow, 0, (nsb / 1024 + 1) * 1024));
	std::vector<uint32_t> h(n);
	std::mt19937_64 rng(1);
	for (auto &v : h) v = (uint32_t)(rng() % nleaves);
	std::sort(h.begin(), h.end()); h.erase(std::unique(h.begin(), h.end()), h.end());
	// the kernel's work lists hold runs of ascending leaves (a string tile's inserts): shuffle runs of 256, not single leaves
	{
		std::vector<uint32_t> o; o.reserve(h.size());
		const size_t R = 256, nr = (h.size() + R - 1) / R;
		std::vector<size_t> perm(nr); for (size_t i = 0; i < nr; ++i) perm[i] = i;
		std::shuffle(perm.begin(), perm.end(), rng);
		for (size_t r : perm) for (size_t i = r * R; i < std::min(h.size(), (r + 1) * R); ++i) o.push_back(h[i]);
		h.swap(o);
	}
	m = (uint32_t)h.size();
	CHK(hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice));
	printf("pool %zu GiB, %u leaf slots, %zu superblocks\n", gib, nleaves, nsb);
	run<3, 0>(); run<2, 0>();
	run<3, 3>(); run<3, 4>(); run<3, 6>();
	run<2, 3>(); run<2, 4>(); run<2, 6>();
	return 0;
}
