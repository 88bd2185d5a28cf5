// leaf_rw.hip -- what bounds an in-place round: N random leaves of G groups (3 plane words each, plane-major) are read whole and the upper half
// is written back; one leaf per G lanes.  Prints time and bytes/s for G = 4..32 so that one can see whether the memory system charges
// per byte or per request.   hipcc --offload-arch=gfx950 -O3 -o leaf_rw leaf_rw.hip && ./leaf_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int G, int ATOM> __global__ __launch_bounds__(256) void k(uint64_t *pool, const uint32_t *idx, uint32_t n, uint32_t *dir, uint32_t *tot)
{
	const uint32_t lane = threadIdx.x & 63, per = 64 / G, g = lane % G;
	const uint32_t nw = gridDim.x * 4;
	for (uint32_t q = (blockIdx.x * 4 + (threadIdx.x >> 6)) * per + lane / G; q < n; q += nw * per) {
		const uint32_t leaf = idx[q];
		uint64_t *p = pool + (uint64_t)leaf * 3 * G + g;
		uint64_t a = p[0], b = p[G], c = p[2 * G];
		if (ATOM == 1 && g < 10) atomicAdd(&dir[(uint64_t)(leaf / 32) * 256 + (g < 7 ? g * 16 + (leaf & 31) / 2 : 200 + g)], 1u);
		if (ATOM == 3) {                                            // what one inserted symbol a costs: its leaf's fill, its own count, the superblock total (a separate small array)
			const uint32_t sym = leaf % 5;
			if (g == 0) atomicAdd(&dir[(uint64_t)(leaf / 32) * 128 + (leaf & 31) / 2], 1u);
			if (g == 1) atomicAdd(&dir[(uint64_t)(leaf / 32) * 128 + (1 + sym) * 16 + (leaf & 31) / 2], 1u);
			if (g == 2) atomicAdd(&tot[(uint64_t)(leaf / 32) * 4 + sym / 2], 1u);
		}
		if (ATOM == 2) {                                            // ... without the superblock total
			const uint32_t sym = leaf % 5;
			if (g == 0) atomicAdd(&dir[(uint64_t)(leaf / 32) * 128 + (leaf & 31) / 2], 1u);
			if (g == 1) atomicAdd(&dir[(uint64_t)(leaf / 32) * 128 + (1 + sym) * 16 + (leaf & 31) / 2], 1u);
		}
		a = (a << 1) | (b >> 63); b = (b << 1) | (c >> 63); c = (c << 1) ^ a;
		if (g >= G / 2) { p[0] = a; p[G] = b; p[2 * G] = c; }
	}
}

template <int G, int ATOM> void run(uint64_t *pool, uint32_t *idx, uint32_t n, uint32_t *dir, size_t pool_bytes, uint32_t *tot = nullptr)
{
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const uint32_t nleaves = (uint32_t)(pool_bytes / (24 * G));
	std::vector<uint32_t> h(n);
	std::mt19937_64 rng(1);
	for (auto &v : h) v = (uint32_t)(rng() % nleaves);
	std::sort(h.begin(), h.end()); h.erase(std::unique(h.begin(), h.end()), h.end());
	std::shuffle(h.begin(), h.end(), rng);
	const uint32_t m = (uint32_t)h.size();
	CHK(hipMemcpy(idx, h.data(), m * 4, hipMemcpyHostToDevice));
	for (int grid : {2048, 8192}) {
		float best = 1e9;
		for (int it = 0; it < 5; ++it) {
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL((k<G, ATOM>), dim3(grid), dim3(256), 0, 0, pool, idx, m, dir, tot);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
		}
		printf("G=%2d (%4d B/leaf) atom=%d grid=%5d: %u leaves  %.1f us  %.2f TB/s (read + half written)\n", G, 24 * G, ATOM, grid, m, best * 1e3, m * 36.0 * G / (best * 1e-3) / 1e12);
	}
}

int main()
{
	const size_t pool_bytes = 6ull << 30;                      // 16 G symbols of index
	uint64_t *pool; uint32_t *idx, *dir;
	CHK(hipMalloc(&pool, pool_bytes)); CHK(hipMemset(pool, 1, pool_bytes));
	const uint32_t n = 1000000;
	CHK(hipMalloc(&idx, n * 4)); CHK(hipMalloc(&dir, (pool_bytes / 96 / 32 + 1) * 1024)); CHK(hipMemset(dir, 0, (pool_bytes / 96 / 32 + 1) * 1024));
	run<4, 0>(pool, idx, n, dir, pool_bytes); run<8, 0>(pool, idx, n, dir, pool_bytes); run<16, 0>(pool, idx, n, dir, pool_bytes); run<32, 0>(pool, idx, n, dir, pool_bytes);
	run<8, 1>(pool, idx, n, dir, pool_bytes); run<16, 1>(pool, idx, n, dir, pool_bytes);
	uint32_t *tot; CHK(hipMalloc(&tot, (pool_bytes / 384 / 32 + 1) * 16)); CHK(hipMemset(tot, 0, (pool_bytes / 384 / 32 + 1) * 16));
	run<16, 2>(pool, idx, n, dir, pool_bytes, tot); run<16, 3>(pool, idx, n, dir, pool_bytes, tot);
	return 0;
}
