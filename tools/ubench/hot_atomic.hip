// hot_atomic.hip -- what a tile kernel of 2048 workgroups costs when every workgroup (a) does nothing, (b) appends to ONE global
// counter with a returning atomic (k_part_sparse's work list, ctl->nwork), (c) spreads the same appends over C counters, and what the
// chain "load -> dependent load -> dependent load" of one thread per lane costs on a buffer of a few hundred MB (the directory gathers
// of k_part_sparse / k_advance).  hipcc --offload-arch=gfx950 -O3 -o hot_atomic hot_atomic.hip && ./hot_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_app(uint32_t *cnt, uint32_t ncnt, uint32_t *out, uint32_t cstride = 32)
{
	__shared__ uint32_t s_base;
	if (MODE == 0) { if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x; return; }
	if (threadIdx.x == 0) s_base = atomicAdd(&cnt[(blockIdx.x % ncnt) * cstride], 300u);
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = s_base;
}

// DEPTH dependent random reads per thread, 2 threads' worth per lane when TWO (independent chains side by side)
template <int DEPTH, bool TWO> __global__ __launch_bounds__(256) void k_chain(const uint32_t *tab, uint32_t mask, uint32_t *out, uint32_t n)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	// neighbouring lanes start at neighbouring words (as inserts in ascending order do), every level jumps somewhere else
	uint32_t a = (i * 8u) & mask, b = (i * 8u + 0x9e3779u * 64u) & mask;
#pragma unroll
	for (int d = 0; d < DEPTH; ++d) {
		a = (tab[a] + (uint32_t)d) & mask;
		if (TWO) b = (tab[b] + (uint32_t)d) & mask;
	}
	out[i] = a + (TWO ? b : 0u);
}

template <class F> float timeit(F f, int reps = 200)
{
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int i = 0; i < 10; ++i) f();
	CHK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) f();
	CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
	float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
	return ms * 1000.f / reps;
}

int main()
{
	uint32_t *cnt, *out;
	CHK(hipMalloc(&cnt, 4096 * 4)); CHK(hipMemset(cnt, 0, 4096 * 4));
	CHK(hipMalloc(&out, 1 << 24));
	for (int nb : {256, 2048, 8192}) {
		printf("%5d workgroups: empty %.1f us", nb, timeit([&] { hipLaunchKernelGGL(k_app<0>, dim3(nb), dim3(256), 0, 0, cnt, 1u, out); }));
		for (uint32_t c : {1u, 4u, 16u, 64u})
			printf(" | %u counter(s) %.1f us", c, timeit([&] { hipLaunchKernelGGL(k_app<1>, dim3(nb), dim3(256), 0, 0, cnt, c, out); }));
		printf("\n      counters one word apart (same line):");
		for (uint32_t c : {4u, 16u, 64u})
			printf(" | %u counter(s) %.1f us", c, timeit([&] { hipLaunchKernelGGL(k_app<1>, dim3(nb), dim3(256), 0, 0, cnt, c, out, 1u); }));
		printf("\n");
	}
	// dependent gathers: tables of 4 M .. 512 M words (16 MB .. 2 GB), from hipMalloc and from hipMemCreate + hipMemMap (how the engine's
	// pools are mapped); entries point at random lines: 64 line requests per wave and level (20 us per level and million = 50 G lines/s)
	const uint32_t n = 1u << 20;
	printf("1 M threads, dependent random 4-byte reads (every lane a random line of its own): us per level = (depth 3 - depth 1) / 2\n");
	for (int vmm = 0; vmm < 2; ++vmm) for (uint32_t lg : {22u, 26u, 29u}) {
		const uint32_t nw = 1u << lg, mask = nw - 1;
		std::vector<uint32_t> h(nw);
		uint64_t x = 88172645463325252ull;
		for (uint32_t i = 0; i < nw; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (((uint32_t)(x >> 20) & ~63u) + (i & 7u) * 8u) & mask; }
		uint32_t *tab = nullptr; size_t bytes = (size_t)nw * 4, gran = 0;
		hipMemGenericAllocationHandle_t hd;
		if (!vmm) CHK(hipMalloc(&tab, bytes));
		else {
			hipMemAllocationProp pr = {}; pr.type = hipMemAllocationTypePinned; pr.location.type = hipMemLocationTypeDevice; pr.location.id = 0;
			CHK(hipMemGetAllocationGranularity(&gran, &pr, hipMemAllocationGranularityRecommended));
			bytes = (bytes + gran - 1) / gran * gran;
			CHK(hipMemAddressReserve((void**)&tab, bytes, gran, nullptr, 0));
			CHK(hipMemCreate(&hd, bytes, &pr, 0));
			CHK(hipMemMap(tab, bytes, 0, hd, 0));
			hipMemAccessDesc ad = {}; ad.location = pr.location; ad.flags = hipMemAccessFlagsProtReadWrite;
			CHK(hipMemSetAccess(tab, bytes, &ad, 1));
		}
		CHK(hipMemcpy(tab, h.data(), (size_t)nw * 4, hipMemcpyHostToDevice));
		const float d1 = timeit([&] { hipLaunchKernelGGL((k_chain<1, false>), dim3(n / 256), dim3(256), 0, 0, tab, mask, out, n); });
		const float d3 = timeit([&] { hipLaunchKernelGGL((k_chain<3, false>), dim3(n / 256), dim3(256), 0, 0, tab, mask, out, n); });
		const float t3 = timeit([&] { hipLaunchKernelGGL((k_chain<3, true>), dim3(n / 512), dim3(256), 0, 0, tab, mask, out, n / 2); });
		printf("  %s %5zu MB (granularity %zu KB): depth 1 %.1f us, depth 3 %.1f us -> %.1f us per level; two chains per thread, half the threads: depth 3 %.1f us\n",
			vmm ? "hipMemMap" : "hipMalloc", (size_t)nw * 4 >> 20, gran >> 10, d1, d3, (d3 - d1) / 2, t3);
		if (!vmm) CHK(hipFree(tab)); else { CHK(hipMemUnmap(tab, bytes)); CHK(hipMemRelease(hd)); CHK(hipMemAddressFree(tab, bytes)); }
	}
	return 0;
}
