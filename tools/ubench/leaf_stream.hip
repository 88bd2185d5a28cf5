// leaf_stream.hip -- what a dense rewrite would gain from leaves that use 320 of their 384 bytes: a copy pool -> pool, one lane per group
// (three plane words, plane-major, as k_merge reads and writes them), (a) all 48 words of every leaf, (b) words 0..39 only (two planes + a
// 64-byte block in the first half of the third line; the second half of that line is never touched).  Do the memory side's requests follow
// the bytes (64-byte sectors) or the 128-byte lines?   hipcc --offload-arch=gfx950 -O3 -o leaf_stream leaf_stream.hip && ./leaf_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_copy(const uint64_t *src, uint64_t *dst, uint64_t nleaves)
{
	const uint64_t gw = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int ln = threadIdx.x & 63, g = ln & 15;
	const uint64_t leaf = gw * 4 + (ln >> 4);
	if (leaf >= nleaves) return;
	const uint64_t *p = src + leaf * 48 + g;
	uint64_t *q = dst + leaf * 48 + g;
	const uint64_t a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 16);
	uint64_t c = 0;
	if (MODE == 0 || g < 8) c = __builtin_nontemporal_load(p + 32);
	__builtin_nontemporal_store(a + 1, q); __builtin_nontemporal_store(b ^ a, q + 16);
	if (MODE == 0 || g < 8) __builtin_nontemporal_store(c + b, q + 32);
}

int main()
{
	const uint64_t nleaves = 12ull << 20;                      // 4.5 GB per side
	uint64_t *a, *b;
	CHK(hipMalloc(&a, nleaves * 384)); CHK(hipMalloc(&b, nleaves * 384));
	CHK(hipMemset(a, 1, nleaves * 384)); CHK(hipMemset(b, 2, nleaves * 384));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
		const unsigned grid = (unsigned)((nleaves + 15) / 16);
		auto run = [&] { if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, a, b, nleaves); else hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, nleaves); };
		run();
		CHK(hipEventRecord(e0));
		for (int i = 0; i < 10; ++i) run();
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
		printf("%s: %.3f ms per pass over %.2f GB of leaf slots (read + write) = %.2f TB/s of slot bytes, %.2f TB/s of touched bytes\n", mode ? "320 of 384 bytes" : "all 384 bytes   ", ms,
			2.0 * nleaves * 384 / 1e9, 2.0 * nleaves * 384 / ms / 1e9, 2.0 * nleaves * (mode ? 320 : 384) / ms / 1e9);
	}
	return 0;
}
