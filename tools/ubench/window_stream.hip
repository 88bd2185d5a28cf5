// window_stream.hip -- which lines of a 1536-byte window (4 leaves x 3 planes x 128 bytes) can a dense rewrite leave alone and gain by it?
// A copy pool -> pool, one wave per window, lane = group (as k_merge reads and writes), nontemporal whole-line accesses:
//   0  all 12 lines                                                    (today's 3-plane leaves)
//   1  leaf-major, 10 of 12: planes 0, 1 of the four leaves + lines 2 and 5 (an exception block where planes 2 of leaves 0, 1 lie); lines 8, 11 untouched
//   2  window-major, the first 10 lines of every 12 (planes 0, 1 of the four leaves back to back, then a 256-byte block), the last 2 untouched
//   3  dense stride: 10 lines per window, windows back to back (1280 bytes)
//   4  leaf-major, 9 of 12 (planes 0, 1 + line 2 only)
//   5  dense stride 8 lines (two planes, nothing else: the floor)
// hipcc --offload-arch=gfx950 -O3 -o window_stream window_stream.hip && ./window_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k_copy(const uint64_t *src, uint64_t *dst, uint64_t nwin)
{
	const uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (w >= nwin) return;
	const int ln = threadIdx.x & 63, lf = ln >> 4, g = ln & 15;
	constexpr int STRIDE = MODE == 3 ? 160 : (MODE == 5 ? 128 : 192);      // words per window
	const uint64_t *p = src + w * STRIDE;
	uint64_t *q = dst + w * STRIDE;
	int o0, o1, ox = -1;
	if (MODE == 0 || MODE == 1 || MODE == 4) { o0 = lf * 48 + g; o1 = o0 + 16; }
	else { o0 = lf * 32 + g; o1 = o0 + 16; }
	if (MODE == 0) ox = lf * 48 + 32 + g;
	if (MODE == 1 && ln < 32) ox = lf * 48 + 32 + g;
	if (MODE == 4 && ln < 16) ox = 32 + g;
	if ((MODE == 2 || MODE == 3) && ln < 32) ox = 128 + ln;
	const uint64_t a = __builtin_nontemporal_load(p + o0), b = __builtin_nontemporal_load(p + o1);
	uint64_t c = 0;
	if (ox >= 0) c = __builtin_nontemporal_load(p + ox);
	__builtin_nontemporal_store(a + 1, q + o0); __builtin_nontemporal_store(b ^ a, q + o1);
	if (ox >= 0) __builtin_nontemporal_store(c + b, q + ox);
}

int main()
{
	const uint64_t nwin = 3ull << 20;                          // 4.8 GB per side at 1536 bytes per window
	uint64_t *a, *b;
	CHK(hipMalloc(&a, nwin * 1536)); CHK(hipMalloc(&b, nwin * 1536));
	CHK(hipMemset(a, 1, nwin * 1536)); CHK(hipMemset(b, 2, nwin * 1536));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int lines[6] = { 12, 10, 10, 10, 9, 8 };
	const char *name[6] = { "all 12 lines", "leaf-major 10 of 12 (skip 8, 11)", "window-major first 10 of 12", "dense stride 10 lines", "leaf-major 9 of 12", "dense stride 8 lines" };
	for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 6; ++mode) {
		const unsigned grid = (unsigned)((nwin + 3) / 4);
		auto run = [&] {
			switch (mode) {
			case 0: hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, a, b, nwin); break;
			case 1: hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, nwin); break;
			case 2: hipLaunchKernelGGL(k_copy<2>, dim3(grid), dim3(256), 0, 0, a, b, nwin); break;
			case 3: hipLaunchKernelGGL(k_copy<3>, dim3(grid), dim3(256), 0, 0, a, b, nwin); break;
			case 4: hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, nwin); break;
			default: hipLaunchKernelGGL(k_copy<5>, dim3(grid), dim3(256), 0, 0, a, b, nwin); break;
			}
		};
		run();
		CHK(hipEventRecord(e0));
		for (int i = 0; i < 10; ++i) run();
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
		printf("%-36s %.3f ms per pass over %.1f M windows = %.2f TB/s of touched bytes; relative to all 12 lines at the same rate: %.2f\n", name[mode], ms, nwin / 1048576.0,
			2.0 * nwin * lines[mode] * 128 / ms / 1e9, lines[mode] / 12.0);
	}
	return 0;
}
