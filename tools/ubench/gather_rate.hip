// gather_rate.hip -- how many 128-byte lines per clock does a CU's address unit take?  Every lane of a wave loads one dword, the lanes
// `stride` dwords apart (1: one or two lines per load instruction; 32: every lane a line of its own), out of a cache-resident array
// (2 MiB: no HBM in the picture), REP independent loads per thread and step, STEPS dependent steps (the next address depends on the loaded
// value, as in a search).  What bounds k_part: the lowest levels of 2.4 M bisections are stride-32-like loads.
// hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int REP> __global__ __launch_bounds__(256) void k_gather(const uint32_t *a, uint32_t mask, uint32_t stride, int steps, uint32_t *out)
{
	const uint32_t t = blockIdx.x * 256 + threadIdx.x;
	uint32_t pos = (t * stride) & mask, acc = 0;
	for (int s = 0; s < steps; ++s) {
		uint32_t v[REP];
#pragma unroll
		for (int r = 0; r < REP; ++r) v[r] = a[(pos + (uint32_t)r * 8191u * stride) & mask];
#pragma unroll
		for (int r = 0; r < REP; ++r) acc += v[r];
		pos = (pos + 977u * stride + (acc & 1u)) & mask;            // (a holds even numbers: the address chain is data dependent and stays put)
	}
	if (acc == 0xdeadbeefu) out[t] = acc;
}

int main()
{
	const uint32_t n = 1u << 19;                                    // 2 MiB of dwords
	uint32_t *a, *out;
	CHK(hipMalloc(&a, n * 4)); CHK(hipMalloc(&out, 1u << 24));
	CHK(hipMemset(a, 0, n * 4));
	hipDeviceProp_t pr; CHK(hipGetDeviceProperties(&pr, 0));
	const double ghz = pr.clockRate / 1e6;
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int blocks = 256 * 8 * 4, steps = 64;
	for (int rep : { 1, 7 }) for (uint32_t stride : { 1u, 2u, 4u, 8u, 16u, 32u }) {
		auto run = [&] {
			if (rep == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, a, n - 1, stride, steps, out);
			else hipLaunchKernelGGL(k_gather<7>, dim3(blocks), dim3(256), 0, 0, a, n - 1, stride, steps, out);
		};
		run();
		CHK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) run(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
		const double wave_loads = (double)blocks * 4 * steps * rep, lines = wave_loads * (stride >= 32 ? 64 : (stride * 64 + 31) / 32);
		printf("%d load(s) per step, lanes %2u dwords apart: %.3f ms, %.2f wave-loads per CU and 1000 clocks, %.3f lines per CU and clock (%d CUs at %.2f GHz)\n", rep, stride, ms,
			wave_loads / pr.multiProcessorCount / (ms * 1e-3 * ghz * 1e9) * 1000, lines / pr.multiProcessorCount / (ms * 1e-3 * ghz * 1e9), pr.multiProcessorCount, ghz);
	}
	return 0;
}
