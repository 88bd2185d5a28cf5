// elem_stream.hip -- what do the string-tile kernels' access shapes cost?  n strings, per string: read 1 byte (A) + 4 bytes (L), write 1 byte + 4 bytes
// + 1 byte (the fused k_sym), with
//   0  one string per lane and instruction (byte / dword loads and stores, 8 strings per lane in a wave-per-tile kernel: x = c * 64 + lane)
//   1  two strings per thread, 256-thread blocks (x = h * 256 + thread): the block kernels
//   2  four CONSECUTIVE strings per lane (dword loads of the bytes, 16-byte loads of the dwords)
//   3  sixteen consecutive strings per lane (16-byte loads of the bytes, 4 x 16-byte loads of the dwords)
// hipcc --offload-arch=gfx950 -O3 -o elem_stream elem_stream.hip && ./elem_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k0(const uint8_t *A, const uint32_t *L, uint8_t *A2, uint32_t *E, uint8_t *IA, uint64_t n)
{
	const uint64_t base = (uint64_t)blockIdx.x * 512;
	const int ln = threadIdx.x;
	uint32_t a[8], l[8];
#pragma unroll
	for (int c = 0; c < 8; ++c) { const uint64_t k = base + c * 64 + ln; a[c] = 0; l[c] = 0; if (k < n) { a[c] = A[k]; l[c] = L[k]; } }
#pragma unroll
	for (int c = 0; c < 8; ++c) { const uint64_t k = base + c * 64 + ln; if (k < n) { A2[k] = (uint8_t)(a[c] | 0x80); E[k] = l[c] - (uint32_t)k; IA[k] = (uint8_t)(a[c] & 7); } }
}
__global__ __launch_bounds__(256) void k1(const uint8_t *A, const uint32_t *L, uint8_t *A2, uint32_t *E, uint8_t *IA, uint64_t n)
{
	const uint64_t base = (uint64_t)blockIdx.x * 512;
	uint32_t a[2], l[2];
#pragma unroll
	for (int h = 0; h < 2; ++h) { const uint64_t k = base + h * 256 + threadIdx.x; a[h] = 0; l[h] = 0; if (k < n) { a[h] = A[k]; l[h] = L[k]; } }
	__syncthreads();
#pragma unroll
	for (int h = 0; h < 2; ++h) { const uint64_t k = base + h * 256 + threadIdx.x; if (k < n) { A2[k] = (uint8_t)(a[h] | 0x80); E[k] = l[h] - (uint32_t)k; IA[k] = (uint8_t)(a[h] & 7); } }
}
__global__ __launch_bounds__(256) void k2(const uint8_t *A, const uint32_t *L, uint8_t *A2, uint32_t *E, uint8_t *IA, uint64_t n)
{
	const uint64_t k = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
	if (k + 4 > n) return;
	const uint32_t a = *(const uint32_t*)(A + k);
	const uint4 l = *(const uint4*)(L + k);
	*(uint32_t*)(A2 + k) = a | 0x80808080u;
	*(uint4*)(E + k) = make_uint4(l.x - (uint32_t)k, l.y - (uint32_t)k - 1, l.z - (uint32_t)k - 2, l.w - (uint32_t)k - 3);
	*(uint32_t*)(IA + k) = a & 0x07070707u;
}
__global__ __launch_bounds__(256) void k3(const uint8_t *A, const uint32_t *L, uint8_t *A2, uint32_t *E, uint8_t *IA, uint64_t n)
{
	const uint64_t k = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
	if (k + 16 > n) return;
	const uint4 a = *(const uint4*)(A + k);
	uint4 l[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) l[i] = *(const uint4*)(L + k + 4 * i);
	*(uint4*)(A2 + k) = make_uint4(a.x | 0x80808080u, a.y | 0x80808080u, a.z | 0x80808080u, a.w | 0x80808080u);
#pragma unroll
	for (int i = 0; i < 4; ++i) *(uint4*)(E + k + 4 * i) = make_uint4(l[i].x - (uint32_t)k, l[i].y - 1, l[i].z - 2, l[i].w - 3);
	*(uint4*)(IA + k) = make_uint4(a.x & 0x07070707u, a.y & 0x07070707u, a.z & 0x07070707u, a.w & 0x07070707u);
}

// 4: persistent waves -- 256-thread blocks of four independent waves, a wave walks tiles w, w + W, ... (8 strings per lane), the next tile's loads
//    issued before the current tile's stores
__global__ __launch_bounds__(256) void k4(const uint8_t *A, const uint32_t *L, uint8_t *A2, uint32_t *E, uint8_t *IA, uint64_t n)
{
	const int ln = threadIdx.x & 63;
	const uint64_t W = (uint64_t)gridDim.x * 4, ntile = (n + 511) / 512;
	uint64_t tile = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	uint32_t a[8], l[8], an[8], lx[8];
	auto load = [&](uint64_t t, uint32_t *aa, uint32_t *ll) {
#pragma unroll
		for (int c = 0; c < 8; ++c) { const uint64_t k = t * 512 + c * 64 + ln; aa[c] = 0; ll[c] = 0; if (k < n) { aa[c] = A[k]; ll[c] = L[k]; } }
	};
	if (tile >= ntile) return;
	load(tile, a, l);
	for (;;) {
		const uint64_t nt = tile + W;
		const bool more = nt < ntile;
		if (more) load(nt, an, lx);
#pragma unroll
		for (int c = 0; c < 8; ++c) { const uint64_t k = tile * 512 + c * 64 + ln; if (k < n) { A2[k] = (uint8_t)(a[c] | 0x80); E[k] = l[c] - (uint32_t)k; IA[k] = (uint8_t)(a[c] & 7); } }
		if (!more) return;
#pragma unroll
		for (int c = 0; c < 8; ++c) { a[c] = an[c]; l[c] = lx[c]; }
		tile = nt;
	}
}

int main()
{
	const uint64_t n = 40ull << 20;
	uint8_t *A, *A2, *IA; uint32_t *L, *E;
	CHK(hipMalloc(&A, n)); CHK(hipMalloc(&A2, n)); CHK(hipMalloc(&IA, n)); CHK(hipMalloc(&L, n * 4)); CHK(hipMalloc(&E, n * 4));
	CHK(hipMemset(A, 1, n)); CHK(hipMemset(L, 1, n * 4));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const char *name[5] = { "wave per 512 strings, 8 per lane, byte/dword accesses", "256-thread block per 512 strings, 2 per thread", "4 consecutive strings per lane (dword / 16-byte accesses)", "16 consecutive strings per lane", "persistent waves (2048 blocks x 4), 8 per lane, next tile prefetched" };
	for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 5; ++mode) {
		auto run = [&] {
			if (mode == 0) hipLaunchKernelGGL(k0, dim3((unsigned)((n + 511) / 512)), dim3(64), 0, 0, A, L, A2, E, IA, n);
			else if (mode == 1) hipLaunchKernelGGL(k1, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, A, L, A2, E, IA, n);
			else if (mode == 2) hipLaunchKernelGGL(k2, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, A, L, A2, E, IA, n);
			else if (mode == 4) hipLaunchKernelGGL(k4, dim3(2048), dim3(256), 0, 0, A, L, A2, E, IA, n);
			else hipLaunchKernelGGL(k3, dim3((unsigned)((n / 16 + 255) / 256)), dim3(256), 0, 0, A, L, A2, E, IA, n);
		};
		run();
		CHK(hipEventRecord(e0));
		for (int i = 0; i < 10; ++i) run();
		CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
		float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
		printf("%-62s %.4f ms for %.0f M strings x 11 bytes = %.2f TB/s\n", name[mode], ms, n / 1048576.0, 11.0 * n / ms / 1e9);
	}
	return 0;
}
