// micro-benchmark: issue cost of a few VALU instructions on gfx950 (wave64), relative to v_add_u32.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 4096
template <int OP> __global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t s)
{
	uint32_t a = threadIdx.x + s, b = threadIdx.x * 3 + 1, c = s | 1, d = threadIdx.x ^ s;
	uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d;
	for (int i = 0; i < N_IT; ++i) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (OP == 0) { a += b; b += c; c += d; d += a; }                                   // 4 x v_add_u32
			if (OP == 1) { x = x << (a & 63); y = y << (b & 63); a += 1; b += 1; }             // 2 x v_lshlrev_b64 (+2 add)
			if (OP == 2) { a = a * b; b = b * c; c = c * d; d = d * a; }                       // 4 x v_mul_lo_u32
			if (OP == 3) { a = __popc(b) + a; b = __popc(c) + b; c = __popc(d) + c; d = __popc(a) + d; }   // 4 x v_bcnt
			if (OP == 4) { x += y; y += x; a += 1; b += 1; }                                   // 2 x 64-bit add (+2 add)
			if (OP == 5) { x = x >> 4; y = y >> 2; x ^= y; y += 1; }                          // const 64-bit shifts
			if (OP == 6) { a = __builtin_amdgcn_update_dpp(0, a, 0x111, 0xf, 0xf, false) + a; b = __builtin_amdgcn_update_dpp(0, b, 0x112, 0xf, 0xf, false) + b;
			               c = __builtin_amdgcn_update_dpp(0, c, 0x114, 0xf, 0xf, false) + c; d = __builtin_amdgcn_update_dpp(0, d, 0x118, 0xf, 0xf, false) + d; }
			if (OP == 7) { a = (a & b) | (~a & c); b = (b & c) | (~b & d); c = (c & d) | (~c & a); d = (d & a) | (~d & b); }   // 4 x v_bfi
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = x + y + a + b + c + d;
}
template <int OP> void run(const char *name, uint64_t *out)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 32;          // 32 blocks (x4 waves) per CU
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1u);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3u);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double groups = (double)blocks * 4 * N_IT * 8;       // per wave: N_IT*8 groups of ~4 instructions
	printf("%-28s %8.3f ms  -> %.3f ns per (wave, group of 4 instr) per chip; %.2f G wave-instr/s (counting 4)\n", name, ms, ms * 1e6 / groups, groups * 4 / (ms * 1e-3) / 1e9);
}
int main()
{
	uint64_t *out; hipMalloc(&out, 256 * 32 * 256 * 8);
	run<0>("4x v_add_u32", out); run<1>("2x lshl_b64 var + 2 add", out); run<2>("4x v_mul_lo_u32", out); run<3>("4x v_bcnt+add", out);
	run<4>("2x add64 + 2 add", out); run<5>("2 shr64 const + xor64 + add64", out); run<6>("4x dpp add", out); run<7>("4x bfi-like", out);
	return 0;
}
