// diag/chain_time.hip — time per term of mpc_wave_chain_add (the in-order reduction of kernels_prog.h): one wave alone, one wave
// per SIMD, several waves per SIMD; data from HBM/L2 with the product's 8-chunk lookahead, and from registers.
#include "../muscle_amd/csrc/mpc_platform.h"
#include <cstdio>
#include <vector>
#define G 8
__global__ void __launch_bounds__(256) k_mem(const float *in, float *out, int nchunks)
{
	const unsigned lane = threadIdx.x & 63u;
	const float *src = in + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 * nchunks;
	float total = 0.0f, cur[G], nxt[G];
	for (int g = 0; g < G; ++g) cur[g] = g < nchunks ? src[g * 64 + lane] : 0.0f;
	for (int c0 = 0; c0 < nchunks; c0 += G) {
#pragma unroll
		for (int g = 0; g < G; ++g) nxt[g] = (c0 + G + g < nchunks) ? src[(c0 + G + g) * 64 + lane] : 0.0f;
#pragma unroll
		for (int g = 0; g < G; ++g) if (c0 + g < nchunks) total = mpc_wave_chain_add(total, cur[g]);
#pragma unroll
		for (int g = 0; g < G; ++g) cur[g] = nxt[g];
	}
	if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = total;
}
__global__ void __launch_bounds__(256) k_reg(float *out, int nchunks, float v)
{
	float total = 0.0f;
	for (int c = 0; c < nchunks; ++c) total = mpc_wave_chain_add(total, v);
	if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = total;
}
// the form of rounds 1-4 (wave_shr:1 of the partial sums: a DPP read of the just-written register per term), from memory
__global__ void __launch_bounds__(256) k_shift(const float *in, float *out, int nchunks)
{
	const unsigned lane = threadIdx.x & 63u;
	const float *src = in + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 * nchunks;
	float total = 0.0f, cur[G], nxt[G];
	for (int g = 0; g < G; ++g) cur[g] = g < nchunks ? src[g * 64 + lane] : 0.0f;
	for (int c0 = 0; c0 < nchunks; c0 += G) {
#pragma unroll
		for (int g = 0; g < G; ++g) nxt[g] = (c0 + G + g < nchunks) ? src[(c0 + G + g) * 64 + lane] : 0.0f;
#pragma unroll
		for (int g = 0; g < G; ++g) if (c0 + g < nchunks) total = mpc_wave_chain_add_shift(total, cur[g]);
#pragma unroll
		for (int g = 0; g < G; ++g) cur[g] = nxt[g];
	}
	if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = total;
}
// the same chain as scalar operands: 64 v_readlane (independent of the chain) + 64 dependent v_add_f32 per chunk, lane-uniform total
__device__ __forceinline__ float chain_rl(float total, float v)
{
#pragma unroll
	for (int k = 0; k < 64; ++k) total += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
	return total;
}
__global__ void __launch_bounds__(256) k_rl(const float *in, float *out, int nchunks)
{
	const unsigned lane = threadIdx.x & 63u;
	const float *src = in + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 * nchunks;
	float total = 0.0f, cur[G], nxt[G];
	for (int g = 0; g < G; ++g) cur[g] = g < nchunks ? src[g * 64 + lane] : 0.0f;
	for (int c0 = 0; c0 < nchunks; c0 += G) {
#pragma unroll
		for (int g = 0; g < G; ++g) nxt[g] = (c0 + G + g < nchunks) ? src[(c0 + G + g) * 64 + lane] : 0.0f;
#pragma unroll
		for (int g = 0; g < G; ++g) if (c0 + g < nchunks) total = chain_rl(total, cur[g]);
#pragma unroll
		for (int g = 0; g < G; ++g) cur[g] = nxt[g];
	}
	if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = total;
}
int main()
{
	const int nchunks = 4096; // 262144 terms per wave
	float *din, *dout;
	const size_t maxwaves = 256 * 8 * 4;
	hipMalloc(&din, maxwaves * 64 * nchunks * 4); hipMalloc(&dout, maxwaves * 4);
	hipMemset(din, 0, maxwaves * 64 * nchunks * 4);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	struct { int grid, block; const char *what; } cfg[] = { {1, 64, "one wave"}, {256, 256, "one wave per SIMD"}, {512, 256, "2 per SIMD"},
		{1024, 256, "4 per SIMD"}, {2048, 256, "8 per SIMD"} };
	for (auto &c : cfg) for (int mode = 0; mode < 4; ++mode) {
		float ms = 0;
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(a);
			if (mode == 0) hipLaunchKernelGGL(k_reg, dim3(c.grid), dim3(c.block), 0, 0, dout, nchunks, 0.5f);
			else if (mode == 3) hipLaunchKernelGGL(k_shift, dim3(c.grid), dim3(c.block), 0, 0, din, dout, nchunks);
			else if (mode == 2) hipLaunchKernelGGL(k_rl, dim3(c.grid), dim3(c.block), 0, 0, din, dout, nchunks);
			else hipLaunchKernelGGL(k_mem, dim3(c.grid), dim3(c.block), 0, 0, din, dout, nchunks);
			hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
		}
		const double terms = 64.0 * nchunks;
		printf("%-20s %s: %.3f ms, %.1f ns per term per wave = %.1f cycles at 2.4 GHz\n", c.what, mode == 3 ? "shift (rounds 1-4)" : mode == 2 ? "readlane" : mode ? "memory  " : "register", ms,
			ms * 1e6 / terms, ms * 1e6 / terms * 2.4);
	}
	return 0;
}
