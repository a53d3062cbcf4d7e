// diag/dma_order_test.hip — two hardware assumptions a barrier-free LDS-DMA pipeline needs (relax_stream_kernel):
//  (1) ORDER: after `s_waitcnt vmcnt(k)` the oldest (issued - k) global_load_lds transfers of the wave have landed in LDS;
//  (2) VISIBILITY: what they wrote is visible to ANOTHER wave of the workgroup that saw a flag the issuing wave wrote (ds_write) after
//      that wait — no workgroup barrier in between.
// One workgroup = 2 waves per CU-slot: wave 1 issues N transfers of 1 KiB (round r: pattern f(r, chunk, lane)), then for k = N-1 .. 0
// waits vmcnt(k) and publishes "N - k landed"; wave 0 spins on the flag and checks every chunk as soon as it is published. Rounds
// alternate two LDS areas so that stale data of the previous round is a detectable error. MODE 1: the producer waits vmcnt(0) once and
// publishes everything (only assumption 2 is exercised).
//   hipcc --offload-arch=gfx950 -O2 diag/dma_order_test.hip -o diag/dma_order_test && diag/dma_order_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define N 12          // transfers in flight per round
#define ROUNDS 2000
typedef unsigned u32;

__device__ __forceinline__ void wait_vmcnt(u32 k)
{
#define C(v) if (k == v) { asm volatile("s_waitcnt vmcnt(" #v ")" ::: "memory"); return; }
	C(11) C(10) C(9) C(8) C(7) C(6) C(5) C(4) C(3) C(2) C(1)
#undef C
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ u32 pat(u32 word) { return word * 2654435761u + 12345u; }
// where chunk c of round r of workgroup b starts (in words, a multiple of 256): odd chunks in a hot 1 MB region (cache hits), even
// chunks anywhere in the buffer (misses) — transfers of very different latency in one queue
__device__ __forceinline__ u32 chunk_word(u32 b, u32 r, u32 c, u32 total_chunks)
{
	u32 h = (b * 7919u + r) * 2246822519u + c * 3266489917u; h ^= h >> 15; h *= 2654435761u; h ^= h >> 13;
	return ((c & 1u) ? h % 1024u : h % total_chunks) * 256u;
}
__global__ void fill_kernel(u32 *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pat((u32)i); }

template <int MODE> __global__ void __launch_bounds__(128) dma_order_kernel(const u32 *src, u32 total_chunks, u32 *errors, u32 *stale)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	volatile u32 *flag = (volatile u32 *)smem; // [0]: transfers published so far (counted over all rounds), [1]: rounds the consumer finished
	unsigned char *area = smem + 64;           // 2 areas x N chunks x 1 KiB
	const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x < 2) flag[threadIdx.x] = 0;
	__syncthreads();
	if (wave == 1) {
		for (u32 r = 0; r < ROUNDS; ++r) {
			// the area of round r was last read in round r - 2: wait until the consumer is done with it
			while (r >= 2 && flag[1] < r - 1) __builtin_amdgcn_s_sleep(1);
			for (u32 c = 0; c < N; ++c)
				__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + chunk_word(blockIdx.x, r, c, total_chunks) + lane * 4),
					(__attribute__((address_space(3))) void *)(area + ((r & 1u) * N + c) * 1024), 16, 0, 0);
			if (MODE == 1) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (lane == 0) flag[0] = (r + 1) * N;
			} else {
				for (u32 k = N; k-- > 0;) {
					wait_vmcnt(k); // at most k outstanding: the oldest N - k have landed
					if (lane == 0) flag[0] = r * N + (N - k);
				}
			}
		}
	} else {
		u32 bad = 0, old = 0;
		for (u32 r = 0; r < ROUNDS; ++r) {
			for (u32 c = 0; c < N; ++c) {
				while (flag[0] < r * N + c + 1) __builtin_amdgcn_s_sleep(1);
				const uint4 got = *(const uint4 *)(area + ((r & 1u) * N + c) * 1024 + lane * 16);
				const u32 w0 = chunk_word(blockIdx.x, r, c, total_chunks) + lane * 4;
				if (got.x != pat(w0) || got.y != pat(w0 + 1) || got.z != pat(w0 + 2) || got.w != pat(w0 + 3)) {
					++bad;
					if (r >= 2 && got.x == pat(chunk_word(blockIdx.x, r - 2, c, total_chunks) + lane * 4)) ++old;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			if (lane == 0) flag[1] = r + 1;
		}
		if (bad) atomicAdd(errors, bad);
		if (old) atomicAdd(stale, old);
	}
}

int main()
{
	const size_t words = (size_t)128 << 20; // 512 MB: far beyond L2 + MALL
	const u32 total_chunks = (u32)(words / 256);
	u32 *d = nullptr, *e = nullptr;
	if (hipMalloc(&d, words * 4) != hipSuccess || hipMalloc(&e, 16) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
	hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, d, words);
	for (int mode = 0; mode < 2; ++mode) {
		(void)hipMemset(e, 0, 16);
		const size_t smem = 64 + 2 * N * 1024;
		if (mode == 0) hipLaunchKernelGGL(dma_order_kernel<0>, dim3(1024), dim3(128), smem, 0, d, total_chunks, e, e + 1);
		else hipLaunchKernelGGL(dma_order_kernel<1>, dim3(1024), dim3(128), smem, 0, d, total_chunks, e, e + 1);
		hipError_t rc = hipDeviceSynchronize();
		u32 r[2] = {0, 0};
		(void)hipMemcpy(r, e, 8, hipMemcpyDeviceToHost);
		printf("mode %d (%s): %s, %u lane-chunks wrong of %llu (%u of them held the data of two rounds before); sources: odd chunks from a hot 1 MB, even chunks anywhere in 512 MB\n", mode,
			mode == 0 ? "publish after vmcnt(k), k = N-1..0" : "publish after vmcnt(0)", hipGetErrorString(rc), r[0],
			(unsigned long long)1024 * ROUNDS * N * 64, r[1]);
	}
	return 0;
}
