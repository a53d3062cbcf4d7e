// diag/hello.hip — toolchain/box sanity check: one trivial kernel, device facts.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o) { o[threadIdx.x] = __shfl_up((int)threadIdx.x, 1) + (int)__popcll(__ballot(1)); }
int main()
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	printf("devices %d (%s)\n", n, hipGetErrorString(e));
	hipDeviceProp_t p;
	hipGetDeviceProperties(&p, 0);
	size_t f, t;
	hipMemGetInfo(&f, &t);
	printf("%s arch %s CUs %d clock %d kHz mem free %.1f / %.1f GB LDS/block %zu regs/block %d\n", p.name, p.gcnArchName,
		p.multiProcessorCount, p.clockRate, f / 1e9, t / 1e9, p.sharedMemPerBlock, p.regsPerBlock);
	int *d, h[64];
	hipMalloc(&d, 256);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	e = hipDeviceSynchronize();
	hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
	printf("kernel: %s h[0]=%d h[5]=%d (expect 64, 68)\n", hipGetErrorString(e), h[0], h[5]);
	return 0;
}
