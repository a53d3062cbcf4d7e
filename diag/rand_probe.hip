// diag/rand_probe.hip — does bringing up the HIP runtime consume libc rand()? Prints the first
// values of rand() in a process that (a) never touches HIP, (b) initialises HIP and launches a
// kernel first. If the runtime draws from the same generator the two sequences differ.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
__global__ void k(int *o) { o[threadIdx.x] = threadIdx.x; }
int main(int argc, char **argv)
{
	const bool use_hip = argc > 1 && !strcmp(argv[1], "hip");
	if (use_hip) {
		int n = 0;
		(void)hipGetDeviceCount(&n);
		int *d = nullptr;
		(void)hipMalloc(&d, 256);
		hipStream_t st;
		(void)hipStreamCreate(&st);
		hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, d);
		(void)hipStreamSynchronize(st);
		usleep(20000);
	}
	printf("%s:", use_hip ? "after HIP init" : "no HIP       ");
	for (int i = 0; i < 6; ++i) printf(" %d", rand());
	printf("\n");
	return 0;
}
