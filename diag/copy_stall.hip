// diag: what makes the FIRST small host<->device copy after some other operation take milliseconds on an idle MI355X?
// (relax_band's tile cutter waits 16 - 25 ms for a 16 KB upload right after the store build: profiles/r10k)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void heavy(float *a, const float *b, unsigned long long n, int rounds)
{
	// HBM-bound with ALU work on every CU: the power state of a relax / store-build phase
	for (int r = 0; r < rounds; ++r)
		for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
			float v = b[i];
			for (int k = 0; k < 16; ++k) v = v * 1.0001f + 0.5f;
			a[i] = v;
		}
}
__global__ void spin(unsigned long long cycles, unsigned *out) { unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	hipStream_t st; CK(hipStreamCreate(&st));
	void *hp; CK(hipHostMalloc(&hp, 1 << 20));
	void *dp; CK(hipMalloc(&dp, 1 << 20));
	void *big1, *big2; CK(hipMalloc(&big1, 6ull << 30)); CK(hipMalloc(&big2, 6ull << 30));
	auto probe = [&](const char *what) {
		double t0 = now();
		hipMemcpyAsync(dp, hp, 16384, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
		double t1 = now();
		hipMemcpyAsync(dp, hp, 16384, hipMemcpyHostToDevice, st); hipStreamSynchronize(st);
		double t2 = now();
		hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, st, 1000ull, (unsigned *)dp); hipStreamSynchronize(st);
		double t3 = now();
		printf("%-44s first copy %.3f ms, second copy %.3f ms, tiny kernel %.3f ms\n", what, t1 - t0, t2 - t1, t3 - t2);
	};
	probe("start");
	probe("again");
	hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, 5000000ull, (unsigned *)nullptr); CK(hipStreamSynchronize(st)); // 50 ms at 100 MHz
	probe("after a 50 ms kernel");
	CK(hipMemcpyAsync(big2, big1, 6ull << 30, hipMemcpyDeviceToDevice, st)); CK(hipStreamSynchronize(st));
	probe("after a 6 GB device-to-device copy");
	{ void *t; CK(hipMalloc(&t, 2ull << 30)); CK(hipFree(t)); }
	probe("after hipMalloc + hipFree of 2 GB");
	{ void *t; CK(hipMalloc(&t, 2ull << 30)); CK(hipMemsetAsync(t, 0, 2ull << 30, st)); CK(hipStreamSynchronize(st)); CK(hipFree(t)); }
	probe("after malloc + memset + free of 2 GB");
	for (int ms : {1, 10, 100, 1000}) { std::this_thread::sleep_for(std::chrono::milliseconds(ms)); char b[64]; snprintf(b, sizeof b, "after %d ms of idle", ms); probe(b); }
	for (int gap_us : {0, 50, 200, 1000, 5000}) {
		for (int rep = 0; rep < 3; ++rep) {
			hipLaunchKernelGGL(heavy, dim3(2048), dim3(256), 0, st, (float *)big1, (const float *)big2, (6ull << 30) / 4, 4);
			CK(hipStreamSynchronize(st));
			if (gap_us) std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
			char b[80]; snprintf(b, sizeof b, "after a heavy kernel + %d us of idle (rep %d)", gap_us, rep); probe(b);
		}
	}
	std::vector<char> pageable(1 << 20);
	{ double t0 = now(); hipMemcpyAsync(dp, pageable.data(), 16384, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); printf("pageable 16 KB upload: %.3f ms\n", now() - t0); }
	{ double t0 = now(); hipMemcpyAsync(pageable.data(), dp, 16384, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); printf("pageable 16 KB download: %.3f ms\n", now() - t0); }
	return 0;
}
