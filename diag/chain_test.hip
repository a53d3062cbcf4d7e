// diag/chain_test.hip — mpc_wave_chain_add (mpc_platform.h) against the sequential float sum it must reproduce bit for bit.
#include "../muscle_amd/csrc/mpc_platform.h"
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float *in, float *out, int nchunks)
{
	float t = 0.0f;
	for (int c = 0; c < nchunks; ++c) t = mpc_wave_chain_add(t, in[c * 64 + (threadIdx.x & 63)]);
	out[threadIdx.x] = t;
}
int main()
{
	const int nchunks = 500;
	std::vector<float> h(nchunks * 64);
	unsigned s = 12345;
	for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (float)(s >> 8) * (1.0f / 16777216.0f) * ((s & 64) ? 1.0f : 0.01f); }
	volatile float ref = 0.0f;
	for (float x : h) ref = ref + x;
	float *din, *dout;
	hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 256 * 4);
	hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, din, dout, nchunks);
	float got[256];
	hipError_t e = hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost);
	float r = ref;
	printf("%s: device %.9g (lane 0) %.9g (lane 200), host sequential %.9g: %s\n", hipGetErrorString(e), got[0], got[200], r,
		(!memcmp(&got[0], &r, 4) && !memcmp(&got[200], &r, 4)) ? "BIT-IDENTICAL" : "DIFFERENT");
	return 0;
}
