// diag/alloc_after_exit.hip — what does a process pay for device memory that ANOTHER process released a moment ago?
//   alloc_after_exit hold <GB>            allocate <GB> in 4 GB pieces, write them, exit (the release is the driver's)
//   alloc_after_exit take <pieces> <GB>   allocate <pieces> buffers of <GB> each, time every hipMalloc and a first write
//   alloc_after_exit vmm <pieces> <GB>    ONE address range of pieces x GB backed by <pieces> physical allocations (hipMemCreate + hipMemMap)
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/alloc_after_exit diag/alloc_after_exit.hip   (diag/alloc_after_exit.sh runs the series)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	const double t0 = now();
	(void)hipFree(nullptr);
	const double t_init = now() - t0;
	if (!strcmp(argv[1], "hold")) {
		const int gb = atoi(argv[2]);
		std::vector<void *> p;
		for (int g = 0; g < gb; g += 4) {
			void *q = nullptr;
			if (hipMalloc(&q, (size_t)4 << 30) != hipSuccess) { printf("hold: hipMalloc failed\n"); return 1; }
			(void)hipMemset(q, 1, (size_t)4 << 30);
			p.push_back(q);
		}
		(void)hipDeviceSynchronize();
		printf("hold: %d GB written, exiting without freeing\n", gb);
		return 0;
	}
	const int pieces = atoi(argv[2]);
	const double gb = argc > 3 ? atof(argv[3]) : 4.0;
	const size_t bytes = (size_t)(gb * (double)(1ull << 30));
	if (!strcmp(argv[1], "vmm")) {
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = 0;
		size_t gran = 0;
		if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { printf("vmm: no granularity\n"); return 1; }
		const size_t piece = (bytes + gran - 1) / gran * gran, total = piece * (size_t)pieces;
		const double a = now();
		void *va = nullptr;
		if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) { printf("vmm: reserve failed\n"); return 1; }
		double worst = 0;
		for (int i = 0; i < pieces; ++i) {
			hipMemGenericAllocationHandle_t h;
			const double b = now();
			if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) { printf("vmm: create failed\n"); return 1; }
			if (hipMemMap((char *)va + (size_t)i * piece, piece, 0, h, 0) != hipSuccess) { printf("vmm: map failed\n"); return 1; }
			const double d = now() - b;
			worst = d > worst ? d : worst;
		}
		hipMemAccessDesc acc = {};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { printf("vmm: set access failed\n"); return 1; }
		const double t_alloc = now() - a;
		const double w = now();
		(void)hipMemsetAsync(va, 0, total, nullptr);
		(void)hipDeviceSynchronize();
		const double t_write = now() - w;
		printf("vmm  %2d x %5.2f GB (granularity %zu KB): HIP init %.3f s, reserve + create + map + access %.3f s (worst piece %.3f), first write %.3f s\n", pieces, gb, gran >> 10, t_init, t_alloc, worst, t_write);
		return 0;
	}
	double worst = 0, total = 0;
	std::vector<void *> p;
	for (int i = 0; i < pieces; ++i) {
		void *q = nullptr;
		const double a = now();
		if (hipMalloc(&q, bytes) != hipSuccess) { printf("take: hipMalloc failed\n"); return 1; }
		const double d = now() - a;
		worst = d > worst ? d : worst; total += d;
		p.push_back(q);
	}
	const double a = now();
	for (void *q : p) (void)hipMemsetAsync(q, 0, bytes, nullptr);
	(void)hipDeviceSynchronize();
	const double t_write = now() - a;
	printf("take %2d x %5.2f GB: HIP init %.3f s, hipMalloc total %.3f s (worst %.3f), first write %.3f s\n", pieces, gb, t_init, total, worst, t_write);
	return 0;
}
