// tests/emu/hip_emu.cpp — TEST INFRASTRUCTURE ONLY. See hip_emu.h.
#include "hip_emu.h"

thread_local dim3 threadIdx;
thread_local dim3 blockIdx;
dim3 blockDim;
dim3 gridDim;

namespace emu {
BlockState *g_block = nullptr;
thread_local WaveState *t_wave = nullptr;
thread_local unsigned t_lane = 0;

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()> &body)
{
	const unsigned nthreads = block.x;
	if (nthreads == 0 || grid.x == 0)
		return;
	BlockState bs;
	pthread_barrier_init(&bs.bar, nullptr, nthreads);
	const unsigned nwaves = (nthreads + 63) / 64;
	for (unsigned w = 0; w < nwaves; ++w) {
		WaveState *ws = new WaveState;
		ws->nthreads = (int)std::min(64u, nthreads - w * 64);
		pthread_barrier_init(&ws->bar, nullptr, ws->nthreads);
		bs.waves.push_back(ws);
	}
	std::vector<unsigned char> dyn(smem + 64);
	bs.dyn_smem = (unsigned char *)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
	g_block = &bs;
	blockDim = block;
	gridDim = grid;
	std::vector<std::thread> th;
	th.reserve(nthreads);
	for (unsigned t = 0; t < nthreads; ++t) {
		th.emplace_back([&, t]() {
			threadIdx = dim3(t, 0, 0);
			t_wave = bs.waves[t / 64];
			t_lane = t % 64;
			for (unsigned b = 0; b < grid.x; ++b) {
				blockIdx = dim3(b, 0, 0);
				body();
				pthread_barrier_wait(&bs.bar); // all threads of the block finish before the next block
			}
		});
	}
	for (auto &t : th)
		t.join();
	for (auto *w : bs.waves) {
		pthread_barrier_destroy(&w->bar);
		delete w;
	}
	pthread_barrier_destroy(&bs.bar);
	g_block = nullptr;
}
} // namespace emu
