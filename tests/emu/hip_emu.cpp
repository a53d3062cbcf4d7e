// tests/emu/hip_emu.cpp — TEST INFRASTRUCTURE ONLY. See hip_emu.h.
#include "hip_emu.h"

#include <mutex>
#include <sys/mman.h>

thread_local dim3 threadIdx;
thread_local dim3 blockIdx;
dim3 blockDim;
dim3 gridDim;

// Context switch between fibers of one OS thread (System V x86-64): callee-saved registers and the
// stack pointer; the floating-point control state is the same everywhere and is left alone.
extern "C" void emu_switch(void **save_sp, void *load_sp);
asm(R"(
	.text
	.globl emu_switch
	.type emu_switch, @function
emu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size emu_switch, .-emu_switch
)");

namespace emu {
thread_local BlockState *g_block = nullptr;
thread_local WaveState *t_wave = nullptr;
thread_local unsigned t_lane = 0;

namespace {
constexpr size_t STACK_BYTES = 256 * 1024; // mapped lazily: only touched pages cost memory

struct Fiber {
	void *sp = nullptr;
	char *stack = nullptr;
	unsigned tid = 0;
	bool done = false;
	struct Dma { void *dst; const void *src; unsigned bytes; };
	std::vector<Dma> dma; // EMU_DMA=late: copies issued and not yet waited for
};

struct Worker {
	std::vector<Fiber> fibers;
	Fiber *cur = nullptr;
	void *sched_sp = nullptr;
	const std::function<void()> *body = nullptr;
	~Worker()
	{
		for (Fiber &f : fibers)
			if (f.stack) munmap(f.stack, STACK_BYTES);
	}
};
thread_local Worker *t_worker = nullptr;

void fiber_entry()
{
	Worker *w = t_worker;
	(*w->body)();
	dma_complete(true); // a thread that ends with transfers in flight: they land
	w->cur->done = true;
	emu_switch(&w->cur->sp, w->sched_sp); // never resumed
	abort();
}

void run_block(Worker &w, unsigned bid, dim3 block, size_t smem)
{
	const unsigned nthreads = block.x;
	BlockState bs;
	bs.bar.n = (int)nthreads;
	const unsigned nwaves = (nthreads + 63) / 64;
	bs.waves.resize(nwaves);
	for (unsigned k = 0; k < nwaves; ++k) {
		bs.waves[k].nthreads = (int)std::min(64u, nthreads - k * 64);
		bs.waves[k].bar.n = bs.waves[k].nthreads;
	}
	std::vector<unsigned char> dyn(smem + 64);
	bs.dyn_smem = (unsigned char *)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
	bs.dyn_size = smem;
	g_block = &bs;
	if (w.fibers.size() < nthreads) w.fibers.resize(nthreads);
	for (unsigned t = 0; t < nthreads; ++t) {
		Fiber &f = w.fibers[t];
		if (!f.stack) {
			void *m = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
			if (m == MAP_FAILED) { perror("emu: mmap fiber stack"); abort(); }
			f.stack = (char *)m;
		}
		f.tid = t;
		f.done = false;
		// initial frame: six callee-saved registers, then the entry point as the return address; after the
		// `ret` the stack pointer is 8 modulo 16, as at any function entry
		uintptr_t top = ((uintptr_t)f.stack + STACK_BYTES) & ~(uintptr_t)15;
		void **sp = (void **)(top - 8);
		*sp = nullptr;                // "return address" of fiber_entry (it never returns)
		*--sp = (void *)&fiber_entry; // popped by emu_switch's ret
		for (int k = 0; k < 6; ++k) *--sp = nullptr;
		f.sp = sp;
	}
	// Order in which the fibers get the OS thread in each round. Between two barriers a fiber runs without interruption,
	// so a data race inside a barrier interval shows as a result that depends on this order: EMU_SCHED=reverse and
	// EMU_SCHED=random (a new shuffle every round, fixed seed) make the tests look at other orders than 0..n-1.
	static const int sched_mode = [] {
		const char *m = getenv("EMU_SCHED");
		return (m && !strcmp(m, "reverse")) ? 1 : (m && !strcmp(m, "random")) ? 2 : 0;
	}();
	std::vector<unsigned> order(nthreads);
	for (unsigned t = 0; t < nthreads; ++t) order[t] = sched_mode == 1 ? nthreads - 1 - t : t;
	uint64_t rng = 0x9e3779b97f4a7c15ull ^ (uint64_t)bid;
	unsigned remaining = nthreads;
	while (remaining) {
		if (sched_mode == 2)
			for (unsigned k = nthreads - 1; k > 0; --k) { // Fisher-Yates with xorshift64
				rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
				std::swap(order[k], order[rng % (k + 1)]);
			}
		for (unsigned oi = 0; oi < nthreads; ++oi) {
			const unsigned t = order[oi];
			Fiber &f = w.fibers[t];
			if (f.done) continue;
			w.cur = &f;
			threadIdx = dim3(t, 0, 0);
			blockIdx = dim3(bid, 0, 0);
			t_wave = &bs.waves[t / 64];
			t_lane = t % 64;
			emu_switch(&w.sched_sp, f.sp);
			if (f.done) --remaining;
		}
	}
	g_block = nullptr;
}
} // namespace

bool g_dma_late = false, g_dma_never = false; // never: the waits deliver nothing (only the thread's end does) — the negative control of the late mode
void dma_enqueue(void *dst, const void *src, unsigned bytes) { t_worker->cur->dma.push_back({dst, src, bytes}); }
void dma_complete(bool data_too)
{
	Fiber *f = t_worker ? t_worker->cur : nullptr;
	if (!f) return;
	size_t keep = 0;
	for (auto &c : f->dma) {
		if (data_too || c.bytes < 16) memcpy(c.dst, c.src, c.bytes);
		else f->dma[keep++] = c;
	}
	f->dma.resize(keep);
}

void yield()
{
	Worker *w = t_worker;
	emu_switch(&w->cur->sp, w->sched_sp);
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()> &body)
{
	if (block.x == 0 || grid.x == 0)
		return;
	static std::mutex one_launch; // blockDim/gridDim are process globals: launches of different host threads take turns
	std::lock_guard<std::mutex> guard(one_launch);
	blockDim = block;
	gridDim = grid;
	{ const char *m = getenv("EMU_DMA"); g_dma_late = m && (!strcmp(m, "late") || !strcmp(m, "never")); g_dma_never = m && !strcmp(m, "never"); }
	static const unsigned hw = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
	const unsigned nworkers = std::min(hw, grid.x);
	auto work = [&](unsigned wi) {
		Worker w;
		w.body = &body;
		t_worker = &w;
		for (unsigned b = wi; b < grid.x; b += nworkers)
			run_block(w, b, block, smem);
		t_worker = nullptr;
	};
	if (nworkers == 1) { // no OS thread at all for single-block launches
		work(0);
		return;
	}
	std::vector<std::thread> th;
	for (unsigned wi = 0; wi < nworkers; ++wi) th.emplace_back(work, wi);
	for (auto &t : th) t.join();
}
} // namespace emu
