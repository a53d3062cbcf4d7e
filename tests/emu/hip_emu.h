// tests/emu/hip_emu.h — TEST INFRASTRUCTURE ONLY (never part of the product library).
//
// A tiny SIMT emulator: just enough of the HIP runtime + device intrinsics for the sources in
// muscle_amd/csrc to be compiled by g++ (-DMPC_EMU) into tests/emu/libmpcgpu_emu.so, so kernel
// logic (indexing, skew schedules, boundary cases, orchestration, buffer sizing) can be checked
// against the oracle in this GPU-less container BEFORE a gpurun call is spent. Every GPU thread of a
// block is a fiber (own stack, hand-written x86-64 context switch, no system calls) of one OS thread;
// wave collectives (shuffles, ballot) and __syncthreads are cooperative barriers: a fiber that has to wait
// yields to the next one. Blocks are spread over a few OS threads. Slow by design: tiny inputs only.
// The product path (muscle_amd/csrc/libmpcgpu.so, hipcc --offload-arch=gfx950) never sees this.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Barrier {
	int n = 0, count = 0;
	unsigned gen = 0;
};
struct WaveState {
	Barrier bar;
	uint64_t xbuf[64];
	int nthreads;
};
struct BlockState {
	Barrier bar;
	std::vector<WaveState> waves;
	unsigned char *dyn_smem;
	size_t dyn_size;
};
extern thread_local BlockState *g_block; // the block this OS thread is running
extern thread_local WaveState *t_wave;   // of the running fiber
extern thread_local unsigned t_lane;
void yield(); // give the OS thread to the next fiber of the block
// LDS-DMA stand-in. Default: the copy happens at issue. EMU_DMA=late (read at every launch): the 16 destination bytes are
// poisoned at issue and the copy happens when the ISSUING thread waits (mpc_dma_wait), i.e. as late as the hardware may deliver —
// a kernel that reads staged data before its wait + barrier then computes with garbage and fails its parity test.
extern bool g_dma_late, g_dma_never;
void dma_enqueue(void *dst, const void *src, unsigned bytes);
void dma_complete(bool data_too);
// all fibers of a barrier live on one OS thread: plain counters
static inline void barrier_wait(Barrier &b)
{
	const unsigned g = b.gen;
	if (++b.count == b.n) { b.count = 0; ++b.gen; }
	else while (b.gen == g) yield();
}
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()> &body);
} // namespace emu

extern thread_local dim3 threadIdx;
extern thread_local dim3 blockIdx;
extern dim3 blockDim; // one launch at a time
extern dim3 gridDim;

#define MPC_LAUNCH(kern, grid, block, smem, stream, ...) \
	emu::launch(dim3(grid), dim3(block), (smem), [=]() { kern(__VA_ARGS__); })
#define MPC_DYN_SMEM(name) unsigned char *name = emu::g_block->dyn_smem

// ---- device intrinsics -------------------------------------------------------------------
static inline void __syncthreads() { emu::barrier_wait(emu::g_block->bar); }

template <class T> static inline T emu_xchg(T v, int src_lane_delta, bool absolute)
{
	static_assert(sizeof(T) <= 8, "emu shuffle: <= 8 bytes");
	emu::WaveState *w = emu::t_wave;
	uint64_t raw = 0;
	memcpy(&raw, &v, sizeof(T));
	w->xbuf[emu::t_lane] = raw;
	emu::barrier_wait(w->bar);
	int src = absolute ? src_lane_delta : (int)emu::t_lane + src_lane_delta;
	T r = v;
	if (src >= 0 && src < w->nthreads)
		memcpy(&r, &w->xbuf[src], sizeof(T));
	emu::barrier_wait(w->bar);
	return r;
}
template <class T> static inline T __shfl_up(T v, unsigned d) { return emu_xchg(v, -(int)d, false); }
template <class T> static inline T __shfl_down(T v, unsigned d) { return emu_xchg(v, (int)d, false); }
template <class T> static inline T __shfl(T v, int src) { return emu_xchg(v, src, true); }
#define MPC_OPAQUE(v) ((void)0)
#define MPC_OPAQUE_S(v) ((void)0)
#define MPC_KERNARG_AGAIN(p) (&(p))
#define MPC_SCHED_BARRIER() ((void)0)
#define MPC_WAVE_LDS_ORDER() ((void)__shfl(0, 0)) // a rendezvous of the WAVE (any collective is one): the emulator runs lanes one after the other between synchronisation points
#define MPC_WAVE_FENCE() ((void)0) // emulated lanes meet at every shuffle
template <class T> static inline T mpc_read_lane(T v, unsigned l) { return __shfl(v, (int)l); }
static inline unsigned mpc_row16_ror8(unsigned v) { return __shfl(v, (int)((emu::t_lane & ~15u) | ((emu::t_lane + 8u) & 15u))); }
static inline unsigned mpc_row16_scan_add(unsigned v)
{
	for (int d = 1; d < 16; d <<= 1) {
		const unsigned o = __shfl_up(v, d);
		if ((int)(emu::t_lane & 15u) >= d) v += o;
	}
	return v;
}
static inline unsigned mpc_lane_gather(unsigned v, unsigned byte_index) { return __shfl(v, (int)(byte_index / 4u)); }
template <class T> static inline T mpc_lane_up1(T v) { return __shfl_up(v, 1); }
template <class T> static inline T mpc_lane_down1(T v) { return __shfl_down(v, 1); }
static inline float mpc_lane_up1_fill(float v, float fill) { const float r = __shfl_up(v, 1); return emu::t_lane == 0 ? fill : r; }
// the same left-to-right chain as the device's 64 DPP steps, from one exchange (every lane adds the 64 values in lane order)
static inline float mpc_wave_chain_add(float total, float v)
{
	emu::WaveState *w = emu::t_wave;
	uint64_t raw = 0;
	memcpy(&raw, &v, 4);
	w->xbuf[emu::t_lane] = raw;
	emu::barrier_wait(w->bar);
	volatile float acc = total; // volatile: no reassociation, no extended precision
	for (int i = 0; i < 64; ++i) { float x = 0.0f; if (i < w->nthreads) memcpy(&x, &w->xbuf[i], 4); acc = acc + x; }
	emu::barrier_wait(w->bar);
	return acc;
}
static inline float mpc_wave_scan_max_nonneg(float v)
{
	for (int d = 1; d < 64; d <<= 1) {
		const float o = __shfl_up(v, d);
		if ((int)emu::t_lane >= d) v = v > o ? v : o;
	}
	return v;
}
static inline unsigned mpc_cvt_u32_sat(float f) { return !(f > 0.0f) ? 0u : (f >= 4294967296.0f ? 0xffffffffu : (unsigned)f); }
struct __attribute__((aligned(16))) MpcQuad { unsigned x, y, z, w; };
static inline unsigned mpc_lds_addr(const void *p) { return (unsigned)((const unsigned char *)p - emu::g_block->dyn_smem); }
// (an LDS read beyond the allocation returns zeros on the hardware: a merge fed with poison may hop anywhere)
static inline MpcQuad mpc_lds_load16(unsigned addr)
{
	if ((size_t)addr + 16 > emu::g_block->dyn_size) return MpcQuad{0, 0, 0, 0};
	return *(const MpcQuad *)(emu::g_block->dyn_smem + addr);
}
static inline unsigned mpc_lds_load4(unsigned addr) { return (size_t)addr + 4 > emu::g_block->dyn_size ? 0u : *(const unsigned *)(emu::g_block->dyn_smem + addr); }
typedef const unsigned *mpc_const_u32p;
#define MPC_CONST_U32(p) ((mpc_const_u32p)(p))
static inline unsigned mpc_write_lane(unsigned v, unsigned sv, unsigned l) { return emu::t_lane == l ? sv : v; }
// LDS-DMA stand-in: synchronous copy (the emulator has no asynchronous memory pipeline; what it checks is addressing)
static inline void mpc_dma16(const void *gsrc, void *lds_wave_base)
{
	unsigned char *dst = (unsigned char *)lds_wave_base + 16 * emu::t_lane;
	if (emu::g_dma_late) { memset(dst, 0xee, 16); emu::dma_enqueue(dst, gsrc, 16); }
	else memcpy(dst, gsrc, 16);
}
static inline void mpc_dma4(const void *gsrc, void *lds_wave_base)
{
	unsigned char *dst = (unsigned char *)lds_wave_base + 4 * emu::t_lane;
	if (emu::g_dma_late) { memset(dst, 0xee, 4); emu::dma_enqueue(dst, gsrc, 4); }
	else memcpy(dst, gsrc, 4);
}
// EMU_DMA=never (the negative control): a wait delivers only the 4-byte transfers (tables of block numbers: withheld, the
// poison would be used as addresses), never the 16-byte data transfers
static inline void mpc_dma_wait() { if (emu::g_dma_late) emu::dma_complete(!emu::g_dma_never); }
static inline unsigned mpc_wave_first(unsigned v) { return __shfl(v, 0); }
static inline unsigned long long mpc_clock() { return 0ull; }
static inline unsigned long long __ballot(int pred)
{
	emu::WaveState *w = emu::t_wave;
	w->xbuf[emu::t_lane] = pred ? 1 : 0;
	emu::barrier_wait(w->bar);
	unsigned long long m = 0;
	for (int i = 0; i < w->nthreads; ++i)
		if (w->xbuf[i]) m |= (1ull << i);
	emu::barrier_wait(w->bar);
	return m;
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }

static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned *p, unsigned v)
{
	unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
	while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
	return o;
}
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }

#define MPC_VERSION_STRING "mpcgpu 0.2 (SIMT emulator build: tests only)"
#define MPC_ALN_THREADS 128 // fewer OS threads per workgroup; calc_aln_kernel is written for any multiple of 64

// ---- host runtime subset -----------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef void *hipStream_t;
struct emuEvent { std::chrono::steady_clock::time_point t; };
typedef emuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { int multiProcessorCount; size_t totalGlobalMem; char name[64]; char gcnArchName[64]; };

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { if (d) *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
	memset(p, 0, sizeof(*p));
	p->multiProcessorCount = 2; // keep emulated grids tiny
	p->totalGlobalMem = (size_t)8 << 30;
	strcpy(p->name, "SIMT emulator (tests only)");
	strcpy(p->gcnArchName, "emu");
	return hipSuccess;
}
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)4 << 30; *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *nb, const void *, int, size_t) { *nb = 1; return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
#define hipStreamNonBlocking 1u
#define hipEventDisableTiming 2u
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; } // (non-null: "created")
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new emuEvent; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new emuEvent; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; } // (everything runs at its launch here)
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
	*ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
	return hipSuccess;
}

// stand-in for the product's rocprim::radix_sort_pairs ("device" memory is host memory here): stable order by key bits [0, end_bit)
template <class EnsureTmp>
static inline hipError_t mpc_sort_pairs(EnsureTmp, const unsigned *keys_in, unsigned *keys_out, const float *vals_in,
	float *vals_out, size_t n, unsigned end_bit, hipStream_t)
{
	const unsigned mask = end_bit >= 32 ? ~0u : ((1u << end_bit) - 1u);
	std::vector<size_t> idx(n);
	for (size_t q = 0; q < n; ++q) idx[q] = q;
	std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return (keys_in[x] & mask) < (keys_in[y] & mask); });
	for (size_t q = 0; q < n; ++q) { keys_out[q] = keys_in[idx[q]]; vals_out[q] = vals_in[idx[q]]; }
	return hipSuccess;
}
static inline void *mpc_dl_open(const char *) { return nullptr; } // no RCCL on the emulator: groups exchange by "peer" copies
static inline void *mpc_dl_sym(void *, const char *) { return nullptr; }
static inline const char *mpc_dl_error() { return "emulator build"; }
static inline void mpc_enable_peer(int, int) {}
