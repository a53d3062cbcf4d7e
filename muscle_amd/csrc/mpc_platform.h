// mpc_platform.h — the only place that knows whether we are compiled by hipcc for gfx950 (the
// product) or by g++ against tests/emu/hip_emu.h (MPC_EMU: the test-only SIMT emulator used to
// validate kernel logic on a GPU-less box; never shipped, never a fallback).
#pragma once
#ifdef MPC_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#include <cstring>
#include <dlfcn.h>
#include <rocprim/device/device_radix_sort.hpp>
#define MPC_VERSION_STRING "mpcgpu 0.2 (HIP gfx950)"
#define MPC_ALN_THREADS 1024 // workgroup of calc_aln_kernel (kernels_aln.h)
// STABLE sort of n (key, value) records by key bits [0, end_bit) — records with equal keys keep their input order, which
// kernels_prog.h relies on: rocprim::radix_sort_pairs, called directly (bulk data movement). ensure_tmp(bytes) must return device scratch of at least that size (or nullptr on failure).
template <class EnsureTmp>
inline hipError_t mpc_sort_pairs(EnsureTmp ensure_tmp, const unsigned *keys_in, unsigned *keys_out, const float *vals_in,
	float *vals_out, size_t n, unsigned end_bit, hipStream_t stream)
{
	size_t tmp_bytes = 0;
	hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
	if (e != hipSuccess) return e;
	void *tmp = ensure_tmp(tmp_bytes < 16 ? (size_t)16 : tmp_bytes);
	if (!tmp) return hipErrorOutOfMemory;
	return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
}
// shared library by name / symbol by name (librccl is loaded on first use: mpcgpu_group.cpp)
inline void *mpc_dl_open(const char *name) { return dlopen(name, RTLD_NOW | RTLD_GLOBAL); }
inline void *mpc_dl_sym(void *lib, const char *name) { return dlsym(lib, name); }
inline const char *mpc_dl_error() { const char *e = dlerror(); return e ? e : "unknown error"; }
// direct access from device `dev` to device `peer` when the hardware offers it (xGMI); "already enabled" is not an error
inline void mpc_enable_peer(int dev, int peer)
{
	(void)hipSetDevice(dev);
	int can = 0;
	if (hipDeviceCanAccessPeer(&can, dev, peer) == hipSuccess && can && hipDeviceEnablePeerAccess(peer, 0) != hipSuccess) (void)hipGetLastError();
}
#define MPC_LAUNCH(kern, grid, block, smem, stream, ...) \
	hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (smem), (stream), __VA_ARGS__)
#define MPC_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// Lane shifts by one across the whole 64-lane wave as DPP moves (v_mov_b32_dpp wave_shr:1 /
// wave_shl:1, gfx9 family) instead of ds_bpermute: no LDS round trip, a few cycles of latency.
// Lane 0 (resp. 63) has no source and keeps its own value, like __shfl_up/__shfl_down.
__device__ __forceinline__ int mpc_lane_up1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int mpc_lane_down1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float mpc_lane_up1(float v) { return __builtin_bit_cast(float, mpc_lane_up1(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ float mpc_lane_down1(float v) { return __builtin_bit_cast(float, mpc_lane_down1(__builtin_bit_cast(int, v))); }
// the same shift with lane 0 receiving `fill` (the DPP move keeps the old destination value in lanes without a source)
__device__ __forceinline__ float mpc_lane_up1_fill(float v, float fill)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// ((((total + v[0]) + v[1]) + ...) + v[63]) over the wave's 64 lane values, strictly left to right, as a wave-uniform result
// (`total` is wave-uniform). The accumulator lives in every lane of a DPP row of 16 lanes; term k of the row is added with
// v_add_f32_dpp acc, v, acc row_newbcast:k — the DPP operand is v (lane k of the row, broadcast to the row), the accumulator is an
// ordinary operand, so consecutive adds depend on each other through a plain register read and issue back to back: ~5 cycles per term.
// After a row's 16 terms its lane 15 hands the accumulator to the next row (v_mov_b32_dpp row_bcast:15 — a DPP read of a register just
// written: the two wait states that needs are spelled out, the hazard recognizer does not look into inline asm). Rounds 1-4 shifted the
// partial sums instead (p = wave_shr:1(p) + w): a DPP read of the just-written register and two wait states PER TERM, ~13 cycles
// (diag/chain_time.hip measures both). Every term is added exactly once, in sequence, each add rounded on its own; a + b == b + a bit for bit.
__device__ __forceinline__ float mpc_wave_chain_add(float total, float v)
{
	float acc = total;
	asm volatile(
		"s_nop 1\n\t"
		".irp k,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:\\k row_mask:0x1 bank_mask:0xf\n\t.endr\n\t"
		"s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0x2 bank_mask:0xf\n\t"
		".irp k,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:\\k row_mask:0x2 bank_mask:0xf\n\t.endr\n\t"
		"s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0x4 bank_mask:0xf\n\t"
		".irp k,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:\\k row_mask:0x4 bank_mask:0xf\n\t.endr\n\t"
		"s_nop 1\n\tv_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0x8 bank_mask:0xf\n\t"
		".irp k,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15\n\tv_add_f32_dpp %0, %1, %0 row_newbcast:\\k row_mask:0x8 bank_mask:0xf\n\t.endr"
		: "+v"(acc) : "v"(v));
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 63));
}
// the form of rounds 1-4 (kept for diag/chain_time.hip's comparison): lane 0's term becomes total + v[0], then 64 steps of
// p = shift_up(p) + w, a lane without a source reading 0.0f; after 64 steps lane 63 holds the whole chain
__device__ __forceinline__ float mpc_wave_chain_add_shift(float total, float v)
{
	const float w = (threadIdx.x & 63u) == 0u ? total + v : v;
	float p = 0.0f;
	asm volatile(".rept 64\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t.endr" : "+v"(p) : "v"(w));
	return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 63));
}
// Inclusive prefix maximum over the wave for values >= 0 (identity 0.0f): DPP row shifts inside the
// rows of 16 lanes, then row_bcast:15 / row_bcast:31 across rows. max is exact and associative, so
// the order of combination does not matter.
__device__ __forceinline__ float mpc_wave_scan_max_nonneg(float v)
{
#define MPC_DPP_MAX(ctrl, rmask)                                                                              \
	v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, false)))
	MPC_DPP_MAX(0x111, 0xf); // row_shr:1
	MPC_DPP_MAX(0x112, 0xf); // row_shr:2
	MPC_DPP_MAX(0x114, 0xf); // row_shr:4
	MPC_DPP_MAX(0x118, 0xf); // row_shr:8
	MPC_DPP_MAX(0x142, 0xa); // row_bcast:15 -> rows 1 and 3
	MPC_DPP_MAX(0x143, 0xc); // row_bcast:31 -> rows 2 and 3
#undef MPC_DPP_MAX
	return v;
}
// inclusive prefix sum inside every DPP row of 16 lanes (v_add_u32 with row_shr:1,2,4,8 sources; a lane without a source adds 0)
__device__ __forceinline__ unsigned mpc_row16_scan_add(unsigned v)
{
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
	return v;
}
// lane l receives the value of lane (l + 8) mod 16 of its DPP row (row_ror:8)
__device__ __forceinline__ unsigned mpc_row16_ror8(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned mpc_lds_load4(unsigned addr) { return *(const __attribute__((address_space(3))) unsigned *)(unsigned long long)addr; }
// value of lane `l` (wave-uniform index) as a scalar: v_readlane_b32, no LDS
__device__ __forceinline__ unsigned mpc_read_lane(unsigned v, unsigned l) { return (unsigned)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ float mpc_read_lane(float v, unsigned l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), (int)l)); }
// cross-lane gather: lane l receives v of lane byte_index[l] / 4 (ds_bpermute_b32: the LDS crossbar, no LDS memory)
__device__ __forceinline__ unsigned mpc_lane_gather(unsigned v, unsigned byte_index) { return (unsigned)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)v); }
// optimisation barrier on one VGPR value (no code): stops hoisting of what is derived from it
#define MPC_OPAQUE(v) asm volatile("" : "+v"(v))
#define MPC_OPAQUE_S(v) asm volatile("" : "+s"(v)) // the same for a wave-uniform value in a scalar register
// The kernel's (single, struct) argument read from the kernarg segment AGAIN, through a pointer the compiler cannot connect with
// the argument it already loaded: pointers an epilogue needs are then scalar loads there instead of registers held across the
// kernel's main loop. Use as MPC_KERNARG_AGAIN(p)->field.
template <class T> __device__ __forceinline__ const __attribute__((address_space(4))) T *mpc_kernarg_again(const T &)
{
	unsigned long long k = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
	asm volatile("" : "+s"(k));
	return (const __attribute__((address_space(4))) T *)k;
}
#define MPC_KERNARG_AGAIN(p) mpc_kernarg_again(p)
// orders this wave's earlier global stores before its later global loads (other lanes' data): s_waitcnt only,
// the waves of a workgroup share the CU's L1
#define MPC_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")
// float -> u32, truncating, saturating (negative and NaN -> 0, >= 2^32 -> 0xffffffff): what v_cvt_u32_f32 does. The C cast is
// undefined outside the range, so the instruction is named (plain asm: schedulable, no side effects).
__device__ __forceinline__ unsigned mpc_cvt_u32_sat(float f) { unsigned r; asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(f)); return r; }
// lane `l` (wave-uniform) of v := sv (wave-uniform scalar). A compare and a select (v_writelane_b32 cannot take its value
// and its lane number from two different SGPRs on gfx9-class hardware); one-dimensional workgroups of whole waves assumed.
__device__ __forceinline__ unsigned mpc_write_lane(unsigned v, unsigned sv, unsigned l) { return (threadIdx.x & 63u) == l ? sv : v; }
// LDS-DMA, 16 bytes per lane: lane L's 16 bytes at `gsrc` go to LDS address `lds_wave_base + 16 * L` without passing
// through VGPRs (global_load_lds_dwordx4; the LDS base is wave-uniform and travels in M0). Asynchronous: the data is
// in LDS once the ISSUING wave has waited for vmcnt(0) (mpc_dma_wait) — other waves additionally need a barrier after that.
__device__ __forceinline__ void mpc_dma16(const void *gsrc, void *lds_wave_base)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
		(__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
// the same with 4 bytes per lane: lane L's dword goes to `lds_wave_base + 4 * L` (global_load_lds_dword)
__device__ __forceinline__ void mpc_dma4(const void *gsrc, void *lds_wave_base)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
		(__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void mpc_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// LDS by 32-bit address: the address of a dynamic-LDS location as an integer, and an aligned 16-byte read through such an
// address (ds_read_b128 on the register as it is: no per-access add of the dynamic-LDS base).
struct __attribute__((aligned(16))) MpcQuad { unsigned x, y, z, w; };
__device__ __forceinline__ unsigned mpc_lds_addr(const void *p) { return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)p; }
typedef unsigned mpc_uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MpcQuad mpc_lds_load16(unsigned addr)
{
	const mpc_uint4v v = *(const __attribute__((address_space(3))) mpc_uint4v *)(unsigned long long)addr;
	MpcQuad q; q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
	return q;
}
// ---- relax_var_kernel's merge of one (cell, Z), hand-scheduled (kernels_relaxv.h holds the C++ statement of the same merge:
// MpcRvBlocksCxx — what the emulator runs and what MPCGPU_RELAX_MERGE=cxx selects on the device for A/B).
// Two sets of 2 x 4 VGPRs hold the blocks {p0, p1, c0 | dist << 16, c1} of the two rows: the set of the slot being merged and
// the set the NEXT slot's first blocks are read into meanwhile. ds_read_b128 wants register quadruples and the arithmetic
// wants their single registers, which inline asm can only name when the quadruples are physical registers: v[24:39].
// Per step of the merge (the loop below): 8 compares, 4 selects, 2 products, 2 adds, and — only for the lanes that go on — two
// adds under the advance masks (exec), against 23 VALU instructions from the compiler (two redundant compares, three
// selects and a shift for the advance). "s_orn2 nl, nl, adv" is "the row has a next block, or it is not the row that advances".
typedef unsigned long long mpc_u64s;
struct MpcRvBlocksAsm {
	mpc_uint4v a0, b0, a1, b1; // set 0: v[24:27], v[28:31]; set 1: v[32:35], v[36:39]
	__device__ __forceinline__ void load(int set, unsigned ia, unsigned ib)
	{
		if (set == 0) asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "={v[24:27]}"(a0), "={v[28:31]}"(b0) : "v"(ia), "v"(ib) : "memory");
		else asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "={v[32:35]}"(a1), "={v[36:39]}"(b1) : "v"(ia), "v"(ib) : "memory");
	}
	// after the last slot of a step: the look-ahead reads the last merge issued are still in flight into one of the register sets
	__device__ __forceinline__ void drain() { asm volatile("s_waitcnt lgkmcnt(0)" : "+{v[24:27]}"(a0), "+{v[28:31]}"(b0), "+{v[32:35]}"(a1), "+{v[36:39]}"(b1) : : "memory"); }
#define MPC_RV_MERGE_ASM(A0_, A1_, A2_, A3_, B0_, B1_, B2_, B3_, CURA_, CURB_, NXTA_, NXTB_)                           \
	"s_waitcnt lgkmcnt(0)\n\t" /* this slot's first blocks (read during the previous slot) have landed */               \
	"ds_read_b128 " NXTA_ ", %[nia]\n\t"                                                                                 \
	"ds_read_b128 " NXTB_ ", %[nib]\n\t"                                                                                 \
	"s_mov_b64 %[sv], exec\n"                                                                                            \
	".Lrv_step_%=:\n\t"                                                                                                  \
	"v_cmp_eq_u32_sdwa %[e01], " A2_ ", " B3_ " src0_sel:WORD_0 src1_sel:DWORD\n\t"                                      \
	"v_cmp_eq_u32_sdwa %[e00], " A2_ ", " B2_ " src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                     \
	"v_cmp_eq_u32_e64 %[e11], " A3_ ", " B3_ "\n\t"                                                                      \
	"v_cmp_eq_u32_sdwa %[e10], " A3_ ", " B2_ " src0_sel:DWORD src1_sel:WORD_0\n\t"                                      \
	"v_cmp_le_u32_e64 %[ada], " A3_ ", " B3_ "\n\t"                                                                      \
	"v_cmp_le_u32_e64 %[adb], " B3_ ", " A3_ "\n\t"                                                                      \
	"v_cmp_lt_u32_e64 %[nla], %[kf], " A2_ "\n\t"                                                                        \
	"v_cmp_lt_u32_e64 %[nlb], %[kf], " B2_ "\n\t"                                                                        \
	"v_cndmask_b32_e64 %[t0], 0, " B1_ ", %[e01]\n\t"                                                                    \
	"v_cndmask_b32_e64 %[t1], 0, " B1_ ", %[e11]\n\t"                                                                    \
	"v_cndmask_b32_e64 %[t0], %[t0], " B0_ ", %[e00]\n\t" /* of two equal columns in B the first wins */                 \
	"v_cndmask_b32_e64 %[t1], %[t1], " B0_ ", %[e10]\n\t"                                                                \
	"v_mul_f32_e32 %[t0], " A0_ ", %[t0]\n\t"                                                                            \
	"v_mul_f32_e32 %[t1], " A1_ ", %[t1]\n\t"                                                                            \
	"v_add_f32_e32 %[sum], %[sum], %[t0]\n\t" /* z ascending: relaxflat.cpp:27, product rounded, then added */           \
	"v_add_f32_e32 %[sum], %[sum], %[t1]\n\t"                                                                            \
	"s_orn2_b64 %[nla], %[nla], %[ada]\n\t"                                                                              \
	"s_orn2_b64 %[nlb], %[nlb], %[adb]\n\t"                                                                              \
	"s_and_b64 %[nla], %[nla], %[nlb]\n\t"                                                                               \
	"s_and_b64 exec, exec, %[nla]\n\t" /* the lanes that go on; scc = any */                                             \
	"s_cbranch_scc0 .Lrv_done_%=\n\t"                                                                                    \
	"s_mov_b64 %[nlb], exec\n\t"                                                                                         \
	"s_and_b64 exec, %[nlb], %[ada]\n\t"                                                                                 \
	"v_add_u32_sdwa %[ia], %[ia], " A2_ " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"        \
	"ds_read_b128 " CURA_ ", %[ia]\n\t" /* only the lanes whose row advanced read its next block: the others keep theirs */ \
	"s_and_b64 exec, %[nlb], %[adb]\n\t"                                                                                 \
	"v_add_u32_sdwa %[ib], %[ib], " B2_ " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"        \
	"ds_read_b128 " CURB_ ", %[ib]\n\t"                                                                                  \
	"s_mov_b64 exec, %[nlb]\n\t"                                                                                         \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
	"s_branch .Lrv_step_%=\n"                                                                                            \
	".Lrv_done_%=:\n\t"                                                                                                  \
	"s_mov_b64 exec, %[sv]"
	// merges the slot whose first blocks are in set SET (addresses ia, ib) onto sum; reads the blocks at nia, nib into the other set
	template <int SET> __device__ __forceinline__ void merge(float &sum, unsigned ia, unsigned ib, unsigned nia, unsigned nib)
	{
		mpc_u64s sv, e00, e01, e10, e11, ada, adb, nla, nlb;
		float t0, t1;
		const unsigned kf = 0xffffu;
		if (SET == 0)
			asm volatile(MPC_RV_MERGE_ASM("v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v[24:27]", "v[28:31]", "v[32:35]", "v[36:39]")
				: [sum] "+v"(sum), [ia] "+v"(ia), [ib] "+v"(ib), "+{v[24:27]}"(a0), "+{v[28:31]}"(b0), "=&{v[32:35]}"(a1), "=&{v[36:39]}"(b1),
				  [t0] "=&v"(t0), [t1] "=&v"(t1), [sv] "=&s"(sv), [e00] "=&s"(e00), [e01] "=&s"(e01), [e10] "=&s"(e10), [e11] "=&s"(e11),
				  [ada] "=&s"(ada), [adb] "=&s"(adb), [nla] "=&s"(nla), [nlb] "=&s"(nlb)
				: [nia] "v"(nia), [nib] "v"(nib), [kf] "s"(kf)
				: "vcc", "scc", "memory");
		else
			asm volatile(MPC_RV_MERGE_ASM("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v[32:35]", "v[36:39]", "v[24:27]", "v[28:31]")
				: [sum] "+v"(sum), [ia] "+v"(ia), [ib] "+v"(ib), "+{v[32:35]}"(a1), "+{v[36:39]}"(b1), "=&{v[24:27]}"(a0), "=&{v[28:31]}"(b0),
				  [t0] "=&v"(t0), [t1] "=&v"(t1), [sv] "=&s"(sv), [e00] "=&s"(e00), [e01] "=&s"(e01), [e10] "=&s"(e10), [e11] "=&s"(e11),
				  [ada] "=&s"(ada), [adb] "=&s"(adb), [nla] "=&s"(nla), [nlb] "=&s"(nlb)
				: [nia] "v"(nia), [nib] "v"(nib), [kf] "s"(kf)
				: "vcc", "scc", "memory");
	}
#undef MPC_RV_MERGE_ASM
};
#define MPC_RV_HAVE_ASM 1
// ---- the same merge for relax_band_kernel (kernels_relaxb.h: MpcRbBlocksCxx is its C++ statement). Differences: the addresses
// ia, ib are those of the rows' FIRST blocks plus the records' hop biases — valid block addresses only once a block's distance has
// been added — and the bias of the Y record differs per lane: it is gathered across lanes (ds_bpermute_b32 on the register that
// holds the Y records' biases in lanes 0..7) one slot ahead, inside the statement, so that its latency hides behind the merge like
// that of the next slot's first blocks; the gathered value lives in v40 from statement to statement (`hb`). Every LDS operation
// issued here is covered by the NEXT statement's opening wait or by drain() after the last slot.
struct MpcRbBlocksAsm {
	static constexpr bool WINDOW = false;
	mpc_uint4v a0, b0, a1, b1; // set 0: v[24:27], v[28:31]; set 1: v[32:35], v[36:39]
	unsigned hb;               // v40: hop bias of the next slot's Y record (in flight until the next opening wait)
	// first blocks of the first slot into set 0, and its Y bias (read from the bias table in LDS at hb_addr)
	__device__ __forceinline__ void load(unsigned ia, unsigned ib, unsigned hb_addr)
	{
		asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b32 %2, %5"
			: "={v[24:27]}"(a0), "={v[28:31]}"(b0), "={v40}"(hb) : "v"(ia), "v"(ib), "v"(hb_addr) : "memory");
	}
	__device__ __forceinline__ void drain() { asm volatile("s_waitcnt lgkmcnt(0)" : "+{v[24:27]}"(a0), "+{v[28:31]}"(b0), "+{v[32:35]}"(a1), "+{v[36:39]}"(b1), "+{v40}"(hb) : : "memory"); }
#define MPC_RB_MERGE_HEAD(A0_, A1_, A2_, A3_, B0_, B1_, B2_, B3_, CURA_, CURB_, NXTA_, NXTB_)                           \
	"s_waitcnt lgkmcnt(0)\n\t" /* this slot's first blocks and its Y bias (issued during the previous slot) have landed */ \
	"v_add_u32_e32 %[ib], %[ib], %[hb]\n\t"                                                                              \
	"ds_read_b128 " NXTA_ ", %[nia]\n\t"                                                                                 \
	"ds_read_b128 " NXTB_ ", %[nib]\n\t"                                                                                 \
	"ds_bpermute_b32 %[hb], %[nidx], %[by]\n\t"                                                                          \
	"s_mov_b64 %[sv], exec\n"                                                                                            \
	".Lrv_step_%=:\n\t"                                                                                                  \
	"v_cmp_eq_u32_sdwa %[e01], " A2_ ", " B3_ " src0_sel:WORD_0 src1_sel:DWORD\n\t"                                      \
	"v_cmp_eq_u32_sdwa %[e00], " A2_ ", " B2_ " src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                     \
	"v_cmp_eq_u32_e64 %[e11], " A3_ ", " B3_ "\n\t"                                                                      \
	"v_cmp_eq_u32_sdwa %[e10], " A3_ ", " B2_ " src0_sel:DWORD src1_sel:WORD_0\n\t"                                      \
	"v_cmp_le_u32_e64 %[ada], " A3_ ", " B3_ "\n\t"                                                                      \
	"v_cmp_le_u32_e64 %[adb], " B3_ ", " A3_ "\n\t"                                                                      \
	"v_cmp_lt_u32_e64 %[nla], %[kf], " A2_ "\n\t"                                                                        \
	"v_cmp_lt_u32_e64 %[nlb], %[kf], " B2_ "\n\t"
#define MPC_RB_MERGE_ARITH(A0_, A1_, A2_, A3_, B0_, B1_, B2_, B3_, CURA_, CURB_, NXTA_, NXTB_) \
	"v_cndmask_b32_e64 %[t0], 0, " B1_ ", %[e01]\n\t"                                                                    \
	"v_cndmask_b32_e64 %[t1], 0, " B1_ ", %[e11]\n\t"                                                                    \
	"v_cndmask_b32_e64 %[t0], %[t0], " B0_ ", %[e00]\n\t" /* of two equal columns in B the first wins */                 \
	"v_cndmask_b32_e64 %[t1], %[t1], " B0_ ", %[e10]\n\t"                                                                \
	"v_mul_f32_e32 %[t0], " A0_ ", %[t0]\n\t"                                                                            \
	"v_mul_f32_e32 %[t1], " A1_ ", %[t1]\n\t"                                                                            \
	"v_add_f32_e32 %[sum], %[sum], %[t0]\n\t" /* z ascending: relaxflat.cpp:27, product rounded, then added */           \
	"v_add_f32_e32 %[sum], %[sum], %[t1]\n\t"
#define MPC_RB_MERGE_TAIL(A0_, A1_, A2_, A3_, B0_, B1_, B2_, B3_, CURA_, CURB_, NXTA_, NXTB_) \
	"s_orn2_b64 %[nla], %[nla], %[ada]\n\t"                                                                              \
	"s_orn2_b64 %[nlb], %[nlb], %[adb]\n\t"                                                                              \
	"s_and_b64 %[nla], %[nla], %[nlb]\n\t"                                                                               \
	"s_and_b64 exec, exec, %[nla]\n\t" /* the lanes that go on; scc = any */                                             \
	"s_cbranch_scc0 .Lrv_done_%=\n\t"                                                                                    \
	"s_mov_b64 %[nlb], exec\n\t"                                                                                         \
	"s_and_b64 exec, %[nlb], %[ada]\n\t"                                                                                 \
	"v_add_u32_sdwa %[ia], %[ia], " A2_ " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"        \
	"ds_read_b128 " CURA_ ", %[ia]\n\t" /* only the lanes whose row advanced read its next block: the others keep theirs */ \
	"s_and_b64 exec, %[nlb], %[adb]\n\t"                                                                                 \
	"v_add_u32_sdwa %[ib], %[ib], " B2_ " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"        \
	"ds_read_b128 " CURB_ ", %[ib]\n\t"                                                                                  \
	"s_mov_b64 exec, %[nlb]\n\t"                                                                                         \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
	"s_branch .Lrv_step_%=\n"                                                                                            \
	".Lrv_done_%=:\n\t"                                                                                                  \
	"s_mov_b64 exec, %[sv]"
#ifdef MPC_RB_NOARITH /* measurement build only (MPCGPU_RELAX_DIAG=5): the walk without its selects, products and sums - wrong results */
#define MPC_RB_MERGE_ASM(...) MPC_RB_MERGE_HEAD(__VA_ARGS__) MPC_RB_MERGE_TAIL(__VA_ARGS__)
#else
#define MPC_RB_MERGE_ASM(...) MPC_RB_MERGE_HEAD(__VA_ARGS__) MPC_RB_MERGE_ARITH(__VA_ARGS__) MPC_RB_MERGE_TAIL(__VA_ARGS__)
#endif
	// merges the slot whose first blocks are in set SET (first-block addresses + X bias in ia, Y first-block address in ib) onto
	// sum; reads the blocks at nia, nib into the other set and gathers lane nidx/4 of bias_y for the next slot
	template <int SET> __device__ __forceinline__ void merge(float &sum, unsigned ia, unsigned ib, unsigned nia, unsigned nib, unsigned nidx, unsigned bias_y)
	{
		mpc_u64s sv, e00, e01, e10, e11, ada, adb, nla, nlb;
		float t0, t1;
		const unsigned kf = 0xffffu;
		if (SET == 0)
			asm volatile(MPC_RB_MERGE_ASM("v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v[24:27]", "v[28:31]", "v[32:35]", "v[36:39]")
				: [sum] "+v"(sum), [ia] "+v"(ia), [ib] "+v"(ib), [hb] "+{v40}"(hb), "+{v[24:27]}"(a0), "+{v[28:31]}"(b0), "=&{v[32:35]}"(a1), "=&{v[36:39]}"(b1),
				  [t0] "=&v"(t0), [t1] "=&v"(t1), [sv] "=&s"(sv), [e00] "=&s"(e00), [e01] "=&s"(e01), [e10] "=&s"(e10), [e11] "=&s"(e11),
				  [ada] "=&s"(ada), [adb] "=&s"(adb), [nla] "=&s"(nla), [nlb] "=&s"(nlb)
				: [nia] "v"(nia), [nib] "v"(nib), [nidx] "v"(nidx), [by] "v"(bias_y), [kf] "s"(kf)
				: "vcc", "scc", "memory");
		else
			asm volatile(MPC_RB_MERGE_ASM("v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v[32:35]", "v[36:39]", "v[24:27]", "v[28:31]")
				: [sum] "+v"(sum), [ia] "+v"(ia), [ib] "+v"(ib), [hb] "+{v40}"(hb), "+{v[32:35]}"(a1), "+{v[36:39]}"(b1), "=&{v[24:27]}"(a0), "=&{v[28:31]}"(b0),
				  [t0] "=&v"(t0), [t1] "=&v"(t1), [sv] "=&s"(sv), [e00] "=&s"(e00), [e01] "=&s"(e01), [e10] "=&s"(e10), [e11] "=&s"(e11),
				  [ada] "=&s"(ada), [adb] "=&s"(adb), [nla] "=&s"(nla), [nlb] "=&s"(nlb)
				: [nia] "v"(nia), [nib] "v"(nib), [nidx] "v"(nidx), [by] "v"(bias_y), [kf] "s"(kf)
				: "vcc", "scc", "memory");
	}
#undef MPC_RB_MERGE_ASM
#undef MPC_RB_MERGE_HEAD
#undef MPC_RB_MERGE_ARITH
#undef MPC_RB_MERGE_TAIL
};
#define MPC_RB_HAVE_ASM 1
// ---- the DIRECT-INDEX merge of relax_band_kernel (kernels_relaxb.h: MpcRbWinCxx is its C++ statement). The X row is walked block by
// block as before; the Y row is not walked: it is a WINDOW record's row — descriptor word c0 | span << 12 | off << 17 and
// values val[off + j] for column c0 + j, with a 0.0f guard at off + span (StoreParams::win) — and every X entry (z, P) looks its
// partner up: value = val[off + min(z - c0, span)]. Per X block: 2 x (sub, min, address, ds_read_b32, mul, add) instead of a
// 2 x 2 compare-select step on two block lists, and the number of steps is that of the X row alone. Registers: set 0 = X block
// v[24:27] + descriptor pair v[28:29], set 1 = v[32:35] + v[36:37]; v40 = the value base of the NEXT slot's Y record (in flight).
struct MpcRbWinAsm {
	static constexpr bool WINDOW = true;
	mpc_uint4v a0, a1;
	unsigned d0, d1;
	unsigned hb;
	__device__ __forceinline__ void load(unsigned ia, unsigned yd, unsigned hb_addr)
	{
		asm volatile("ds_read_b128 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5"
			: "={v[24:27]}"(a0), "={v28}"(d0), "={v40}"(hb) : "v"(ia), "v"(yd), "v"(hb_addr) : "memory");
	}
	__device__ __forceinline__ void drain() { asm volatile("s_waitcnt lgkmcnt(0)" : "+{v[24:27]}"(a0), "+{v28}"(d0), "+{v[32:35]}"(a1), "+{v36}"(d1), "+{v40}"(hb) : : "memory"); }
#define MPC_RW_MERGE_ASM(A0_, A1_, A2_, A3_, D0_, CURA_, NXTA_, NXTD_)                                                   \
	"s_waitcnt lgkmcnt(0)\n\t" /* this slot's X block, descriptor and value base have landed */                          \
	"ds_read_b128 " NXTA_ ", %[nia]\n\t"                                                                                 \
	"ds_read_b32 " NXTD_ ", %[nyd]\n\t"                                                                                  \
	"v_and_b32_e32 %[c0], 0xfff, " D0_ "\n\t"                  /* descriptor: c0 | span << 12 | off << 17 */             \
	"v_bfe_u32 %[sp], " D0_ ", 12, 5\n\t"                      /* span: where the guard sits */                          \
	"v_lshrrev_b32_e32 %[vb], 17, " D0_ "\n\t"                                                                           \
	"v_lshl_add_u32 %[vb], %[vb], 2, %[hb]\n\t" /* LDS address of the row's first value */                               \
	"ds_bpermute_b32 %[hb], %[nidx], %[by]\n\t" /* the next slot's value base, into the register this slot's just left */ \
	"s_mov_b64 %[sv], exec\n\t"                                                                                          \
	"v_cmp_eq_u32_e32 vcc, 31, %[sp]\n\t"       /* the escape: a row wider than the field — its span from the next row's offset */ \
	"s_cbranch_vccz .Lrv_narrow_%=\n\t"                                                                                  \
	"s_mov_b64 exec, vcc\n\t"                                                                                            \
	"ds_read_b32 %[j0], %[yd] offset:4\n\t"                                                                              \
	"v_lshrrev_b32_e32 %[j1], 17, " D0_ "\n\t"                                                                           \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
	"v_lshrrev_b32_e32 %[j0], 17, %[j0]\n\t"                                                                             \
	"v_sub_u32_e32 %[j0], %[j0], %[j1]\n\t"                                                                              \
	"v_add_u32_e32 %[sp], -1, %[j0]\n\t"                                                                                 \
	"s_mov_b64 exec, %[sv]\n"                                                                                            \
	".Lrv_narrow_%=:\n"                                                                                                  \
	".Lrv_step_%=:\n\t"                                                                                                  \
	"v_sub_u32_sdwa %[j0], " A2_ ", %[c0] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"        \
	"v_sub_u32_e32 %[j1], " A3_ ", %[c0]\n\t"                                                                            \
	"v_min_u32_e32 %[j0], %[j0], %[sp]\n\t" /* a column outside the window (also: below c0, wrapped) reads the guard */    \
	"v_min_u32_e32 %[j1], %[j1], %[sp]\n\t"                                                                              \
	"v_lshl_add_u32 %[j0], %[j0], 2, %[vb]\n\t"                                                                          \
	"v_lshl_add_u32 %[j1], %[j1], 2, %[vb]\n\t"                                                                          \
	"ds_read_b32 %[t0], %[j0]\n\t"                                                                                       \
	"ds_read_b32 %[t1], %[j1]\n\t"                                                                                       \
	"v_cmp_lt_u32_e64 %[more], %[kf], " A2_ "\n\t" /* the X row has a next block */                                      \
	"s_and_b64 %[more], %[more], exec\n\t"                                                                               \
	"s_cbranch_scc0 .Lrv_last_%=\n\t"                                                                                    \
	"v_mov_b32_e32 %[p0], " A0_ "\n\t" /* the block's probabilities outlive the read of its successor */                 \
	"v_mov_b32_e32 %[p1], " A1_ "\n\t"                                                                                   \
	"s_mov_b64 %[keep], exec\n\t"                                                                                        \
	"s_mov_b64 exec, %[more]\n\t"                                                                                        \
	"v_add_u32_sdwa %[ia], %[ia], " A2_ " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"        \
	"ds_read_b128 " CURA_ ", %[ia]\n\t"                                                                                  \
	"s_mov_b64 exec, %[keep]\n\t"                                                                                        \
	"s_waitcnt lgkmcnt(1)\n\t" /* LDS returns in order: both values are here, the next block may still be on its way */    \
	"v_mul_f32_e32 %[t0], %[p0], %[t0]\n\t"                                                                              \
	"v_mul_f32_e32 %[t1], %[p1], %[t1]\n\t"                                                                              \
	"v_add_f32_e32 %[sum], %[sum], %[t0]\n\t" /* z ascending: relaxflat.cpp:27, product rounded, then added */           \
	"v_add_f32_e32 %[sum], %[sum], %[t1]\n\t"                                                                            \
	"s_mov_b64 exec, %[more]\n\t" /* only the rows that go on */                                                         \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
	"s_branch .Lrv_step_%=\n"                                                                                            \
	".Lrv_last_%=:\n\t"                                                                                                  \
	"s_waitcnt lgkmcnt(0)\n\t"                                                                                           \
	"v_mul_f32_e32 %[t0], " A0_ ", %[t0]\n\t"                                                                            \
	"v_mul_f32_e32 %[t1], " A1_ ", %[t1]\n\t"                                                                            \
	"v_add_f32_e32 %[sum], %[sum], %[t0]\n\t"                                                                            \
	"v_add_f32_e32 %[sum], %[sum], %[t1]\n\t"                                                                            \
	"s_mov_b64 exec, %[sv]"
	// merges the slot whose X block and descriptor are in set SET (ia: address of the X row's first block + the X record's hop
	// bias) onto sum; reads the next slot's X block (nia) and descriptor (nyd) into the other set and gathers its value base
	template <int SET> __device__ __forceinline__ void merge(float &sum, unsigned ia, unsigned yd, unsigned nia, unsigned nyd, unsigned nidx, unsigned bias_y)
	{
		mpc_u64s sv, more, keep;
		float t0, t1, p0, p1;
		unsigned c0, sp, vb, j0, j1;
		const unsigned kf = 0xffffu;
		if (SET == 0)
			asm volatile(MPC_RW_MERGE_ASM("v24", "v25", "v26", "v27", "v28", "v[24:27]", "v[32:35]", "v36")
				: [sum] "+v"(sum), [ia] "+v"(ia), [hb] "+{v40}"(hb), "+{v[24:27]}"(a0), "+{v28}"(d0), "=&{v[32:35]}"(a1), "=&{v36}"(d1),
				  [t0] "=&v"(t0), [t1] "=&v"(t1), [p0] "=&v"(p0), [p1] "=&v"(p1), [c0] "=&v"(c0), [sp] "=&v"(sp), [vb] "=&v"(vb), [j0] "=&v"(j0), [j1] "=&v"(j1),
				  [sv] "=&s"(sv), [more] "=&s"(more), [keep] "=&s"(keep)
				: [nia] "v"(nia), [nyd] "v"(nyd), [yd] "v"(yd), [nidx] "v"(nidx), [by] "v"(bias_y), [kf] "s"(kf)
				: "vcc", "scc", "memory");
		else
			asm volatile(MPC_RW_MERGE_ASM("v32", "v33", "v34", "v35", "v36", "v[32:35]", "v[24:27]", "v28")
				: [sum] "+v"(sum), [ia] "+v"(ia), [hb] "+{v40}"(hb), "+{v[32:35]}"(a1), "+{v36}"(d1), "=&{v[24:27]}"(a0), "=&{v28}"(d0),
				  [t0] "=&v"(t0), [t1] "=&v"(t1), [p0] "=&v"(p0), [p1] "=&v"(p1), [c0] "=&v"(c0), [sp] "=&v"(sp), [vb] "=&v"(vb), [j0] "=&v"(j0), [j1] "=&v"(j1),
				  [sv] "=&s"(sv), [more] "=&s"(more), [keep] "=&s"(keep)
				: [nia] "v"(nia), [nyd] "v"(nyd), [yd] "v"(yd), [nidx] "v"(nidx), [by] "v"(bias_y), [kf] "s"(kf)
				: "vcc", "scc", "memory");
	}
#undef MPC_RW_MERGE_ASM
};
#define MPC_RW_HAVE_ASM 1
// A pointer through which wave-uniform reads of memory that this kernel never writes become scalar loads (s_load_dword*:
// SGPR results, no VGPRs, no vmcnt): the constant address space. (Through a plain global pointer the compiler has to assume
// the kernel's own stores may alias and issues vector loads.)
typedef const __attribute__((address_space(4))) unsigned *mpc_const_u32p;
#define MPC_CONST_U32(p) ((mpc_const_u32p)(unsigned long long)(p))
// no instruction is scheduled across this point (keeps the loads of unrolled prologue / epilogue iterations from being hoisted
// on top of each other: register pressure, not speed, decides those parts)
#define MPC_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// Between LDS writes of one wavefront and its own later LDS reads of what other lanes wrote, in a workgroup that IS one
// wavefront: the hardware executes a wave's LDS instructions in order, so no barrier and no wait is needed — only the compiler
// must not move accesses across this point.
#define MPC_WAVE_LDS_ORDER() __builtin_amdgcn_wave_barrier()
// free-running clock for the phase timers of the diagnostics (s_memrealtime: 100 MHz on gfx950)
__device__ __forceinline__ unsigned long long mpc_clock() { return wall_clock64(); }
// value held by the first active lane, as a wave-uniform scalar (v_readfirstlane_b32 -> SGPR)
__device__ __forceinline__ unsigned mpc_wave_first(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#endif
#include <stdint.h>

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define MPC_WAVE 64
