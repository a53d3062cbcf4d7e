// mpc_platform.h — the only place that knows whether we are compiled by hipcc for gfx950 (the
// product) or by g++ against tests/emu/hip_emu.h (MPC_EMU: the test-only SIMT emulator used to
// validate kernel logic on a GPU-less box; never shipped, never a fallback).
#pragma once
#ifdef MPC_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define MPC_LAUNCH(kern, grid, block, smem, stream, ...) \
	hipLaunchKernelGGL(kern, dim3(grid), dim3(block), (smem), (stream), __VA_ARGS__)
#define MPC_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// optimisation barrier on one VGPR value (no code): stops hoisting of what is derived from it
#define MPC_OPAQUE(v) asm volatile("" : "+v"(v))
// value held by the first active lane, as a wave-uniform scalar (v_readfirstlane_b32 -> SGPR)
__device__ __forceinline__ unsigned mpc_wave_first(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
#endif
#include <stdint.h>

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define MPC_WAVE 64
