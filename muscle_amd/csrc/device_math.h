// device_math.h — float32 log-space arithmetic of the reference, restated op-for-op for the
// device. Compiled with -ffp-contract=off: the reference build (x86-64 baseline) has no FMA, so
// every multiply and add below must round on its own.
#pragma once
#include "mpc_platform.h"

#define MPC_LOG_ZERO (-2e20f)              /* scoretype.h:89 */
#define MPC_LOG_UNDERFLOW 7.5f             /* scoretype.h:96 */
#define MPC_MIN_SPARSE_PROB 0.01f          /* mysparsemx.h:3 */

// LOGEXP1, scoretype.h:100-109: four cubics in Horner form on (..1], (1,2.5], (2.5,4.5], (4.5,7.5].
// Coefficients are selected first, then ONE Horner chain is evaluated (same ops as the taken branch).
__device__ __forceinline__ float mpc_logexp1(float d)
{
	const bool a = d <= 1.00f, b = d <= 2.50f, c = d <= 4.50f;
	const float c3 = a ? -0.009350833524763f : (b ? -0.014532321752540f : (c ? -0.004605031767994f : -0.000458661602210f));
	const float c2 = a ? 0.130659527668286f : (b ? 0.139942324101744f : (c ? 0.063427417320019f : 0.009695946122598f));
	const float c1 = a ? 0.498799810682272f : (b ? 0.495635523139337f : (c ? 0.695956496475118f : 0.930734667215156f));
	const float c0 = a ? 0.693203116424741f : (b ? 0.692140569840976f : (c ? 0.514272634594009f : 0.168037164329057f));
	return ((c3 * d + c2) * d + c1) * d + c0;
}

// LOG_ADD, scoretype.h:119-124. The reference returns hi when lo == LOG_ZERO or hi-lo >= 7.5, else
// LOGEXP1(hi-lo)+lo. The lo == LOG_ZERO test is implied here: if hi > LOG_ZERO then hi-lo >= 1.7e13
// (one ulp of 2e20) >= 7.5; if hi == lo == LOG_ZERO then LOGEXP1(0)+LOG_ZERO rounds back to
// LOG_ZERO == hi. Operands on this path are either exactly LOG_ZERO (LOG_ZERO + score == LOG_ZERO),
// sums of two such (-4e20, only in the total fold) or ordinary scores, so the result is bit-identical
// (tests/test_gpu_parity.py::test_log_add_bits and the emulator tests check it against the oracle).
__device__ __forceinline__ float mpc_la2(float x, float y)
{
	const float lo = x < y ? x : y;
	const float hi = x < y ? y : x;
	const float d = hi - lo;
	const float p = mpc_logexp1(d) + lo;
	return d >= MPC_LOG_UNDERFLOW ? hi : p;
}

// ---- hot-path form of LOG_ADD (kernels_fb.h): same values, fewer instructions ------------------
// The four coefficient sets of LOGEXP1 are picked with 12 selects above. The interval bounds
// (1, 2.5, 4.5, 7.5) are all multiples of 0.5, so ceil(2d) identifies the interval exactly
// ("d <= bound" <=> ceil(2d) <= 2*bound): one 16-entry LDS table indexed by ceil(2d) - 1, fetched with a
// single 16-byte read, replaces the selects (mpc_coef_offset below). lo/hi use
// min/max (no NaNs on this path; equal operands give d = 0 either way). The Horner chain and every
// rounding step are unchanged, so results are bit-identical to mpc_la2.
struct __attribute__((aligned(16))) MpcCoef { float c3, c2, c1, c0; };
#define MPC_COEF_ENTRIES 16

// entry q of the table = coefficient set of the interval that holds d with ceil(2d) - 1 == q (q = 0 also for d == 0)
__device__ __forceinline__ void mpc_coef_table_init(MpcCoef *tab, int q)
{
	MpcCoef c;
	if (q <= 1) { c.c3 = -0.009350833524763f; c.c2 = 0.130659527668286f; c.c1 = 0.498799810682272f; c.c0 = 0.693203116424741f; }
	else if (q <= 4) { c.c3 = -0.014532321752540f; c.c2 = 0.139942324101744f; c.c1 = 0.495635523139337f; c.c0 = 0.692140569840976f; }
	else if (q <= 8) { c.c3 = -0.004605031767994f; c.c2 = 0.063427417320019f; c.c1 = 0.695956496475118f; c.c0 = 0.514272634594009f; }
	else { c.c3 = -0.000458661602210f; c.c2 = 0.009695946122598f; c.c1 = 0.930734667215156f; c.c0 = 0.168037164329057f; }
	tab[q] = c;
}

// Byte offset of d's coefficient set in the table: 16 * (ceil(2d) - 1), from three cheap instructions (measured on
// MI355X, diag/pkbench: floor / min / cvt / shift issue at 0.6 of the add rate, integer add and `and` at the full rate).
// bits(d) + (5 << 23) are the bits of 32d (d >= 0, normal), minus one: the float just below it, p = prev(32d). Then
// trunc(p) = 16 * floor(prev(2d)) + (a fraction part < 16), and floor(prev(2d)) = ceil(2d) - 1 whenever 2d > 0 (prev()
// only matters when 2d is an integer, which is exactly when the two differ) — so masking trunc(p) with 0xf0 leaves
// 16 * (ceil(2d) - 1) for every d in (0, 8]. d == 0 and denormal d give a tiny p (offset 0: the first interval, as it
// should be); d > 8 gives some in-range offset whose result the caller discards (d >= 7.5 returns hi); the conversion
// saturates for the huge d of LOG_ZERO operands (v_cvt_u32_f32: 0xffffffff).
__device__ __forceinline__ u32 mpc_coef_offset(float d)
{
	const float p = __uint_as_float(__float_as_uint(d) + 0x027fffffu);
	return mpc_cvt_u32_sat(p) & 0xf0u;
}

__device__ __forceinline__ float mpc_la2t(float x, float y, const MpcCoef *tab)
{
	const float lo = fminf(x, y);
	const float hi = fmaxf(x, y);
	const float d = hi - lo;
	const MpcCoef c = *(const MpcCoef *)((const unsigned char *)tab + mpc_coef_offset(d));
	const float p = ((c.c3 * d + c.c2) * d + c.c1) * d + c.c0 + lo;
	return d >= MPC_LOG_UNDERFLOW ? hi : p;
}

__device__ __forceinline__ float mpc_la5t(float a, float b, float c, float d, float e, const MpcCoef *tab)
{
	return mpc_la2t(a, mpc_la2t(b, mpc_la2t(c, mpc_la2t(d, e, tab), tab), tab), tab);
}

// 5-ary LOG_ADD nests to the right, scoretype.h:136-139
__device__ __forceinline__ float mpc_la5(float a, float b, float c, float d, float e)
{
	return mpc_la2(a, mpc_la2(b, mpc_la2(c, mpc_la2(d, e))));
}

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, N=32 table + cubic in double, one rounding to
// float) — the libm call at calcposteriorflat.cpp:20. use_fma selects the x86-64 ifunc variant the
// host's libm resolves to (see oracle/mpc_oracle.c:orc_expf_emul for provenance of the contraction
// pattern). Only called for logf(0.01f) <= x < 0, so no special cases are needed.
__device__ static const u64 MPC_EXP2F_TAB[32] = {
	0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
	0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
	0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
	0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
	0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
	0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
	0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
	0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

__device__ __forceinline__ float mpc_expf(float x, int use_fma)
{
	const double InvLn2N = __longlong_as_double(0x40471547652b82feull);
	const double SHIFT = __longlong_as_double(0x4338000000000000ull);
	const double C0 = __longlong_as_double(0x3ebc6af84b912394ull);
	const double C1 = __longlong_as_double(0x3f2ebfce50fac4f3ull);
	const double C2 = __longlong_as_double(0x3f962e42ff0c52d6ull);
	const double xd = (double)x;
	double kd, r, z, y, r2;
	u64 ki;
	if (use_fma) {
		kd = fma(InvLn2N, xd, SHIFT);
		ki = (u64)__double_as_longlong(kd);
		kd = kd - SHIFT;
		r = fma(InvLn2N, xd, -kd);
		z = fma(r, C0, C1);
		r2 = r * r;
		y = fma(r, C2, 1.0);
		y = fma(z, r2, y);
	} else {
		z = InvLn2N * xd;
		kd = z + SHIFT;
		ki = (u64)__double_as_longlong(kd);
		kd = kd - SHIFT;
		r = z - kd;
		z = C0 * r + C1;
		r2 = r * r;
		y = C2 * r + 1.0;
		y = z * r2 + y;
	}
	u64 t = MPC_EXP2F_TAB[ki & 31];
	t += ki << (52 - 5);
	y = y * __longlong_as_double((long long)t);
	return (float)y;
}

// calcposteriorflat.cpp:16-22 for a cell whose Score passed the MIN_SPARSE_SCORE test
__device__ __forceinline__ float mpc_score_to_prob(float score, int use_fma)
{
	return score >= 0.0f ? 1.0f : mpc_expf(score, use_fma);
}
