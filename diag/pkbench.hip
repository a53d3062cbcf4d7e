// diag/pkbench.hip — issue cost of the VALU / LDS instruction forms the fb and relax kernels are made of, measured on
// this GPU: 8 waves per SIMD, 16 independent chains per lane, 64 instructions per loop iteration. Answers "does an
// explicit packed-FP32 formulation of LOG_ADD buy throughput on gfx950"  and prices min/max,
// selects, compares, conversions and DPP moves against a plain v_add_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

// scalar-register form: one asm string with %0 = chain value (in/out), %1 = second operand
#define DEF_KERNEL(NAME, ASM)                                                                     \
	__global__ void __launch_bounds__(256) k_##NAME(float *out, int iters, float seed)            \
	{                                                                                             \
		float a[16];                                                                              \
		const float b = seed + threadIdx.x * 1e-7f;                                               \
		_Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed * (i + 1);                     \
		asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_mov_b64 s[2:3], vcc" ::"v"(a[3]), "v"(b) : "vcc", "s2", "s3"); \
		for (int it = 0; it < iters; ++it) {                                                      \
			REP16(ONE_##NAME) REP16(ONE_##NAME) REP16(ONE_##NAME) REP16(ONE_##NAME)               \
		}                                                                                         \
		float s = 0;                                                                              \
		_Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                                 \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
	}
#define ONE_add(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_sub(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_mul(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_fma(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define ONE_min(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_max(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_minu(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_med3(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define ONE_add_abs(i) asm volatile("v_add_f32_e64 %0, |%0|, -%1" : "+v"(a[i]) : "v"(b));
#define ONE_floor(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define ONE_ceil(i) asm volatile("v_ceil_f32 %0, %0" : "+v"(a[i]));
#define ONE_cvti(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define ONE_cvtu(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
#define ONE_lshl(i) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(a[i]));
#define ONE_and(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_addu(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_lshladd(i) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));
#define ONE_bfe(i) asm volatile("v_bfe_u32 %0, %0, 16, 16" : "+v"(a[i]));
#define ONE_mov(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_cnd_vcc(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define ONE_cnd_sgpr(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[2:3]" : "+v"(a[i]) : "v"(b));
#define ONE_cmp_vcc(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_cmp_sgpr(i) asm volatile("v_cmp_lt_f32_e64 s[4:5], %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "s4", "s5");
#define ONE_cmp_cnd(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_cmpu_sdwa(i) asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:WORD_0 src1_sel:WORD_1\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_dpp_shr(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define ONE_dpp_row(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define ONE_add_dpp(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define ONE_readlane(i) asm volatile("v_readfirstlane_b32 s4, %0\n\tv_add_f32 %0, s4, %0" : "+v"(a[i]) : : "s4");
#define ONE_cmp_nop_cnd(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_smov_cnd(i) asm volatile("s_mov_b64 vcc, s[2:3]\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_cmp_3_cnd(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_cmp_cnd2(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_cmps_cnds(i) asm volatile("v_cmp_lt_f32_e64 s[4:5], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(a[i]) : "v"(b) : "s4", "s5");
#define ONE_cnd_sdwa(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(b) : "vcc");
DEF_KERNEL(cmp_nop_cnd, "") DEF_KERNEL(smov_cnd, "") DEF_KERNEL(cmp_3_cnd, "") DEF_KERNEL(cmp_cnd2, "") DEF_KERNEL(cmps_cnds, "") DEF_KERNEL(cnd_sdwa, "")
DEF_KERNEL(add, "") DEF_KERNEL(sub, "") DEF_KERNEL(mul, "") DEF_KERNEL(fma, "") DEF_KERNEL(min, "") DEF_KERNEL(max, "")
DEF_KERNEL(minu, "") DEF_KERNEL(med3, "") DEF_KERNEL(add_abs, "") DEF_KERNEL(floor, "") DEF_KERNEL(ceil, "") DEF_KERNEL(cvti, "")
DEF_KERNEL(cvtu, "") DEF_KERNEL(lshl, "") DEF_KERNEL(and, "") DEF_KERNEL(addu, "") DEF_KERNEL(lshladd, "") DEF_KERNEL(bfe, "")
DEF_KERNEL(mov, "") DEF_KERNEL(cnd_vcc, "") DEF_KERNEL(cnd_sgpr, "") DEF_KERNEL(cmp_vcc, "") DEF_KERNEL(cmp_sgpr, "")
DEF_KERNEL(cmp_cnd, "") DEF_KERNEL(cmpu_sdwa, "") DEF_KERNEL(dpp_shr, "") DEF_KERNEL(dpp_row, "") DEF_KERNEL(add_dpp, "") DEF_KERNEL(readlane, "")

// packed forms: chains of register pairs
#define DEF_PK(NAME)                                                                              \
	__global__ void __launch_bounds__(256) k_##NAME(float *out, int iters, float seed)            \
	{                                                                                             \
		v2f p[16];                                                                                \
		const float b = seed + threadIdx.x * 1e-7f;                                               \
		v2f pb = {b, b + 1e-7f};                                                                  \
		_Pragma("unroll") for (int i = 0; i < 16; ++i) p[i] = v2f{seed * (i + 1), seed * (i + 1) + 0.5f}; \
		for (int it = 0; it < iters; ++it) {                                                      \
			REP16(ONE_##NAME) REP16(ONE_##NAME) REP16(ONE_##NAME) REP16(ONE_##NAME)               \
		}                                                                                         \
		float s = 0;                                                                              \
		_Pragma("unroll") for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;                      \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                           \
	}
#define ONE_pk_add(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define ONE_pk_mul(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define ONE_pk_fma(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pb));
#define ONE_pk_mov(i) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "+v"(p[i]) : "v"(pb));
DEF_PK(pk_add) DEF_PK(pk_mul) DEF_PK(pk_fma) DEF_PK(pk_mov)

// LDS: 16-byte reads of a 16-entry table at lane-dependent entries (the LOG_ADD coefficient fetch), and of random blocks of a
// 64 KB region (the relax kernel's row gathers); 8-byte and 4-byte forms for comparison. One add per read keeps a dependency.
template <int BYTES, int SPAN>
__global__ void __launch_bounds__(256) k_lds(float *out, int iters, float seed)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	for (int q = threadIdx.x; q < SPAN / 4; q += 256) ((float *)smem)[q] = seed * q;
	__syncthreads();
	unsigned idx = (threadIdx.x * 2654435761u) >> 7;
	float s = 0;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 16; ++u) {
			const unsigned off = ((idx + u * 977u) * 16u) & (SPAN - 1);
			if (BYTES == 16) { const float4 v = *(const float4 *)(smem + off); s += v.x + v.w; }
			if (BYTES == 8) { const float2 v = *(const float2 *)(smem + off); s += v.x + v.y; }
			if (BYTES == 4) { const float v = *(const float *)(smem + off); s += v; }
		}
		idx = idx * 1664525u + 1013904223u;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float *d_out;
static int g_cus;
static double g_clk;
template <class K> static void run(const char *name, K kern, double inst_per_iter, size_t smem = 0, int iters = 20000)
{
	const int blocks = g_cus * 8; // 8 blocks of 4 waves per CU = 8 waves per SIMD
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, 0, d_out, 100, 1.0f);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, 0, d_out, iters, 1.0f);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms;
	(void)hipEventElapsedTime(&ms, e0, e1);
	const double winst = (double)blocks * 4 * iters * inst_per_iter; // wave-instructions
	const double per_simd_per_s = winst / (g_cus * 4.0) / (ms * 1e-3);
	printf("%-22s %8.3f ms  %6.2f nominal cycles per wave-instruction per SIMD  (%5.2fx v_add_f32)\n", name, ms, g_clk / per_simd_per_s,
		g_clk / per_simd_per_s / 2.48);
}

int main()
{
	hipDeviceProp_t pr;
	(void)hipGetDeviceProperties(&pr, 0);
	g_cus = pr.multiProcessorCount;
	g_clk = pr.clockRate * 1e3;
	(void)hipMalloc(&d_out, (size_t)g_cus * 8 * 256 * 4);
	printf("%s, %d CUs, %.0f MHz nominal; two-instruction rows (cmp+add, readlane+add) count both\n", pr.gcnArchName, g_cus, g_clk * 1e-6);
#define RUN(NAME, N) run(#NAME, k_##NAME, N);
	RUN(add, 64) RUN(sub, 64) RUN(mul, 64) RUN(fma, 64) RUN(min, 64) RUN(max, 64) RUN(minu, 64) RUN(med3, 64) RUN(add_abs, 64)
	RUN(floor, 64) RUN(ceil, 64) RUN(cvti, 64) RUN(cvtu, 64) RUN(lshl, 64) RUN(and, 64) RUN(addu, 64) RUN(lshladd, 64) RUN(bfe, 64)
	RUN(mov, 64) RUN(cnd_vcc, 64) RUN(cnd_sgpr, 64) RUN(cmp_vcc, 128) RUN(cmp_sgpr, 128) RUN(cmp_cnd, 128) RUN(cmpu_sdwa, 128)
	RUN(dpp_shr, 64) RUN(dpp_row, 64) RUN(add_dpp, 64) RUN(readlane, 128)
	RUN(cmp_nop_cnd, 128) RUN(smov_cnd, 64) RUN(cmp_3_cnd, 320) RUN(cmp_cnd2, 192) RUN(cmps_cnds, 128) RUN(cnd_sdwa, 128)
	RUN(pk_add, 64) RUN(pk_mul, 64) RUN(pk_fma, 64) RUN(pk_mov, 64)
	run("ds_read_b128 table256B", k_lds<16, 256>, 16, 256, 4000);
	run("ds_read_b128 rand64KB", k_lds<16, 65536>, 16, 65536, 4000);
	run("ds_read_b64 rand64KB", k_lds<8, 65536>, 16, 65536, 4000);
	run("ds_read_b32 rand64KB", k_lds<4, 65536>, 16, 65536, 4000);
	return 0;
}
