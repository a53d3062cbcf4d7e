// kernels_aln.h — posterior-DP alignment with traceback on the device.
//
// Replaces CalcAlnFlat (calcalnflat.cpp:6-46) + TraceBackFlat (tracebackflat.cpp:3-37) with the
// tie order of Best3 (best3.h:5-28): per cell, with B = S(i-1,j-1) + Post(i-1,j-1),
// X = S(i-1,j), Y = S(i,j-1):   if B >= X: (B >= Y ? 'B' : 'Y')  else  (X >= Y ? 'X' : 'Y').
//
// Mapping. Rows are sequential; inside a row S(i,j) = max(T_j, S(i,j-1)) with
// T_j = max(B_j, X_j) and S(i,0) = 0, i.e. a prefix maximum over j — max is exact and
// associative, so a workgroup-wide scan reproduces the sequential recurrence bit for bit, and the
// traceback letter of a cell only needs T_j, its own letter (B if B >= X else X) and the final
// S(i,j-1): T_j >= S(i,j-1) ? letter(T_j) : 'Y'. One 1024-thread workgroup per alignment, thread t
// owns columns [t*C, t*C+C); two DP rows ping-pong in LDS; TB is written to HBM as the
// reference's own bytes ('B','X','Y', row 0 = 'Y', column 0 = 'X': calcalnflat.cpp:15-25); the
// walk back from (LX,LY) is pointer chasing and is done by one lane, then the path is reversed by
// the whole workgroup. This is a latency-bound kernel (2 barriers per row): it exists so that the
// progressive stage can stay on the device, not because one call beats a CPU.
#pragma once
#include "kernels_fb.h" // candidate key layout (dense_post_kernel)
#include "device_math.h"


struct AlnParams {
	const float *post; // LX*LY, row-major
	u32 LX, LY;
	char *tb;          // (LX+1)*(LY+1)
	char *rev;         // LX+LY scratch (path in reverse)
	char *path;        // LX+LY out
	u32 *pathlen;      // out
	float *score;      // out
};

__global__ void __launch_bounds__(MPC_ALN_THREADS) calc_aln_kernel(AlnParams p)
{
	MPC_DYN_SMEM(smem_raw);
	const u32 LX = p.LX, LY = p.LY, W = LY + 1;
	float *rows = (float *)smem_raw;        // 2*W
	float *wmax = rows + 2 * (u64)W;        // one per wave
	u32 *s_n = (u32 *)(wmax + MPC_ALN_THREADS / 64);
	const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
	const u32 C = (W + MPC_ALN_THREADS - 1) / MPC_ALN_THREADS;
	const u32 j0 = tid * C, j1 = (j0 + C < W) ? j0 + C : W; // my columns [j0, j1)
	float *oldr = rows, *newr = rows + W;
	for (u32 j = j0; j < j1; ++j) { oldr[j] = 0.0f; p.tb[j] = 'Y'; } // calcalnflat.cpp:15-19
	__syncthreads();
	for (u32 i = 1; i <= LX; ++i) {
		const float *prow = p.post + (u64)(i - 1) * LY;
		char *tbrow = p.tb + (u64)i * W;
		// pass 1: maximum of T over my columns
		float run = 0.0f; // T_j >= 0 always (X >= 0), and S(i,0) = 0
		for (u32 j = (j0 == 0 ? 1u : j0); j < j1; ++j) {
			const float B = oldr[j - 1] + prow[j - 1];
			const float X = oldr[j];
			const float T = B >= X ? B : X;
			run = T >= run ? T : run;
		}
		// workgroup-wide exclusive prefix maximum of the per-thread maxima
		const float incl = mpc_wave_scan_max_nonneg(run);
		float excl = mpc_lane_up1(incl);
		if (lane == 0) excl = 0.0f;
		if (lane == 63) wmax[wave] = incl;
		__syncthreads();
		for (u32 w = 0; w < wave; ++w) { const float o = wmax[w]; excl = o >= excl ? o : excl; }
		// pass 2: S and the traceback letters
		float Y = excl; // S(i, j0-1)
		for (u32 j = j0; j < j1; ++j) {
			if (j == 0) { newr[0] = 0.0f; tbrow[0] = 'X'; Y = 0.0f; continue; } // calcalnflat.cpp:23-25
			const float B = oldr[j - 1] + prow[j - 1];
			const float X = oldr[j];
			const bool bx = B >= X;            // best3.h:9
			const float T = bx ? B : X;
			const bool ty = T >= Y;            // best3.h:11 / :21
			const float S = ty ? T : Y;
			newr[j] = S;
			tbrow[j] = ty ? (bx ? 'B' : 'X') : 'Y';
			Y = S;
		}
		__syncthreads();
		float *tmp = oldr; oldr = newr; newr = tmp;
	}
	// TraceBackFlat (tracebackflat.cpp:3-37)
	if (tid == 0) {
		*p.score = oldr[LY];
		int i = (int)LX, j = (int)LY;
		u32 n = 0;
		while (i != 0 || j != 0) {
			const char c = p.tb[(u64)i * W + j];
			p.rev[n++] = c;
			if (c == 'B') { --i; --j; } else if (c == 'X') --i; else --j;
		}
		*s_n = n;
		*p.pathlen = n;
	}
	__syncthreads();
	const u32 n = *s_n;
	for (u32 k = tid; k < n; k += MPC_ALN_THREADS) p.path[k] = p.rev[n - 1 - k];
}

// ---- one wavefront per alignment (the common size: both alignments <= 512 columns, <= 600 rows) -----------------------
// The progressive joins and the 100 refinement rounds of a typical run align matrices of ~450 x 450: with one column per
// thread of a 1024-thread workgroup a row costs two workgroup barriers and a cross-wave scan through LDS, and the walk back
// chases ~900 dependent bytes through HBM (round 1: 0.72 ms per call, 0.79 s of a 7.3 s run). Here ONE wave keeps the
// previous DP row in registers, C = 8 consecutive columns per lane: the left neighbour of a lane's first column is a DPP
// shift, the prefix maximum of the row is the same wave scan (exact: max is associative), and there is no barrier at all.
// The traceback letters are 4-bit codes in LDS (two columns per byte, a lane's 8 columns = 4 whole bytes), so the one-lane
// walk back reads LDS, not HBM. Same cells, same comparisons, same tie order as calc_aln_kernel above.
#define MPC_ALNW_C 8                       // columns per lane
#define MPC_ALNW_MAXW (64 * MPC_ALNW_C)    // W = LY + 1 <= 512
#define MPC_ALNW_ROWBYTES (MPC_ALNW_MAXW / 2)
#ifndef MPC_ALNW_PF
#define MPC_ALNW_PF 4                      // rows of Post in flight
#endif

template <int C> // columns per lane: 64 * C >= LY + 1
__device__ __forceinline__ void calc_aln_wave_body_c(const AlnParams &p, unsigned char *smem_raw) // smem_raw: (LX+1) rows of MPC_ALNW_ROWBYTES traceback nibbles
{
	const u32 LX = p.LX, LY = p.LY, W = LY + 1;
	const u32 lane = threadIdx.x & 63u;
	const u32 j0 = lane * C; // my columns [j0, j0 + C)
	float oldr[C];
	u32 code0 = 0; // row 0: 'Y' everywhere (calcalnflat.cpp:15-19); codes: 0 = 'B', 1 = 'X', 2 = 'Y'
#pragma unroll
	for (int c = 0; c < C; ++c) { oldr[c] = 0.0f; code0 |= 2u << (4 * c); }
	((u32 *)smem_raw)[lane] = code0;
	// Post(i-1, j-1) of my columns, MPC_ALNW_PF rows ahead: the row loop is a dependent chain of ~0.2 us per row and a load from
	// HBM/L2 takes ~1-2 us, so one row of lookahead leaves the chain waiting on memory every row (measured: 1.5 us per row).
	// The loads are unconditional (a branch around a load makes the compiler wait for ALL loads in flight, vmcnt(0), at the
	// next use): rows past LX and columns outside 1..LY read a clamped in-range address instead, and what they read is never
	// used (column 0 and the columns past LY get their S and letter without it and stay out of the row maximum).
	float ring[MPC_ALNW_PF][C];
	u32 coff[C]; // clamped column offsets: the same for every row
#pragma unroll
	for (int c = 0; c < C; ++c) {
		const u32 j = j0 + c;
		coff[c] = (j < 1u ? 1u : (j > LY ? LY : j)) - 1u;
	}
	auto load_row = [&](float *dst, u32 i) {
		const float *prow = p.post + (u64)((i <= LX ? i : LX) - 1u) * LY;
#pragma unroll
		for (int c = 0; c < C; ++c) dst[c] = prow[coff[c]];
	};
#pragma unroll
	for (int r = 0; r < MPC_ALNW_PF; ++r) { load_row(ring[r], 1u + r); MPC_SCHED_BARRIER(); } // issued oldest row first: the waits count loads in order
	for (u32 ib = 1; ib <= LX; ib += MPC_ALNW_PF) {
#pragma unroll
	for (int r = 0; r < MPC_ALNW_PF; ++r) {
		const u32 i = ib + r;
		if (i > LX) break; // wave-uniform
		float pvc[C];
#pragma unroll
		for (int c = 0; c < C; ++c) pvc[c] = ring[r][c];
		load_row(ring[r], i + MPC_ALNW_PF);
		MPC_SCHED_BARRIER();
		// S(i-1, j0-1): the previous lane's last column of the previous row
		float left_old = mpc_lane_up1(oldr[C - 1]);
		if (lane == 0) left_old = 0.0f; // unused (column 0 has no B)
		float T[C];
		bool bx[C];
		float run = 0.0f; // T_j >= 0 always (X >= 0), and S(i,0) = 0
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const u32 j = j0 + c;
			const float diag = c == 0 ? left_old : oldr[c - 1];
			const float B = diag + pvc[c];
			const float X = oldr[c];
			bx[c] = B >= X; // best3.h:9
			T[c] = bx[c] ? B : X;
			if (j >= 1 && j <= LY) run = T[c] >= run ? T[c] : run;
		}
		// exclusive prefix maximum over the lanes = S(i, j0-1)
		const float incl = mpc_wave_scan_max_nonneg(run);
		float Y = mpc_lane_up1(incl);
		if (lane == 0) Y = 0.0f;
		u32 codes = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const u32 j = j0 + c;
			float S;
			u32 code;
			if (j == 0) { S = 0.0f; code = 1u; } // calcalnflat.cpp:23-25: column 0 = 'X'
			else {
				const bool ty = T[c] >= Y; // best3.h:11 / :21
				S = ty ? T[c] : Y;
				code = ty ? (bx[c] ? 0u : 1u) : 2u;
			}
			if (j > LY) { S = 0.0f; code = 2u; } // beyond the matrix: never read
			oldr[c] = S;
			codes |= code << (4 * c);
			Y = S;
		}
		*(u32 *)(smem_raw + (u64)i * MPC_ALNW_ROWBYTES + 4 * lane) = codes;
	}
	}
	// score = S(LX, LY): lane LY / C, register LY % C
	float sc = 0.0f;
#pragma unroll
	for (int c = 0; c < C; ++c) if ((u32)c == (LY % C)) sc = oldr[c];
	sc = mpc_read_lane(sc, LY / C);
	__syncthreads(); // one wave: orders the LDS writes above before the walk below
	// TraceBackFlat (tracebackflat.cpp:3-37) out of LDS
	u32 *s_n = (u32 *)(smem_raw + (u64)(LX + 1) * MPC_ALNW_ROWBYTES);
	if (lane == 0) {
		*p.score = sc;
		int i = (int)LX, j = (int)LY;
		u32 n = 0;
		while (i != 0 || j != 0) {
			const u32 jl = (u32)j / (u32)C, jc = (u32)j % (u32)C; // lane and register of column j: a lane's C codes are the low nibbles of its word of the row
			const u32 byte = smem_raw[(u64)i * MPC_ALNW_ROWBYTES + 4u * jl + (jc >> 1)];
			const u32 code = (byte >> (4 * (jc & 1))) & 0xfu;
			const char ch = code == 0u ? 'B' : (code == 1u ? 'X' : 'Y');
			p.rev[n++] = ch;
			if (code == 0u) { --i; --j; } else if (code == 1u) --i; else --j;
		}
		*s_n = n;
		*p.pathlen = n;
	}
	__syncthreads();
	const u32 n = *s_n;
	for (u32 k = lane; k < n; k += 64) p.path[k] = p.rev[n - 1 - k];
	(void)W;
}

// Columns per lane by the matrix's width: the row loop is a dependent chain whose length grows with the columns a lane owns (two
// passes over them around the wave scan), and the joins of a -super7 shrub are ~300 columns wide, those of a 400-residue family ~450:
// 5 / 7 columns per lane instead of 8 (51 148 one-wave alignments are 61 % of that run's kernel time: profiles/r10j). Same cells, same
// comparisons, same tie order; a row of codes stays one word per lane (256 bytes) whatever C is.
__device__ __forceinline__ void calc_aln_wave_body(const AlnParams &p, unsigned char *smem_raw)
{
	const u32 W = p.LY + 1; // wave-uniform
	if (W <= 64u * 4u) calc_aln_wave_body_c<4>(p, smem_raw);
	else if (W <= 64u * 5u) calc_aln_wave_body_c<5>(p, smem_raw);
	else if (W <= 64u * 6u) calc_aln_wave_body_c<6>(p, smem_raw);
	else if (W <= 64u * 7u) calc_aln_wave_body_c<7>(p, smem_raw);
	else calc_aln_wave_body_c<8>(p, smem_raw);
}

__global__ void __launch_bounds__(64) calc_aln_wave_kernel(AlnParams p)
{
	MPC_DYN_SMEM(smem_raw);
	calc_aln_wave_body(p, smem_raw);
}

// the same, one workgroup (= one wavefront) per alignment of a batch (mpcgpu_align_pairs): dynamic LDS sized for the batch's
// longest row sequence
__global__ void __launch_bounds__(64) calc_aln_wave_batch_kernel(const AlnParams *batch)
{
	MPC_DYN_SMEM(smem_raw);
	const AlnParams p = batch[blockIdx.x];
	calc_aln_wave_body(p, smem_raw);
}

// The dense thresholded posterior of CalcPostFlat (calcposteriorflat.cpp:9-26) for the pairs of one stage-A batch, from the
// candidate lists post_rows_kernel left behind (kernels_post.h: (cell key, P bits) of every cell with Score >= MIN_SPARSE_SCORE,
// INCLUDING those MySparseMx::FromPost later drops for P < 0.01 — AlignPairFlat's CalcAlnFlat runs on the dense matrix,
// alignpairflat.cpp:8-12). One workgroup per pair: zero, then scatter.
struct DensePostParams {
	const u32 *pair_x, *pair_y, *seq_len;
	const u64 *cand;
	u32 capc;
	const u32 *cand_cnt;
	u32 long_min;       // key layout: kernels_post.h mpc_key_shift
	const u64 *out_off; // floats: start of pair q's LX x LY matrix
	float *out;
};

__global__ void __launch_bounds__(256) dense_post_kernel(DensePostParams p)
{
	const u32 q = blockIdx.x;
	const u32 LX = p.seq_len[p.pair_x[q]], LY = p.seq_len[p.pair_y[q]];
	float *M = p.out + p.out_off[q];
	const u64 cells = (u64)LX * LY;
	for (u64 e = threadIdx.x; e < cells; e += blockDim.x) M[e] = 0.0f;
	__syncthreads();
	const u32 kshift = LX >= p.long_min ? MPC_KEY_ROW_SHIFT_LONG : MPC_KEY_ROW_SHIFT;
	const u32 c = p.cand_cnt[q];
	const u64 *cand = p.cand + (u64)q * p.capc;
	for (u32 e = threadIdx.x; e < c; e += blockDim.x) {
		const u64 v = cand[e];
		const u32 key = (u32)(v >> 32);
		const u32 row = key >> kshift, col = key & ((1u << kshift) - 1u);
		M[(u64)row * LY + col] = __uint_as_float((u32)v);
	}
}

// ---- several wavefronts, previous row in registers (matrices wider than one wave holds: the joins near the root) ---------
// Gap-rich families reach ~1000 x 1000 at the root and in every refinement round (793 of the 1099 joins of the 1000 x L~400
// run, ~1 ms each with calc_aln_kernel: the rows ping-pong through LDS with two barriers a row and the walk back chases
// ~2000 dependent bytes through L2). Here thread t keeps S(i-1, .) of its 4 columns [4t, 4t+4) in registers, NW = blockDim/64
// waves cover the row. The only values that cross waves are the per-wave maxima of T (one float per wave and row, LDS slots
// double-buffered by row parity: ONE barrier per row); S(i-1, 4t-1), the diagonal input of a thread's first column, is the
// exclusive prefix maximum the thread itself used one row earlier (S(i,j) is the prefix maximum of T), so the previous row
// never travels. Traceback letters are 2-bit codes, 4 columns = one byte per thread and row, written to HBM as the rows
// complete; the walk back stages them through LDS in blocks of `block_rows` rows (bulk copies by the whole workgroup), so
// every dependent step of the walk reads LDS. Same cells, same comparisons, same tie order as calc_aln_kernel.
#define MPC_ALNQ_C 4
#define MPC_ALNQ_PF 4          // rows of Post in flight
#define MPC_ALNQ_HDR 256       // bytes of LDS before the staged block: wave maxima [2][16], walk state

__global__ void __launch_bounds__(1024) calc_aln_quad_kernel(AlnParams p, u32 block_rows)
{
	MPC_DYN_SMEM(smem_raw);
	const u32 LX = p.LX, LY = p.LY;
	const u32 tid = threadIdx.x, lane = tid & 63u, wave = mpc_wave_first(tid >> 6);
	const u32 rowbytes = blockDim.x; // one byte per thread and row
	float *s_tot = (float *)smem_raw;                   // [2][16]
	u32 *s_state = (u32 *)(smem_raw + 128);             // i, j, n, done
	unsigned char *s_blk = smem_raw + MPC_ALNQ_HDR;     // block_rows * rowbytes
	unsigned char *tb = (unsigned char *)p.tb;
	const u32 j0 = tid * MPC_ALNQ_C;
	float oldr[MPC_ALNQ_C];
#pragma unroll
	for (int c = 0; c < MPC_ALNQ_C; ++c) oldr[c] = 0.0f;
	tb[tid] = 0xaau; // row 0: 'Y' everywhere (calcalnflat.cpp:15-19); codes: 0 = 'B', 1 = 'X', 2 = 'Y'
	float left_old = 0.0f; // S(i-1, j0-1)
	float ring[MPC_ALNQ_PF][MPC_ALNQ_C];
	u32 coff[MPC_ALNQ_C]; // clamped column offsets, unconditional loads: see calc_aln_wave_kernel
#pragma unroll
	for (int c = 0; c < MPC_ALNQ_C; ++c) {
		const u32 j = j0 + c;
		coff[c] = (j < 1u ? 1u : (j > LY ? LY : j)) - 1u;
	}
	auto load_row = [&](float *dst, u32 i) {
		const float *prow = p.post + (u64)((i <= LX ? i : LX) - 1u) * LY;
#pragma unroll
		for (int c = 0; c < MPC_ALNQ_C; ++c) dst[c] = prow[coff[c]];
	};
#pragma unroll
	for (int r = 0; r < MPC_ALNQ_PF; ++r) { load_row(ring[r], 1u + r); MPC_SCHED_BARRIER(); } // issued oldest row first: the waits count loads in order
	for (u32 ib = 1; ib <= LX; ib += MPC_ALNQ_PF) {
#pragma unroll
	for (int r = 0; r < MPC_ALNQ_PF; ++r) {
		const u32 i = ib + r;
		if (i > LX) break; // uniform
		float pvc[MPC_ALNQ_C];
#pragma unroll
		for (int c = 0; c < MPC_ALNQ_C; ++c) pvc[c] = ring[r][c];
		load_row(ring[r], i + MPC_ALNQ_PF);
		MPC_SCHED_BARRIER();
		float T[MPC_ALNQ_C];
		bool bx[MPC_ALNQ_C];
		float run = 0.0f; // T_j >= 0 always (X >= 0), and S(i,0) = 0
#pragma unroll
		for (int c = 0; c < MPC_ALNQ_C; ++c) {
			const u32 j = j0 + c;
			const float diag = c == 0 ? left_old : oldr[c - 1];
			const float B = diag + pvc[c];
			const float X = oldr[c];
			bx[c] = B >= X; // best3.h:9
			T[c] = bx[c] ? B : X;
			if (j >= 1 && j <= LY) run = T[c] >= run ? T[c] : run;
		}
		const float incl = mpc_wave_scan_max_nonneg(run);
		float *tot = s_tot + 16u * (i & 1u);
		if (lane == 63) tot[wave] = incl;
		__syncthreads();
		float Y = mpc_lane_up1(incl);
		if (lane == 0) Y = 0.0f;
		for (u32 w = 0; w < wave; ++w) { const float o = tot[w]; Y = o >= Y ? o : Y; }
		left_old = Y; // S(i, j0-1): next row's diagonal input
		u32 codes = 0;
#pragma unroll
		for (int c = 0; c < MPC_ALNQ_C; ++c) {
			const u32 j = j0 + c;
			float S;
			u32 code;
			if (j == 0) { S = 0.0f; code = 1u; } // calcalnflat.cpp:23-25: column 0 = 'X'
			else {
				const bool ty = T[c] >= Y; // best3.h:11 / :21
				S = ty ? T[c] : Y;
				code = ty ? (bx[c] ? 0u : 1u) : 2u;
			}
			if (j > LY) { S = 0.0f; code = 2u; } // beyond the matrix: never read
			oldr[c] = S;
			codes |= code << (2 * c);
			Y = S;
		}
		tb[(u64)i * rowbytes + tid] = (unsigned char)codes;
	}
	}
	// score = S(LX, LY)
	if (tid == LY / MPC_ALNQ_C) {
		float sc = 0.0f;
#pragma unroll
		for (int c = 0; c < MPC_ALNQ_C; ++c) if ((u32)c == (LY % MPC_ALNQ_C)) sc = oldr[c];
		*p.score = sc;
	}
	if (tid == 0) { s_state[0] = LX; s_state[1] = LY; s_state[2] = 0; s_state[3] = 0; }
	__syncthreads(); // also orders the traceback bytes above before the copies below
	// TraceBackFlat (tracebackflat.cpp:3-37), block after block of rows out of LDS
	for (;;) {
		const u32 hi = s_state[0];
		const u32 lo = hi + 1 >= block_rows ? hi + 1 - block_rows : 0u;
		const u32 nwords = (hi - lo + 1) * (rowbytes / 4);
		const u32 *src = (const u32 *)(tb + (u64)lo * rowbytes);
		for (u32 k = tid; k < nwords; k += blockDim.x) ((u32 *)s_blk)[k] = src[k];
		__syncthreads();
		if (tid == 0) {
			int i = (int)hi, j = (int)s_state[1];
			u32 n = s_state[2];
			while ((i != 0 || j != 0) && i >= (int)lo) {
				const u32 byte = s_blk[(u64)(i - (int)lo) * rowbytes + (j >> 2)];
				const u32 code = (byte >> (2 * (j & 3))) & 3u;
				p.rev[n++] = code == 0u ? 'B' : (code == 1u ? 'X' : 'Y');
				if (code == 0u) { --i; --j; } else if (code == 1u) --i; else --j;
			}
			s_state[0] = (u32)(i < 0 ? 0 : i); s_state[1] = (u32)j; s_state[2] = n;
			s_state[3] = (i == 0 && j == 0) ? 1u : 0u;
		}
		__syncthreads();
		if (s_state[3]) break;
	}
	const u32 n = s_state[2];
	if (tid == 0) *p.pathlen = n;
	for (u32 k = tid; k < n; k += blockDim.x) p.path[k] = p.rev[n - 1 - k];
}
