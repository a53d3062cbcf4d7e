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
