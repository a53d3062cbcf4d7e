// kernels_post.h — finish one pair's posterior: probabilities, sparse matrix, EA score.
//
// Replaces, per pair: the expf half of CalcPostFlat (calcposteriorflat.cpp:16-22),
// MySparseMx::FromPost (mysparsemx.cpp:115-152), CalcAlnScoreFlat (calcalnscoreflat.cpp:4-32) and
// EA = Score/min(LX,LY) (calcposteriorflat.cpp:89). One 64-thread workgroup (one wavefront) per
// pair; the candidate list of the fb kernel (unordered, ~1 % of the cells) is sorted in LDS.
//
// Output record of a pair ("packed record", u32 words, the unit the multi-GPU exchange ships):
//   [rowcnt: LX] [colcnt: LY] [ent: 2*nnz = {P bits, col}] [row: nnz] [tperm: nnz]
// ent is row-major (rows ascending, cols ascending) = MySparseMx::m_ValueVec byte for byte;
// row[k] is the row of entry k; tperm[k] is the rank of entry k in column-major order (used to
// build the transposed copy the relax kernel reads).
//
// EA DP on the device: S(i,j) = max(S(i-1,j-1)+P(i,j), S(i-1,j), S(i,j-1)) over the dense
// thresholded posterior (entries with Score >= MIN_SPARSE_SCORE, including those FromPost later
// drops because P < 0.01f). Row-wise it is T(j) = max(S(i-1,j), S(i-1,j-1)+P(i,j)) followed by a
// prefix maximum over j; max is exact and associative, and for P == 0 cells T(j) = S(i-1,j)
// because S(i-1,.) is non-decreasing, so a wave-parallel prefix-max scan reproduces the
// sequential recurrence bit for bit (the only rounding is the one float add per stored entry).
#pragma once
#include "device_math.h"

struct PostParams {
	const u32 *pair_x, *pair_y; // per batch-local pair
	const u32 *seq_len;
	const u64 *cand;
	u32 capc;
	const u32 *cand_cnt;
	int use_fma;
	u32 sort_cap;      // entries that fit the LDS sort buffer (power of two)
	u32 srow_cap;      // floats that fit the LDS DP row
	u64 *sort_scratch; // global fallback: sort_stride u64 per block
	u64 sort_stride;
	float *srow_scratch; // global fallback DP row: srow_stride floats per block
	u64 srow_stride;
	u32 *res;       // per batch-local pair: res_stride words
	u64 res_stride; // words
	u32 *nnz;       // per batch-local pair
	float *ea;      // per batch-local pair
	u32 *flags;     // per batch-local pair: bit0 = candidate overflow
	u32 count;      // pairs in this batch
};

__device__ __forceinline__ u32 mpc_next_pow2(u32 v)
{
	u32 p = 1;
	while (p < v) p <<= 1;
	return p;
}

// ascending bitonic sort of n2 (power of two) u64 keys by one 64-thread workgroup
__device__ __forceinline__ void mpc_bitonic_sort(u64 *buf, u32 n2)
{
	for (u32 k = 2; k <= n2; k <<= 1) {
		for (u32 j = k >> 1; j > 0; j >>= 1) {
			for (u32 q = threadIdx.x; q < n2; q += blockDim.x) {
				const u32 ixj = q ^ j;
				if (ixj > q) {
					const u64 a = buf[q], b = buf[ixj];
					const bool asc = (q & k) == 0;
					if ((a > b) == asc) { buf[q] = b; buf[ixj] = a; }
				}
			}
			__syncthreads();
		}
	}
}

__global__ void __launch_bounds__(64) post_kernel(PostParams p)
{
	MPC_DYN_SMEM(smem_raw);
	u64 *s_sort = (u64 *)smem_raw;
	float *s_row = (float *)(smem_raw + (size_t)p.sort_cap * 8); // 2 rows of srow_cap floats
	const int t = threadIdx.x;

	for (u32 pid = blockIdx.x; pid < p.count; pid += gridDim.x) {
		const u32 LX = p.seq_len[p.pair_x[pid]], LY = p.seq_len[p.pair_y[pid]];
		u32 *rec = p.res + (u64)pid * p.res_stride;
		u32 c = p.cand_cnt[pid];
		if (c > p.capc) { // overflow: reported to the host, which fails loudly
			if (t == 0) { p.flags[pid] = 1u; p.nnz[pid] = 0; p.ea[pid] = 0.0f; }
			continue;
		}
		if (t == 0) p.flags[pid] = 0u;
		const u32 n2 = mpc_next_pow2(c < 2 ? 2 : c);
		u64 *buf = (n2 <= p.sort_cap) ? s_sort : (p.sort_scratch + (u64)blockIdx.x * p.sort_stride);
		float *S = (LY + 1 <= p.srow_cap) ? s_row : (p.srow_scratch + (u64)blockIdx.x * p.srow_stride); // 2*(LY+1) floats
		const u64 *cand = p.cand + (u64)pid * p.capc;
		// probabilities (calcposteriorflat.cpp:16-22); key = (flat index << 32) | P bits
		for (u32 q = t; q < n2; q += 64) {
			u64 key = ~0ull;
			if (q < c) {
				const u64 v = cand[q];
				const float pr = mpc_score_to_prob(__uint_as_float((u32)v), p.use_fma);
				key = (v & 0xffffffff00000000ull) | (u64)__float_as_uint(pr);
			}
			buf[q] = key;
		}
		for (u32 q = t; q < LX + LY; q += 64)
			rec[q] = 0;
		__syncthreads();
		mpc_bitonic_sort(buf, n2);

		// ---- EA score (calcalnscoreflat.cpp:4-32): lane t owns DP columns [t*C, t*C+C).
		// Two DP rows ping-pong (Sp = S(i-1,.), Sn = S(i,.)) so the diagonal reads never race the
		// writes; rows of the posterior without stored cells leave S unchanged and are skipped.
		const u32 C = (LY + 1 + 63) / 64;
		float *Sp = S, *Sn = S + (LY + 1);
		for (u32 q = t; q <= LY; q += 64)
			Sp[q] = 0.0f;
		__syncthreads();
		const u32 c0 = t * C;
		u32 e = 0; // wave-uniform cursor into the sorted candidates
		for (u32 i = 0; i < LX; ++i) {
			const u32 rlo = i * LY, rhi = rlo + LY; // flat-index range of posterior row i
			u32 e1 = e;
			while (e1 < c && (u32)(buf[e1] >> 32) < rhi) ++e1;
			if (e1 > e) {
				u32 kk = e; // first stored cell of this row at or right of my first column
				while (kk < e1 && (u32)(buf[kk] >> 32) - rlo + 1 < c0) ++kk;
				float run = 0.0f;
				for (u32 q = 0; q < C; ++q) {
					const u32 j = c0 + q;
					if (j > LY) break;
					float v = Sp[j]; // X = S(i-1, j)
					if (kk < e1 && (u32)(buf[kk] >> 32) - rlo + 1 == j) {
						const float b = Sp[j - 1] + __uint_as_float((u32)buf[kk]); // B = S(i-1,j-1) + P
						v = fmaxf(v, b);
						++kk;
					}
					run = (q == 0) ? v : fmaxf(run, v);
					Sn[j] = run; // prefix max inside my columns
				}
				// lanes without columns contribute 0 <= every S
				const float incl = mpc_wave_scan_max_nonneg(run);
				float excl = mpc_lane_up1(incl);
				if (t == 0) excl = 0.0f;
				for (u32 q = 0; q < C; ++q) {
					const u32 j = c0 + q;
					if (j > LY) break;
					Sn[j] = fmaxf(Sn[j], excl); // Y = S(i, j-1) folded in
				}
				__syncthreads();
				float *tmp = Sp; Sp = Sn; Sn = tmp;
			}
			e = e1;
		}
		S = Sp;
		const float score = S[LY];
		const u32 mn = LX < LY ? LX : LY;
		const float ea = score / (float)mn; // calcposteriorflat.cpp:89 (uint -> float, IEEE divide)
		__syncthreads();

		// ---- sparsify (mysparsemx.cpp:115-152): keep P >= 0.01f; rank = position among kept
		u32 *rowcnt = rec, *colcnt = rec + LX;
		u32 base = 0;
		// pass 1: count kept
		u32 kept = 0;
		for (u32 q0 = 0; q0 < c; q0 += 64) {
			const u32 q = q0 + t;
			const bool k = q < c && __uint_as_float((u32)buf[q]) >= MPC_MIN_SPARSE_PROB;
			kept += (u32)__popcll(__ballot(k));
		}
		const u32 nnz = kept;
		u32 *ent = rec + LX + LY;
		u32 *rowv = ent + 2 * (u64)nnz;
		u32 *tperm = rowv + nnz;
		for (u32 q0 = 0; q0 < c; q0 += 64) {
			const u32 q = q0 + t;
			const u64 key = q < c ? buf[q] : 0ull;
			const bool k = q < c && __uint_as_float((u32)key) >= MPC_MIN_SPARSE_PROB;
			const u64 bal = __ballot(k);
			if (k) {
				const u32 rank = base + (u32)__popcll(bal & ((1ull << t) - 1ull));
				const u32 idx = (u32)(key >> 32);
				const u32 row = idx / LY, col = idx - row * LY;
				ent[2 * (u64)rank] = (u32)key;
				ent[2 * (u64)rank + 1] = col;
				rowv[rank] = row;
				atomicAdd(&rowcnt[row], 1u);
				atomicAdd(&colcnt[col], 1u);
			}
			base += (u32)__popcll(bal);
		}
		__syncthreads();
		// ---- column-major rank of every kept entry: sort (col*LX + row, rank)
		const u32 m2 = mpc_next_pow2(nnz < 2 ? 2 : nnz);
		for (u32 q = t; q < m2; q += 64) {
			u64 key = ~0ull;
			if (q < nnz)
				key = ((u64)(ent[2 * (u64)q + 1] * LX + rowv[q]) << 32) | (u64)q;
			buf[q] = key;
		}
		__syncthreads();
		mpc_bitonic_sort(buf, m2);
		for (u32 q = t; q < nnz; q += 64)
			tperm[(u32)buf[q]] = q;
		if (t == 0) { p.nnz[pid] = nnz; p.ea[pid] = ea; }
		__syncthreads();
	}
}
