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
#include "kernels_fb.h" // MPC_KEY_ROW_SHIFT: layout of the candidate keys

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
	u32 long_min;   // pairs with LX >= long_min came from a row-block (LONG) fb kernel: 16-bit column keys
};

// candidate key layout of a pair (kernels_fb.h): row << shift | column
__device__ __forceinline__ u32 mpc_key_shift(u32 LX, u32 long_min) { return LX >= long_min ? MPC_KEY_ROW_SHIFT_LONG : MPC_KEY_ROW_SHIFT; }

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
		const u32 kshift = mpc_key_shift(LX, p.long_min);
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
			const u32 rlo = i << kshift; // key range of posterior row i: [rlo, rhi) (64-bit: the last row ends at 2^32)
			const u64 rhi = (u64)(i + 1) << kshift;
			u32 e1 = e;
			while (e1 < c && (buf[e1] >> 32) < rhi) ++e1;
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
				const u32 row = idx >> kshift, col = idx & ((1u << kshift) - 1u);
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

// ---------------------------------------------------------------------------------------------
// post_rows_kernel — same outputs as post_kernel, without the two bitonic sorts and with an EA DP
// that touches LDS three times per row instead of a dozen. Used when LY fits the LDS arrays;
// post_kernel stays the general path.
//
//  * Row-major order by a counting sort on the row (LDS histogram -> scan -> scatter through a per-row
//    cursor) and a tiny insertion sort inside each row (a lane per row, ~3 cells per row); the
//    column-major rank (tperm) the same way on the column.
//  * EA score (calcalnscoreflat.cpp:4-32): PM[j] holds S(i,j) of the current row (non-decreasing in
//    j). Unrolling the recurrence over j gives S(i,j) = max(S(i-1,j), max over stored cells (i,c),
//    c <= j, of S(i-1,c-1) + P(i,c)); cells with P == 0 change nothing because S(i-1,.) is
//    non-decreasing. So a row needs: one LDS read of PM[c-1] per stored cell (all cells of the row at
//    once, from the previous row's values), one add each, a running maximum over the cells in column
//    order, and one pass "PM[j] = max(PM[j], V(j))" over the columns right of the first cell — as far as
//    the right-most column any row has reached so far: beyond it S(i,.) is flat and is kept as that one value.
//    The only rounding is the one add per stored cell, max is exact: bit-identical to the sequential DP.
struct PostRowsParams {
	const u32 *pair_x, *pair_y;
	const u32 *seq_len;
	u64 *cand;        // in: (row << MPC_KEY_ROW_SHIFT | col) << 32 | score bits; probability bits replace the score in place
	u32 capc;
	const u32 *cand_cnt;
	int use_fma;
	u32 lx_cap, ly_cap; // LDS array sizes (entries): rows+1, cols+1
	u32 sort_cap;       // LDS entries for the row-sorted list; larger pairs use sort_scratch
	u32 batch;          // cells of one row handled per EA pass: 64 (a smaller value only to test the multi-pass path)
	u64 *sort_scratch;
	u64 sort_stride;
	u32 *res;
	u64 res_stride;
	u32 *nnz;
	float *ea;
	u32 *flags;
	u32 count;
	u32 long_min; // as in PostParams
};

// phase clock of post_rows_kernel (workgroup 0, lane 0 only)

__global__ void __launch_bounds__(64) post_rows_kernel(PostRowsParams p)
{
	MPC_DYN_SMEM(smem_raw);
	u32 *s_rend = (u32 *)smem_raw;            // lx_cap: per-row counts, then row ENDS (cursor after the scatter)
	u32 *s_cend = s_rend + p.lx_cap;          // ly_cap: same per column
	float *s_pm = (float *)(s_cend + p.ly_cap); // ly_cap: PM[0..LY]
	u64 *s_sorted = (u64 *)(smem_raw + ((((size_t)p.lx_cap + 2 * (size_t)p.ly_cap) * 4 + 7) & ~(size_t)7)); // sort_cap: (col << 32 | P bits), row-major
	const int t = threadIdx.x;

	for (u32 pid = blockIdx.x; pid < p.count; pid += gridDim.x) {
		const u32 LX = p.seq_len[p.pair_x[pid]], LY = p.seq_len[p.pair_y[pid]];
		const u32 kshift = mpc_key_shift(LX, p.long_min);
		u32 *rec = p.res + (u64)pid * p.res_stride;
		const u32 c = p.cand_cnt[pid];
		if (c > p.capc) { // overflow: reported to the host, which retries with a larger capacity
			if (t == 0) { p.flags[pid] = 1u; p.nnz[pid] = 0; p.ea[pid] = 0.0f; }
			continue;
		}
		if (t == 0) p.flags[pid] = 0u;
		u64 *cand = p.cand + (u64)pid * p.capc;
		u64 *sorted = (c <= p.sort_cap) ? s_sorted : (p.sort_scratch + (u64)blockIdx.x * p.sort_stride);
		for (u32 q = t; q <= LX; q += 64) s_rend[q] = 0;
		for (u32 q = t; q <= LY; q += 64) { s_cend[q] = 0; s_pm[q] = 0.0f; }
		for (u32 q = t; q < LX + LY; q += 64) rec[q] = 0;
		__syncthreads();
		// ---- probabilities (calcposteriorflat.cpp:16-22) and the row histogram
		for (u32 q = t; q < c; q += 64) {
			const u64 v = cand[q];
			const float pr = mpc_score_to_prob(__uint_as_float((u32)v), p.use_fma);
			cand[q] = (v & 0xffffffff00000000ull) | (u64)__float_as_uint(pr);
			atomicAdd(&s_rend[(u32)(v >> (32 + kshift))], 1u);
		}
		__syncthreads();
		// exclusive scan of the row counts (in place: s_rend[i] = first slot of row i)
		{
			u32 carry = 0;
			for (u32 a0 = 0; a0 <= LX; a0 += 64) {
				const u32 a = a0 + t;
				const u32 v = (a <= LX) ? s_rend[a] : 0;
				u32 incl = v;
				for (int d = 1; d < 64; d <<= 1) {
					const u32 o = __shfl_up(incl, d);
					if (t >= d) incl += o;
				}
				if (a <= LX) s_rend[a] = carry + incl - v;
				carry += __shfl(incl, 63);
			}
		}
		__syncthreads();
		// scatter through the per-row cursor: afterwards s_rend[i] = END of row i (start of row i+1)
		for (u32 q = t; q < c; q += 64) {
			const u64 v = cand[q];
			const u32 at = atomicAdd(&s_rend[(u32)(v >> (32 + kshift))], 1u);
			sorted[at] = ((v >> 32) & (u64)((1u << kshift) - 1u)) << 32 | (v & 0xffffffffull);
		}
		__syncthreads();
		// columns ascending inside each row (a lane per row; rows hold a handful of cells)
		for (u32 i = t; i < LX; i += 64) {
			const u32 b = i ? s_rend[i - 1] : 0u, e = s_rend[i];
			for (u32 x = b + 1; x < e; ++x) {
				const u64 key = sorted[x];
				u32 y = x;
				while (y > b && sorted[y - 1] > key) { sorted[y] = sorted[y - 1]; --y; }
				sorted[y] = key;
			}
		}
		__syncthreads();
		// ---- EA score. PM is kept explicitly only up to column cmax, the right-most column any row has updated so far; right of
		// it S(i,.) is flat (nothing stored there has been reached yet), PM[j] == PM[cmax]. Stored cells hug the alignment
		// path, so a row updates the few columns between its first cell and that frontier instead of all LY of them.
		// The rows are a dependent chain (row i reads what row i-1 wrote), so what a row costs is the latency of what it has to
		// wait for. Off the chain: the row ends (64 rows per LDS read, then v_readlane) and the row's cells (the next non-empty
		// row's cells start where this row's end, so they are loaded one row ahead). On it: PM[col-1] of the cells -> add ->
		// wave scan -> the update of the few columns up to the frontier. One wavefront is the whole workgroup and LDS executes a
		// wave's accesses in order, so a row's writes need no barrier before the next row's reads.
		u32 cmax = 0; // wave-uniform
		u64 nkey = c ? sorted[(u32)t < c ? (u32)t : c - 1u] : 0ull; // cells [b, b+64) of the next non-empty row
		for (u32 i0 = 0; i0 < LX; i0 += 64) {
			const u32 ri = i0 + (u32)t;
			const u32 rend_v = s_rend[ri < LX ? ri : LX - 1u]; // lane l: end of row i0+l
			u32 b = mpc_wave_first(i0 ? s_rend[i0 - 1u] : 0u);
			const u32 nl = LX - i0 < 64u ? LX - i0 : 64u;
			for (u32 rl = 0; rl < nl; ++rl) {
				const u32 e = mpc_read_lane(rend_v, rl);
				if (e == b) continue;
				if (e - b > p.batch) {
					// more than one batch of cells (rare): every B must still come from the previous row's PM, so they are
					// formed first (s_cend is free until the sparsify step), then the batches update PM one after the other
					for (u32 x = b + (u32)t; x < e; x += 64) {
						const u64 key = sorted[x];
						const u32 cc = (u32)(key >> 32);
						s_cend[x - b] = __float_as_uint(s_pm[cc < cmax ? cc : cmax] + __uint_as_float((u32)key));
					}
					__syncthreads();
					float vprev = 0.0f; // maximum over the cells of this row handled so far
					for (u32 c0 = b; c0 < e; c0 += p.batch) {
						const u32 me = c0 + (u32)t;
						const bool have = me < e && (u32)t < p.batch;
						const u64 key = have ? sorted[me] : 0ull;
						const u32 col = (u32)(key >> 32);
						const float val = have ? __uint_as_float(s_cend[me - b]) : 0.0f;
						const float suffix = s_pm[cmax];
						const float V = fmaxf(mpc_wave_scan_max_nonneg(val), vprev);
						const u32 k = (e - c0 < p.batch) ? (e - c0) : p.batch;
						const u32 last = mpc_read_lane(col, k - 1) + 1u;
						const u32 hi = last > cmax ? last : cmax;
						// from the first column behind the frontier at the latest: a batch that starts right of cmax + 1 leaves a
						// gap (cmax, first cell] that becomes explicit with this row and has to hold the flat suffix S(i-1, cmax)
						const u32 jb0 = mpc_wave_first(col) + 1u;
						for (u32 j0 = jb0 < cmax + 1u ? jb0 : cmax + 1u; j0 <= hi; j0 += 64) {
							const u32 j = j0 + (u32)t;
							float cur = 0.0f;
							for (u32 l = 0; l < k; ++l) {
								const u32 cj = mpc_read_lane(col, l) + 1u;
								const float vj = mpc_read_lane(V, l);
								cur = (j >= cj) ? vj : cur;
							}
							if (j <= hi) s_pm[j] = fmaxf(j <= cmax ? s_pm[j] : suffix, cur);
						}
						cmax = hi;
						vprev = mpc_read_lane(V, k - 1);
						__syncthreads();
					}
					for (u32 x = b + (u32)t; x < e; x += 64) s_cend[x - b] = 0;
					__syncthreads();
					nkey = sorted[e + (u32)t < c ? e + (u32)t : c - 1u];
					b = e;
					continue;
				}
				const u32 k = e - b; // 1..64 cells, lane l holds cell l (columns ascending)
				const bool have = (u32)t < k;
				const u64 key = nkey;
				nkey = sorted[e + (u32)t < c ? e + (u32)t : c - 1u]; // in flight while this row is computed
				const u32 col = have ? (u32)(key >> 32) : 0u;
				const u32 jfirst = mpc_wave_first(col) + 1u;
				// B = S(i-1,j-1) + P with j = col+1 (calcalnscoreflat.cpp:20)
				const float val = have ? s_pm[col < cmax ? col : cmax] + __uint_as_float((u32)key) : 0.0f;
				const float suffix = s_pm[cmax]; // S(i-1, j) for every j >= cmax
				const float V = mpc_wave_scan_max_nonneg(val); // running maximum in column order
				const u32 last = mpc_read_lane(col, k - 1) + 1u; // right-most column the row's cells start at
				const u32 hi = last > cmax ? last : cmax;        // explicit range after this row
				// The update starts at the row's first cell or, when that lies beyond the frontier, right behind the frontier:
				// the columns in between become explicit with this row (cmax = hi below) and must receive S(i-1, cmax) — left
				// unwritten they would keep their initial 0.0f and later rows would read it as S(., j).
				for (u32 j0 = jfirst < cmax + 1u ? jfirst : cmax + 1u; j0 <= hi; j0 += 64) {
					const u32 j = j0 + (u32)t;
					const float old = s_pm[j <= cmax ? j : cmax];
					float cur = 0.0f;
					for (u32 l = 0; l < k; ++l) { // cells ascend in column: the last one with col+1 <= j wins
						const u32 cj = mpc_read_lane(col, l) + 1u;
						const float vj = mpc_read_lane(V, l);
						cur = (j >= cj) ? vj : cur;
					}
					if (j <= hi) s_pm[j] = fmaxf(j <= cmax ? old : suffix, cur); // max(X, Y-chain) of the recurrence
				}
				cmax = hi;
				b = e;
				MPC_WAVE_LDS_ORDER();
			}
		}
		const float score = s_pm[LY < cmax ? LY : cmax];
		const u32 mn = LX < LY ? LX : LY;
		const float ea = score / (float)mn; // calcposteriorflat.cpp:89 (uint -> float, IEEE divide)
		__syncthreads();
		// ---- sparsify (mysparsemx.cpp:115-152): keep P >= 0.01f, row-major rank among the kept
		u32 kept = 0;
		for (u32 q0 = 0; q0 < c; q0 += 64) {
			const u32 q = q0 + t;
			const bool k = q < c && __uint_as_float((u32)sorted[q]) >= MPC_MIN_SPARSE_PROB;
			kept += (u32)__popcll(__ballot(k));
		}
		const u32 nnz = kept;
		u32 *rowcnt = rec, *colcnt = rec + LX;
		u32 *ent = rec + LX + LY;
		u32 *rowv = ent + 2 * (u64)nnz;
		u32 *tperm = rowv + nnz;
		// row of every sorted slot: slots of row i are [s_rend[i-1], s_rend[i])
		u32 base = 0;
		for (u32 i0 = 0; i0 < LX; i0 += 64) { // a lane per row writes that row's kept cells
			const u32 i = i0 + (u32)t;
			u32 mycnt = 0;
			u32 b = 0, e = 0;
			if (i < LX) {
				b = i ? s_rend[i - 1] : 0u; e = s_rend[i];
				for (u32 x = b; x < e; ++x) mycnt += (__uint_as_float((u32)sorted[x]) >= MPC_MIN_SPARSE_PROB) ? 1u : 0u;
			}
			u32 incl = mycnt;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (t >= d) incl += o;
			}
			u32 at = base + incl - mycnt;
			if (i < LX) {
				rowcnt[i] = mycnt;
				for (u32 x = b; x < e; ++x) {
					const u64 key = sorted[x];
					if (__uint_as_float((u32)key) >= MPC_MIN_SPARSE_PROB) {
						const u32 col = (u32)(key >> 32);
						ent[2 * (u64)at] = (u32)key;
						ent[2 * (u64)at + 1] = col;
						rowv[at] = i;
						atomicAdd(&s_cend[col], 1u);
						++at;
					}
				}
			}
			base += __shfl(incl, 63);
		}
		__syncthreads();
		// ---- column-major rank of every kept entry: counting sort on the column, rows ascending inside
		for (u32 q = t; q < LY; q += 64) colcnt[q] = s_cend[q];
		{
			u32 carry = 0;
			for (u32 a0 = 0; a0 <= LY; a0 += 64) {
				const u32 a = a0 + t;
				const u32 v = (a <= LY) ? s_cend[a] : 0;
				u32 incl = v;
				for (int d = 1; d < 64; d <<= 1) {
					const u32 o = __shfl_up(incl, d);
					if (t >= d) incl += o;
				}
				if (a <= LY) s_cend[a] = carry + incl - v;
				carry += __shfl(incl, 63);
			}
		}
		__syncthreads();
		u32 *csorted = (u32 *)sorted; // entry ranks in column-major order (the row-sorted list is no longer needed)
		__syncthreads();
		for (u32 q = t; q < nnz; q += 64) {
			const u32 at = atomicAdd(&s_cend[ent[2 * (u64)q + 1]], 1u);
			csorted[at] = q; // entry rank (row-major); rows ascend with the rank
		}
		__syncthreads();
		for (u32 j = t; j < LY; j += 64) { // ranks ascending inside each column = rows ascending
			const u32 b = j ? s_cend[j - 1] : 0u, e = s_cend[j];
			for (u32 x = b + 1; x < e; ++x) {
				const u32 key = csorted[x];
				u32 y = x;
				while (y > b && csorted[y - 1] > key) { csorted[y] = csorted[y - 1]; --y; }
				csorted[y] = key;
			}
		}
		__syncthreads();
		for (u32 q = t; q < nnz; q += 64) tperm[csorted[q]] = q;
		if (t == 0) { p.nnz[pid] = nnz; p.ea[pid] = ea; }
		__syncthreads();
	}
}
