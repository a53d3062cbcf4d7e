// kernels_prog.h — alignment of two alignments on the device: MSA x MSA posterior build.
//
// Replaces MPCFlat::BuildPost (buildpostflat.cpp:18-106): for all s in MSA1 (outer loop), t in
// MSA2 (inner loop) and every stored entry (i,j,P) of their pairwise matrix,
//     Post[col1(i) * C2 + col2(j)] += w1*w2*P        (w == 1.0f: mpcflat.cpp:324)
// with the roles of row/column swapped when the stored orientation is (t,s)
// (buildpostflat.cpp:78-100). CalcAlnFlat then runs on that dense matrix (kernels_aln.h).
//
// Float addition is not associative and the alignment DP breaks ties on exact values, so every
// output cell must receive its contributions in the reference's order: s ascending (position in
// MSA1), then t ascending (position in MSA2); one pair contributes at most once to a cell. The
// device form: (1) every stored entry of every (s,t) pair becomes a (cell, P) record, written pair after pair in the
// reference's loop order (s-major), so the records of any one cell already stand in the order they must be added in; (2) one
// STABLE radix sort of the records by cell alone (rocprim::radix_sort_pairs on 32-bit keys, bits [0, bits(cells)) — bulk
// data movement, not arithmetic; stability is what carries the (s,t) order into each run); (3) the runs of equal cells are
// listed (build_post_heads_kernel) and one WAVE per run adds it up front to back through its lanes
// (build_post_reduce_kernel), starting from 0.0f like the reference's zeroed matrix. The probabilities are the current ones of the packed records (after the last commit).
#pragma once
#include "kernels_store.h"

struct BuildPostParams {
	StoreParams s;
	const u32 *seq1, *seq2; // sequence index (InitPairs numbering) of every row of MSA1 / MSA2
	u32 n1, n2;
	const u32 *p2c1, *p2c2;         // position -> column maps, concatenated per row
	const u64 *p2c1_off, *p2c2_off; // start of each row's map
	u32 C2;
	const u64 *coff; // n1*n2+1: first record of pair (a,b)
	u32 *keys; // cell of every record
	float *vals;
	const float *w1, *w2; // sequence weights of the rows of MSA1 / MSA2 (buildpostflat.cpp:41,52), nullptr = 1.0f
	float *post;          // C1*C2 output matrix, zeroed here (buildpostflat.cpp:27-30) — this launch precedes the sort and the reduction
	u64 cells;
	u32 *counters;        // {runs, next run} of the reduction, zeroed here
};

// zeroes what the reduction adds into / counts with: a slice per workgroup of the generating launch (saves two memset launches
// per join; the joins inside a shrub of 32 sequences are launch-bound)
__device__ __forceinline__ void build_post_zero(float *post, u64 cells, u32 *counters)
{
	for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < cells; q += (u64)gridDim.x * blockDim.x) post[q] = 0.0f;
	if (blockIdx.x == 0 && threadIdx.x < 2u) counters[threadIdx.x] = 0u;
}

// one 64-thread workgroup per (a,b)
__global__ void __launch_bounds__(64) build_post_gen_kernel(BuildPostParams p)
{
	const u64 total = (u64)p.n1 * p.n2;
	build_post_zero(p.post, p.cells, p.counters);
	for (u64 ab = blockIdx.x; ab < total; ab += gridDim.x) {
		const u32 a = (u32)(ab / p.n2), b = (u32)(ab % p.n2);
		const u32 S = p.seq1[a], T = p.seq2[b];
		const bool fwd = S < T;
		const u64 k = fwd ? mpc_pair_pos(p.s, S, T) : mpc_pair_pos(p.s, T, S);
		const u32 LX = p.s.seq_len[fwd ? S : T], LY = p.s.seq_len[fwd ? T : S];
		const u32 *rec = p.s.packed + p.s.pbase[k];
		const u32 nnz = (u32)(p.s.vbase[k + 1] - p.s.vbase[k]);
		const u32 *ent = rec + LX + LY;
		const u32 *rowv = ent + 2 * (u64)nnz;
		const u32 *m1 = p.p2c1 + p.p2c1_off[a], *m2 = p.p2c2 + p.p2c2_off[b];
		u32 *keys = p.keys + p.coff[ab];
		float *vals = p.vals + p.coff[ab];
		const bool weighted = p.w1 != nullptr;
		const float w12 = weighted ? p.w1[a] * p.w2[b] : 1.0f; // w1*w2 rounded first: buildpostflat.cpp:74 / :96
		for (u32 q = threadIdx.x; q < nnz; q += 64) {
			const u32 row = rowv[q], col = ent[2 * (u64)q + 1];
			// stored (S,T): rows are positions of S (MSA1); stored (T,S): rows are positions of T (MSA2)
			const u32 c1 = fwd ? m1[row] : m1[col];
			const u32 c2 = fwd ? m2[col] : m2[row];
			keys[q] = c1 * p.C2 + c2;
			const float P = __uint_as_float(ent[2 * (u64)q]);
			vals[q] = weighted ? w12 * P : P;
		}
	}
}

// In-order sum of every output cell's run in the sorted records, ONE WAVE PER RUN. The sum of a cell is a chain
//     (((0 + v0) + v1) + v2) ...                      (buildpostflat.cpp:27-30 zeroes Post, :74 / :96 add, w1*w2 == 1.0f)
// whose order is fixed (float addition is not associative and CalcAlnFlat breaks ties on exact values), and the cells on the
// alignment path receive a contribution from almost every (s,t) pair — 250 000 terms at the root of a 1000-sequence tree — so
// the work cannot be one thread per cell (round 1: a serial load -> add chain per thread, 7.7 of the 11.8 s of the
// progressive + refinement tail at 1000 x L~400, profiles/r02g_e2e_timing.log). Here the wave loads 64 consecutive terms at
// once (coalesced) and runs the chain through its lanes (mpc_wave_chain_add, mpc_platform.h: one DPP add per term,
// every term added exactly once in sequence; lanes past the end of the run add +0.0f, exact for these non-negative sums).
//
// Run boundaries of the sorted records: record q opens a run when its cell differs from record q-1's. The opener appends
// {cell, q} to the list of runs (one atomic per wave: the lanes that open a run are counted by a ballot) and closes the
// previous run (run_end[previous cell] = q); the last record closes its own. Cells without records never appear: the matrix
// is zeroed beforehand (buildpostflat.cpp:27-30) and only the ~10^4..10^5 cells near the alignment path of the ~10^6 of a
// 1000-column join get a wave in the reduction below.
__global__ void __launch_bounds__(256) build_post_heads_kernel(const u32 *keys, u64 count, u32 *run_end, u32 *heads, u32 *nheads)
{
	const u32 lane = threadIdx.x & 63u;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	for (u64 q0 = (u64)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); q0 < count; q0 += stride) { // whole waves enter together
		const u64 q = q0 + lane;
		bool head = false;
		u32 cell = 0;
		if (q < count) {
			cell = keys[q];
			const u32 prev = q ? keys[q - 1] : 0u;
			head = q == 0 || cell != prev;
			if (head && q) run_end[prev] = (u32)q;
			if (q + 1 == count) run_end[cell] = (u32)count;
		}
		const u64 bal = __ballot(head);
		if (bal) {
			u32 base = 0;
			if (lane == 0) base = atomicAdd(nheads, (u32)__popcll(bal));
			base = mpc_wave_first(base);
			if (head) {
				const u32 at = base + (u32)__popcll(bal & ((1ull << lane) - 1ull));
				heads[2 * (u64)at] = cell;
				heads[2 * (u64)at + 1] = (u32)q;
			}
		}
	}
}

#define MPC_BP_GROUP 8
// The launch is as long as its slowest SIMD: nearly all records of a join near the root sit in the ~1000-2000 cells along the
// alignment path (~10^5 terms each), a lone wave adds a term every ~7 cycles (round 5's row-broadcast chain; ~14 with the shifted
// partial sums of rounds 1-4; with two waves on a SIMD 12.6 per wave, four share its issue: diag/chain_time.hip), and runs dealt out statically pile several
// heavy ones onto one SIMD (measured: 4.6 ms for a 1.4 ms critical path). So: few resident waves (the host launches two per
// SIMD) that PULL runs from a queue — whichever wave finishes takes the next run.
__global__ void __launch_bounds__(256) build_post_reduce_kernel(const float *vals, const u32 *run_end, const u32 *heads, const u32 *nheads,
	u32 *next_run, float *post)
{
	const u32 lane = threadIdx.x & 63u;
	const u32 nh = mpc_wave_first(*nheads);
	for (;;) {
		const u32 h = mpc_wave_first(atomicAdd(next_run, lane == 0 ? 1u : 0u)); // lane 0 adds 1, the others 0: one atomic per wave (kernels_fb.h)
		if (h >= nh) break;
		const u32 cell = mpc_wave_first(heads[2 * (u64)h]);
		const u64 lo = mpc_wave_first(heads[2 * (u64)h + 1]), hi = mpc_wave_first(run_end[cell]);
		// The chain is serial and the heaviest run (250 000 terms at the root of a 1000-sequence tree) is the critical path of
		// the launch, so its loads must never be waited for: groups of MPC_BP_GROUP chunks of 64 terms, the next
		// group in flight while this one is added (chunks past the end are neither loaded nor added).
		float total = 0.0f; // wave-uniform
		float cur[MPC_BP_GROUP], nxt[MPC_BP_GROUP];
#pragma unroll
		for (int g = 0; g < MPC_BP_GROUP; ++g) {
			cur[g] = 0.0f;
			if (lo + 64u * g < hi) cur[g] = (lo + 64u * g + lane < hi) ? vals[lo + 64u * g + lane] : 0.0f;
		}
		for (u64 q0 = lo; q0 < hi; q0 += 64u * MPC_BP_GROUP) {
			const u64 q1 = q0 + 64u * MPC_BP_GROUP;
#pragma unroll
			for (int g = 0; g < MPC_BP_GROUP; ++g) {
				nxt[g] = 0.0f;
				if (q1 + 64u * g < hi) nxt[g] = (q1 + 64u * g + lane < hi) ? vals[q1 + 64u * g + lane] : 0.0f;
			}
#pragma unroll
			for (int g = 0; g < MPC_BP_GROUP; ++g)
				if (q0 + 64u * g < hi) total = mpc_wave_chain_add(total, cur[g]);
#pragma unroll
			for (int g = 0; g < MPC_BP_GROUP; ++g) cur[g] = nxt[g];
		}
		if (lane == 0) post[cell] = total;
	}
}

// ---- list form: CalcPosteriorFlat3 (buildposterior3flat.cpp:19-85) over an explicit pair list whose
// packed records (kernels_post.h) lie back to back in one shard, X = the MSA1 sequence of the pair.
// Contributions are added in pair-list order (buildposterior3flat.cpp:41, :81): the records are written in that order and the
// sort by cell is stable.
struct BuildPostListParams {
	const u32 *seq_len;
	const u32 *packed;
	const u32 *seq1, *seq2; // per pair: sequence indices
	u32 npairs;
	const u32 *p2c1, *p2c2; // per pair: position -> column maps, concatenated
	const u64 *off1, *off2; // start of each pair's map
	const u64 *coff;        // first record of each pair in keys/vals
	const u64 *rbase;       // word offset of each pair's packed record
	const u32 *nnz;
	u32 C2;
	u32 *keys; // cell of every record
	float *vals;
	float *post; // as in BuildPostParams
	u64 cells;
	u32 *counters;
};

__global__ void __launch_bounds__(64) build_post_list_gen_kernel(BuildPostListParams p)
{
	build_post_zero(p.post, p.cells, p.counters);
	for (u32 q = blockIdx.x; q < p.npairs; q += gridDim.x) {
		const u32 LX = p.seq_len[p.seq1[q]], LY = p.seq_len[p.seq2[q]];
		const u32 nnz = (u32)(p.coff[q + 1] - p.coff[q]);
		const u32 *ent = p.packed + p.rbase[q] + LX + LY;
		const u32 *rowv = ent + 2 * (u64)nnz;
		const u32 *m1 = p.p2c1 + p.off1[q], *m2 = p.p2c2 + p.off2[q];
		u32 *keys = p.keys + p.coff[q];
		float *vals = p.vals + p.coff[q];
		for (u32 e = threadIdx.x; e < nnz; e += 64) {
			keys[e] = m1[rowv[e]] * p.C2 + m2[ent[2 * (u64)e + 1]];
			vals[e] = __uint_as_float(ent[2 * (u64)e]);
		}
	}
}

// ---- small joins: BuildPost by rows, ONE launch ------------------------------------------------------------------------
// The joins inside a shrub of <= 32 sequences (progressive joins + the 100 refinement rounds of MPCFlat::Run: 131 per shrub,
// 50 788 in a 10 000-sequence -super7 run) carry microseconds of work each, and the record / sort / run-list / reduce
// pipeline above costs them ~8 launches apiece: what a join then takes is launch throughput, not arithmetic
// (profiles/r02q_join_overhead.log: 0.9 ms per join on 8 worker contexts). For joins of few pairs this kernel builds the whole
// matrix in one launch, from inputs it reads in page-locked host memory (no upload): one wave per output row col1.
//   * Which entries feed row col1: for every (a, b) of MSA1 x MSA2, the position pos of sequence S = seq1[a] that stands in
//     column col1 (none if S has a gap there), and row pos of the ORDERED matrix M(S,T), T = seq2[b] — the variable-size
//     record store of the relax (kernels_store.h) holds every ordered pair by row, so both stored orientations of
//     buildpostflat.cpp:56-100 are one access. A lane takes one (a, b); 64 pairs at a time, in (a, b) order.
//   * Order of addition (buildpostflat.cpp: s outer, t inner; float addition is not associative): the lanes' entries are laid
//     into an LDS list in lane order (a wave scan of the counts) — that IS (s, t) order — and added one after the other, the
//     lane that owns column col2 % 64 adding into its register for chunk col2 / 64. One pair touches a cell at most once.
#define MPC_BPR_CAP 1024u   // LDS list: entries of one chunk of 64 pairs
#define MPC_BPR_GAP 0xffffffffu

struct BuildPostRowsParams {
	StoreParams s;
	const u32 *seq1, *seq2; // sequence index of every row of MSA1 / MSA2
	u32 n1, n2;
	const u32 *c2p1; // n1 x C1: position of MSA1's row a that stands in column col1, MPC_BPR_GAP if none
	const u32 *p2c2; // position -> column maps of MSA2's rows, concatenated
	const u32 *off2; // n2 + 1: start of each row's map
	u32 C1, C2;
	const float *w1, *w2; // nullptr = all 1.0f
	float *post;          // C1 x C2
	u32 *err;             // set to 1 when a chunk's entries exceed the list (the host then takes the general path)
};

// one output row (col1) of one join: the body of both kernels below
template <int CH, class PARAMS> // 64 * CH >= C2: columns a lane accumulates in registers
__device__ __forceinline__ void build_post_one_row(const PARAMS &p, u32 col1, u32 *lkey, float *lval)
{
	const u32 lane = threadIdx.x;
	const u32 n = p.s.n;
	const u32 npairs = p.n1 * p.n2;
	const unsigned char *padb = (const unsigned char *)p.s.pad;
	float acc[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) acc[c] = 0.0f; // buildpostflat.cpp:27-30
	for (u32 ab0 = 0; ab0 < npairs; ab0 += 64u) {
		const u32 ab = ab0 + lane;
		const bool valid = ab < npairs;
		const u32 a = valid ? ab / p.n2 : 0u, b = valid ? ab % p.n2 : 0u;
		const u32 pos = valid ? p.c2p1[(u64)a * p.C1 + col1] : MPC_BPR_GAP;
		const bool have = pos != MPC_BPR_GAP;
		const u32 S = p.seq1[a], T = p.seq2[b];
		const unsigned char *row0 = padb + 16 * ((u64)p.s.rec_off[mpc_rec_index(n, S, T)] + (have ? pos : 0u));
		// pass 1: stored entries of the row (a block's second entry is real unless it repeats the first column or is the
		// empty-row sentinel; a block's first entry is real unless the row is empty)
		u32 cnt = 0;
		if (have) {
			const unsigned char *blk = row0;
			for (;;) {
				const MpcQuad v = *(const MpcQuad *)blk;
				const u32 c0 = v.z & 0xffffu, dist = v.z >> 16;
				cnt += (c0 != MPC_PAD_SENTINEL ? 1u : 0u) + ((v.w != c0 && v.w != MPC_PAD_SENTINEL) ? 1u : 0u);
				if (dist == 0u) break;
				blk += dist;
			}
		}
		u32 incl = cnt;
		for (int d = 1; d < 64; d <<= 1) {
			const u32 o = __shfl_up(incl, d);
			if (lane >= (u32)d) incl += o;
		}
		const u32 total = mpc_wave_first(__shfl(incl, 63));
		if (total > MPC_BPR_CAP) { // not expected for the joins the host sends here; reported, not guessed
			if (lane == 0) *p.err = 1u;
			continue;
		}
		// pass 2: the entries, in lane order = (a, b) order, rows' entries in column order
		if (have && cnt) {
			u32 at = incl - cnt;
			const u32 *m2 = p.p2c2 + p.off2[b];
			const float w12 = p.w1 ? p.w1[a] * p.w2[b] : 1.0f; // w1*w2 rounded first: buildpostflat.cpp:74 / :96
			const unsigned char *blk = row0;
			for (;;) {
				const MpcQuad v = *(const MpcQuad *)blk;
				const u32 c0 = v.z & 0xffffu, dist = v.z >> 16;
				if (c0 != MPC_PAD_SENTINEL) {
					lkey[at] = m2[c0];
					lval[at] = p.w1 ? w12 * __uint_as_float(v.x) : __uint_as_float(v.x);
					++at;
				}
				if (v.w != c0 && v.w != MPC_PAD_SENTINEL) {
					lkey[at] = m2[v.w];
					lval[at] = p.w1 ? w12 * __uint_as_float(v.y) : __uint_as_float(v.y);
					++at;
				}
				if (dist == 0u) break;
				blk += dist;
			}
		}
		MPC_WAVE_LDS_ORDER();
		// the additions, strictly in list order
		for (u32 e = 0; e < total; ++e) {
			const u32 k = mpc_wave_first(lkey[e]);
			const float v = lval[e];
			const u32 ch = k >> 6;
			const bool mine = (k & 63u) == lane;
#pragma unroll
			for (int c = 0; c < CH; ++c)
				if (ch == (u32)c) acc[c] = mine ? acc[c] + v : acc[c]; // ch is wave-uniform: scalar branches
		}
		MPC_WAVE_LDS_ORDER();
	}
	float *out = p.post + (u64)col1 * p.C2;
#pragma unroll
	for (int c = 0; c < CH; ++c)
		if ((u32)c * 64u + lane < p.C2) out[(u32)c * 64u + lane] = acc[c];
}

template <int CH>
__global__ void __launch_bounds__(64) build_post_rows_kernel(BuildPostRowsParams p)
{
	MPC_DYN_SMEM(smem_raw); // 8 * MPC_BPR_CAP bytes: the list of one chunk of pairs
	u32 *lkey = (u32 *)smem_raw;
	float *lval = (float *)(smem_raw + 4 * MPC_BPR_CAP);
	for (u32 col1 = blockIdx.x; col1 < p.C1; col1 += gridDim.x) build_post_one_row<CH>(p, col1, lkey, lval);
}

// The same for a LIST of independent joins in one launch (mpcgpu_align_alns_batch: the joins of one level of a guide tree): workgroup
// w builds row rows[2 w + 1] of join rows[2 w]; `batch` holds one parameter record per join (page-locked host memory, like the
// joins' inputs). A join's matrix lies at its own place in the batch's buffer (BuildPostRowsParams::post).
template <int CH>
__global__ void __launch_bounds__(64) build_post_rows_batch_kernel(const BuildPostRowsParams *batch, const u32 *rows, u32 nrows)
{
	MPC_DYN_SMEM(smem_raw);
	u32 *lkey = (u32 *)smem_raw;
	float *lval = (float *)(smem_raw + 4 * MPC_BPR_CAP);
	for (u32 w = blockIdx.x; w < nrows; w += gridDim.x) {
		const u32 C2 = batch[rows[2 * w]].C2;
		if (CH == 8 ? C2 > 512u : C2 <= 512u) continue; // (the launch with 8 columns per lane takes the joins up to 512 columns, the other the rest)
		const BuildPostRowsParams p = batch[rows[2 * w]];
		build_post_one_row<CH>(p, rows[2 * w + 1], lkey, lval);
	}
}
