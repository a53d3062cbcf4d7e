// kernels_relax.h — consistency relax, LDS-tiled ("relax_tile_kernel").
//
// Replaces MPCFlat::ConsPair (conspairflat.cpp:10-110) -> RelaxFlat_{XZ_ZY,ZX_ZY,XZ_YZ}
// (relaxflat.cpp:4-94) -> MySparseMx::UpdateFromPost (mysparsemx.cpp:87-113) for a TILE of pairs
// at a time. Same arithmetic as relax_kernel (kernels_store.h): per stored cell (x,y) of (X,Y)
//     acc = 2*P_XY(x,y);  for Z = 0..N-1: acc += sum_z M(X,Z)(x,z) * M(Y,Z)(y,z)  (z ascending)
//     P'  = acc / N
// product rounded, then added (no FMA), Z ascending, z ascending: bit-identical to the reference.
//
// Why a tile. One cell-per-thread gather (relax_kernel) re-reads M(X,Z) for every Y and M(Y,Z)
// for every X: 2 matrices (~16 KB at L=400, r=2) per (pair,Z), ~8 TB per iteration at N=1000, all
// of it as short latency-bound gathers. Here a workgroup of 1024 threads owns the pairs
// {X in [x0,x0+nx)} x {Y in [y0,y0+ny)}, X<Y (nx,ny <= 4, <= 16 register "slots" of 1024 cells)
// and walks Z = 0..N-1 once: for every Z the nx+ny matrices M(S,Z) (row pointers + entries, both
// one fixed-size 16-byte-aligned record in the padded layout, kernels_store.h) are streamed with
// one global_load_dwordx4 per thread per matrix into LDS and every pair of the tile is served from LDS — (nx+ny)/(nx*ny) = 0.5 matrices per (pair,Z) instead of 2, and the
// loads of step Z+1 are in flight (staged in registers) while step Z is computed. Accumulators
// and cell coordinates stay in VGPRs for the whole walk. The host orders the tile list in 8x8
// super-tiles and deals consecutive tiles to the same XCD (block b runs on XCD b % 8), so the
// workgroups sharing an L2 read the same 64 sequences' slabs at about the same Z.
#pragma once
#include "kernels_store.h"

#define MPC_RT_THREADS 1024
#define MPC_RT_SLOTS 16
#define MPC_RT_ROW 4 // entries per row the register-matched fast path handles

struct RelaxTileParams {
	StoreParams s;    // s.pad / s.pad_stride / s.lcap1 / s.ecap describe the padded records
	const u32 *tiles; // 4 u32 per tile: x0, nx, y0, ny
	u32 ntiles;
	u64 k0, k1; // only pairs in [k0,k1) are relaxed (multi-GPU shard)
};

struct __attribute__((aligned(16))) MpcU4 { u32 x, y, z, w; };

// MAXSEQ: matrices resident per step; NLD: 16-byte loads per thread per matrix (record bytes <=
// NLD * 16 KiB). A pair with more than 1024 cells takes several slots — the host splits any tile
// that would need more than MPC_RT_SLOTS slots.
template <int MAXSEQ, int NLD>
__global__ void __launch_bounds__(MPC_RT_THREADS) relax_tile_kernel(RelaxTileParams p)
{
	MPC_DYN_SMEM(smem_raw);
	const StoreParams &s = p.s;
	const u32 tid = threadIdx.x;
	const u32 n = s.n;
	const u32 mat_dwords = s.pad_stride;
	const u32 lcap1 = s.lcap1;
	u32 *lds = (u32 *)smem_raw;

	// XCD-aware static schedule: the tile list is cut into 8 contiguous ranges, one per XCD
	// (block b is placed on XCD b % 8 — affinity only, any placement is correct); the workgroups of
	// one XCD walk their range round-robin, so at any time they sit on consecutive tiles.
	const u32 G = gridDim.x < 8u ? gridDim.x : 8u;
	const u32 xcd = blockIdx.x % G, lb = blockIdx.x / G, per_xcd = (gridDim.x - xcd + G - 1u) / G;
	const u32 chunk = (p.ntiles + G - 1u) / G;
	const u32 t_begin = xcd * chunk, t_end = (t_begin + chunk < p.ntiles) ? t_begin + chunk : p.ntiles;

	for (u32 tl = t_begin + lb; tl < t_end; tl += per_xcd) {
		const u32 x0 = p.tiles[4 * tl], nx = p.tiles[4 * tl + 1], y0 = p.tiles[4 * tl + 2], ny = p.tiles[4 * tl + 3];
		// resident sequences: the X range, then the part of the Y range not already in it
		u32 seq[MAXSEQ];
		u32 nseq = 0;
#pragma unroll
		for (int i = 0; i < MAXSEQ; ++i) seq[i] = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i)
			if ((u32)i < nx && nseq < (u32)MAXSEQ) {
#pragma unroll
				for (int q = 0; q < MAXSEQ; ++q) if ((u32)q == nseq) seq[q] = x0 + i;
				++nseq;
			}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const u32 Y = y0 + i;
			if ((u32)i < ny && !(Y >= x0 && Y < x0 + nx) && nseq < (u32)MAXSEQ) {
#pragma unroll
				for (int q = 0; q < MAXSEQ; ++q) if ((u32)q == nseq) seq[q] = Y;
				++nseq;
			}
		}
		// ---- slots: (pair, 1024-cell chunk) -> this thread's cell, accumulator, LDS matrix slots.
		// The enumeration order is a pure function of the tile, so the epilogue repeats it instead of
		// keeping 16 entry indices alive in registers through the Z walk.
		float acc[MPC_RT_SLOTS];
		u32 xy[MPC_RT_SLOTS];    // (x << 16) | y ; 0xffffffff = no cell
		u32 sl_ab[MPC_RT_SLOTS]; // wave-uniform: (lds matrix of X) | (lds matrix of Y) << 8 ; 0xffff = unused slot
#pragma unroll
		for (int q = 0; q < MPC_RT_SLOTS; ++q) { acc[q] = 0.0f; xy[q] = 0xffffffffu; sl_ab[q] = 0xffffu; }
		auto for_each_slot = [&](auto &&fn) {
			u32 slot = 0;
			for (u32 ix = 0; ix < nx; ++ix) {
				for (u32 iy = 0; iy < ny; ++iy) {
					const u32 X = x0 + ix, Y = y0 + iy;
					if (X >= Y) continue;
					const u64 k = mpc_pair_index(n, X, Y);
					if (k < p.k0 || k >= p.k1) continue;
					const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
					// LDS matrix slots of X and Y
					u32 mb = 0;
					if (Y >= x0 && Y < x0 + nx) mb = Y - x0;
					else {
						u32 before = 0; // Y's rank among the Y-range sequences that are not in the X range
						for (u32 j = 0; j < iy; ++j) { const u32 Yj = y0 + j; if (!(Yj >= x0 && Yj < x0 + nx)) ++before; }
						mb = nx + before;
					}
					for (u32 c0 = 0; c0 < nnz; c0 += MPC_RT_THREADS) {
						fn(slot, k, X, Y, nnz, c0 + tid, ix | (mb << 8));
						++slot;
					}
				}
			}
		};
		for_each_slot([&](u32 slot, u64 k, u32 X, u32 Y, u32 nnz, u32 idx, u32 ab) {
			const u32 *ent = s.packed + s.pbase[k] + s.seq_len[X] + s.seq_len[Y];
			float a0 = 0.0f;
			u32 c = 0xffffffffu;
			if (idx < nnz) {
				a0 = __uint_as_float(ent[2 * (u64)idx]) * 2.0f; // conspairflat.cpp:29-30
				c = (ent[2 * (u64)nnz + idx] << 16) | ent[2 * (u64)idx + 1];
			}
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q)
				if ((u32)q == slot) { acc[q] = a0; xy[q] = c; sl_ab[q] = ab; }
		});

		// ---- walk Z with register-staged prefetch: record (A,Z) of the padded layout is copied
		// verbatim, 16 bytes per thread per load, from pad + (A*n+Z)*stride (wave-uniform base in
		// SGPRs, advanced by one record per step) to LDS matrix slot i.
		MpcU4 st[MAXSEQ][NLD];
		const u32 rec_bytes = mat_dwords * 4u;
		auto stage_load = [&](u32 Z) {
#pragma unroll
			for (int i = 0; i < MAXSEQ; ++i) {
				if ((u32)i < nseq) {
					const unsigned char *src = (const unsigned char *)(s.pad + ((u64)seq[i] * n + Z) * (u64)mat_dwords);
#pragma unroll
					for (int r = 0; r < NLD; ++r) {
						const u32 off = (tid + (u32)r * MPC_RT_THREADS) * 16u;
						MpcU4 v; v.x = 0; v.y = 0; v.z = 0; v.w = 0;
						if (off < rec_bytes) v = *(const MpcU4 *)(src + off);
						st[i][r] = v;
					}
				}
			}
		};
		auto stage_store = [&]() {
#pragma unroll
			for (int i = 0; i < MAXSEQ; ++i) {
				if ((u32)i < nseq) {
					unsigned char *m = (unsigned char *)(lds + (u32)i * mat_dwords);
#pragma unroll
					for (int r = 0; r < NLD; ++r) {
						const u32 off = (tid + (u32)r * MPC_RT_THREADS) * 16u;
						if (off < rec_bytes) *(MpcU4 *)(m + off) = st[i][r];
					}
				}
			}
		};

		stage_load(0);
		for (u32 Z = 0; Z < n; ++Z) {
			__syncthreads(); // every wave is done reading step Z-1 from LDS
			stage_store();
			__syncthreads();
			if (Z + 1 < n) stage_load(Z + 1); // in flight while step Z is computed
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				if (sl_ab[q] != 0xffffu) { // wave-uniform
					const u32 *ma = lds + (sl_ab[q] & 0xffu) * mat_dwords;
					const u32 *mb = lds + (sl_ab[q] >> 8) * mat_dwords;
					u32 c = xy[q];
					MPC_OPAQUE(c); // keep one register per slot: recompute the two LDS addresses per step
					const bool have = c != 0xffffffffu;
					if (__ballot(have) == 0) continue; // this wave holds no cell of the slot
					const u32 x = have ? (c >> 16) : 0u, y = have ? (c & 0xffffu) : 0u;
					u32 a = ma[x], b = mb[y];
					u32 na = ma[x + 1] - a, nb = mb[y + 1] - b;
					if (!have) { na = 0; nb = 0; }
					// lcap1 is even and the LDS base 16-byte aligned: entries are 8-byte aligned (ds_read_b64)
					const MpcEnt *ea = (const MpcEnt *)__builtin_assume_aligned(ma + lcap1, 8);
					const MpcEnt *eb = (const MpcEnt *)__builtin_assume_aligned(mb + lcap1, 8);
					// Block merge of the two sorted rows, MPC_RT_ROW entries of each per step: all LDS reads
					// of a step are in flight together and the match is done in registers, so a cell costs
					// ceil(na/4)+ceil(nb/4)-1 dependent LDS round trips (1 for ~97 % of the rows) instead of
					// one per merged entry — and a wave only waits for its longest row pair at that rate.
					// Row a is walked in ascending z; the partner of an entry in row b (columns are distinct
					// within a row: at most one) is picked by compares. Columns ascend in both rows, so
					// matches are monotone and the block order preserves the reference's order of additions
					// (relaxflat.cpp:16-29 / :41-58 / :78-92: z ascending). An unmatched or absent entry
					// contributes pa * 0.0f = +0.0f, which leaves the strictly positive sum bit-for-bit
					// unchanged (the XZ_YZ form of the reference adds such explicit zeros itself).
					float sum = acc[q];
					u32 ia = 0, ib = 0;
					while (ia < na && ib < nb) {
						const u32 ra = na - ia, rb = nb - ib; // entries left in each row (>= 1)
						MpcEnt va[MPC_RT_ROW], vb[MPC_RT_ROW];
#pragma unroll
						for (int r = 0; r < MPC_RT_ROW; ++r) { va[r] = ea[a + ia + r]; vb[r] = eb[b + ib + r]; } // reads past a row end are masked
						u32 ca[MPC_RT_ROW], cb[MPC_RT_ROW];
#pragma unroll
						for (int r = 0; r < MPC_RT_ROW; ++r) {
							ca[r] = ((u32)r < ra) ? va[r].c : 0xffffffffu;
							cb[r] = ((u32)r < rb) ? vb[r].c : 0xfffffffeu;
						}
#pragma unroll
						for (int r = 0; r < MPC_RT_ROW; ++r) {
							float pb = 0.0f;
#pragma unroll
							for (int t2 = MPC_RT_ROW - 1; t2 >= 0; --t2) pb = (ca[r] == cb[t2]) ? __uint_as_float(vb[t2].p) : pb;
							const float pa = ((u32)r < ra) ? __uint_as_float(va[r].p) : 0.0f;
							sum += pa * pb; // relaxflat.cpp:27 (w == 1.0f): product rounded, then added
						}
						// largest column present in each block decides which row moves on
						u32 amax = ca[0], bmax = cb[0];
#pragma unroll
						for (int r = 1; r < MPC_RT_ROW; ++r) {
							amax = ((u32)r < ra) ? ca[r] : amax;
							bmax = ((u32)r < rb) ? cb[r] : bmax;
						}
						ia += (amax <= bmax) ? (u32)MPC_RT_ROW : 0u;
						ib += (bmax <= amax) ? (u32)MPC_RT_ROW : 0u;
					}
					acc[q] = sum;
				}
			}
		}
		// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
		for_each_slot([&](u32 slot, u64 k, u32, u32, u32 nnz, u32 idx, u32) {
			float v = 0.0f;
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q)
				if ((u32)q == slot) v = acc[q];
			if (idx < nnz) s.vnext[s.vbase[k] + idx] = v / (float)n; // uint -> float, IEEE divide (mysparsemx.cpp:108)
		});
		__syncthreads();
	}
}
