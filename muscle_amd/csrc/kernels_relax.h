// kernels_relax.h — consistency relax, LDS-tiled: relax_dense_kernel (default, dense records: at the end of this
// file) and relax_tile_kernel (row-pointer records, MPCGPU_PAD=rows; described first because the dense kernel only
// changes how a cell reaches its two rows).
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
// {X in [x0,x0+nx)} x {Y in [y0,y0+ny)}, X<Y (nx,ny <= 4). The cells of all those pairs are laid
// end to end and dealt to (slot, lane): cell g lives in slot g/1024, lane g%1024 — every lane of
// every used slot but the last holds a cell (~12 slots for 16 pairs at L=400, r=2). Accumulator and
// packed cell descriptor stay in VGPRs for the whole walk over Z = 0..N-1: for every Z the nx+ny
// matrices M(S,Z) (each one fixed-size 16-byte-aligned record in the padded layout,
// kernels_store.h) are streamed with one global_load_dwordx4 per thread per matrix into LDS and
// every cell of the tile is served from LDS — (nx+ny)/(nx*ny) = 0.5 matrices per (pair,Z) instead
// of 2, with the loads of step Z+1 in flight (staged in registers) while step Z is computed. The
// host orders the tile list in 8x8 super-tiles and deals consecutive tiles to the same XCD (block b
// runs on XCD b % 8), so the workgroups sharing an L2 read the same 64 sequences' records at about
// the same Z.
#pragma once
#include "kernels_store.h"

#define MPC_RT_THREADS 1024 // default workgroup size (MPCGPU_RELAX_WG=512: two 512-thread workgroups per CU on 4x2 blocks)
#define MPC_RT_SLOTS 16
#define MPC_RT_ROW MPC_PAD_ROW // entries of each row handled per merge step (= the block size of the padded layout)
#define MPC_RT_MAXLEN 8191u   // cell descriptor packs x:13 | y:13 | matrix of X:3 | matrix of Y:3; columns < MPC_PAD_SENTINEL

struct RelaxTileParams {
	StoreParams s;    // s.pad / s.pad_stride / s.lcap1 / s.ecap describe the padded records
	const u32 *tiles; // 4 u32 per tile: x0, nx, y0, ny
	u32 ntiles;
	u64 k0, k1; // only pairs in [k0,k1) are relaxed (multi-GPU shard)
	u32 buf_units; // relax_dense_kernel: 16-byte blocks between the two LDS staging buffers, 0 = one buffer
};

struct __attribute__((aligned(16))) MpcU4 { u32 x, y, z, w; };

// MAXSEQ: matrices resident per step; THREADS: workgroup size = cells per slot; NLD: 16-byte loads
// per thread per matrix (record bytes <= NLD * THREADS * 16). The host splits any tile whose cells would need more than MPC_RT_SLOTS slots.
template <int MAXSEQ, int NLD, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? 1 : 4) relax_tile_kernel(RelaxTileParams p)
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
		// ---- cells: the pairs of the tile in (ix, iy) order, their cells end to end; cell g ->
		// slot g / 1024, lane g % 1024. The enumeration is a pure function of the tile, so the
		// epilogue repeats it instead of keeping 16 entry indices alive through the Z walk. Lanes
		// past the last cell get a dummy descriptor (cell (0,0) of matrix 0: valid LDS reads, result
		// never stored), so the walk needs no per-lane validity tests.
		float acc[MPC_RT_SLOTS];
		u32 xy[MPC_RT_SLOTS]; // x | y << 13 | (lds matrix of X) << 26 | (lds matrix of Y) << 29
#pragma unroll
		for (int q = 0; q < MPC_RT_SLOTS; ++q) { acc[q] = 1.0f; xy[q] = 0u; }
		auto for_each_pair = [&](auto &&fn) {
			u32 base = 0;
			for (u32 ix = 0; ix < nx; ++ix) {
				for (u32 iy = 0; iy < ny; ++iy) {
					const u32 X = x0 + ix, Y = y0 + iy;
					if (X >= Y) continue;
					const u64 k = mpc_pair_index(n, X, Y);
					if (k < p.k0 || k >= p.k1) continue;
					const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
					u32 mb = 0; // LDS matrix slot of Y
					if (Y >= x0 && Y < x0 + nx) mb = Y - x0;
					else {
						u32 before = 0; // Y's rank among the Y-range sequences that are not in the X range
						for (u32 j = 0; j < iy; ++j) { const u32 Yj = y0 + j; if (!(Yj >= x0 && Yj < x0 + nx)) ++before; }
						mb = nx + before;
					}
					fn(k, X, Y, nnz, base, (ix << 26) | (mb << 29));
					base += nnz;
				}
			}
			return base;
		};
		// cells of the tile (wave-uniform)
		const u32 total = for_each_pair([&](u64 k, u32 X, u32 Y, u32 nnz, u32 base, u32 ab) {
			const u32 *ent = s.packed + s.pbase[k] + s.seq_len[X] + s.seq_len[Y];
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				const u32 g = (u32)q * THREADS + tid;
				if (g >= base && g - base < nnz) {
					const u32 idx = g - base;
					acc[q] = __uint_as_float(ent[2 * (u64)idx]) * 2.0f; // conspairflat.cpp:29-30
					xy[q] = ent[2 * (u64)nnz + idx] | (ent[2 * (u64)idx + 1] << 13) | ab;
				}
			}
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
						const u32 off = (tid + (u32)r * THREADS) * 16u;
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
						const u32 off = (tid + (u32)r * THREADS) * 16u;
						if (off < rec_bytes) *(MpcU4 *)(m + off) = st[i][r];
					}
				}
			}
		};

		// first lane of my wave, as a scalar: the "does my wave hold cells of slot q" tests below become scalar
		// branches instead of exec-mask regions (the value is the same in all 64 lanes)
		const u32 wave_first = mpc_wave_first(tid & ~63u);
		stage_load(0);
		for (u32 Z = 0; Z < n; ++Z) {
			__syncthreads(); // every wave is done reading step Z-1 from LDS
			stage_store();
			__syncthreads();
			if (Z + 1 < n) stage_load(Z + 1); // in flight while step Z is computed
			// Software pipeline over the slots: the row pointers of slot q+1 are requested from LDS
			// before the entries of slot q, so that round trip overlaps slot q's work (LDS returns in
			// order; waiting for the younger entry reads of slot q also retires these).
			u32 nx_oa = 0, nx_ob = 0, nx_a = 0, nx_b = 0, nx_na = 0, nx_nb = 0;
			auto fetch_rows = [&](int q) {
				u32 c = xy[q];
				MPC_OPAQUE(c); // keep one register per slot: recompute the LDS addresses per step
				const u32 x = c & 0x1fffu, y = (c >> 13) & 0x1fffu;
				nx_oa = __umul24((c >> 26) & 7u, mat_dwords);
				nx_ob = __umul24(c >> 29, mat_dwords);
				const u32 *ma = lds + nx_oa, *mb = lds + nx_ob;
				nx_a = ma[x]; nx_b = mb[y];               // first block of each row
				nx_na = ma[x + 1] - nx_a; nx_nb = mb[y + 1] - nx_b; // blocks in each row
			};
			fetch_rows(0); // unconditional: a lane without a cell holds the dummy descriptor (valid LDS reads)
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				if ((u32)q * THREADS + wave_first < total) { // wave-uniform: my wave holds cells of this slot
					const u32 na = nx_na, nb = nx_nb;
					// a block is 16 bytes and the entry region starts 16-byte aligned: one ds_read_b128 per block
					const MpcU4 *ea = (const MpcU4 *)__builtin_assume_aligned(lds + nx_oa + lcap1, 16) + nx_a;
					const MpcU4 *eb = (const MpcU4 *)__builtin_assume_aligned(lds + nx_ob + lcap1, 16) + nx_b;
					if (q + 1 < MPC_RT_SLOTS) fetch_rows(q + 1); // unconditional, see fetch_rows(0)
					// Block merge of the two sorted rows, one block of MPC_PAD_ROW = 2 entries of each per
					// step (rows are stored in whole blocks, an odd tail is filled with {0.0f, sentinel
					// column}): both blocks are fetched with one aligned 16-byte LDS read each, in flight
					// together, and matched in registers. Row a is walked in ascending z; the partner of an
					// entry in row b (columns are distinct within a row: at most one) is picked by compares.
					// No validity masks: a sentinel only ever equals another sentinel and both carry
					// P = 0.0f. Columns ascend in both rows, so matches are monotone and the block order
					// preserves the reference's order of additions (relaxflat.cpp:16-29 / :41-58 / :78-92:
					// z ascending). An unmatched entry (and a sentinel) contributes pa * 0.0f = +0.0f, which
					// leaves the strictly positive sum bit-for-bit unchanged (the XZ_YZ form of the
					// reference adds such zeros itself).
					float sum = acc[q];
					u32 ia = 0, ib = 0;
					while (ia < na && ib < nb) {
						const MpcU4 va = ea[ia], vb = eb[ib]; // {p0, c0, p1, c1}
						const float pb0 = (va.y == vb.y) ? __uint_as_float(vb.x) : ((va.y == vb.w) ? __uint_as_float(vb.z) : 0.0f);
						const float pb1 = (va.w == vb.y) ? __uint_as_float(vb.x) : ((va.w == vb.w) ? __uint_as_float(vb.z) : 0.0f);
						sum += __uint_as_float(va.x) * pb0; // relaxflat.cpp:27 (w == 1.0f): product rounded, then added
						sum += __uint_as_float(va.z) * pb1;
						// the last column of each block (a sentinel once the row has ended) decides which row moves on
						ia += (va.w <= vb.w) ? 1u : 0u;
						ib += (vb.w <= va.w) ? 1u : 0u;
					}
					acc[q] = sum;
				}
			}
		}
		// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
		for_each_pair([&](u64 k, u32, u32, u32 nnz, u32 base, u32) {
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				const u32 g = (u32)q * THREADS + tid;
				if (g >= base && g - base < nnz)
					s.vnext[s.vbase[k] + (g - base)] = acc[q] / (float)n; // uint -> float, IEEE divide (mysparsemx.cpp:108)
			}
		});
		__syncthreads();
	}
}

// ---- dense-record variant (the default; MPCGPU_PAD=rows selects relax_tile_kernel above) ----------------------
// Same tiles, slots, staging and arithmetic as relax_tile_kernel; only the way a cell finds its two rows differs.
// In the dense padded layout (kernels_store.h) row a of a record starts at block a, so a cell keeps the LDS block
// indices of its two rows (matrix slot * blocks per record + row, 16 bits each) and a merge step is: two aligned
// 16-byte LDS reads, a 2x2 column compare, two products, two adds, and one test whether the row that has to move on
// has another block (the distance to it rides in the upper half of the block's first column word; the two
// probabilities of a block are adjacent dwords, so the two products are one packed multiply). No row
// pointers, no block counts, no matrix-base arithmetic: rows of up to two entries — most rows — take one step.
// PF (tuning variant, MPCGPU_RELAX_PF=1): the first blocks of slot q+1's two rows are requested from LDS before slot q's
// arithmetic, so the round trip of the common one-step merge hides behind it (8 more VGPRs).
// DIAG (measurement only, MPCGPU_RELAX_DIAG, results are wrong): 1 = staging and barriers without the merges, 2 = the merges
// without staging or barriers (every step reads the records of Z = 0) — splits the launch time into its two halves.
template <int MAXSEQ, int NLD, int THREADS, bool PF = false, int DIAG = 0>
__global__ void __launch_bounds__(THREADS, THREADS == 1024 ? 1 : 4) relax_dense_kernel(RelaxTileParams p)
{
	MPC_DYN_SMEM(smem_raw);
	const StoreParams &s = p.s;
	const u32 tid = threadIdx.x;
	const u32 n = s.n;
	const u32 mat_dwords = s.pad_stride;
	const u32 rec_units = mat_dwords >> 2; // 16-byte blocks per record
	u32 *lds = (u32 *)smem_raw;

	const u32 G = gridDim.x < 8u ? gridDim.x : 8u;
	const u32 xcd = blockIdx.x % G, lb = blockIdx.x / G, per_xcd = (gridDim.x - xcd + G - 1u) / G;
	const u32 chunk = (p.ntiles + G - 1u) / G;
	const u32 t_begin = xcd * chunk, t_end = (t_begin + chunk < p.ntiles) ? t_begin + chunk : p.ntiles;

	for (u32 tl = t_begin + lb; tl < t_end; tl += per_xcd) {
		const u32 x0 = p.tiles[4 * tl], nx = p.tiles[4 * tl + 1], y0 = p.tiles[4 * tl + 2], ny = p.tiles[4 * tl + 3];
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
		float acc[MPC_RT_SLOTS];
		u32 ab[MPC_RT_SLOTS]; // LDS block index of row x of M(X,.) | of row y of M(Y,.) << 16; 0 = the dummy cell
#pragma unroll
		for (int q = 0; q < MPC_RT_SLOTS; ++q) { acc[q] = 1.0f; ab[q] = 0u; }
		auto for_each_pair = [&](auto &&fn) {
			u32 base = 0;
			for (u32 ix = 0; ix < nx; ++ix) {
				for (u32 iy = 0; iy < ny; ++iy) {
					const u32 X = x0 + ix, Y = y0 + iy;
					if (X >= Y) continue;
					const u64 k = mpc_pair_index(n, X, Y);
					if (k < p.k0 || k >= p.k1) continue;
					const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
					u32 mb = 0; // LDS matrix slot of Y
					if (Y >= x0 && Y < x0 + nx) mb = Y - x0;
					else {
						u32 before = 0;
						for (u32 j = 0; j < iy; ++j) { const u32 Yj = y0 + j; if (!(Yj >= x0 && Yj < x0 + nx)) ++before; }
						mb = nx + before;
					}
					fn(k, X, Y, nnz, base, ix * rec_units, mb * rec_units);
					base += nnz;
				}
			}
			return base;
		};
		const u32 total = for_each_pair([&](u64 k, u32 X, u32 Y, u32 nnz, u32 base, u32 ua, u32 ub) {
			const u32 *ent = s.packed + s.pbase[k] + s.seq_len[X] + s.seq_len[Y];
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				const u32 g = (u32)q * THREADS + tid;
				if (g >= base && g - base < nnz) {
					const u32 idx = g - base;
					acc[q] = __uint_as_float(ent[2 * (u64)idx]) * 2.0f; // conspairflat.cpp:29-30
					ab[q] = (ua + ent[2 * (u64)nnz + idx]) | ((ub + ent[2 * (u64)idx + 1]) << 16);
				}
			}
		});

		MpcU4 st[MAXSEQ][NLD];
		const u32 rec_bytes = mat_dwords * 4u;
		auto stage_load = [&](u32 Z) {
#pragma unroll
			for (int i = 0; i < MAXSEQ; ++i) {
				if ((u32)i < nseq) {
					const unsigned char *src = (const unsigned char *)(s.pad + ((u64)seq[i] * n + Z) * (u64)mat_dwords);
#pragma unroll
					for (int r = 0; r < NLD; ++r) {
						const u32 off = (tid + (u32)r * THREADS) * 16u;
						MpcU4 v; v.x = 0; v.y = 0; v.z = 0; v.w = 0;
						if (off < rec_bytes) v = *(const MpcU4 *)(src + off);
						st[i][r] = v;
					}
				}
			}
		};
		auto stage_store = [&](u32 buf_dwords) {
#pragma unroll
			for (int i = 0; i < MAXSEQ; ++i) {
				if ((u32)i < nseq) {
					unsigned char *m = (unsigned char *)(lds + buf_dwords + (u32)i * mat_dwords);
#pragma unroll
					for (int r = 0; r < NLD; ++r) {
						const u32 off = (tid + (u32)r * THREADS) * 16u;
						if (off < rec_bytes) *(MpcU4 *)(m + off) = st[i][r];
					}
				}
			}
		};

		const u32 wave_first = mpc_wave_first(tid & ~63u); // scalar: slot tests are scalar branches
		const MpcU4 *blocks0 = (const MpcU4 *)__builtin_assume_aligned(lds, 16);
		// Two LDS staging buffers when they fit (p.buf_units != 0): step Z is computed out of buffer Z&1 while the
		// records of step Z+1, loaded into registers during that computation, go into the other buffer — one
		// barrier per step (nobody reads that other buffer any more: its readers, step Z-1, passed the previous
		// barrier; nobody reads it yet: its readers, step Z+1, wait at this step's barrier). With one buffer the
		// store has to wait for all readers of the previous step and the readers for the store: two barriers.
		const u32 bstride = p.buf_units;
		stage_load(0);
		if (bstride) {
			__syncthreads(); // the previous tile's readers are done
			stage_store(0);
			__syncthreads();
			if (n > 1) stage_load(1);
		}
		for (u32 Z = 0; Z < n; ++Z) {
			if (!bstride && (DIAG != 2 || Z == 0)) {
				__syncthreads(); // every wave is done reading step Z-1 from LDS
				stage_store(0);
				__syncthreads();
				if (Z + 1 < n) stage_load(Z + 1); // in flight while step Z is computed
			}
			const MpcU4 *blocks = blocks0 + (bstride ? (Z & 1u) * bstride : 0u);
			if (PF) {
				MpcU4 nva, nvb; // first blocks of the next slot's rows, in flight
				u32 nia = 0, nib = 0;
				auto prefetch = [&](int q) {
					u32 c = ab[q];
					MPC_OPAQUE(c);
					nia = c & 0xffffu; nib = c >> 16;
					nva = blocks[nia]; nvb = blocks[nib];
				};
				prefetch(0); // unconditional: a lane without a cell holds the dummy descriptor (block 0 of matrix 0)
#pragma unroll
				for (int q = 0; q < MPC_RT_SLOTS; ++q) {
					if ((u32)q * THREADS + wave_first < total) {
						u32 ia = nia, ib = nib;
						MpcU4 va = nva, vb = nvb;
						if (q + 1 < MPC_RT_SLOTS) prefetch(q + 1);
						float sum = acc[q];
						for (;;) { // the same merge as below, the loads moved to the end of the step
							const u32 ca0 = va.z & 0xffffu, cb0 = vb.z & 0xffffu;
							const float pb0 = (ca0 == cb0) ? __uint_as_float(vb.x) : ((ca0 == vb.w) ? __uint_as_float(vb.y) : 0.0f);
							const float pb1 = (va.w == cb0) ? __uint_as_float(vb.x) : ((va.w == vb.w) ? __uint_as_float(vb.y) : 0.0f);
							sum += __uint_as_float(va.x) * pb0;
							sum += __uint_as_float(va.y) * pb1;
							const bool adv_a = va.w <= vb.w, adv_b = vb.w <= va.w;
							const u32 da = va.z >> 16, db = vb.z >> 16;
							if ((adv_a && da == 0u) || (adv_b && db == 0u)) break;
							ia += adv_a ? da : 0u;
							ib += adv_b ? db : 0u;
							va = blocks[ia]; vb = blocks[ib];
						}
						acc[q] = sum;
					}
				}
			} else
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				if (DIAG != 1 && (u32)q * THREADS + wave_first < total) { // wave-uniform: my wave holds cells of this slot
					u32 c = ab[q];
					MPC_OPAQUE(c); // one register per slot: the two block indices are unpacked per step
					u32 ia = c & 0xffffu, ib = c >> 16;
					float sum = acc[q];
					// Block merge of the two sorted rows (see relax_tile_kernel for the order-of-additions argument: z
					// ascending, unmatched entries and sentinels add +0.0f). The row whose last column is not larger
					// moves to its next block; when that row has none the merge is over (everything left in the other
					// row lies beyond it).
					for (;;) {
						const MpcU4 va = blocks[ia], vb = blocks[ib]; // {p0, p1, c0 | delta << 16, c1}
						const u32 ca0 = va.z & 0xffffu, cb0 = vb.z & 0xffffu;
						const float pb0 = (ca0 == cb0) ? __uint_as_float(vb.x) : ((ca0 == vb.w) ? __uint_as_float(vb.y) : 0.0f);
						const float pb1 = (va.w == cb0) ? __uint_as_float(vb.x) : ((va.w == vb.w) ? __uint_as_float(vb.y) : 0.0f);
						sum += __uint_as_float(va.x) * pb0; // relaxflat.cpp:27 (w == 1.0f): product rounded, then added
						sum += __uint_as_float(va.y) * pb1;
						const bool adv_a = va.w <= vb.w, adv_b = vb.w <= va.w;
						const u32 da = va.z >> 16, db = vb.z >> 16; // blocks to the next block of the row, 0: none
						if ((adv_a && da == 0u) || (adv_b && db == 0u)) break;
						ia += adv_a ? da : 0u;
						ib += adv_b ? db : 0u;
					}
					acc[q] = sum;
				}
			}
			if (bstride) {
				if (Z + 1 < n) stage_store(((Z + 1) & 1u) * bstride * 4u); // records of step Z+1 -> the other buffer
				__syncthreads();
				if (Z + 2 < n) stage_load(Z + 2); // in flight while step Z+1 is computed
			}
		}
		// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
		for_each_pair([&](u64 k, u32, u32, u32 nnz, u32 base, u32, u32) {
#pragma unroll
			for (int q = 0; q < MPC_RT_SLOTS; ++q) {
				const u32 g = (u32)q * THREADS + tid;
				if (g >= base && g - base < nnz)
					s.vnext[s.vbase[k] + (g - base)] = acc[q] / (float)n; // uint -> float, IEEE divide (mysparsemx.cpp:108)
			}
		});
		__syncthreads();
	}
}
