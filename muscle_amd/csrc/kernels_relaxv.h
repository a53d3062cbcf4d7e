// kernels_relaxv.h — consistency relax over variable-size dense records: relax_var_kernel (the default).
//
// Replaces MPCFlat::ConsPair (conspairflat.cpp:10-110) -> RelaxFlat_{XZ_ZY,ZX_ZY,XZ_YZ} (relaxflat.cpp:4-94) ->
// MySparseMx::UpdateFromPost (mysparsemx.cpp:87-113) for a TILE of pairs at a time, with the arithmetic and the order of
// additions of relax_kernel (kernels_store.h): per stored cell (x,y) of (X,Y)
//     acc = 2*P_XY(x,y);  for Z = 0..N-1: acc += sum_z M(X,Z)(x,z) * M(Y,Z)(y,z)  (z ascending);  P' = acc / N
// product rounded, then added (no FMA): bit-identical to the reference.
//
// What the round-2 measurements of relax_dense_kernel (fixed-size records, register staging, two barriers per Z;
// profiles/r02a_*) said, and what this kernel does about each:
//  * every record was padded to the worst of the N^2 (13.3 KB at 1000 x L~400 against 8.1 KB mean): 40 % of the HBM
//    traffic, of the LDS writes and of the LDS footprint carried nothing. Records are now exactly len(A) first blocks +
//    their own overflow blocks (kernels_store.h, var_*), found through a block-offset table read with scalar loads one
//    step ahead, and packed back to back in LDS in the order of the tile's sequences.
//  * staging went HBM -> VGPR -> ds_write_b128 -> LDS between two barriers (0.8 us of every 8.7 us step, 32 VGPRs).
//    It is now LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write issue): each wave moves 1 KiB per instruction.
//  * with the smaller footprint two staging buffers fit the CU's 160 KB for almost every tile (the host checks each
//    tile's worst step with var_tile_fit_kernel and splits the few that do not): the DMA of step Z+1 is in flight
//    while step Z is computed and a step costs ONE barrier. With one buffer (nbuf = 1; two 512-thread workgroups per CU,
//    each hiding the other's DMA wait) it costs two.
//  * a cell found its rows through two per-lane 16-bit block indices relative to fixed record slots; records now sit
//    at step-dependent LDS addresses, so the cells of a pair are laid out in whole waves (a pair's cell range is rounded
//    up to 64): the two records a (wave, slot) reads are wave-uniform, their LDS bases are two v_readlane of a
//    register that holds the step's 8 record bases, and a lane keeps only the byte offsets of its two rows.
#pragma once
#include "kernels_store.h"

#define MPC_RV_MAXSEQ 8        // records resident per step (4 + 4 sequences)
#define MPC_RV_MAXLEN 4095u    // a cell keeps 16 * row and 16 * column in 16 bits each; a block's distance field is 16 bits of bytes
#define MPC_RV_TAB_BYTES 512   // pair table of the tile at the head of the dynamic LDS: 16 pairs x 8 dwords
#define MPC_RV_WAVE 64u

struct RelaxVarParams {
	StoreParams s;
	const u32 *tiles; // 4 u32 per tile: x0, nx, y0, ny
	u32 ntiles;
	u64 k0, k1;    // only pairs in [k0,k1) are relaxed (multi-GPU shard)
	u32 nbuf;      // LDS staging buffers: 2 = DMA of step Z+1 under the merges of step Z, 1 = DMA, wait, merge
	u32 buf_bytes; // capacity of one staging buffer (the second starts buf_bytes after the first)
	u32 *tile_next; // 8 counters, zeroed before the launch: next tile of each XCD's range
};

// THREADS: workgroup size; MAXSLOTS: cells per lane (the host splits any tile whose wave-aligned cells need more);
// DIAG (measurement only, results wrong): 1 = staging and barriers only.
// WGS: workgroups per CU the register allocation has to allow (waves per SIMD = WGS * THREADS / 256).
template <int THREADS, int MAXSLOTS, int WGS, int DIAG = 0>
__global__ void __launch_bounds__(THREADS, WGS * THREADS / 256) relax_var_kernel(RelaxVarParams p)
{
	MPC_DYN_SMEM(smem_raw);
	const StoreParams &s = p.s;
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63u;
	const u32 n = s.n;
	constexpr u32 NWAVES = THREADS / 64;
	const u32 wave = mpc_wave_first(tid >> 6); // scalar
	u32 *ptab = (u32 *)smem_raw;                       // [16][8]: cell base, nnz, sel, k lo, k hi, cells (aligned), -, -
	const unsigned char *padb = (const unsigned char *)s.pad;
	mpc_const_u32p rec_off = MPC_CONST_U32(s.rec_off); // read with scalar loads
	const u32 lds0 = mpc_lds_addr(smem_raw);            // 32-bit LDS address of the dynamic LDS

	// XCD-aware schedule (block b runs on XCD b % 8 — affinity only): the tile list is cut into 8 contiguous ranges and the
	// workgroups of one XCD take the tiles of their range one after the other from a counter (neighbouring tiles share the Y
	// records in that XCD's L2); a workgroup whose range is used up takes from the other ranges, so nobody waits at the end for
	// a workgroup that happened to get the dearer tiles (tiles differ: 4x4 / 4x2, stored cells).
	const u32 G = gridDim.x < 8u ? gridDim.x : 8u;
	const u32 xcd = blockIdx.x % G;
	const u32 chunk = (p.ntiles + G - 1u) / G;

	for (;;) {
		__syncthreads(); // the previous tile is done with the pair table and the staging buffers
		if (tid == 0) {
			u32 got = 0xffffffffu;
			for (u32 k = 0; k < G && got == 0xffffffffu; ++k) {
				const u32 r = (xcd + k) % G;
				const u32 t_begin = r * chunk, t_end = (t_begin + chunk < p.ntiles) ? t_begin + chunk : p.ntiles;
				if (t_begin >= t_end) continue;
				const u32 t = atomicAdd(&p.tile_next[r], 1u);
				if (t < t_end - t_begin) got = t_begin + t;
			}
			ptab[8 * 15 + 7] = got;
		}
		__syncthreads();
		const u32 tl = mpc_wave_first(ptab[8 * 15 + 7]);
		if (tl == 0xffffffffu) break;
		const u32 x0 = p.tiles[4 * tl], nx = p.tiles[4 * tl + 1], y0 = p.tiles[4 * tl + 2], ny = p.tiles[4 * tl + 3];
		// resident sequences: the X range, then the part of the Y range not already in it (wave-uniform, SGPRs)
		u32 seq[MPC_RV_MAXSEQ];
		u32 nseq = 0;
#pragma unroll
		for (int i = 0; i < MPC_RV_MAXSEQ; ++i) seq[i] = 0;
#pragma unroll
		for (int i = 0; i < 4; ++i)
			if ((u32)i < nx) {
#pragma unroll
				for (int q = 0; q < MPC_RV_MAXSEQ; ++q) if ((u32)q == nseq) seq[q] = x0 + i;
				++nseq;
			}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const u32 Y = y0 + i;
			if ((u32)i < ny && !(Y >= x0 && Y < x0 + nx)) {
#pragma unroll
				for (int q = 0; q < MPC_RV_MAXSEQ; ++q) if ((u32)q == nseq) seq[q] = Y;
				++nseq;
			}
		}
		// ---- pair table: pair (ix, iy) of the tile -> first cell, stored cells, the LDS record slots of its two sequences.
		// Lane q of wave 0 looks after pair q; a 16-lane inclusive scan lays the pairs' cell ranges end to end, each rounded
		// up to whole waves so that the 64 cells of any (wave, slot) belong to ONE pair.
		if (tid < 64u) {
			const u32 ix = lane / 4u, iy = lane % 4u;
			const u32 X = x0 + ix, Y = y0 + iy;
			u32 nnz = 0, sel = 0;
			u64 k = 0;
			if (lane < 16u && ix < nx && iy < ny && X < Y) {
				k = mpc_pair_index(n, X, Y);
				if (k >= p.k0 && k < p.k1) nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
				u32 mb = 0; // record slot of Y
				if (Y >= x0 && Y < x0 + nx) mb = Y - x0;
				else {
					u32 before = 0; // Y's rank among the Y-range sequences that are not in the X range
					for (u32 j = 0; j < iy; ++j) { const u32 Yj = y0 + j; if (!(Yj >= x0 && Yj < x0 + nx)) ++before; }
					mb = nx + before;
				}
				sel = ix | (mb << 8);
			}
			const u32 cells = (nnz + MPC_RV_WAVE - 1u) & ~(MPC_RV_WAVE - 1u);
			u32 incl = cells;
			for (int d = 1; d < 16; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (lane >= (u32)d) incl += o;
			}
			if (lane < 16u) {
				u32 *e = ptab + 8 * lane;
				e[0] = incl - cells; e[1] = nnz; e[2] = sel; e[3] = (u32)k; e[4] = (u32)(k >> 32); e[5] = cells;
			}
			if (lane == 15u) ptab[8 * 15 + 6] = incl; // all cells of the tile
		}
		__syncthreads();
		const u32 total = mpc_wave_first(ptab[8 * 15 + 6]);

		// ---- my cells: slot q of this wave covers cells [q*THREADS + wave*64, +64) — one pair (found by a scalar scan of the
		// table), or nothing. Accumulator and row offsets stay in VGPRs for the whole walk; the pair's two record slots go into
		// lane q of `vsel` (one register per wave for all slots).
		float acc[MAXSLOTS];
		u32 xy[MAXSLOTS]; // 16 * x | (16 * y) << 16: byte offsets of row x of M(X,.) and row y of M(Y,.) inside their records
		u32 vsel = 0;
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q) {
			MPC_SCHED_BARRIER();
			acc[q] = 1.0f; xy[q] = 0u;
			const u32 g0 = (u32)q * THREADS + wave * 64u;
			if (g0 < total) {
				u32 pi = 0;
				for (u32 j = 1; j < 16u; ++j) if (mpc_wave_first(ptab[8 * j]) <= g0 && mpc_wave_first(ptab[8 * j + 5]) != 0u) pi = j;
				const u32 base = mpc_wave_first(ptab[8 * pi]), nnz = mpc_wave_first(ptab[8 * pi + 1]), sel = mpc_wave_first(ptab[8 * pi + 2]);
				const u64 k = (u64)mpc_wave_first(ptab[8 * pi + 3]) | ((u64)mpc_wave_first(ptab[8 * pi + 4]) << 32);
				vsel = mpc_write_lane(vsel, sel, (u32)q);
				const u32 idx = g0 + lane - base;
				if (idx < nnz) {
					const u32 *ent = s.packed + s.pbase[k] + s.seq_len[s.pair_x[k]] + s.seq_len[s.pair_y[k]];
					acc[q] = __uint_as_float(ent[2 * (u64)idx]) * 2.0f; // conspairflat.cpp:29-30
					xy[q] = (ent[2 * (u64)nnz + idx] << 4) | (ent[2 * (u64)idx + 1] << 20);
				}
			}
		}

		// ---- walk Z. Record (A,Z) = blocks [rec_off[A*n+Z], rec_off[A*n+Z+1]) of `pad`; the step's records are packed
		// back to back in the staging buffer in the order of seq[]. cur_off/cur_sz describe the step whose DMA is issued
		// next; they are loaded (scalar loads: wave-uniform addresses) one step before they are used.
		u32 nxt_off[MPC_RV_MAXSEQ], nxt_sz[MPC_RV_MAXSEQ];
		auto load_table = [&](u32 Z) {
#pragma unroll
			for (int i = 0; i < MPC_RV_MAXSEQ; ++i) {
				nxt_off[i] = 0; nxt_sz[i] = 0;
				if ((u32)i < nseq) {
					mpc_const_u32p ro = rec_off + ((u64)seq[i] * n + Z);
					const u32 a = ro[0], b = ro[1];
					nxt_off[i] = a; nxt_sz[i] = b - a;
				}
			}
		};
		// issues the DMA of the step described by nxt_* into staging buffer `buf`; returns the register holding the LDS byte
		// address of record i in lane i
		auto issue_dma = [&](u32 buf) -> u32 {
			u32 vb = 0;
			u32 at = MPC_RV_TAB_BYTES + buf * p.buf_bytes; // byte offset in the dynamic LDS
#pragma unroll
			for (int i = 0; i < MPC_RV_MAXSEQ; ++i) {
				if ((u32)i < nseq) {
					vb = mpc_write_lane(vb, lds0 + at, (u32)i);
					const u32 sz = nxt_sz[i];
					for (u32 c0 = wave * 64u; c0 < sz; c0 += NWAVES * 64u) { // chunks of 64 blocks = 1 KiB, dealt round-robin to the waves
						const u32 blk = c0 + lane;
						if (blk < sz)
							mpc_dma16(padb + 16 * ((u64)nxt_off[i] + blk), smem_raw + at + 16 * c0);
					}
					at += 16u * sz;
				}
			}
			return vb;
		};

		u32 vbase_cur = 0, vbase_nxt = 0; // lane i: LDS byte address of record i of the step being merged / being staged
		const u32 wave_first = wave * 64u;
		if (p.nbuf == 2) {
			load_table(0);
			vbase_cur = issue_dma(0);
			if (n > 1) load_table(1);
			mpc_dma_wait();
		}
		for (u32 Z = 0; Z < n; ++Z) {
			if (p.nbuf == 2) {
				__syncthreads(); // step Z's records have landed (every wave waited for its own DMA) and step Z-1's readers are done
				if (Z + 1 < n) {
					vbase_nxt = issue_dma((Z + 1) & 1u);
					if (Z + 2 < n) load_table(Z + 2);
				}
			} else {
				if (Z == 0) load_table(0);
				__syncthreads(); // step Z-1's readers are done
				vbase_cur = issue_dma(0);
				if (Z + 1 < n) load_table(Z + 1);
				mpc_dma_wait();
				__syncthreads();
			}
#pragma unroll
			for (int q = 0; q < MAXSLOTS; ++q) {
				if (DIAG != 1 && (u32)q * THREADS + wave_first < total) { // wave-uniform: my wave holds cells of this slot
					const u32 sel = mpc_read_lane(vsel, (u32)q);
					const u32 sa = mpc_read_lane(vbase_cur, sel & 0xffu), sb = mpc_read_lane(vbase_cur, sel >> 8);
					u32 c = xy[q];
					MPC_OPAQUE(c); // one register per slot: the two row offsets are unpacked per step
					u32 ia = sa + (c & 0xffffu), ib = sb + (c >> 16);
					float sum = acc[q];
					// Block merge of the two sorted rows, one block of 2 entries of each per step: two aligned 16-byte LDS reads,
					// a 2x2 column compare, two products, two adds (z ascending: relaxflat.cpp:16-29 / :41-58 / :78-92; an unmatched
					// entry and a sentinel {0.0f, 0x1fff} contribute pa * 0.0f = +0.0f, which leaves the strictly positive sum
					// unchanged bit for bit). The row whose last column is not larger moves to its next block (the distance in
					// bytes rides in the upper half of the block's first column word); when that row has none the merge is over.
					for (;;) {
						const MpcQuad va = mpc_lds_load16(ia), vb = mpc_lds_load16(ib); // {p0, p1, c0 | dist << 16, c1}
						const u32 ca0 = va.z & 0xffffu, cb0 = vb.z & 0xffffu;
						const float pb0 = (ca0 == cb0) ? __uint_as_float(vb.x) : ((ca0 == vb.w) ? __uint_as_float(vb.y) : 0.0f);
						const float pb1 = (va.w == cb0) ? __uint_as_float(vb.x) : ((va.w == vb.w) ? __uint_as_float(vb.y) : 0.0f);
						sum += __uint_as_float(va.x) * pb0; // relaxflat.cpp:27 (w == 1.0f): product rounded, then added
						sum += __uint_as_float(va.y) * pb1;
						const bool adv_a = va.w <= vb.w, adv_b = vb.w <= va.w;
						const u32 da = va.z >> 16, db = vb.z >> 16; // bytes to the next block of the row, 0: none
						if ((adv_a && da == 0u) || (adv_b && db == 0u)) break;
						ia += adv_a ? da : 0u;
						ib += adv_b ? db : 0u;
					}
					acc[q] = sum;
				}
			}
			if (p.nbuf == 2) {
				mpc_dma_wait(); // my part of step Z+1's records is in LDS
				vbase_cur = vbase_nxt;
			}
		}
		// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q) {
			MPC_SCHED_BARRIER();
			const u32 g0 = (u32)q * THREADS + wave * 64u;
			if (g0 < total) {
				u32 pi = 0;
				for (u32 j = 1; j < 16u; ++j) if (mpc_wave_first(ptab[8 * j]) <= g0 && mpc_wave_first(ptab[8 * j + 5]) != 0u) pi = j;
				const u32 base = mpc_wave_first(ptab[8 * pi]), nnz = mpc_wave_first(ptab[8 * pi + 1]);
				const u64 k = (u64)mpc_wave_first(ptab[8 * pi + 3]) | ((u64)mpc_wave_first(ptab[8 * pi + 4]) << 32);
				const u32 idx = g0 + lane - base;
				if (idx < nnz)
					s.vnext[s.vbase[k] + idx] = acc[q] / (float)n; // uint -> float, IEEE divide (mysparsemx.cpp:108)
			}
		}
	}
}
