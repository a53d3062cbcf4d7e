// kernels_relaxv.h — consistency relax over variable-size dense records: relax_var_kernel (the default).
//
// Replaces MPCFlat::ConsPair (conspairflat.cpp:10-110) -> RelaxFlat_{XZ_ZY,ZX_ZY,XZ_YZ} (relaxflat.cpp:4-94) ->
// MySparseMx::UpdateFromPost (mysparsemx.cpp:87-113) for a TILE of pairs at a time, with the arithmetic and the order of
// additions of relax_kernel (kernels_store.h): per stored cell (x,y) of (X,Y)
//     acc = 2*P_XY(x,y);  for Z = 0..N-1: acc += sum_z M(X,Z)(x,z) * M(Y,Z)(y,z)  (z ascending);  P' = acc / N
// product rounded, then added (no FMA): bit-identical to the reference.
//
// Shape of the kernel (round 3; what rounds 1-2 measured is in DESIGN.md 4.3):
//  * A workgroup owns the pairs {X in [x0,x0+nx)} x {Y in [y0,y0+ny)}, X < Y, and walks Z = 0..N-1 once. Per step it needs
//    the records M(S,Z) of its <= 8 sequences S. Records are stored Z-major (kernels_store.h), so these are TWO contiguous
//    runs of HBM — the X range, and the part of the Y range beyond it — staged HBM -> LDS by LDS-DMA
//    (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPRs, no ds_write) exactly as they lie in memory. Where each
//    record starts is read from 5 + 5 consecutive entries of the block-offset table by ONE vector load per step (lane i =
//    record i), one step ahead; the lane-indexed result IS the register the merges take their record bases from.
//  * The cells of the tile's pairs are laid end to end over (slot, lane), every pair rounded up to whole waves, so the 64
//    cells of a (wave, slot) belong to ONE pair: its two record bases come out of that register (one gather per step). A lane keeps,
//    per slot, the accumulator and the byte offsets of its cell's two rows (16*x | 16*y << 16) in VGPRs for the whole walk.
//  * Per (cell, Z): a block merge of row x of M(X,Z) and row y of M(Y,Z), 2 entries per block and step. The FIRST step of
//    every slot is straight-line code for all lanes, and the two LDS reads of slot q+1's first step are issued before slot
//    q's arithmetic, so a wave always has reads in flight. Only lanes whose rows go on (a third entry on the side that has
//    to advance) enter the loop for the further steps. Blocks with one entry repeat their column (kernels_store.h), so
//    advance / stop decisions compare real last columns: 2.45 steps per wave and (cell, Z) at 1000 x L~400, was 2.76.
#pragma once
#include "kernels_store.h"
#include <type_traits>

#define MPC_RV_MAXSEQ 8        // records resident per step (4 + 4 sequences)
#define MPC_RV_MAXLEN 4095u    // a cell keeps 16 * row and 16 * column in 16 bits each; a block's distance field is 16 bits of bytes
#define MPC_RV_TAB_BYTES 512   // pair table of the tile at the head of the dynamic LDS: 16 pairs x 8 dwords
#define MPC_RV_WAVE 64u
#define MPC_RV_YLANE 8u        // lanes 0..4 of the step table: X run (record i -> lane i); lanes 8..12: Y run

struct RelaxVarParams {
	StoreParams s;
	const u32 *tiles; // 4 u32 per tile: x0, nx, y0, ny
	u32 ntiles;
	u64 k0, k1;    // only pairs in [k0,k1) are relaxed (multi-GPU shard)
	u32 nbuf;      // LDS staging buffers: 2 = DMA of step Z+1 under the merges of step Z, 1 = DMA, wait, merge
	u32 buf_bytes; // capacity of one staging buffer (the second starts buf_bytes after the first)
	u32 *tile_next; // 8 counters, zeroed before the launch: next tile of each XCD's range
};

// One merge step on the blocks va (row x of M(X,Z)) and vb (row y of M(Y,Z)), block = {p0, p1, c0 | dist << 16, c1}:
// the 2x2 column compare, two products, two adds in z order (relaxflat.cpp:16-29 / :41-58 / :78-92: an unmatched entry, a
// repeated column and an empty block contribute pa * 0.0f = +0.0f, which leaves the strictly positive sum unchanged bit for
// bit). Of two equal columns in vb the first wins (the second is the zero-probability repeat).
#define MPC_RV_TERMS(sum, va, vb)                                                                                           \
	do {                                                                                                                \
		const u32 ca0_ = (va).z & 0xffffu, cb0_ = (vb).z & 0xffffu;                                                     \
		float pb0_ = (ca0_ == (vb).w) ? __uint_as_float((vb).y) : 0.0f;                                                 \
		pb0_ = (ca0_ == cb0_) ? __uint_as_float((vb).x) : pb0_;                                                          \
		float pb1_ = ((va).w == (vb).w) ? __uint_as_float((vb).y) : 0.0f;                                               \
		pb1_ = ((va).w == cb0_) ? __uint_as_float((vb).x) : pb1_;                                                        \
		sum += __uint_as_float((va).x) * pb0_; /* relaxflat.cpp:27 (w == 1.0f): product rounded, then added */           \
		sum += __uint_as_float((va).y) * pb1_;                                                                          \
	} while (0)

// The merge of one (cell, Z) in C++: the statement of what MpcRvBlocksAsm (mpc_platform.h) does with hand-scheduled
// instructions. The emulator runs this one; on the device it is the MPCGPU_RELAX_MERGE=cxx instantiation (A/B).
__device__ __forceinline__ void mpc_rv_merge_cxx(float &sum, MpcQuad va, MpcQuad vb, u32 ia, u32 ib)
{
	MPC_RV_TERMS(sum, va, vb);
	// The row whose last column is not larger moves to its next block (the distance in bytes rides in the upper half of
	// the block's first column word; 0: the row ends here); when that row has none the merge is over. Most lanes stop
	// after the first step.
	bool adv_a = va.w <= vb.w, adv_b = vb.w <= va.w;
	u32 da = va.z >> 16, db = vb.z >> 16;
	bool more = !((adv_a && da == 0u) || (adv_b && db == 0u));
	while (more) {
		// only the row that advanced is read again (as the hand-scheduled merge does under its advance masks): relax_band_kernel
		// passes first-block addresses that are already biased for the hop — valid addresses only after a distance has been added
		if (adv_a) { ia += da; va = mpc_lds_load16(ia); }
		if (adv_b) { ib += db; vb = mpc_lds_load16(ib); }
		MPC_RV_TERMS(sum, va, vb);
		adv_a = va.w <= vb.w; adv_b = vb.w <= va.w;
		da = va.z >> 16; db = vb.z >> 16;
		more = !((adv_a && da == 0u) || (adv_b && db == 0u));
	}
}
// one slot at a time, the next slot's first blocks in flight
struct MpcRvBlocksCxx {
	MpcQuad a[2], b[2];
	__device__ __forceinline__ void load(int set, u32 ia, u32 ib) { a[set] = mpc_lds_load16(ia); b[set] = mpc_lds_load16(ib); }
	__device__ __forceinline__ void drain() {}
	template <int SET> __device__ __forceinline__ void merge(float &sum, u32 ia, u32 ib, u32 nia, u32 nib)
	{
		const MpcQuad va = a[SET], vb = b[SET];
		load(SET ^ 1, nia, nib); // the next slot's first blocks: in flight during this slot's arithmetic
		mpc_rv_merge_cxx(sum, va, vb, ia, ib);
	}
};
#ifndef MPC_RV_HAVE_ASM
typedef MpcRvBlocksCxx MpcRvBlocksAsm; // the emulator has the C++ statement only
#endif

// THREADS: workgroup size; MAXSLOTS: cells per lane (the host splits any tile whose wave-aligned cells need more);
// DIAG (measurement only, results wrong): 1 = staging and barriers only, 2 = merges only (step 0's records for every step,
// no further staging, no barriers), 3 = as 2 with the two barriers per step.
// WGS: workgroups per CU the register allocation has to allow (waves per SIMD = WGS * THREADS / 256).
// BLOCKS: MpcRvBlocksAsm (hand-scheduled merge, the default) or MpcRvBlocksCxx.
template <int THREADS, int MAXSLOTS, int WGS, int DIAG = 0, class BLOCKS = MpcRvBlocksAsm>
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
		const u32 x0 = mpc_wave_first(p.tiles[4 * tl]), nx = mpc_wave_first(p.tiles[4 * tl + 1]);
		const u32 y0 = mpc_wave_first(p.tiles[4 * tl + 2]), ny = mpc_wave_first(p.tiles[4 * tl + 3]);
		u32 ys, nys; // the run of Y records beyond the X range (kernels_store.h: mpc_tile_runs)
		mpc_tile_runs(x0, nx, y0, ny, &ys, &nys);
		// ---- pair table: pair (ix, iy) of the tile -> first cell, stored cells, the step-table lanes of its two records.
		// Lane q of wave 0 looks after pair q; a 16-lane inclusive scan lays the pairs' cell ranges end to end, each rounded
		// up to whole waves so that the 64 cells of any (wave, slot) belong to ONE pair.
		if (tid < 64u) {
			const u32 ix = lane / 4u, iy = lane % 4u;
			const u32 X = x0 + ix, Y = y0 + iy;
			u32 nnz = 0, sel = 0;
			u64 k = 0;
			if (lane < 16u && ix < nx && iy < ny && X < Y) {
				k = mpc_pair_pos(s, X, Y);
				if (k >= p.k0 && k < p.k1) nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
				const u32 mb = (Y < x0 + nx) ? Y - x0 : MPC_RV_YLANE + (Y - ys); // Y >= x0 here (X < Y), so Y is in one of the runs
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
		// slots of this wave that hold cells: slot q covers cells [q*THREADS + wave*64, +64)
		const u32 wave_first = wave * 64u;
		const u32 nact = total > wave_first ? (total - wave_first + THREADS - 1u) / THREADS : 0u;

		// ---- my cells: one pair per (wave, slot) (found by a scalar scan of the table), or nothing. Accumulator and row
		// offsets stay in VGPRs for the whole walk; the pair's two step-table lanes go into lane q of `vsel`.
		float acc[MAXSLOTS];
		u32 xy[MAXSLOTS]; // 16 * x | (16 * y) << 16: byte offsets of row x of M(X,.) and row y of M(Y,.) inside their records
		u32 vsel_a = 0, vsel_b = 0; // lane q: 4 * (step-table lane of slot q's X record / Y record): index registers of the gather below
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q) {
			MPC_SCHED_BARRIER();
			acc[q] = 1.0f; xy[q] = 0u;
			const u32 g0 = (u32)q * THREADS + wave_first;
			if ((u32)q < nact) {
				u32 pi = 0;
				for (u32 j = 1; j < 16u; ++j) if (mpc_wave_first(ptab[8 * j]) <= g0 && mpc_wave_first(ptab[8 * j + 5]) != 0u) pi = j;
				const u32 base = mpc_wave_first(ptab[8 * pi]), nnz = mpc_wave_first(ptab[8 * pi + 1]), sel = mpc_wave_first(ptab[8 * pi + 2]);
				const u64 k = (u64)mpc_wave_first(ptab[8 * pi + 3]) | ((u64)mpc_wave_first(ptab[8 * pi + 4]) << 32);
				vsel_a = mpc_write_lane(vsel_a, 4u * (sel & 0xffu), (u32)q);
				vsel_b = mpc_write_lane(vsel_b, 4u * (sel >> 8), (u32)q);
				const u32 idx = g0 + lane - base;
				if (idx < nnz) {
					const u32 *ent = s.packed + s.pbase[k] + s.seq_len[s.pair_x[k]] + s.seq_len[s.pair_y[k]];
					acc[q] = __uint_as_float(ent[2 * (u64)idx]) * 2.0f; // conspairflat.cpp:29-30
					xy[q] = (ent[2 * (u64)nnz + idx] << 4) | (ent[2 * (u64)idx + 1] << 20);
				}
			}
		}

		// ---- walk Z. The step table of step Z, one entry per lane: lane i (i <= nx) = block offset of record (x0+i, Z) — lane
		// nx: the end of the X run; lane 8+j (j <= nys) the same for the Y run. Loaded one step ahead with one vector load.
		const u32 tab_a = lane < MPC_RV_YLANE ? x0 + (lane < nx ? lane : nx) : ys + (lane - MPC_RV_YLANE < nys ? lane - MPC_RV_YLANE : nys);
		const u32 *tab_src = s.rec_off + tab_a; // + Z*n per step (rec_off has n*n+1 entries: the run ends are in range)
		auto load_table = [&](u32 Z) -> u32 { return tab_src[(u64)Z * n]; };
		// Issues the DMA of the step whose table is `tab` into staging buffer `buf`: the X run, then the Y run, as they lie in
		// HBM, in chunks of 64 blocks = 1 KiB dealt round-robin to the waves. Returns the register that holds in lane i the LDS
		// byte address of record i (X run) / in lane 8+j that of record j of the Y run.
		auto issue_dma = [&](u32 tab, u32 buf) -> u32 {
			const u32 xs = mpc_read_lane(tab, 0u), xe = mpc_read_lane(tab, nx);
			const u32 yb = mpc_read_lane(tab, MPC_RV_YLANE), ye = mpc_read_lane(tab, MPC_RV_YLANE + nys);
			const u32 at = MPC_RV_TAB_BYTES + buf * p.buf_bytes; // byte offset of the buffer in the dynamic LDS
			const u32 xlen = xe - xs;
			{
				for (u32 c0 = wave * 64u; c0 < xlen; c0 += NWAVES * 64u) {
					const u32 blk = c0 + lane;
					if (blk < xlen) mpc_dma16(padb + 16 * ((u64)xs + blk), smem_raw + at + 16 * c0);
				}
				const u32 ylen = ye - yb;
				// the waves that got the fewest chunks of the X run take the first of the Y run
				const u32 shift = (xlen + 63u) / 64u % NWAVES;
				const u32 w2 = wave >= shift ? wave - shift : wave + NWAVES - shift;
				for (u32 c0 = w2 * 64u; c0 < ylen; c0 += NWAVES * 64u) {
					const u32 blk = c0 + lane;
					if (blk < ylen) mpc_dma16(padb + 16 * ((u64)yb + blk), smem_raw + at + 16 * (xlen + c0));
				}
			}
			const u32 rel = lane < MPC_RV_YLANE ? tab - xs : tab - yb + xlen;
			return lds0 + at + 16u * rel;
		};

		u32 tab_nxt = 0;
		u32 vbase_cur = 0, vbase_nxt = 0; // lane i: LDS byte address of record i of the step being merged / being staged
		if (p.nbuf == 2) {
			const u32 tab0 = load_table(0);
			vbase_cur = issue_dma(tab0, 0);
			if (n > 1) tab_nxt = load_table(1);
			mpc_dma_wait();
		} else tab_nxt = load_table(0);
		for (u32 Z = 0; Z < n; ++Z) {
			if (DIAG == 2 && Z > 0) {
				// measurement only: every step merges step 0's records
			} else if (DIAG == 3 && Z > 0) {
				// measurement only: step 0's records, but the two barriers of a step are kept (what the waves of a workgroup lose by
				// waiting for each other, without the staging)
				__syncthreads();
				__syncthreads();
			} else if (p.nbuf == 2) {
				__syncthreads(); // step Z's records have landed (every wave waited for its own DMA) and step Z-1's readers are done
				if (DIAG != 2 && Z + 1 < n) {
					vbase_nxt = issue_dma(tab_nxt, (Z + 1) & 1u);
					if (Z + 2 < n) tab_nxt = load_table(Z + 2);
				}
			} else {
				__syncthreads(); // step Z-1's readers are done
				vbase_cur = issue_dma(tab_nxt, 0);
				if (Z + 1 < n) tab_nxt = load_table(Z + 1);
				mpc_dma_wait();
				__syncthreads();
			}
			if (DIAG != 1 && nact != 0u) {
				// lane q of base_a / base_b: LDS address of slot q's two records at this step (one cross-lane gather each per step;
				// a slot then takes its bases with two v_readlane of a constant lane)
				const u32 base_a = mpc_lane_gather(vbase_cur, vsel_a), base_b = mpc_lane_gather(vbase_cur, vsel_b);
				BLOCKS blk;
				auto addr_a = [&](int q) -> u32 { return mpc_read_lane(base_a, (u32)q) + (xy[q] & 0xffffu); }; // one register per slot:
				auto addr_b = [&](int q) -> u32 { return mpc_read_lane(base_b, (u32)q) + (xy[q] >> 16); };     // the row offsets are unpacked per step
				// first blocks of slot 0; from then on slot q+1's are read while slot q is merged
				u32 nia = addr_a(0), nib = addr_b(0);
				blk.load(0, nia, nib);
				// slots 0 .. nact-1, unrolled by recursion over the slot number (a loop with an early exit is not unrolled, and
				// acc[] / xy[] must stay registers)
				auto slot = [&](auto &&self, auto qc) __attribute__((always_inline)) {
					constexpr int q = decltype(qc)::value;
					if constexpr (q < MAXSLOTS) {
						if ((u32)q >= nact) return; // wave-uniform: the slots a wave holds cells of are the first nact
						const u32 ia = nia, ib = nib;
						// slot q+1's first blocks, unconditionally (a slot without cells has base lane 0 and offsets 0: row 0 of record 0)
						constexpr int qn = q + 1 < MAXSLOTS ? q + 1 : q;
						nia = addr_a(qn); nib = addr_b(qn);
						float sum = acc[q];
						blk.template merge<q & 1>(sum, ia, ib, nia, nib);
						acc[q] = sum;
						self(self, std::integral_constant<int, q + 1>{});
					}
				};
				slot(slot, std::integral_constant<int, 0>{});
				blk.drain(); // the last slot's look-ahead reads land before v24..v39 can mean anything else
			}
			if (p.nbuf == 2 && DIAG != 2) {
				mpc_dma_wait(); // my part of step Z+1's records is in LDS
				if (Z + 1 < n) vbase_cur = vbase_nxt;
			}
		}
		// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q) {
			MPC_SCHED_BARRIER();
			const u32 g0 = (u32)q * THREADS + wave_first;
			if ((u32)q < nact) {
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
