// kernels_store.h — device-resident all-pairs sparse posterior store, relax, commit, export.
//
// Layout in HBM. The reference keeps one MySparseMx per unordered pair and reaches
// P_YZ(y,z) for Z>Y through a column->row range search (relaxflat.cpp:62-94, mysparsemx.cpp:238).
// Here every ORDERED pair (A,Z) gets a row-major CSR matrix M(A,Z) (rows = positions of A, cols =
// positions of Z; the transposed copies carry bit-identical floats), grouped into one "slab" per
// sequence A with Z ascending and an empty matrix at Z == A:
//     rp  [rp_base[A] + Z*(L_A+1) + a]   u32  entry offset of row a of M(A,Z), relative to slab A
//     ent [ent_base[A] + offset]         {P bits, col} 8 bytes
// With that, ConsPair (conspairflat.cpp:10-110) for one stored cell (x,y) of pair (X,Y) is
//     acc = 2*P_XY(x,y);  for Z = 0..N-1:  acc += sum_z  M(X,Z)(x,z) * M(Y,Z)(y,z)   (z ascending)
//     P'  = acc / N
// i.e. a merge of two short sorted rows per Z; rows x..x+k of M(X,Z) are contiguous, so a wave of
// consecutive cells reads a few contiguous cache lines per Z. The three RelaxFlat_* variants
// (relaxflat.cpp:4-94) collapse into this one form because both factors are always read by row; the
// accumulation order per cell (Z ascending, then z ascending; product rounded, then added — no FMA)
// is the reference's, so results are bit-identical. Z == X and Z == Y contribute nothing because
// M(X,X) and M(Y,Y) are empty, matching the `continue` at conspairflat.cpp:39-40.
#pragma once
#include "device_math.h"

struct __attribute__((aligned(8))) MpcEnt { u32 p; u32 c; }; // {float bits, column}: MySparseMx entry layout (mysparsemx.h:56-82)

struct StoreParams {
	u32 n;
	const u32 *seq_len;
	u64 npairs;
	const u32 *pair_x, *pair_y; // all pairs (InitPairs order)
	// packed records of all pairs (kernels_post.h layout) and their bases
	u32 *packed;
	const u64 *pbase; // npairs+1, word offsets into packed
	const u64 *vbase; // npairs+1, canonical entry index of the pair's first entry
	// slabs
	u32 *rp;
	const u64 *rp_base; // n+1
	MpcEnt *ent;
	const u64 *ent_base; // n+1
	const u32 *mbase;    // n*(n+1): entry offset of M(A,Z) inside slab A at [A*(n+1)+Z]; [..+n] = slab size
	// values of the next iteration, canonical order
	float *vnext;
	// Records for the LDS-tiled relax (relax_var_kernel, kernels_relaxv.h), used INSTEAD of the slabs when a run fits their
	// limits: one record per ORDERED pair (A,Z), a 16-byte-aligned verbatim copy of what the kernel wants in LDS:
	//   [first block of every row: len(A) blocks, row a at block a][its overflow blocks]
	// block = 16 bytes = {P0 bits, P1 bits, col0 | dist << 16, col1}: 2 entries (the two probabilities adjacent), dist = bytes to the
	// next block of the same row (0 in its last). A block that holds ONE entry (the last block of a row with an odd count)
	// repeats its column with a zero probability, {P0, 0.0f, col0 | 0, col0}: its last column is then the row's real last
	// column, which is what the merge's advance / stop decisions compare (a sentinel there kept the merge running one more
	// step whenever the other row went on: 2.76 -> 2.45 steps per wave at 1000 x L~400, oracle statistics), and the repeated
	// column contributes P * 0.0f = +0.0f. The block of an EMPTY row is {0.0f, 0.0f, MPC_PAD_SENTINEL, MPC_PAD_SENTINEL}. A row
	// is reached by its index alone (no row pointers); rows of up to 2 entries — most of them — never leave the first-block
	// region. Records are stored Z-MAJOR: record (A,Z) starts at block rec_off[Z*n+A] of `pad` (rec_off: n*n+1 entries, 16-byte
	// units; mpc_rec_index), so the records a relax tile needs at step Z — those of 4 consecutive sequences A — are ONE
	// contiguous run in HBM, staged by one stream of LDS-DMA chunks and described by 5 consecutive table entries. A record is
	// exactly as long as it needs to be: 8.1 KB on average at 1000 x L~400 (padding every record to the worst one cost 13.3 KB).
	// pos_f / pos_t give, per canonical entry, its entry index (block * 2 + slot) inside the record of (X,Y) and of (Y,X):
	// written by var_build, used by commit.
	u32 *pad;
	u32 lcap1; // >= the longest sequence: LDS scratch of var_build_kernel
	unsigned short *pos_f, *pos_t;
	const u32 *rec_off;
	// band index of the records' overflow regions (relax_band_kernel, kernels_relaxb.h): ovf_off[(Z*n + A) * nb1 + b] = block index
	// in `pad` where the overflow blocks of rows >= HB*b of record (A,Z) start (overflow blocks are stored in row order); the
	// record's end when HB*b >= len(A). HB = MPC_RB_HB rows, nb1 = ceil(longest sequence / HB) + 1. Written by var_build_kernel;
	// null: not wanted.
	u32 *ovf_off;
	u32 nb1;
	// WINDOW records (kernels_relaxb.h, the direct-index merge; null: not built): a second copy of every ordered pair's matrix for
	// the role of the Y operand, in which a row is looked up BY COLUMN instead of being walked:
	//   record (A,Z) = [desc: len(A)+1 words, padded to 16 bytes][values, + one spare block], at block wrec_off[Z*n+A] of `win`
	//   desc[a] = c0 | span << 12 | off << 17: first stored column of row a (0: none; 12 bits), span = last - first + 1 columns (0 for
	//   an empty row; 5 bits) and the dword offset of its values in the value area (15 bits) — ONE word per row, one LDS read per
	//   (cell, Z) (until round 4's last profile: c0 | off << 12, the span taken from the next row's word: a two-word read whose
	//   bank conflicts the row-block cell order doubled). A span field of 31 is an ESCAPE: the span is then off(a + 1) - off - 1,
	//   one more read for the lanes that meet such a row (a fragment against a full-length sequence: spans of 60). Stores with a
	//   value area > 32767 dwords keep block records for the Y operand (win_size_kernel's flag). desc[len(A)] = total << 17. Row a
	//   owns dwords off .. off + span:
	//   the probability of column c0 + j at off + j (0.0f where that column is not stored), and 0.0f at off + span — the GUARD that
	//   every column outside the window is clamped to: value(z) = val[off + min(z - c0 (unsigned), span)].
	// wv_off[(Z*n+A)*nb1 + b] = block of `win` that holds the first value of row MPC_RB_HB*b (the record's end beyond the last row).
	// pos_wf / pos_wt: per canonical entry, the dword of its probability inside the value area of the record of (X,Y) / (Y,X).
	u32 *win;
	const u32 *wrec_off;
	u32 *wv_off;
	unsigned short *pos_wf, *pos_wt;
	// PAIR ORDER (mpcgpu_set_pair_order; multi-GPU block partition): the position k of pair (X,Y) in every per-pair array
	// above. nrect == 0: MPCFlat::InitPairs order (mpcflat.cpp:145-155, closed form). Otherwise the pairs are enumerated rectangle by
	// rectangle, row-major inside each: rects[6 r ..] = {xa, xb, ya, yb, base lo, base hi}; a rectangle is either off the diagonal
	// (ya >= xb: all (x,y) of [xa,xb) x [ya,yb)) or a triangle (xa == ya, xb == yb: the pairs x < y inside [xa,xb)). A rank of a
	// block-partitioned run owns consecutive rectangles, i.e. ONE contiguous range of positions.
	u32 nrect;
	const u32 *rects;
	// PARTIAL STORE (mpcgpu_store_import_part): need[A] != 0 where the records (A, Z) of sequence A exist — a rank whose pairs touch
	// only the sequences of its blocks builds, reads and commits the records of those sequences only. null: every sequence.
	const u8 *need;
};

#define MPC_PAD_ROW 2 // entries per block (16 bytes = one ds_read_b128)
#define MPC_WIN_MAXSPAN 31u   // window records: a row's span field (5 bits); 31 = look at the next row's offset
#define MPC_WIN_MAXOFF 32767u  // and its value offset (15 bits)
#define MPC_RB_HB 8u  // rows per index band of the band tables (overflow offsets here; cell offsets, y ranges: kernels_relaxb.h)
#define MPC_PAD_SENTINEL 0x1fffu // larger than any column: sequences in the padded layout are <= 8191 long

__device__ __forceinline__ u64 mpc_pair_index(u32 n, u32 i, u32 j) // i<j, mpcflat.cpp:145-155 order
{
	return (u64)i * n - ((u64)i * (i + 1)) / 2 + (j - i - 1);
}

// position of pair (X,Y), X < Y, in the store's pair order (StoreParams::rects)
__device__ __forceinline__ u64 mpc_pair_pos_f(u32 n, u32 nrect, const u32 *rects, u32 X, u32 Y)
{
	if (nrect == 0u) return mpc_pair_index(n, X, Y);
	for (u32 r = 0; r < nrect; ++r) {
		const u32 *q = rects + 6u * r;
		const u32 xa = q[0], xb = q[1], ya = q[2], yb = q[3];
		if (X < xa || X >= xb || Y < ya || Y >= yb) continue;
		const u64 base = (u64)q[4] | ((u64)q[5] << 32);
		return ya >= xb ? base + (u64)(X - xa) * (yb - ya) + (Y - ya) : base + mpc_pair_index(xb - xa, X - xa, Y - xa);
	}
	return ~0ull; // (mpcgpu_set_pair_order checks that the rectangles cover every pair)
}
__device__ __forceinline__ u64 mpc_pair_pos(const StoreParams &s, u32 X, u32 Y) { return mpc_pair_pos_f(s.n, s.nrect, s.rects, X, Y); }
__device__ __forceinline__ bool mpc_need(const StoreParams &s, u32 A) { return s.need == nullptr || s.need[A] != 0; }

// index of record (A,Z) in rec_off: Z-major (see StoreParams::pad)
__device__ __forceinline__ u64 mpc_rec_index(u32 n, u32 A, u32 Z) { return (u64)Z * n + A; }

// One 64-thread workgroup per ordered pair (A,Z): row pointers (exclusive scan of the per-row or
// per-column counts) and entries (row-major copy, or column-major through tperm).
__global__ void __launch_bounds__(64) slab_build_kernel(StoreParams s)
{
	const int t = threadIdx.x;
	const u64 total = (u64)s.n * s.n;
	for (u64 b = blockIdx.x; b < total; b += gridDim.x) {
		const u32 A = (u32)(b / s.n), Z = (u32)(b % s.n);
		const u32 LA = s.seq_len[A];
		u32 *rp = s.rp + s.rp_base[A] + (u64)Z * (LA + 1);
		const u32 mb = s.mbase[(u64)A * (s.n + 1) + Z];
		if (A == Z) {
			for (u32 a = t; a <= LA; a += 64) rp[a] = mb;
			continue;
		}
		const bool fwd = A < Z;
		const u64 k = fwd ? mpc_pair_pos(s, A, Z) : mpc_pair_pos(s, Z, A);
		const u32 *rec = s.packed + s.pbase[k];
		const u32 LX = fwd ? LA : s.seq_len[Z]; // rows of the stored (unordered) pair
		const u32 LY = fwd ? s.seq_len[Z] : LA;
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		const u32 *cnt = fwd ? rec : rec + LX; // rowcnt or colcnt, LA entries
		u32 carry = mb;
		for (u32 a0 = 0; a0 <= LA; a0 += 64) {
			const u32 a = a0 + t;
			const u32 v = (a < LA) ? cnt[a] : 0;
			u32 incl = v;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (t >= d) incl += o;
			}
			if (a <= LA) rp[a] = carry + incl - v;
			carry += __shfl(incl, 63);
		}
		const u32 *e = rec + LX + LY;
		const u32 *rowv = e + 2 * (u64)nnz;
		const u32 *tperm = rowv + nnz;
		MpcEnt *dst = s.ent + s.ent_base[A] + mb;
		for (u32 q = t; q < nnz; q += 64) {
			MpcEnt v;
			v.p = e[2 * (u64)q];
			if (fwd) { v.c = e[2 * (u64)q + 1]; dst[q] = v; }
			else { v.c = rowv[q]; dst[tperm[q]] = v; }
		}
	}
}

// ---- variable-size dense records ---------------------------------------------------------------------------------
// blocks of record (A,Z): len(A) + sum_a max(ceil(cnt[a]/2) - 1, 0); one 64-thread workgroup per record
__global__ void __launch_bounds__(64) var_size_kernel(StoreParams s, u32 *sizes)
{
	const int t = threadIdx.x;
	const u64 total = (u64)s.n * s.n;
	for (u64 b = blockIdx.x; b < total; b += gridDim.x) {
		const u32 Z = (u32)(b / s.n), A = (u32)(b % s.n); // b == mpc_rec_index(n, A, Z)
		const u32 LA = s.seq_len[A];
		u32 mine = 0;
		if (!mpc_need(s, A)) { if (t == 0) sizes[b] = 0u; continue; } // a partial store: no record for this sequence
		if (A != Z) {
			const bool fwd = A < Z;
			const u64 k = fwd ? mpc_pair_pos(s, A, Z) : mpc_pair_pos(s, Z, A);
			const u32 *rec = s.packed + s.pbase[k];
			const u32 *cnt = fwd ? rec : rec + s.seq_len[Z];
			for (u32 a = t; a < LA; a += 64) {
				const u32 nb = (cnt[a] + MPC_PAD_ROW - 1) / MPC_PAD_ROW;
				mine += nb > 1 ? nb - 1 : 0;
			}
			for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d);
		}
		if (t == 0) sizes[b] = LA + mine;
	}
}

// One 64-thread workgroup per ordered pair (A,Z). Dynamic LDS: 2*lcap1 u32 (lcap1 >= the longest sequence).
__global__ void __launch_bounds__(64) var_build_kernel(StoreParams s)
{
	MPC_DYN_SMEM(smem_raw);
	u32 *s_start = (u32 *)smem_raw; // entries before row a in the packed (unpadded) order
	u32 *s_ovf = s_start + s.lcap1; // overflow blocks before row a
	const int t = threadIdx.x;
	const u64 total = (u64)s.n * s.n;
	for (u64 b = blockIdx.x; b < total; b += gridDim.x) {
		const u32 Z = (u32)(b / s.n), A = (u32)(b % s.n); // b == mpc_rec_index(n, A, Z)
		const u32 LA = s.seq_len[A];
		if (!mpc_need(s, A)) continue; // (wave-uniform) a partial store: this sequence has no records, no band entries, no positions
		u32 *rec_out = s.pad + 4 * (u64)s.rec_off[b];
		const u32 units = s.rec_off[b + 1] - s.rec_off[b];
		for (u32 q = t; q < units; q += 64) { // every block starts as an empty one
			rec_out[4 * q] = 0u; rec_out[4 * q + 1] = 0u; rec_out[4 * q + 2] = MPC_PAD_SENTINEL; rec_out[4 * q + 3] = MPC_PAD_SENTINEL;
		}
		if (A == Z) { // empty matrix: conspairflat.cpp:39-40 skips Z == X and Z == Y
			if (s.ovf_off) for (u32 q = t; q < s.nb1; q += 64) s.ovf_off[b * s.nb1 + q] = s.rec_off[b + 1];
			continue;
		}
		const bool fwd = A < Z;
		const u64 k = fwd ? mpc_pair_pos(s, A, Z) : mpc_pair_pos(s, Z, A);
		const u32 *rec = s.packed + s.pbase[k];
		const u32 LX = fwd ? LA : s.seq_len[Z]; // rows of the stored (unordered) pair
		const u32 LY = fwd ? s.seq_len[Z] : LA;
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		const u32 *cnt = fwd ? rec : rec + LX; // rowcnt or colcnt, LA entries
		u32 carry = 0, carry_o = 0;
		for (u32 a0 = 0; a0 < LA; a0 += 64) {
			const u32 a = a0 + t;
			const u32 v = (a < LA) ? cnt[a] : 0;
			const u32 nb = (v + MPC_PAD_ROW - 1) / MPC_PAD_ROW;
			const u32 vo = nb > 1 ? nb - 1 : 0;
			u32 incl = v, incl_o = vo;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d), oo = __shfl_up(incl_o, d);
				if (t >= d) { incl += o; incl_o += oo; }
			}
			if (a < LA) {
				s_start[a] = carry + incl - v;
				s_ovf[a] = carry_o + incl_o - vo;
			}
			carry += __shfl(incl, 63);
			carry_o += __shfl(incl_o, 63);
		}
		__syncthreads(); // empty blocks and scans are in place before the entries go in
		if (s.ovf_off)
			for (u32 q = t; q < s.nb1; q += 64)
				s.ovf_off[b * s.nb1 + q] = MPC_RB_HB * q < LA ? s.rec_off[b] + LA + s_ovf[MPC_RB_HB * q] : s.rec_off[b + 1];
		const u32 *e = rec + LX + LY;
		const u32 *rowv = e + 2 * (u64)nnz;
		const u32 *tperm = rowv + nnz;
		unsigned short *pos = (fwd ? s.pos_f : s.pos_t) + s.vbase[k];
		for (u32 q = t; q < nnz; q += 64) {
			const u32 pbits = e[2 * (u64)q];
			const u32 col = e[2 * (u64)q + 1], row = rowv[q];
			const u32 r = fwd ? row : col;       // row of this entry in M(A,Z)
			const u32 rank = fwd ? q : tperm[q]; // its rank in M(A,Z)'s row-major order
			const u32 c = fwd ? col : row;
			const u32 within = rank - s_start[r];
			const u32 j = within / MPC_PAD_ROW, slot = within % MPC_PAD_ROW;
			const u32 nb = (cnt[r] + MPC_PAD_ROW - 1) / MPC_PAD_ROW;
			const u32 ovf0 = LA + s_ovf[r]; // first overflow block of this row
			const u32 unit = j == 0 ? r : ovf0 + (j - 1);
			const u32 delta = j + 1 < nb ? (j == 0 ? ovf0 - r : 1u) : 0u; // blocks to the row's next block
			rec_out[4 * unit + slot] = pbits;
			rec_out[4 * unit + 2 + slot] = slot == 0 ? (c | (delta << 20)) : c; // << 16 and * 16: bytes
			if (slot == 0 && within + 1 == cnt[r]) rec_out[4 * unit + 3] = c; // one-entry block: its column once more (P1 stays 0.0f)
			pos[q] = (unsigned short)(unit * MPC_PAD_ROW + slot);
		}
		__syncthreads(); // s_start is reused by the next record
	}
}

// ---- window records (see StoreParams::win) -----------------------------------------------------------------------------------
// value dwords of window record (A,Z): sum over rows of (span + 1); sizes[b] = blocks of the record = desc blocks + value blocks
// + 1 spare; *too_wide is set when a record does not fit the descriptor word (a value area > MPC_WIN_MAXOFF dwords). One wave per record.
__global__ void __launch_bounds__(64) win_size_kernel(StoreParams s, u32 *sizes, u32 *vals_total, u32 *too_wide)
{
	const int t = threadIdx.x;
	const u64 total = (u64)s.n * s.n;
	for (u64 b = blockIdx.x; b < total; b += gridDim.x) {
		const u32 Z = (u32)(b / s.n), A = (u32)(b % s.n);
		const u32 LA = s.seq_len[A];
		u32 mine = 0;
		if (!mpc_need(s, A)) { if (t == 0) { sizes[b] = 0u; vals_total[b] = 0u; } continue; } // a partial store
		if (A != Z) {
			// the span of row a of M(A,Z) = last - first + 1 of the stored columns: from the record of (A,Z) in block form (first block
			// of the row: its first column; the row's last block: its last column) — already built
			const u32 *rec = s.pad + 4 * (u64)s.rec_off[b];
			for (u32 a = t; a < LA; a += 64) {
				u32 blk = a;
				const u32 c_first = rec[4 * blk + 2] & 0xffffu;
				u32 span = 0;
				if (c_first != MPC_PAD_SENTINEL) {
					for (u32 d = rec[4 * blk + 2] >> 20; d != 0u; d = rec[4 * blk + 2] >> 20) blk += d; // distance in blocks (bytes / 16)
					span = rec[4 * blk + 3] - c_first + 1u;
				}
				mine += span + 1u;
			}
			for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d);
		} else mine = LA; // empty matrix: every row is its guard alone
		if (t == 0) {
			sizes[b] = (LA + 1u + 3u) / 4u + (mine + 3u) / 4u + 1u;
			vals_total[b] = mine;
			if (mine > MPC_WIN_MAXOFF) atomicOr(too_wide, 1u);
		}
	}
}

// Builds window record (A,Z) from the block-form record of the same ordered pair, and wv_off. One wave per record; dynamic LDS:
// lcap1 + 1 words (the rows' value offsets).
__global__ void __launch_bounds__(64) win_build_kernel(StoreParams s)
{
	MPC_DYN_SMEM(smem_raw);
	u32 *s_off = (u32 *)smem_raw;
	const u32 t = threadIdx.x;
	const u64 total = (u64)s.n * s.n;
	for (u64 b = blockIdx.x; b < total; b += gridDim.x) {
		const u32 A = (u32)(b % s.n);
		const u32 LA = s.seq_len[A];
		if (!mpc_need(s, A)) continue; // (wave-uniform) a partial store
		const u32 *rec = s.pad + 4 * (u64)s.rec_off[b];
		u32 *wrec = s.win + 4 * (u64)s.wrec_off[b];
		const u32 dblocks = (LA + 1u + 3u) / 4u;
		u32 *desc = wrec, *vals = wrec + 4 * (u64)dblocks;
		// spans and first columns per row, exclusive scan of span + 1
		u32 carry = 0;
		for (u32 a0 = 0; a0 <= LA; a0 += 64) {
			const u32 a = a0 + t;
			u32 c_first = 0, span = 0;
			if (a < LA) {
				u32 blk = a;
				const u32 cf = rec[4 * blk + 2] & 0xffffu;
				if (cf != MPC_PAD_SENTINEL) {
					for (u32 d = rec[4 * blk + 2] >> 20; d != 0u; d = rec[4 * blk + 2] >> 20) blk += d;
					c_first = cf; span = rec[4 * blk + 3] - cf + 1u;
				}
			}
			const u32 v = a < LA ? span + 1u : 0u;
			u32 incl = v;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (t >= (u32)d) incl += o;
			}
			const u32 off = carry + incl - v;
			if (a <= LA) { desc[a] = c_first | ((span < MPC_WIN_MAXSPAN ? span : MPC_WIN_MAXSPAN) << 12) | (off << 17); s_off[a] = off; }
			carry += __shfl(incl, 63);
		}
		for (u32 q = LA + 1u + t; q < 4u * dblocks; q += 64) desc[q] = carry << 17; // padding words repeat the end marker
		__syncthreads();
		const u32 vtotal = carry, vblocks = (vtotal + 3u) / 4u + 1u;
		for (u32 q = t; q < 4u * vblocks; q += 64) vals[q] = 0u; // 0.0f everywhere: guards and the gaps inside the windows
		if (s.wv_off)
			for (u32 q = t; q < s.nb1; q += 64)
				s.wv_off[b * s.nb1 + q] = MPC_RB_HB * q < LA ? s.wrec_off[b] + dblocks + s_off[MPC_RB_HB * q] / 4u : s.wrec_off[b + 1] - 1u;
		__syncthreads();
		// the probabilities: every entry of the block-form record goes to off(row) + (col - first col of the row)
		for (u32 a = t; a < LA; a += 64) {
			u32 blk = a;
			const u32 cf = rec[4 * blk + 2] & 0xffffu;
			if (cf == MPC_PAD_SENTINEL) continue;
			const u32 off = s_off[a];
			for (;;) {
				const u32 c0 = rec[4 * blk + 2] & 0xffffu, c1 = rec[4 * blk + 3];
				vals[off + (c0 - cf)] = rec[4 * blk];
				if (c1 != c0) vals[off + (c1 - cf)] = rec[4 * blk + 1]; // (a one-entry block repeats its column with 0.0f)
				const u32 d = rec[4 * blk + 2] >> 20;
				if (d == 0u) break;
				blk += d;
			}
		}
		__syncthreads(); // s_off is reused by the next record
	}
}

// pos_wf / pos_wt: the dword (inside the value area of its window record) of every canonical entry. One wave per pair.
__global__ void __launch_bounds__(64) win_pos_kernel(StoreParams s)
{
	const u32 t = threadIdx.x;
	for (u64 k = blockIdx.x; k < s.npairs; k += gridDim.x) {
		const u32 X = s.pair_x[k], Y = s.pair_y[k];
		const u32 LX = s.seq_len[X], LY = s.seq_len[Y];
		const u32 *prec = s.packed + s.pbase[k];
		const u32 *ent = prec + LX + LY;
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		const u32 *dxy = s.win + 4 * (u64)s.wrec_off[mpc_rec_index(s.n, X, Y)];
		const u32 *dyx = s.win + 4 * (u64)s.wrec_off[mpc_rec_index(s.n, Y, X)];
		const bool nx = mpc_need(s, X), ny = mpc_need(s, Y); // (a partial store has the records of the needed sequences only)
		if (!nx && !ny) continue;
		for (u32 q = t; q < nnz; q += 64) {
			const u32 col = ent[2 * (u64)q + 1], row = ent[2 * (u64)nnz + q];
			if (nx) { const u32 wf = dxy[row]; s.pos_wf[s.vbase[k] + q] = (unsigned short)((wf >> 17) + (col - (wf & 0xfffu))); }
			if (ny) { const u32 wt = dyx[col]; s.pos_wt[s.vbase[k] + q] = (unsigned short)((wt >> 17) + (row - (wt & 0xfffu))); }
		}
	}
}

// Largest LDS footprint of a tile's records over one walk: out[t] = max over Z of the blocks of the two runs relax_var_kernel
// stages per step — records (x0..x0+nx-1, Z) and, of the Y range, the records beyond the X range (y >= x0+nx; a Y inside the X
// range is already resident, a Y below it has no pair X < Y in the tile). One wave per tile, lanes stride over Z.
__device__ __forceinline__ void mpc_tile_runs(u32 x0, u32 nx, u32 y0, u32 ny, u32 *ys, u32 *nys)
{
	const u32 lo = y0 > x0 + nx ? y0 : x0 + nx, hi = y0 + ny;
	*ys = lo;
	*nys = hi > lo ? hi - lo : 0u;
}

__global__ void __launch_bounds__(64) var_tile_fit_kernel(StoreParams s, const u32 *tiles, u32 ntiles, u32 *out)
{
	const u32 t = threadIdx.x, n = s.n;
	for (u32 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
		const u32 x0 = tiles[4 * tl], nx = tiles[4 * tl + 1], y0 = tiles[4 * tl + 2], ny = tiles[4 * tl + 3];
		u32 ys, nys;
		mpc_tile_runs(x0, nx, y0, ny, &ys, &nys);
		u32 best = 0;
		for (u32 Z = t; Z < n; Z += 64) {
			const u64 bx = mpc_rec_index(n, x0, Z);
			u32 sum = s.rec_off[bx + nx] - s.rec_off[bx];
			if (nys) { const u64 by = mpc_rec_index(n, ys, Z); sum += s.rec_off[by + nys] - s.rec_off[by]; }
			best = sum > best ? sum : best;
		}
		for (int d = 32; d >= 1; d >>= 1) { const u32 o = __shfl_down(best, d); best = o > best ? o : best; }
		if (t == 0) out[tl] = best;
	}
}

__device__ __forceinline__ u64 mpc_find_pair(const u64 *vbase, u64 lo, u64 hi, u64 e)
{
	// largest k in [lo,hi) with vbase[k] <= e (pairs without entries are skipped naturally)
	while (hi - lo > 1) {
		const u64 mid = (lo + hi) >> 1;
		if (vbase[mid] <= e) lo = mid; else hi = mid;
	}
	return lo;
}

// One thread per stored cell of the pairs [k0,k1).
__global__ void __launch_bounds__(256) relax_kernel(StoreParams s, u64 k0, u64 k1)
{
	const u64 first = s.vbase[k0], last = s.vbase[k1];
	const u32 n = s.n;
	for (u64 e = first + (u64)blockIdx.x * blockDim.x + threadIdx.x; e < last; e += (u64)gridDim.x * blockDim.x) {
		const u64 k = mpc_find_pair(s.vbase, k0, k1, e);
		const u32 X = s.pair_x[k], Y = s.pair_y[k];
		const u32 LX = s.seq_len[X], LY = s.seq_len[Y];
		const u32 *rec = s.packed + s.pbase[k];
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		const u32 idx = (u32)(e - s.vbase[k]);
		const u32 *ent = rec + LX + LY;
		const float pxy = __uint_as_float(ent[2 * (u64)idx]);
		const u32 y = ent[2 * (u64)idx + 1];
		const u32 x = ent[2 * (u64)nnz + idx];
		float acc = pxy * 2.0f; // conspairflat.cpp:29-30
		const u32 *rpx = s.rp + s.rp_base[X] + x;
		const u32 *rpy = s.rp + s.rp_base[Y] + y;
		const MpcEnt *ex = s.ent + s.ent_base[X];
		const MpcEnt *ey = s.ent + s.ent_base[Y];
		const u32 sxs = LX + 1, sys = LY + 1;
		for (u32 Z = 0; Z < n; ++Z) {
			u32 a = rpx[0], a1 = rpx[1];
			u32 b = rpy[0], b1 = rpy[1];
			rpx += sxs; rpy += sys;
			if (a == a1 || b == b1) continue;
			MpcEnt va = ex[a], vb = ey[b];
			for (;;) {
				if (va.c == vb.c) {
					acc += __uint_as_float(va.p) * __uint_as_float(vb.p); // relaxflat.cpp:27 (w == 1.0f)
					if (++a == a1 || ++b == b1) break;
					va = ex[a]; vb = ey[b];
				} else if (va.c < vb.c) {
					if (++a == a1) break;
					va = ex[a];
				} else {
					if (++b == b1) break;
					vb = ey[b];
				}
			}
		}
		s.vnext[e] = acc / (float)n; // mysparsemx.cpp:108 (uint -> float, IEEE divide)
	}
}

// Make vnext current everywhere it is stored: both slab orientations and the packed record
// (the swap of consflat.cpp:22). One thread per stored cell in the canonical entry range [e0, e1) (everything: [0, vbase[npairs])).
__global__ void __launch_bounds__(256) commit_kernel(StoreParams s, u64 e0, u64 e1)
{
	const u64 last = e1;
	for (u64 e = e0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; e < last; e += (u64)gridDim.x * blockDim.x) {
		const u64 k = mpc_find_pair(s.vbase, 0, s.npairs, e);
		const u32 X = s.pair_x[k], Y = s.pair_y[k];
		const u32 LX = s.seq_len[X], LY = s.seq_len[Y];
		u32 *rec = s.packed + s.pbase[k];
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		const u32 idx = (u32)(e - s.vbase[k]);
		u32 *ent = rec + LX + LY;
		const u32 pb = __float_as_uint(s.vnext[e]);
		const u32 tq = ent[3 * (u64)nnz + idx];
		ent[2 * (u64)idx] = pb;
		s.ent[s.ent_base[X] + s.mbase[(u64)X * (s.n + 1) + Y] + idx].p = pb;
		s.ent[s.ent_base[Y] + s.mbase[(u64)Y * (s.n + 1) + X] + tq].p = pb;
	}
}

// The commit of the record layout — the packed record of the pair, both orientations' block records and window records — one WAVE PER
// PAIR (round 6; the per-entry form of rounds 2-5, like commit_kernel above, found its pair by a binary search over vbase, 19
// dependent loads per thread at 499 500 pairs, and visited every entry even where nothing was to be written: 24.2 -> 16.1 ms per
// step on one GPU, 19.6 -> 7.0 ms on a rank of eight, profiles/r12c). A wave takes pair k of
// [k0, k1) and walks its entries inside [e0, e1) (coalesced reads of vnext and the position arrays); a PARTIAL store skips the pairs
// that touch none of its sequences and are not its own, and writes the packed record only for its own pairs [own0, own1) — the relax
// kernel reads P_XY of its own pairs there; everyone else's packed values are refreshed from vnext when somebody asks for them
// (packed_refresh_kernel: mpcgpu.cpp, refresh_packed).
__global__ void __launch_bounds__(64) commit_pairs_kernel(StoreParams s, u64 k0, u64 k1, u64 e0, u64 e1, u64 own0, u64 own1, int lazy_packed)
{
	const u32 t = threadIdx.x;
	for (u64 k = k0 + blockIdx.x; k < k1; k += gridDim.x) {
		const u32 X = s.pair_x[k], Y = s.pair_y[k];
		const bool nx = mpc_need(s, X), ny = mpc_need(s, Y);
		const bool pk = !lazy_packed || (k >= own0 && k < own1);
		if (!nx && !ny && !pk) continue;
		const u64 vb = s.vbase[k], ve = s.vbase[k + 1];
		const u64 a = vb > e0 ? vb : e0, b = ve < e1 ? ve : e1;
		if (a >= b) continue;
		const u32 LX = s.seq_len[X], LY = s.seq_len[Y];
		u32 *ent = s.packed + s.pbase[k] + LX + LY;
		u32 *rx = s.pad + 4 * (u64)s.rec_off[mpc_rec_index(s.n, X, Y)], *ry = s.pad + 4 * (u64)s.rec_off[mpc_rec_index(s.n, Y, X)];
		u32 *wx = s.win ? s.win + 4 * ((u64)s.wrec_off[mpc_rec_index(s.n, X, Y)] + (LX + 1u + 3u) / 4u) : nullptr;
		u32 *wy = s.win ? s.win + 4 * ((u64)s.wrec_off[mpc_rec_index(s.n, Y, X)] + (LY + 1u + 3u) / 4u) : nullptr;
		for (u64 e = a + t; e < b; e += 64) {
			const u32 pb = __float_as_uint(s.vnext[e]);
			if (pk) ent[2 * (e - vb)] = pb;
			// entry `pos` of a record = block pos / 2, slot pos % 2: its probability is dword block * 4 + slot
			if (nx) { const u32 pf = s.pos_f[e]; rx[(pf >> 1) * 4u + (pf & 1u)] = pb; if (wx) wx[s.pos_wf[e]] = pb; }
			if (ny) { const u32 pt = s.pos_t[e]; ry[(pt >> 1) * 4u + (pt & 1u)] = pb; if (wy) wy[s.pos_wt[e]] = pb; }
		}
	}
}

// packed P of the pairs [k0, k1) OUTSIDE [own0, own1), entries [e0, e1), from vnext: what commit_pairs_kernel left out when it
// committed that range of entries lazily
__global__ void __launch_bounds__(64) packed_refresh_kernel(StoreParams s, u64 k0, u64 k1, u64 e0, u64 e1, u64 own0, u64 own1)
{
	const u32 t = threadIdx.x;
	for (u64 k = k0 + blockIdx.x; k < k1; k += gridDim.x) {
		if (k >= own0 && k < own1) continue;
		const u64 vb = s.vbase[k], ve = s.vbase[k + 1];
		const u64 a = vb > e0 ? vb : e0, b = ve < e1 ? ve : e1;
		u32 *ent = s.packed + s.pbase[k] + s.seq_len[s.pair_x[k]] + s.seq_len[s.pair_y[k]];
		for (u64 e = a + t; e < b; e += 64) ent[2 * (e - vb)] = __float_as_uint(s.vnext[e]);
	}
}

// Pack one batch of post_kernel records (fixed stride) into the packed buffer.
__global__ void __launch_bounds__(256) pack_kernel(const u32 *res, u64 res_stride, const u64 *dst_base,
	const u64 *rec_words, u32 *packed, u32 count)
{
	for (u32 pid = blockIdx.x; pid < count; pid += gridDim.x) {
		const u32 *src = res + (u64)pid * res_stride;
		u32 *dst = packed + dst_base[pid];
		const u64 w = rec_words[pid];
		for (u64 q = threadIdx.x; q < w; q += blockDim.x) dst[q] = src[q];
	}
}

// Export pairs [k0,k1) in MySparseMx layout: offsets (LX+1 per pair, exclusive scan of rowcnt) and
// values ({P,col} per entry) into two contiguous staging arrays. klist != null: the pairs are klist[q], q in [k0,k1), and their
// values go to entry val_base[q] (a custom pair order: an InitPairs range is a list of positions).
__global__ void __launch_bounds__(64) export_kernel(StoreParams s, u64 k0, u64 k1, const u64 *off_base,
	u32 *out_off, u32 *out_val, const u64 *klist, const u64 *val_base)
{
	const int t = threadIdx.x;
	for (u64 q = k0 + blockIdx.x; q < k1; q += gridDim.x) {
		const u64 k = klist ? klist[q] : q;
		const u32 LX = s.seq_len[s.pair_x[k]], LY = s.seq_len[s.pair_y[k]];
		const u32 *rec = s.packed + s.pbase[k];
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		u32 *oo = out_off + off_base[q - k0];
		u32 carry = 0;
		for (u32 a0 = 0; a0 <= LX; a0 += 64) {
			const u32 a = a0 + t;
			const u32 v = (a < LX) ? rec[a] : 0;
			u32 incl = v;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (t >= d) incl += o;
			}
			if (a <= LX) oo[a] = carry + incl - v;
			carry += __shfl(incl, 63);
		}
		const u32 *ent = rec + LX + LY;
		u32 *ov = out_val + 2 * (klist ? val_base[q] : s.vbase[k] - s.vbase[k0]);
		for (u32 i = t; i < 2 * nnz; i += 64) ov[i] = ent[i];
	}
}
