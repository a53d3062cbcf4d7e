// kernels_relaxb.h — consistency relax over BAND tiles of the variable-size records: relax_band_kernel (the default since round 4).
//
// Same arithmetic, same order of additions and the same merge step as relax_var_kernel (kernels_relaxv.h; reference:
// conspairflat.cpp:10-110 -> relaxflat.cpp:4-94 -> mysparsemx.cpp:87-113): per stored cell (x,y) of (X,Y)
//     acc = 2*P_XY(x,y);  for Z = 0..N-1: acc += sum_z M(X,Z)(x,z) * M(Y,Z)(y,z)  (z ascending);  P' = acc / N
// What changed is WHICH cells a workgroup owns and WHAT it stages per step. relax_var_kernel owned all cells of <= 4x4 pairs and
// staged their <= 8 whole records per step (65 KB of the 80 KB a workgroup has: no room for step Z+1, a quarter of the kernel an
// exposed DMA wait; on wide-row data two 24 KB records per pair-step and no reuse at all). Stored cells hug the alignment
// diagonal, so the cells of the ROW BAND [r0,r1) of a pair need only rows [r0,r1) of M(X,Z) and a contiguous row range
// [ylo,yhi) of M(Y,Z). A band tile is
//     {X in [x0,x0+nx)} x {Y in [y0,y0+ny)} x rows [r0,r1) of the X sequences,   nx, ny <= 8:
// up to 64 pairs share 16 PARTIAL records per step — at 1000 x L~400: 8x8 pairs x ~100 rows, 33 KB per step for 15 pair-
// equivalents (whole records: 65 KB for 16), and two steps fit the LDS: the DMA of step Z+1 runs under the merges of step Z.
//
// A partial record is two pieces, because a record is [first block of every row][overflow blocks in row order]
// (kernels_store.h): rows [a0,a1) of the first-block region (row a at block a: no table needed) and the overflow blocks of the
// index bands (MPC_RB_HB = 8 rows) those rows lie in (ovf_off: where the overflow of band b of record (A,Z) starts, written by
// var_build_kernel). In LDS the first pieces of all records come first, at offsets that do NOT depend on Z — a cell's two row
// addresses inside a step's buffer are constants of the walk, no per-step base — and the overflow pieces follow back to back,
// at offsets that do. The only thing the merge needs per step and record is the HOP BIAS: a first block's distance field is
// relative to the record as it lies in HBM; (address of the first block in LDS) + bias + distance is its overflow block in
// LDS. The bias of a slot's X record is a scalar (cells are laid out X-major and every X group is rounded up to whole waves,
// so the 64 cells of a (wave, slot) share their X record); the Y record differs per lane: one cross-lane gather per slot.
// Inside an X group the cells go in blocks of 8 rows, a block's cells pair after pair (see "cell order" in the kernel).
//
// The Y operand comes in one of two forms: block records walked with the X row (two-list merge, MpcRbBlocksAsm), or WINDOW
// records looked up by column (direct-index merge, MpcRbWinAsm: the default where rows are narrow) — same tiles, same staging.
//
// Buffers: a step's pieces are placed at the bottom or at the top of the staging area, alternating; the next step is
// prefetched when both fit (the host cuts the bands so that they do on average), otherwise — wide-row data: one step fills most
// of the area — it is staged after the merges, as relax_var_kernel did, and the CU's second workgroup fills the wait.
#pragma once
#include "kernels_relaxv.h"

// MPC_RB_HB (kernels_store.h): rows per index band — overflow offsets, cell offsets and y ranges are kept per band
#define MPC_RB_MAXN 8u          // sequences per side of a tile
#define MPC_RB_TILE_WORDS 16u   // x0, nx, y0, ny, r0, r1, first-piece blocks, slots | (X records with cells) << 8, then ylo | yhi << 16 per Y (yhi exclusive)
#define MPC_RB_TAB_BYTES 2560u  // tables at the head of the dynamic LDS: pairs 64 x 16 B, records 16 x 32 B, groups, misc, step tables, biases
#define MPC_RB_PTAB 0u
#define MPC_RB_RTAB 1024u
#define MPC_RB_GTAB 1536u       // 8 x {group base, group end}
#define MPC_RB_MISC 1600u       // tile number, total cells, first-piece blocks
#define MPC_RB_TTAB 1664u       // step tables of two steps: 2 x 64 words
#define MPC_RB_BTAB 2176u       // hop biases of two steps: 2 x 16 words (.. 2304; the rest is spare)
#define MPC_RB_MAXFIRST 4095u   // blocks of first pieces per step: a cell keeps its two byte offsets in 16 bits each

// relax_band_kernel's merge of one (cell, Z) in C++ (what MpcRbBlocksAsm, mpc_platform.h, does with hand-scheduled instructions;
// the emulator runs this one, the device can: MPCGPU_RELAX_MERGE=cxx). ia: address of the X row's first block + the X record's hop
// bias; ib: address of the Y row's first block — its record's bias, gathered one slot ahead, is added here.
struct MpcRbBlocksCxx {
	static constexpr bool WINDOW = false;
	MpcQuad a[2], b[2];
	u32 hb;
	__device__ __forceinline__ void load(u32 ia, u32 ib, u32 hb_addr) { a[0] = mpc_lds_load16(ia); b[0] = mpc_lds_load16(ib); hb = mpc_lds_load4(hb_addr); } // hb_addr: LDS address of the first slot's Y bias
	__device__ __forceinline__ void drain() {}
	template <int SET> __device__ __forceinline__ void merge(float &sum, u32 ia, u32 ib, u32 nia, u32 nib, u32 nidx, u32 bias_y)
	{
		const MpcQuad va = a[SET], vb = b[SET];
		ib += hb;
		a[SET ^ 1] = mpc_lds_load16(nia); b[SET ^ 1] = mpc_lds_load16(nib); // the next slot's first blocks and Y bias: in flight during this slot's arithmetic
		hb = mpc_lane_gather(bias_y, nidx);
		mpc_rv_merge_cxx(sum, va, vb, ia, ib);
	}
};
#ifndef MPC_RB_HAVE_ASM
typedef MpcRbBlocksCxx MpcRbBlocksAsm; // the emulator has the C++ statement only
#endif

// The DIRECT-INDEX merge in C++ (MpcRbWinAsm, mpc_platform.h, is the hand-scheduled form): the X row is walked block by block,
// every X entry (z, P) looks its partner up in the Y row's WINDOW (StoreParams::win): value = val[off + min(z - c0, span)] — a
// column the row does not store reads 0.0f (inside the window) or the 0.0f guard at off + span (outside, also below c0: the
// unsigned difference wraps). P * 0.0f = +0.0f leaves the strictly positive sum unchanged, so the additions that count are those of
// the reference, in z order. yd: LDS address of the Y row's descriptor word (this slot's: read again only for a wide row); hb: LDS address of the value
// area of the cell's Y record as it lies in this step's buffer (gathered one slot ahead).
struct MpcRbWinCxx {
	static constexpr bool WINDOW = true;
	MpcQuad a[2];
	u32 d0[2];
	u32 hb;
	__device__ __forceinline__ void load(u32 ia, u32 yd, u32 hb_addr) { a[0] = mpc_lds_load16(ia); d0[0] = mpc_lds_load4(yd); hb = mpc_lds_load4(hb_addr); }
	__device__ __forceinline__ void drain() {}
	template <int SET> __device__ __forceinline__ void merge(float &sum, u32 ia, u32 yd, u32 nia, u32 nyd, u32 nidx, u32 bias_y)
	{
		MpcQuad va = a[SET];
		const u32 D0 = d0[SET], base = hb;
		a[SET ^ 1] = mpc_lds_load16(nia); d0[SET ^ 1] = mpc_lds_load4(nyd);
		hb = mpc_lane_gather(bias_y, nidx);
		const u32 c0 = D0 & 0xfffu, vb = base + 4u * (D0 >> 17); // kernels_store.h: c0 | span << 12 | off << 17
		u32 span = (D0 >> 12) & 31u;
		if (span == MPC_WIN_MAXSPAN) span = (mpc_lds_load4(yd + 4u) >> 17) - (D0 >> 17) - 1u; // the escape: a wide row
		for (;;) {
			u32 j0 = (va.z & 0xffffu) - c0, j1 = va.w - c0;
			j0 = j0 < span ? j0 : span; j1 = j1 < span ? j1 : span;
			const float v0 = __uint_as_float(mpc_lds_load4(vb + 4u * j0)), v1 = __uint_as_float(mpc_lds_load4(vb + 4u * j1));
			sum += __uint_as_float(va.x) * v0; // relaxflat.cpp:27 (w == 1.0f): product rounded, then added
			sum += __uint_as_float(va.y) * v1;
			const u32 d = va.z >> 16;
			if (d == 0u) break;
			ia += d;
			va = mpc_lds_load16(ia);
		}
	}
};
#ifndef MPC_RW_HAVE_ASM
typedef MpcRbWinCxx MpcRbWinAsm;
#endif

// Wave scans of the walk kernel's prologue through ds_bpermute with the lane number passed in: the caller passes a lane number
// that went through MPC_OPAQUE inside the tile loop — __shfl_up's own lane number (and the byte index made of it) is a loop
// invariant the compiler keeps in two VGPRs across the walk.
__device__ __forceinline__ u32 rb_lane_up(u32 v, u32 d, u32 ln) { return mpc_lane_gather(v, 4u * ((ln - d) & 63u)); } // value of lane ln - d
__device__ __forceinline__ u32 rb_lane_at(u32 v, u32 l) { return mpc_lane_gather(v, 4u * l); }

struct RelaxBandParams {
	StoreParams s;
	const u32 *ovf_off;  // [(Z*n + A) * nb1 + b]: block index (in `pad`) where the overflow blocks of rows >= b*HB of record (A,Z) start; the record's end for b*HB >= len(A)
	u32 nb1;             // entries per record / per pair in the band tables: ceil(max_len / HB) + 1
	const u32 *cell_off; // [k * nb1 + b]: stored cells of pair k in rows < b*HB (canonical entry order is row-major)
	const u32 *tiles;    // MPC_RB_TILE_WORDS per tile
	u32 ntiles;
	u64 k0, k1;          // only pairs in [k0,k1) are relaxed (multi-GPU shard)
	u32 cap_bytes;       // staging area (after the tables)
	u32 *tile_next;      // 8 counters, zeroed before the launch: next tile of each XCD's range
	u32 by_rows;         // cell order inside an X group: G > 0 = blocks of G rows, a block's cells pair after pair; 0 = pair after pair
};

// ---- band tables -------------------------------------------------------------------------------------------------------------
// cell_off (see RelaxBandParams) and yr[k * nb1 + b] = ymin | ymax << 16 over the stored cells of pair k in rows [b*HB, (b+1)*HB)
// (0xffff | 0 when there are none). One wave per pair; the columns of a row ascend (MySparseMx order).
__global__ void __launch_bounds__(64) band_index_kernel(StoreParams s, u32 nb1, u32 *cell_off, u32 *yr)
{
	const u32 t = threadIdx.x;
	for (u64 k = blockIdx.x; k < s.npairs; k += gridDim.x) {
		if (!mpc_need(s, s.pair_x[k]) || !mpc_need(s, s.pair_y[k])) continue; // a partial store relaxes pairs of its own sequences only
		const u32 LX = s.seq_len[s.pair_x[k]], LY = s.seq_len[s.pair_y[k]];
		const u32 *rec = s.packed + s.pbase[k];
		const u32 *ent = rec + LX + LY;
		const u32 nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
		u32 carry = 0;
		for (u32 a0 = 0; a0 < LX; a0 += 64) {
			const u32 a = a0 + t;
			const u32 v = a < LX ? rec[a] : 0u;
			u32 incl = v;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (t >= (u32)d) incl += o;
			}
			const u32 start = carry + incl - v;
			u32 lo = v ? ent[2 * (u64)start + 1] : 0xffffu;
			u32 hi = v ? ent[2 * (u64)(start + v - 1) + 1] : 0u;
			for (int d = (int)MPC_RB_HB / 2; d >= 1; d >>= 1) { // the rows of a band are consecutive lanes
				const u32 ol = __shfl_down(lo, d), oh = __shfl_down(hi, d);
				lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
			}
			if ((t & (MPC_RB_HB - 1u)) == 0u && a < LX) {
				cell_off[k * nb1 + a / MPC_RB_HB] = start;
				yr[k * nb1 + a / MPC_RB_HB] = lo | (hi << 16);
			}
			carry += __shfl(incl, 63);
		}
		for (u32 b = (LX + MPC_RB_HB - 1u) / MPC_RB_HB + t; b < nb1; b += 64) { cell_off[k * nb1 + b] = nnz; yr[k * nb1 + b] = 0xffffu; }
	}
}

// Per (sequence A, band b), over all Z: ovf_sum[A*nb1+b] = sum_Z (overflow blocks of record (A,Z) in rows < b*HB) — the mean the
// tile cutter estimates a step's LDS need with — and ovf_maxc[A*nb1+b] = sum over the GROUPS of MPC_RB_BG index bands that start
// before band b of max_Z (overflow blocks of the group): an upper bound of any step's need for a range of whole groups (a range
// is rounded out to groups; per single index band the sum of maxima was too loose to ever settle the question). One wave per
// sequence.
#define MPC_RB_BG 4u
// (win != 0: the same over the VALUE areas of the window records: ovf_off = wv_off, a record's dynamic part starts after its
// descriptor blocks)
__global__ void __launch_bounds__(64) ovf_stats_kernel(StoreParams s, const u32 *ovf_off, u32 nb1, u32 *ovf_sum, u32 *ovf_maxc, int win)
{
	const u32 t = threadIdx.x, n = s.n;
	const u32 *rstart = win ? s.wrec_off : s.rec_off;
	for (u32 A = blockIdx.x; A < n; A += gridDim.x) {
		if (!mpc_need(s, A)) continue; // a partial store: no records of this sequence
		const u32 LA = win ? (s.seq_len[A] + 1u + 3u) / 4u : s.seq_len[A]; // blocks of the static part of a record
		for (u32 b = t; b < nb1; b += 64) {
			u32 sum = 0, umax = 0u, lmin = 0xffffffffu;
			for (u32 Z = 0; Z < n; ++Z) {
				const u64 r = mpc_rec_index(n, A, Z);
				const u32 pre = ovf_off[r * nb1 + b] - (rstart[r] + LA); // dynamic blocks of record (A,Z) before band b
				sum += pre;
				if (Z != A) { umax = pre > umax ? pre : umax; lmin = pre < lmin ? pre : lmin; } // (the record (A,A) is empty: it stages nothing)
			}
			ovf_sum[(u64)A * nb1 + b] = sum;
			// round 6: the largest and the smallest prefix over Z — max_Z (pre(e1) - pre(e0)) <= umax(e1) - lmin(e0): an upper bound of a
			// record's dynamic piece for ANY band range, within ~1.7x of the mean where the group-wise maxima below were 3.5x (window value
			// areas), so that most tiles no longer need the exact walk over Z (band_fit_kernel)
			ovf_sum[(u64)n * nb1 + (u64)A * nb1 + b] = umax;
			ovf_sum[2ull * n * nb1 + (u64)A * nb1 + b] = lmin == 0xffffffffu ? 0u : lmin;
		}
		const u32 ng = (nb1 - 1u + MPC_RB_BG - 1u) / MPC_RB_BG; // groups of bands 0 .. nb1-2 (band nb1-1 is the end marker)
		u32 carry = 0;
		if (t == 0u) ovf_maxc[(u64)A * nb1] = 0u;
		for (u32 g0 = 0; g0 < ng; g0 += 64) {
			const u32 g = g0 + t;
			u32 mx = 0;
			if (g < ng) {
				const u32 lo = g * MPC_RB_BG, hi = lo + MPC_RB_BG < nb1 - 1u ? lo + MPC_RB_BG : nb1 - 1u;
				for (u32 Z = 0; Z < n; ++Z) {
					const u64 r = mpc_rec_index(n, A, Z);
					const u32 d = ovf_off[r * nb1 + hi] - ovf_off[r * nb1 + lo];
					mx = d > mx ? d : mx;
				}
			}
			u32 incl = mx;
			for (int d = 1; d < 64; d <<= 1) {
				const u32 o = __shfl_up(incl, d);
				if (t >= (u32)d) incl += o;
			}
			if (g < ng)
				for (u32 b = g * MPC_RB_BG + 1u; b <= (g + 1u) * MPC_RB_BG && b < nb1; ++b) ovf_maxc[(u64)A * nb1 + b] = carry + incl;
			carry += __shfl(incl, 63);
		}
	}
}

// ---- tiles: statistics of one band tile, by one wave (lane = pair ix*8+iy) --------------------------------------------------------
struct RbTileTabs {
	const u32 *cell_off, *yr, *ovf_sum, *ovf_maxc;
	const u32 *ysum, *ymaxc; // the same two tables for the Y records' dynamic pieces: the overflow blocks again, or (win) the value areas of the window records
	u32 win;                 // the Y records are staged as window records: descriptor rows (4 per block) + values
	u32 nb1, threads;
	u64 k0, k1;
};
struct RbTileStats {
	u32 cells;       // stored cells of the tile's pairs in the band
	u32 slots;       // cells per lane the tile needs (X groups rounded up to whole waves)
	u32 first;       // blocks of the first pieces
	u32 est, bound;  // blocks per step: mean over Z (rounded up) / upper bound
	u32 yr;          // lane iy (< 8): ylo | yhi << 16 of Y record iy (yhi exclusive; 0 when it has no cell)
	u32 xmask;       // bit ix: X record ix has cells in the band (an X sequence without any is not staged: the last one of a tile on
	                 // the diagonal, the sequences of a neighbouring block at the ragged edge of a rank's blocks — whose records a
	                 // partial store does not even hold)
};
__device__ __forceinline__ u32 rb_wave_sum(u32 v) { for (int d = 1; d < 64; d <<= 1) v += __shfl(v, (int)((threadIdx.x & 63u) ^ (u32)d)); return v; }

// c, lo, hi: per lane, the cells of the lane's pair in the band and their column range (lo > hi: none); b0, b1: the band range.
__device__ __forceinline__ RbTileStats rb_tile_stats(const StoreParams &s, const RbTileTabs &tb, u32 x0, u32 nx, u32 y0, u32 ny, u32 b0, u32 b1,
	u32 c, u32 lo, u32 hi)
{
	const u32 lane = threadIdx.x & 63u, ix = lane >> 3, iy = lane & 7u, n = s.n;
	RbTileStats r;
	// Y ranges: over the X of the tile (lanes with the same iy: xor 8, 16, 32)
	for (int d = 8; d < 64; d <<= 1) {
		const u32 ol = __shfl(lo, (int)(lane ^ (u32)d)), oh = __shfl(hi, (int)(lane ^ (u32)d));
		lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
	}
	const bool yany = lo <= hi;
	const u32 ylo = yany ? lo : 0u, yhi = yany ? hi + 1u : 0u;
	r.yr = ylo | (yhi << 16);
	// cells per X group (lanes with the same ix: xor 1, 2, 4), each rounded up to whole waves
	u32 g = c;
	for (int d = 1; d < 8; d <<= 1) g += __shfl(g, (int)(lane ^ (u32)d));
	const u32 grnd = (g + 63u) & ~63u;
	r.cells = rb_wave_sum(iy == 0u ? g : 0u);
	const u32 rounded = rb_wave_sum(iy == 0u ? grnd : 0u);
	r.slots = (rounded + tb.threads - 1u) / tb.threads;
	// pieces: lane ix*8 speaks for X record ix, lane iy (ix == 0) for Y record iy
	u32 first = 0, sum = 0, mxc = 0, mx2 = 0, my2 = 0;
	const u64 N1 = (u64)n * tb.nb1; // ovf_sum / ysum are three tables back to back: sums, largest prefix, smallest prefix
	r.xmask = rb_wave_sum((iy == 0u && ix < nx && g != 0u) ? 1u << ix : 0u);
	if (iy == 0u && ix < nx && g != 0u) {
		const u32 A = x0 + ix, LA = s.seq_len[A];
		const u32 a0 = b0 * MPC_RB_HB, a1 = (b1 * MPC_RB_HB < LA) ? b1 * MPC_RB_HB : LA;
		if (a1 > a0) {
			const u32 e1 = (a1 + MPC_RB_HB - 1u) / MPC_RB_HB;
			first = a1 - a0;
			sum = tb.ovf_sum[(u64)A * tb.nb1 + e1] - tb.ovf_sum[(u64)A * tb.nb1 + b0];
			mxc = tb.ovf_maxc[(u64)A * tb.nb1 + e1] - tb.ovf_maxc[(u64)A * tb.nb1 + b0 / MPC_RB_BG * MPC_RB_BG];
			const u32 hi2 = tb.ovf_sum[N1 + (u64)A * tb.nb1 + e1], lo2 = tb.ovf_sum[2 * N1 + (u64)A * tb.nb1 + b0];
			mx2 = hi2 > lo2 ? hi2 - lo2 : 0u;
		}
	}
	u32 fy = 0, sy = 0, my = 0;
	if (ix == 0u && iy < ny && yany) {
		const u32 A = y0 + iy;
		const u32 e0 = ylo / MPC_RB_HB, e1 = (yhi + MPC_RB_HB - 1u) / MPC_RB_HB;
		// window records: descriptors of rows [ylo & ~3, yhi] (4 per block) and the value blocks of the bands, + the block the range ends in
		fy = tb.win ? (yhi + 1u - (ylo & ~3u) + 3u) / 4u : yhi - ylo;
		sy = tb.ysum[(u64)A * tb.nb1 + e1] - tb.ysum[(u64)A * tb.nb1 + e0] + (tb.win ? n : 0u);
		my = tb.ymaxc[(u64)A * tb.nb1 + e1] - tb.ymaxc[(u64)A * tb.nb1 + e0 / MPC_RB_BG * MPC_RB_BG] + (tb.win ? 1u : 0u);
		const u32 hi2 = tb.ysum[N1 + (u64)A * tb.nb1 + e1], lo2 = tb.ysum[2 * N1 + (u64)A * tb.nb1 + e0];
		my2 = (hi2 > lo2 ? hi2 - lo2 : 0u) + (tb.win ? 1u : 0u);
	}
	r.first = rb_wave_sum(first + fy);
	const u32 tot = rb_wave_sum(sum + sy);
	r.est = r.first + (tot + n - 1u) / n;
	const u32 bound_groups = r.first + rb_wave_sum(mxc + my);
	const u32 bound_prefix = r.first + rb_wave_sum(mx2 + my2); // (ovf_stats_kernel: largest prefix at the range's end - smallest at its start)
	r.bound = bound_prefix < bound_groups ? bound_prefix : bound_groups;
	return r;
}

// the lane's pair of tile (x0, nx, y0, ny): its index, or ~0 when the lane has none inside [k0,k1)
__device__ __forceinline__ u64 rb_lane_pair(const StoreParams &s, const RbTileTabs &tb, u32 x0, u32 nx, u32 y0, u32 ny)
{
	const u32 lane = threadIdx.x & 63u, ix = lane >> 3, iy = lane & 7u;
	const u32 X = x0 + ix, Y = y0 + iy;
	if (ix >= nx || iy >= ny || X >= Y) return ~0ull;
	const u64 k = mpc_pair_pos(s, X, Y);
	return (k >= tb.k0 && k < tb.k1) ? k : ~0ull;
}

// Cuts the super-tiles (x0, nx, y0, ny) of `cand` (4 words each) into row bands of about equal numbers of cells: the fewest bands
// such that none needs more than `max_slots` cells per lane or more than `target` blocks per step (mean over Z) — a band of one
// index band that does is emitted as it is (the host splits it by sequences). write == 0: count[c] = bands of candidate c;
// write == 1: the bands' tile words 0..5 go to tiles + MPC_RB_TILE_WORDS * base[c]. One wave per candidate; dynamic LDS: nb1 words.
__global__ void __launch_bounds__(64) band_cut_kernel(StoreParams s, RbTileTabs tb, const u32 *cand, u32 ncand, u32 max_slots, u32 target,
	int write, u32 *count, const u32 *base, u32 *tiles)
{
	MPC_DYN_SMEM(smem_raw);
	u32 *cum = (u32 *)smem_raw; // cum[b]: cells of the super-tile's pairs in rows < b*HB
	const u32 lane = threadIdx.x & 63u;
	for (u32 ci = blockIdx.x; ci < ncand; ci += gridDim.x) {
		const u32 x0 = cand[4 * ci], nx = cand[4 * ci + 1], y0 = cand[4 * ci + 2], ny = cand[4 * ci + 3];
		const u64 k = rb_lane_pair(s, tb, x0, nx, y0, ny);
		const u32 *co = tb.cell_off + (k == ~0ull ? 0ull : k) * tb.nb1, *yrp = tb.yr + (k == ~0ull ? 0ull : k) * tb.nb1;
		u32 maxlen = 0;
		for (u32 q = 0; q < nx; ++q) { const u32 l = s.seq_len[x0 + q]; maxlen = l > maxlen ? l : maxlen; }
		const u32 nb = (maxlen + MPC_RB_HB - 1u) / MPC_RB_HB;
		MPC_WAVE_LDS_ORDER(); // the previous candidate's readers are done with cum
		for (u32 b = 0; b <= nb; ++b) {
			const u32 v = rb_wave_sum(k == ~0ull ? 0u : co[b]);
			if (lane == 0u) cum[b] = v;
		}
		MPC_WAVE_LDS_ORDER();
		const u32 T = cum[nb];
		u32 emitted = 0;
		if (T != 0u) {
			// band [b0, b1) of the cut into `parts`: the index band boundary nearest to the even share
			auto boundary = [&](u32 i, u32 parts, u32 b0) -> u32 {
				if (i == parts) return nb;
				const u32 goal = (u32)(((u64)T * i) / parts);
				u32 b1 = b0 + 1u;
				while (b1 < nb && cum[b1] < goal) ++b1;
				if (b1 > b0 + 1u && cum[b1] - goal > goal - cum[b1 - 1u]) --b1;
				return b1;
			};
			auto fits = [&](u32 b0, u32 b1) -> bool {
				u32 c = 0, lo = 0xffffu, hi = 0u;
				if (k != ~0ull) {
					c = co[b1] - co[b0];
					for (u32 b = b0; b < b1; ++b) {
						if (co[b + 1] == co[b]) continue;
						const u32 w = yrp[b];
						lo = (w & 0xffffu) < lo ? (w & 0xffffu) : lo; hi = (w >> 16) > hi ? (w >> 16) : hi;
					}
				}
				const RbTileStats st = rb_tile_stats(s, tb, x0, nx, y0, ny, b0, b1, c, lo, hi);
				return st.slots <= max_slots && st.est <= target;
			};
			auto emit = [&](u32 b0, u32 b1) {
				if (cum[b1] == cum[b0]) return;
				if (write && lane == 0u) {
					u32 *t = tiles + (u64)MPC_RB_TILE_WORDS * (base[ci] + emitted);
					t[0] = x0; t[1] = nx; t[2] = y0; t[3] = ny; t[4] = b0 * MPC_RB_HB; t[5] = b1 * MPC_RB_HB;
				}
				++emitted;
			};
			// greedy first: a band takes index bands while it fits (running cell counts and column ranges: one evaluation per index
			// band). That cut is always valid, but its last band is whatever was left over (25 index bands as 6+6+6+6+1); when an even
			// cut into the same number of bands — or one more — fits everywhere, that one is taken. (Raising the number of bands until
			// an even cut fits is not an option: one dense region would shred the whole super-tile into single index bands.)
			u32 *gcut = cum + (nb + 1u); // the greedy cut's boundaries (second half of the dynamic LDS)
			u32 greedy = 0, nbnd = 0;    // bands with cells / boundaries stored
			{
				u32 b0 = 0, c = 0, lo = 0xffffu, hi = 0u;
				for (u32 b = 0; b < nb; ++b) {
					const u32 cb = k == ~0ull ? 0u : co[b + 1] - co[b];
					const u32 w = (k == ~0ull || !cb) ? 0xffffu : yrp[b];
					const u32 blo = cb ? (w & 0xffffu) : 0xffffu, bhi = cb ? (w >> 16) : 0u;
					const u32 nc = c + cb, nlo = blo < lo ? blo : lo, nhi = bhi > hi ? bhi : hi;
					if (b > b0) {
						const RbTileStats st = rb_tile_stats(s, tb, x0, nx, y0, ny, b0, b + 1u, nc, nlo, nhi);
						if (st.slots > max_slots || st.est > target) { // band b does not go in: [b0, b) is closed
							if (lane == 0u) gcut[nbnd] = b;
							++nbnd;
							if (cum[b] != cum[b0]) ++greedy;
							b0 = b; c = cb; lo = blo; hi = bhi;
							continue;
						}
					}
					c = nc; lo = nlo; hi = nhi;
				}
				if (lane == 0u) gcut[nbnd] = nb;
				++nbnd;
				if (cum[nb] != cum[b0]) ++greedy;
			}
			MPC_WAVE_LDS_ORDER();
			u32 parts = 0;
			for (u32 cand_parts = greedy; cand_parts <= greedy + 1u && cand_parts <= nb && !parts; ++cand_parts) {
				bool ok = cand_parts != 0u;
				for (u32 i = 1, b0 = 0; i <= cand_parts && ok; ++i) {
					const u32 b1 = boundary(i, cand_parts, b0);
					if (cum[b1] != cum[b0] && b1 - b0 > 1u && !fits(b0, b1)) ok = false;
					b0 = b1;
				}
				if (ok) parts = cand_parts;
			}
			if (parts) {
				for (u32 i = 1, b0 = 0; i <= parts; ++i) { const u32 b1 = boundary(i, parts, b0); emit(b0, b1); b0 = b1; }
			} else {
				for (u32 i = 0, b0 = 0; i < nbnd; ++i) { const u32 b1 = gcut[i]; emit(b0, b1); b0 = b1; } // the greedy cut itself
			}
		}
		if (!write && lane == 0u) count[ci] = emitted;
	}
}

// Fills tile words 6..15 (first-piece blocks, slots, Y ranges) of tiles whose words 0..5 are set, and out[4t..] = slots, mean
// blocks per step, upper bound of the blocks of any step, cells. One wave per tile.
__global__ void __launch_bounds__(64) band_eval_kernel(StoreParams s, RbTileTabs tb, u32 *tiles, u32 ntiles, u32 *out)
{
	const u32 lane = threadIdx.x & 63u;
	for (u32 t = blockIdx.x; t < ntiles; t += gridDim.x) {
		u32 *tw = tiles + (u64)MPC_RB_TILE_WORDS * t;
		const u32 x0 = tw[0], nx = tw[1], y0 = tw[2], ny = tw[3], r0 = tw[4], r1 = tw[5];
		const u32 b0 = r0 / MPC_RB_HB, b1 = (r1 + MPC_RB_HB - 1u) / MPC_RB_HB < tb.nb1 - 1u ? (r1 + MPC_RB_HB - 1u) / MPC_RB_HB : tb.nb1 - 1u;
		const u64 k = rb_lane_pair(s, tb, x0, nx, y0, ny);
		u32 c = 0, lo = 0xffffu, hi = 0u;
		if (k != ~0ull) {
			const u32 *co = tb.cell_off + k * tb.nb1, *yrp = tb.yr + k * tb.nb1;
			c = co[b1] - co[b0];
			for (u32 b = b0; b < b1; ++b) {
				if (co[b + 1] == co[b]) continue;
				const u32 w = yrp[b];
				lo = (w & 0xffffu) < lo ? (w & 0xffffu) : lo; hi = (w >> 16) > hi ? (w >> 16) : hi;
			}
		}
		const RbTileStats st = rb_tile_stats(s, tb, x0, nx, y0, ny, b0, b1, c, lo, hi);
		if (lane < 8u) tw[8 + lane] = st.yr;
		if (lane == 0u) {
			tw[6] = st.first; tw[7] = st.slots | (st.xmask << 8);
			out[4 * t] = st.slots; out[4 * t + 1] = st.est; out[4 * t + 2] = st.bound; out[4 * t + 3] = st.cells;
		}
	}
}

// The 16 records of a tile as relax_band_kernel stages them: record i < 8 is X sequence x0+i with rows [r0, min(r1, len)),
// record 8+j is Y sequence y0+j with rows [ylo_j, yhi_j). Returns the sequence, the first block of the static piece inside the
// record (*blk0), its blocks (*blocks; 0: nothing is staged), the index bands [e0, e1) whose dynamic blocks belong to the piece,
// and *arow: the row the piece's first block starts with. Block form: one block per row. Window form (win, Y records only): the
// static piece is the descriptors of rows [ylo & ~3, yhi] — 4 per block, the pair a cell reads ends with row yhi's successor.
__device__ __forceinline__ void rb_record(const u32 *seq_len, const u32 *tw, u32 i, u32 win, u32 *S, u32 *blk0, u32 *blocks, u32 *e0, u32 *e1, u32 *arow)
{
	const u32 x0 = tw[0], nx = tw[1], y0 = tw[2], ny = tw[3], r0 = tw[4], r1 = tw[5];
	u32 a0 = 0, a1 = 0, A = x0;
	if (i < MPC_RB_MAXN) {
		if (i < nx && ((tw[7] >> (8u + i)) & 1u)) { A = x0 + i; const u32 LA = seq_len[A]; a0 = r0; a1 = r1 < LA ? r1 : LA; } // (word 7: slots | X mask << 8, band_eval_kernel)
	} else if (i - MPC_RB_MAXN < ny) {
		A = y0 + (i - MPC_RB_MAXN);
		const u32 w = tw[8 + (i - MPC_RB_MAXN)];
		a0 = w & 0xffffu; a1 = w >> 16;
	}
	if (a1 <= a0) { a0 = 0; a1 = 0; }
	*S = A;
	*e0 = a0 / MPC_RB_HB; *e1 = a1 > a0 ? (a1 + MPC_RB_HB - 1u) / MPC_RB_HB : a0 / MPC_RB_HB;
	if (win && i >= MPC_RB_MAXN) {
		*arow = a0 & ~3u; *blk0 = a0 / 4u;
		*blocks = a1 > a0 ? (a1 + 1u - (a0 & ~3u) + 3u) / 4u : 0u;
	} else { *arow = a0; *blk0 = a0; *blocks = a1 - a0; }
}

// out[t] = max over Z of the blocks tile list[t] stages at step Z (first pieces + overflow pieces): the exact LDS need of its
// worst step. One wave per tile, lanes stride over Z.
// (round 6: four waves per workgroup take four CONSECUTIVE tiles of the list — the row bands of one super-tile follow each other, read the
// same 16 sequences' table rows at neighbouring band entries, and one CU's L1 then serves what four CUs' used to fetch from L2)
__global__ void __launch_bounds__(256) band_fit_kernel(StoreParams s, const u32 *ovf_off, u32 nb1, const u32 *tiles, const u32 *list, u32 nlist, u32 *out, u32 win)
{
	// lane = (Z parity, bound, record): the 16 records' band entries of one Z lie in two neighbourhoods of the tables (consecutive
	// sequences), so a step of the loop touches a handful of cache lines — with lanes over Z every load was 64 lines
	const u32 lane = threadIdx.x & 63u, n = s.n, li = lane & 15u, hi_bound = (lane >> 4) & 1u, par = lane >> 5;
	for (u32 q = blockIdx.x * 4u + (threadIdx.x >> 6); q < nlist; q += gridDim.x * 4u) {
		const u32 *tw = tiles + (u64)MPC_RB_TILE_WORDS * list[q];
		u32 S, blk0, blocks, e0, e1, arow;
		rb_record(s.seq_len, tw, li, win, &S, &blk0, &blocks, &e0, &e1, &arow);
		u32 first = (lane < 16u) ? blocks : 0u;
		first = rb_wave_sum(first);
		const bool wrec = win && li >= MPC_RB_MAXN;
		const u32 *tab = wrec ? s.wv_off : ovf_off;
		const u32 e = hi_bound ? e1 : e0;
		u32 best = 0;
		// (every lane runs the same number of turns: the shuffles below are wave-wide. Four steps of Z per turn, their loads issued
		// together: the walk is a chain of dependent table reads — 500 round trips to L2 per tile one at a time)
		for (u32 Z0 = par; Z0 < n + par; Z0 += 8u) {
			u32 v4[4];
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const u32 Z = Z0 + 2u * (u32)u, Zc = Z < n ? Z : n - 1u;
				v4[u] = tab[(mpc_rec_index(n, S, Zc)) * nb1 + e];
			}
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				const u32 v = v4[u];
				const u32 lo = __shfl(v, (int)(lane & ~16u)), hb = __shfl(v, (int)(lane | 16u));
				u32 d = (e1 != e0) ? hb - lo + (wrec ? 1u : 0u) : 0u;
				for (int k = 1; k < 16; k <<= 1) d += __shfl(d, (int)((lane & ~15u) | ((li + (u32)k) & 15u))); // sum over the 16 records (rotation inside the row)
				best = d > best ? d : best;
			}
		}
		for (int k = 32; k >= 1; k >>= 1) { const u32 o = __shfl(best, (int)(lane ^ (u32)k)); best = o > best ? o : best; }
		if (lane == 0u) out[q] = first + best;
	}
}

// ---- the relax --------------------------------------------------------------------------------------------------------------------
// THREADS: workgroup size; MAXSLOTS: cells per lane; WGS: workgroups per CU the register allocation has to allow;
// DIAG (measurement only, results wrong; compiled only with MPC_RELAX_DIAG_BUILD): 1 = staging and barriers only, 2 = merges only
// (step 0's records for every step), 3 = as 2 with the barriers; BLOCKS: MpcRbBlocksAsm (hand-scheduled merge) or MpcRbBlocksCxx.
template <int THREADS, int MAXSLOTS, int WGS, int DIAG = 0, class BLOCKS = MpcRbBlocksAsm>
__global__ void __launch_bounds__(THREADS, WGS * THREADS / 256) relax_band_kernel(RelaxBandParams p)
{
	MPC_DYN_SMEM(smem_raw);
	const StoreParams &s = p.s;
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63u;
	const u32 n = s.n;
	constexpr u32 NWAVES = THREADS / 64;
	constexpr bool WIN = BLOCKS::WINDOW;      // the Y records are window records, looked up by column (MpcRbWinAsm / MpcRbWinCxx)
	constexpr u32 YREGS = (MAXSLOTS + 5) / 6; // 4 * iy of a slot's cell: 5 bits, 6 slots per register
	const u32 wave = mpc_wave_first(tid >> 6); // scalar
	u32 *ptab = (u32 *)(smem_raw + MPC_RB_PTAB); // [64][4]: first cell, cells, pair, first entry
	u32 *rtab = (u32 *)(smem_raw + MPC_RB_RTAB); // [16][8]: first-piece block offset, rows, row0 - offset, sequence, e0, e1
	u32 *gtab = (u32 *)(smem_raw + MPC_RB_GTAB); // [8][2]: first cell of X group, end of its cells
	u32 *misc = (u32 *)(smem_raw + MPC_RB_MISC);
	u32 *ttab = (u32 *)(smem_raw + MPC_RB_TTAB);
	u32 *btab = (u32 *)(smem_raw + MPC_RB_BTAB);
	const unsigned char *padb = (const unsigned char *)s.pad;
	unsigned char *stage = smem_raw + MPC_RB_TAB_BYTES;
	const u32 lds_stage = mpc_lds_addr(stage);

	// XCD-aware schedule, as relax_var_kernel: 8 contiguous ranges of the tile list, a counter each, stealing at the end
	const u32 G = gridDim.x < 8u ? gridDim.x : 8u;
	const u32 xcd = blockIdx.x % G;
	const u32 chunk = (p.ntiles + G - 1u) / G;

	for (;;) {
		__syncthreads(); // the previous tile is done with the tables and the staging area
		if (tid == 0) {
			u32 got = 0xffffffffu;
			for (u32 k = 0; k < G && got == 0xffffffffu; ++k) {
				const u32 r = (xcd + k) % G;
				const u32 t_begin = r * chunk, t_end = (t_begin + chunk < p.ntiles) ? t_begin + chunk : p.ntiles;
				if (t_begin >= t_end) continue;
				const u32 t = atomicAdd(&p.tile_next[r], 1u);
				if (t < t_end - t_begin) got = t_begin + t;
			}
			misc[0] = got;
		}
		__syncthreads();
		const u32 tl = mpc_wave_first(misc[0]);
		if (tl == 0xffffffffu) break;
		// the prologue's pointers come from the kernarg segment again, tile by tile: held in scalar registers across the walk (they
		// were: 70 of them spilled into lanes of two VGPRs) they cost the walk an accumulator
		const auto pa = MPC_KERNARG_AGAIN(p);
		const u32 *tw = pa->tiles + (u64)MPC_RB_TILE_WORDS * tl;
		const u32 x0 = mpc_wave_first(tw[0]), nx = mpc_wave_first(tw[1]), y0 = mpc_wave_first(tw[2]), ny = mpc_wave_first(tw[3]);
		const u32 r0 = mpc_wave_first(tw[4]), r1 = mpc_wave_first(tw[5]);

		// ---- tables, by wave 0: lane = pair ix*8+iy for the cells, lane = record for the pieces
		if (tid < 64u) {
			const u32 ix = lane >> 3, iy = lane & 7u;
			const u32 X = x0 + ix, Y = y0 + iy;
			u32 cnt = 0, e0 = 0, kk = 0;
			if (ix < nx && iy < ny && X < Y) {
				const u64 k = mpc_pair_pos_f(n, pa->s.nrect, pa->s.rects, X, Y);
				if (k >= pa->k0 && k < pa->k1) {
					const u32 b0 = r0 / MPC_RB_HB, b1r = (r1 + MPC_RB_HB - 1u) / MPC_RB_HB, b1 = b1r < pa->nb1 - 1u ? b1r : pa->nb1 - 1u;
					e0 = pa->cell_off[k * pa->nb1 + b0];
					cnt = pa->cell_off[k * pa->nb1 + b1] - e0;
					kk = (u32)k;
				}
			}
			// cells laid end to end pair after pair inside an X group; every group rounded up to whole waves
			u32 ln = lane;
			MPC_OPAQUE(ln);
			u32 w = cnt;
			for (u32 d = 1; d < 8; d <<= 1) { const u32 o = rb_lane_up(w, d, ln); if (iy >= d) w += o; }
			const u32 gtot = rb_lane_at(w, ln | 7u);
			const u32 grnd = (gtot + MPC_RV_WAVE - 1u) & ~(MPC_RV_WAVE - 1u);
			u32 gi = iy == 0u ? grnd : 0u;
			for (u32 d = 1; d < 64; d <<= 1) { const u32 o = rb_lane_up(gi, d, ln); if (ln >= d) gi += o; }
			const u32 gbase = rb_lane_at(gi, ln & ~7u) - grnd;
			const u32 total = rb_lane_at(gi, 63u);
			u32 *e = ptab + 4 * lane;
			e[0] = gbase + w - cnt; e[1] = cnt; e[2] = kk; e[3] = e0;
			if (iy == 0u) { gtab[2 * ix] = gbase; gtab[2 * ix + 1] = gbase + gtot; }
			// pieces
			u32 S, row0, rows, b0, b1, arow;
			rb_record(pa->s.seq_len, tw, lane & 15u, WIN ? 1u : 0u, &S, &row0, &rows, &b0, &b1, &arow);
			u32 fi = lane < 16u ? rows : 0u;
			for (u32 d = 1; d < 16; d <<= 1) { const u32 o = rb_lane_up(fi, d, ln); if (ln >= d) fi += o; }
			const u32 ftot = rb_lane_at(fi, 15u);
			if (lane < 16u) {
				u32 *r = rtab + 8 * lane;
				const u32 fst = fi - rows;
				// [0] place of the static piece in a step's buffer (blocks), [1] its blocks, [2] its first block in the record minus [0],
				// [3] sequence, [4] / [5] index bands of the dynamic piece, [6] first row of the static piece, [7] the record's bias
				// constant: [2] again for block records (hop bias), the descriptor blocks of a window record (its value area starts there)
				r[0] = fst; r[1] = rows; r[2] = row0 - fst; r[3] = S; r[4] = b0; r[5] = b1; r[6] = arow;
				r[7] = (WIN && lane >= MPC_RB_MAXN) ? (pa->s.seq_len[S] + 1u + 3u) / 4u : row0 - fst;
			}
			if (lane == 0u) { misc[1] = total; misc[2] = ftot; }
		}
		__syncthreads();
		const u32 total = mpc_wave_first(misc[1]);
		const u32 ftot = mpc_wave_first(misc[2]); // blocks of the first pieces: where the overflow pieces start
		const u32 wave_first = wave * 64u;
		u32 nact = total > wave_first ? (total - wave_first + THREADS - 1u) / THREADS : 0u;

		// ---- my cells. Slot q of this wave covers cells [q*THREADS + wave*64, +64): all of ONE X group. Lanes past the group's
		// last cell repeat it (their sums are dropped): every lane merges real rows.
		float acc[MAXSLOTS];
		u32 xy[MAXSLOTS];       // byte offsets of the cell's two first blocks inside a step's buffer: X row | Y row << 16
		u32 yreg[YREGS];        // 4 * iy of every slot's cell, 5 bits each: the lane of its Y record in the bias gather
		u32 sel_a[(MAXSLOTS + 9) / 10]; // ix of every slot of this wave, 3 bits each (wave-uniform: scalar registers)
#pragma unroll
		for (int j = 0; j < (MAXSLOTS + 9) / 10; ++j) sel_a[j] = 0u;
#pragma unroll
		for (int j = 0; j < (int)YREGS; ++j) yreg[j] = 0u;
		// ---- cell order inside an X group. The 64 cells of a (wave, slot) run their merge loop as often as the LONGEST X row among
		// them has blocks, and at r ~ 2 a fifth of the rows of a record have a second block: laid out pair after pair, 64 consecutive
		// cells are ~35 different rows of X and nearly every (wave, slot) pays two or three rounds for the few lanes that need them
		// (40 x 400 synthetic posteriors: 2.0 - 2.2 rounds per slot). Laid out ROW BY ROW — row x of all the group's pairs, then row
		// x + 1 — the same 64 cells are ~5 rows of X: 1.4 - 1.5 rounds, 21 % fewer LDS and VALU instructions on the device
		// (SQ_INSTS_LDS 1.13e11 -> 0.90e11 per two launches) — and slower: the 32 lanes of a half wave then read descriptors and
		// values of 8 different Y records, SQ_LDS_BANK_CONFLICT 1.0e11 -> 2.3e11 cycles. In between: BLOCKS of G rows, the cells of
		// a block pair after pair (G = 8: runs of ~14 cells of one pair, ~10 X rows per slot, 1.6 rounds) keep the conflicts of the
		// pair order and most of the saved rounds: relax per step 880 ms (pairs), 929 (G = 1), 849 (G = 8), 851 (16), 886 (32).
		// Which lane owns which cell changes nothing in any sum. The order needs, per group, the cells before row r (T) and per pair
		// (PJ): prefix sums of the packed records' row counts over the band, built in the staging area before the walk and again
		// after it (13 cells per lane look themselves up twice per tile; the table does not live across the walk).
		u32 rows_g, gx0, gnx, band0, R1, tstride, gbytes; // rows_g: rows per block of the cell order; gx0, gnx: the tile's X sequences; R1: prefix entries per group (rows of the band + 1); tstride: bytes of a group's T array (u16); gbytes: + its PJ array, 8 x u16 per row
		bool by_rows;
		auto row_geometry = [&](auto P) { // (computed where it is used, from the tile's words: five values that would otherwise live across the walk)
			u32 t = misc[0];
			MPC_OPAQUE(t);
			const u32 *w = P->tiles + (u64)MPC_RB_TILE_WORDS * mpc_wave_first(t);
			const u32 ra = mpc_wave_first(w[4]), rb = mpc_wave_first(w[5]);
			gx0 = mpc_wave_first(w[0]); gnx = mpc_wave_first(w[1]);
			const u32 bb = (rb + MPC_RB_HB - 1u) / MPC_RB_HB;
			band0 = ra / MPC_RB_HB;
			R1 = ((bb < P->nb1 - 1u ? bb : P->nb1 - 1u) - band0) * MPC_RB_HB + 1u;
			tstride = (2u * R1 + 15u) & ~15u;
			gbytes = tstride + 16u * R1;
			rows_g = P->by_rows;
			by_rows = rows_g != 0u && MPC_RB_MAXN * gbytes <= P->cap_bytes;
		};
		row_geometry(pa);
		auto build_rows = [&](auto P) { // P: the kernel's argument (the epilogue reads it from the kernarg segment again)
			constexpr u32 PPW = 64u / NWAVES; // pairs of the tile per wave; lanes = rows
			static_assert(64u % NWAVES == 0u, "relax_band_kernel: 64 pairs are dealt over the waves");
			for (u32 pp = 0; pp < PPW; ++pp) {
				const u32 pr = wave * PPW + pp, gx = pr >> 3, j = pr & 7u;
				const u32 cnt = mpc_wave_first(ptab[4 * pr + 1]), k = mpc_wave_first(ptab[4 * pr + 2]);
				unsigned short *PJ = (unsigned short *)(stage + gx * gbytes + tstride);
				const u32 *rec = P->s.packed + P->s.pbase[k]; // rowcnt: the first LX words of the packed record (kernels_post.h)
				const u32 LX = P->s.seq_len[gx0 + (gx < gnx ? gx : 0u)];
				u32 carry = 0;
				u32 ln = lane;
				MPC_OPAQUE(ln);
				if (ln == 0u) PJ[j] = 0;
				for (u32 c0 = 0; c0 + 1u < R1; c0 += 64u) {
					const u32 r = c0 + ln, x = band0 * MPC_RB_HB + r;
					const u32 v = (cnt != 0u && r + 1u < R1 && x < LX) ? rec[x] : 0u;
					u32 incl = v;
					for (u32 d = 1; d < 64; d <<= 1) { const u32 o = rb_lane_up(incl, d, ln); if (ln >= d) incl += o; }
					if (r + 1u < R1) PJ[8u * (r + 1u) + j] = (unsigned short)(carry + incl);
					carry += mpc_wave_first(rb_lane_at(incl, 63u));
				}
			}
			__syncthreads();
			for (u32 i = tid; i < MPC_RB_MAXN * R1; i += THREADS) {
				const u32 gx = i / R1, r = i % R1;
				const unsigned short *PJ = (const unsigned short *)(stage + gx * gbytes + tstride) + 8u * r;
				u32 t = 0;
				for (u32 j = 0; j < MPC_RB_MAXN; ++j) t += PJ[j];
				((unsigned short *)(stage + gx * gbytes))[r] = (unsigned short)t;
			}
			__syncthreads();
		};
		if (by_rows) build_rows(pa);
		// the cell of (slot q, this lane); false: a repeat. *rowo: the cell's row when the order knows it, else 0xffffffff
		auto find_cell = [&](u32 q, u32 *kout, u32 *eout, u32 *ixo, u32 *iyo, u32 *rowo) -> bool {
			const u32 g0 = q * THREADS + wave_first;
			u32 ix = 0;
			for (u32 j = 1; j < MPC_RB_MAXN; ++j) if (mpc_wave_first(gtab[2 * j]) <= g0 && mpc_wave_first(gtab[2 * j + 1]) > mpc_wave_first(gtab[2 * j])) ix = j;
			const u32 gend = mpc_wave_first(gtab[2 * ix + 1]);
			u32 ln = lane;
			MPC_OPAQUE(ln); // (the epilogue's searches must not be the prologue's, kept in 12 registers across the walk)
			const u32 g = g0 + ln;
			const u32 ge = g < gend ? g : gend - 1u;
			u32 iy = 0;
			if (by_rows) {
				const u32 d = ge - mpc_wave_first(gtab[2 * ix]);
				const unsigned short *T = (const unsigned short *)(stage + ix * gbytes);
				const unsigned short *PJ = (const unsigned short *)(stage + ix * gbytes + tstride);
				const u32 rows = R1 - 1u, nblk = (rows + rows_g - 1u) / rows_g;
				u32 lo = 0, hi = nblk; // blocks of rows_g rows: cells before block lo <= d < cells before block hi
				while (hi - lo > 1u) {
					const u32 mid = (lo + hi) >> 1;
					if ((u32)T[mid * rows_g] <= d) lo = mid; else hi = mid;
				}
				const u32 ra = lo * rows_g, rb = ra + rows_g < rows ? ra + rows_g : rows;
				const unsigned short *A = PJ + 8u * ra, *B = PJ + 8u * rb;
				u32 dd = d - (u32)T[ra], before = 0;
				bool found = false;
				for (u32 j = 0; j < MPC_RB_MAXN; ++j) { // inside a block: pair after pair
					const u32 a = A[j], c = (u32)B[j] - a;
					if (!found) {
						if (dd < c) { iy = j; before = a; found = true; }
						else dd -= c;
					}
				}
				u32 r = ra; // the row inside the pair's run
				while (r + 1u < rb && (u32)PJ[8u * (r + 1u) + iy] - before <= dd) ++r;
				const u32 *e = ptab + 4 * (8 * ix + iy);
				*kout = e[2]; *eout = e[3] + before + dd; *ixo = ix; *iyo = iy; *rowo = band0 * MPC_RB_HB + r;
				return g < gend;
			}
			for (u32 j = 0; j < MPC_RB_MAXN; ++j) {
				const u32 b = ptab[4 * (8 * ix + j)], c = ptab[4 * (8 * ix + j) + 1];
				if (c != 0u && b <= ge && ge - b < c) iy = j;
			}
			const u32 *e = ptab + 4 * (8 * ix + iy);
			*kout = e[2]; *eout = e[3] + (ge - e[0]); *ixo = ix; *iyo = iy; *rowo = 0xffffffffu;
			return g < gend;
		};
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q) {
			MPC_SCHED_BARRIER();
			acc[q] = 1.0f; xy[q] = 0u;
			if ((u32)q < nact) {
				u32 k, e, ix, iy, rw;
				find_cell((u32)q, &k, &e, &ix, &iy, &rw);
				const u32 nnz = (u32)(pa->s.vbase[(u64)k + 1] - pa->s.vbase[k]);
				const u32 *ent = pa->s.packed + pa->s.pbase[k] + pa->s.seq_len[pa->s.pair_x[k]] + pa->s.seq_len[pa->s.pair_y[k]];
				acc[q] = __uint_as_float(ent[2 * (u64)e]) * 2.0f; // conspairflat.cpp:29-30
				const u32 col = ent[2 * (u64)e + 1], row = rw != 0xffffffffu ? rw : ent[2 * (u64)nnz + e];
				// row - row0 + first-piece offset, in bytes (rtab[..][2] = row0 - offset)
				const u32 xo = (row - rtab[8 * ix + 2]) << 4;
				const u32 yo = WIN ? (rtab[8 * (MPC_RB_MAXN + iy)] << 4) + ((col - rtab[8 * (MPC_RB_MAXN + iy) + 6]) << 2) // the row's descriptor
				                   : (col - rtab[8 * (MPC_RB_MAXN + iy) + 2]) << 4;                                        // the row's first block
				xy[q] = xo | (yo << 16);
				yreg[q / 6] |= (4u * iy) << (5 * (q % 6));
				sel_a[q / 10] |= mpc_wave_first(ix) << (3 * (q % 10));
			}
		}

		// ---- walk Z. The step table of step Z, one word per lane: lane i (< 16) = rec_off of record i, lane 16+i / 32+i = ovf_off at
		// the first / past the last index band of its piece. Wave 0 fetches it two steps ahead by LDS-DMA (4 bytes per lane) into
		// one of two slots; every wave reads it from there: neither a pointer nor a loaded table lives in registers across the merges.
		// (`ln`: the lane number through an optimisation barrier — what is derived from it is recomputed per step; hoisted out of the
		// walk, those per-lane constants were 15 registers that lived across the merges without being used there)
		auto issue_table = [&](u32 Zt) { // wave 0 only
			u32 ln = lane;
			MPC_OPAQUE(ln);
			const u32 li = ln & 15u, role = ln >> 4;
			const u32 S = rtab[8 * li + 3], b0 = rtab[8 * li + 4], b1 = rtab[8 * li + 5];
			const u32 rec = Zt * n + S; // (n * n < 2^32: build_var_store; the PRODUCT with nb1 is formed in 64 bits — a store of skewed lengths has more table entries than blocks)
			// one base pointer and an element offset (a select between two POINTERS became a two-entry table in scratch memory, read
			// back with a load whose wait — vmcnt(0) — also waited for the prefetch the wave had just issued)
			const bool wide = role == 1u || role == 2u;
			const bool wrec = WIN && li >= MPC_RB_MAXN; // a window record: its own record table and band table
			const long long off = (wide ? (long long)((wrec ? s.wv_off : p.ovf_off) - s.rec_off) + (long long)rec * (long long)p.nb1 + (long long)(role == 1u ? b0 : b1)
			                            : (wrec ? (long long)(s.wrec_off - s.rec_off) : 0ll) + (long long)rec);
			if (ln < 48u) mpc_dma4(s.rec_off + off, ttab + 64u * (Zt & 1u));
		};
		// Everything the staging of a step needs comes out of ONE round of LDS reads — the step's table (lane i = record i, in every
		// row of 16 lanes) and the records' constants — followed by register work only (a DPP row scan, v_readlane): the chain of
		// dependent LDS round trips this block used to be (uniform reads per piece, a cross-lane scan through the LDS crossbar) was
		// a microsecond per step on every wave's critical path. Returns the step's length in blocks; *bias: lane i = hop bias of
		// record i (bytes); pieces of this wave (s_src, s_len, s_dst: blocks): piece pc < 16 is the first piece of record pc, else
		// the overflow piece of record pc - 16; a wave stages pieces wave, wave + NWAVES, ...
		constexpr u32 PIECES = 32u / NWAVES;
		u32 s_src[PIECES], s_len[PIECES], s_dst[PIECES];
		auto step_vectors = [&](u32 Zs, u32 *bias) -> u32 {
			u32 ln = lane;
			MPC_OPAQUE(ln);
			const u32 li = ln & 15u;
			const u32 *tt = ttab + 64u * (Zs & 1u);
			const u32 R = tt[li], Oa = tt[16u + li], Ob = tt[32u + li];
			const u32 rt_first = rtab[8 * li], rt_rows = rtab[8 * li + 1], rt_c = rtab[8 * li + 2], rt_d = rtab[8 * li + 7];
			const u32 ovl = Ob - Oa + ((WIN && li >= MPC_RB_MAXN && rt_rows != 0u) ? 1u : 0u); // (value blocks: + the block the range ends in)
			const u32 incl = mpc_row16_scan_add(ovl); // the 16 records are the 16 lanes of a DPP row
			const u32 O = ftot + incl - ovl;
			const u32 src0 = R + rt_c + rt_first;
#pragma unroll
			for (u32 j = 0; j < PIECES; ++j) {
				const u32 pc = wave + j * NWAVES, rec = pc & 15u;
				const bool fp = pc < 16u; // wave-uniform
				s_src[j] = fp ? mpc_read_lane(src0, rec) : mpc_read_lane(Oa, rec);
				s_len[j] = fp ? mpc_read_lane(rt_rows, rec) : mpc_read_lane(ovl, rec);
				s_dst[j] = fp ? mpc_read_lane(rt_first, rec) : mpc_read_lane(O, rec);
			}
			// block record: hop bias (first-block address + bias + distance = the overflow block); window record: where its value area
			// would start in the buffer (+ the buffer's LDS address: stage_next), so that value d of the area is at bias + 4 d
			*bias = (O + R - Oa + rt_d) << 4;
			return ftot + mpc_read_lane(incl, 15u);
		};
		auto issue_dma = [&](u32 at) { // at: byte offset of the step's buffer in the staging area
			u32 ln = lane;
			MPC_OPAQUE(ln);
#pragma unroll
			for (u32 j = 0; j < PIECES; ++j) {
				const unsigned char *src = (WIN && ((wave + j * NWAVES) & 15u) >= MPC_RB_MAXN) ? (const unsigned char *)s.win : padb;
				for (u32 c0 = 0; c0 < s_len[j]; c0 += 64u)
					if (c0 + ln < s_len[j]) mpc_dma16(src + 16 * (u64)(s_src[j] + c0 + ln), stage + at + 16 * (s_dst[j] + c0));
			}
		};
		// window records: the Y lanes' biases become LDS addresses (the merge adds 4 * offset to them, nothing else)
		auto y_abs = [&](u32 bias, u32 at) -> u32 { u32 ln = lane; MPC_OPAQUE(ln); return (WIN && (ln & 15u) >= MPC_RB_MAXN) ? bias + lds_stage + at : bias; };
		// the biases of a step go through LDS as well (every wave computes the same 16 words and writes them to the same place)
		auto put_bias = [&](u32 Zs, u32 bias) { u32 ln = lane; MPC_OPAQUE(ln); if (ln < 16u) btab[16u * (Zs & 1u) + ln] = bias; };

		u32 cur_at = 0, cur_len, nxt_at = 0, nxt_len = 0;
		if (wave == 0u) {
			issue_table(0);
			if (n > 1) issue_table(1);
			mpc_dma_wait();
		}
		__syncthreads();
		{
			u32 b;
			cur_len = 16u * step_vectors(0, &b);
			put_bias(0, y_abs(b, 0u));
			issue_dma(0);
			mpc_dma_wait();
		}
		constexpr bool STAGING = DIAG < 2 || DIAG == 4; // DIAG 2, 3: step 0's records for every step
		bool pre = false;
		// vectors of step Z+1, its place (bottom and top of the staging area alternate), the prefetch when it fits beside step Z,
		// and — wave 0 — the table of step Z+2
		auto stage_next = [&](u32 Z) {
			pre = false;
			if (Z + 1 >= n) return;
			u32 b;
			nxt_len = 16u * step_vectors(Z + 1, &b);
			if (cur_at == 0u) { nxt_at = p.cap_bytes - nxt_len; pre = nxt_len <= p.cap_bytes && nxt_at >= cur_len; }
			else { nxt_at = 0u; pre = nxt_len <= cur_at; }
			put_bias(Z + 1, y_abs(b, pre ? nxt_at : 0u)); // (a step that is not prefetched is staged at the bottom, below)
			if (pre) issue_dma(nxt_at);
#ifdef MPC_RELAX_DIAG_BUILD
			if (tid == 0) atomicAdd(&p.tile_next[pre ? 9 : 8], 1u); // measurement build: steps prefetched / staged late
#endif
			if (wave == 0u && Z + 2 < n) issue_table(Z + 2);
		};
		unsigned long long tm_bar = 0, tm_wait = 0, tm_stage = 0, tm_all = 0, tm0 = 0; // DIAG 4: where a wave's time goes (100 MHz ticks)
		if (DIAG == 4) tm0 = mpc_clock();
		for (u32 Z = 0; Z < n; ++Z) {
			unsigned long long tq = 0;
			if (DIAG == 4) tq = mpc_clock();
			if (STAGING) __syncthreads(); // step Z's pieces and step Z+1's table have landed (every wave waited for its own DMA); step Z-1's readers are done
			else if (DIAG == 3 || Z == 0) __syncthreads();
			if (DIAG == 4) tm_bar += mpc_clock() - tq;
			bool staged = false;
			if (DIAG != 1 && nact != 0u) {
				const u32 sb = lds_stage + cur_at;
				u32 ln = lane;
				MPC_OPAQUE(ln);
				const u32 *bt = btab + 16u * ((STAGING ? Z : 0u) & 1u);
				const u32 bias_cur = bt[ln & 15u];                  // lane i: hop bias of record i at this step
				const u32 bias_y = mpc_row16_ror8(bias_cur);        // lane j < 8: hop bias of Y record j (record 8 + j: the row rotated by 8 lanes)
#pragma unroll
				for (int j = 0; j < (int)YREGS; ++j) MPC_OPAQUE(yreg[j]); // the 5-bit fields are unpacked per step (hoisted, they would be 12 more live registers)
#pragma unroll
				for (int j = 0; j < (MAXSLOTS + 9) / 10; ++j) MPC_OPAQUE_S(sel_a[j]); // likewise the scalar selectors
				MPC_OPAQUE_S(nact);
				BLOCKS blk;
				auto addr_a = [&](int q) -> u32 { return sb + (xy[q] & 0xffffu); };
				auto addr_b = [&](int q) -> u32 { return sb + (xy[q] >> 16); };
				auto idx_b = [&](int q) -> u32 { return (yreg[q / 6] >> (5 * (q % 6))) & 31u; }; // 4 * (lane of bias_y that holds the cell's Y record)
				u32 nia = addr_a(0), nib = addr_b(0);
				blk.load(nia, nib, mpc_lds_addr(bt + MPC_RB_MAXN) + idx_b(0)); // (the first slot's Y bias straight from the table: one round of LDS reads opens the step)
				auto slot = [&](auto &&self, auto qc) __attribute__((always_inline)) {
					constexpr int q = decltype(qc)::value;
					if constexpr (q < MAXSLOTS) {
						if ((u32)q >= nact) return; // wave-uniform
						const u32 ia = nia + mpc_read_lane(bias_cur, (sel_a[q / 10] >> (3 * (q % 10))) & 7u), ib = nib; // X record: scalar bias
						constexpr int qn = q + 1 < MAXSLOTS ? q + 1 : q;
						nia = addr_a(qn); nib = addr_b(qn);
						float sum = acc[q];
						blk.template merge<q & 1>(sum, ia, ib, nia, nib, idx_b(qn), bias_y);
						acc[q] = sum;
						// the next step's staging goes out after the first slot: its chain of table reads and scalar work then runs beside the
						// other waves' merges instead of holding every wave of the workgroup at the top of the step
						if (STAGING && q == 0) {
							unsigned long long ts = 0;
							if (DIAG == 4) ts = mpc_clock();
							stage_next(Z); staged = true;
							if (DIAG == 4) tm_stage += mpc_clock() - ts;
						}
						self(self, std::integral_constant<int, q + 1>{});
					}
				};
				slot(slot, std::integral_constant<int, 0>{});
				blk.drain(); // the last slot's look-ahead reads have landed before their registers mean anything else
			}
			if (STAGING && !staged) stage_next(Z);
			if (STAGING && Z + 1 < n) {
				if (!pre) { // the next step did not fit beside this one: stage it now that this one's readers are done
					__syncthreads();
					u32 b;
					nxt_len = 16u * step_vectors(Z + 1, &b);
					nxt_at = 0u;
					issue_dma(0u);
				}
				unsigned long long tw = 0;
				if (DIAG == 4) tw = mpc_clock();
				mpc_dma_wait();
				if (DIAG == 4) tm_wait += mpc_clock() - tw;
				cur_at = nxt_at; cur_len = nxt_len;
			}
		}
		if (DIAG == 4 && lane == 0u) { // per wave number, summed over workgroups and tiles: walk, barrier, staging block, DMA wait
			tm_all = mpc_clock() - tm0;
			unsigned long long *tt = (unsigned long long *)(p.tile_next + 16) + 4 * wave;
			atomicAdd(tt + 0, tm_all); atomicAdd(tt + 1, tm_bar); atomicAdd(tt + 2, tm_stage); atomicAdd(tt + 3, tm_wait);
		}
		// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
		const auto pk = MPC_KERNARG_AGAIN(p);
		row_geometry(pk);
		if (by_rows) {
			__syncthreads(); // the last step's readers are done with the staging area
			build_rows(pk);
		}
#pragma unroll
		for (int q = 0; q < MAXSLOTS; ++q) {
			MPC_SCHED_BARRIER();
			if ((u32)q < nact) {
				u32 k, e, ix, iy, rw;
				if (find_cell((u32)q, &k, &e, &ix, &iy, &rw))
					pk->s.vnext[pk->s.vbase[k] + e] = acc[q] / (float)pk->s.n; // uint -> float, IEEE divide (mysparsemx.cpp:108)
			}
		}
	}
}
