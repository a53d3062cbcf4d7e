// kernels_fb.h — pair-HMM forward + backward + posterior thresholding, one wavefront per pair.
//
// Replaces CalcFwdFlat (fwdflat3.cpp:12-153), CalcBwdFlat (bwdflat3.cpp:10-184),
// CalcTotalProbFlat (totalprobflat.cpp:3-16) and the Score/threshold half of CalcPostFlat
// (calcposteriorflat.cpp:9-26) of the reference; the exp/sort/sparsify half is kernels_post.h.
//
// Mapping (CDNA4, 64-wide wavefront): lane t owns H consecutive rows of X (i = t*H+1 .. t*H+H,
// H = ceil(LX/64) <= MPC_HMAX) and sweeps the columns of Y as a systolic array: at step s lane t
// computes column j = s - t, so the three DP neighbours (i-1,j-1), (i-1,j), (i,j-1) are either in
// the lane's own registers or in lane t-1's registers from the previous step (one wave shift of
// 5 floats + the Y letter per step, as DPP wave_shr/wave_shl moves). All 5H state values live in VGPRs; LDS only holds the
// compacted emission tables (A*A match + A insert scores). Only the forward M plane ever touches
// HBM: it is written step-major [(step*H + r)*64 + lane] so both the forward store and the
// backward load (which visits the same (lane,row,column) at forward-step index j+t, uniform over
// the wave) are fully coalesced 256-byte rows. The backward sweep mirrors the schedule (lane T-1
// leads, data flows t+1 -> t) and emits every cell with Score >= MIN_SPARSE_SCORE as an 8-byte
// candidate {flat index, score} through a wave-aggregated append.
//
// Structure-profile ("mega") emissions, MEGA = true: when a .mega input is loaded the reference's CalcPost
// runs Mega::CalcFwdFlat_mega / CalcBwdFlat_mega (calcpost.cpp:14-22; fwdflat_mega.cpp:14-165,
// bwdflat_mega.cpp:13-193) — the same recurrences with Emit = Mega::GetInsScore (mega.cpp:273-285) /
// Mega::GetMatchScore (mega.cpp:341-363): a left fold, from 0, over the features f of
// Table_f[letter(s)] * Weight_f. Here the products are formed once (mega_prepare_kernel: the fold of the
// insert score per sequence position, and the pair tables pre-multiplied by their weight — each product
// is one rounded multiply either way), every position carries its <= 8 feature letters packed in a
// u64, and a cell's match score is the in-order sum of <= 8 LDS lookups.
//
// Row blocks, LONG = true (X longer than 64*MPC_HMAX rows — and, by the host's choice, from 769 rows on, where a single block
// would need 13+ rows per lane and 177+ VGPRs): the rows are cut into blocks of 64*H; the
// forward sweep runs block after block, lane 63's last row (all five states, every column) going through
// a small HBM line buffer to lane 0 of the next block, which reads it back 64 columns at a time (one
// coalesced load per state every 64 steps + one v_readlane per step). The backward sweep runs the blocks
// in reverse with lane 0's first row (M, IX, JX) as the line buffer for lane 63 of the block above. The
// forward M plane keeps one region per block. Cells, expressions and results are the same; candidate keys
// use 16 bits for the column (MPC_KEY_ROW_SHIFT_LONG). The line buffers alternate by block parity and a
// workgroup-scope fence separates a block's stores from the next block's loads.
//
// Every cell value is a fixed expression of its three neighbours, so the wavefront order does
// not change results: outputs are bit-identical to the row-major CPU sweep (given no FMA
// contraction, -ffp-contract=off). Border handling: the reference's special-cased border
// formulas are reproduced by the generic recurrences over virtual LOG_ZERO neighbours
// (LOG_ZERO + score == LOG_ZERO exactly, LOG_ADD(LOG_ZERO, v) == v) plus the few genuinely special
// cells, each cited below.
#pragma once
#include "device_math.h"

#define MPC_HMAX 16
#define MPC_KEY_ROW_SHIFT 22 // rows < 64*MPC_HMAX = 2^10, columns < 2^22 (checked in mpcgpu_set_seqs)
#define MPC_KEY_COL_MASK ((1u << MPC_KEY_ROW_SHIFT) - 1u)
#define MPC_KEY_ROW_SHIFT_LONG 16 // row-block (LONG) kernels: rows and columns < 2^16 (checked by the host)
#define MPC_KEY_COL_MASK_LONG ((1u << MPC_KEY_ROW_SHIFT_LONG) - 1u)
#define MPC_FB_COEF_BYTES (MPC_COEF_ENTRIES * 16) // static LDS of fb_kernel (the host's occupancy / capacity arithmetic adds it)
#define MPC_MEGA_FMAX 8 // features per position (one byte each in a u64)

struct FbParams {
	// sequences (compact alphabet codes 0..A-1)
	const u8 *seq_code;
	const u64 *seq_off;
	const u32 *seq_len;
	// PairHMM constants (hmmscores.h:1-16 names)
	float tSM, tSI, tSJ, tMM, tMI, tMJ, tII, tIM, tJJ, tJM;
	float thr; // MIN_SPARSE_SCORE
	int A;     // alphabet size
	const float *match; // A*A
	const float *ins;   // A
	// work list of this launch
	const u32 *pair_x, *pair_y; // per batch-local pair: sequence indices
	const u32 *order;           // batch-local pair ids handled by this launch (same H for all)
	u32 count;
	u32 *queue; // work-queue head (zeroed before launch)
	// scratch / outputs
	float *fm_scratch; // forward M plane, one slot per resident wave
	u64 fm_stride;     // floats per slot
	u64 *cand;         // candidates, capc per batch-local pair: (cell key << 32) | score bits,
	                   // cell key = (i-1) << MPC_KEY_ROW_SHIFT | (j-1): row-major order, no division to unpack
	u32 capc;
	u32 *cand_cnt; // per batch-local pair (may exceed capc: overflow, detected by the host)
	float *total;  // per batch-local pair: log total probability (diagnostic / tests)
	// row blocks (LONG kernels only)
	float *bnd;      // line buffers, one slot per resident wave: [2 parities][5 fwd + 3 bwd states][bnd_ld]
	u64 bnd_stride;  // floats per slot
	u32 bnd_ld;      // >= LYmax + 2
	u64 fm_block;    // floats of fm_scratch per row block
	// structure-profile emissions (MEGA kernels only); positions are indexed like seq_code
	const u64 *mg_prof;  // letters of feature f in bits [8f, 8f+8)
	const float *mg_ins; // Mega::GetInsScore of the position
	const float *mg_tab; // weight-multiplied pair tables back to back + one 0.0f (what unused features read)
	u32 mg_tab_floats;   // including the trailing zero
	u32 mg_base[MPC_MEGA_FMAX];  // first float of feature f's table
	u32 mg_alpha[MPC_MEGA_FMAX]; // its row length (0 for unused features: every lookup hits the zero)
};

// Mega::GetMatchScore (mega.cpp:341-363): Score = 0; Score += LogProbMx_f[lx][ly] * Weight_f for f ascending.
// yi[f] = mg_base[f] + letter_f(y). Unused features add the table's 0.0f, which leaves the sum unchanged
// (it can never be -0.0f: it starts at +0.0f).
__device__ __forceinline__ float mpc_mega_match(const float *s_tab, const u32 *alpha, u64 xl, const u32 *yi)
{
	float m = 0.0f;
#pragma unroll
	for (int f = 0; f < MPC_MEGA_FMAX; ++f) {
		const u32 xf = (u32)(xl >> (8 * f)) & 0xffu;
		m += s_tab[xf * alpha[f] + yi[f]];
	}
	return m;
}

// With 8 rows per lane the kernel is held to 128 VGPRs (4 waves per SIMD: the second launch-bounds argument; it needs 129
// without it, and 3 waves per SIMD cost 19 %): 400 x L~480 fb 156 -> 139 ms. Fewer rows fit anyway (7 rows: 117;
// given the bound the compiler schedules them differently and the L~400 headline loses 1.7 %, so they are left alone).
template <int H, bool MEGA, bool LONG>
__global__ void __launch_bounds__(256, (H == 8 && !MEGA && !LONG) ? 4 : 1) fb_kernel(FbParams p)
{
	MPC_DYN_SMEM(smem_raw);
	// LOGEXP1 coefficient table: statically allocated, so its LDS address is a compile-time constant that rides in the
	// offset field of every ds_read_b128 (no per-LOG_ADD base add); the dynamic part below follows it
	__shared__ MpcCoef s_coef[MPC_COEF_ENTRIES];
	float *s_match = (float *)smem_raw;                          // A*A (MEGA: the feature tables)
	float *s_ins = s_match + p.A * p.A;                          // A   (MEGA: unused)
	if (threadIdx.x < MPC_COEF_ENTRIES)
		mpc_coef_table_init(s_coef, (int)threadIdx.x);
	if (MEGA) {
		for (u32 q = threadIdx.x; q < p.mg_tab_floats; q += blockDim.x)
			s_match[q] = p.mg_tab[q];
	} else {
		for (int q = threadIdx.x; q < p.A * p.A; q += blockDim.x)
			s_match[q] = p.match[q];
		for (int q = threadIdx.x; q < p.A; q += blockDim.x)
			s_ins[q] = p.ins[q];
	}
	__syncthreads();

	const int t = threadIdx.x & 63;
	const u32 waves_per_block = blockDim.x >> 6;
	const u32 slot = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
	float *fm = p.fm_scratch + (u64)slot * p.fm_stride;
	const float LZ = MPC_LOG_ZERO;
	const float tSM = p.tSM, tSI = p.tSI, tSJ = p.tSJ, tMM = p.tMM, tMI = p.tMI, tMJ = p.tMJ;
	const float tII = p.tII, tIM = p.tIM, tJJ = p.tJJ, tJM = p.tJM;
	const int A = p.A;
	u32 mg_base[MPC_MEGA_FMAX], mg_alpha[MPC_MEGA_FMAX]; // wave-uniform (SGPRs)
#pragma unroll
	for (int f = 0; f < MPC_MEGA_FMAX; ++f) { mg_base[f] = MEGA ? p.mg_base[f] : 0u; mg_alpha[f] = MEGA ? p.mg_alpha[f] : 0u; }

	for (;;) {
		// Work queue: lane 0 adds 1, the other lanes add 0 (branch-free; the compiler's atomic optimizer
		// folds the wave into one atomic), and lane 0's return value — the grab index — is broadcast
		// through readfirstlane so that it and everything derived from it (pair id, lengths, trip
		// counts) is wave-uniform in SGPRs and the loop exit is a scalar branch. Do NOT rewrite this as
		// `if (lane == 0) atomicAdd` + shuffle: a lane-dependent branch at the head of a loop that
		// contains wave shuffles gets rotated into two back edges and the lanes lose reconvergence
		// (observed on gfx950: lanes 1..63 spin forever).
		const u32 qi = mpc_wave_first(atomicAdd(p.queue, t == 0 ? 1u : 0u));
		if (qi >= p.count)
			break;
		const u32 pid = p.order[qi];
		const u32 sx = p.pair_x[pid], sy = p.pair_y[pid];
		const int LX = (int)p.seq_len[sx], LY = (int)p.seq_len[sy];
		const u8 *X = p.seq_code + p.seq_off[sx];
		const u8 *Y = p.seq_code + p.seq_off[sy];
		constexpr int R = 64 * H;                   // rows per block
		const int NB = LONG ? (LX + R - 1) / R : 1; // row blocks (1 unless LONG)
		float *bnd = LONG ? p.bnd + (u64)slot * p.bnd_stride : nullptr;
		const u32 ld = LONG ? p.bnd_ld : 0u;
		int T = (LX + H - 1) / H; // lanes that own at least one row (LONG: of the current block)

		// ------------------------------------------------------------------ forward
		float cM[H], cIX[H], cJX[H], cIY[H], cJY[H]; // own rows at the previous column
		float insx[H];
		int mrow[H];
		u64 xl[MEGA ? H : 1];
		// MEGA: packed letters and insert score of every position
		const u64 *PX = MEGA ? p.mg_prof + p.seq_off[sx] : nullptr, *PY = MEGA ? p.mg_prof + p.seq_off[sy] : nullptr;
		const float *IX = MEGA ? p.mg_ins + p.seq_off[sx] : nullptr, *IY = MEGA ? p.mg_ins + p.seq_off[sy] : nullptr;
		u32 ylo_prev = 0, yhi_prev = 0; // MEGA: the column's packed letters and insert score travel with it
		float insy_prev = 0.0f;
		for (int b = 0; b < NB; ++b) { // row block b: rows i0+1 .. i0+R (one pass unless LONG)
		const int i0 = LONG ? b * R : 0;
		if (LONG) T = ((LX - i0 < R ? LX - i0 : R) + H - 1) / H;
		float *fmb = LONG ? fm + (u64)b * p.fm_block : fm;
		const float *bnd_in = LONG ? bnd + (u64)(b & 1) * 8 * ld : nullptr; // row i0, written by block b-1
		float *bnd_out = LONG ? bnd + (u64)((b + 1) & 1) * 8 * ld : nullptr; // row i0+R, for block b+1
		float pfM = LZ, pfIX = LZ, pfJX = LZ, pfIY = LZ, pfJY = LZ;          // LONG: 64 columns of row i0, one per lane
#pragma unroll
		for (int r = 0; r < H; ++r) {
			const int i = i0 + t * H + r + 1;
			if (MEGA) {
				xl[MEGA ? r : 0] = (i <= LX) ? PX[i - 1] : 0ull;
				insx[r] = (i <= LX) ? IX[i - 1] : 0.0f; // fwdflat_mega.cpp:113
			} else {
				const int xc = (i <= LX) ? (int)X[i - 1] : 0;
				insx[r] = s_ins[xc];
				mrow[r] = xc * A;
			}
			cM[r] = cIX[r] = cJX[r] = cIY[r] = cJY[r] = LZ;
		}
		float uM = LZ, uIX = LZ, uJX = LZ, uIY = LZ, uJY = LZ; // row t*H at column j-1 (diagonal of r=0)
		float gIY = LZ, gJY = LZ;                              // lane 0: row-0 chain (fwdflat3.cpp:81-93)
		int yprev = 0;
		ylo_prev = 0; yhi_prev = 0; insy_prev = 0.0f;
		const int nsteps = LY + T;
		for (int s = 0; s < nsteps; ++s) {
			const int j = s - t;
			// row t*H at column j comes from lane t-1's last row of the previous step
			float nM = mpc_lane_up1(cM[H - 1]);
			float nIX = mpc_lane_up1(cIX[H - 1]);
			float nJX = mpc_lane_up1(cJX[H - 1]);
			float nIY = mpc_lane_up1(cIY[H - 1]);
			float nJY = mpc_lane_up1(cJY[H - 1]);
			int yc = 0;
			u32 ylo = 0, yhi = 0;
			float insy;
			u32 yi[MEGA ? MPC_MEGA_FMAX : 1];
			if (MEGA) {
				ylo = (u32)mpc_lane_up1((int)ylo_prev);
				yhi = (u32)mpc_lane_up1((int)yhi_prev);
				insy = mpc_lane_up1(insy_prev);
				const bool incol = (s >= 1 && s <= LY);
				const u64 yload = incol ? PY[s - 1] : 0ull; // lane 0: column j = s
				const float iload = incol ? IY[s - 1] : 0.0f; // fwdflat_mega.cpp:120
				if (t == 0) { ylo = (u32)yload; yhi = (u32)(yload >> 32); insy = iload; }
#pragma unroll
				for (int f = 0; f < MPC_MEGA_FMAX; ++f)
					yi[f] = mg_base[f] + (((f < 4 ? ylo : yhi) >> (8 * (f & 3))) & 0xffu);
			} else {
				yc = mpc_lane_up1(yprev);
				const int yload = (s >= 1 && s <= LY) ? (int)Y[s - 1] : 0; // lane 0: letter of column j = s
				if (t == 0)
					yc = yload;
				insy = s_ins[yc];
			}
			if (LONG && b > 0) {
				// row i0, the last row of the block above, comes back from the line buffer
				if ((s & 63) == 0) {
					const int jj = s + t;
					const bool in = jj <= LY;
					pfM = in ? bnd_in[0 * ld + jj] : LZ; pfIX = in ? bnd_in[1 * ld + jj] : LZ; pfJX = in ? bnd_in[2 * ld + jj] : LZ;
					pfIY = in ? bnd_in[3 * ld + jj] : LZ; pfJY = in ? bnd_in[4 * ld + jj] : LZ;
				}
				const float bM = mpc_read_lane(pfM, s & 63), bIX = mpc_read_lane(pfIX, s & 63), bJX = mpc_read_lane(pfJX, s & 63);
				const float bIY = mpc_read_lane(pfIY, s & 63), bJY = mpc_read_lane(pfJY, s & 63);
				if (t == 0) { nM = bM; nIX = bIX; nJX = bJX; nIY = bIY; nJY = bJY; }
			} else
			if (t == 0) {
				// row 0 (fwdflat3.cpp:35-39, :44-45, :57-65, :81-93): M=IX=JX=LOG_ZERO,
				// IY(0,1)=tSI+Ins(y1), IY(0,j)=IY(0,j-1)+tII+Ins(yj)
				nM = LZ; nIX = LZ; nJX = LZ;
				if (j <= 0) { nIY = LZ; nJY = LZ; }
				else if (j == 1) { nIY = tSI + insy; nJY = tSJ + insy; }
				else { nIY = gIY + tII + insy; nJY = gJY + tJJ + insy; }
				gIY = nIY; gJY = nJY;
			}
			float dM = uM, dIX = uIX, dJX = uJX, dIY = uIY, dJY = uJY; // (i-1, j-1)
			float upM = nM, upIX = nIX, upJX = nJX;                     // (i-1, j)
			float *fmrow = fmb + ((u64)s * H) * 64 + t;
#pragma unroll
			for (int r = 0; r < H; ++r) {
				const float oM = cM[r], oIX = cIX[r], oJX = cJX[r], oIY = cIY[r], oJY = cJY[r]; // (i, j-1)
				const float m = MEGA ? mpc_mega_match(s_match, mg_alpha, xl[MEGA ? r : 0], yi) // fwdflat_mega.cpp:121
				                     : s_match[mrow[r] + yc];
				// fwdflat3.cpp:116-145. No per-cell border tests: for j <= 0 (column 0 and the columns a
				// lane "computes" before it has started) every input that should be LOG_ZERO is exactly
				// LOG_ZERO, LOG_ZERO + score == LOG_ZERO and LOG_ADD(LOG_ZERO, v) == v, so the same
				// expressions give M = IY = JY = LOG_ZERO and the column-0 chains
				// IX(i,0) = IX(i-1,0) + tII + Ins(x_i), JX likewise (fwdflat3.cpp:48-55, :67-79). The only
				// genuinely special cells are in row 1 (lane 0, r == 0), below. Rows past LX and columns
				// past LY compute garbage that nothing reads.
				float vM = mpc_la5t(dM + tMM, dIX + tIM, dJX + tJM, dIY + tIM, dJY + tJM, s_coef) + m;
				float vIX = mpc_la2t(upIX + tII, upM + tMI, s_coef) + insx[r];
				float vJX = mpc_la2t(upJX + tJJ, upM + tMJ, s_coef) + insx[r];
				float vIY = mpc_la2t(oIY + tII, oM + tMI, s_coef) + insy;
				float vJY = mpc_la2t(oJY + tJJ, oM + tMJ, s_coef) + insy;
				if (r == 0) {
					const bool row1 = (t == 0) && (!LONG || b == 0);
					if (row1 && j == 0) { vIX = tSI + insx[0]; vJX = tSJ + insx[0]; } // fwdflat3.cpp:42-43
					if (row1 && j == 1) vM = tSM + m;                                 // fwdflat3.cpp:111-112
				}
				cM[r] = vM; cIX[r] = vIX; cJX[r] = vJX; cIY[r] = vIY; cJY[r] = vJY;
				fmrow[r * 64] = vM;
				dM = oM; dIX = oIX; dJX = oJX; dIY = oIY; dJY = oJY;
				upM = vM; upIX = vIX; upJX = vJX;
			}
			uM = nM; uIX = nIX; uJX = nJX; uIY = nIY; uJY = nJY;
			yprev = yc;
			ylo_prev = ylo; yhi_prev = yhi; insy_prev = insy;
			if (LONG && b + 1 < NB && t == 63 && j >= 0 && j <= LY) { // row i0+R for the next block (column j = s-63)
				bnd_out[0 * ld + j] = cM[H - 1]; bnd_out[1 * ld + j] = cIX[H - 1]; bnd_out[2 * ld + j] = cJX[H - 1];
				bnd_out[3 * ld + j] = cIY[H - 1]; bnd_out[4 * ld + j] = cJY[H - 1];
			}
		}
		if (LONG) MPC_WAVE_FENCE();
		} // row blocks (forward)
		// F(LX,LY,*) sits in lane T-1, row (LX-1)%H, after its last step (column LY).
		float eM = LZ, eIX = LZ, eJX = LZ, eIY = LZ, eJY = LZ;
		{
			const int rl = (LX - 1) % H;
#pragma unroll
			for (int r = 0; r < H; ++r)
				if (r == rl) { eM = cM[r]; eIX = cIX[r]; eJX = cJX[r]; eIY = cIY[r]; eJY = cJY[r]; }
			eM = __shfl(eM, T - 1); eIX = __shfl(eIX, T - 1); eJX = __shfl(eJX, T - 1);
			eIY = __shfl(eIY, T - 1); eJY = __shfl(eJY, T - 1);
		}
		// totalprobflat.cpp:3-16 with B(LX,LY,*) = start scores (bwdflat3.cpp:53-61); state order
		// M, IX, IY, JX, JY (pairhmm.h:11-19), left fold from LOG_ZERO.
		float total = LZ;
		total = mpc_la2t(total, eM + tSM, s_coef);
		total = mpc_la2t(total, eIX + tSI, s_coef);
		total = mpc_la2t(total, eIY + tSI, s_coef);
		total = mpc_la2t(total, eJX + tSJ, s_coef);
		total = mpc_la2t(total, eJY + tSJ, s_coef);
		if (t == 0)
			p.total[pid] = total;

		// ------------------------------------------------------------------ backward + posterior
		// Row i uses the emissions of x_{i+1}=X[i] and y_{j+1}=Y[j] (bwdflat3.cpp:46,64).
		u64 *cand = p.cand + (u64)pid * p.capc;
		u32 ncand = 0;
		for (int bb = 0; bb < NB; ++bb) { // row blocks bottom-up (one pass unless LONG)
		const int b = NB - 1 - bb;
		const int i0 = LONG ? b * R : 0;
		if (LONG) T = ((LX - i0 < R ? LX - i0 : R) + H - 1) / H;
		const float *fmb = LONG ? fm + (u64)b * p.fm_block : fm;
		const float *bnd_in = LONG ? bnd + ((u64)(bb & 1) * 8 + 5) * ld : nullptr; // row i0+R+1 (M, IX, JX), written by block b+1
		float *bnd_out = LONG ? bnd + ((u64)((bb + 1) & 1) * 8 + 5) * ld : nullptr; // row i0+1, for block b-1
		float pfM = LZ, pfIX = LZ, pfJX = LZ; // LONG: 64 columns of the row below the block, one per lane
#pragma unroll
		for (int r = 0; r < H; ++r) {
			const int i = i0 + t * H + r + 1;
			if (MEGA) {
				xl[MEGA ? r : 0] = (i < LX) ? PX[i] : 0ull;
				insx[r] = (i < LX) ? IX[i] : 0.0f; // bwdflat_mega.cpp:55
			} else {
				const int xc = (i < LX) ? (int)X[i] : 0;
				insx[r] = s_ins[xc];
				mrow[r] = xc * A;
			}
			cM[r] = cIX[r] = cJX[r] = cIY[r] = cJY[r] = LZ; // virtual column LY+1
		}
		float gM = LZ; // row (t+1)*H+1 at column j+1: diagonal of r=H-1
		int ynext_prev = 0;
		ylo_prev = 0; yhi_prev = 0; insy_prev = 0.0f;
		const int bsteps = LY + T - 1;
		for (int s = 0; s < bsteps; ++s) {
			const int j = LY - s + (T - 1 - t);
			// row (t+1)*H+1 at column j: lane t+1's first row from the previous step
			float nM = mpc_lane_down1(cM[0]);
			float nIX = mpc_lane_down1(cIX[0]);
			float nJX = mpc_lane_down1(cJX[0]);
			if (LONG && bb > 0) {
				// the first row of the block below comes back from the line buffer (lane 63 is at column LY - s)
				if ((s & 63) == 0) {
					const int jj = LY - s - t;
					const bool in = jj >= 1;
					pfM = in ? bnd_in[0 * ld + jj] : LZ; pfIX = in ? bnd_in[1 * ld + jj] : LZ; pfJX = in ? bnd_in[2 * ld + jj] : LZ;
				}
				const float bM = mpc_read_lane(pfM, s & 63), bIX = mpc_read_lane(pfIX, s & 63), bJX = mpc_read_lane(pfJX, s & 63);
				if (t == 63) { nM = bM; nIX = bIX; nJX = bJX; }
			} else
			if (t == 63) { nM = LZ; nIX = LZ; nJX = LZ; } // nothing below the wave: virtual row
			const int jl = LY - s; // column of the leading lane T-1
			int yc = 0;
			u32 ylo = 0, yhi = 0;
			float insy;
			u32 yi[MEGA ? MPC_MEGA_FMAX : 1];
			if (MEGA) {
				ylo = (u32)mpc_lane_down1((int)ylo_prev);
				yhi = (u32)mpc_lane_down1((int)yhi_prev);
				insy = mpc_lane_down1(insy_prev);
				const bool incol = (jl >= 0 && jl < LY);
				const u64 yload = incol ? PY[jl] : 0ull;      // y_{j+1} of the leading lane's column
				const float iload = incol ? IY[jl] : 0.0f;    // bwdflat_mega.cpp:78
				if (t >= T - 1) { ylo = (u32)yload; yhi = (u32)(yload >> 32); insy = iload; }
#pragma unroll
				for (int f = 0; f < MPC_MEGA_FMAX; ++f)
					yi[f] = mg_base[f] + (((f < 4 ? ylo : yhi) >> (8 * (f & 3))) & 0xffu);
			} else {
				yc = mpc_lane_down1(ynext_prev);
				const int yload = (jl >= 0 && jl < LY) ? (int)Y[jl] : 0;
				if (t >= T - 1)
					yc = yload; // leading lane (and idle lanes beyond it)
				insy = s_ins[yc];
			}
			const int sf = j + t; // forward step that stored column j of this lane (uniform: LY-s+T-1)
			const float *fmrow = fmb + ((u64)(sf < 0 ? 0 : sf) * H) * 64 + t;
			float dgM = gM;                           // M(i+1, j+1)
			float dnIX = nIX, dnJX = nJX;             // (i+1, j)
			bool anyhit = false;
			float sc[H];
#pragma unroll
			for (int r = H - 1; r >= 0; --r) {
				const int i = i0 + t * H + r + 1;
				const float oM = cM[r], oIY = cIY[r], oJY = cJY[r]; // (i, j+1)
				// bwdflat3.cpp:75-79
				const float xM = dgM + (MEGA ? mpc_mega_match(s_match, mg_alpha, xl[MEGA ? r : 0], yi) // bwdflat_mega.cpp:79-80
				                             : s_match[mrow[r] + yc]);
				const float xIX = dnIX + insx[r];
				const float xJX = dnJX + insx[r];
				const float xIY = oIY + insy;
				const float xJY = oJY + insy;
				// bwdflat3.cpp:81-118 (interior); the right column (:132-153) and bottom row (:155-176)
				// formulas fall out of the same expressions over LOG_ZERO virtual neighbours.
				float vM = mpc_la5t(tMM + xM, tMI + xIX, tMJ + xJX, tMI + xIY, tMJ + xJY, s_coef);
				float vIX = mpc_la2t(tII + xIX, tIM + xM, s_coef);
				float vJX = mpc_la2t(tJJ + xJX, tJM + xM, s_coef);
				float vIY = mpc_la2t(tII + xIY, tIM + xM, s_coef);
				float vJY = mpc_la2t(tJJ + xJY, tJM + xM, s_coef);
				if (i == LX && j == LY) { // bwdflat3.cpp:53-61
					vM = tSM; vIX = tSI; vIY = tSI; vJX = tSJ; vJY = tSJ;
				}
				// No "outside the matrix" masking: rows below LX and columns right of LY start as
				// LOG_ZERO and only ever combine LOG_ZERO inputs, so they stay exactly LOG_ZERO (the
				// virtual neighbours the border formulas bwdflat3.cpp:132-176 need); column 0 and beyond
				// is never read back (posteriors use i,j >= 1).
				// calcposteriorflat.cpp:14: Score = F_M + B_M - Total
				const float f = fmrow[r * 64];
				const float score = (f + vM) - total;
				sc[r] = score;
				anyhit = anyhit || ((i <= LX) && (j >= 1) && (j <= LY) && score >= p.thr);
				dgM = oM; // becomes M(i, j+1) = diagonal of row i-1
				cM[r] = vM; cIX[r] = vIX; cJX[r] = vJX; cIY[r] = vIY; cJY[r] = vJY;
				dnIX = vIX; dnJX = vJX;
			}
			gM = nM;
			ynext_prev = yc;
			ylo_prev = ylo; yhi_prev = yhi; insy_prev = insy;
			if (LONG && b > 0 && t == 0 && j >= 1 && j <= LY) { // row i0+1 for the block above
				bnd_out[0 * ld + j] = cM[0]; bnd_out[1 * ld + j] = cIX[0]; bnd_out[2 * ld + j] = cJX[0];
			}
			if (__ballot(anyhit)) {
#pragma unroll
				for (int r = 0; r < H; ++r) {
					const int i = i0 + t * H + r + 1;
					const bool hit = (i <= LX) && (j >= 1) && (j <= LY) && (sc[r] >= p.thr);
					const u64 bal = __ballot(hit);
					if (bal) {
						const u32 pos = ncand + (u32)__popcll(bal & ((1ull << t) - 1ull));
						if (hit && pos < p.capc) {
							const u32 idx = ((u32)(i - 1) << (LONG ? MPC_KEY_ROW_SHIFT_LONG : MPC_KEY_ROW_SHIFT)) | (u32)(j - 1);
							cand[pos] = ((u64)idx << 32) | (u64)__float_as_uint(sc[r]);
						}
						ncand += (u32)__popcll(bal);
					}
				}
			}
		}
		if (LONG) MPC_WAVE_FENCE();
		} // row blocks (backward)
		if (t == 0)
			p.cand_cnt[pid] = ncand;
	}
}

// ---- preparation of the structure-profile tables (once per mpcgpu_set_mega) ---------------------------------
struct MegaPrepParams {
	u32 nfeat;
	const u32 *alpha;    // [nfeat]
	const float *weight; // [nfeat]
	const float *lp;     // log-probabilities per feature, back to back
	const u32 *lp_off;   // [nfeat]
	const float *mx;     // A_f x A_f log-probability matrices, back to back
	const u32 *mx_off;   // [nfeat + 1]
	const u8 *letters;   // all positions of all sequences, nfeat letters per position
	u64 npos;
	u64 *prof;           // out: packed letters per position
	float *ins;          // out: Mega::GetInsScore per position
	float *tab;          // out: mx[q] * weight[f(q)], then one 0.0f
};

__global__ void __launch_bounds__(256) mega_prepare_kernel(MegaPrepParams p)
{
	const u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x, stride = (u64)gridDim.x * blockDim.x;
	const u32 ntab = p.mx_off[p.nfeat];
	for (u64 q = gid; q <= ntab; q += stride) {
		if (q == ntab) { p.tab[q] = 0.0f; continue; }
		u32 f = 0;
		while (f + 1 < p.nfeat && q >= p.mx_off[f + 1]) ++f;
		p.tab[q] = p.mx[q] * p.weight[f]; // mega.cpp:359-360 (the product; the fold is mpc_mega_match)
	}
	for (u64 pos = gid; pos < p.npos; pos += stride) {
		const u8 *l = p.letters + pos * p.nfeat;
		u64 packed = 0;
		float score = 0.0f; // mega.cpp:277-284
		for (u32 f = 0; f < p.nfeat; ++f) {
			packed |= (u64)l[f] << (8 * f);
			score += p.lp[p.lp_off[f] + l[f]] * p.weight[f];
		}
		p.prof[pos] = packed;
		p.ins[pos] = score;
	}
}
