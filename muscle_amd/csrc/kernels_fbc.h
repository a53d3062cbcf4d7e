// kernels_fbc.h — fb_chain_kernel: the forward + backward + threshold sweep of kernels_fb.h for a CHAIN of pairs that share
// their row sequence X (in the all-pairs order of mpcflat.cpp:145-155 the pairs (i, i+1), (i, i+2), ... do), without the
// systolic fill and drain between the pairs of the chain.
//
// fb_kernel spends LY + T steps per sweep where only LY columns carry work (T = lanes that own rows: 58 of 458 steps at
// L ~ 400): lane t idles t steps before its first column and T-1-t after its last. Here the Y sequences of the chain are laid
// on ONE virtual column axis V = 0 .. Vtot-1, pair k at V = base_k + j (j = 0 .. LY_k: column 0 included), base_{k+1} =
// base_k + LY_k + 1, and the wave sweeps the whole axis once forward (lane t at V = s - t) and once backward (lane t at
// V = Vtot-1 - s + (T-1-t)): lane t starts pair k+1 the step after it finished pair k. X, and with it the rows a lane owns,
// its emission rows and T, are the same for every pair of the chain; what changes per pair travels with the column (the Y
// letter, shifted from lane to lane as before) or is wave-uniform bookkeeping of the FIRST and LAST lane (which pair they are
// in, the forward totals). The only per-lane events:
//   * forward, a lane reaching V = base_k (column 0 of pair k >= 1): its state registers still hold the previous pair's last
//     column; they are set to LOG_ZERO first — the virtual column "-1" every cell of column 0 is computed from in fb_kernel
//     (kernels_fb.h:260-266), so column 0 and everything after it come out as they do there;
//   * backward, the same V is column 0 of pair k (computed and never read, as in fb_kernel) and stands where pair k-1 needs
//     its virtual column LY+1 = LOG_ZERO (kernels_fb.h:336): the registers are set to LOG_ZERO AFTER that step.
// Columns are at least T apart from one boundary to the next (the host only chains pairs with LY + 1 >= T), so at most one
// boundary is inside the wave at any step, known as a scalar; a cell's pair is "the one above or the one below the boundary".
// The forward M plane is stored by step as before (one plane of Vtot + T steps per wave); the backward sweep meets (lane, V)
// at forward step V + t = Vtot-1 - s + T-1, uniform over the wave, so its loads stay coalesced rows.
// Every cell is the same expression of the same neighbours as in fb_kernel: results are bit-identical (the GPU and emulator
// suites run every all-pairs set through both kernels, MPCGPU_FB_CHAIN=0 selects fb_kernel alone).
#pragma once
#include "kernels_fb.h"

#define MPC_CHAIN_MAX 16 // pairs per chain
#define MPC_CHAIN_TAB_WORDS 8
#define MPC_CHAIN_TAB_BYTES (MPC_CHAIN_MAX * MPC_CHAIN_TAB_WORDS * 4) // per wave: {LY, base, pid, sy, total bits} per pair

struct FbChainParams {
	FbParams f;            // f.order: chain members back to back; f.count: number of chains; f.queue: chain queue
	const u32 *chain_first; // per chain: first entry of f.order
	const u32 *chain_cnt;   // per chain: members (1 .. MPC_CHAIN_MAX), all with the same pair_x
};

template <int H>
__global__ void __launch_bounds__(256, (H == 8) ? 4 : 1) fb_chain_kernel(FbChainParams cp)
{
	const FbParams &p = cp.f;
	MPC_DYN_SMEM(smem_raw);
	__shared__ MpcCoef s_coef[MPC_COEF_ENTRIES];
	float *s_match = (float *)smem_raw; // A*A
	float *s_ins = s_match + p.A * p.A; // A
	u32 *s_tab_all = (u32 *)(s_ins + p.A); // per-wave chain tables
	if (threadIdx.x < MPC_COEF_ENTRIES)
		mpc_coef_table_init(s_coef, (int)threadIdx.x);
	for (int q = threadIdx.x; q < p.A * p.A; q += blockDim.x)
		s_match[q] = p.match[q];
	for (int q = threadIdx.x; q < p.A; q += blockDim.x)
		s_ins[q] = p.ins[q];
	__syncthreads();

	const int t = threadIdx.x & 63;
	const u32 waves_per_block = blockDim.x >> 6;
	const u32 wave = threadIdx.x >> 6;
	const u32 slot = blockIdx.x * waves_per_block + wave;
	float *fm = p.fm_scratch + (u64)slot * p.fm_stride;
	u32 *s_tab = s_tab_all + wave * (MPC_CHAIN_MAX * MPC_CHAIN_TAB_WORDS); // word w of pair k: s_tab[k * 8 + w]
	const float LZ = MPC_LOG_ZERO;
	const float tSM = p.tSM, tSI = p.tSI, tSJ = p.tSJ, tMM = p.tMM, tMI = p.tMI, tMJ = p.tMJ;
	const float tII = p.tII, tIM = p.tIM, tJJ = p.tJJ, tJM = p.tJM;
	const int A = p.A;
	const int NONE_HI = 0x7fffffff, NONE_LO = -0x40000000;
	// wave-uniform reads of the chain table
	auto tabw = [&](int k, int w) -> u32 { return mpc_wave_first(s_tab[k * MPC_CHAIN_TAB_WORDS + w]); };

	for (;;) {
		const u32 qi = mpc_wave_first(atomicAdd(p.queue, t == 0 ? 1u : 0u)); // see fb_kernel
		if (qi >= p.count)
			break;
		const u32 first = cp.chain_first[qi];
		const int C = (int)cp.chain_cnt[qi];
		const u32 pid0 = p.order[first];
		const u32 sx = p.pair_x[pid0];
		const int LX = (int)p.seq_len[sx];
		const u8 *X = p.seq_code + p.seq_off[sx];
		const int T = (LX + H - 1) / H;
		// chain table: lane k describes pair k; the bases by a scan over the <= 16 lanes
		int Vtot;
		{
			const bool mine = t < C;
			const u32 pid = mine ? p.order[first + (u32)t] : 0u;
			const u32 sy = mine ? p.pair_y[pid] : 0u;
			const int ly = mine ? (int)p.seq_len[sy] : -1;
			int incl = ly + 1; // columns 0 .. LY
			for (int d = 1; d < MPC_CHAIN_MAX; d <<= 1) {
				const int o = __shfl_up(incl, d);
				if (t >= d) incl += o;
			}
			MPC_WAVE_LDS_ORDER(); // the previous chain's last reads of the table come first
			if (mine) {
				u32 *e = s_tab + t * MPC_CHAIN_TAB_WORDS;
				e[0] = (u32)ly; e[1] = (u32)(incl - (ly + 1)); e[2] = pid; e[3] = sy; e[4] = 0u;
			}
			MPC_WAVE_LDS_ORDER();
			Vtot = (int)mpc_wave_first((u32)__shfl(incl, C - 1));
		}
		auto LYof = [&](int k) { return (int)tabw(k, 0); };
		auto baseof = [&](int k) { return (int)tabw(k, 1); };
		auto Yof = [&](int k) { return p.seq_code + p.seq_off[tabw(k, 3)]; };

		// ------------------------------------------------------------------ forward
		float cM[H], cIX[H], cJX[H], cIY[H], cJY[H]; // own rows at the previous column
		float insx[H];
		int mrow[H];
#pragma unroll
		for (int r = 0; r < H; ++r) {
			const int i = t * H + r + 1;
			const int xc = (i <= LX) ? (int)X[i - 1] : 0;
			insx[r] = s_ins[xc];
			mrow[r] = xc * A;
			cM[r] = cIX[r] = cJX[r] = cIY[r] = cJY[r] = LZ;
		}
		float uM = LZ, uIX = LZ, uJX = LZ, uIY = LZ, uJY = LZ; // row t*H at column V-1 (diagonal of r=0)
		float gIY = LZ, gJY = LZ;                              // lane 0: row-0 chain (fwdflat3.cpp:81-93)
		int yprev = 0;
		// lane 0's pair (it loads the letters), the boundary inside the wave, the pair whose last column lane T-1 reaches next
		int k0 = 0, base0 = 0, LY0 = LYof(0), next0 = (C > 1) ? baseof(1) : NONE_HI;
		const u8 *Y0 = Yof(0);
		int kb = 1, vb = (C > 1) ? baseof(1) : NONE_HI; // first boundary not yet passed by lane T-1
		int ke = 0, ve = LYof(0);                        // V of pair ke's last column
		const int nsteps = Vtot + T - 1;
		for (int s = 0; s < nsteps; ++s) {
			const int V = s - t;
			if (s == next0) { // lane 0 enters pair k0+1 at its column 0
				++k0; base0 = next0; LY0 = LYof(k0); Y0 = Yof(k0);
				next0 = (k0 + 1 < C) ? baseof(k0 + 1) : NONE_HI;
			}
			const int j0 = s - base0; // lane 0's column within its pair
			float nM = mpc_lane_up1(cM[H - 1]);
			float nIX = mpc_lane_up1(cIX[H - 1]);
			float nJX = mpc_lane_up1(cJX[H - 1]);
			float nIY = mpc_lane_up1(cIY[H - 1]);
			float nJY = mpc_lane_up1(cJY[H - 1]);
			int yc = mpc_lane_up1(yprev);
			const int yload = (j0 >= 1 && j0 <= LY0) ? (int)Y0[j0 - 1] : 0; // lane 0: letter of its column
			if (t == 0)
				yc = yload;
			const float insy = s_ins[yc];
			if (t == 0) { // row 0: kernels_fb.h:243-251
				nM = LZ; nIX = LZ; nJX = LZ;
				if (j0 <= 0) { nIY = LZ; nJY = LZ; }
				else if (j0 == 1) { nIY = tSI + insy; nJY = tSJ + insy; }
				else { nIY = gIY + tII + insy; nJY = gJY + tJJ + insy; }
				gIY = nIY; gJY = nJY;
			}
			if (V == vb) { // column 0 of the next pair: what lies to the left of it is the virtual column of LOG_ZEROs
#pragma unroll
				for (int r = 0; r < H; ++r) cM[r] = cIX[r] = cJX[r] = cIY[r] = cJY[r] = LZ;
				uM = uIX = uJX = uIY = uJY = LZ;
			}
			float dM = uM, dIX = uIX, dJX = uJX, dIY = uIY, dJY = uJY; // (i-1, j-1)
			float upM = nM, upIX = nIX, upJX = nJX;                     // (i-1, j)
			float *fmrow = fm + ((u64)s * H) * 64 + t;
#pragma unroll
			for (int r = 0; r < H; ++r) {
				const float oM = cM[r], oIX = cIX[r], oJX = cJX[r], oIY = cIY[r], oJY = cJY[r]; // (i, j-1)
				const float m = s_match[mrow[r] + yc];
				// fwdflat3.cpp:116-145, as in fb_kernel
				float vM = mpc_la5t(dM + tMM, dIX + tIM, dJX + tJM, dIY + tIM, dJY + tJM, s_coef) + m;
				float vIX = mpc_la2t(upIX + tII, upM + tMI, s_coef) + insx[r];
				float vJX = mpc_la2t(upJX + tJJ, upM + tMJ, s_coef) + insx[r];
				float vIY = mpc_la2t(oIY + tII, oM + tMI, s_coef) + insy;
				float vJY = mpc_la2t(oJY + tJJ, oM + tMJ, s_coef) + insy;
				if (r == 0) {
					if (t == 0 && j0 == 0) { vIX = tSI + insx[0]; vJX = tSJ + insx[0]; } // fwdflat3.cpp:42-43
					if (t == 0 && j0 == 1) vM = tSM + m;                                 // fwdflat3.cpp:111-112
				}
				cM[r] = vM; cIX[r] = vIX; cJX[r] = vJX; cIY[r] = vIY; cJY[r] = vJY;
				fmrow[r * 64] = vM;
				dM = oM; dIX = oIX; dJX = oJX; dIY = oIY; dJY = oJY;
				upM = vM; upIX = vIX; upJX = vJX;
			}
			uM = nM; uIX = nIX; uJX = nJX; uIY = nIY; uJY = nJY;
			yprev = yc;
			const int VT = s - (T - 1); // lane T-1's column
			if (VT == ve) {
				// F(LX, LY, *) of pair ke sits in lane T-1, row (LX-1)%H: totalprobflat.cpp:3-16 as in fb_kernel
				float eM = LZ, eIX = LZ, eJX = LZ, eIY = LZ, eJY = LZ;
				const int rl = (LX - 1) % H;
#pragma unroll
				for (int r = 0; r < H; ++r)
					if (r == rl) { eM = cM[r]; eIX = cIX[r]; eJX = cJX[r]; eIY = cIY[r]; eJY = cJY[r]; }
				eM = __shfl(eM, T - 1); eIX = __shfl(eIX, T - 1); eJX = __shfl(eJX, T - 1);
				eIY = __shfl(eIY, T - 1); eJY = __shfl(eJY, T - 1);
				float total = LZ;
				total = mpc_la2t(total, eM + tSM, s_coef);
				total = mpc_la2t(total, eIX + tSI, s_coef);
				total = mpc_la2t(total, eIY + tSI, s_coef);
				total = mpc_la2t(total, eJX + tSJ, s_coef);
				total = mpc_la2t(total, eJY + tSJ, s_coef);
				const u32 pid_e = tabw(ke, 2);
				if (t == 0) {
					s_tab[ke * MPC_CHAIN_TAB_WORDS + 4] = __float_as_uint(total);
					p.total[pid_e] = total;
				}
				MPC_WAVE_LDS_ORDER();
				++ke;
				ve = (ke < C) ? baseof(ke) + LYof(ke) : NONE_HI;
			}
			if (VT == vb) { // lane T-1 has just started the next pair: the boundary after it comes into view
				++kb;
				vb = (kb < C) ? baseof(kb) : NONE_HI;
			}
		}
		MPC_WAVE_LDS_ORDER();

		// ------------------------------------------------------------------ backward + posterior
		// Row i uses the emissions of x_{i+1}=X[i] and y_{j+1}=Y[j] (bwdflat3.cpp:46,64).
#pragma unroll
		for (int r = 0; r < H; ++r) {
			const int i = t * H + r + 1;
			const int xc = (i < LX) ? (int)X[i] : 0;
			insx[r] = s_ins[xc];
			mrow[r] = xc * A;
			cM[r] = cIX[r] = cJX[r] = cIY[r] = cJY[r] = LZ; // virtual column LY+1 of the last pair
		}
		float gM = LZ; // row (t+1)*H+1 at column V+1: diagonal of r=H-1
		int ynext_prev = 0;
		// the leading lane's pair (letters, the corner cell); the pairs above (A) and below (B) the boundary inside the wave
		int kl = C - 1, basel = baseof(kl), LYl = LYof(kl);
		const u8 *Yl = Yof(kl);
		int ka = C - 1;
		int baseA = baseof(ka), baseB = (ka >= 1) ? baseof(ka - 1) : 0;
		int vbB = (ka >= 1) ? baseA : NONE_LO; // the boundary inside (or ahead of) the wave: column 0 of pair A
		u32 pidA = tabw(ka, 2), pidB = (ka >= 1) ? tabw(ka - 1, 2) : 0u;
		float totA = __uint_as_float(tabw(ka, 4)), totB = (ka >= 1) ? __uint_as_float(tabw(ka - 1, 4)) : 0.0f;
		u32 ncA = 0, ncB = 0;
		const int bsteps = Vtot + T - 2;
		for (int s = 0; s < bsteps; ++s) {
			const int Vlead = Vtot - 1 - s; // lane T-1's column
			const int V = Vlead + (T - 1 - t);
			if (Vlead < basel && kl > 0) { // the leading lane moves into the pair below
				--kl; basel = baseof(kl); LYl = LYof(kl); Yl = Yof(kl);
			}
			const int jl = Vlead - basel;
			float nM = mpc_lane_down1(cM[0]);
			float nIX = mpc_lane_down1(cIX[0]);
			float nJX = mpc_lane_down1(cJX[0]);
			if (t == 63) { nM = LZ; nIX = LZ; nJX = LZ; } // nothing below the wave: virtual row
			int yc = mpc_lane_down1(ynext_prev);
			const int yload = (jl >= 0 && jl < LYl) ? (int)Yl[jl] : 0; // y_{j+1} of the leading lane's column
			if (t >= T - 1)
				yc = yload; // leading lane (and idle lanes beyond it)
			const float insy = s_ins[yc];
			const bool corner = jl == LYl; // lane T-1 stands on (., LY) of its pair: bwdflat3.cpp:53-61 for row LX
			const int sf = Vlead + T - 1;  // forward step that stored this lane's column V (= V + t)
			const float *fmrow = fm + ((u64)(sf < 0 ? 0 : sf) * H) * 64 + t;
			const bool lo = V < vbB; // below the boundary: pair B
			const int j = V - (lo ? baseB : baseA);
			const float total = lo ? totB : totA;
			const bool incol = (j >= 1) && (V < Vtot);
			float dgM = gM;               // M(i+1, j+1)
			float dnIX = nIX, dnJX = nJX; // (i+1, j)
			bool anyhit = false;
			float sc[H];
#pragma unroll
			for (int r = H - 1; r >= 0; --r) {
				const int i = t * H + r + 1;
				const float oM = cM[r], oIY = cIY[r], oJY = cJY[r]; // (i, j+1)
				// bwdflat3.cpp:75-79
				const float xM = dgM + s_match[mrow[r] + yc];
				const float xIX = dnIX + insx[r];
				const float xJX = dnJX + insx[r];
				const float xIY = oIY + insy;
				const float xJY = oJY + insy;
				// bwdflat3.cpp:81-118 and, over LOG_ZERO virtual neighbours, :132-176 — as in fb_kernel
				float vM = mpc_la5t(tMM + xM, tMI + xIX, tMJ + xJX, tMI + xIY, tMJ + xJY, s_coef);
				float vIX = mpc_la2t(tII + xIX, tIM + xM, s_coef);
				float vJX = mpc_la2t(tJJ + xJX, tJM + xM, s_coef);
				float vIY = mpc_la2t(tII + xIY, tIM + xM, s_coef);
				float vJY = mpc_la2t(tJJ + xJY, tJM + xM, s_coef);
				if (i == LX && corner) { // row LX lives in lane T-1, whose column this is
					vM = tSM; vIX = tSI; vIY = tSI; vJX = tSJ; vJY = tSJ;
				}
				// calcposteriorflat.cpp:14: Score = F_M + B_M - Total
				const float f = fmrow[r * 64];
				const float score = (f + vM) - total;
				sc[r] = score;
				anyhit = anyhit || ((i <= LX) && incol && score >= p.thr);
				dgM = oM;
				cM[r] = vM; cIX[r] = vIX; cJX[r] = vJX; cIY[r] = vIY; cJY[r] = vJY;
				dnIX = vIX; dnJX = vJX;
			}
			gM = nM;
			ynext_prev = yc;
			if (V == vbB) { // this was column 0 of pair A; for pair B it stands where the virtual column LY+1 is
#pragma unroll
				for (int r = 0; r < H; ++r) cM[r] = cIX[r] = cJX[r] = cIY[r] = cJY[r] = LZ;
			}
			if (__ballot(anyhit)) {
				u64 *candA = p.cand + (u64)pidA * p.capc, *candB = p.cand + (u64)pidB * p.capc;
				const u64 below = (1ull << t) - 1ull;
				if (vbB >= Vlead && vbB <= Vlead + (T - 1)) { // the boundary is inside the wave: two lists grow
#pragma unroll
					for (int r = 0; r < H; ++r) {
						const int i = t * H + r + 1;
						const bool hit = (i <= LX) && incol && (sc[r] >= p.thr);
						const u64 bal = __ballot(hit);
						if (bal) {
							const u64 balB = __ballot(hit && lo), balA = bal & ~balB;
							const u32 pos = lo ? ncB + (u32)__popcll(balB & below) : ncA + (u32)__popcll(balA & below);
							if (hit && pos < p.capc) {
								const u32 idx = ((u32)(i - 1) << MPC_KEY_ROW_SHIFT) | (u32)(j - 1);
								(lo ? candB : candA)[pos] = ((u64)idx << 32) | (u64)__float_as_uint(sc[r]);
							}
							ncA += (u32)__popcll(balA);
							ncB += (u32)__popcll(balB);
						}
					}
				} else { // every lane is in the same pair: as in fb_kernel
					const bool allB = Vlead + (T - 1) < vbB; // wave-uniform
					u64 *cand1 = allB ? candB : candA;
					u32 nc1 = allB ? ncB : ncA;
#pragma unroll
					for (int r = 0; r < H; ++r) {
						const int i = t * H + r + 1;
						const bool hit = (i <= LX) && incol && (sc[r] >= p.thr);
						const u64 bal = __ballot(hit);
						if (bal) {
							const u32 pos = nc1 + (u32)__popcll(bal & below);
							if (hit && pos < p.capc) {
								const u32 idx = ((u32)(i - 1) << MPC_KEY_ROW_SHIFT) | (u32)(j - 1);
								cand1[pos] = ((u64)idx << 32) | (u64)__float_as_uint(sc[r]);
							}
							nc1 += (u32)__popcll(bal);
						}
					}
					if (allB) ncB = nc1; else ncA = nc1;
				}
			}
			if (Vlead + (T - 1) == vbB) { // lane 0 has left pair A: its list is complete, B becomes A
				if (t == 0)
					p.cand_cnt[pidA] = ncA;
				--ka;
				pidA = pidB; ncA = ncB; totA = totB; baseA = baseB;
				vbB = (ka >= 1) ? baseA : NONE_LO;
				ncB = 0;
				if (ka >= 1) { baseB = baseof(ka - 1); pidB = tabw(ka - 1, 2); totB = __uint_as_float(tabw(ka - 1, 4)); }
				else baseB = 0;
			}
		}
		if (t == 0)
			p.cand_cnt[pidA] = ncA;
	}
}
