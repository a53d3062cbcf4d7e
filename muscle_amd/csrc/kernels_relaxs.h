// kernels_relaxs.h — consistency relax with the staging taken off the merging waves: relax_stream_kernel.
//
// Same arithmetic, same order of additions and same records as relax_var_kernel (kernels_relaxv.h; reference:
// conspairflat.cpp:10-110, relaxflat.cpp:4-94, mysparsemx.cpp:87-113) — what differs is who moves the records and when.
// relax_var_kernel stages the 8 records of a step between two barriers, and the 65 KB they take of a workgroup's 80 KB leave
// no room for the next step's: a quarter of its time is the exposed wait. Here
//  * a tile is 2 X sequences x 8 Y sequences, its pairs ordered Y-major: (X0,Y0) (X1,Y0) (X0,Y1) ... — so only ONE Y record is
//    in use at a time. The two X records of a step (one contiguous run of HBM) are double-buffered over the steps, the Y
//    records stream through a ring of 3 slots: 2 x ~19 KB + 3 x ~10.5 KB of LDS;
//  * wave 15 of the workgroup is the PRODUCER: it alone issues the LDS-DMA, in the order the slots become free, waits for the
//    oldest transfer in flight (s_waitcnt vmcnt(k), k = the chunk instructions issued after it) and publishes it by a counter in
//    LDS. Waves 0..14 are CONSUMERS: they merge, and at every change of Y record they add 1 to that ring slot's "done" counter
//    and look at the "published" counter — no barrier inside the walk, so a wave that has one 64-cell unit more than its
//    neighbours in one quarter of a step and one less in the next never makes anybody wait (with barriers at the quarters a
//    fifth of the lanes would idle: 37 units for 16 waves).
// A record is overwritten only after all 15 consumers have counted themselves out of the previous occupant of its slot;
// a consumer reads a record only after the producer has seen its transfer land. Counters only grow within a tile.
#pragma once
#include "kernels_relaxv.h"

#define MPC_RS_NX 2u            // X records resident per step
#define MPC_RS_NY 8u            // Y records streamed per step
#define MPC_RS_TAB_BYTES 1152u  // pair table (16 pairs x 8 dwords) + control block (32 dwords) + block-offset entries of 8 steps x 16 dwords
// Steps of lead of those entries over the consumers: the producer looks an entry up when a slot is free, i.e. when every consumer
// has STARTED step S - 2 (X run of step S) or the step of Y record g - ring (record g: up to 4 steps back when a tile has one Y
// record per step), and step S + LEAD's entries replace step S - 3's in the ring of 8 when a consumer starts step S.
#define MPC_RS_LEAD 5u
#define MPC_RS_OFFTAB_BYTE 640u // ... the latter: entry e of step S at dword (S & 7) * 16 + e; e <= nx: X run, 4 + j (j <= ny): Y records
#define MPC_RS_CTRL_BYTE 512u   // the control block inside that area
#define MPC_RS_SPIN_LIMIT (1u << 24)
// control block (u32 indices): X steps / Y records published by each producer wave, LDS address of the second X record in each X buffer,
// progress of every consumer wave = the Y records (g = Z * ny + j, counted over the whole walk) it is done with
#define MPC_RS_PUBX 0    // 4 words: X steps published by producer 0..3 (a record is there when ALL producers have published it)
#define MPC_RS_PUBY 4    // 4 words: Y records published by producer 0..3
#define MPC_RS_XB1 8     // 2 words
#define MPC_RS_PROG 12   // up to 15 words: progress of the consumer waves
// ring slot of Y record g in a ring of 3 or 4 slots
__device__ __forceinline__ u32 mpc_rs_slot(u32 g, u32 ring) { return ring == 4u ? (g & 3u) : g % 3u; }

struct RelaxStreamParams {
	StoreParams s;
	const u32 *tiles; // 6 u32 per tile: x0, nx, y0, ny | ring slots << 8, capacity of one X buffer, of one ring slot (16-byte blocks, the tile's worst step)
	u32 ntiles;
	u64 k0, k1;       // only pairs in [k0,k1) are relaxed (multi-GPU shard)
	u32 *tile_next;   // 8 counters, zeroed before the launch: next tile of each XCD's range
	u32 diag;         // measurement only (results wrong), bits: 1 = the consumers do the protocol but no merges, 2 = they do not wait for records, 4 = the producers issue no transfer
	u32 *err;         // set when a wait gave up (a protocol error: results are wrong; the host fails the call)
};

// The producer wave of relax_stream_kernel, as a function of its own: its handful of scalars get registers of their own instead
// of sharing the consumers' allocation (inlined, the compiler kept the next transfer's LDS address in a spill slot and every
// reload waited for vmcnt(0): one transfer in flight at a time). Every argument is wave-uniform.
template <u32 CW, u32 NP>
__device__ __attribute__((noinline)) void mpc_rs_producer(const unsigned char *padb, u32 lds0_, u32 *err, u32 pid_, u32 diag_,
	u32 n_, u32 nx_, u32 ny_, u32 ring_, u32 xb_bytes_, u32 yb_bytes_)
{
	const u32 n = mpc_wave_first(n_), nx = mpc_wave_first(nx_), ny = mpc_wave_first(ny_);
	const u32 ring = mpc_wave_first(ring_), xb_bytes = mpc_wave_first(xb_bytes_), yb_bytes = mpc_wave_first(yb_bytes_);
	const u32 ring0 = MPC_RS_TAB_BYTES + 2u * xb_bytes;
	// (arguments of a function arrive in vector registers: pointers are made scalar again, so that the table reads below are
	// scalar loads and not vector loads waited for with vmcnt)
	auto uniform_ptr = [](const void *q) { const u64 v = (u64)q; return (const void *)((u64)mpc_wave_first((u32)v) | ((u64)mpc_wave_first((u32)(v >> 32)) << 32)); };
	padb = (const unsigned char *)uniform_ptr(padb);
	const u32 diag = mpc_wave_first(diag_); // measurement only: bit 2 = no transfer is issued (the protocol alone)
	const u32 pid = mpc_wave_first(pid_); // this producer's number: it moves chunks pid, pid + NP, ... of every record
	const u32 lds0 = mpc_wave_first(lds0_), ctrl = lds0 + MPC_RS_CTRL_BYTE; // LDS addresses (never a generic pointer: see mpc_lds_vread)

	// ---- the producer: transfers in the order their slots become free. Y record g = Z * ny + j goes to ring slot g % 3,
	// free once the 15 consumers are done with record g - 3; the X run of step Z goes to X buffer Z & 1, free once they
	// are done with step Z - 2. Order: X(0) X(1) Y(0) Y(1) Y(2), then per finished record g: Y(g+3), and X(Z+2) when g
	// ends step Z.
	const u32 totalY = n * ny;
	// Where a record starts and ends comes from LDS: the consumers keep the block-offset entries of the next steps there (scalar
	// loads from the table in HBM cost a microsecond each — every step is another 4 KB row — and vector loads would be waited for
	// with vmcnt, i.e. together with the transfers in flight). The entries of a step are in place before its records' slots can be free.
	const u32 offtab = lds0 + MPC_RS_OFFTAB_BYTE;
	u32 zx = 0, gy = 0, gy_step = 0, gy_j = 0, pubx = 0, puby = 0, flen = 0, idle = 0;
	u64 fifo = 0; // transfers in flight, oldest first: 16 bits each = chunk instructions of this wave | kind << 8 (1 = X)
	while (pubx < n || puby < totalY) {
		if (flen < 4u && (zx < n || gy < totalY)) {
			// the next transfer in slot-release order
			const long long trigY = gy < totalY ? (long long)gy - (long long)ring : (1ll << 40);
			const long long trigX = zx < n ? (zx < 2u ? -8ll : (long long)(zx - 1u) * ny - 1) : (1ll << 40);
			const bool nxt_x = trigX < trigY;
			// the slowest consumer's progress: lane l reads wave l's word, a 16-lane minimum
			const u32 pl = mpc_lane_fresh();
			const u32 minprog = mpc_row16_min_u32(pl < CW ? mpc_lds_vread(ctrl + 4u * (MPC_RS_PROG + pl)) : 0xffffffffu);
			const bool free_now = nxt_x ? (zx < 2u || minprog >= (zx - 1u) * ny) : (gy < ring || minprog >= gy + 1u - ring);
			if (free_now) {
				u32 src, len, at, x1 = 0;
				if (nxt_x) {
					const u32 row = offtab + 64u * (zx & 7u);
					const u32 xs = mpc_wave_first(mpc_lds_vread(row)), x1s = mpc_wave_first(mpc_lds_vread(row + (nx > 1u ? 4u : 0u))), xe = mpc_wave_first(mpc_lds_vread(row + 4u * nx));
					src = xs; len = xe - xs; at = MPC_RS_TAB_BYTES + (zx & 1u) * xb_bytes; x1 = x1s - xs;
				} else {
					const u32 row = offtab + 64u * (gy_step & 7u) + 4u * (4u + gy_j);
					const u32 ys = mpc_wave_first(mpc_lds_vread(row)), ye = mpc_wave_first(mpc_lds_vread(row + 4u));
					src = ys; len = ye - ys; at = ring0 + mpc_rs_slot(gy, ring) * yb_bytes;
				}
				u32 mine = 0; // chunks of 64 blocks = 1 KiB, one LDS-DMA instruction each
				for (u32 c0 = 64u * pid; c0 < len; c0 += 64u * NP, ++mine)
					if (c0 + pl < len && !(diag & 4u)) mpc_dma16_at(padb + 16 * ((u64)src + c0 + pl), lds0 + at + 16u * c0);
				fifo |= (u64)(mine | (nxt_x ? 0x100u : 0u)) << (16u * flen);
				++flen;
				if (nxt_x) {
					if (pl == 0 && pid == 0u) mpc_lds_vwrite(ctrl + 4u * (MPC_RS_XB1 + (zx & 1u)), lds0 + at + 16u * x1);
					++zx;
				} else {
					++gy;
					if (++gy_j == ny) { gy_j = 0; ++gy_step; }
				}
				idle = 0;
				continue;
			}
		}
		if (flen) {
			u32 younger = 0;
			for (u32 e = 1; e < flen; ++e) younger += (u32)(fifo >> (16u * e)) & 0xffu;
			mpc_wait_vmcnt(younger); // the oldest transfer has landed
			const u32 pl2 = mpc_lane_fresh();
			if ((fifo >> 8) & 1u) { ++pubx; if (pl2 == 0) mpc_lds_vwrite(ctrl + 4u * (MPC_RS_PUBX + pid), pubx); }
			else { ++puby; if (pl2 == 0) mpc_lds_vwrite(ctrl + 4u * (MPC_RS_PUBY + pid), puby); }
			fifo >>= 16; --flen;
			idle = 0;
		} else {
			if (++idle > MPC_RS_SPIN_LIMIT) { if (mpc_lane_fresh() == 0) *err = 1u; break; }
			MPC_SPIN_PAUSE();
		}
	}
}

// THREADS: workgroup size (two workgroups per CU); its last wave is the producer, the others (CW) are consumers: slot q of
// consumer wave w holds cells [q * 64 * CW + 64 * w, + 64). MAXSLOTS: cells per lane (3 bits per slot in a 64-bit map: <= 21).
template <int THREADS, int NPROD, int MAXSLOTS, class BLOCKS = MpcRvBlocksAsm>
__global__ void __launch_bounds__(THREADS, THREADS / 128) relax_stream_kernel(RelaxStreamParams p)
{
	constexpr u32 CW = THREADS / 64 - NPROD, SLOT_CELLS = CW * 64u;
	static_assert(NPROD >= 1 && NPROD <= 4 && CW <= 15, "control block");
	static_assert(MAXSLOTS * 3 <= 64, "slot map");
	MPC_DYN_SMEM(smem_raw);
	const StoreParams &s = p.s;
	const u32 tid = threadIdx.x;
	const u32 lane = tid & 63u;
	const u32 n = s.n;
	const u32 wave = mpc_wave_first(tid >> 6);
	const bool producer = wave >= CW;
	u32 *ptab = (u32 *)smem_raw; // [16][8]: cell base, nnz, sel, k lo, k hi, cells (aligned), -, -
	volatile u32 *ctrl = (volatile u32 *)(smem_raw + MPC_RS_CTRL_BYTE);
	const unsigned char *padb = (const unsigned char *)s.pad;
	const u32 lds0 = mpc_lds_addr(smem_raw);

	const u32 G = gridDim.x < 8u ? gridDim.x : 8u; // tile schedule: as relax_var_kernel
	const u32 xcd = blockIdx.x % G;
	const u32 chunk = (p.ntiles + G - 1u) / G;

	for (;;) {
		__syncthreads(); // the previous tile is done with the table, the control block and the buffers (every transfer was published)
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
		if (tid < 32u) ctrl[tid] = 0u;
		__syncthreads();
		const u32 tl = mpc_wave_first(ptab[8 * 15 + 7]);
		if (tl == 0xffffffffu) break;
		const u32 x0 = mpc_wave_first(p.tiles[6 * tl]), nx = mpc_wave_first(p.tiles[6 * tl + 1]);
		const u32 y0 = mpc_wave_first(p.tiles[6 * tl + 2]), nyr = mpc_wave_first(p.tiles[6 * tl + 3]);
		const u32 ny = nyr & 0xffu, ring = nyr >> 8; // 3 or 4 ring slots
		const u32 xb_bytes = 16u * mpc_wave_first(p.tiles[6 * tl + 4]), yb_bytes = 16u * mpc_wave_first(p.tiles[6 * tl + 5]);
		const u32 ring0 = MPC_RS_TAB_BYTES + 2u * xb_bytes; // byte offset of ring slot 0 in the dynamic LDS
		// ---- pair table, Y-major: lane q of wave 0 looks after pair (ix = q % 2, iy = q / 2)
		if (tid < 64u) {
			const u32 ix = lane % MPC_RS_NX, iy = lane / MPC_RS_NX;
			const u32 X = x0 + ix, Y = y0 + iy;
			u32 nnz = 0, sel = 0;
			u64 k = 0;
			if (lane < 16u && ix < nx && iy < ny && X < Y) {
				k = mpc_pair_index(n, X, Y);
				if (k >= p.k0 && k < p.k1) nnz = (u32)(s.vbase[k + 1] - s.vbase[k]);
				sel = ix | ((MPC_RV_YLANE + iy) << 8);
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
		// ---- block-offset entries of steps 0..MPC_RS_LEAD (the consumers add a step per step of the walk): entry e of step S
		volatile u32 *offtab = (volatile u32 *)(smem_raw + MPC_RS_OFFTAB_BYTE); // volatile: ordered against the progress words
		if (tid >= 64u && tid < 64u + 16u * (MPC_RS_LEAD + 1u)) {
			const u32 S = (tid - 64u) / 16u, e = tid % 16u;
			if (S < n && (e <= nx || (e >= 4u && e - 4u <= ny)))
				offtab[S * 16u + e] = s.rec_off[(u64)S * n + (e < 4u ? x0 + e : y0 + (e - 4u))];
		}
		__syncthreads();
		const u32 total = mpc_wave_first(ptab[8 * 15 + 6]);
		// consumer wave w, slot q: cells [q * SLOT_CELLS + w * 64, +64)
		const u32 wave_first = wave * 64u;
		const u32 nact = (!producer && total > wave_first) ? (total - wave_first + SLOT_CELLS - 1u) / SLOT_CELLS : 0u;

		if (producer) {
			mpc_rs_producer<CW, (u32)NPROD>(padb, lds0, p.err, wave - CW, p.diag, n, nx, ny, ring, xb_bytes, yb_bytes);
		} else {
			// (the producer holds no cells: accumulators, row offsets and the final division exist in the consumers' branch only, so
			// that the producer's few registers are never spilled — a reload there would wait for every transfer in flight)
			float acc[MAXSLOTS];
			u32 xy[MAXSLOTS];
			u32 vsel_a = 0, vsel_b = 0;
			u64 qmap = 0; // 3 bits per slot: the Y record (quarter of the step) of the slot's pair
	#pragma unroll
			for (int q = 0; q < MAXSLOTS; ++q) {
				MPC_SCHED_BARRIER();
				acc[q] = 1.0f; xy[q] = 0u;
				const u32 g0 = (u32)q * SLOT_CELLS + wave_first;
				if ((u32)q < nact) {
					u32 pi = 0;
					for (u32 j = 1; j < 16u; ++j) if (mpc_wave_first(ptab[8 * j]) <= g0 && mpc_wave_first(ptab[8 * j + 5]) != 0u) pi = j;
					const u32 base = mpc_wave_first(ptab[8 * pi]), nnz = mpc_wave_first(ptab[8 * pi + 1]), sel = mpc_wave_first(ptab[8 * pi + 2]);
					const u64 k = (u64)mpc_wave_first(ptab[8 * pi + 3]) | ((u64)mpc_wave_first(ptab[8 * pi + 4]) << 32);
					vsel_a = mpc_write_lane(vsel_a, 4u * (sel & 0xffu), (u32)q);
					vsel_b = mpc_write_lane(vsel_b, 4u * (sel >> 8), (u32)q);
					qmap |= (u64)((sel >> 8) - MPC_RV_YLANE) << (3 * q);
					const u32 idx = g0 + lane - base;
					if (idx < nnz) {
						const u32 *ent = s.packed + s.pbase[k] + s.seq_len[s.pair_x[k]] + s.seq_len[s.pair_y[k]];
						acc[q] = __uint_as_float(ent[2 * (u64)idx]) * 2.0f; // conspairflat.cpp:29-30
						xy[q] = (ent[2 * (u64)nnz + idx] << 4) | (ent[2 * (u64)idx + 1] << 20);
					}
				}
			}

			// ---- the consumers: walk Z, one Y record after the other inside a step. A wave tells the producer how far it is by ONE
			// word, "done with every Y record before g" (the producer takes the minimum over the waves: a wave that holds no cell of
			// a record, or none at all, never holds a slot back), and reads a record only once it has seen it published.
			u32 knownX = 0, knownY = 0; // what this wave has seen published
			auto wait_pub = [&](u32 idx, u32 need, u32 &known) {
				if (p.diag & 2u) known = need;
				for (u32 spins = 0; known < need; ++spins) {
					u32 m = ctrl[idx]; // published = by every producer
					for (u32 e = 1; e < (u32)NPROD; ++e) { const u32 o = ctrl[idx + e]; m = o < m ? o : m; }
					known = mpc_wave_first(m);
					if (known >= need) break;
					if (spins > MPC_RS_SPIN_LIMIT) { if (lane == 0) *p.err = 1u; known = need; break; }
					MPC_SPIN_PAUSE();
				}
			};
			auto progress = [&](u32 g) { if (lane == 0) ctrl[MPC_RS_PROG + wave] = g; };
			// table duty: this wave keeps entries `wave` and `wave + CW` of the block-offset entries in LDS MPC_RS_LEAD steps ahead (a scalar
			// load at the start of a step, written at the start of the next one) — every consumer wave, also one without cells
			mpc_const_u32p roff = MPC_CONST_U32(s.rec_off);
			const u32 e0 = wave, e1 = wave + CW;
			const bool ok0 = e0 <= nx || (e0 >= 4u && e0 - 4u <= ny), ok1 = e1 < 16u && (e1 <= nx || (e1 >= 4u && e1 - 4u <= ny));
			const u32 s0 = e0 < 4u ? x0 + e0 : y0 + (e0 - 4u), s1 = e1 < 4u ? x0 + e1 : y0 + (e1 - 4u);
			u32 pend0 = 0, pend1 = 0;
			for (u32 Z = 0; Z < n; ++Z) {
				const u32 gz = Z * ny;
				wait_pub(MPC_RS_PUBX, Z + 1u, knownX); // (X(Z) is published once everybody is done with step Z - 2: nobody runs far ahead)
				if (Z > 0u && Z + MPC_RS_LEAD < n && lane == 0) {
					if (ok0) offtab[((Z + MPC_RS_LEAD) & 7u) * 16u + e0] = pend0;
					if (ok1) offtab[((Z + MPC_RS_LEAD) & 7u) * 16u + e1] = pend1;
				}
				if (Z + MPC_RS_LEAD + 1u < n) {
					if (ok0) pend0 = roff[(u64)(Z + MPC_RS_LEAD + 1u) * n + s0];
					if (ok1) pend1 = roff[(u64)(Z + MPC_RS_LEAD + 1u) * n + s1];
				}
				if (nact == 0u) { progress(gz + ny); continue; }
				const u32 xb0 = lds0 + MPC_RS_TAB_BYTES + (Z & 1u) * xb_bytes, xb1 = mpc_wave_first(ctrl[MPC_RS_XB1 + (Z & 1u)]);
				// lane i: LDS address of X record i; lane 8 + j: of Y record j (its ring slot at this step)
				const u32 vbase = lane == 0u ? xb0 : lane == 1u ? xb1 : lds0 + ring0 + mpc_rs_slot(gz + ((lane - MPC_RV_YLANE) & 7u), ring) * yb_bytes;
				const u32 base_a = mpc_lane_gather(vbase, vsel_a), base_b = mpc_lane_gather(vbase, vsel_b);
				BLOCKS blk;
				auto addr_a = [&](int q) -> u32 { return mpc_read_lane(base_a, (u32)q) + (xy[q] & 0xffffu); };
				auto addr_b = [&](int q) -> u32 { return mpc_read_lane(base_b, (u32)q) + (xy[q] >> 16); };
				u32 curj = (u32)qmap & 7u; // the Y record in use
				if (curj != 0u) progress(gz + curj);
				wait_pub(MPC_RS_PUBY, gz + curj + 1u, knownY);
				u32 nia = addr_a(0), nib = addr_b(0);
				blk.load(0, nia, nib);
				auto slot = [&](auto &&self, auto qc) __attribute__((always_inline)) {
					constexpr int q = decltype(qc)::value;
					if constexpr (q < MAXSLOTS) {
						if ((u32)q >= nact) return;
						const u32 ia = nia, ib = nib;
						constexpr int qn = q + 1 < MAXSLOTS ? q + 1 : q;
						nia = addr_a(qn); nib = addr_b(qn);
						// the next slot's Y record: when it differs and this wave has not seen it published yet, the read of its first
						// blocks that the merge below issues may find the transfer still on its way — it is then repeated after the wait
						const u32 jn = (u32)(q + 1) < nact ? (u32)(qmap >> (3 * qn)) & 7u : curj;
						const bool seen = knownY >= gz + jn + 1u;
						if (!(p.diag & 1u)) {
							float sum = acc[q];
							blk.template merge<q & 1>(sum, ia, ib, nia, nib);
							acc[q] = sum;
						}
						if (jn != curj) {
							progress(gz + jn);
							curj = jn;
							if (!seen) {
								wait_pub(MPC_RS_PUBY, gz + jn + 1u, knownY);
								blk.load(qn & 1, nia, nib);
							}
						}
						self(self, std::integral_constant<int, q + 1>{});
					}
				};
				slot(slot, std::integral_constant<int, 0>{});
				progress(gz + ny);
			}
			// ---- UpdateFromPost (mysparsemx.cpp:87-113): P' = acc / N on the frozen pattern
	#pragma unroll
			for (int q = 0; q < MAXSLOTS; ++q) {
				MPC_SCHED_BARRIER();
				const u32 g0 = (u32)q * SLOT_CELLS + wave_first;
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
}

// Worst step of a stream tile: out[2t] = max over Z of the blocks of the X run (records x0 .. x0+nx-1), out[2t+1] = max over Z
// and j of the blocks of ONE Y record. One wave per tile, lanes stride over Z.
__global__ void __launch_bounds__(64) stream_tile_fit_kernel(StoreParams s, const u32 *tiles, u32 ntiles, u32 *out)
{
	const u32 t = threadIdx.x, n = s.n;
	for (u32 tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
		const u32 x0 = tiles[6 * tl], nx = tiles[6 * tl + 1], y0 = tiles[6 * tl + 2], ny = tiles[6 * tl + 3];
		u32 bx = 0, by = 0;
		for (u32 Z = t; Z < n; Z += 64) {
			const u64 ix = mpc_rec_index(n, x0, Z), iy = mpc_rec_index(n, y0, Z);
			const u32 xr = s.rec_off[ix + nx] - s.rec_off[ix];
			bx = xr > bx ? xr : bx;
			for (u32 j = 0; j < ny; ++j) { const u32 yr = s.rec_off[iy + j + 1] - s.rec_off[iy + j]; by = yr > by ? yr : by; }
		}
		for (int d = 32; d >= 1; d >>= 1) {
			const u32 ox = __shfl_down(bx, d), oy = __shfl_down(by, d);
			bx = ox > bx ? ox : bx; by = oy > by ? oy : by;
		}
		if (t == 0) { out[2 * tl] = bx; out[2 * tl + 1] = by; }
	}
}
