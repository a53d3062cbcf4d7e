// mpcgpu_group.cpp — several GPUs of one node inside ONE process (the drop-in binary, hostcxx/): the N(N-1)/2 pair schedule
// of MPCFlat::CalcPosteriors (mpcflat.cpp:239-251) and of MPCFlat::ConsIter (consflat.cpp:5-23) sharded over the devices,
// with the two exchanges the sharding needs:
//   after stage A   every device receives every other device's packed sparse posteriors (the all-gather before relax),
//   after each relax iteration  every device receives every other device's slice of the new probabilities.
// Built on the public C ABI of include/mpcgpu.h only (one mpcgpu_ctx per device, one host thread per device while a
// sharded step runs). Transport, chosen at group creation:
//   "rccl"  RCCL over xGMI: one communicator per device (ncclCommInitAll), every exchange ONE group of point-to-point
//           ncclSend / ncclRecv with exact sizes — rank r sends its segment to each peer and receives each peer's segment at
//           its final offset (no padding to the largest shard, no second copy). xGMI is a full mesh of point-to-point
//           links, so the 7 sends of a rank travel on 7 different links. librccl is dlopen()ed on first use: a process
//           that never makes a group (bench.py, the tests of the single-GPU path) does not need it.
//   "peer"  hipMemcpyPeerAsync from the owner into every peer's buffer (also the only form that works when two contexts sit
//           on the SAME device: that is how the tests exercise this file on a one-GPU box; MPCGPU_GROUP_TRANSPORT=peer).
#include "../../include/mpcgpu.h"
#include "mpc_platform.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// the slice of the RCCL API used here (rccl.h: ncclCommInitAll :236, ncclCommDestroy :260, ncclSend :700, ncclRecv :722,
// ncclGroupStart :923, ncclGroupEnd :933); ncclChar = 0, ncclFloat = 7 (:459-466)
struct Rccl {
	void *lib = nullptr;
	int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
	int (*CommDestroy)(void *comm) = nullptr;
	int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
	int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	bool load(std::string &why)
	{
		const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
		for (const char *nm : names) { lib = mpc_dl_open(nm); if (lib) break; }
		if (!lib) { why = std::string("librccl: ") + mpc_dl_error(); return false; }
		auto sym = [&](const char *s) { void *p = mpc_dl_sym(lib, s); if (!p) why = std::string("librccl lacks ") + s; return p; };
		*(void **)&CommInitAll = sym("ncclCommInitAll");
		*(void **)&CommDestroy = sym("ncclCommDestroy");
		*(void **)&Send = sym("ncclSend");
		*(void **)&Recv = sym("ncclRecv");
		*(void **)&GroupStart = sym("ncclGroupStart");
		*(void **)&GroupEnd = sym("ncclGroupEnd");
		*(void **)&GetErrorString = sym("ncclGetErrorString");
		return CommInitAll && CommDestroy && Send && Recv && GroupStart && GroupEnd && GetErrorString;
	}
};

} // namespace

// One host thread per rank, started once per group and parked between the phases of a sharded step (stage A, the two exchanges,
// store import, relax, commit: a step used to start and join a fresh set of threads for each of them).
struct RankPool {
	std::vector<std::thread> th;
	std::mutex mu;
	std::condition_variable wake, done;
	std::function<int(uint32_t)> fn;
	uint64_t gen = 0;
	uint32_t pending = 0;
	bool stop = false;
	std::vector<int> rc;
	explicit RankPool(uint32_t R) : rc(R, 0)
	{
		for (uint32_t r = 0; r < R; ++r)
			th.emplace_back([this, r]() {
				uint64_t seen = 0;
				for (;;) {
					std::function<int(uint32_t)> job;
					{
						std::unique_lock<std::mutex> lk(mu);
						wake.wait(lk, [&]() { return stop || gen != seen; });
						if (stop) return;
						seen = gen;
						job = fn;
					}
					const int v = job(r);
					std::lock_guard<std::mutex> lk(mu);
					rc[r] = v;
					if (--pending == 0) done.notify_all();
				}
			});
	}
	int run(const std::function<int(uint32_t)> &f)
	{
		std::unique_lock<std::mutex> lk(mu);
		fn = f;
		pending = (uint32_t)th.size();
		std::fill(rc.begin(), rc.end(), 0);
		++gen;
		wake.notify_all();
		done.wait(lk, [&]() { return pending == 0; });
		for (int v : rc) if (v) return v;
		return 0;
	}
	~RankPool()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		wake.notify_all();
		for (auto &t : th) t.join();
	}
};

struct mpcgpu_group {
	RankPool *pool = nullptr;
	std::vector<int> dev;
	std::vector<mpcgpu_ctx *> ctx;
	std::vector<hipStream_t> xs;   // exchange stream per rank
	std::vector<void *> gbuf;      // gathered packed shards per rank (the store of that rank keeps reading it)
	std::vector<size_t> gcap;
	std::vector<uint64_t> k0, k1;  // pair shard of each rank
	bool use_rccl = false;
	Rccl rccl;
	std::vector<void *> comm;
	std::string err, transport_note;
	uint32_t n = 0;
	std::vector<uint32_t> len;
	bool have_store = false;
};

namespace {

std::string g_group_create_err;
std::mutex g_err_mu; // the per-rank host threads of a sharded step may fail at the same time

int gfail(mpcgpu_group *g, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	std::lock_guard<std::mutex> guard(g_err_mu);
	if (g) g->err = buf; else g_group_create_err = buf;
	return 1;
}

// runs fn(rank) on one host thread per rank; returns the first non-zero result
template <class F> int per_rank(mpcgpu_group *g, F fn)
{
	const size_t R = g->ctx.size();
	if (R == 1) return fn(0u);
	if (!g->pool) g->pool = new RankPool((uint32_t)R);
	return g->pool->run(std::function<int(uint32_t)>(fn));
}

// Contiguous pair ranges balanced by DP cells sum (LX+1)(LY+1), InitPairs order (mpcflat.cpp:139-159): the same cuts as
// muscle_amd/mpcflat.py shard_bounds (the Python path of bench.py), so both hosts shard alike.
void shard_bounds(const std::vector<uint32_t> &len, uint32_t world, std::vector<uint64_t> &cuts)
{
	const uint32_t n = (uint32_t)len.size();
	std::vector<uint64_t> w;
	w.reserve((size_t)n * (n - 1) / 2);
	uint64_t run = 0;
	for (uint32_t i = 0; i < n; ++i)
		for (uint32_t j = i + 1; j < n; ++j) { run += (uint64_t)(len[i] + 1) * (len[j] + 1); w.push_back(run); }
	const uint64_t total = run, np = w.size();
	cuts.assign(1, 0);
	for (uint32_t r = 1; r < world; ++r) {
		const uint64_t target = total * r / world;
		cuts.push_back((uint64_t)(std::lower_bound(w.begin(), w.end(), target) - w.begin()));
	}
	cuts.push_back(np);
	for (size_t r = 1; r < cuts.size(); ++r) cuts[r] = std::max(cuts[r], cuts[r - 1]);
}

// ---- block partition -------------------------------------------------------------------------------------------------
// A rank that owns a contiguous range of InitPairs pairs relaxes pairs (X, Y) with Y anywhere behind X: it reads the records of
// (nearly) every sequence, so every rank imports, builds and commits the WHOLE store — the part of a rank's step that does not
// shrink with the number of ranks. Cut in two dimensions instead: the sequences fall into g groups of equal weight (sum of L + 1),
// the pair triangle into g (g + 1) / 2 blocks (group i x group j), and a rank owns whole blocks — its pairs then touch the
// sequences of a few groups only, and it needs the records of those (mpcgpu_store_import_part).
//   world = g (g - 1) / 2 + g / 2, g even (2, 8, 18 ...): one off-diagonal block per rank, and one rank per two diagonal blocks
//     (a triangle weighs half a square): at 8 ranks every rank touches 2 of 4 groups — HALF of the store.
//   any other world: g = world groups, rank i owns the triangle of group i and the blocks {i, i + d}, d = 1 .. (g - 1) / 2 (indices
//     mod g); for even g the g / 2 antipodal blocks {i, i + g / 2} are cut in two by rows, one half for either rank.
// The pairs are enumerated rank by rank, block by block (mpcgpu_set_pair_order), so that a rank's pairs are one contiguous range
// of positions and everything per pair — shards, values — stays one segment per rank.
struct PlanRect { uint32_t xa, xb, ya, yb; };

static uint64_t rect_pairs(const PlanRect &q) { return q.ya >= q.xb ? (uint64_t)(q.xb - q.xa) * (q.yb - q.ya) : (uint64_t)(q.xb - q.xa) * (q.xb - q.xa - 1) / 2; }

static bool plan_blocks(const std::vector<uint32_t> &len, uint32_t world, std::vector<std::vector<PlanRect>> &per_rank)
{
	const uint32_t n = (uint32_t)len.size();
	uint32_t g = 0;
	bool paired = false;
	for (uint32_t e = 2; e * (e - 1) / 2 + e / 2 <= world; e += 2) if (e * (e - 1) / 2 + e / 2 == world) { g = e; paired = true; }
	if (!g) g = world;
	if (world < 2 || n < 4 * g) return false; // too few sequences for groups: contiguous ranges
	std::vector<uint64_t> cum(n + 1, 0);
	for (uint32_t i = 0; i < n; ++i) cum[i + 1] = cum[i] + len[i] + 1;
	auto cut_at = [&](uint32_t lo, uint32_t hi, uint64_t num, uint64_t den) { // first index in [lo, hi] whose prefix weight reaches num / den of the range's
		const uint64_t target = cum[lo] + (cum[hi] - cum[lo]) * num / den;
		return (uint32_t)(std::lower_bound(cum.begin() + lo, cum.begin() + hi + 1, target) - cum.begin());
	};
	std::vector<uint32_t> gb(g + 1, 0);
	for (uint32_t j = 1; j < g; ++j) gb[j] = std::max(cut_at(0, n, j, g), gb[j - 1] + 1);
	gb[g] = n;
	for (uint32_t j = g; j-- > 1;) if (gb[j] >= gb[j + 1]) gb[j] = gb[j + 1] - 1; // (every group keeps at least one sequence)
	auto tri = [&](uint32_t i) { return PlanRect{gb[i], gb[i + 1], gb[i], gb[i + 1]}; };
	auto blk = [&](uint32_t i, uint32_t j) { if (i > j) std::swap(i, j); return PlanRect{gb[i], gb[i + 1], gb[j], gb[j + 1]}; };
	per_rank.assign(world, {});
	if (paired) {
		uint32_t r = 0;
		for (uint32_t t = 0; t < g; t += 2) { per_rank[r].push_back(tri(t)); per_rank[r].push_back(tri(t + 1)); ++r; }
		for (uint32_t i = 0; i < g; ++i)
			for (uint32_t j = i + 1; j < g; ++j) per_rank[r++].push_back(blk(i, j));
		return true;
	}
	for (uint32_t i = 0; i < g; ++i) {
		per_rank[i].push_back(tri(i));
		for (uint32_t d = 1; d <= (g - 1) / 2; ++d) per_rank[i].push_back(blk(i, (i + d) % g));
	}
	if (g % 2 == 0)
		for (uint32_t i = 0; i < g / 2; ++i) { // the antipodal block, cut by rows at half of its weight
			PlanRect b = blk(i, i + g / 2);
			const uint32_t mid = std::min(std::max(cut_at(b.xa, b.xb, 1, 2), b.xa), b.xb);
			if (mid > b.xa) per_rank[i].push_back(PlanRect{b.xa, mid, b.ya, b.yb});
			if (b.xb > mid) per_rank[i + g / 2].push_back(PlanRect{mid, b.xb, b.ya, b.yb});
		}
	return true;
}

#define GHIP(g, call)                                                                                           \
	do {                                                                                                        \
		hipError_t e_ = (call);                                                                                 \
		if (e_ != hipSuccess) return gfail((g), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

// Segment r (seg_bytes[r] bytes at src[r] on rank r's device) goes to dst[d] + seg_off[r] on every rank d. With
// in_place the segment already sits at its final place on its owner (dst[r] + seg_off[r] == src[r]).
// wait == false: the transfers are queued on the exchange streams and the caller synchronises those later (the next piece of
// stage A runs meanwhile).
int all_gather_segments(mpcgpu_group *g, const std::vector<const void *> &src, const std::vector<void *> &dst,
	const std::vector<uint64_t> &seg_off, const std::vector<uint64_t> &seg_bytes, bool in_place, bool wait = true)
{
	const uint32_t R = (uint32_t)g->ctx.size();
	if (g->use_rccl) {
		// one group: every rank sends its segment to every peer and receives every peer's segment at its final offset
		int rc = g->rccl.GroupStart();
		if (rc) return gfail(g, "ncclGroupStart: %s", g->rccl.GetErrorString(rc));
		hipError_t herr = hipSuccess; // a HIP error inside the group must not leave RCCL with an open group: record it, close, then fail
		for (uint32_t r = 0; r < R && !rc && herr == hipSuccess; ++r) {
			herr = hipSetDevice(g->dev[r]);
			if (herr == hipSuccess && !in_place && seg_bytes[r])
				herr = hipMemcpyAsync((char *)dst[r] + seg_off[r], src[r], seg_bytes[r], hipMemcpyDeviceToDevice, g->xs[r]);
			if (herr != hipSuccess) break;
			for (uint32_t d = 0; d < R && !rc; ++d) {
				if (d == r) continue;
				if (seg_bytes[r]) rc = g->rccl.Send(src[r], seg_bytes[r], 0 /* ncclChar */, (int)d, g->comm[r], g->xs[r]);
				if (!rc && seg_bytes[d]) rc = g->rccl.Recv((char *)dst[r] + seg_off[d], seg_bytes[d], 0, (int)d, g->comm[r], g->xs[r]);
			}
		}
		const int rc2 = g->rccl.GroupEnd();
		if (herr != hipSuccess) return gfail(g, "exchange: %s", hipGetErrorString(herr));
		if (rc || rc2) return gfail(g, "RCCL exchange: %s", g->rccl.GetErrorString(rc ? rc : rc2));
		for (uint32_t r = 0; r < R && wait; ++r) {
			GHIP(g, hipSetDevice(g->dev[r]));
			GHIP(g, hipStreamSynchronize(g->xs[r]));
		}
		return 0;
	}
	// peer copies: the owner pushes its segment into every rank's buffer (its own included unless in place)
	return per_rank(g, [&](uint32_t r) -> int {
		GHIP(g, hipSetDevice(g->dev[r]));
		for (uint32_t d = 0; d < R; ++d) {
			if (!seg_bytes[r] || (d == r && in_place)) continue;
			GHIP(g, hipMemcpyPeerAsync((char *)dst[d] + seg_off[r], g->dev[d], src[r], g->dev[r], seg_bytes[r], g->xs[r]));
		}
		if (wait) GHIP(g, hipStreamSynchronize(g->xs[r]));
		return 0;
	});
}

} // namespace

extern "C" {

int mpcgpu_plan_partition(uint32_t n, const uint32_t *lens, uint32_t world, uint32_t max_rects, uint32_t *rects, uint32_t *nrects,
	uint64_t *rank_pos)
{
	if (!lens || !nrects || !rank_pos || world == 0 || n < 2) return 1;
	const std::vector<uint32_t> len(lens, lens + n);
	std::vector<std::vector<PlanRect>> per_rank;
	if (!plan_blocks(len, world, per_rank)) {
		std::vector<uint64_t> cuts;
		shard_bounds(len, world, cuts);
		*nrects = 0;
		for (uint32_t r = 0; r <= world; ++r) rank_pos[r] = cuts[r];
		return 0;
	}
	uint32_t nr = 0;
	uint64_t pos = 0;
	for (uint32_t r = 0; r < world; ++r) {
		rank_pos[r] = pos;
		for (const PlanRect &q : per_rank[r]) {
			if (rect_pairs(q) == 0) continue;
			if (nr >= max_rects || !rects) return 2; // the caller's array is too small
			rects[4 * nr] = q.xa; rects[4 * nr + 1] = q.xb; rects[4 * nr + 2] = q.ya; rects[4 * nr + 3] = q.yb;
			++nr;
			pos += rect_pairs(q);
		}
	}
	rank_pos[world] = pos;
	*nrects = nr;
	return pos == (uint64_t)n * (n - 1) / 2 ? 0 : 3;
}

const char *mpcgpu_group_last_error(const mpcgpu_group *g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }
uint32_t mpcgpu_group_size(const mpcgpu_group *g) { return g ? (uint32_t)g->ctx.size() : 0; }
mpcgpu_ctx *mpcgpu_group_ctx(mpcgpu_group *g, uint32_t rank) { return (g && rank < g->ctx.size()) ? g->ctx[rank] : nullptr; }
const char *mpcgpu_group_transport(const mpcgpu_group *g) { return !g ? "" : g->use_rccl ? "rccl" : "peer"; }

void mpcgpu_group_destroy(mpcgpu_group *g)
{
	if (!g) return;
	delete g->pool; // parks no more: the rank threads end before their contexts do
	g->pool = nullptr;
	for (size_t r = 0; r < g->ctx.size(); ++r) {
		(void)hipSetDevice(g->dev[r]);
		if (r < g->comm.size() && g->comm[r]) (void)g->rccl.CommDestroy(g->comm[r]);
		if (g->ctx[r]) mpcgpu_destroy(g->ctx[r]); // before the gather buffer its store reads
		if (r < g->gbuf.size() && g->gbuf[r]) (void)hipFree(g->gbuf[r]);
		if (r < g->xs.size() && g->xs[r]) (void)hipStreamDestroy(g->xs[r]);
	}
	delete g;
}

int mpcgpu_group_create(mpcgpu_group **out, uint32_t ndev, const int *devices)
{
	if (!out || ndev == 0 || !devices) return gfail(nullptr, "mpcgpu_group_create: bad arguments");
	mpcgpu_group *g = new mpcgpu_group;
	g->dev.assign(devices, devices + ndev);
	g->ctx.assign(ndev, nullptr);
	g->xs.assign(ndev, nullptr);
	g->gbuf.assign(ndev, nullptr);
	g->gcap.assign(ndev, 0);
	for (uint32_t r = 0; r < ndev; ++r) {
		if (mpcgpu_create(&g->ctx[r], devices[r]) != 0) {
			gfail(nullptr, "mpcgpu_group_create: device %d: %s", devices[r], mpcgpu_last_error(nullptr));
			mpcgpu_group_destroy(g);
			return 1;
		}
		if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreate(&g->xs[r]) != hipSuccess) {
			gfail(nullptr, "mpcgpu_group_create: stream on device %d", devices[r]);
			mpcgpu_group_destroy(g);
			return 1;
		}
	}
	bool distinct = true;
	for (uint32_t a = 0; a < ndev; ++a)
		for (uint32_t b = a + 1; b < ndev; ++b) if (devices[a] == devices[b]) distinct = false;
	for (uint32_t a = 0; a < ndev; ++a) // direct access between every pair of distinct devices (xGMI)
		for (uint32_t b = 0; b < ndev; ++b)
			if (devices[a] != devices[b]) mpc_enable_peer(devices[a], devices[b]);
	const char *want = getenv("MPCGPU_GROUP_TRANSPORT");
	const bool want_peer = want && !strcmp(want, "peer"), want_rccl = want && !strcmp(want, "rccl");
	if ((ndev > 1 || want_rccl) && distinct && !want_peer) { // (a one-device group only makes a communicator on request: the loader's test)
		std::string why;
		if (g->rccl.load(why)) {
			g->comm.assign(ndev, nullptr);
			const int rc = g->rccl.CommInitAll(g->comm.data(), (int)ndev, devices);
			if (rc == 0) g->use_rccl = true;
			else { g->transport_note = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(rc); g->comm.clear(); }
		} else g->transport_note = why;
		if (!g->use_rccl && want_rccl) {
			gfail(nullptr, "mpcgpu_group_create: RCCL requested but unavailable: %s", g->transport_note.c_str());
			mpcgpu_group_destroy(g);
			return 1;
		}
	}
	*out = g;
	return 0;
}

int mpcgpu_group_set_hmm(mpcgpu_group *g, const float start[5], const float trans[25], const float match[256 * 256],
	const float ins[256], float min_sparse_score, int expf_variant)
{
	if (!g) return 1;
	for (size_t r = 0; r < g->ctx.size(); ++r)
		if (mpcgpu_set_hmm(g->ctx[r], start, trans, match, ins, min_sparse_score, expf_variant))
			return gfail(g, "rank %zu: %s", r, mpcgpu_last_error(g->ctx[r]));
	return 0;
}

int mpcgpu_group_set_seqs(mpcgpu_group *g, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens)
{
	if (!g) return 1;
	g->have_store = false;
	for (size_t r = 0; r < g->ctx.size(); ++r)
		if (mpcgpu_set_seqs(g->ctx[r], n, seqs, lens)) return gfail(g, "rank %zu: %s", r, mpcgpu_last_error(g->ctx[r]));
	g->n = n;
	g->len.assign(lens, lens + n);
	return 0;
}

int mpcgpu_group_set_mega(mpcgpu_group *g, uint32_t nfeat, const uint32_t *alpha, const float *weight,
	const float *const *logprobs, const float *const *logprob_mx, const uint8_t *const *profiles)
{
	if (!g) return 1;
	for (size_t r = 0; r < g->ctx.size(); ++r)
		if (mpcgpu_set_mega(g->ctx[r], nfeat, alpha, weight, logprobs, logprob_mx, profiles))
			return gfail(g, "rank %zu: %s", r, mpcgpu_last_error(g->ctx[r]));
	return 0;
}

int mpcgpu_group_calc_posteriors(mpcgpu_group *g)
{
	if (!g) return 1;
	if (g->n < 2) return gfail(g, "mpcgpu_group_calc_posteriors: call mpcgpu_group_set_seqs first");
	const uint32_t R = (uint32_t)g->ctx.size();
	g->have_store = false;
	// ---- the partition: blocks of the pair triangle (mpcgpu_plan_partition), every context enumerates its pairs rank by rank
	std::vector<uint32_t> rects(4 * (size_t)(R * (R / 2 + 3) + 4));
	std::vector<uint64_t> pos(R + 1, 0);
	uint32_t nrects = 0;
	if (mpcgpu_plan_partition(g->n, g->len.data(), R, (uint32_t)(rects.size() / 4), rects.data(), &nrects, pos.data()))
		return gfail(g, "mpcgpu_group_calc_posteriors: no partition of %u sequences over %u ranks", g->n, R);
	g->k0.assign(pos.begin(), pos.end() - 1);
	g->k1.assign(pos.begin() + 1, pos.end());
	{ // (every context enumerates its pairs in this order: on the ranks' own threads, side by side)
		const int rc0 = per_rank(g, [&](uint32_t r) -> int {
			if (mpcgpu_set_pair_order(g->ctx[r], nrects, rects.data())) return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
			return 0;
		});
		if (rc0) return rc0;
	}
	if (R == 1) { // nothing to exchange
		if (mpcgpu_calc_posteriors(g->ctx[0], g->k0[0], g->k1[0]) || mpcgpu_build_store(g->ctx[0]))
			return gfail(g, "%s", mpcgpu_last_error(g->ctx[0]));
		g->have_store = true;
		return 0;
	}
	// ---- stage A in PIECES (MPCGPU_GROUP_PIECES, default 1: at 8 ranks two pieces cost more stage-A time — small launches of fb_chain_kernel — than the exchange they hide, profiles/r12b): a rank's range is cut by DP cells, and the all-gather of piece p
	// travels (exchange streams) while piece p + 1 is computed (library streams) — only the last piece's exchange is exposed
	const uint32_t P = (uint32_t)std::min(std::max(getenv("MPCGPU_GROUP_PIECES") ? atoi(getenv("MPCGPU_GROUP_PIECES")) : 1, 1), 16);
	std::vector<uint64_t> cutp((size_t)R * (P + 1), 0); // cutp[r * (P + 1) + p]: first position of piece p of rank r
	{
		// DP cells per position, in position order: the pairs of the rectangles (or InitPairs order)
		std::vector<uint64_t> cum(1, 0);
		cum.reserve((size_t)g->n * (g->n - 1) / 2 + 1);
		auto add = [&](uint32_t x, uint32_t y) { cum.push_back(cum.back() + (uint64_t)(g->len[x] + 1) * (g->len[y] + 1)); };
		if (!nrects) { for (uint32_t i = 0; i < g->n; ++i) for (uint32_t j = i + 1; j < g->n; ++j) add(i, j); }
		else
			for (uint32_t q = 0; q < nrects; ++q) {
				const uint32_t xa = rects[4 * q], xb = rects[4 * q + 1], ya = rects[4 * q + 2], yb = rects[4 * q + 3];
				for (uint32_t x = xa; x < xb; ++x) for (uint32_t y = ya >= xb ? ya : x + 1; y < yb; ++y) add(x, y);
			}
		for (uint32_t r = 0; r < R; ++r) {
			uint64_t *cp = &cutp[(size_t)r * (P + 1)];
			cp[0] = g->k0[r]; cp[P] = g->k1[r];
			for (uint32_t p = 1; p < P; ++p) {
				const uint64_t target = cum[g->k0[r]] + (cum[g->k1[r]] - cum[g->k0[r]]) * p / P;
				cp[p] = (uint64_t)(std::lower_bound(cum.begin() + g->k0[r], cum.begin() + g->k1[r] + 1, target) - cum.begin());
				cp[p] = std::min(std::max(cp[p], cp[p - 1]), g->k1[r]);
			}
		}
	}
	std::vector<uint64_t> sk0, sk1, sbytes, soff; // the shards as they lie in every rank's gather buffer: piece after piece, rank after rank
	uint64_t total = 0;
	bool pending = false; // an exchange is in flight on the exchange streams
	auto finish_exchange = [&]() -> int {
		if (!pending) return 0;
		pending = false;
		for (uint32_t r = 0; r < R; ++r) { GHIP(g, hipSetDevice(g->dev[r])); GHIP(g, hipStreamSynchronize(g->xs[r])); }
		return 0;
	};
	for (uint32_t p = 0; p < P; ++p) {
		std::vector<uint64_t> bytes(R, 0);
		std::vector<const void *> src(R, nullptr);
		int rc = per_rank(g, [&](uint32_t r) -> int {
			void *ptr = nullptr;
			const uint64_t *cp = &cutp[(size_t)r * (P + 1)];
			if (mpcgpu_calc_posteriors(g->ctx[r], cp[p], cp[p + 1]) || mpcgpu_shard_info(g->ctx[r], &bytes[r], &ptr) || mpcgpu_synchronize(g->ctx[r]))
				return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
			src[r] = ptr;
			return 0;
		});
		if (rc) return rc;
		uint64_t piece = 0;
		std::vector<uint64_t> off(R, 0);
		for (uint32_t r = 0; r < R; ++r) { off[r] = total + piece; piece += (bytes[r] + 15) & ~15ull; }
		// room for this piece and, by its size, for the ones to come (a buffer that turns out too small is replaced: its content is copied)
		const uint64_t want = total + piece, guess = total + piece * (P - p) + (piece * (P - p)) / 8 + 4096;
		for (uint32_t r = 0; r < R; ++r) {
			if (g->gcap[r] >= want) continue;
			if ((rc = finish_exchange())) return rc;
			GHIP(g, hipSetDevice(g->dev[r]));
			void *nb = nullptr;
			GHIP(g, hipMalloc(&nb, guess));
			if (g->gbuf[r] && total) GHIP(g, hipMemcpy(nb, g->gbuf[r], total, hipMemcpyDeviceToDevice));
			if (g->gbuf[r]) GHIP(g, hipFree(g->gbuf[r]));
			g->gbuf[r] = nb; g->gcap[r] = guess;
		}
		// the own piece leaves the context's shard buffer now (the next stage A overwrites it): a device-local copy on the exchange
		// stream, then the sends read the gather buffer
		if ((rc = finish_exchange())) return rc;
		for (uint32_t r = 0; r < R; ++r) {
			GHIP(g, hipSetDevice(g->dev[r]));
			if (bytes[r]) GHIP(g, hipMemcpyAsync((char *)g->gbuf[r] + off[r], src[r], bytes[r], hipMemcpyDeviceToDevice, g->xs[r]));
			GHIP(g, hipStreamSynchronize(g->xs[r]));
			src[r] = (const char *)g->gbuf[r] + off[r];
		}
		std::vector<void *> dst(g->gbuf.begin(), g->gbuf.end());
		rc = all_gather_segments(g, src, dst, off, bytes, true, false);
		if (rc) return rc;
		pending = true;
		for (uint32_t r = 0; r < R; ++r) {
			const uint64_t *cp = &cutp[(size_t)r * (P + 1)];
			sk0.push_back(cp[p]); sk1.push_back(cp[p + 1]); sbytes.push_back(bytes[r]); soff.push_back(off[r]);
		}
		total += piece;
	}
	int rc = finish_exchange();
	if (rc) return rc;
	// ---- every device builds its store from the gathered shards: the records of the sequences its own pairs touch
	rc = per_rank(g, [&](uint32_t r) -> int {
		if (mpcgpu_store_import_part(g->ctx[r], (uint32_t)sk0.size(), sk0.data(), sk1.data(), sbytes.data(), soff.data(), g->gbuf[r], g->k0[r], g->k1[r]))
			return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
		return 0;
	});
	if (rc) return rc;
	g->have_store = true;
	return 0;
}

int mpcgpu_group_cons_iter(mpcgpu_group *g)
{
	if (!g) return 1;
	if (!g->have_store) return gfail(g, "mpcgpu_group_cons_iter: no store (call mpcgpu_group_calc_posteriors)");
	const uint32_t R = (uint32_t)g->ctx.size();
	// ---- relax the own shard on every device
	int rc = per_rank(g, [&](uint32_t r) -> int {
		if (mpcgpu_cons_iter(g->ctx[r], g->k0[r], g->k1[r]) || mpcgpu_synchronize(g->ctx[r]))
			return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
		return 0;
	});
	if (rc) return rc;
	if (R == 1) { // ---- the swap of consflat.cpp:22
		if (mpcgpu_cons_commit(g->ctx[0]) || mpcgpu_synchronize(g->ctx[0])) return gfail(g, "rank 0: %s", mpcgpu_last_error(g->ctx[0]));
		return 0;
	}
	// ---- all-gather of the new probabilities, in place in every device's values array (canonical entry order). Every device
	// first queues the swap of consflat.cpp:22 for the slice it relaxed itself — it runs under the exchange — and commits the
	// other devices' slices once they are in: every entry exactly once.
	std::vector<const void *> src(R, nullptr);
	std::vector<void *> dst(R, nullptr);
	std::vector<uint64_t> off(R, 0), bytes(R, 0), first(R, 0), count(R, 0), total(R, 0);
	for (uint32_t r = 0; r < R; ++r) {
		void *vp = nullptr;
		if (mpcgpu_values_info(g->ctx[r], &vp, &total[r]) || mpcgpu_values_slice(g->ctx[r], g->k0[r], g->k1[r], &first[r], &count[r]))
			return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
		dst[r] = vp;
		off[r] = first[r] * 4;
		bytes[r] = count[r] * 4;
		src[r] = (const char *)vp + first[r] * 4;
	}
	rc = per_rank(g, [&](uint32_t r) -> int {
		if (mpcgpu_cons_commit_range(g->ctx[r], first[r], count[r])) return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
		return 0;
	});
	if (rc) return rc;
	rc = all_gather_segments(g, src, dst, off, bytes, true);
	if (rc) return rc;
	return per_rank(g, [&](uint32_t r) -> int {
		if (mpcgpu_cons_commit_range(g->ctx[r], 0, first[r]) ||
		    mpcgpu_cons_commit_range(g->ctx[r], first[r] + count[r], total[r] - first[r] - count[r]) || mpcgpu_synchronize(g->ctx[r]))
			return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
		return 0;
	});
}

} // extern "C"
