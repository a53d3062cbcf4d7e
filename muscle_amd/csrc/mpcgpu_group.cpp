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

#define GHIP(g, call)                                                                                           \
	do {                                                                                                        \
		hipError_t e_ = (call);                                                                                 \
		if (e_ != hipSuccess) return gfail((g), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

// Segment r (seg_bytes[r] bytes at src[r] on rank r's device) goes to dst[d] + seg_off[r] on every rank d. With
// in_place the segment already sits at its final place on its owner (dst[r] + seg_off[r] == src[r]).
int all_gather_segments(mpcgpu_group *g, const std::vector<const void *> &src, const std::vector<void *> &dst,
	const std::vector<uint64_t> &seg_off, const std::vector<uint64_t> &seg_bytes, bool in_place)
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
		for (uint32_t r = 0; r < R; ++r) {
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
		GHIP(g, hipStreamSynchronize(g->xs[r]));
		return 0;
	});
}

} // namespace

extern "C" {

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
	std::vector<uint64_t> cuts;
	shard_bounds(g->len, R, cuts);
	g->k0.assign(cuts.begin(), cuts.end() - 1);
	g->k1.assign(cuts.begin() + 1, cuts.end());
	if (R == 1) { // nothing to exchange
		if (mpcgpu_calc_posteriors(g->ctx[0], g->k0[0], g->k1[0]) || mpcgpu_build_store(g->ctx[0]))
			return gfail(g, "%s", mpcgpu_last_error(g->ctx[0]));
		g->have_store = true;
		return 0;
	}
	// ---- stage A on every device's shard
	std::vector<uint64_t> bytes(R, 0);
	std::vector<const void *> src(R, nullptr);
	int rc = per_rank(g, [&](uint32_t r) -> int {
		void *p = nullptr;
		if (mpcgpu_calc_posteriors(g->ctx[r], g->k0[r], g->k1[r]) || mpcgpu_shard_info(g->ctx[r], &bytes[r], &p) ||
			mpcgpu_synchronize(g->ctx[r]))
			return gfail(g, "rank %u: %s", r, mpcgpu_last_error(g->ctx[r]));
		src[r] = p;
		return 0;
	});
	if (rc) return rc;
	// ---- all-gather of the packed shards: every device gets [shard 0 | shard 1 | ...]
	std::vector<uint64_t> off(R, 0);
	uint64_t total = 0;
	for (uint32_t r = 0; r < R; ++r) { off[r] = total; total += bytes[r]; }
	for (uint32_t r = 0; r < R; ++r) {
		GHIP(g, hipSetDevice(g->dev[r]));
		if (g->gcap[r] < total) {
			if (g->gbuf[r]) GHIP(g, hipFree(g->gbuf[r]));
			g->gbuf[r] = nullptr; g->gcap[r] = 0;
			GHIP(g, hipMalloc(&g->gbuf[r], total + total / 16 + 256));
			g->gcap[r] = total + total / 16 + 256;
		}
	}
	std::vector<void *> dst(g->gbuf.begin(), g->gbuf.end());
	rc = all_gather_segments(g, src, dst, off, bytes, false);
	if (rc) return rc;
	// ---- every device builds its store from the gathered shards
	rc = per_rank(g, [&](uint32_t r) -> int {
		if (mpcgpu_store_import(g->ctx[r], R, g->k0.data(), g->k1.data(), bytes.data(), g->gbuf[r]))
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
