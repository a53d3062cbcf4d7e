// mpcgpu.cpp — host side of libmpcgpu.so: the C ABI of include/mpcgpu.h over the HIP kernels in
// kernels_fb.h (pair-HMM forward/backward, letter and structure-profile emissions), kernels_post.h (probabilities, sparsify, EA),
// kernels_store.h (packed records, padded / slab stores, gather relax, commit, export),
// kernels_relaxv.h (LDS-tiled relax over variable-size records), kernels_aln.h (posterior-DP alignment + traceback) and
// kernels_prog.h (MSA x MSA posterior build). Built by hipcc (-x hip) for gfx950. There is no CPU
// implementation behind this API: without a HIP device mpcgpu_create() fails. rocPRIM's radix sort
// (rocprim::radix_sort_pairs, called directly) is used for one bulk data-movement step (kernels_prog.h); everything else is
// hand-written.
#include "../../include/mpcgpu.h"
#include "kernels_fb.h"
#include "kernels_fbc.h"
#include "kernels_post.h"
#include "kernels_store.h"
#include "kernels_relaxv.h"
#include "kernels_relaxb.h"
#include "kernels_aln.h"
#include "kernels_prog.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

std::string g_create_err;

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	hipError_t ensure(size_t bytes, bool keep = false, hipStream_t st = nullptr)
	{
		if (bytes <= cap) return hipSuccess;
		size_t ncap = keep ? std::max(bytes, cap + cap / 2) : bytes;
		void *np = nullptr;
		hipError_t e = hipMalloc(&np, ncap ? ncap : 1);
		if (e != hipSuccess) return e;
		if (keep && p && cap) {
			e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
			if (e == hipSuccess) e = hipStreamSynchronize(st);
			if (e != hipSuccess) { (void)hipFree(np); return e; }
		}
		if (p) (void)hipFree(p);
		p = np;
		cap = ncap;
		return hipSuccess;
	}
	void release()
	{
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
	}
	// scratch whose size changes from call to call (the joins of a progressive alignment grow): room to spare, so that most calls
	// find it large enough (hipMalloc / hipFree synchronise the device)
	hipError_t ensure_grow(size_t bytes) { return bytes <= cap ? hipSuccess : ensure(std::max(bytes, cap + cap / 2)); }
	template <class T> T *as() const { return (T *)p; }
};

// page-locked host staging: copies to and from it are asynchronous, and it outlives the call that filled it
struct HostBuf {
	void *p = nullptr;
	size_t cap = 0;
	hipError_t ensure(size_t bytes)
	{
		if (bytes <= cap) return hipSuccess;
		const size_t ncap = std::max(bytes, cap + cap / 2);
		void *np = nullptr;
		hipError_t e = hipHostMalloc(&np, ncap ? ncap : 1);
		if (e != hipSuccess) return e;
		if (p) (void)hipHostFree(p);
		p = np;
		cap = ncap;
		return hipSuccess;
	}
	void release()
	{
		if (p) (void)hipHostFree(p);
		p = nullptr;
		cap = 0;
	}
	template <class T> T *as() const { return (T *)p; }
};

struct TimedSpan { hipEvent_t a, b; int fam; };

} // namespace

struct mpcgpu_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipDeviceProp_t prop;
	std::string err;

	// HMM (row 0)
	bool have_hmm = false;
	float start[5], trans[25], thr = 0;
	std::vector<float> match256, ins256;
	int use_fma = 1;

	// sequences
	u32 n = 0;
	std::vector<std::vector<u8>> raw;
	std::vector<u32> len;
	int A = 0;
	int code_of[256];
	DevBuf d_seq_code, d_seq_off, d_seq_len, d_match, d_ins;
	// structure-profile emissions (mpcgpu_set_mega); reset by every set_seqs
	bool have_mega = false;
	u32 mg_nfeat = 0, mg_tab_floats = 0, mg_base[MPC_MEGA_FMAX], mg_alpha[MPC_MEGA_FMAX];
	DevBuf d_mg_prof, d_mg_ins, d_mg_tab, d_mg_in;
	u64 npairs = 0;
	DevBuf d_pair_x, d_pair_y; // all pairs
	std::vector<u32> h_pair_x, h_pair_y;
	// pair order (mpcgpu_set_pair_order): empty = MPCFlat::InitPairs order. order_rects: 4 words per rectangle {xa, xb, ya, yb},
	// order_base[r] = position of its first pair; d_rects: 6 words per rectangle for the kernels (StoreParams::rects); ext2pos[k] =
	// position of InitPairs pair k (the getters speak InitPairs numbers, the sharding calls positions)
	std::vector<u32> order_rects;
	std::vector<u64> order_base;
	std::vector<u32> ext2pos;
	DevBuf d_rects;
	// partial store (mpcgpu_store_import_part): the sequences whose records exist, and the positions this context relaxes
	bool partial = false;
	bool packed_stale = false; // commits since the import wrote the packed records of the own pairs only (refresh_packed)
	std::vector<std::pair<u64, u64> > lazy_ranges; // ... for these ranges of entries [first, end), merged
	std::vector<u8> need;
	DevBuf d_need;
	u64 own_k0 = 0, own_k1 = 0;

	// shard state (stage A output of this context)
	bool have_shard = false;
	bool shard_is_list = false; // the shard holds an explicit pair list (mpcgpu_align_msas), not a range of InitPairs
	u64 sh_k0 = 0, sh_k1 = 0;
	DevBuf d_shard;             // [header][records]
	u64 shard_bytes = 0;
	std::vector<u32> sh_nnz;
	std::vector<float> sh_ea;

	// store state (all pairs)
	bool have_store = false;
	const u32 *st_packed = nullptr; // device: words base that pbase refers to
	std::vector<u64> h_pbase, h_vbase;
	std::vector<u32> all_nnz;
	std::vector<float> all_ea;
	DevBuf d_pbase, d_vbase, d_rp, d_rp_base, d_ent, d_ent_base, d_mbase, d_vnext, d_own_packed;
	u64 total_entries = 0;
	u32 max_nnz = 0, max_len = 0;
	HostBuf h_bp_in, h_aln_res;
	size_t aln_smem_set[4] = {0, 0, 0, 0}; // largest dynamic LDS each CalcAlnFlat kernel has been allowed so far
	DevBuf d_tile_next, d_bp_in, d_aln_res, d_bp_seq, d_bp_map, d_bp_off, d_bp_coff, d_bp_keys, d_bp_vals, d_bp_tmp, d_bp_runs;
	DevBuf d_tiles, d_pad, d_pos, d_aln_post, d_aln_tb, d_aln_rev;
	bool have_pad = false;       // variable-size dense records + relax_var_kernel (else: slabs + gather relax)
	u32 pad_lcap1 = 0;           // longest sequence (LDS scratch of var_build_kernel)
	DevBuf d_rec_off, d_sizes, d_tilefit;
	u32 var_max_rec_blocks = 0;  // largest record, 16-byte blocks
	u64 var_total_blocks = 0;
	u32 var_threads = 1024, var_nbuf = 2, var_buf_bytes = 0;
	std::string store_desc, tiles_desc, relax_kernel_name; // mpcgpu_relax_info
	bool relax_fallback = false;
	// tile list of the LDS-tiled relax, cached per pair range (the sparsity pattern is frozen)
	std::vector<u32> h_tiles;
	std::vector<u32> h_tiles2; // tiles of the pairs that only fit the one-workgroup-per-CU geometry (var_mixed)
	DevBuf d_tiles2;
	bool var_mixed = false;    // two launches per relax: the configured geometry + 1 x 1024 threads / 160 KB for what does not fit it
	u64 tiles_k0 = ~0ull, tiles_k1 = ~0ull;
	u32 tiles_bx = 0, tiles_by = 0;
	// band tiles (relax_band_kernel, kernels_relaxb.h): band tables of the store, the tile list of the cached pair range
	bool band_ok = false;      // the band tables exist for this store
	bool var_pairs_ok = true;  // every pair's two whole records fit a tile of relax_var_kernel (else: band tiles or nothing)
	u32 band_nb1 = 0;
	DevBuf d_ovf_off, d_cell_off, d_yr, d_ovf_sum, d_ovf_maxc, d_btiles, d_bt_out, d_bt_count, d_bt_list;
	std::vector<u32> h_btiles;
	u64 btiles_k0 = ~0ull, btiles_k1 = ~0ull;
	bool win_ok = false;       // window records exist for this store (the direct-index merge: kernels_relaxb.h)
	DevBuf d_win, d_wrec_off, d_wv_off, d_pos_w, d_wsum, d_wmaxc, d_wflag;
	u64 win_total_blocks = 0;
	const void *band_fn = nullptr; // relax_band: kernel, LDS size and occupancy of the last launch (runtime queries cached)
	size_t band_smem = 0;
	int band_occ = 0;

	// scratch
	DevBuf d_bnd;
	DevBuf d_queue, d_order, d_bx, d_by, d_fm, d_cand, d_cand_cnt, d_total, d_res, d_nnz, d_ea, d_flags,
		d_sort_scratch, d_srow_scratch, d_dstbase, d_recwords, d_exp_off, d_exp_val, d_exp_offbase, d_exp_klist, d_exp_valbase;

	// measurement
	std::vector<TimedSpan> spans;
	bool timing = true; // mpcgpu_timers_enable
	float ms[MPCGPU_NKERNELS] = {0};
	u64 launches[MPCGPU_NKERNELS] = {0};
	u64 work_cells = 0, work_entry_z = 0;
	u64 last_post_cells = 0; // cells of the dense matrix in d_aln_post (mpcgpu_get_last_post)
	u64 sa_b0 = 0, sa_B = 0;  // the last stage-A batch: first pair, pairs; candidate capacity, finishing kernel, key layout
	u32 sa_capc = 0, sa_long_min = 0;
	bool sa_post_rows = false;
	std::vector<u32> list_x, list_y; // the pairs of the last list stage
	// mpcgpu_align_pairs runs its list in chunks (and halves a chunk that stage A had to split): the caller's WHOLE list and the
	// index of the first pair of the chunk the last stage A ran on — mpcgpu_get_list_sparse(q) indexes the caller's list
	std::vector<u32> ap_x, ap_y;
	u32 list_q0 = 0;
	bool ap_keep = false; // set around the stage_a calls that serve the align-pairs list (every other list stage forgets it)
	HostBuf h_ap;             // mpcgpu_align_pairs: kernel parameters and results, page-locked
	// Host vectors of megabytes that take part in device copies are KEPT (members, not locals). glibc gives such a block back to the
	// kernel when it is freed (munmap above the mmap threshold, or a trim of the heap's top: both thresholds move at run time), the
	// runtime had it registered for DMA (a pageable source / destination of hipMemcpyAsync), and the kernel driver answers the MMU
	// invalidation by EVICTING and later restoring the process's queues: the next dispatch starts 10 - 30 ms late. Measured with
	// in-kernel clocks (profiles/r10k_rank_time.log): the first kernel after the store build ran 12 - 27 ms after its launch in some
	// processes, every or every other step — the 4 MB offset tables of build_var_store died at its return. (Telling the allocator
	// to keep everything mapped — mallopt — removes it too; the library leaves the allocator to its host program, the drop-in binary
	// sets it: no cost to -align, profiles/r11i_align_malloc_ab.log.)
	std::vector<u32> v_flags;               // stage A: per-pair flags of a batch as they are read back
	std::vector<u64> v_dstbase, v_recw;     // stage A: where a batch's records go in the shard
	std::vector<u8> v_shdr;                 // stage A: the shard's header as it is uploaded
	std::vector<u8> v_hdr;                  // mpcgpu_store_import: a shard's header as it is read back
	std::vector<u32> v_off, v_woff;         // build_var_store: block offsets of the n x n records / window records
	std::vector<u32> v_words, v_out, v_w2, v_o2, v_okw; // relax_band: tile words and their statistics while the tiles are cut
	HostBuf h_bt;             // relax_band's tile cutter: the small host <-> device transfers of a cut, page-locked (a copy into pageable memory
	                          // right after a launch was measured at 24 ms on an otherwise idle device: profiles/r10k)
	DevBuf d_ap_off;
	DevBuf d_chain_first, d_chain_cnt; // fb_chain_kernel's work list (kernels_fbc.h)
	// stage A's batch pipeline (mpcgpu_stage_a.inc): the index arrays of the NEXT batch (its sweeps are queued while this batch's sizes
	// are on their way to the host), the stream those sizes come back on and the event behind the finishing kernel
	DevBuf d_bx_n, d_by_n, d_order_n, d_chain_first_n, d_chain_cnt_n;
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_post = nullptr;
	u64 sa_pairs = 0, sa_chained = 0, sa_chains = 0; // last stage A: pairs, pairs that ran in chains, chains
	double aa_trace_t[5] = {0, 0, 0, 0, 0}; // MPCGPU_TRACE & 4: host seconds of mpcgpu_align_alns' phases
	u64 aa_trace_n = 0;
};

namespace {

int fail(mpcgpu_ctx *c, const char *fmt, ...)
{
	char buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (c) c->err = buf; else g_create_err = buf;
	return 1;
}

#define HIPCHK(c, call)                                                                          \
	do {                                                                                         \
		hipError_t e_ = (call);                                                                  \
		if (e_ != hipSuccess)                                                                    \
			return fail((c), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
	} while (0)

// MPCGPU_TRACE=1: every timed span is synchronised and reported on stderr as it completes
// (diagnostics only: serialises the stream).
// MPCGPU_TRACE: bit 0 = the spans and decisions above; bit 1 (MPCGPU_TRACE=3) = also the first 64 band tiles; bit 2 (MPCGPU_TRACE=4 or 5)
// = host wall time of the phases of stage A and of mpcgpu_align_alns (rounds 3-5 had a knob for each).
int trace_level()
{
	static int lv = -1;
	if (lv < 0) { const char *s = getenv("MPCGPU_TRACE"); lv = (s && *s) ? atoi(s) : 0; if (lv < 0) lv = 0; }
	return lv;
}
bool trace_on() { return (trace_level() & 1) != 0; }
bool trace_host() { return (trace_level() & 4) != 0; }

int span_begin(mpcgpu_ctx *c, int fam, TimedSpan *sp)
{
	sp->fam = (c->timing || trace_on()) ? fam : -1;
	if (sp->fam < 0) return 0;
	if (trace_on()) { fprintf(stderr, "[mpcgpu] launch family %d ...\n", fam); fflush(stderr); }
	HIPCHK(c, hipEventCreate(&sp->a));
	HIPCHK(c, hipEventCreate(&sp->b));
	HIPCHK(c, hipEventRecord(sp->a, c->stream));
	return 0;
}
int span_end(mpcgpu_ctx *c, TimedSpan *sp)
{
	if (sp->fam < 0) return 0;
	HIPCHK(c, hipEventRecord(sp->b, c->stream));
	if (trace_on()) { // before the events can be folded away below
		HIPCHK(c, hipStreamSynchronize(c->stream));
		float t = 0;
		HIPCHK(c, hipEventElapsedTime(&t, sp->a, sp->b));
		fprintf(stderr, "[mpcgpu] family %d done: %.3f ms\n", sp->fam, t);
		fflush(stderr);
	}
	c->spans.push_back(*sp);
	c->launches[sp->fam] += 1;
	// a caller that never reads the timers (the drop-in: thousands of joins) must not accumulate events: fold the
	// finished spans into the totals now and then, without waiting for the running ones
	if (c->spans.size() >= 64) {
		size_t keep = 0;
		for (size_t q = 0; q < c->spans.size(); ++q) {
			TimedSpan &t = c->spans[q];
			float ms = 0;
			if (hipEventQuery(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
				c->ms[t.fam] += ms;
				(void)hipEventDestroy(t.a);
				(void)hipEventDestroy(t.b);
			} else c->spans[keep++] = t;
		}
		(void)hipGetLastError(); // hipErrorNotReady of the queries is not an error
		c->spans.resize(keep);
	}
	return 0;
}
int spans_collect(mpcgpu_ctx *c)
{
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for (auto &sp : c->spans) {
		float t = 0;
		HIPCHK(c, hipEventElapsedTime(&t, sp.a, sp.b));
		c->ms[sp.fam] += t;
		(void)hipEventDestroy(sp.a);
		(void)hipEventDestroy(sp.b);
	}
	c->spans.clear();
	return 0;
}

template <class T> int upload(mpcgpu_ctx *c, DevBuf &b, const std::vector<T> &v)
{
	HIPCHK(c, b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T)));
	if (!v.empty())
		HIPCHK(c, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
	return 0;
}

u32 next_pow2(u32 v)
{
	u32 p = 1;
	while (p < v) p <<= 1;
	return p;
}

int env_int(const char *name, int dflt)
{
	const char *s = getenv(name);
	return (s && *s) ? atoi(s) : dflt;
}

// Dynamic LDS beyond the default 64 KB (alphabets of more than ~125 compacted letters: (A*A + A) floats; gfx950 has 160 KB per CU)
// has to be asked for per kernel function; asked once per function and size (the call costs a good fraction of a millisecond).
void ensure_dyn_smem(const void *fn, size_t smem)
{
	if (smem <= 64u * 1024u) return;
	static std::mutex mu;
	static std::map<std::pair<int, const void *>, size_t> have; // the attribute is per DEVICE and function (several devices in one process: mpcgpu_group, MUSCLE_GPU_DEVICES)
	std::lock_guard<std::mutex> lk(mu);
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
	size_t &h = have[std::make_pair(dev, fn)];
	if (h >= smem) return;
	if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) == hipSuccess) h = smem;
	else (void)hipGetLastError();
}

template <int H, bool MEGA, bool LONG = false> void launch_fb(const FbParams &p, u32 grid, u32 block, size_t smem, hipStream_t st)
{
	auto kern = fb_kernel<H, MEGA, LONG>;
	ensure_dyn_smem((const void *)kern, smem);
	MPC_LAUNCH(kern, grid, block, smem, st, p);
}

template <int H, bool MEGA, bool LONG = false> int occ_fb(u32 block, size_t smem)
{
	int nb = 0;
	ensure_dyn_smem((const void *)fb_kernel<H, MEGA, LONG>, smem);
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fb_kernel<H, MEGA, LONG>, (int)block, smem) != hipSuccess || nb < 1)
		nb = 1;
	return nb;
}

// Row-block kernels (sequences X longer than 64*MPC_HMAX): H = MPC_LONG_H rows per lane; H = 1 exists so that
// tests reach several blocks with short sequences (MPCGPU_FB_LONG_H=1 MPCGPU_FB_LONG_MIN=<rows>).
#define MPC_LONG_H 7
#define MPC_LONG_H_SMALL 4 // 166 VGPRs (3 waves per SIMD) where H = 7 needs 217 (2): used when memory lets more than 2 waves per SIMD be resident
int occ_fb_long(int H, bool mega, u32 block, size_t smem)
{
	if (H == 1) return mega ? occ_fb<1, true, true>(block, smem) : occ_fb<1, false, true>(block, smem);
	if (H == MPC_LONG_H_SMALL) return mega ? occ_fb<MPC_LONG_H_SMALL, true, true>(block, smem) : occ_fb<MPC_LONG_H_SMALL, false, true>(block, smem);
	return mega ? occ_fb<MPC_LONG_H, true, true>(block, smem) : occ_fb<MPC_LONG_H, false, true>(block, smem);
}

void launch_fb_long(int H, bool mega, const FbParams &p, u32 grid, u32 block, size_t smem, hipStream_t st)
{
	if (H == 1) { if (mega) launch_fb<1, true, true>(p, grid, block, smem, st); else launch_fb<1, false, true>(p, grid, block, smem, st); }
	else if (H == MPC_LONG_H_SMALL) { if (mega) launch_fb<MPC_LONG_H_SMALL, true, true>(p, grid, block, smem, st); else launch_fb<MPC_LONG_H_SMALL, false, true>(p, grid, block, smem, st); }
	else if (mega) launch_fb<MPC_LONG_H, true, true>(p, grid, block, smem, st);
	else launch_fb<MPC_LONG_H, false, true>(p, grid, block, smem, st);
}

int occ_fb_h(int H, bool mega, u32 block, size_t smem)
{
	switch (H) {
#define MPC_CASE(h) case h: return mega ? occ_fb<h, true>(block, smem) : occ_fb<h, false>(block, smem);
	MPC_CASE(1) MPC_CASE(2) MPC_CASE(3) MPC_CASE(4) MPC_CASE(5) MPC_CASE(6) MPC_CASE(7) MPC_CASE(8)
	MPC_CASE(9) MPC_CASE(10) MPC_CASE(11) MPC_CASE(12) MPC_CASE(13) MPC_CASE(14) MPC_CASE(15) MPC_CASE(16)
#undef MPC_CASE
	default: return 1;
	}
}

void launch_fb_h(int H, bool mega, const FbParams &p, u32 grid, u32 block, size_t smem, hipStream_t st)
{
	switch (H) {
#define MPC_CASE(h) case h: if (mega) launch_fb<h, true>(p, grid, block, smem, st); else launch_fb<h, false>(p, grid, block, smem, st); break;
	MPC_CASE(1) MPC_CASE(2) MPC_CASE(3) MPC_CASE(4) MPC_CASE(5) MPC_CASE(6) MPC_CASE(7) MPC_CASE(8)
	MPC_CASE(9) MPC_CASE(10) MPC_CASE(11) MPC_CASE(12) MPC_CASE(13) MPC_CASE(14) MPC_CASE(15) MPC_CASE(16)
#undef MPC_CASE
	default: break;
	}
}

// fb_chain_kernel (kernels_fbc.h): chains of pairs that share their row sequence
template <int H> void launch_fbc(const FbChainParams &p, u32 grid, u32 block, size_t smem, hipStream_t st)
{
	auto kern = fb_chain_kernel<H>;
	ensure_dyn_smem((const void *)kern, smem);
	MPC_LAUNCH(kern, grid, block, smem, st, p);
}

template <int H> int occ_fbc(u32 block, size_t smem)
{
	int nb = 0;
	ensure_dyn_smem((const void *)fb_chain_kernel<H>, smem);
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)fb_chain_kernel<H>, (int)block, smem) != hipSuccess || nb < 1)
		nb = 1;
	return nb;
}

int occ_fbc_h(int H, u32 block, size_t smem)
{
	switch (H) {
#define MPC_CASE(h) case h: return occ_fbc<h>(block, smem);
	MPC_CASE(1) MPC_CASE(2) MPC_CASE(3) MPC_CASE(4) MPC_CASE(5) MPC_CASE(6) MPC_CASE(7) MPC_CASE(8)
	MPC_CASE(9) MPC_CASE(10) MPC_CASE(11) MPC_CASE(12) MPC_CASE(13) MPC_CASE(14) MPC_CASE(15) MPC_CASE(16)
#undef MPC_CASE
	default: return 1;
	}
}

void launch_fbc_h(int H, const FbChainParams &p, u32 grid, u32 block, size_t smem, hipStream_t st)
{
	switch (H) {
#define MPC_CASE(h) case h: launch_fbc<h>(p, grid, block, smem, st); break;
	MPC_CASE(1) MPC_CASE(2) MPC_CASE(3) MPC_CASE(4) MPC_CASE(5) MPC_CASE(6) MPC_CASE(7) MPC_CASE(8)
	MPC_CASE(9) MPC_CASE(10) MPC_CASE(11) MPC_CASE(12) MPC_CASE(13) MPC_CASE(14) MPC_CASE(15) MPC_CASE(16)
#undef MPC_CASE
	default: break;
	}
}

// shard buffer layout: [u64 npairs][u64 record_words_total][u32 nnz[np]][f32 ea[np]] pad to 8 | records
u64 shard_header_bytes(u64 np) { return ((16 + np * 8 + 7) / 8) * 8; }
u64 rec_words(u32 LX, u32 LY, u32 nnz) { return (u64)LX + LY + 4 * (u64)nnz; }

void fill_store_params(mpcgpu_ctx *c, StoreParams &s)
{
	s.n = c->n;
	s.seq_len = c->d_seq_len.as<u32>();
	s.npairs = c->npairs;
	s.pair_x = c->d_pair_x.as<u32>();
	s.pair_y = c->d_pair_y.as<u32>();
	s.packed = (u32 *)c->st_packed;
	s.pbase = c->d_pbase.as<u64>();
	s.vbase = c->d_vbase.as<u64>();
	s.rp = c->d_rp.as<u32>();
	s.rp_base = c->d_rp_base.as<u64>();
	s.ent = c->d_ent.as<MpcEnt>();
	s.ent_base = c->d_ent_base.as<u64>();
	s.mbase = c->d_mbase.as<u32>();
	s.vnext = c->d_vnext.as<float>();
	s.pad = c->d_pad.as<u32>();
	s.lcap1 = c->pad_lcap1;
	s.pos_f = c->d_pos.as<unsigned short>();
	s.pos_t = c->d_pos.as<unsigned short>() + c->total_entries;
	s.rec_off = c->d_rec_off.as<u32>();
	s.ovf_off = c->band_ok ? c->d_ovf_off.as<u32>() : nullptr;
	s.nb1 = c->band_nb1;
	s.win = c->win_ok ? c->d_win.as<u32>() : nullptr;
	s.wrec_off = c->d_wrec_off.as<u32>();
	s.wv_off = c->win_ok ? c->d_wv_off.as<u32>() : nullptr;
	s.pos_wf = c->d_pos_w.as<unsigned short>();
	s.pos_wt = c->d_pos_w.as<unsigned short>() + c->total_entries;
	s.nrect = (u32)(c->order_rects.size() / 4);
	s.rects = s.nrect ? c->d_rects.as<u32>() : nullptr;
	s.need = c->partial ? c->d_need.as<u8>() : nullptr;
}

// position of pair (X,Y), X < Y, in the context's pair order (the host twin of mpc_pair_pos, kernels_store.h)
u64 pair_pos(const mpcgpu_ctx *c, u32 X, u32 Y)
{
	const u32 n = c->n;
	auto tri = [](u32 m, u32 i, u32 j) { return (u64)i * m - ((u64)i * (i + 1)) / 2 + (j - i - 1); }; // InitPairs order (mpcflat.cpp:145-155)
	if (c->order_rects.empty()) return tri(n, X, Y);
	for (size_t r = 0; r < c->order_rects.size() / 4; ++r) {
		const u32 *q = &c->order_rects[4 * r];
		if (X < q[0] || X >= q[1] || Y < q[2] || Y >= q[3]) continue;
		return q[2] >= q[1] ? c->order_base[r] + (u64)(X - q[0]) * (q[3] - q[2]) + (Y - q[2]) : c->order_base[r] + tri(q[1] - q[0], X - q[0], Y - q[0]);
	}
	return ~0ull;
}

#include "mpcgpu_relax.inc"
#include "mpcgpu_store.inc"
} // namespace

// The fallback layout: compact CSR slabs per sequence + the gather kernel (relax_kernel). Built from the packed records, which
// always hold the current values: also the way out when a store of dense records turns out not to tile (mpcgpu_cons_iter).
// The packed records of the pairs a sharded context does not relax itself are committed lazily (commit_pairs_kernel): before anything
// reads them — the getters, a rebuild of the records — they take their values from the values array, which holds the committed
// probability of every such pair from the last exchange on (mpcgpu_cons_iter writes the context's own slice only).
static int refresh_packed(mpcgpu_ctx *c)
{
	if (!c->packed_stale) return 0;
	StoreParams sp;
	fill_store_params(c, sp);
	// exactly the ranges of entries that were committed since the last refresh (a caller that has committed its own slice only and
	// asks for a foreign pair's matrix gets that pair's LAST committed values, not whatever the values array holds beyond them)
	for (const std::pair<u64, u64> &r : c->lazy_ranges) {
		const u64 ka = (u64)(std::upper_bound(c->h_vbase.begin(), c->h_vbase.end(), r.first) - c->h_vbase.begin()) - 1;
		const u64 kb = std::min<u64>((u64)(std::lower_bound(c->h_vbase.begin(), c->h_vbase.end(), r.second) - c->h_vbase.begin()), c->npairs);
		if (kb <= ka) continue;
		MPC_LAUNCH(packed_refresh_kernel, (u32)std::min<u64>(kb - ka, (u64)c->prop.multiProcessorCount * 64), 64, 0, c->stream, sp, ka, kb, r.first, r.second,
			c->own_k0, c->own_k1);
		HIPCHK(c, hipGetLastError());
	}
	c->lazy_ranges.clear();
	c->packed_stale = false;
	return 0;
}

static int build_slab_store(mpcgpu_ctx *c)
{
	const u32 n = c->n;
	if (refresh_packed(c)) return 1; // (the slabs are built from the packed records)
	c->own_k0 = 0; c->own_k1 = c->npairs; // (and hold everything: commits are complete from here on)
	c->have_pad = false;
	c->band_ok = false;
	c->partial = false; // (the slabs hold every sequence: a partial store that falls back here is a complete one)
	TimedSpan ts;
	c->d_pad.release();
	c->d_pos.release();
	// the band and window tables of a dense-record store that turned out not to tile go with it (nn * nb1 words each: GBs)
	c->win_ok = false;
	c->d_ovf_off.release(); c->d_cell_off.release(); c->d_yr.release(); c->d_ovf_sum.release(); c->d_ovf_maxc.release();
	c->d_win.release(); c->d_pos_w.release(); c->d_wv_off.release(); c->d_wsum.release(); c->d_wmaxc.release(); c->d_wrec_off.release(); c->d_rec_off.release();
	c->d_btiles.release();
	{
		const char *rm = getenv("MPCGPU_RELAX");
		c->store_desc = "CSR slabs per sequence; relax_kernel (one thread per stored cell gathers its rows from HBM: the slow path, ~5x the LDS-tiled kernels)";
		c->tiles_desc.clear();
		c->relax_kernel_name = "relax_kernel";
		c->relax_fallback = !(rm && !strcmp(rm, "gather"));
	}
	// ---- slab geometry: per ordered pair entry counts -> mbase (within slab), slab bases
	std::vector<u32> mbase((size_t)n * (n + 1), 0);
	std::vector<u64> ent_base(n + 1, 0), rp_base(n + 1, 0);
	{
		std::vector<u64> slab(n, 0);
		// nnz(A,Z) = nnz of the unordered pair
		for (u32 A = 0; A < n; ++A) {
			u64 run = 0;
			for (u32 Z = 0; Z < n; ++Z) {
				if (run > 0xffffffffull) return fail(c, "mpcgpu_store_import: slab of sequence %u exceeds 2^32 entries", A);
				mbase[(size_t)A * (n + 1) + Z] = (u32)run;
				if (Z != A) {
					const u64 k = A < Z ? pair_pos(c, A, Z) : pair_pos(c, Z, A);
					run += c->all_nnz[k];
				}
			}
			if (run > 0xffffffffull) return fail(c, "mpcgpu_store_import: slab of sequence %u exceeds 2^32 entries", A);
			mbase[(size_t)A * (n + 1) + n] = (u32)run;
			slab[A] = run;
		}
		for (u32 A = 0; A < n; ++A) {
			ent_base[A + 1] = ent_base[A] + slab[A];
			rp_base[A + 1] = rp_base[A] + (u64)n * (c->len[A] + 1);
		}
	}
	HIPCHK(c, c->d_rp.ensure(rp_base[n] * 4));
	HIPCHK(c, c->d_ent.ensure(std::max<u64>(ent_base[n], 1) * 8));
	if (upload(c, c->d_mbase, mbase) || upload(c, c->d_ent_base, ent_base) || upload(c, c->d_rp_base, rp_base)) return 1;
	StoreParams sp;
	fill_store_params(c, sp);
	if (span_begin(c, 2, &ts)) return 1;
	const u64 blocks = (u64)n * n;
	MPC_LAUNCH(slab_build_kernel, (u32)std::min<u64>(blocks, (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &ts)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->have_store = true;
	return 0;
}

extern "C" {

const char *mpcgpu_version(void) { return MPC_VERSION_STRING; }

const char *mpcgpu_last_error(const mpcgpu_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mpcgpu_create(mpcgpu_ctx **out, int device_ordinal)
{
	if (!out) return fail(nullptr, "mpcgpu_create: out is NULL");
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(nullptr, "mpcgpu_create: no HIP device available (%s); this library has no CPU path",
			e == hipSuccess ? "device count 0" : hipGetErrorString(e));
	if (device_ordinal < 0 || device_ordinal >= ndev)
		return fail(nullptr, "mpcgpu_create: device %d out of range (have %d)", device_ordinal, ndev);
	mpcgpu_ctx *c = new mpcgpu_ctx;
	c->device = device_ordinal;
	if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&c->prop, device_ordinal) != hipSuccess ||
		hipStreamCreate(&c->stream) != hipSuccess) {
		delete c;
		return fail(nullptr, "mpcgpu_create: cannot initialise device %d", device_ordinal);
	}
	for (int i = 0; i < 256; ++i) c->code_of[i] = -1;
	*out = c;
	return 0;
}

void mpcgpu_destroy(mpcgpu_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (auto &sp : c->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
	DevBuf *all[] = {&c->d_seq_code, &c->d_seq_off, &c->d_seq_len, &c->d_match, &c->d_ins, &c->d_pair_x, &c->d_pair_y,
		&c->d_mg_prof, &c->d_mg_ins, &c->d_mg_tab, &c->d_mg_in, &c->d_bnd,
		&c->d_shard, &c->d_pbase, &c->d_vbase, &c->d_rp, &c->d_rp_base, &c->d_ent, &c->d_ent_base, &c->d_mbase,
		&c->d_vnext, &c->d_own_packed, &c->d_queue, &c->d_order, &c->d_bx, &c->d_by, &c->d_fm, &c->d_cand,
		&c->d_cand_cnt, &c->d_total, &c->d_res, &c->d_nnz, &c->d_ea, &c->d_flags, &c->d_sort_scratch,
		&c->d_srow_scratch, &c->d_dstbase, &c->d_recwords, &c->d_exp_off, &c->d_exp_val, &c->d_exp_offbase, &c->d_tiles, &c->d_pad, &c->d_pos, &c->d_bp_seq, &c->d_bp_map, &c->d_bp_off, &c->d_bp_coff,
		&c->d_tile_next, &c->d_bp_in, &c->d_aln_res, &c->d_bp_keys, &c->d_bp_vals, &c->d_bp_tmp, &c->d_bp_runs, &c->d_aln_post, &c->d_aln_tb, &c->d_aln_rev,
		&c->d_rec_off, &c->d_sizes, &c->d_tilefit};
	for (DevBuf *b : all) b->release();
	c->h_bp_in.release();
	c->h_aln_res.release();
	c->h_ap.release();
	c->d_ap_off.release();
	c->d_chain_first.release(); c->d_chain_cnt.release();
	c->d_bx_n.release(); c->d_by_n.release(); c->d_order_n.release(); c->d_chain_first_n.release(); c->d_chain_cnt_n.release();
	c->d_rects.release(); c->d_need.release(); c->d_exp_klist.release(); c->d_exp_valbase.release();
	c->d_tiles2.release();
	if (c->ev_post) (void)hipEventDestroy(c->ev_post);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	(void)hipStreamDestroy(c->stream);
	delete c;
}

int mpcgpu_synchronize(mpcgpu_ctx *c)
{
	if (!c) return 1;
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_set_hmm(mpcgpu_ctx *c, const float start[5], const float trans[25], const float match[65536],
	const float ins[256], float min_sparse_score, int expf_variant)
{
	if (!c) return 1;
	if (!start || !trans || !match || !ins) return fail(c, "mpcgpu_set_hmm: NULL table");
	memcpy(c->start, start, sizeof(c->start));
	memcpy(c->trans, trans, sizeof(c->trans));
	c->match256.assign(match, match + 65536);
	c->ins256.assign(ins, ins + 256);
	c->thr = min_sparse_score;
	if (expf_variant < 0) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
		__builtin_cpu_init();
		c->use_fma = (__builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2")) ? 1 : 0; // glibc ifunc-fma.h
#else
		c->use_fma = 0;
#endif
	} else
		c->use_fma = expf_variant ? 1 : 0;
	c->have_hmm = true;
	c->have_shard = c->have_store = false;
	return 0;
}

static int set_seqs_impl(mpcgpu_ctx *c, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens, bool with_pairs)
{
	if (!c) return 1;
	if (!c->have_hmm) return fail(c, "mpcgpu_set_seqs: call mpcgpu_set_hmm first");
	if (n < 2) return fail(c, "mpcgpu_set_seqs: need at least 2 sequences (got %u)", n);
	if (with_pairs && (u64)n * n > 0xffffffffull) return fail(c, "mpcgpu_set_seqs: too many sequences (%u)", n);
	HIPCHK(c, hipSetDevice(c->device));
	c->have_shard = c->have_store = false;
	c->have_mega = false;
	c->n = n;
	c->raw.assign(n, {});
	c->len.assign(lens, lens + n);
	for (int i = 0; i < 256; ++i) c->code_of[i] = -1;
	c->A = 0;
	u32 maxl = 0, max2 = 0;
	for (u32 i = 0; i < n; ++i) {
		if (lens[i] == 0) return fail(c, "mpcgpu_set_seqs: sequence %u is empty", i);
		if (lens[i] > MPC_KEY_COL_MASK) return fail(c, "mpcgpu_set_seqs: sequence %u is longer than %u", i, MPC_KEY_COL_MASK);
		c->raw[i].assign(seqs[i], seqs[i] + lens[i]);
		for (u32 at = 0; at < lens[i]; ++at) {
			const u8 b = c->raw[i][at];
			// (the reference indexes m_MatchScore[256][256] by a plain `char`, fwdflat3.cpp:102-109: bytes >= 128 are negative indices
			// there — undefined behaviour, not a feature; all 128 seven-bit values are taken)
			if (b >= 128) return fail(c, "mpcgpu_set_seqs: sequence %u holds non-ASCII byte %u at position %u", i, (unsigned)b, at);
			if (c->code_of[b] < 0) {
				c->code_of[b] = c->A++;
			}
		}
		if (lens[i] > maxl) { max2 = maxl; maxl = lens[i]; } else if (lens[i] > max2) max2 = lens[i];
	}
	// calcposteriorflat.cpp:54-61
	if (double(maxl) * double(max2) * 5 + 100 > double(INT_MAX))
		return fail(c, "HMM overflow, sequence lengths %u, %u (max ~21k)", maxl, max2);
	std::vector<u8> code;
	std::vector<u64> off(n + 1, 0);
	for (u32 i = 0; i < n; ++i) {
		off[i] = code.size();
		for (u8 b : c->raw[i]) code.push_back((u8)c->code_of[b]);
	}
	off[n] = code.size();
	std::vector<float> cm((size_t)c->A * c->A), ci(c->A);
	int byte_of[128];
	for (int b = 0; b < 256; ++b) if (c->code_of[b] >= 0) byte_of[c->code_of[b]] = b;
	for (int a = 0; a < c->A; ++a) {
		ci[a] = c->ins256[byte_of[a]];
		for (int b = 0; b < c->A; ++b) cm[(size_t)a * c->A + b] = c->match256[(size_t)byte_of[a] * 256 + byte_of[b]];
	}
	if (upload(c, c->d_seq_code, code) || upload(c, c->d_seq_off, off) || upload(c, c->d_seq_len, c->len) ||
		upload(c, c->d_match, cm) || upload(c, c->d_ins, ci))
		return 1;
	c->npairs = 0;
	c->h_pair_x.clear(); c->h_pair_y.clear();
	c->order_rects.clear(); c->order_base.clear(); c->ext2pos.clear(); // (a pair order belongs to one set of sequences)
	c->partial = false;
	if (!with_pairs) { // explicit pair lists only (mpcgpu_align_msas)
		HIPCHK(c, hipStreamSynchronize(c->stream));
		return 0;
	}
	// MPCFlat::InitPairs (mpcflat.cpp:139-159)
	c->npairs = (u64)n * (n - 1) / 2;
	c->h_pair_x.resize(c->npairs);
	c->h_pair_y.resize(c->npairs);
	u64 k = 0;
	for (u32 i = 0; i < n; ++i)
		for (u32 j = i + 1; j < n; ++j) { c->h_pair_x[k] = i; c->h_pair_y[k] = j; ++k; }
	if (upload(c, c->d_pair_x, c->h_pair_x) || upload(c, c->d_pair_y, c->h_pair_y)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_set_seqs(mpcgpu_ctx *c, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens)
{
	return set_seqs_impl(c, n, seqs, lens, true);
}

int mpcgpu_set_seqs_registry(mpcgpu_ctx *c, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens)
{
	return set_seqs_impl(c, n, seqs, lens, false);
}

int mpcgpu_set_pair_order(mpcgpu_ctx *c, uint32_t nrects, const uint32_t *rects)
{
	if (!c) return 1;
	if (c->n == 0 || c->npairs == 0) return fail(c, "mpcgpu_set_pair_order: call mpcgpu_set_seqs first");
	if (nrects && !rects) return fail(c, "mpcgpu_set_pair_order: no rectangles");
	HIPCHK(c, hipSetDevice(c->device));
	const u32 n = c->n;
	c->have_shard = c->have_store = false;
	c->partial = false;
	if (c->order_rects.size() == 4 * (size_t)nrects && (nrects == 0 || !memcmp(c->order_rects.data(), rects, 16 * (size_t)nrects)))
		return 0; // the order the context already has (mpcgpu_set_seqs leaves InitPairs order)
	c->order_rects.clear(); c->order_base.clear(); c->ext2pos.clear();
	u64 k = 0;
	if (nrects == 0) { // back to MPCFlat::InitPairs order (mpcflat.cpp:145-155)
		for (u32 i = 0; i < n; ++i)
			for (u32 j = i + 1; j < n; ++j) { c->h_pair_x[k] = i; c->h_pair_y[k] = j; ++k; }
	} else {
		std::vector<u8> seen(((u64)n * n + 7) / 8, 0); // every pair exactly once
		std::vector<u64> base(nrects);
		for (u32 r = 0; r < nrects; ++r) {
			const u32 xa = rects[4 * r], xb = rects[4 * r + 1], ya = rects[4 * r + 2], yb = rects[4 * r + 3];
			const bool tri = xa == ya && xb == yb;
			if (xa > xb || ya > yb || xb > n || yb > n || (!tri && ya < xb))
				return fail(c, "mpcgpu_set_pair_order: rectangle %u = [%u,%u) x [%u,%u) is neither off the diagonal nor a triangle", r, xa, xb, ya, yb);
			base[r] = k;
			for (u32 x = xa; x < xb; ++x)
				for (u32 y = tri ? x + 1 : ya; y < yb; ++y) {
					const u64 bit = (u64)x * n + y;
					if (k >= c->npairs || (seen[bit >> 3] >> (bit & 7)) & 1) return fail(c, "mpcgpu_set_pair_order: pair (%u,%u) is listed twice", x, y);
					seen[bit >> 3] |= (u8)(1u << (bit & 7));
					c->h_pair_x[k] = x; c->h_pair_y[k] = y; ++k;
				}
		}
		if (k != c->npairs) return fail(c, "mpcgpu_set_pair_order: the rectangles hold %llu of %llu pairs", (u64)k, (u64)c->npairs);
		c->order_rects.assign(rects, rects + 4 * (size_t)nrects);
		c->order_base = base;
		std::vector<u32> dev(6 * (size_t)nrects);
		for (u32 r = 0; r < nrects; ++r) {
			for (u32 q = 0; q < 4; ++q) dev[6 * r + q] = rects[4 * r + q];
			dev[6 * r + 4] = (u32)(base[r] & 0xffffffffull); dev[6 * r + 5] = (u32)(base[r] >> 32);
		}
		if (upload(c, c->d_rects, dev)) return 1;
		HIPCHK(c, hipStreamSynchronize(c->stream)); // `dev` dies with this block
		c->ext2pos.resize(c->npairs);
		for (u64 q = 0; q < c->npairs; ++q) {
			const u32 x = c->h_pair_x[q], y = c->h_pair_y[q];
			c->ext2pos[(u64)x * n - ((u64)x * (x + 1)) / 2 + (y - x - 1)] = (u32)q;
		}
	}
	if (upload(c, c->d_pair_x, c->h_pair_x) || upload(c, c->d_pair_y, c->h_pair_y)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_pair_position(mpcgpu_ctx *c, uint32_t x, uint32_t y, uint64_t *pos)
{
	if (!c) return 1;
	if (x >= y || y >= c->n) return fail(c, "mpcgpu_pair_position: need x < y < n");
	if (pos) *pos = pair_pos(c, x, y);
	return 0;
}

int mpcgpu_set_mega(mpcgpu_ctx *c, uint32_t nfeat, const uint32_t *alpha, const float *weight,
	const float *const *logprobs, const float *const *logprob_mx, const uint8_t *const *profiles)
{
	if (!c) return 1;
	if (c->n == 0) return fail(c, "mpcgpu_set_mega: call mpcgpu_set_seqs first");
	HIPCHK(c, hipSetDevice(c->device));
	c->have_shard = c->have_store = false;
	c->have_mega = false;
	if (nfeat == 0) return 0; // back to byte-sequence emissions
	if (nfeat > MPC_MEGA_FMAX) return fail(c, "mpcgpu_set_mega: %u features; this build supports at most %d", nfeat, MPC_MEGA_FMAX);
	if (!alpha || !weight || !logprobs || !logprob_mx || !profiles) return fail(c, "mpcgpu_set_mega: NULL argument");
	// one upload: [alpha | weight | lp_off | mx_off | lp | mx | letters]
	std::vector<u32> lp_off(nfeat), mx_off(nfeat + 1);
	u32 nlp = 0, nmx = 0;
	for (u32 f = 0; f < nfeat; ++f) {
		if (alpha[f] == 0 || alpha[f] > 256) return fail(c, "mpcgpu_set_mega: feature %u has alphabet size %u", f, alpha[f]);
		lp_off[f] = nlp; mx_off[f] = nmx;
		nlp += alpha[f]; nmx += alpha[f] * alpha[f];
	}
	mx_off[nfeat] = nmx;
	const size_t fb_lds = MPC_FB_COEF_BYTES + ((size_t)nmx + 1) * sizeof(float);
	if (fb_lds > 64 * 1024) return fail(c, "mpcgpu_set_mega: the feature tables (%u floats) do not fit the kernel's LDS budget", nmx);
	u64 npos = 0;
	for (u32 i = 0; i < c->n; ++i) npos += c->len[i];
	std::vector<u8> blob;
	auto put = [&](const void *src, size_t bytes) {
		const size_t at = blob.size();
		blob.resize(at + ((bytes + 15) & ~(size_t)15));
		memcpy(blob.data() + at, src, bytes);
		return at;
	};
	const size_t o_alpha = put(alpha, 4ull * nfeat), o_weight = put(weight, 4ull * nfeat);
	const size_t o_lpoff = put(lp_off.data(), 4ull * nfeat), o_mxoff = put(mx_off.data(), 4ull * (nfeat + 1));
	std::vector<float> lp(nlp), mx(nmx);
	for (u32 f = 0; f < nfeat; ++f) {
		if (!logprobs[f] || !logprob_mx[f]) return fail(c, "mpcgpu_set_mega: NULL table of feature %u", f);
		memcpy(lp.data() + lp_off[f], logprobs[f], 4ull * alpha[f]);
		memcpy(mx.data() + mx_off[f], logprob_mx[f], 4ull * alpha[f] * alpha[f]);
	}
	const size_t o_lp = put(lp.data(), 4ull * nlp), o_mx = put(mx.data(), 4ull * nmx);
	std::vector<u8> letters((size_t)npos * nfeat);
	size_t at = 0;
	for (u32 i = 0; i < c->n; ++i) {
		if (!profiles[i]) return fail(c, "mpcgpu_set_mega: NULL profile of sequence %u", i);
		const size_t bytes = (size_t)c->len[i] * nfeat;
		for (size_t q = 0; q < bytes; ++q)
			if (profiles[i][q] >= alpha[q % nfeat])
				return fail(c, "mpcgpu_set_mega: sequence %u position %zu: letter %u of feature %zu is outside its alphabet (%u)",
					i, q / nfeat, (unsigned)profiles[i][q], q % nfeat, alpha[q % nfeat]);
		memcpy(letters.data() + at, profiles[i], bytes);
		at += bytes;
	}
	const size_t o_let = put(letters.data(), letters.size());
	if (upload(c, c->d_mg_in, blob)) return 1;
	HIPCHK(c, c->d_mg_prof.ensure(npos * 8));
	HIPCHK(c, c->d_mg_ins.ensure(npos * 4));
	HIPCHK(c, c->d_mg_tab.ensure(((u64)nmx + 1) * 4));
	MegaPrepParams mp;
	const u8 *base = c->d_mg_in.as<u8>();
	mp.nfeat = nfeat;
	mp.alpha = (const u32 *)(base + o_alpha); mp.weight = (const float *)(base + o_weight);
	mp.lp = (const float *)(base + o_lp); mp.lp_off = (const u32 *)(base + o_lpoff);
	mp.mx = (const float *)(base + o_mx); mp.mx_off = (const u32 *)(base + o_mxoff);
	mp.letters = base + o_let; mp.npos = npos;
	mp.prof = c->d_mg_prof.as<u64>(); mp.ins = c->d_mg_ins.as<float>(); mp.tab = c->d_mg_tab.as<float>();
	const u32 grid = (u32)std::min<u64>((std::max<u64>(npos, nmx + 1) + 255) / 256, 4096);
	MPC_LAUNCH(mega_prepare_kernel, grid, 256, 0, c->stream, mp);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->mg_nfeat = nfeat;
	c->mg_tab_floats = nmx + 1;
	for (u32 f = 0; f < MPC_MEGA_FMAX; ++f) {
		c->mg_base[f] = f < nfeat ? mx_off[f] : nmx; // unused features read the trailing 0.0f
		c->mg_alpha[f] = f < nfeat ? alpha[f] : 0;
	}
	c->have_mega = true;
	return 0;
}

uint64_t mpcgpu_pair_count(const mpcgpu_ctx *c) { return c ? c->npairs : 0; }

#include "mpcgpu_stage_a.inc"
#include "mpcgpu_exchange.inc"
#include "mpcgpu_joins.inc"
int mpcgpu_relax_info(mpcgpu_ctx *c, char *buf, uint32_t buflen, int *is_fallback)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_relax_info: no store");
	if (buf && buflen) {
		std::string d = c->store_desc;
		if (!c->tiles_desc.empty()) d += "; " + c->tiles_desc;
		if (!c->relax_kernel_name.empty()) d += "; kernel=" + c->relax_kernel_name; // the instantiation the last relax launched (as rocprofv3 names it)
		snprintf(buf, buflen, "%s", d.c_str());
	}
	if (is_fallback) *is_fallback = c->relax_fallback ? 1 : 0;
	return 0;
}

int mpcgpu_store_info(mpcgpu_ctx *c, uint64_t out[6])
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_store_info: no store");
	u64 packed_words = 0;
	for (u64 k = 0; k < c->npairs; ++k) packed_words += rec_words(c->len[c->h_pair_x[k]], c->len[c->h_pair_y[k]], c->all_nnz[k]);
	out[0] = c->have_pad ? c->var_total_blocks * 16 : 0;                 // block records (both orientations of every pair this store holds)
	out[1] = (c->have_pad && c->win_ok) ? c->win_total_blocks * 16 : 0;  // window records
	out[2] = packed_words * 4;                                           // packed records of all pairs
	out[3] = c->total_entries;                                           // stored posteriors of all pairs
	out[4] = c->h_vbase[c->own_k1] - c->h_vbase[c->own_k0];              // ... of the pairs this context relaxes
	u32 held = 0;
	for (u32 i = 0; i < c->n; ++i) held += (!c->partial || c->need[i]) ? 1u : 0u;
	out[5] = held;                                                       // sequences whose records the store holds
	return 0;
}

int mpcgpu_stage_a_info(mpcgpu_ctx *c, uint64_t *pairs, uint64_t *chained_pairs, uint64_t *chains)
{
	if (!c) return 1;
	if (pairs) *pairs = c->sa_pairs;
	if (chained_pairs) *chained_pairs = c->sa_chained;
	if (chains) *chains = c->sa_chains;
	return 0;
}

int mpcgpu_timers_reset(mpcgpu_ctx *c)
{
	if (!c) return 1;
	if (spans_collect(c)) return 1;
	for (int i = 0; i < MPCGPU_NKERNELS; ++i) { c->ms[i] = 0; c->launches[i] = 0; }
	return 0;
}

int mpcgpu_timers_enable(mpcgpu_ctx *c, int on)
{
	if (!c) return 1;
	c->timing = on != 0;
	return 0;
}

int mpcgpu_timers_get(mpcgpu_ctx *c, float ms[MPCGPU_NKERNELS], uint64_t launches[MPCGPU_NKERNELS])
{
	if (!c) return 1;
	if (spans_collect(c)) return 1;
	for (int i = 0; i < MPCGPU_NKERNELS; ++i) { ms[i] = c->ms[i]; launches[i] = c->launches[i]; }
	return 0;
}

int mpcgpu_work_get(mpcgpu_ctx *c, uint64_t *dp_cells, uint64_t *relax_entry_z, uint64_t *store_entries)
{
	if (!c) return 1;
	if (dp_cells) *dp_cells = c->work_cells;
	if (relax_entry_z) *relax_entry_z = c->work_entry_z;
	if (store_entries) *store_entries = c->total_entries;
	return 0;
}

} // extern "C"
