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
	DevBuf d_tile_next, d_bp_in, d_aln_res, d_post_prof, d_bp_seq, d_bp_map, d_bp_off, d_bp_coff, d_bp_keys, d_bp_vals, d_bp_tmp, d_bp_runs;
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
	u64 sa_pairs = 0, sa_chained = 0, sa_chains = 0; // last stage A: pairs, pairs that ran in chains, chains
	double aa_trace_t[5] = {0, 0, 0, 0, 0}; // MPCGPU_TRACE_HOST: host seconds of mpcgpu_align_alns' phases
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
bool trace_on()
{
	static int on = -1;
	if (on < 0) { const char *s = getenv("MPCGPU_TRACE"); on = (s && *s && *s != '0') ? 1 : 0; }
	return on == 1;
}

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

// ---- variable-size records + relax_var_kernel (kernels_relaxv.h) -------------------------------------------------------
// LDS of one workgroup: the pair table + nbuf staging buffers. 1024-thread workgroups own the CU's 160 KB, 512-thread ones
// run two per CU.
void var_lds_geometry(u32 geo, u32 nbuf, u32 *buf_bytes, size_t *smem)
{
	// MPCGPU_RELAX_LDS_KB: tests shrink the budget to reach tile splitting with short sequences (_1024: the one-workgroup geometry alone)
	const u64 lds_cap = (u64)(geo == 1024 ? env_int("MPCGPU_RELAX_LDS_KB_1024", env_int("MPCGPU_RELAX_LDS_KB", 160)) : env_int("MPCGPU_RELAX_LDS_KB", 80)) * 1024;
	*buf_bytes = (u32)(((lds_cap - MPC_RV_TAB_BYTES) / nbuf) & ~15ull);
	*smem = MPC_RV_TAB_BYTES + (size_t)nbuf * *buf_bytes;
}

// workgroup sizes of relax_var_kernel: 1024 (one per CU), or two per CU of 512 / 640 / 768 threads (4 / 5 / 6 waves per SIMD:
// 128 / 96 / 80 VGPRs); slots = cells per lane a tile may need (about 12.7 k wave-aligned cells per 4x4 tile at L~400)
// geometry id = threads per workgroup, except 2048 = two 1024-thread workgroups per CU (8 waves per SIMD, 64 VGPRs)
// cells per lane of the default geometry (two 1024-thread workgroups per CU, 64 VGPRs): 13. With 14 the compiler keeps five row-offset
// registers in spill slots and reloads them inside the walk (each reload waits for vmcnt(0)); 13 has none, and although 2 600 more
// 4x4 tiles then split into 4x2 the two iterations at 1000 x L~400 take 1171 ms against 1195 (12: 1282, most tiles split) — profiles/r05g
u32 var_slots_2048() { const int v = env_int("MPCGPU_RELAX_DIAG", 0) ? 13 : env_int("MPCGPU_RELAX_SLOTS_2048", 13); return v == 12 ? 12u : v == 14 ? 14u : 13u; } // (the measurement-only DIAG kernels exist with 13)
u32 var_max_slots(u32 geo) { return geo == 1024 ? 16u : geo == 2048 ? var_slots_2048() : geo == 768 ? 18u : 26u; }
u32 var_geo_from_env()
{
	// default: two 1024-thread workgroups per CU (8 waves per SIMD, 64 VGPRs, 13 cells per lane: var_slots_2048). Round 2 (profiles/r02e, r02h):
	// 768 x 2 1748 ms per two iterations at 1000 x L~400 against 1984 (512 x 2), 2017 (1024 x 2: spills in the walk), 2060 (1024 x 1,
	// two staging buffers). With round 3's walk (no spills at 64 VGPRs) 1024 x 2 is level or ahead: 1192 against 1203 ms (768 x 2)
	// on the synthetic family, 12.45 against 12.94 s on real data (profiles/r04a, r04e)
	const int t = env_int("MPCGPU_RELAX_WG", 2048);
	return t == 512 ? 512u : t == 1024 ? 1024u : t == 768 ? 768u : 2048u;
}

template <int TH, int SL, int WGS, int DG = 0, class BL = MpcRvBlocksAsm> void launch_relax_var(const RelaxVarParams &rp, u32 grid, size_t smem, hipStream_t st)
{
	auto kern = relax_var_kernel<TH, SL, WGS, DG, BL>;
	MPC_LAUNCH(kern, grid, TH, smem, st, rp);
}

// launches relax_var_kernel of geometry `geo` (see var_max_slots) over a tile list
static int relax_var_launch(mpcgpu_ctx *c, const StoreParams &sp, u64 k0, u64 k1, u32 geo, u32 nbuf, const DevBuf &d_tiles, u32 ntiles,
	u32 counter_slot, bool primary)
{
	const u32 threads = geo == 2048 ? 1024u : geo;
	u32 buf_bytes = 0;
	size_t smem = 0;
	var_lds_geometry(geo, nbuf, &buf_bytes, &smem);
	RelaxVarParams rp;
	rp.s = sp; rp.tiles = d_tiles.as<u32>(); rp.ntiles = ntiles;
	rp.k0 = k0; rp.k1 = k1; rp.nbuf = nbuf; rp.buf_bytes = buf_bytes;
	rp.tile_next = c->d_tile_next.as<u32>() + 8 * counter_slot;
	// MPCGPU_RELAX_DIAG=1|2|3 (staging only / merges only / merges + barriers): measurement kernels whose results are WRONG by design;
	// they exist only in a library built with -DMPC_RELAX_DIAG_BUILD (make diag), which also says so on stderr at every launch
	int diag = primary ? env_int("MPCGPU_RELAX_DIAG", 0) : 0;
#ifndef MPC_RELAX_DIAG_BUILD
	if (diag) return fail(c, "MPCGPU_RELAX_DIAG needs a library built with -DMPC_RELAX_DIAG_BUILD (measurement kernels: wrong results by design)");
#else
	if (diag) { fprintf(stderr, "[mpcgpu] WARNING: MPCGPU_RELAX_DIAG=%d: measurement kernel, the relax results are WRONG by design\n", diag); c->relax_fallback = true; }
#endif
	const char *merge_env = getenv("MPCGPU_RELAX_MERGE"); // "cxx": the compiler's code for the merge instead of the hand-scheduled one (A/B, 768 geometry)
	const bool merge_cxx = merge_env && !strcmp(merge_env, "cxx");
	const void *fn = geo == 1024 ? (const void *)relax_var_kernel<1024, 16, 1>
#ifdef MPC_RELAX_DIAG_BUILD
	               : geo == 2048 && diag ? (diag == 1 ? (const void *)relax_var_kernel<1024, 13, 2, 1> : diag == 2 ? (const void *)relax_var_kernel<1024, 13, 2, 2> : (const void *)relax_var_kernel<1024, 13, 2, 3>)
	               : geo == 768 && diag ? (diag == 1 ? (const void *)relax_var_kernel<768, 18, 2, 1> : diag == 2 ? (const void *)relax_var_kernel<768, 18, 2, 2> : (const void *)relax_var_kernel<768, 18, 2, 3>)
#endif
	               : geo == 2048 ? (var_slots_2048() == 14 ? (const void *)relax_var_kernel<1024, 14, 2> : var_slots_2048() == 12 ? (const void *)relax_var_kernel<1024, 12, 2> : (const void *)relax_var_kernel<1024, 13, 2>)
	               : geo == 768 ? (merge_cxx ? (const void *)relax_var_kernel<768, 18, 2, 0, MpcRvBlocksCxx> : (const void *)relax_var_kernel<768, 18, 2>)
	               : (const void *)relax_var_kernel<512, 26, 2>;
	HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
	if (primary) {
		char kn[128];
		snprintf(kn, sizeof(kn), "relax_var_kernel<%u, %u, %d, %d, %s>", threads, geo == 1024 ? 16u : geo == 2048 ? var_slots_2048() : geo == 768 ? 18u : 26u,
			geo == 1024 ? 1 : 2, (geo == 768 || geo == 2048) ? diag : 0, (geo == 768 && merge_cxx && !diag) ? "MpcRvBlocksCxx" : "MpcRvBlocksAsm");
		c->relax_kernel_name = kn;
	}
	int occ = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, (int)threads, smem) != hipSuccess || occ < 1) occ = 1;
	u32 grid = std::max(std::min<u32>(rp.ntiles, (u32)c->prop.multiProcessorCount * (u32)occ), 1u);
	if (trace_on()) {
		fprintf(stderr, "[mpcgpu] relax var: tiles=%u wg=%u (geometry %u) nbuf=%u buf=%u B lds=%zu B occ=%d grid=%u max_nnz=%u\n", rp.ntiles, threads, geo, nbuf, buf_bytes, smem, occ, grid, c->max_nnz);
		fflush(stderr);
	}
	TimedSpan ts;
	if (span_begin(c, 3, &ts)) return 1;
	if (geo == 1024) launch_relax_var<1024, 16, 1>(rp, grid, smem, c->stream);
#ifdef MPC_RELAX_DIAG_BUILD
	else if (geo == 2048 && diag == 1) launch_relax_var<1024, 13, 2, 1>(rp, grid, smem, c->stream);
	else if (geo == 2048 && diag == 2) launch_relax_var<1024, 13, 2, 2>(rp, grid, smem, c->stream);
	else if (geo == 2048 && diag == 3) launch_relax_var<1024, 13, 2, 3>(rp, grid, smem, c->stream);
	else if (geo == 768 && diag == 1) launch_relax_var<768, 18, 2, 1>(rp, grid, smem, c->stream);
	else if (geo == 768 && diag == 2) launch_relax_var<768, 18, 2, 2>(rp, grid, smem, c->stream);
	else if (geo == 768 && diag == 3) launch_relax_var<768, 18, 2, 3>(rp, grid, smem, c->stream);
#endif
	else if (geo == 2048) {
		if (var_slots_2048() == 14) launch_relax_var<1024, 14, 2>(rp, grid, smem, c->stream);
		else if (var_slots_2048() == 12) launch_relax_var<1024, 12, 2>(rp, grid, smem, c->stream);
		else launch_relax_var<1024, 13, 2>(rp, grid, smem, c->stream);
	}
	else if (geo == 768) {
		if (merge_cxx) launch_relax_var<768, 18, 2, 0, MpcRvBlocksCxx>(rp, grid, smem, c->stream);
		else launch_relax_var<768, 18, 2>(rp, grid, smem, c->stream);
	}
	else launch_relax_var<512, 26, 2>(rp, grid, smem, c->stream);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &ts)) return 1;
	return 0;
}

// ---- band tiles + relax_band_kernel (kernels_relaxb.h) ---------------------------------------------------------------------
#ifdef MPC_RELAX_DIAG_BUILD
#define MPC_RB_DIAG_CASES(TH, SL) \
	if (diag == 1) { fn = (const void *)relax_band_kernel<TH, SL, 2, 1>; if (go) MPC_LAUNCH((relax_band_kernel<TH, SL, 2, 1>), grid, TH, smem, c->stream, rp); } \
	else if (diag == 2) { fn = (const void *)relax_band_kernel<TH, SL, 2, 2>; if (go) MPC_LAUNCH((relax_band_kernel<TH, SL, 2, 2>), grid, TH, smem, c->stream, rp); } \
	else if (diag == 3) { fn = (const void *)relax_band_kernel<TH, SL, 2, 3>; if (go) MPC_LAUNCH((relax_band_kernel<TH, SL, 2, 3>), grid, TH, smem, c->stream, rp); } \
	else if (diag == 4) { fn = (const void *)relax_band_kernel<TH, SL, 2, 4>; if (go) MPC_LAUNCH((relax_band_kernel<TH, SL, 2, 4>), grid, TH, smem, c->stream, rp); } \
	else
#else
#define MPC_RB_DIAG_CASES(TH, SL)
#endif
constexpr u32 kBandSlotsWin = 15; // cells per lane of the direct-index merge (its two-word descriptor sets became one word each: registers for two more cells)
constexpr u32 kBandThreads = 1024, kBandSlots = 13; // two 1024-thread workgroups per CU (8 waves per SIMD, 64 VGPRs), 13 cells per lane (14: spill reloads inside the walk, and every reload waits for vmcnt(0) - the prefetch)

// 0 = launched (or nothing to do), 1 = error, 2 = not for band tiles (the caller runs relax_var)
int relax_band(mpcgpu_ctx *c, const StoreParams &sp, u64 k0, u64 k1)
{
	const u32 n = c->n, nb1 = c->band_nb1;
	// geometry: 2 x 1024 threads per CU (default) or 4 x 512 (MPCGPU_RELAX_WG=512): 8 waves per SIMD either way; a barrier of the
	// walk then holds 8 waves instead of 16, and three other workgroups fill a waiting one's issue slots
	const u32 bthreads = env_int("MPCGPU_RELAX_WG", 1024) == 512 ? 512u : 1024u;
	// (measured and removed again, profiles/r10a_rdrp_geometry_sweep.log: ONE 1024-thread workgroup per CU with the CU's 160 KB and 26 cells per
	// lane — 8x4 bands, 10.7 cells per lane, every step prefetched, 6.6 B per cell-step instead of 11.7 — is 18 % SLOWER on real data
	// (rdrp-500: 1818 against 1545 ms per two iterations): the walk's merges are chains of dependent LDS reads and want 8 waves per SIMD)
	const u32 lds_bytes = (u32)std::max(env_int("MPCGPU_RELAX_LDS_KB", bthreads == 512 ? 40 : 80), 3) * 1024u;
	const u32 cap = (lds_bytes - MPC_RB_TAB_BYTES) & ~15u, cap_blocks = cap / 16;
	const u32 cus = (u32)c->prop.multiProcessorCount;
	// the direct-index merge (window records for the Y operand) where the store has them; the 512-thread geometry and the
	// measurement kernels exist for the block walk only
	const bool use_win = c->win_ok && bthreads == 1024 && !env_int("MPCGPU_RELAX_DIAG", 0);
	const u32 kernel_slots = use_win ? kBandSlotsWin : kBandSlots;
	const u32 max_slots = (u32)std::min<int>(std::max(env_int("MPCGPU_RELAX_SLOTS", (int)kernel_slots), 1), (int)kernel_slots);
	if (c->btiles_k0 != k0 || c->btiles_k1 != k1) {
		c->btiles_k0 = c->btiles_k1 = ~0ull;
		const auto t_cut0 = std::chrono::steady_clock::now();
		auto lap_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_cut0).count(); };
		RbTileTabs tb;
		tb.cell_off = c->d_cell_off.as<u32>(); tb.yr = c->d_yr.as<u32>(); tb.ovf_sum = c->d_ovf_sum.as<u32>(); tb.ovf_maxc = c->d_ovf_maxc.as<u32>();
		tb.nb1 = nb1; tb.threads = bthreads; tb.k0 = k0; tb.k1 = k1;
		tb.win = use_win ? 1u : 0u;
		tb.ysum = use_win ? c->d_wsum.as<u32>() : tb.ovf_sum; tb.ymaxc = use_win ? c->d_wmaxc.as<u32>() : tb.ovf_maxc;
		// tile words of a list of tiles whose words 0..5 are set: Y ranges, first-piece blocks, slots; out: slots, mean blocks, bound, cells
		auto eval_tiles = [&](std::vector<u32> &words, std::vector<u32> &out) -> int {
			const u32 nt = (u32)(words.size() / MPC_RB_TILE_WORDS);
			out.assign((size_t)nt * 4, 0u);
			if (!nt) return 0;
			if (upload(c, c->d_btiles, words)) return 1;
			HIPCHK(c, c->d_bt_out.ensure((size_t)nt * 16));
			MPC_LAUNCH(band_eval_kernel, std::min<u32>(nt, cus * 32), 64, 0, c->stream, sp, tb, c->d_btiles.as<u32>(), nt, c->d_bt_out.as<u32>());
			HIPCHK(c, hipGetLastError());
			HIPCHK(c, hipMemcpyAsync(words.data(), c->d_btiles.p, words.size() * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipMemcpyAsync(out.data(), c->d_bt_out.p, out.size() * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			return 0;
		};
		auto pair_index = [&](u32 X, u32 Y) -> u64 { return (u64)X * n - ((u64)X * (X + 1)) / 2 + (Y - X - 1); }; // InitPairs order (mpcflat.cpp:145-155), X < Y
		// does the block of sequences [x0, x0+cx) x [y0, y0+cy) hold a pair whose position may lie in [k0, k1)? (a conservative test: the
		// kernels check every pair's own position)
		auto block_in_range = [&](u32 x0, u32 cx, u32 y0, u32 cy) -> bool {
			if (c->order_rects.empty()) {
				// InitPairs order: pair indices grow with X first — the block's pairs lie between its first row's first and its last row's last pair
				const u32 xl = std::min(x0 + cx - 1, y0 + cy - 2);
				return !(pair_index(x0, std::max(y0, x0 + 1)) >= k1 || pair_index(xl, y0 + cy - 1) < k0);
			}
			for (size_t r = 0; r < c->order_rects.size() / 4; ++r) { // block order: the rectangles whose positions meet [k0, k1) and whose sequences meet the block's
				const u32 *q = &c->order_rects[4 * r];
				const u64 cnt = q[2] >= q[1] ? (u64)(q[1] - q[0]) * (q[3] - q[2]) : (u64)(q[1] - q[0]) * (q[1] - q[0] - 1) / 2;
				if (c->order_base[r] >= k1 || c->order_base[r] + cnt <= k0) continue;
				if (x0 < q[1] && x0 + cx > q[0] && y0 < q[3] && y0 + cy > q[2]) return true;
			}
			return false;
		};
		u64 last_cut_candidates = 0; // super-tiles with cells in the last cut
		// super-tiles of nx x ny sequences cut into row bands of <= max_slots cells per lane and <= target blocks per step (mean)
		auto cut = [&](u32 nx, u32 ny, u32 target, std::vector<u32> &words, std::vector<u32> &out) -> int {
			std::vector<u32> cand;
			const u32 nbx = (n + nx - 1) / nx, nby = (n + ny - 1) / ny;
			for (u32 xb = 0; xb < nbx; ++xb)
				for (u32 yb = 0; yb < nby; ++yb) {
					const u32 x0 = xb * nx, cx = std::min(nx, n - x0), y0 = yb * ny, cy = std::min(ny, n - y0);
					if (y0 + cy <= x0 + 1) continue; // no pair X < Y in this block
					// a rank of a sharded run relaxes [k0, k1) only: blocks whose pairs all lie outside that range are not candidates
					if (!block_in_range(x0, cx, y0, cy)) continue;
					cand.insert(cand.end(), {x0, cx, y0, cy});
				}
			const u32 nc = (u32)(cand.size() / 4);
			words.clear();
			if (!nc) { out.clear(); return 0; }
			// The cut's small inputs and outputs (candidates in, bands per candidate out, list bases in) live in ONE page-locked record that
			// the kernels read and write in place: three transfers and a wait fewer per cut. (The FIRST wait of a cut ends 16 - 27 ms late in
			// some processes, every or every other step, whatever is queued first — copy or kernel, polled or blocking wait; not under
			// rocprofv3, not with torch initialised before the context: profiles/r10k_rank_time.log. Not understood, not fixed by this.)
			HIPCHK(c, c->h_bt.ensure(cand.size() * 4 + (size_t)nc * 8));
			u32 *cnt = c->h_bt.as<u32>(), *base = cnt + nc, *hcand = base + nc; // [cnt nc][base nc][cand 4 nc]
			memcpy(hcand, cand.data(), cand.size() * 4);
			const u32 grid = std::min<u32>(nc, cus * 32);
			MPC_LAUNCH(band_cut_kernel, grid, 64, (size_t)(nb1 + 1) * 8, c->stream, sp, tb, (const u32 *)hcand, nc, max_slots, target, 0, cnt,
				(const u32 *)nullptr, (u32 *)nullptr);
			HIPCHK(c, hipGetLastError());
			HIPCHK(c, hipStreamSynchronize(c->stream));
			u64 tot = 0;
			last_cut_candidates = 0;
			for (u32 q = 0; q < nc; ++q) { base[q] = (u32)tot; tot += cnt[q]; last_cut_candidates += cnt[q] ? 1 : 0; }
			if (tot > 0x7fffffffull / MPC_RB_TILE_WORDS) return fail(c, "mpcgpu_cons_iter: too many band tiles");
			words.assign((size_t)tot * MPC_RB_TILE_WORDS, 0u);
			if (!tot) { out.clear(); return 0; }

			HIPCHK(c, c->d_btiles.ensure(words.size() * 4));
			MPC_LAUNCH(band_cut_kernel, grid, 64, (size_t)(nb1 + 1) * 8, c->stream, sp, tb, (const u32 *)hcand, nc, max_slots, target, 1, (u32 *)nullptr,
				(const u32 *)base, c->d_btiles.as<u32>());
			HIPCHK(c, hipGetLastError());
			HIPCHK(c, hipMemcpyAsync(words.data(), c->d_btiles.p, words.size() * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream)); // `base` dies with this frame
			return eval_tiles(words, out);
		};
	struct Score { double fill, bytes_per_cell, in_target; u64 tiles; };
		auto score = [&](const std::vector<u32> &out, u32 target) {
			Score sc = {0, 0, 0, out.size() / 4};
			u64 cells = 0, est = 0, ok = 0;
			for (size_t t = 0; t + 3 < out.size(); t += 4) { cells += out[t + 3]; est += out[t + 1]; ok += out[t + 1] <= target ? 1 : 0; }
			if (sc.tiles) { sc.fill = (double)cells / ((double)sc.tiles * max_slots * bthreads); sc.in_target = (double)ok / (double)sc.tiles; }
			sc.bytes_per_cell = cells ? 16.0 * (double)est / (double)cells : 0.0;
			return sc;
		};
		// Shape of the super-tiles and the target the bands are cut to. MPCGPU_RELAX_SHAPE=nx,ny[,kb]: forced.
		// Narrow rows (1000 x L~400: 2 cells per row, y ranges that follow the diagonal): 8x8 super-tiles whose steps leave room for
		// the next step beside the current one fill the register slots — taken at once. Otherwise (real data: 7 cells per row, the
		// cells of 50 rows spread over 200 rows of the partner, and steps that vary by a factor of 1.6 around their mean) every
		// shape of the menu is cut in both modes and priced: a tile-step costs a fixed part (staging block, barrier; plus the exposed
		// transfer when the next step cannot be prefetched) and a part per cell slot. Shapes with more X than Y sequences are on the
		// menu because the X pieces are the band's rows only, the Y pieces the whole range those rows' cells reach.
		static const u32 menu[10][2] = {{8, 8}, {8, 4}, {8, 2}, {8, 1}, {4, 4}, {4, 2}, {4, 1}, {2, 2}, {2, 1}, {1, 1}};
		std::vector<u32> &words = c->v_words, &out = c->v_out; // (kept: see mpcgpu_ctx)
		words.clear(); out.clear();
		u32 use_nx = 0, use_ny = 0, use_target = 0;
		const u32 margin = std::min<u32>(cap_blocks / 16, 64); // blocks: steps vary around the mean
		const u32 half = cap_blocks / 2 > margin ? cap_blocks / 2 - margin : cap_blocks / 2, full = cap_blocks > 2 * margin ? cap_blocks - 2 * margin : cap_blocks;
		if (const char *sh = getenv("MPCGPU_RELAX_SHAPE")) {
			unsigned a = 0, b = 0, kb = 0;
			const int got = sscanf(sh, "%u,%u,%u", &a, &b, &kb);
			if (got >= 2 && a >= 1 && a <= MPC_RB_MAXN && b >= 1 && b <= MPC_RB_MAXN) {
				use_nx = a; use_ny = b; use_target = got == 3 && kb ? std::min<u32>(kb * 64, cap_blocks) : half;
				if (cut(use_nx, use_ny, use_target, words, out)) return 1;
			}
		}
		if (!use_nx && n <= 64) {
			// few sequences (the shrubs of -super7, the clusters of -super5): the 8 x 8 super-tiles with ALL their rows as one band
			// each, evaluated in one pass; taken when every one of them fits the cell slots and leaves room for the next step
			std::vector<u32> &w2 = c->v_w2, &o2 = c->v_o2; // (kept: see mpcgpu_ctx)
			w2.clear(); o2.clear();
			for (u32 x0 = 0; x0 < n; x0 += 8)
				for (u32 y0 = x0; y0 < n; y0 += 8) {
					u32 nw[MPC_RB_TILE_WORDS] = {x0, std::min(8u, n - x0), y0, std::min(8u, n - y0), 0u, (nb1 - 1) * (u32)MPC_RB_HB};
					if (y0 + nw[3] > x0 + 1) w2.insert(w2.end(), nw, nw + MPC_RB_TILE_WORDS);
				}
			if (eval_tiles(w2, o2)) return 1;
			bool ok = !w2.empty();
			for (size_t t = 0; t + 3 < o2.size(); t += 4) ok = ok && o2[t] <= max_slots && o2[t + 1] <= half;
			if (ok) { words.swap(w2); out.swap(o2); use_nx = 8; use_ny = 8; use_target = half; }
		}
		if (!use_nx) {
			if (cut(8, 8, half, words, out)) return 1;
			const Score sc = score(out, half);
			if (trace_on()) fprintf(stderr, "[mpcgpu] band tiles 8x8, two steps resident: %llu tiles, fill %.2f, %.2f B/cell-step, %.0f %% within target\n",
				(unsigned long long)sc.tiles, sc.fill, sc.bytes_per_cell, 100 * sc.in_target);
			if (!sc.tiles) { c->h_btiles.clear(); c->btiles_k0 = k0; c->btiles_k1 = k1; return 0; } // no cell in [k0,k1)
			if (sc.fill >= 0.70 && sc.in_target >= 0.90) { use_nx = 8; use_ny = 8; use_target = half; }
			// few cells (the shrubs of -super7: 32 sequences): nothing was cut — one band per super-tile — so no other shape or target
			// gives fewer tile-steps, and the search below (20 more cuts) is skipped
			else if (sc.in_target >= 1.0 && sc.tiles == last_cut_candidates) { use_nx = 8; use_ny = 8; use_target = half; }
		}
		if (!use_nx) {
			// worst step / mean step of this data set: the exact worst steps of a sample of 4x2 tiles cut to the whole area (95th percentile)
			double r95 = 1.0;
			{
				if (cut(4, 2, full, words, out)) return 1;
				const u32 nt = (u32)(words.size() / MPC_RB_TILE_WORDS);
				std::vector<u32> sample;
				const u32 stride = std::max(nt / 4096u, 1u);
				for (u32 t = 0; t < nt; t += stride) if (out[4 * t + 3]) sample.push_back(t);
				if (!sample.empty()) {
					if (upload(c, c->d_btiles, words) || upload(c, c->d_bt_list, sample)) return 1;
					HIPCHK(c, c->d_bt_count.ensure(sample.size() * 4));
					MPC_LAUNCH(band_fit_kernel, std::min<u32>((u32)sample.size(), cus * 32), 64, 0, c->stream, sp, c->d_ovf_off.as<u32>(), nb1,
						c->d_btiles.as<u32>(), c->d_bt_list.as<u32>(), (u32)sample.size(), c->d_bt_count.as<u32>(), tb.win);
					HIPCHK(c, hipGetLastError());
					std::vector<u32> worst(sample.size());
					HIPCHK(c, hipMemcpyAsync(worst.data(), c->d_bt_count.p, sample.size() * 4, hipMemcpyDeviceToHost, c->stream));
					HIPCHK(c, hipStreamSynchronize(c->stream));
					std::vector<double> ratio;
					for (size_t q = 0; q < sample.size(); ++q) if (out[4 * sample[q] + 1]) ratio.push_back((double)worst[q] / (double)out[4 * sample[q] + 1]);
					if (!ratio.empty()) { std::sort(ratio.begin(), ratio.end()); r95 = std::max(ratio[std::min(ratio.size() - 1, (size_t)(0.95 * (double)ratio.size()))], 1.0); }
				}
			}
			const u32 single = std::max(std::min((u32)((double)cap_blocks / r95 * 0.98), full), 1u); // mean step such that the worst one still fits
			// cost of a tile-step in units of one cell slot of this data (a slot's merges grow with the rows' entries)
			u64 rows = 0;
			for (u32 i = 0; i + 1 < n; ++i) rows += (u64)c->len[i] * (n - 1 - i);
			const double per_row = rows ? (double)c->total_entries / (double)rows : 2.0;
			const double slot_us = 0.3 + 0.2 * per_row, fixed = 1.0 / slot_us, exposed = 1.5 / slot_us;
			if (trace_on()) fprintf(stderr, "[mpcgpu] band tiles: %.1f cells per row; worst step / mean step = %.3f (95th percentile): one step resident = %u blocks mean\n", per_row, r95, single);
			double best = 0;
			bool have = false;
			std::vector<u32> &w2 = c->v_w2, &o2 = c->v_o2; // (kept: see mpcgpu_ctx)
			w2.clear(); o2.clear();
			for (u32 mode = 0; mode < 2; ++mode)
				for (u32 m = 0; m < 10; ++m) {
					const u32 target = mode == 0 ? half : single;
					if (cut(menu[m][0], menu[m][1], target, w2, o2)) return 1;
					u64 slots = 0, over = 0;
					const u64 nt = o2.size() / 4;
					for (size_t t = 0; t + 3 < o2.size(); t += 4) { slots += std::min(o2[t], max_slots); over += o2[t + 1] > target ? 1 : 0; }
					if (!nt) continue;
					// tiles over the target (single index bands that do not fit) will be split by sequences: charged double
					const double cost = ((double)nt + (double)over) * (fixed + (mode ? exposed : 0.0)) + (double)slots;
					if (trace_on()) fprintf(stderr, "[mpcgpu] band tiles %ux%u, %s: %llu tiles (%llu over the target), %.1f cells per lane, cost %.3g\n",
						menu[m][0], menu[m][1], mode ? "one step resident" : "two steps resident", (unsigned long long)nt, (unsigned long long)over,
						(double)slots / (double)nt, cost);
					if (!have || cost < best) { have = true; best = cost; words.swap(w2); out.swap(o2); use_nx = menu[m][0]; use_ny = menu[m][1]; use_target = target; }
				}
			if (!have) { c->h_btiles.clear(); c->btiles_k0 = k0; c->btiles_k1 = k1; return 0; }
		}
		// every tile must fit: cells per lane, 16-bit first-piece offsets, and its WORST step in the staging area (upper bound
		// first; the exact maximum over Z only where the bound does not settle it). What does not fit is halved: band, then Y, then X.
		std::vector<u32> &okw = c->v_okw;
		okw.clear();
		u64 nsplit = 0;
		if (!words.empty() && upload(c, c->d_btiles, words)) return 1; // (the device copy is that of the last shape tried)
		for (int round = 0; round < 24 && !words.empty(); ++round) {
			const u32 nt = (u32)(words.size() / MPC_RB_TILE_WORDS);
			std::vector<u32> need, exact;
			for (u32 t = 0; t < nt; ++t)
				if (out[4 * t] <= max_slots && words[(size_t)t * MPC_RB_TILE_WORDS + 6] <= MPC_RB_MAXFIRST && out[4 * t + 2] > cap_blocks) need.push_back(t);
			if (!need.empty()) {
				if (trace_on()) {
					u64 bsum = 0, esum = 0;
					for (u32 t : need) { bsum += out[4 * t + 2]; esum += out[4 * t + 1]; }
					fprintf(stderr, "[mpcgpu] band tiles: %zu of %u tiles need the exact worst step (mean bound %.0f blocks, mean step %.0f, area %u)\n",
						need.size(), nt, (double)bsum / need.size(), (double)esum / need.size(), cap_blocks);
				}
				if (upload(c, c->d_bt_list, need)) return 1;
				HIPCHK(c, c->d_bt_count.ensure(need.size() * 4));
				MPC_LAUNCH(band_fit_kernel, std::min<u32>((u32)need.size(), cus * 32), 64, 0, c->stream, sp, c->d_ovf_off.as<u32>(), nb1,
					c->d_btiles.as<u32>(), c->d_bt_list.as<u32>(), (u32)need.size(), c->d_bt_count.as<u32>(), tb.win);
				HIPCHK(c, hipGetLastError());
				exact.resize(need.size());
				HIPCHK(c, hipMemcpyAsync(exact.data(), c->d_bt_count.p, need.size() * 4, hipMemcpyDeviceToHost, c->stream));
				HIPCHK(c, hipStreamSynchronize(c->stream));
				for (size_t q = 0; q < need.size(); ++q) out[4 * need[q] + 2] = exact[q]; // the bound becomes the exact worst step
			}
			std::vector<u32> next;
			for (u32 t = 0; t < nt; ++t) {
				const u32 *w = &words[(size_t)t * MPC_RB_TILE_WORDS];
				if (out[4 * t + 3] == 0) continue; // no cell
				if (out[4 * t] <= max_slots && w[6] <= MPC_RB_MAXFIRST && out[4 * t + 2] <= cap_blocks) { okw.insert(okw.end(), w, w + MPC_RB_TILE_WORDS); continue; }
				++nsplit;
				auto push = [&](u32 x0, u32 nx, u32 y0, u32 ny, u32 r0, u32 r1) {
					u32 nw[MPC_RB_TILE_WORDS] = {x0, nx, y0, ny, r0, r1};
					next.insert(next.end(), nw, nw + MPC_RB_TILE_WORDS);
				};
				const u32 x0 = w[0], nx = w[1], y0 = w[2], ny = w[3], r0 = w[4], r1 = w[5];
				const u32 hb = (r1 - r0) / MPC_RB_HB;
				if (hb > 1) { const u32 mid = r0 + (hb / 2) * MPC_RB_HB; push(x0, nx, y0, ny, r0, mid); push(x0, nx, y0, ny, mid, r1); }
				else if (ny > 1) { push(x0, nx, y0, ny / 2, r0, r1); push(x0, nx, y0 + ny / 2, ny - ny / 2, r0, r1); }
				else if (nx > 1) { push(x0, nx / 2, y0, ny, r0, r1); push(x0 + nx / 2, nx - nx / 2, y0, ny, r0, r1); }
				else {
					if (trace_on()) fprintf(stderr, "[mpcgpu] band tiles: rows [%u,%u) of pair (%u,%u) do not fit (slots %u, first %u, worst step %u blocks of %u)\n",
						r0, r1, x0, y0, out[4 * t], w[6], out[4 * t + 2], cap_blocks);
					return 2;
				}
			}
			words.swap(next);
			if (eval_tiles(words, out)) return 1;
		}
		if (!words.empty()) return 2;
		u64 ntail_split = 0;
		{
			// The tail of the launch: the kernel deals the list to the 8 XCDs in contiguous chunks (a counter each), and a chunk's
			// LAST tiles — one per resident workgroup of the XCD — are the ones that finish alone. They are cut in two by rows (any
			// part of a tile is a valid tile): half the tail, which is one tile of ~62 per workgroup on one GPU and one of ~8 on a rank
			// of eight. MPCGPU_RELAX_TAIL=0: off.
			const u32 W = MPC_RB_TILE_WORDS;
			const size_t nt = okw.size() / W;
			const u32 per_xcd = std::max(cus * 2u / 8u, 1u); // resident workgroups of an XCD (two per CU)
			if (env_int("MPCGPU_RELAX_TAIL", 1) != 0 && nt >= (size_t)per_xcd * 8u * 3u) {
				const size_t chunk = (nt + 7) / 8;
				std::vector<u32> split;
				split.reserve(okw.size() + (size_t)per_xcd * 8u * W);
				for (size_t c0 = 0; c0 < nt; c0 += chunk) {
					const size_t c1 = std::min(c0 + chunk, nt), body = c1 - c0 > per_xcd ? c1 - per_xcd : c0;
					split.insert(split.end(), okw.begin() + c0 * W, okw.begin() + body * W);
					for (size_t t = body; t < c1; ++t) {
						const u32 *w = &okw[t * W];
						const u32 hb = (w[5] - w[4] + MPC_RB_HB - 1) / MPC_RB_HB;
						if (hb < 2) { split.insert(split.end(), w, w + W); continue; }
						const u32 mid = w[4] + (hb / 2) * MPC_RB_HB;
						u32 a[MPC_RB_TILE_WORDS] = {w[0], w[1], w[2], w[3], w[4], mid}, b[MPC_RB_TILE_WORDS] = {w[0], w[1], w[2], w[3], mid, w[5]};
						split.insert(split.end(), a, a + W);
						split.insert(split.end(), b, b + W);
						++ntail_split;
					}
				}
				okw.swap(split);
			}
		}
		{
			u64 cells = 0, est = 0;
			std::vector<u32> &o2 = c->v_o2, &w2 = c->v_w2; // (kept: see mpcgpu_ctx)
			w2 = okw;
			if (eval_tiles(w2, o2)) return 1; // (fills words 6.. of the halves; also leaves the final list's statistics for the description)
			if (ntail_split) { // halves without a cell go; the others are tiles like any other
				okw.clear();
				for (size_t t = 0; t < w2.size() / MPC_RB_TILE_WORDS; ++t)
					if (o2[4 * t + 3]) okw.insert(okw.end(), w2.begin() + t * MPC_RB_TILE_WORDS, w2.begin() + (t + 1) * MPC_RB_TILE_WORDS);
			}
			for (size_t t = 0; t + 3 < o2.size(); t += 4) { cells += o2[t + 3]; est += o2[t + 1]; }
			const size_t nt = okw.size() / MPC_RB_TILE_WORDS;
			char b[320];
			snprintf(b, sizeof(b), "%zu band tiles of <= %ux%u pairs (%llu split), target %u B per step of %u B staging (%s), mean step %.0f B, %.1f of %u cells per lane, %.2f B per cell-step",
				nt, use_nx, use_ny, (unsigned long long)nsplit, use_target * 16, cap, use_target <= cap_blocks / 2 ? "two steps resident" : "one step resident",
				nt ? 16.0 * (double)est / (double)nt : 0.0, nt ? (double)cells / ((double)nt * bthreads) : 0.0, max_slots, cells ? 16.0 * (double)est / (double)cells : 0.0);
			c->tiles_desc = b;
		}
		c->h_btiles.swap(okw);
		if (trace_on() && env_int("MPCGPU_TRACE_TILES", 0))
			for (size_t t = 0; t < c->h_btiles.size() / MPC_RB_TILE_WORDS && t < (size_t)env_int("MPCGPU_TRACE_TILES", 0); ++t) {
				const u32 *w = &c->h_btiles[t * MPC_RB_TILE_WORDS];
				fprintf(stderr, "[mpcgpu] tile %zu: X %u+%u Y %u+%u rows [%u,%u) first %u slots %u; Y rows", t, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7] & 0xffu);
				for (u32 j = 0; j < w[3]; ++j) fprintf(stderr, " [%u,%u)", w[8 + j] & 0xffffu, w[8 + j] >> 16);
				fprintf(stderr, "\n");
			}
		if (upload(c, c->d_btiles, c->h_btiles)) return 1;
		HIPCHK(c, hipStreamSynchronize(c->stream));
		if (trace_on()) fprintf(stderr, "[mpcgpu] band tiles: cut, checked and uploaded in %.2f ms\n", lap_ms());
		c->btiles_k0 = k0; c->btiles_k1 = k1;
	}
	const u32 ntiles = (u32)(c->h_btiles.size() / MPC_RB_TILE_WORDS);
	if (!ntiles) return 0;
	HIPCHK(c, c->d_tile_next.ensure(160 * 4));
	HIPCHK(c, hipMemsetAsync(c->d_tile_next.p, 0, 160 * 4, c->stream));
	RelaxBandParams rp;
	rp.s = sp; rp.ovf_off = c->d_ovf_off.as<u32>(); rp.nb1 = nb1; rp.cell_off = c->d_cell_off.as<u32>();
	rp.tiles = c->d_btiles.as<u32>(); rp.ntiles = ntiles; rp.k0 = k0; rp.k1 = k1; rp.cap_bytes = cap;
	rp.tile_next = c->d_tile_next.as<u32>();
	{ // cell order inside an X group (kernels_relaxb.h): blocks of G rows, a block's cells pair after pair. MPCGPU_RELAX_ORDER = G, or "pairs"
		// (no blocks: the layout until round 4's last profile). 1000 x 400, relax per step: pairs 880 ms, G = 1: 929, 2: 920, 4: 885,
		// 8: 849, 16: 851, 32: 886 (profiles/r09b_order_sweep.log). The two-list walk on wide rows (rdrp, <= 4x2 pairs, 3 cells per lane) gains nothing from
		// blocks: 12 577 ms against 12 280 pair after pair (profiles/r09c) — its default stays "pairs".
		const char *order_env = getenv("MPCGPU_RELAX_ORDER");
		const u32 order_default = use_win ? 8u : 0u;
		rp.by_rows = !order_env ? order_default : !strcmp(order_env, "pairs") ? 0u : (u32)atoi(order_env) > 0 ? (u32)atoi(order_env) : order_default;
	}
	const size_t smem = MPC_RB_TAB_BYTES + (size_t)cap;
	const int diag = env_int("MPCGPU_RELAX_DIAG", 0); // measurement only (results wrong): needs a library built with -DMPC_RELAX_DIAG_BUILD
#ifndef MPC_RELAX_DIAG_BUILD
	if (diag) return fail(c, "MPCGPU_RELAX_DIAG needs a library built with -DMPC_RELAX_DIAG_BUILD (measurement kernels: wrong results by design)");
#else
	if (diag) { fprintf(stderr, "[mpcgpu] WARNING: MPCGPU_RELAX_DIAG=%d: measurement kernel, the relax results are WRONG by design\n", diag); c->relax_fallback = true; }
#endif
	const char *merge_env = getenv("MPCGPU_RELAX_MERGE"); // "cxx": the compiler's code for the merge instead of the hand-scheduled one (A/B)
	const bool merge_cxx = merge_env && !strcmp(merge_env, "cxx");
	u32 grid = 1;
	const void *fn = nullptr;
	for (int go = 0; go < 2; ++go) { // pass 0: which instantiation (attributes, occupancy); pass 1: launch
		TimedSpan ts;
		if (go && span_begin(c, 3, &ts)) return 1;
		if (use_win) { // window records for the Y operand: the direct-index merge
			if (merge_cxx) { fn = (const void *)relax_band_kernel<kBandThreads, kBandSlotsWin, 2, 0, MpcRbWinCxx>; if (go) MPC_LAUNCH((relax_band_kernel<kBandThreads, kBandSlotsWin, 2, 0, MpcRbWinCxx>), grid, kBandThreads, smem, c->stream, rp); }
			else { fn = (const void *)relax_band_kernel<kBandThreads, kBandSlotsWin, 2, 0, MpcRbWinAsm>; if (go) MPC_LAUNCH((relax_band_kernel<kBandThreads, kBandSlotsWin, 2, 0, MpcRbWinAsm>), grid, kBandThreads, smem, c->stream, rp); }
		}
		else if (bthreads == 512) { fn = (const void *)relax_band_kernel<512, kBandSlots, 4>; if (go) MPC_LAUNCH((relax_band_kernel<512, kBandSlots, 4>), grid, 512, smem, c->stream, rp); }
		else
		MPC_RB_DIAG_CASES(kBandThreads, kBandSlots)
		if (merge_cxx) { fn = (const void *)relax_band_kernel<kBandThreads, kBandSlots, 2, 0, MpcRbBlocksCxx>; if (go) MPC_LAUNCH((relax_band_kernel<kBandThreads, kBandSlots, 2, 0, MpcRbBlocksCxx>), grid, kBandThreads, smem, c->stream, rp); }
		else { fn = (const void *)relax_band_kernel<kBandThreads, kBandSlots, 2>; if (go) MPC_LAUNCH((relax_band_kernel<kBandThreads, kBandSlots, 2>), grid, kBandThreads, smem, c->stream, rp); }
		if (!go) {
			// (the two runtime queries cost a good fraction of a millisecond: once per context, kernel and LDS size — a -super7 run
			// relaxes 400 small stores on every worker context)
			int occ = 0;
			if (c->band_fn == fn && c->band_smem == smem) occ = c->band_occ;
			else {
				HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
				if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, (int)bthreads, smem) != hipSuccess || occ < 1) occ = 1;
				c->band_fn = fn; c->band_smem = smem; c->band_occ = occ;
			}
			grid = std::max(std::min<u32>(ntiles, cus * (u32)occ), 1u);
			char kn[128];
			snprintf(kn, sizeof(kn), "relax_band_kernel<%u, %u, %u, %d, %s>", bthreads, kernel_slots, bthreads == 512 ? 4u : 2u, bthreads == 512 ? 0 : diag,
				use_win ? (merge_cxx ? "MpcRbWinCxx" : "MpcRbWinAsm") : merge_cxx && !diag && bthreads == 1024 ? "MpcRbBlocksCxx" : "MpcRbBlocksAsm");
			c->relax_kernel_name = kn;
			if (trace_on()) { fprintf(stderr, "[mpcgpu] relax band: %s; lds=%zu B occ=%d grid=%u\n", c->tiles_desc.c_str(), smem, occ, grid); fflush(stderr); }
		} else {
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &ts)) return 1;
		}
	}
#ifdef MPC_RELAX_DIAG_BUILD
	if (trace_on()) {
		u32 cnt[160];
		HIPCHK(c, hipMemcpyAsync(cnt, c->d_tile_next.p, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		fprintf(stderr, "[mpcgpu] relax band (measurement build): %u steps prefetched beside the current one, %u staged after the merges\n", cnt[9], cnt[8]);
		if (diag == 4) {
			const unsigned long long *t = (const unsigned long long *)(cnt + 16);
			fprintf(stderr, "[mpcgpu] relax band timers per wave number (share of the walk: barrier / staging block / DMA wait):");
			for (int w = 0; w < 16; ++w)
				fprintf(stderr, " %d: %.1f/%.1f/%.1f", w, 100.0 * t[4 * w + 1] / std::max<double>(t[4 * w], 1), 100.0 * t[4 * w + 2] / std::max<double>(t[4 * w], 1), 100.0 * t[4 * w + 3] / std::max<double>(t[4 * w], 1));
			fprintf(stderr, "\n");
		}
	}
#endif
	return 0;
}

int relax_var(mpcgpu_ctx *c, const StoreParams &sp, u64 k0, u64 k1)
{
	const u32 n = c->n;
	auto pidx = [&](u32 i, u32 j) { return pair_pos(c, i, j); };
	// Tiles of geometry (geo, nbuf) out of a list of candidate tiles: a tile is kept when its cells fit the register slots and its
	// records of one step, packed back to back, fit one staging buffer at EVERY step (the worst step of every tile is measured on
	// the device); others are split (Y first, then X) and measured again. Single pairs that still do not fit go to `leftover`
	// (when given: the second geometry takes them) or fail the call.
	auto build_tiles = [&](u32 geo, u32 nbuf, std::vector<u32> cand, std::vector<u32> &ok, std::vector<u32> *leftover) -> int {
		const u32 threads = geo == 2048 ? 1024u : geo;
		u32 buf_bytes = 0;
		size_t smem = 0;
		var_lds_geometry(geo, nbuf, &buf_bytes, &smem);
		const u32 max_slots = (u32)std::min<int>(std::max(env_int("MPCGPU_RELAX_SLOTS", (int)var_max_slots(geo)), 1), (int)var_max_slots(geo));
		// slots a tile needs: the cells of its pairs in [k0,k1), every pair rounded up to whole waves, in chunks of `threads`
		auto tile_slots = [&](u32 x0, u32 nx, u32 y0, u32 ny) {
			u64 cells = 0;
			for (u32 X = x0; X < x0 + nx; ++X)
				for (u32 Y = std::max(y0, X + 1); Y < y0 + ny; ++Y) {
					const u64 k = pidx(X, Y);
					if (k >= k0 && k < k1) cells += ((u64)c->all_nnz[k] + 63) & ~63ull;
				}
			return (u32)((cells + threads - 1) / threads);
		};
		std::vector<u32> tiles;
		bool too_big = false;
		std::function<void(u32, u32, u32, u32)> emit = [&](u32 x0, u32 nx, u32 y0, u32 ny) {
			const u32 slots = tile_slots(x0, nx, y0, ny);
			if (slots == 0) return;
			if (slots <= max_slots) { tiles.insert(tiles.end(), {x0, nx, y0, ny}); return; }
			if (ny > 1) { emit(x0, nx, y0, ny / 2); emit(x0, nx, y0 + ny / 2, ny - ny / 2); }
			else if (nx > 1) { emit(x0, nx / 2, y0, ny); emit(x0 + nx / 2, nx - nx / 2, y0, ny); }
			else if (leftover) leftover->insert(leftover->end(), {x0, nx, y0, ny});
			else too_big = true;
		};
		for (size_t t = 0; t + 3 < cand.size(); t += 4) emit(cand[t], cand[t + 1], cand[t + 2], cand[t + 3]);
		if (too_big) return fail(c, "mpcgpu_cons_iter: a pair has more than %u stored cells (tile slot budget)", max_slots * threads);
		const u32 budget_blocks = buf_bytes / 16;
		for (int round = 0; round < 8 && !tiles.empty(); ++round) {
			const u32 nt = (u32)(tiles.size() / 4);
			if (upload(c, c->d_tiles, tiles)) return 1;
			HIPCHK(c, c->d_tilefit.ensure((size_t)nt * 4));
			MPC_LAUNCH(var_tile_fit_kernel, std::min<u32>(nt, (u32)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp, c->d_tiles.as<u32>(), nt,
				c->d_tilefit.as<u32>());
			HIPCHK(c, hipGetLastError());
			std::vector<u32> fit(nt);
			HIPCHK(c, hipMemcpyAsync(fit.data(), c->d_tilefit.p, (size_t)nt * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			std::vector<u32> next;
			u32 nsplit = 0;
			for (u32 t = 0; t < nt; ++t) {
				const u32 x0 = tiles[4 * t], nx = tiles[4 * t + 1], y0 = tiles[4 * t + 2], ny = tiles[4 * t + 3];
				if (fit[t] <= budget_blocks) { ok.insert(ok.end(), {x0, nx, y0, ny}); continue; }
				++nsplit;
				auto push = [&](u32 a, u32 b, u32 cc, u32 d) { if (tile_slots(a, b, cc, d)) next.insert(next.end(), {a, b, cc, d}); };
				if (ny > 1) { push(x0, nx, y0, ny / 2); push(x0, nx, y0 + ny / 2, ny - ny / 2); }
				else if (nx > 1) { push(x0, nx / 2, y0, ny); push(x0 + nx / 2, nx - nx / 2, y0, ny); }
				else if (leftover) leftover->insert(leftover->end(), {x0, nx, y0, ny});
				else return fail(c, "mpcgpu_cons_iter: the two records of pair (%u,%u) need %u bytes of LDS at some step, one staging buffer holds %u",
					x0, y0, fit[t] * 16, buf_bytes);
			}
			if (trace_on() && nsplit) { fprintf(stderr, "[mpcgpu] relax var: %u of %u tiles over the LDS budget (%u B), split\n", nsplit, nt, buf_bytes); fflush(stderr); }
			tiles.swap(next);
		}
		if (!tiles.empty()) return fail(c, "mpcgpu_cons_iter: tile splitting did not converge");
		return 0;
	};
	auto describe = [](const std::vector<u32> &ok) {
		u32 hist[5][5] = {{0}};
		for (size_t t = 0; t + 3 < ok.size(); t += 4) hist[std::min(ok[t + 1], 4u)][std::min(ok[t + 3], 4u)]++;
		char b[256];
		int o = snprintf(b, sizeof(b), "%zu tiles:", ok.size() / 4);
		for (u32 a = 4; a >= 1; --a)
			for (u32 bb = 4; bb >= 1; --bb)
				if (hist[a][bb] && o < (int)sizeof(b) - 24) o += snprintf(b + o, sizeof(b) - o, " %ux%u x %u", a, bb, hist[a][bb]);
		return std::string(b);
	};
	if (c->tiles_k0 != k0 || c->tiles_k1 != k1 || c->tiles_bx != 4 || c->tiles_by != 4) {
		c->tiles_k0 = c->tiles_k1 = ~0ull;
		// X blocks of 4, Y blocks of 4, walked in 8x8 super-tiles (the workgroups of an XCD read the same sequences' records)
		std::vector<u32> cand;
		const u32 nbx = (n + 3) / 4, nby = (n + 3) / 4;
		for (u32 sx = 0; sx < nbx; sx += 8)
			for (u32 sy = 0; sy < nby; sy += 8)
				for (u32 xb = sx; xb < std::min(sx + 8, nbx); ++xb)
					for (u32 yb = sy; yb < std::min(sy + 8, nby); ++yb) {
						const u32 x0 = xb * 4, nx = std::min(4u, n - x0), y0 = yb * 4, ny = std::min(4u, n - y0);
						if (y0 + ny <= x0 + 1) continue; // no pair X < Y in this block
						cand.insert(cand.end(), {x0, nx, y0, ny});
					}
		std::vector<u32> ok, ok2, left;
		if (build_tiles(c->var_threads, c->var_nbuf, cand, ok, c->var_mixed ? &left : nullptr)) return 1;
		if (!left.empty() && build_tiles(1024, 1, left, ok2, nullptr)) return 1;
		c->tiles_desc = describe(ok);
		if (!ok2.empty()) c->tiles_desc += "; + 1 x 1024-thread workgroup per CU, 1 staging buffer of 160 KB for " + describe(ok2);
		c->h_tiles.swap(ok);
		c->h_tiles2.swap(ok2);
		if (upload(c, c->d_tiles, c->h_tiles)) return 1;
		if (!c->h_tiles2.empty() && upload(c, c->d_tiles2, c->h_tiles2)) return 1;
		HIPCHK(c, hipStreamSynchronize(c->stream)); // the source of the async copy lives in the context; drained before any rebuild
		c->tiles_k0 = k0; c->tiles_k1 = k1; c->tiles_bx = 4; c->tiles_by = 4;
	}
	if (c->h_tiles.empty() && c->h_tiles2.empty()) return 0;
	HIPCHK(c, c->d_tile_next.ensure(16 * 4));
	HIPCHK(c, hipMemsetAsync(c->d_tile_next.p, 0, 16 * 4, c->stream));
	if (!c->h_tiles.empty() && relax_var_launch(c, sp, k0, k1, c->var_threads, c->var_nbuf, c->d_tiles, (u32)(c->h_tiles.size() / 4), 0, true)) return 1;
	if (!c->h_tiles2.empty() && relax_var_launch(c, sp, k0, k1, 1024, 1, c->d_tiles2, (u32)(c->h_tiles2.size() / 4), 1, c->h_tiles.empty())) return 1;
	return 0;
}

// Builds the variable-size record store (rec_off table, records, entry positions). 0 = built (c->pad_var set), 1 = error,
// 2 = this run does not fit the layout's limits (the caller falls back to the fixed-size records / the gather kernel).
int build_var_store(mpcgpu_ctx *c)
{
	const u32 n = c->n;
	if (c->max_len > MPC_RV_MAXLEN) return 2;
	// geometry: the configured one (default two 768-thread workgroups per CU, 80 KB of LDS each); when the largest pair does not
	// fit it — long or poorly aligned sequences: wide posterior rows, records of tens of KB — one 1024-thread workgroup per CU
	// with the whole 160 KB as ONE staging buffer, before giving the run to the fallback layouts
	u32 threads = var_geo_from_env(); // geometry id (see var_max_slots)
	u32 nbuf = (u32)std::min(std::max(env_int("MPCGPU_RELAX_NBUF", threads == 1024 ? 2 : 1), 1), 2);
	auto slots_ok = [&](u32 geo) { return (((u64)c->max_nnz + 63) & ~63ull) <= (u64)var_max_slots(geo) * (geo == 2048 ? 1024u : geo); };
	StoreParams sp0;
	fill_store_params(c, sp0);
	const u64 nn = (u64)n * n;
	HIPCHK(c, c->d_sizes.ensure(nn * 4));
	MPC_LAUNCH(var_size_kernel, (u32)std::min<u64>(nn, (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp0, c->d_sizes.as<u32>());
	HIPCHK(c, hipGetLastError());
	std::vector<u32> &off = c->v_off; // (kept: see mpcgpu_ctx)
	off.resize(nn + 1);
	HIPCHK(c, hipMemcpyAsync(off.data() + 1, c->d_sizes.p, nn * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	off[0] = 0;
	u32 max_rec = 0;
	u64 run = 0;
	for (u64 b = 0; b < nn; ++b) { // exclusive scan in place: off[b+1] holds size(b) on entry
		const u32 sz = off[b + 1];
		max_rec = std::max(max_rec, sz);
		run += sz;
		if (run > 0xffffffffull) return 2; // block offsets are 32-bit (64 GB of records)
		off[b + 1] = (u32)run;
	}
	if (max_rec > 4095u) return 2; // a block's distance field holds 16 bits of bytes
	u32 buf_bytes = 0;
	size_t smem = 0;
	var_lds_geometry(threads, nbuf, &buf_bytes, &smem);
	c->var_mixed = false;
	const char *tiles_mode = getenv("MPCGPU_RELAX_TILES"); // "pairs": whole-record tiles of relax_var_kernel only; "band": band tiles whatever the size
	// MPCGPU_RELAX_SMALL_PAIRS=<n> (default 0 = never; the drop-in binary sets 40): stores of <= n sequences whose pairs fit the
	// whole-record tiles take those. A shrub of -super7 (<= 32 sequences, 412 of them in a 10 000-sequence run) is relaxed in 0.3 ms
	// either way, but the band path builds window records and band tables and cuts its tiles on the device first: 4.6 ms of launches
	// and round trips per store against 0.06 (profiles/r10d_small_store_time.log) — the 0.7 s that run lost in round 4.
	const int small_n = env_int("MPCGPU_RELAX_SMALL_PAIRS", 0);
	const bool small_pairs = small_n > 0 && n <= (u32)small_n && 2ull * max_rec * 16 <= buf_bytes && slots_ok(threads) &&
		!(tiles_mode && !strcmp(tiles_mode, "band"));
	const bool want_band = !(tiles_mode && !strcmp(tiles_mode, "pairs")) && !small_pairs && c->npairs < 0xffffffffull;
	bool pairs_ok = true;
	if (2ull * max_rec * 16 > buf_bytes || !slots_ok(threads)) { // not every single pair (two records, its cells) fits a tile of this geometry
		u32 bb1 = 0;
		size_t sm1 = 0;
		var_lds_geometry(1024, 1, &bb1, &sm1);
		if (2ull * max_rec * 16 > bb1 || !slots_ok(1024)) {
			if (!want_band) return 2;
			pairs_ok = false; // whole-record tiles are not an option for this run; band tiles may still be (relax_band)
		}
		// MPCGPU_RELAX_MIXED (default 1): keep the configured geometry for the pairs that fit it (two workgroups per CU: one's
		// staging overlaps the other's merges) and give the rest to a second launch of the one-workgroup geometry; 0: everything
		// to the one-workgroup geometry
		if (!pairs_ok) {}
		else if (!(threads == 1024 && nbuf == 1) && env_int("MPCGPU_RELAX_MIXED", 1)) {
			c->var_mixed = true; // (such runs end up with tiles of one pair: two records of 20..40 KB per step for ~3 slots of cells)
		}
		else { threads = 1024; nbuf = 1; buf_bytes = bb1; smem = sm1; }
	}
	const u64 pad_bytes = run * 16 + 4 * std::max<u64>(c->total_entries, 1);
	size_t freeb = 0, totb = 0;
	HIPCHK(c, hipMemGetInfo(&freeb, &totb));
	if (!(pad_bytes <= c->d_pad.cap + c->d_pos.cap || pad_bytes + ((u64)2 << 30) <= (u64)freeb)) return 2;
	c->d_rp.release(); c->d_ent.release(); c->d_mbase.release(); // slabs of an earlier run are not needed
	HIPCHK(c, c->d_pad.ensure(std::max<u64>(run, 1) * 16));
	HIPCHK(c, c->d_pos.ensure(4 * std::max<u64>(c->total_entries, 1))); // pos_f then pos_t, u16 each
	if (upload(c, c->d_rec_off, off)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream)); // `off` dies with this frame
	c->have_pad = true;
	c->pad_lcap1 = c->max_len;
	c->var_threads = threads; c->var_nbuf = nbuf; c->var_buf_bytes = buf_bytes;
	c->var_max_rec_blocks = max_rec; c->var_total_blocks = run;
	c->tiles_k0 = c->tiles_k1 = ~0ull;
	{
		char b[512];
		snprintf(b, sizeof(b), "variable-size dense records: %u x %u records, %.2f GB, mean %.0f B, largest %u B; relax_var_kernel, %s, %u staging buffer%s of %u B",
			n, n, (double)run * 16 / 1e9, (double)run * 16 / (double)nn, max_rec * 16,
			threads == 2048 ? "2 x 1024-thread workgroups per CU" : threads == 1024 ? "1 x 1024-thread workgroup per CU" :
			threads == 768 ? "2 x 768-thread workgroups per CU" : "2 x 512-thread workgroups per CU",
			nbuf, nbuf == 1 ? "" : "s", buf_bytes);
		c->store_desc = b;
		if (c->var_mixed) c->store_desc += " (pairs whose records do not fit it: 1 x 1024-thread workgroup per CU with 160 KB, second launch)";
		c->tiles_desc.clear(); c->relax_kernel_name.clear(); c->relax_fallback = false;
	}
	// band tables for relax_band_kernel (kernels_relaxb.h; MPCGPU_RELAX_TILES=pairs: whole-record tiles of relax_var_kernel only)
	c->band_ok = false;
	c->btiles_k0 = c->btiles_k1 = ~0ull;
	{
		const u32 nb1 = (c->max_len + MPC_RB_HB - 1) / MPC_RB_HB + 1;
		const u64 tab_bytes = (nn + 2 * c->npairs + 2ull * n) * nb1 * 4;
		size_t free2 = 0, tot2 = 0;
		HIPCHK(c, hipMemGetInfo(&free2, &tot2));
		if (want_band && (c->d_ovf_off.cap >= nn * nb1 * 4 || tab_bytes + ((u64)1 << 30) <= (u64)free2)) {
			HIPCHK(c, c->d_ovf_off.ensure(nn * nb1 * 4));
			HIPCHK(c, c->d_cell_off.ensure(std::max<u64>(c->npairs, 1) * nb1 * 4));
			HIPCHK(c, c->d_yr.ensure(std::max<u64>(c->npairs, 1) * nb1 * 4));
			HIPCHK(c, c->d_ovf_sum.ensure((u64)n * nb1 * 4));
			HIPCHK(c, c->d_ovf_maxc.ensure((u64)n * nb1 * 4));
			c->band_ok = true;
			c->band_nb1 = nb1;
			char b[256];
			snprintf(b, sizeof(b), "variable-size dense records: %u x %u records, %.2f GB, mean %.0f B, largest %u B, band index of %u rows",
				n, n, (double)run * 16 / 1e9, (double)run * 16 / (double)nn, max_rec * 16, (unsigned)MPC_RB_HB);
			c->store_desc = b; // (the tiles and the kernel are described when the first relax has cut them: relax_band)
		}
		if (!c->band_ok && !pairs_ok) { c->have_pad = false; return 2; }
		c->var_pairs_ok = pairs_ok;
	}
	StoreParams sp;
	fill_store_params(c, sp);
	if (trace_on()) {
		fprintf(stderr, "[mpcgpu] store: variable-size dense records, %u x %u records, %.2f GB (mean %.0f B, largest %u B), wg=%u nbuf=%u\n", n, n,
			(double)run * 16 / 1e9, (double)run * 16 / (double)nn, max_rec * 16, threads, nbuf);
		fflush(stderr);
	}
	TimedSpan ts;
	if (span_begin(c, 2, &ts)) return 1;
	MPC_LAUNCH(var_build_kernel, (u32)std::min<u64>(nn, (u64)c->prop.multiProcessorCount * 32), 64, (size_t)std::max(sp.lcap1, 1u) * 8, c->stream, sp);
	HIPCHK(c, hipGetLastError());
	// ---- window records (the Y operand of the direct-index merge): a second copy of the store whose rows are looked up by column.
	// Built when the rows are narrow — the windows then cost about what the blocks cost (1000 x L~400: 93 % of the rows span <= 4
	// columns); wide-row data (rdrp: half of the rows span >= 40 columns) keeps the walk of two block lists. MPCGPU_RELAX_FORM=walk: never.
	c->win_ok = false;
	{
		const char *form = getenv("MPCGPU_RELAX_FORM");
		if (c->band_ok && !(form && !strcmp(form, "walk"))) {
			HIPCHK(c, c->d_sizes.ensure(nn * 4));
			HIPCHK(c, c->d_tilefit.ensure(nn * 4)); // (scratch: value dwords per record)
			HIPCHK(c, c->d_wflag.ensure(4));
			HIPCHK(c, hipMemsetAsync(c->d_wflag.p, 0, 4, c->stream));
			MPC_LAUNCH(win_size_kernel, (u32)std::min<u64>(nn, (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp, c->d_sizes.as<u32>(),
				c->d_tilefit.as<u32>(), c->d_wflag.as<u32>());
			HIPCHK(c, hipGetLastError());
			std::vector<u32> &woff = c->v_woff; // (kept: see mpcgpu_ctx)
			woff.resize(nn + 1);
			u32 wide = 0;
			HIPCHK(c, hipMemcpyAsync(woff.data() + 1, c->d_sizes.p, nn * 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipMemcpyAsync(&wide, c->d_wflag.p, 4, hipMemcpyDeviceToHost, c->stream));
			HIPCHK(c, hipStreamSynchronize(c->stream));
			woff[0] = 0;
			u64 wrun = 0;
			for (u64 b = 0; b < nn; ++b) { wrun += woff[b + 1]; woff[b + 1] = (u32)std::min<u64>(wrun, 0xffffffffull); }
			const double ratio = (double)wrun / (double)std::max<u64>(run, 1);
			const double max_ratio = (double)env_int("MPCGPU_RELAX_WIN_PCT", 125) / 100.0;
			size_t free3 = 0, tot3 = 0;
			HIPCHK(c, hipMemGetInfo(&free3, &tot3));
			const u64 need = wrun * 16 + 4 * std::max<u64>(c->total_entries, 1) + nn * c->band_nb1 * 4;
			if (!wide && wrun <= 0xffffffffull && ratio <= max_ratio && (c->d_win.cap >= wrun * 16 || need + ((u64)1 << 30) <= (u64)free3)) {
				HIPCHK(c, c->d_win.ensure(std::max<u64>(wrun, 1) * 16));
				HIPCHK(c, c->d_pos_w.ensure(4 * std::max<u64>(c->total_entries, 1)));
				HIPCHK(c, c->d_wv_off.ensure(nn * c->band_nb1 * 4));
				HIPCHK(c, c->d_wsum.ensure((u64)n * c->band_nb1 * 4));
				HIPCHK(c, c->d_wmaxc.ensure((u64)n * c->band_nb1 * 4));
				if (upload(c, c->d_wrec_off, woff)) return 1;
				HIPCHK(c, hipStreamSynchronize(c->stream)); // `woff` dies with this block
				c->win_ok = true;
				c->win_total_blocks = wrun;
				fill_store_params(c, sp);
				MPC_LAUNCH(win_build_kernel, (u32)std::min<u64>(nn, (u64)c->prop.multiProcessorCount * 32), 64, (size_t)(std::max(sp.lcap1, 1u) + 1) * 4, c->stream, sp);
				HIPCHK(c, hipGetLastError());
				MPC_LAUNCH(win_pos_kernel, (u32)std::min<u64>(std::max<u64>(c->npairs, 1), (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp);
				HIPCHK(c, hipGetLastError());
				MPC_LAUNCH(ovf_stats_kernel, std::min<u32>(n, (u32)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp, c->d_wv_off.as<u32>(), c->band_nb1,
					c->d_wsum.as<u32>(), c->d_wmaxc.as<u32>(), 1);
				HIPCHK(c, hipGetLastError());
				char wb[160];
				snprintf(wb, sizeof(wb), " + window records for the Y operand (%.2f GB, %.0f %% of the blocks)", (double)wrun * 16 / 1e9, 100.0 * ratio);
				c->store_desc += wb;
			} else if (trace_on())
				fprintf(stderr, "[mpcgpu] store: no window records (%s; they would take %.0f %% of the block records' %.2f GB)\n",
					wide ? "a record's windows exceed 65535 values" : "rows too wide", 100.0 * ratio, (double)run * 16 / 1e9);
		}
	}
	if (c->band_ok) {
		MPC_LAUNCH(band_index_kernel, (u32)std::min<u64>(std::max<u64>(c->npairs, 1), (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp, c->band_nb1,
			c->d_cell_off.as<u32>(), c->d_yr.as<u32>());
		HIPCHK(c, hipGetLastError());
		MPC_LAUNCH(ovf_stats_kernel, std::min<u32>(n, (u32)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp, c->d_ovf_off.as<u32>(), c->band_nb1,
			c->d_ovf_sum.as<u32>(), c->d_ovf_maxc.as<u32>(), 0);
		HIPCHK(c, hipGetLastError());
	}
	if (span_end(c, &ts)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

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
	MPC_LAUNCH(packed_refresh_kernel, (u32)std::min<u64>(std::max<u64>(c->npairs, 1), (u64)c->prop.multiProcessorCount * 64), 64, 0, c->stream, sp, c->own_k0, c->own_k1);
	HIPCHK(c, hipGetLastError());
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
		&c->d_tile_next, &c->d_bp_in, &c->d_aln_res, &c->d_post_prof, &c->d_bp_keys, &c->d_bp_vals, &c->d_bp_tmp, &c->d_bp_runs, &c->d_aln_post, &c->d_aln_tb, &c->d_aln_rev,
		&c->d_rec_off, &c->d_sizes, &c->d_tilefit};
	for (DevBuf *b : all) b->release();
	c->h_bp_in.release();
	c->h_aln_res.release();
	c->h_ap.release();
	c->d_ap_off.release();
	c->d_chain_first.release(); c->d_chain_cnt.release();
	c->d_tiles2.release();
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

// the constant part of a forward/backward launch: sequences, PairHMM scores, tables, candidate buffers (the work list, queue and
// forward-plane scratch are set per launch)
static void fill_fb_params(mpcgpu_ctx *c, FbParams &fp, const u32 *pair_x, const u32 *pair_y, u32 capc, bool mega)
{
	fp.seq_code = c->d_seq_code.as<u8>(); fp.seq_off = c->d_seq_off.as<u64>(); fp.seq_len = c->d_seq_len.as<u32>();
	fp.tSM = c->start[0]; fp.tSI = c->start[1]; fp.tSJ = c->start[3]; // pairhmm.h:11-19: M,IX,IY,JX,JY
	fp.tMM = c->trans[0 * 5 + 0]; fp.tMI = c->trans[0 * 5 + 1]; fp.tMJ = c->trans[0 * 5 + 3];
	fp.tII = c->trans[1 * 5 + 1]; fp.tIM = c->trans[1 * 5 + 0];
	fp.tJJ = c->trans[3 * 5 + 3]; fp.tJM = c->trans[3 * 5 + 0];
	fp.thr = c->thr; fp.A = c->A; fp.match = c->d_match.as<float>(); fp.ins = c->d_ins.as<float>();
	fp.pair_x = pair_x; fp.pair_y = pair_y;
	fp.cand = c->d_cand.as<u64>(); fp.capc = capc; fp.cand_cnt = c->d_cand_cnt.as<u32>();
	fp.total = c->d_total.as<float>();
	fp.mg_prof = mega ? c->d_mg_prof.as<u64>() : nullptr; fp.mg_ins = mega ? c->d_mg_ins.as<float>() : nullptr;
	fp.mg_tab = mega ? c->d_mg_tab.as<float>() : nullptr; fp.mg_tab_floats = mega ? c->mg_tab_floats : 0;
	for (u32 f = 0; f < MPC_MEGA_FMAX; ++f) { fp.mg_base[f] = mega ? c->mg_base[f] : 0; fp.mg_alpha[f] = mega ? c->mg_alpha[f] : 0; }
	fp.bnd = nullptr; fp.bnd_stride = 0; fp.bnd_ld = 0; fp.fm_block = 0;
}

// Stage A over an explicit list of (x,y) sequence-index pairs (host arrays of np entries): the packed
// shard of those pairs, in list order, ends up in c->d_shard with sh_nnz / sh_ea.
static int stage_a(mpcgpu_ctx *c, u64 np, const u32 *px, const u32 *py)
{
	HIPCHK(c, hipSetDevice(c->device));
	c->have_shard = c->have_store = false;
	c->shard_is_list = true;
	c->list_x.assign(px, px + np); c->list_y.assign(py, py + np); // mpcgpu_get_list_sparse
	c->list_q0 = 0;
	if (!c->ap_keep) { c->ap_x.clear(); c->ap_y.clear(); }
	c->sh_k0 = 0; c->sh_k1 = np;
	c->sh_nnz.assign(np, 0);
	c->sh_ea.assign(np, 0.0f);
	c->work_cells = 0;
	c->sa_pairs = np; c->sa_chained = c->sa_chains = 0;
	const u64 hdr = shard_header_bytes(np);
	if (np == 0) {
		HIPCHK(c, c->d_shard.ensure(hdr));
		u64 h2[2] = {0, 0};
		HIPCHK(c, hipMemcpyAsync(c->d_shard.p, h2, 16, hipMemcpyHostToDevice, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		c->shard_bytes = hdr;
		c->have_shard = true;
		return 0;
	}
	// geometry of the shard
	u32 LXmax = 0, LYmax = 0;
	for (u64 k = 0; k < np; ++k) {
		const u32 LX = c->len[px[k]], LY = c->len[py[k]];
		LXmax = std::max(LXmax, LX); LYmax = std::max(LYmax, LY);
		c->work_cells += (u64)(LX + 1) * (LY + 1);
	}
	// X longer than 64*MPC_HMAX rows: row-block (LONG) kernels, 16-bit row/column candidate keys
	const int long_h_env = env_int("MPCGPU_FB_LONG_H", 0); // 0 = chosen below, 1 / 4 / 7 = forced (1: tests reach several blocks with short sequences)
	// Row sequences from 769 residues on take the row-block kernels: one block of 13..16 rows per lane needs 177..219 VGPRs (2 waves
	// per SIMD), blocks of 4 rows per lane 166 (3 waves): 300 x L~1000 fb 439 -> 366 ms; up to 12 rows per lane (<= 167 VGPRs)
	// the single block wins (400 x L~600: 228 against 297 ms). 1025 is where a single block stops being possible.
	const u32 long_min = (u32)std::min(std::max(env_int("MPCGPU_FB_LONG_MIN", 64 * 12 + 1), 2), 64 * MPC_HMAX + 1);
	u32 LXlong = 0, LYlong = 0; // extents over the LONG pairs
	for (u64 k = 0; k < np; ++k) {
		const u32 LX = c->len[px[k]], LY = c->len[py[k]];
		if (LX >= long_min) { LXlong = std::max(LXlong, LX); LYlong = std::max(LYlong, LY); }
	}
	if (LXlong > MPC_KEY_COL_MASK_LONG || LYlong > MPC_KEY_COL_MASK_LONG)
		return fail(c, "mpcgpu_calc_posteriors: a pair of %u x %u positions is beyond this build's limit of %u per sequence "
			"once the row sequence is longer than %u", LXlong, LYlong, MPC_KEY_COL_MASK_LONG, long_min - 1);
	const u32 Lmax = std::max(LXmax, LYmax);
	u32 capc = (u32)std::max(env_int("MPCGPU_CAND_PER_ROW", 12), 1) * Lmax;
	capc = std::max(capc, 1024u);
	const int waves_per_block = 4, block = 64 * waves_per_block;
	const u32 cus = (u32)c->prop.multiProcessorCount;
	const bool mega = c->have_mega;
	const size_t fb_smem = (mega ? (size_t)c->mg_tab_floats : (size_t)c->A * c->A + c->A) * sizeof(float) +
		(size_t)env_int("MPCGPU_FB_LDS_PAD_KB", 0) * 1024; // the pad (measurement only) lowers the resident workgroups per CU

	// Host side of a batch (sizing, bins, launch order): prepared for batch b+1 while the device runs batch b.
	struct BatchPrep {
		bool valid = false;
		u64 b0 = 0, B = 0;
		u32 capc = 0;
		std::vector<u32> bx, by, order;
		u32 hcount[MPC_HMAX + 2];
		// chains of pairs with the same row sequence (fb_chain_kernel): members in `order` behind the single pairs
		std::vector<u32> chain_first, chain_cnt;
		u32 ccount[MPC_HMAX + 1];   // chains per rows-per-lane bin
		u32 cvmax[MPC_HMAX + 1];    // longest virtual column axis of a bin's chains
	};
	// Chains: consecutive pairs of the list with the same row sequence (the all-pairs order is full of them), each with
	// LY + 1 >= T (kernels_fbc.h), up to MPCGPU_FB_CHAIN_MAX (default 16) pairs and as many columns as the forward M planes of the
	// resident waves may take (a quarter of the free memory, 32 GB at most). MPCGPU_FB_CHAIN=0: every pair on its own (fb_kernel).
	const bool chain_on = !mega && env_int("MPCGPU_FB_CHAIN", 1) != 0;
	const u32 chain_max = (u32)std::min(std::max(env_int("MPCGPU_FB_CHAIN_MAX", 16), 2), MPC_CHAIN_MAX);
	const bool chain_grade = env_int("MPCGPU_FB_CHAIN_GRADE", 1) != 0; // 0: no shorter chains at the end of a launch (tests)
	const size_t fbc_smem = ((size_t)c->A * c->A + c->A) * sizeof(float) + (size_t)waves_per_block * MPC_CHAIN_TAB_BYTES;
	u32 chain_vcap[MPC_HMAX + 1];
	for (u32 H = 0; H <= MPC_HMAX; ++H) chain_vcap[H] = 0;
	if (chain_on) {
		size_t freeb = 0, totb = 0;
		HIPCHK(c, hipMemGetInfo(&freeb, &totb));
		// (a quarter of what is free, 32 GB at most — and no more than MPCGPU_SCRATCH_GB where that is set: several contexts on one
		// device, e.g. the eight of tests/test_gpu_parity.py::test_group_of_eight_contexts_config3_digests, each see the same free memory)
		const char *scratch_set = getenv("MPCGPU_SCRATCH_GB");
		const u64 fm_budget = std::min<u64>(std::min<u64>((u64)32 << 30, (scratch_set && *scratch_set) ? (u64)std::max(atoi(scratch_set), 1) << 30 : ~0ull),
			(u64)((freeb + c->d_fm.cap) * 0.25));
		for (u32 H = 1; H <= MPC_HMAX; ++H) {
			const u64 waves = (u64)cus * (u32)occ_fbc_h((int)H, block, fbc_smem) * waves_per_block;
			const u64 steps = fm_budget / (waves * H * 64 * 4);
			chain_vcap[H] = steps > 64 + 2 ? (u32)std::min<u64>(steps - 64, 1u << 24) : 0;
		}
	}
	auto prepare = [&](u64 b0, BatchPrep &P) -> int {
		// ---- batch sizing: candidates + fixed-stride records per pair
		const u64 res_stride = (u64)LXmax + LYmax + 4 * (u64)capc;
		const u64 per_pair = (u64)capc * 8 + res_stride * 4 + 64;
		size_t freeb = 0, totb = 0;
		HIPCHK(c, hipMemGetInfo(&freeb, &totb));
		// the scratch of the previous batch (or of an overflow retry) is already owned and gets reused: count it as available
		const u64 owned = (u64)c->d_cand.cap + c->d_res.cap + c->d_fm.cap;
		// 16 GB: four batches at 1000 x L~400. Fewer, larger batches save the tails of waves that finish alone (24 GB / 3 batches: fb
		// 551 -> 545 ms, step 1832 -> 1822; 32 / 2: 541, 1840: the first batch's host preparation is not covered by device work) —
		// but the scratch is allocated inside the first call, and with 24 GB that call's allocations took 1.2 s longer (some hipMalloc
		// crosses a slow path): `muscle_gpu -align` of the same 1000 sequences 5.07 -> 6.42 s; 8 GB: 4.97 s, 4 GB: 5.22 s
		// (profiles/r05a, r05c, r05d)
		u64 budget = std::min<u64>((u64)env_int("MPCGPU_SCRATCH_GB", 16) << 30, (u64)((freeb + owned) * 0.4));
		u64 B = std::max<u64>(1, std::min<u64>(np - b0, budget / per_pair));
		B = std::min<u64>(B, 1u << 22);
		P.b0 = b0; P.B = B; P.capc = capc;
		// ---- bin by H, order by work (longest first)
		// one 64-bit key per pair: bin (5 bits) | work, descending (37 bits) | index (22 bits) — a plain integer sort (with a
		// three-array comparator it cost 9 ms per 125 000 pairs)
		P.bx.resize(B); P.by.resize(B); P.order.resize(B);
		std::vector<u64> keys;
		keys.reserve(B);
		for (u32 h = 0; h < MPC_HMAX + 2; ++h) P.hcount[h] = 0; // bin MPC_HMAX+1: the row-block (LONG) pairs
		for (u32 h = 0; h <= MPC_HMAX; ++h) P.ccount[h] = P.cvmax[h] = 0;
		P.chain_first.clear(); P.chain_cnt.clear();
		for (u64 q = 0; q < B; ++q) { P.bx[q] = px[b0 + q]; P.by[q] = py[b0 + q]; }
		// chains first: runs of consecutive pairs with the same row sequence. Every pair the chain kernel can take goes to it (a
		// pair on its own is a chain of one), so a bin is ONE launch; the chains of a launch are served longest first, and the
		// last ones are cut shorter (4, 2, 1 pairs for about one round of the resident waves each) so that the waves finish together
		struct Chain { u32 q0, cnt, H; u64 work; };
		std::vector<Chain> chains;
		std::vector<unsigned char> chained(B, 0);
		if (chain_on) {
			auto work_of = [&](u32 q0, u32 cnt, u32 H, u32 T) {
				u64 V = 0;
				for (u32 k = 0; k < cnt; ++k) V += c->len[P.by[q0 + k]] + 1;
				return (V + T) * H;
			};
			u64 q = 0;
			while (q < B) {
				const u32 LX = c->len[P.bx[q]];
				const u32 H = (LX + 63) / 64;
				if (LX >= long_min || H < 1 || H > MPC_HMAX || c->len[P.by[q]] + 1 > chain_vcap[H]) { ++q; continue; }
				const u32 T = (LX + H - 1) / H;
				u64 e = q;
				u64 V = 0;
				while (e < B && P.bx[e] == P.bx[q] && e - q < chain_max) {
					const u32 LY = c->len[P.by[e]];
					if (LY + 1 < T || V + LY + 1 > chain_vcap[H]) break;
					V += LY + 1;
					++e;
				}
				if (e == q) e = q + 1; // a pair too short to chain: on its own
				chains.push_back({(u32)q, (u32)(e - q), H, 0});
				for (u64 k = q; k < e; ++k) chained[k] = 1;
				q = e;
			}
			for (Chain &ch : chains) { const u32 LX = c->len[P.bx[ch.q0]]; ch.work = work_of(ch.q0, ch.cnt, ch.H, (LX + ch.H - 1) / ch.H); }
			auto by_bin_and_work = [](const Chain &a, const Chain &b) { return a.H != b.H ? a.H < b.H : a.work != b.work ? a.work > b.work : a.q0 < b.q0; };
			std::sort(chains.begin(), chains.end(), by_bin_and_work);
			// the short end of every bin
			std::vector<Chain> graded;
			graded.reserve(chains.size() * 2);
			size_t lo = 0;
			while (lo < chains.size()) {
				size_t hi = lo;
				while (hi < chains.size() && chains[hi].H == chains[lo].H) ++hi;
				const u32 H = chains[lo].H;
				const u64 waves = (u64)cus * (u32)occ_fbc_h((int)H, block, fbc_smem) * waves_per_block;
				const u64 gw = std::max<u64>(waves * (u64)std::max(env_int("MPCGPU_FB_CHAIN_GRADE_Q", 4), 1) / 4, 1); // pairs per grade, in quarters of a round of resident waves
				u64 seen = 0; // pairs, counted from the end of the bin
				for (size_t k = hi; k-- > lo;) {
					const Chain &ch = chains[k];
					const u32 piece = !chain_grade ? ch.cnt : seen < gw ? 1u : seen < 3 * gw ? 2u : seen < 7 * gw ? 4u : ch.cnt;
					seen += ch.cnt;
					const u32 LX = c->len[P.bx[ch.q0]];
					for (u32 o = 0; o < ch.cnt; o += piece) {
						const u32 n = std::min(piece, ch.cnt - o);
						graded.push_back({ch.q0 + o, n, H, work_of(ch.q0 + o, n, H, (LX + H - 1) / H)});
					}
				}
				lo = hi;
			}
			std::sort(graded.begin(), graded.end(), by_bin_and_work);
			chains.swap(graded);
			for (const Chain &ch : chains) {
				u64 V = 0;
				for (u32 k = 0; k < ch.cnt; ++k) V += c->len[P.by[ch.q0 + k]] + 1;
				P.cvmax[ch.H] = std::max<u32>(P.cvmax[ch.H], (u32)V);
			}
		}
		for (u64 q = 0; q < B; ++q) {
			if (chained[q]) continue;
			const u32 LX = c->len[P.bx[q]], LY = c->len[P.by[q]];
			const bool lng = LX >= long_min;
			const u32 H = lng ? MPC_HMAX + 1 : (LX + 63) / 64;
			P.hcount[H]++;
			const u64 wk = lng ? (u64)LX * LY : (u64)(LY + (LX + H - 1) / H) * H; // < 2^37 (lengths < 2^16 when LONG, < 2^22 otherwise with H <= 16)
			keys.push_back(((u64)H << 59) | ((((u64)1 << 37) - 1 - wk) << 22) | q);
		}
		std::sort(keys.begin(), keys.end());
		u64 at = 0;
		for (; at < keys.size(); ++at) P.order[at] = (u32)(keys[at] & (((u64)1 << 22) - 1));
		// the chains: by bin, longest first; their members follow the other pairs in `order`
		for (const Chain &ch : chains) {
			P.ccount[ch.H]++;
			P.chain_first.push_back((u32)at);
			P.chain_cnt.push_back(ch.cnt);
			for (u32 k = 0; k < ch.cnt; ++k) P.order[at++] = ch.q0 + k;
		}
		P.valid = true;
		return 0;
	};
	BatchPrep cur, nxt;
	const bool host_trace = env_int("MPCGPU_TRACE_HOST", 0) != 0; // diagnostics: host wall time between the device phases of a batch
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double lap_t[6] = {0, 0, 0, 0, 0, 0}, t_prev = host_trace ? now() : 0.0;
	auto lap = [&](int k) { if (host_trace) { const double t = now(); lap_t[k] += t - t_prev; t_prev = t; } };
	u64 words_done = 0; // record words packed so far
	u64 done = 0;
	while (done < np) {
		if (!(cur.valid && cur.b0 == done && cur.capc == capc) && prepare(done, cur)) return 1;
		const u64 res_stride = (u64)LXmax + LYmax + 4 * (u64)capc;
		const u64 B = cur.B;
		const std::vector<u32> &bx = cur.bx, &by = cur.by, &order = cur.order;
		const u32 *hcount = cur.hcount;
		lap(0);
		if (upload(c, c->d_bx, bx) || upload(c, c->d_by, by) || upload(c, c->d_order, order)) return 1;
		if (!cur.chain_first.empty() && (upload(c, c->d_chain_first, cur.chain_first) || upload(c, c->d_chain_cnt, cur.chain_cnt))) return 1;
		HIPCHK(c, c->d_cand.ensure(B * capc * 8));
		HIPCHK(c, c->d_cand_cnt.ensure(B * 4));
		HIPCHK(c, c->d_total.ensure(B * 4));
		HIPCHK(c, c->d_res.ensure(B * res_stride * 4));
		HIPCHK(c, c->d_nnz.ensure(B * 4));
		HIPCHK(c, c->d_ea.ensure(B * 4));
		HIPCHK(c, c->d_flags.ensure(B * 4));
		HIPCHK(c, c->d_queue.ensure(4 * (2 * MPC_HMAX + 4))); // single pairs per bin, then chains per bin
		HIPCHK(c, hipMemsetAsync(c->d_queue.p, 0, 4 * (2 * MPC_HMAX + 4), c->stream));

		FbParams fp;
		fill_fb_params(c, fp, c->d_bx.as<u32>(), c->d_by.as<u32>(), capc, mega);
		// row-list post kernel (no sorts, 3 LDS trips per EA row) when LY fits its LDS arrays; MPCGPU_POST=sort forces the general one
		const char *post_mode = getenv("MPCGPU_POST");
		// (up to ~12 000 positions: three arrays of one word per position + the sorted-list buffer in the CU's LDS)
		const bool post_rows = !(post_mode && !strcmp(post_mode, "sort")) &&
			((size_t)LXmax + 2 + 2 * ((size_t)LYmax + 2)) * 4 + 8 + 8 * (size_t)std::max(env_int("MPCGPU_POST_SORT_CAP", 1024), 2) <= 150 * 1024;

		TimedSpan sp;
		u32 pos = 0;
		if (hcount[MPC_HMAX + 1]) { // ---- row-block pairs (the order lists them last: bins ascend)
			const u32 cnt = hcount[MPC_HMAX + 1];
			u32 first = 0;
			for (u32 H = 1; H <= MPC_HMAX; ++H) first += hcount[H];
			const u32 ld = (LYlong + 2 + 63) & ~63u;
			auto planes = [&](u32 h, u32 *nb, u64 *blk) { *nb = (LXlong + 64 * h - 1) / (64 * h); *blk = (u64)(LYlong + 64) * h * 64; return *blk * *nb; };
			// resident waves are bounded by the forward M planes they keep (LX*LY floats each)
			size_t freeb2 = 0, totb2 = 0;
			HIPCHK(c, hipMemGetInfo(&freeb2, &totb2));
			// (up to 45 % of what is free, counting the plane buffer already owned: 36 MB per 3000 x 3000 pair — with the 16 GB the
			// other scratch is held to, 444 waves were resident where the chip takes 2048. The buffer stays allocated — hipMalloc and
			// hipFree of ~100 GB take seconds — and is given back only when the store needs the room: mpcgpu_store_import)
			const char *scratch_env = getenv("MPCGPU_SCRATCH_GB");
			const u64 fm_budget = std::min<u64>((scratch_env && *scratch_env) ? (u64)atoi(scratch_env) << 30 : ~0ull, (u64)((freeb2 + c->d_fm.cap) * 0.45));
			// rows per lane: 7 (217 VGPRs, 2 waves per SIMD), or 4 (166 VGPRs, 3 waves per SIMD; more blocks, more line-buffer
			// traffic) when the pairs and the memory for their forward planes can keep more than 2 waves per SIMD busy
			// (100 x L~3000: 517 -> 407 ms; 64 x L~6000, 875 waves fit: 1985 ms with 7 rows, 2190 with 4)
			u32 long_h = long_h_env == 1 ? 1u : long_h_env == MPC_LONG_H_SMALL ? (u32)MPC_LONG_H_SMALL : (u32)MPC_LONG_H;
			u32 nbmax = 0;
			u64 fm_block = 0;
			if (long_h_env == 0) {
				const u64 stride_small = planes(MPC_LONG_H_SMALL, &nbmax, &fm_block);
				const u64 waves_small = std::min<u64>(cnt, fm_budget / (stride_small * 4 + 16ull * ld * 4));
				if (waves_small > (u64)cus * 4 * 2) long_h = MPC_LONG_H_SMALL;
			}
			const u64 fm_stride = planes(long_h, &nbmax, &fm_block);
			const u64 max_waves = fm_budget / (fm_stride * 4 + 16ull * ld * 4);
			if (max_waves < 1)
				return fail(c, "mpcgpu_calc_posteriors: not enough device memory for the forward plane of a %u x %u pair", LXlong, LYlong);
			const u32 occ = (u32)occ_fb_long((int)long_h, mega, block, fb_smem);
			u32 grid = std::min<u32>((cnt + waves_per_block - 1) / waves_per_block, cus * occ);
			grid = (u32)std::max<u64>(std::min<u64>(grid, max_waves / waves_per_block), 1);
			const u32 wpb = max_waves < (u64)waves_per_block ? (u32)max_waves : (u32)waves_per_block; // fewer waves per workgroup when memory is that tight
			HIPCHK(c, c->d_fm.ensure((u64)grid * wpb * fm_stride * 4));
			HIPCHK(c, c->d_bnd.ensure((u64)grid * wpb * 16 * ld * 4));
			if (trace_on()) {
				fprintf(stderr, "[mpcgpu] fb row blocks: H=%u pairs=%u blocks<=%u grid=%u x %u waves occ=%u fm=%.1f MB\n", long_h, cnt, nbmax,
					grid, wpb, occ, (double)grid * wpb * fm_stride * 4 / 1048576.0);
				fflush(stderr);
			}
			fp.order = c->d_order.as<u32>() + first; fp.count = cnt;
			fp.queue = c->d_queue.as<u32>() + MPC_HMAX + 1;
			fp.fm_scratch = c->d_fm.as<float>(); fp.fm_stride = fm_stride; fp.fm_block = fm_block;
			fp.bnd = c->d_bnd.as<float>(); fp.bnd_stride = 16ull * ld; fp.bnd_ld = ld;
			if (span_begin(c, 0, &sp)) return 1;
			launch_fb_long((int)long_h, mega, fp, grid, 64 * wpb, fb_smem, c->stream);
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &sp)) return 1;
			fp.bnd = nullptr; fp.bnd_stride = 0; fp.bnd_ld = 0; fp.fm_block = 0;
		}
		for (u32 H = 1; H <= MPC_HMAX; ++H) {
			if (!hcount[H]) continue;
			const u32 cnt = hcount[H];
			// persistent waves: exactly as many workgroups as the chip keeps resident (VGPR-limited)
			const u32 occ = (u32)occ_fb_h((int)H, mega, block, fb_smem);
			u32 grid = std::min<u32>((cnt + waves_per_block - 1) / waves_per_block, cus * occ);
			grid = std::max(grid, 1u);
			const u64 fm_stride = (u64)(LYmax + 64) * H * 64;
			HIPCHK(c, c->d_fm.ensure((u64)grid * waves_per_block * fm_stride * 4));
			if (trace_on()) {
				fprintf(stderr, "[mpcgpu] fb H=%u pairs=%u grid=%u block=%d occ=%u capc=%u batch=%llu fm=%.1f MB\n", H, cnt, grid,
					block, occ, capc, B, (double)grid * waves_per_block * fm_stride * 4 / 1048576.0);
				fflush(stderr);
			}
			fp.order = c->d_order.as<u32>() + pos; fp.count = cnt;
			fp.queue = c->d_queue.as<u32>() + H;
			fp.fm_scratch = c->d_fm.as<float>(); fp.fm_stride = fm_stride;
			if (span_begin(c, 0, &sp)) return 1;
			launch_fb_h((int)H, mega, fp, grid, block, fb_smem, c->stream);
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &sp)) return 1;
			pos += cnt;
		}
		u32 cpos = 0;
		u64 batch_chains = 0, batch_chained = 0;
		for (u32 c2 : cur.chain_cnt) if (c2 >= 2) { ++batch_chains; batch_chained += c2; }
		for (u32 H = 1; H <= MPC_HMAX; ++H) { // chains (kernels_fbc.h)
			if (!cur.ccount[H]) continue;
			const u32 cnt = cur.ccount[H];
			const u32 occ = (u32)occ_fbc_h((int)H, block, fbc_smem);
			const u32 grid = std::max(std::min<u32>((cnt + waves_per_block - 1) / waves_per_block, cus * occ), 1u);
			const u64 fm_stride = (u64)(cur.cvmax[H] + 64) * H * 64;
			HIPCHK(c, c->d_fm.ensure((u64)grid * waves_per_block * fm_stride * 4));
			if (trace_on()) {
				fprintf(stderr, "[mpcgpu] fb chains H=%u chains=%u grid=%u occ=%u longest axis=%u fm=%.1f MB\n", H, cnt, grid, occ, cur.cvmax[H],
					(double)grid * waves_per_block * fm_stride * 4 / 1048576.0);
				fflush(stderr);
			}
			FbChainParams cp;
			cp.f = fp;
			cp.f.order = c->d_order.as<u32>(); cp.f.count = cnt;
			cp.f.queue = c->d_queue.as<u32>() + (MPC_HMAX + 2) + H;
			cp.f.fm_scratch = c->d_fm.as<float>(); cp.f.fm_stride = fm_stride;
			cp.chain_first = c->d_chain_first.as<u32>() + cpos; cp.chain_cnt = c->d_chain_cnt.as<u32>() + cpos;
			if (span_begin(c, 0, &sp)) return 1;
			launch_fbc_h((int)H, cp, grid, block, fbc_smem, c->stream);
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &sp)) return 1;
			cpos += cnt;
		}
		// ---- finish: probabilities, sort, EA, sparsify
		if (post_rows) {
			PostRowsParams pr;
			pr.pair_x = c->d_bx.as<u32>(); pr.pair_y = c->d_by.as<u32>(); pr.seq_len = c->d_seq_len.as<u32>();
			pr.cand = c->d_cand.as<u64>(); pr.capc = capc; pr.cand_cnt = c->d_cand_cnt.as<u32>();
			pr.use_fma = c->use_fma;
			pr.lx_cap = LXmax + 2; pr.ly_cap = LYmax + 2;
			// LDS list of a pair's candidates (larger lists cost resident waves, pairs that exceed it sort through HBM scratch; per
			// batch of 125 000 pairs at L~400: 512 entries 16.9 ms, 768: 12.3, 896: 11.9, 1024: 11.7, 1280: 12.7, 1408 (holds every
			// pair): 14.0, 1664: 15.8)
			const u32 sort_cap = (u32)std::max(env_int("MPCGPU_POST_SORT_CAP", 1024), 2);
			pr.sort_cap = std::min<u32>(capc, sort_cap);
			pr.sort_stride = capc;
			pr.batch = (u32)std::min(std::max(env_int("MPCGPU_POST_BATCH", 64), 1), 64);
			const size_t fixed_lds = ((((size_t)pr.lx_cap + 2 * (size_t)pr.ly_cap) * 4 + 7) & ~(size_t)7);
			if (fixed_lds + (size_t)pr.sort_cap * 8 > 150 * 1024) pr.sort_cap = (u32)((150 * 1024 - fixed_lds) / 8); // long sequences: the arrays per position come first
			const size_t smem = fixed_lds + (size_t)pr.sort_cap * 8;
			if (smem > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void *)post_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
			int pocc = 0;
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pocc, (const void *)post_rows_kernel, 64, smem) != hipSuccess || pocc < 1) pocc = 1;
			// The persistent grid must not be larger than what is really resident, or its last workgroups run as a second round
			// (measured, per batch of 125 000 pairs: 15 248 B of LDS, 10 workgroups per CU reported: 12.7 ms; 15 760 B, still 10
			// reported: 22.5 ms; 17 296 B, 9 reported: 14.1 ms — 64-thread workgroups stop fitting at ~152 KB per CU, not 160)
			pocc = std::max(1, std::min(pocc, (int)((152 * 1024) / smem)));
			const u32 pgrid = (u32)std::min<u64>(B, (u64)cus * (u32)pocc);
			HIPCHK(c, c->d_sort_scratch.ensure(capc > pr.sort_cap ? (u64)pgrid * pr.sort_stride * 8 : 8));
			pr.sort_scratch = c->d_sort_scratch.as<u64>();
			pr.res = c->d_res.as<u32>(); pr.res_stride = res_stride;
			pr.nnz = c->d_nnz.as<u32>(); pr.ea = c->d_ea.as<float>(); pr.flags = c->d_flags.as<u32>();
			pr.count = (u32)B;
			pr.long_min = long_min;
			pr.prof = nullptr;
			const bool post_prof = env_int("MPCGPU_POST_PROFILE", 0) != 0; // measurement only: phase clocks of workgroup 0
			if (post_prof) {
				HIPCHK(c, c->d_post_prof.ensure(8 * 8));
				HIPCHK(c, hipMemsetAsync(c->d_post_prof.p, 0, 64, c->stream));
				pr.prof = c->d_post_prof.as<u64>();
			}
			if (trace_on()) { fprintf(stderr, "[mpcgpu] post rows: list of %u candidates in LDS, lds=%zu B blocks/CU=%d grid=%u\n", pr.sort_cap, smem, pocc, pgrid); fflush(stderr); }
			if (span_begin(c, 1, &sp)) return 1;
			MPC_LAUNCH(post_rows_kernel, pgrid, 64, smem, c->stream, pr);
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &sp)) return 1;
			if (post_prof) {
				u64 ticks[8];
				HIPCHK(c, hipMemcpyAsync(ticks, pr.prof, 64, hipMemcpyDeviceToHost, c->stream));
				HIPCHK(c, hipStreamSynchronize(c->stream));
				const u64 npairs0 = (B + pgrid - 1) / pgrid; // pairs workgroup 0 handled
				fprintf(stderr, "[mpcgpu] post_rows_kernel, workgroup 0, %llu pairs, us per pair (100 MHz clock): prob+histogram %.1f, scan+scatter %.1f, "
					"row sort %.1f, EA %.1f, kept entries %.1f, column ranks %.1f\n", (u64)npairs0, ticks[0] / 100.0 / npairs0, ticks[1] / 100.0 / npairs0,
					ticks[2] / 100.0 / npairs0, ticks[3] / 100.0 / npairs0, ticks[4] / 100.0 / npairs0, ticks[5] / 100.0 / npairs0);
			}
		} else {
		PostParams pp;
		pp.pair_x = c->d_bx.as<u32>(); pp.pair_y = c->d_by.as<u32>(); pp.seq_len = c->d_seq_len.as<u32>();
		pp.cand = c->d_cand.as<u64>(); pp.capc = capc; pp.cand_cnt = c->d_cand_cnt.as<u32>();
		pp.use_fma = c->use_fma;
		// LDS sort buffer capacity (entries, power of two): pairs with more candidates sort in the
		// global scratch. The kernel is latency-bound (one wave per pair), so LDS per workgroup trades
		// against resident waves; MPCGPU_POST_SORT_CAP overrides for tuning.
		pp.sort_cap = std::min<u32>(next_pow2(capc), next_pow2((u32)std::max(env_int("MPCGPU_POST_SORT_CAP", 1024), 2)));
		pp.srow_cap = std::min<u32>(LYmax + 1, 2048u);
		const size_t psmem0 = (size_t)pp.sort_cap * 8 + (size_t)pp.srow_cap * 2 * 4;
		int pocc = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pocc, (const void *)post_kernel, 64, psmem0) != hipSuccess || pocc < 1) pocc = 8;
		if (psmem0) pocc = std::max(1, std::min(pocc, (int)((152 * 1024) / psmem0))); // as for post_rows_kernel above
		const u32 pgrid = (u32)std::min<u64>(B, (u64)cus * (u32)pocc);
		if (trace_on()) { fprintf(stderr, "[mpcgpu] post: sort_cap=%u lds=%zu B blocks/CU=%d grid=%u\n", pp.sort_cap, psmem0, pocc, pgrid); fflush(stderr); }
		pp.sort_stride = next_pow2(capc);
		pp.srow_stride = 2 * ((u64)LYmax + 1);
		const bool need_sort_scr = next_pow2(capc) > pp.sort_cap, need_srow_scr = LYmax + 1 > pp.srow_cap;
		HIPCHK(c, c->d_sort_scratch.ensure(need_sort_scr ? (u64)pgrid * pp.sort_stride * 8 : 8));
		HIPCHK(c, c->d_srow_scratch.ensure(need_srow_scr ? (u64)pgrid * pp.srow_stride * 4 : 8));
		pp.sort_scratch = c->d_sort_scratch.as<u64>(); pp.srow_scratch = c->d_srow_scratch.as<float>();
		pp.res = c->d_res.as<u32>(); pp.res_stride = res_stride;
		pp.nnz = c->d_nnz.as<u32>(); pp.ea = c->d_ea.as<float>(); pp.flags = c->d_flags.as<u32>();
		pp.count = (u32)B;
		pp.long_min = long_min;
		const size_t psmem = (size_t)pp.sort_cap * 8 + (size_t)pp.srow_cap * 2 * 4;
		if (span_begin(c, 1, &sp)) return 1;
		MPC_LAUNCH(post_kernel, pgrid, 64, psmem, c->stream, pp);
		HIPCHK(c, hipGetLastError());
		if (span_end(c, &sp)) return 1;
		}
		// ---- the next batch's host work, while the device runs this one (before the copies below: a copy into pageable host
		// memory returns only when it is done)
		lap(1);
		nxt.valid = false;
		if (done + B < np && prepare(done + B, nxt)) return 1;
		lap(2);
		// ---- sizes back, overflow check, pack
		std::vector<u32> &flags = c->v_flags; // (kept: see mpcgpu_ctx)
		flags.resize(B);
		HIPCHK(c, hipMemcpyAsync(&c->sh_nnz[done], c->d_nnz.p, B * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(&c->sh_ea[done], c->d_ea.p, B * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipMemcpyAsync(flags.data(), c->d_flags.p, B * 4, hipMemcpyDeviceToHost, c->stream));
		HIPCHK(c, hipStreamSynchronize(c->stream));
		lap(3);
		bool overflow = false;
		for (u64 q = 0; q < B; ++q) overflow = overflow || (flags[q] & 1u);
		if (overflow) {
			if (capc >= LXmax * (u64)LYmax)
				return fail(c, "mpcgpu_calc_posteriors: candidate overflow at full capacity (internal error)");
			capc = (u32)std::min<u64>((u64)capc * 2, (u64)LXmax * LYmax);
			continue; // redo this batch with a larger candidate capacity
		}
		std::vector<u64> &dstbase = c->v_dstbase, &recw = c->v_recw; // (kept: see mpcgpu_ctx)
		dstbase.resize(B); recw.resize(B);
		u64 w = words_done;
		for (u64 q = 0; q < B; ++q) {
			recw[q] = rec_words(c->len[bx[q]], c->len[by[q]], c->sh_nnz[done + q]);
			dstbase[q] = hdr / 4 + w;
			w += recw[q];
		}
		// capacity estimate for the whole shard from the words seen so far
		const double per = double(w) / double(done + B);
		const u64 est = hdr + (u64)(per * 1.05 * double(np) + 1024) * 4;
		HIPCHK(c, c->d_shard.ensure(std::max<u64>(est, hdr + w * 4), true, c->stream));
		if (upload(c, c->d_dstbase, dstbase) || upload(c, c->d_recwords, recw)) return 1;
		if (span_begin(c, 1, &sp)) return 1;
		MPC_LAUNCH(pack_kernel, (u32)std::min<u64>(B, (u64)cus * 8), 256, 0, c->stream, c->d_res.as<u32>(), res_stride,
			c->d_dstbase.as<u64>(), c->d_recwords.as<u64>(), c->d_shard.as<u32>(), (u32)B);
		HIPCHK(c, hipGetLastError());
		if (span_end(c, &sp)) return 1;
		HIPCHK(c, hipStreamSynchronize(c->stream));
		words_done = w;
		// what the LAST batch left in the scratch buffers (mpcgpu_align_pairs reads the candidate lists of a one-batch stage)
		c->sa_b0 = done; c->sa_B = B; c->sa_capc = capc; c->sa_post_rows = post_rows; c->sa_long_min = long_min;
		done += B;
		c->sa_chains += batch_chains; c->sa_chained += batch_chained;
		std::swap(cur, nxt);
		lap(4);
	}
	if (host_trace)
		fprintf(stderr, "[mpcgpu] stage A host seconds: prepare (first batch / retries) %.4f, uploads + launches %.4f, next batch prepared %.4f, "
			"waiting for the device %.4f, sizes -> pack -> wait %.4f\n", lap_t[0], lap_t[1], lap_t[2], lap_t[3], lap_t[4]);
	// header
	std::vector<u8> &h = c->v_shdr; // (kept: see mpcgpu_ctx)
	h.assign(hdr, 0);
	u64 h2[2] = {np, words_done};
	memcpy(h.data(), h2, 16);
	memcpy(h.data() + 16, c->sh_nnz.data(), np * 4);
	memcpy(h.data() + 16 + np * 4, c->sh_ea.data(), np * 4);
	HIPCHK(c, hipMemcpyAsync(c->d_shard.p, h.data(), hdr, hipMemcpyHostToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	c->shard_bytes = hdr + words_done * 4;
	c->have_shard = true;
	return 0;
}

int mpcgpu_calc_posteriors(mpcgpu_ctx *c, uint64_t k0, uint64_t k1)
{
	if (!c) return 1;
	if (c->n == 0) return fail(c, "mpcgpu_calc_posteriors: call mpcgpu_set_seqs first");
	if (k0 > k1 || k1 > c->npairs) return fail(c, "mpcgpu_calc_posteriors: bad pair range [%llu,%llu)", (u64)k0, (u64)k1);
	const int rc = stage_a(c, k1 - k0, c->h_pair_x.data() + k0, c->h_pair_y.data() + k0);
	c->shard_is_list = false;
	c->sh_k0 = k0; c->sh_k1 = k1;
	return rc;
}

int mpcgpu_shard_info(mpcgpu_ctx *c, uint64_t *bytes, void **dev_ptr)
{
	if (!c) return 1;
	if (!c->have_shard) return fail(c, "mpcgpu_shard_info: no shard (call mpcgpu_calc_posteriors)");
	if (bytes) *bytes = c->shard_bytes;
	if (dev_ptr) *dev_ptr = c->d_shard.p;
	return 0;
}

int mpcgpu_shard_entries(mpcgpu_ctx *c, uint64_t *entries)
{
	if (!c) return 1;
	if (!c->have_shard) return fail(c, "mpcgpu_shard_entries: no shard (call mpcgpu_calc_posteriors)");
	u64 sum = 0;
	for (u32 v : c->sh_nnz) sum += v;
	if (entries) *entries = sum;
	return 0;
}

int mpcgpu_shard_export(mpcgpu_ctx *c, void *dev_dst)
{
	if (!c) return 1;
	if (!c->have_shard) return fail(c, "mpcgpu_shard_export: no shard (call mpcgpu_calc_posteriors)");
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemcpyAsync(dev_dst, c->d_shard.p, c->shard_bytes, hipMemcpyDeviceToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_store_import_part(mpcgpu_ctx *c, uint32_t nshards, const uint64_t *k0, const uint64_t *k1, const uint64_t *bytes,
	const uint64_t *offsets, void *dev_all, uint64_t own_k0, uint64_t own_k1)
{
	if (!c) return 1;
	if (c->n == 0) return fail(c, "mpcgpu_store_import: call mpcgpu_set_seqs first");
	if (own_k0 > own_k1 || own_k1 > c->npairs) return fail(c, "mpcgpu_store_import_part: bad own range [%llu,%llu)", (u64)own_k0, (u64)own_k1);
	HIPCHK(c, hipSetDevice(c->device));
	c->have_store = false;
	c->tiles_k0 = c->tiles_k1 = ~0ull;
	const u32 n = c->n;
	// the forward planes of a row-block stage A may hold a large share of the device: give them back if the store (roughly four
	// times the packed records) could not be allocated next to them
	if (c->d_fm.cap > (8ull << 30)) {
		u64 packed_bytes = 0;
		for (u32 s = 0; s < nshards; ++s) packed_bytes += bytes[s];
		size_t free_now = 0, total_now = 0;
		HIPCHK(c, hipMemGetInfo(&free_now, &total_now));
		if ((u64)free_now < 5 * packed_bytes + (4ull << 30)) c->d_fm.release();
	}
	// ---- read shard headers, check coverage. The shards may lie anywhere in dev_all (offsets[s]; null: back to back in the order
	// given) and come in any order: sorted by their first position they must tile [0, pairs)
	std::vector<u32> by_k(nshards);
	for (u32 s = 0; s < nshards; ++s) by_k[s] = s;
	std::sort(by_k.begin(), by_k.end(), [&](u32 a, u32 b) { return k0[a] != k0[b] ? k0[a] < k0[b] : k1[a] < k1[b]; });
	std::vector<u64> at(nshards, 0);
	{
		u64 run = 0;
		for (u32 s = 0; s < nshards; ++s) { at[s] = offsets ? offsets[s] : run; run += bytes[s]; if (at[s] & 3) return fail(c, "mpcgpu_store_import: shard %u is not word-aligned", s); }
	}
	c->all_nnz.assign(c->npairs, 0);
	c->all_ea.assign(c->npairs, 0.0f);
	c->h_pbase.assign(c->npairs + 1, 0);
	c->h_vbase.assign(c->npairs + 1, 0);
	u64 expect = 0, end_max = 0;
	// all headers are read back with ONE wait (a pair-sharded run with stage A in pieces imports world x pieces shards)
	std::vector<u64> hoff(nshards + 1, 0);
	for (u32 s = 0; s < nshards; ++s) {
		if (k1[s] < k0[s] || k1[s] > c->npairs) return fail(c, "mpcgpu_store_import: shard %u is [%llu,%llu)", s, (u64)k0[s], (u64)k1[s]);
		const u64 hdr = shard_header_bytes(k1[s] - k0[s]);
		if (bytes[s] < hdr) return fail(c, "mpcgpu_store_import: shard %u too small", s);
		hoff[s + 1] = hoff[s] + hdr;
	}
	std::vector<u8> &h = c->v_hdr; // (kept: see mpcgpu_ctx — 8 bytes per pair)
	h.resize(hoff[nshards]);
	for (u32 s = 0; s < nshards; ++s)
		HIPCHK(c, hipMemcpyAsync(h.data() + hoff[s], (const u8 *)dev_all + at[s], hoff[s + 1] - hoff[s], hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	for (u32 q = 0; q < nshards; ++q) {
		const u32 s = by_k[q];
		if (k0[s] != expect)
			return fail(c, "mpcgpu_store_import: shards must tile [0,%llu) (shard %u is [%llu,%llu), expected one that starts at %llu)",
				(u64)c->npairs, s, (u64)k0[s], (u64)k1[s], (u64)expect);
		const u64 np = k1[s] - k0[s];
		const u64 hdr = hoff[s + 1] - hoff[s];
		const u8 *hs = h.data() + hoff[s];
		u64 h2[2];
		memcpy(h2, hs, 16);
		if (h2[0] != np || hdr + h2[1] * 4 != bytes[s])
			return fail(c, "mpcgpu_store_import: shard %u header mismatch (pairs %llu vs %llu, bytes %llu vs %llu)", s,
				(u64)h2[0], (u64)np, (u64)(hdr + h2[1] * 4), (u64)bytes[s]);
		if (np) {
			memcpy(&c->all_nnz[k0[s]], hs + 16, np * 4);
			memcpy(&c->all_ea[k0[s]], hs + 16 + np * 4, np * 4);
		}
		u64 w = (at[s] + hdr) / 4;
		for (u64 k = k0[s]; k < k1[s]; ++k) {
			c->h_pbase[k] = w;
			w += rec_words(c->len[c->h_pair_x[k]], c->len[c->h_pair_y[k]], c->all_nnz[k]);
		}
		if (w * 4 != at[s] + bytes[s]) return fail(c, "mpcgpu_store_import: shard %u record sizes do not add up", s);
		end_max = std::max(end_max, at[s] + bytes[s]);
		expect = k1[s];
	}
	if (expect != c->npairs) return fail(c, "mpcgpu_store_import: shards cover %llu of %llu pairs", (u64)expect, (u64)c->npairs);
	c->h_pbase[c->npairs] = end_max / 4; // (a sentinel only: a pair's record length follows from its lengths and its entry count)
	for (u64 k = 0; k < c->npairs; ++k) c->h_vbase[k + 1] = c->h_vbase[k] + c->all_nnz[k];
	c->total_entries = c->h_vbase[c->npairs];
	c->max_nnz = 0;
	for (u64 k = 0; k < c->npairs; ++k) c->max_nnz = std::max(c->max_nnz, c->all_nnz[k]);
	c->max_len = 0;
	for (u32 i = 0; i < n; ++i) c->max_len = std::max(c->max_len, c->len[i]);
	c->st_packed = (const u32 *)dev_all;
	HIPCHK(c, c->d_vnext.ensure(std::max<u64>(c->total_entries, 1) * 4));
	if (upload(c, c->d_pbase, c->h_pbase) || upload(c, c->d_vbase, c->h_vbase)) return 1;
	// ---- a PARTIAL store: this context relaxes the positions [own_k0, own_k1) only, so it needs the records (A, Z) of the
	// sequences A those pairs touch and of no other (a rank of a block-partitioned run: the sequences of its blocks — half of the
	// store at 8 ranks). Everything per pair (packed records, values) stays complete.
	c->own_k0 = own_k0; c->own_k1 = own_k1;
	c->packed_stale = false;
	c->partial = !(own_k0 == 0 && own_k1 == c->npairs);
	if (c->partial) {
		c->need.assign(n, 0);
		for (u64 k = own_k0; k < own_k1; ++k) { c->need[c->h_pair_x[k]] = 1; c->need[c->h_pair_y[k]] = 1; }
		u32 cnt = 0;
		for (u32 i = 0; i < n; ++i) cnt += c->need[i];
		if (cnt == n) c->partial = false; // (the rank's pairs touch every sequence: nothing to leave out)
		else if (upload(c, c->d_need, c->need)) return 1;
	}
	// ---- layout for relax: padded records + LDS-tiled kernel when a tile fits the LDS and the
	// records fit HBM; otherwise compact slabs + the gather kernel (MPCGPU_RELAX=gather forces it).
	// Both are device paths with identical results.
	c->have_pad = false;
	{
		// variable-size dense records + the LDS-tiled relax_var_kernel; a run beyond that layout's limits (sequences longer
		// than 4095, records that do not fit the CU's LDS two at a time, ...) gets compact CSR slabs + the gather kernel
		// (MPCGPU_RELAX=gather forces it). Both are device paths with identical results.
		const char *mode = getenv("MPCGPU_RELAX");
		if (!(mode && !strcmp(mode, "gather"))) {
			const int rc = build_var_store(c);
			if (rc == 1) return 1;
			if (rc == 0) { c->have_store = true; return 0; }
		}
	}
	return build_slab_store(c);
}

int mpcgpu_store_import(mpcgpu_ctx *c, uint32_t nshards, const uint64_t *k0, const uint64_t *k1,
	const uint64_t *bytes, void *dev_all)
{
	if (!c) return 1;
	return mpcgpu_store_import_part(c, nshards, k0, k1, bytes, nullptr, dev_all, 0, c->npairs);
}

int mpcgpu_store_complete(mpcgpu_ctx *c)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_store_complete: no store");
	HIPCHK(c, hipSetDevice(c->device));
	if (refresh_packed(c)) return 1;
	if (!c->partial) { c->own_k0 = 0; c->own_k1 = c->npairs; return 0; }
	// the packed records hold the current values (every commit writes them): the records of ALL sequences are built from them
	c->partial = false;
	c->own_k0 = 0; c->own_k1 = c->npairs;
	c->have_store = false;
	c->tiles_k0 = c->tiles_k1 = ~0ull;
	const int rc = build_var_store(c);
	if (rc == 1) return 1;
	if (rc == 0) { c->have_store = true; return 0; }
	return build_slab_store(c);
}

int mpcgpu_build_store(mpcgpu_ctx *c)
{
	if (!c) return 1;
	if (!c->have_shard || c->shard_is_list || c->sh_k0 != 0 || c->sh_k1 != c->npairs)
		return fail(c, "mpcgpu_build_store: needs this context's shard to cover all pairs "
			"(multi-GPU callers use mpcgpu_store_import)");
	const uint64_t k0 = 0, k1 = c->npairs, bytes = c->shard_bytes;
	return mpcgpu_store_import(c, 1, &k0, &k1, &bytes, c->d_shard.p);
}

int mpcgpu_values_info(mpcgpu_ctx *c, void **dev_ptr, uint64_t *total_count)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_values_info: no store");
	if (dev_ptr) *dev_ptr = c->d_vnext.p;
	if (total_count) *total_count = c->total_entries;
	return 0;
}

int mpcgpu_values_slice(mpcgpu_ctx *c, uint64_t k0, uint64_t k1, uint64_t *first, uint64_t *count)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_values_slice: no store");
	if (k0 > k1 || k1 > c->npairs) return fail(c, "mpcgpu_values_slice: bad pair range");
	if (first) *first = c->h_vbase[k0];
	if (count) *count = c->h_vbase[k1] - c->h_vbase[k0];
	return 0;
}

int mpcgpu_values_export(mpcgpu_ctx *c, uint64_t first, uint64_t count, void *dev_dst)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_values_export: no store");
	if (first + count > c->total_entries) return fail(c, "mpcgpu_values_export: range out of bounds");
	HIPCHK(c, hipSetDevice(c->device));
	if (count)
		HIPCHK(c, hipMemcpyAsync(dev_dst, c->d_vnext.as<float>() + first, count * 4, hipMemcpyDeviceToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_values_import(mpcgpu_ctx *c, uint64_t first, uint64_t count, const void *dev_src)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_values_import: no store");
	if (first + count > c->total_entries) return fail(c, "mpcgpu_values_import: range out of bounds");
	HIPCHK(c, hipSetDevice(c->device));
	if (count)
		HIPCHK(c, hipMemcpyAsync(c->d_vnext.as<float>() + first, dev_src, count * 4, hipMemcpyDeviceToDevice, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_cons_iter(mpcgpu_ctx *c, uint64_t k0, uint64_t k1)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_cons_iter: no store (call mpcgpu_build_store / mpcgpu_store_import)");
	if (k0 > k1 || k1 > c->npairs) return fail(c, "mpcgpu_cons_iter: bad pair range");
	if (c->partial && k0 != k1 && (k0 < c->own_k0 || k1 > c->own_k1))
		return fail(c, "mpcgpu_cons_iter: [%llu,%llu) is outside the range [%llu,%llu) this partial store was imported for (mpcgpu_store_import_part)",
			(u64)k0, (u64)k1, (u64)c->own_k0, (u64)c->own_k1);
	HIPCHK(c, hipSetDevice(c->device));
	const u64 cnt = c->h_vbase[k1] - c->h_vbase[k0];
	c->work_entry_z = cnt * c->n;
	if (cnt == 0) return 0;
	StoreParams sp;
	fill_store_params(c, sp);
	if (c->have_pad) {
		if (c->band_ok) {
			int r = relax_band(c, sp, k0, k1);
			if (r == 2 && c->win_ok) {
				// no band fits with the Y rows as windows (one wide row can be most of the LDS): the same tiles with the Y rows as block
				// lists, i.e. the two-list walk; the window records are dropped
				c->win_ok = false;
				c->d_win.release(); c->d_pos_w.release(); c->d_wv_off.release(); c->d_wsum.release(); c->d_wmaxc.release(); c->d_wrec_off.release();
				{ const size_t at = c->store_desc.find(" + window records"); if (at != std::string::npos) c->store_desc.erase(at); }
				c->btiles_k0 = c->btiles_k1 = ~0ull;
				fill_store_params(c, sp);
				r = relax_band(c, sp, k0, k1);
			}
			if (r != 2) return r;
			c->band_ok = false; // this store's rows do not cut into band tiles that fit: whole-record tiles from here on
		}
		if (c->var_pairs_ok) return relax_var(c, sp, k0, k1);
		if (build_slab_store(c)) return 1; // neither: CSR slabs + the gather kernel (same results)
		fill_store_params(c, sp);
	}
	TimedSpan ts;
	if (span_begin(c, 3, &ts)) return 1;
	const u32 block = 256;
	const u64 blocks = (cnt + block - 1) / block;
	MPC_LAUNCH(relax_kernel, (u32)std::min<u64>(blocks, (u64)c->prop.multiProcessorCount * 64), block, 0, c->stream, sp,
		(u64)k0, (u64)k1);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &ts)) return 1;
	return 0;
}

int mpcgpu_cons_commit_range(mpcgpu_ctx *c, uint64_t first, uint64_t count)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_cons_commit: no store");
	if (first + count > c->total_entries) return fail(c, "mpcgpu_cons_commit_range: range out of bounds");
	HIPCHK(c, hipSetDevice(c->device));
	if (count == 0) return 0;
	StoreParams sp;
	fill_store_params(c, sp);
	TimedSpan ts;
	if (span_begin(c, 4, &ts)) return 1;
	const u32 block = 256;
	const u64 blocks = (count + block - 1) / block;
	const u32 grid = (u32)std::min<u64>(blocks, (u64)c->prop.multiProcessorCount * 64);
	if (c->have_pad && env_int("MPCGPU_COMMIT", 1) != 0) {
		// one wave per pair (kernels_store.h: commit_pairs_kernel): the pairs whose entries meet [first, first + count)
		const u64 ka = (u64)(std::upper_bound(c->h_vbase.begin(), c->h_vbase.end(), (u64)first) - c->h_vbase.begin()) - 1;
		const u64 kb = (u64)(std::lower_bound(c->h_vbase.begin(), c->h_vbase.end(), (u64)(first + count)) - c->h_vbase.begin());
		// a context that relaxes a part of the pairs only (a rank of a sharded run) writes the packed records of ITS pairs — what its
		// relax reads — and leaves the others' to refresh_packed (whoever reads them asks for it)
		const bool lazy = !(c->own_k0 == 0 && c->own_k1 == c->npairs);
		if (lazy) c->packed_stale = true;
		const u64 pairs = std::min<u64>(kb, c->npairs) - ka;
		MPC_LAUNCH(commit_pairs_kernel, (u32)std::min<u64>(std::max<u64>(pairs, 1), (u64)c->prop.multiProcessorCount * 64), 64, 0, c->stream, sp, ka,
			std::min<u64>(kb, c->npairs), (u64)first, (u64)(first + count), c->own_k0, c->own_k1, lazy ? 1 : 0);
	}
	else if (c->have_pad) MPC_LAUNCH(commit_pad_kernel, grid, block, 0, c->stream, sp, (u64)first, (u64)(first + count));
	else MPC_LAUNCH(commit_kernel, grid, block, 0, c->stream, sp, (u64)first, (u64)(first + count));
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &ts)) return 1;
	return 0;
}

int mpcgpu_cons_commit(mpcgpu_ctx *c)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_cons_commit: no store");
	return mpcgpu_cons_commit_range(c, 0, c->total_entries);
}

// The getters speak MPCFlat::InitPairs pair numbers (mpcflat.cpp:145-155) whatever order the context keeps its pairs in
// (mpcgpu_set_pair_order): position of InitPairs pair k
static inline u64 pos_of(const mpcgpu_ctx *c, u64 k) { return c->ext2pos.empty() ? k : (u64)c->ext2pos[k]; }

int mpcgpu_get_ea(mpcgpu_ctx *c, uint64_t k0, uint64_t k1, float *ea)
{
	if (!c) return 1;
	if (k0 > k1 || k1 > c->npairs) return fail(c, "mpcgpu_get_ea: bad pair range");
	const bool shard_ok = c->have_shard && !c->shard_is_list;
	if (c->ext2pos.empty()) {
		if (c->have_store) { memcpy(ea, &c->all_ea[k0], (k1 - k0) * 4); return 0; }
		if (shard_ok && k0 >= c->sh_k0 && k1 <= c->sh_k1) { memcpy(ea, &c->sh_ea[k0 - c->sh_k0], (k1 - k0) * 4); return 0; }
		return fail(c, "mpcgpu_get_ea: range [%llu,%llu) not available", (u64)k0, (u64)k1);
	}
	for (u64 k = k0; k < k1; ++k) {
		const u64 q = pos_of(c, k);
		if (c->have_store) ea[k - k0] = c->all_ea[q];
		else if (shard_ok && q >= c->sh_k0 && q < c->sh_k1) ea[k - k0] = c->sh_ea[q - c->sh_k0];
		else return fail(c, "mpcgpu_get_ea: pair %llu not available", (u64)k);
	}
	return 0;
}

int mpcgpu_get_nnz(mpcgpu_ctx *c, uint64_t k0, uint64_t k1, uint32_t *nnz)
{
	if (!c) return 1;
	if (k0 > k1 || k1 > c->npairs) return fail(c, "mpcgpu_get_nnz: bad pair range");
	const bool shard_ok = c->have_shard && !c->shard_is_list;
	if (c->ext2pos.empty()) {
		if (c->have_store) { memcpy(nnz, &c->all_nnz[k0], (k1 - k0) * 4); return 0; }
		if (shard_ok && k0 >= c->sh_k0 && k1 <= c->sh_k1) { memcpy(nnz, &c->sh_nnz[k0 - c->sh_k0], (k1 - k0) * 4); return 0; }
		return fail(c, "mpcgpu_get_nnz: range [%llu,%llu) not available", (u64)k0, (u64)k1);
	}
	for (u64 k = k0; k < k1; ++k) {
		const u64 q = pos_of(c, k);
		if (c->have_store) nnz[k - k0] = c->all_nnz[q];
		else if (shard_ok && q >= c->sh_k0 && q < c->sh_k1) nnz[k - k0] = c->sh_nnz[q - c->sh_k0];
		else return fail(c, "mpcgpu_get_nnz: pair %llu not available", (u64)k);
	}
	return 0;
}

int mpcgpu_get_sparse_range(mpcgpu_ctx *c, uint64_t k0, uint64_t k1, uint32_t *offsets, void *values)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_get_sparse_range: no store (call mpcgpu_build_store / mpcgpu_store_import)");
	if (k0 > k1 || k1 > c->npairs) return fail(c, "mpcgpu_get_sparse_range: bad pair range");
	if (k0 == k1) return 0;
	HIPCHK(c, hipSetDevice(c->device));
	if (refresh_packed(c)) return 1;
	const bool listed = !c->ext2pos.empty(); // a custom pair order: the InitPairs range is a list of positions
	std::vector<u64> offbase(k1 - k0), valbase, klist;
	if (listed) { valbase.resize(k1 - k0); klist.resize(k1 - k0); }
	u64 o = 0, nval = 0;
	for (u64 k = k0; k < k1; ++k) {
		const u64 q = pos_of(c, k);
		offbase[k - k0] = o; o += c->len[c->h_pair_x[q]] + 1;
		if (listed) { klist[k - k0] = q; valbase[k - k0] = nval; }
		nval += c->all_nnz[q];
	}
	HIPCHK(c, c->d_exp_off.ensure(o * 4));
	HIPCHK(c, c->d_exp_val.ensure(std::max<u64>(nval, 1) * 8));
	if (upload(c, c->d_exp_offbase, offbase)) return 1;
	if (listed && (upload(c, c->d_exp_klist, klist) || upload(c, c->d_exp_valbase, valbase))) return 1;
	StoreParams sp;
	fill_store_params(c, sp);
	MPC_LAUNCH(export_kernel, (u32)std::min<u64>(k1 - k0, (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, sp,
		(u64)(listed ? 0 : k0), (u64)(listed ? k1 - k0 : k1), c->d_exp_offbase.as<u64>(), c->d_exp_off.as<u32>(), c->d_exp_val.as<u32>(),
		listed ? c->d_exp_klist.as<u64>() : (const u64 *)nullptr, listed ? c->d_exp_valbase.as<u64>() : (const u64 *)nullptr);
	HIPCHK(c, hipGetLastError());
	HIPCHK(c, hipMemcpyAsync(offsets, c->d_exp_off.p, o * 4, hipMemcpyDeviceToHost, c->stream));
	if (nval) HIPCHK(c, hipMemcpyAsync(values, c->d_exp_val.p, nval * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_get_sparse(mpcgpu_ctx *c, uint64_t k, uint32_t *offsets, void *values)
{
	return mpcgpu_get_sparse_range(c, k, k + 1, offsets, values);
}

// CalcAlnFlat + TraceBackFlat on a dense LX x LY matrix already in device memory
static int run_calc_aln(mpcgpu_ctx *c, const float *d_post, uint32_t LX, uint32_t LY, char *path, uint32_t *pathlen, float *score)
{
	const u64 W = (u64)LY + 1;
	// one wavefront with the previous row in registers and the traceback codes in LDS when the matrix is small enough
	// (the progressive joins and refinement rounds of L~400 families are), else the workgroup kernel
	const size_t smem_wave = (size_t)(LX + 1) * MPC_ALNW_ROWBYTES + 16;
	const int pick = env_int("MPCGPU_ALN_KERNEL", 0); // 0 = by size, 1 = one wave, 2 = several waves, 3 = LDS rows
	const bool wave = W <= MPC_ALNW_MAXW && smem_wave <= 160u * 1024u && (pick == 0 || pick == 1);
	// several waves, 4 columns per thread, previous row in registers: up to 4096 columns
	const u32 qthreads = (u32)((W + 4 * 64 - 1) / (4 * 64)) * 64;
	const bool quad = !wave && qthreads <= 1024 && (pick == 0 || pick == 2);
	const u32 qrows = quad ? (u32)std::min<u64>((u64)LX + 1, (150u * 1024u) / qthreads) : 0;
	const size_t smem = wave ? smem_wave : quad ? (size_t)MPC_ALNQ_HDR + (size_t)qrows * qthreads : (size_t)(2 * W + MPC_ALN_THREADS / 64 + 4) * 4;
	if (smem > 160u * 1024u)
		return fail(c, "mpcgpu_calc_aln: %u columns exceed the LDS-resident DP rows of this build", LY);
	HIPCHK(c, c->d_aln_tb.ensure_grow(((u64)LX + 1) * (quad ? (u64)qthreads : W))); // letters per cell, or one byte per thread and row
	HIPCHK(c, c->d_aln_rev.ensure_grow((u64)LX + LY));
	// one result record {path length, score, path}: one copy back, one wait
	const u64 res_bytes = 8 + (u64)LX + LY;
	HIPCHK(c, c->h_aln_res.ensure(res_bytes));
	AlnParams ap;
	ap.post = d_post; ap.LX = LX; ap.LY = LY;
	ap.tb = c->d_aln_tb.as<char>(); ap.rev = c->d_aln_rev.as<char>();
	// the record is written straight into page-locked host memory (device-visible: hipHostMalloc): no copy back, one wait
	ap.pathlen = c->h_aln_res.as<u32>(); ap.score = c->h_aln_res.as<float>() + 1; ap.path = c->h_aln_res.as<char>() + 8;
	const int which = wave ? 0 : quad ? 1 : 2;
	if (smem > c->aln_smem_set[which]) { // raise the kernel's dynamic-LDS limit only when this call needs more than any before
		(void)hipFuncSetAttribute(wave ? (const void *)calc_aln_wave_kernel : quad ? (const void *)calc_aln_quad_kernel : (const void *)calc_aln_kernel,
			hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		c->aln_smem_set[which] = smem;
	}
	if (trace_on()) { fprintf(stderr, "[mpcgpu] calc_aln %u x %u: %s\n", LX, LY, wave ? "one wave" : quad ? "waves, rows in registers" : "rows in LDS"); fflush(stderr); }
	TimedSpan ts_aln;
	if (span_begin(c, 8, &ts_aln)) return 1;
	if (wave) MPC_LAUNCH(calc_aln_wave_kernel, 1, 64, smem, c->stream, ap);
	else if (quad) MPC_LAUNCH(calc_aln_quad_kernel, 1, qthreads, smem, c->stream, ap, qrows);
	else MPC_LAUNCH(calc_aln_kernel, 1, MPC_ALN_THREADS, smem, c->stream, ap);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &ts_aln)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream));
	const u32 n_path = c->h_aln_res.as<u32>()[0];
	if (n_path > LX + LY) return fail(c, "mpcgpu_calc_aln: path length %u out of range (internal error)", n_path);
	*pathlen = n_path;
	if (score) memcpy(score, c->h_aln_res.as<char>() + 4, 4);
	memcpy(path, c->h_aln_res.as<char>() + 8, n_path);
	return 0;
}

// The finishing kernels (kernels_post.h) on ONE caller-supplied list of cells with Score >= MIN_SPARSE_SCORE — what fb_kernel
// would have emitted for a pair. Lets tests reach shapes of the candidate list the pair-HMM never produces (rows whose first
// cell lies beyond the EA frontier, empty rows, one-column matrices) and compare both kernels with the dense
// CalcAlnScoreFlat (calcalnscoreflat.cpp:4-32) / MySparseMx::FromPost (mysparsemx.cpp:115-152).
int mpcgpu_post_scores(mpcgpu_ctx *c, uint32_t LX, uint32_t LY, uint32_t ncand, const uint32_t *rows, const uint32_t *cols,
	const float *scores, int kernel, uint32_t batch, float *ea, uint32_t *nnz, uint32_t *offsets, void *values)
{
	if (!c) return 1;
	if (!c->have_hmm) return fail(c, "mpcgpu_post_scores: set_hmm first (expf variant)");
	if (LX == 0 || LY == 0 || LX > MPC_KEY_COL_MASK_LONG || LY > MPC_KEY_COL_MASK_LONG) return fail(c, "mpcgpu_post_scores: bad shape %u x %u", LX, LY);
	HIPCHK(c, hipSetDevice(c->device));
	const u32 capc = std::max<u32>(ncand, 1);
	std::vector<u64> cand(capc, 0);
	const u32 long_min = LX > 1023u ? LX : 0xffffffffu; // 22-bit column keys unless the rows need more than 10 bits
	const u32 kshift = LX >= long_min ? MPC_KEY_ROW_SHIFT_LONG : MPC_KEY_ROW_SHIFT;
	for (u32 q = 0; q < ncand; ++q) {
		if (rows[q] >= LX || cols[q] >= LY) return fail(c, "mpcgpu_post_scores: cell %u out of range", q);
		u32 bits;
		memcpy(&bits, &scores[q], 4);
		cand[q] = ((u64)((rows[q] << kshift) | cols[q]) << 32) | bits;
	}
	const std::vector<u32> lens = {LX, LY}, zero = {0}, one = {1}, cnt = {ncand};
	DevBuf d_len, d_x, d_y, d_cnt, d_cand, d_res, d_out, d_sort, d_srow;
	auto free_all = [&]() { for (DevBuf *b : {&d_len, &d_x, &d_y, &d_cnt, &d_cand, &d_res, &d_out, &d_sort, &d_srow}) b->release(); };
	const u64 res_stride = (u64)LX + LY + 4 * (u64)capc;
	int rc = 0;
	do {
		if (upload(c, d_len, lens) || upload(c, d_x, zero) || upload(c, d_y, one) || upload(c, d_cnt, cnt) || upload(c, d_cand, cand)) { rc = 1; break; }
		if (d_res.ensure(res_stride * 4) != hipSuccess || d_out.ensure(16) != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: out of device memory"); break; }
		if (kernel == 0) {
			PostRowsParams pr;
			pr.pair_x = d_x.as<u32>(); pr.pair_y = d_y.as<u32>(); pr.seq_len = d_len.as<u32>();
			pr.cand = d_cand.as<u64>(); pr.capc = capc; pr.cand_cnt = d_cnt.as<u32>();
			pr.use_fma = c->use_fma;
			pr.lx_cap = LX + 2; pr.ly_cap = LY + 2;
			pr.sort_cap = std::min<u32>(capc, 1024u); pr.sort_stride = capc;
			pr.batch = std::min<u32>(std::max<u32>(batch, 1u), 64u);
			const size_t fixed_lds = ((((size_t)pr.lx_cap + 2 * (size_t)pr.ly_cap) * 4 + 7) & ~(size_t)7);
			if (fixed_lds + 16 > 150 * 1024) { rc = fail(c, "mpcgpu_post_scores: %u x %u does not fit the row-list kernel", LX, LY); break; }
			if (fixed_lds + (size_t)pr.sort_cap * 8 > 150 * 1024) pr.sort_cap = (u32)((150 * 1024 - fixed_lds) / 8);
			const size_t smem = fixed_lds + (size_t)pr.sort_cap * 8;
			if (smem > 64 * 1024 && hipFuncSetAttribute((const void *)post_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: LDS"); break; }
			if (d_sort.ensure((u64)capc * 8 + 8) != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: out of device memory"); break; }
			pr.sort_scratch = d_sort.as<u64>();
			pr.res = d_res.as<u32>(); pr.res_stride = res_stride;
			pr.nnz = d_out.as<u32>(); pr.ea = d_out.as<float>() + 1; pr.flags = d_out.as<u32>() + 2;
			pr.count = 1; pr.long_min = long_min; pr.prof = nullptr;
			MPC_LAUNCH(post_rows_kernel, 1, 64, smem, c->stream, pr);
		} else {
			PostParams pp;
			pp.pair_x = d_x.as<u32>(); pp.pair_y = d_y.as<u32>(); pp.seq_len = d_len.as<u32>();
			pp.cand = d_cand.as<u64>(); pp.capc = capc; pp.cand_cnt = d_cnt.as<u32>();
			pp.use_fma = c->use_fma;
			pp.sort_cap = std::min<u32>(next_pow2(std::max<u32>(capc, 2)), 1024u);
			pp.srow_cap = std::min<u32>(LY + 1, 2048u);
			pp.sort_stride = next_pow2(std::max<u32>(capc, 2));
			pp.srow_stride = 2 * ((u64)LY + 1);
			if (d_sort.ensure(pp.sort_stride * 8) != hipSuccess || d_srow.ensure(pp.srow_stride * 4) != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: out of device memory"); break; }
			pp.sort_scratch = d_sort.as<u64>(); pp.srow_scratch = d_srow.as<float>();
			pp.res = d_res.as<u32>(); pp.res_stride = res_stride;
			pp.nnz = d_out.as<u32>(); pp.ea = d_out.as<float>() + 1; pp.flags = d_out.as<u32>() + 2;
			pp.count = 1; pp.long_min = long_min;
			const size_t psmem = (size_t)pp.sort_cap * 8 + (size_t)pp.srow_cap * 2 * 4;
			MPC_LAUNCH(post_kernel, 1, 64, psmem, c->stream, pp);
		}
		if (hipGetLastError() != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: launch failed"); break; }
		u32 out[4] = {0, 0, 0, 0};
		if (hipMemcpyAsync(out, d_out.p, 12, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: copy failed"); break; }
		if (out[2] & 1u) { rc = fail(c, "mpcgpu_post_scores: candidate overflow"); break; }
		if (nnz) *nnz = out[0];
		if (ea) memcpy(ea, &out[1], 4);
		if (offsets && values) { // MySparseMx layout: offsets[LX+1], {P, col} per entry
			std::vector<u32> rec(res_stride);
			if (hipMemcpyAsync(rec.data(), d_res.p, res_stride * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(c, "mpcgpu_post_scores: copy failed"); break; }
			u32 acc = 0;
			for (u32 i = 0; i < LX; ++i) { offsets[i] = acc; acc += rec[i]; }
			offsets[LX] = acc;
			memcpy(values, rec.data() + LX + LY, (size_t)out[0] * 8);
		}
	} while (0);
	free_all();
	return rc;
}

int mpcgpu_calc_aln(mpcgpu_ctx *c, const float *post, uint32_t LX, uint32_t LY, char *path, uint32_t *pathlen, float *score)
{
	if (!c) return 1;
	if (!post || !path || !pathlen) return fail(c, "mpcgpu_calc_aln: NULL argument");
	if (LX == 0 || LY == 0) return fail(c, "mpcgpu_calc_aln: empty matrix (%u x %u)", LX, LY);
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, c->d_aln_post.ensure((u64)LX * LY * 4));
	HIPCHK(c, hipMemcpyAsync(c->d_aln_post.p, post, (u64)LX * LY * 4, hipMemcpyHostToDevice, c->stream));
	c->last_post_cells = (u64)LX * LY;
	return run_calc_aln(c, c->d_aln_post.as<float>(), LX, LY, path, pathlen, score); // syncs: the caller's buffer is done with
}

static u32 bits_for(u64 v) { u32 b = 1; while ((v >> b) != 0) ++b; return b; } // bits to hold values 0..v

int mpcgpu_align_alns(mpcgpu_ctx *c, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2, uint32_t C1,
	uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2, char *path, uint32_t *pathlen, float *score)
{
	return mpcgpu_align_alns_w(c, n1, seq1, n2, seq2, C1, C2, pos2col1, pos2col2, nullptr, nullptr, path, pathlen, score);
}

// Buffers of the in-order reduction (kernels_prog.h): end of every cell's run, the list of runs, {runs, next run}. The
// generating kernel zeroes the output matrix and the two counters.
struct RunBufs { u32 *run_end, *heads, *counters; };
static int prepare_runs(mpcgpu_ctx *c, u64 M, u64 cells, RunBufs *rb)
{
	if (cells > 0xffffffffull) return fail(c, "mpcgpu_align_alns: %llu cells exceed this build's cell index", (u64)cells);
	const u64 maxruns = std::min<u64>(M, cells);
	HIPCHK(c, c->d_bp_runs.ensure_grow((cells + 2 * maxruns + 2) * 4));
	rb->run_end = c->d_bp_runs.as<u32>(); rb->heads = rb->run_end + cells; rb->counters = rb->heads + 2 * maxruns;
	return 0;
}
// post[cell] = the cell's records added in key order, 0 where there are none: list the runs of the sorted records, one wave per run.
static int reduce_runs(mpcgpu_ctx *c, const RunBufs &rb, const u32 *keys_sorted, const float *vals_sorted, u64 M, u64 cells)
{
	if (!M) return 0;
	const u64 maxruns = std::min<u64>(M, cells);
	const u32 grid_cap = (u32)c->prop.multiProcessorCount * 8;
	MPC_LAUNCH(build_post_heads_kernel, (u32)std::min<u64>((M + 255) / 256, grid_cap), 256, 0, c->stream, keys_sorted, (u64)M, rb.run_end, rb.heads,
		rb.counters);
	HIPCHK(c, hipGetLastError());
	// two waves per SIMD pulling runs from a queue (kernels_prog.h); MPCGPU_BP_WAVES: resident waves per SIMD
	const int bp_waves = std::min(std::max(env_int("MPCGPU_BP_WAVES", 2), 1), 8);
	const u32 red_grid = (u32)c->prop.multiProcessorCount * (u32)bp_waves;
	MPC_LAUNCH(build_post_reduce_kernel, (u32)std::min<u64>((maxruns + 3) / 4, red_grid), 256, 0, c->stream, vals_sorted, (const u32 *)rb.run_end,
		(const u32 *)rb.heads, (const u32 *)rb.counters, rb.counters + 1, c->d_aln_post.as<float>());
	HIPCHK(c, hipGetLastError());
	return 0;
}

// BuildPost on the device store (+ CalcAlnFlat when path != NULL): the body of mpcgpu_align_alns_w and mpcgpu_build_post
static int build_post_impl(mpcgpu_ctx *c, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2, uint32_t C1,
	uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2, const float *w1, const float *w2, char *path,
	uint32_t *pathlen, float *score)
{
	if (!c) return 1;
	if (!c->have_store) return fail(c, "mpcgpu_align_alns: no store (call mpcgpu_build_store / mpcgpu_store_import)");
	// BuildPost reads the records of any sequence: a partial store (a rank of a block-partitioned run) is completed first — once,
	// from the packed records, which hold the current values
	if ((c->partial || c->packed_stale) && mpcgpu_store_complete(c)) return 1;
	if (!seq1 || !seq2 || !pos2col1 || !pos2col2 || (path && !pathlen)) return fail(c, "mpcgpu_align_alns: NULL argument");
	if (n1 == 0 || n2 == 0 || C1 == 0 || C2 == 0) return fail(c, "mpcgpu_align_alns: empty alignment");
	HIPCHK(c, hipSetDevice(c->device));
	const u32 n = c->n;
	static const bool host_trace = env_int("MPCGPU_TRACE_HOST", 0) != 0; // diagnostics: host wall time of this call's phases, summed
	double *acc_t = c->aa_trace_t; // per context: the shrub workers of -super7 call this concurrently on their own contexts
	u64 &acc_n = c->aa_trace_n;
	auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	double t_prev = host_trace ? now() : 0.0;
	auto lap = [&](int k) { if (host_trace) { const double t = now(); acc_t[k] += t - t_prev; t_prev = t; } };
	// ---- host: everything the kernels need from this call in ONE page-locked record, one copy:
	// [off: n1+n2+1 u64][coff: n1*n2+1 u64][seqs: n1+n2 u32][maps: len1+len2 u32][weights: n1+n2 f32, when not all 1.0f]
	u64 len1 = 0, len2 = 0;
	for (u32 a = 0; a < n1; ++a) {
		if (seq1[a] >= n) return fail(c, "mpcgpu_align_alns: sequence index %u out of range", seq1[a]);
		len1 += c->len[seq1[a]];
	}
	for (u32 b = 0; b < n2; ++b) {
		if (seq2[b] >= n) return fail(c, "mpcgpu_align_alns: sequence index %u out of range", seq2[b]);
		len2 += c->len[seq2[b]];
	}
	if ((w1 != nullptr) != (w2 != nullptr)) return fail(c, "mpcgpu_align_alns_w: give both weight arrays or neither");
	bool weighted = false; // all 1.0f (what MPCFlat::Run sets): (1*1)*P == P, skip the multiply
	if (w1) {
		for (u32 a = 0; a < n1; ++a) weighted = weighted || w1[a] != 1.0f;
		for (u32 b = 0; b < n2; ++b) weighted = weighted || w2[b] != 1.0f;
	}
	const u64 npairs12 = (u64)n1 * n2;
	// ---- small joins: the whole matrix in one launch, inputs read from page-locked host memory (kernels_prog.h:
	// build_post_rows_kernel). Needs the variable-size record store (every ordered pair by row).
	{
		const char *bp_mode = getenv("MPCGPU_BP"); // "sort": always the general path; "rows": the row kernel whenever its limits allow
		const bool want_rows = !(bp_mode && !strcmp(bp_mode, "sort"));
		const u64 pair_limit = (bp_mode && !strcmp(bp_mode, "rows")) ? ~0ull : (u64)std::max(env_int("MPCGPU_BP_ROWS_PAIRS", 2048), 1);
		if (want_rows && c->have_pad && npairs12 <= pair_limit && C2 <= 1024u && (u64)n1 * C1 <= (1u << 26)) {
			const u64 r_seqs = 0, r_c2p = r_seqs + 4 * ((u64)n1 + n2), r_off2 = r_c2p + 4 * (u64)n1 * C1, r_maps = r_off2 + 4 * ((u64)n2 + 1),
				r_w = r_maps + 4 * len2, r_err = r_w + (weighted ? 4 * ((u64)n1 + n2) : 0), r_bytes = r_err + 4;
			HIPCHK(c, c->h_bp_in.ensure(r_bytes));
			char *hin = c->h_bp_in.as<char>();
			u32 *seqs = (u32 *)(hin + r_seqs), *c2p = (u32 *)(hin + r_c2p), *off2 = (u32 *)(hin + r_off2), *maps2 = (u32 *)(hin + r_maps);
			memcpy(seqs, seq1, 4 * (size_t)n1);
			memcpy(seqs + n1, seq2, 4 * (size_t)n2);
			for (u64 q = 0; q < (u64)n1 * C1; ++q) c2p[q] = MPC_BPR_GAP;
			u64 at = 0;
			for (u32 a = 0; a < n1; ++a) {
				const u32 L = c->len[seq1[a]];
				for (u32 pos = 0; pos < L; ++pos) {
					const u32 col = pos2col1[at + pos];
					if (col >= C1) return fail(c, "mpcgpu_align_alns: column map of MSA1 out of range");
					c2p[(u64)a * C1 + col] = pos;
				}
				at += L;
			}
			off2[0] = 0;
			for (u32 b = 0; b < n2; ++b) off2[b + 1] = off2[b] + c->len[seq2[b]];
			memcpy(maps2, pos2col2, 4 * len2);
			for (u64 q = 0; q < len2; ++q) if (maps2[q] >= C2) return fail(c, "mpcgpu_align_alns: column map of MSA2 out of range");
			for (u32 a = 0; a < n1; ++a)
				for (u32 b = 0; b < n2; ++b) if (seq1[a] == seq2[b]) return fail(c, "mpcgpu_align_alns: sequence %u is in both alignments", seq1[a]);
			if (weighted) {
				float *w = (float *)(hin + r_w);
				memcpy(w, w1, 4 * (size_t)n1);
				memcpy(w + n1, w2, 4 * (size_t)n2);
			}
			*(u32 *)(hin + r_err) = 0u;
			lap(0);
			const u64 cells = (u64)C1 * C2;
			HIPCHK(c, c->d_aln_post.ensure_grow(cells * 4));
			BuildPostRowsParams rp;
			fill_store_params(c, rp.s);
			rp.seq1 = seqs; rp.seq2 = seqs + n1; rp.n1 = n1; rp.n2 = n2;
			rp.c2p1 = c2p; rp.p2c2 = maps2; rp.off2 = off2; rp.C1 = C1; rp.C2 = C2;
			rp.w1 = weighted ? (const float *)(hin + r_w) : nullptr; rp.w2 = weighted ? rp.w1 + n1 : nullptr;
			rp.post = c->d_aln_post.as<float>(); rp.err = (u32 *)(hin + r_err);
			const u32 grid = std::min<u32>(C1, (u32)c->prop.multiProcessorCount * 8u);
			if (trace_on()) { fprintf(stderr, "[mpcgpu] build_post %u x %u rows, %u x %u columns: row kernel, grid %u\n", n1, n2, C1, C2, grid); fflush(stderr); }
			TimedSpan ts_rows;
			if (span_begin(c, 5, &ts_rows)) return 1;
			if (C2 <= 512u) MPC_LAUNCH(build_post_rows_kernel<8>, grid, 64, 8 * MPC_BPR_CAP, c->stream, rp);
			else MPC_LAUNCH(build_post_rows_kernel<16>, grid, 64, 8 * MPC_BPR_CAP, c->stream, rp);
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &ts_rows)) return 1;
			lap(2);
			c->last_post_cells = cells;
			int rc_rows = 0;
			if (!path) HIPCHK(c, hipStreamSynchronize(c->stream));
			else rc_rows = run_calc_aln(c, c->d_aln_post.as<float>(), C1, C2, path, pathlen, score); // ends with a wait for the stream
			lap(3);
			if (*(volatile u32 *)(hin + r_err) == 0u) return rc_rows;
			// a chunk of pairs overflowed the row kernel's list (very wide posterior rows): the general path below redoes the join
		}
	}
	const u64 o_off = 0, o_coff = o_off + 8 * ((u64)n1 + n2 + 1), o_seqs = o_coff + 8 * (npairs12 + 1), o_maps = o_seqs + 4 * ((u64)n1 + n2),
		o_w = o_maps + 4 * (len1 + len2), in_bytes = o_w + (weighted ? 4 * ((u64)n1 + n2) : 0);
	HIPCHK(c, c->h_bp_in.ensure(in_bytes));
	char *hin = c->h_bp_in.as<char>();
	u64 *off = (u64 *)(hin + o_off), *coff = (u64 *)(hin + o_coff);
	u32 *seqs = (u32 *)(hin + o_seqs), *maps = (u32 *)(hin + o_maps);
	memcpy(seqs, seq1, 4 * (size_t)n1);
	memcpy(seqs + n1, seq2, 4 * (size_t)n2);
	off[0] = 0;
	for (u32 a = 0; a < n1 + n2; ++a) off[a + 1] = off[a] + c->len[seqs[a]];
	memcpy(maps, pos2col1, len1 * 4);
	memcpy(maps + len1, pos2col2, len2 * 4);
	for (u64 q = 0; q < len1; ++q) if (maps[q] >= C1) return fail(c, "mpcgpu_align_alns: column map of MSA1 out of range");
	for (u64 q = len1; q < len1 + len2; ++q) if (maps[q] >= C2) return fail(c, "mpcgpu_align_alns: column map of MSA2 out of range");
	coff[0] = 0;
	for (u32 a = 0; a < n1; ++a)
		for (u32 b = 0; b < n2; ++b) {
			const u32 S = seqs[a], T = seqs[n1 + b];
			if (S == T) return fail(c, "mpcgpu_align_alns: sequence %u is in both alignments", S);
			const u64 k = S < T ? (u64)S * n - ((u64)S * (S + 1)) / 2 + (T - S - 1) : (u64)T * n - ((u64)T * (T + 1)) / 2 + (S - T - 1);
			coff[(u64)a * n2 + b + 1] = coff[(u64)a * n2 + b] + c->all_nnz[k];
		}
	if (weighted) {
		float *w = (float *)(hin + o_w);
		memcpy(w, w1, 4 * (size_t)n1);
		memcpy(w + n1, w2, 4 * (size_t)n2);
	}
	const u64 M = coff[npairs12];
	const u64 cells = (u64)C1 * C2;
	if (cells > 0xffffffffull) return fail(c, "mpcgpu_align_alns: %llu cells exceed this build's cell index", (u64)cells);
	const u32 bc = bits_for(cells - 1);
	if (M > 0xffffffffull) return fail(c, "mpcgpu_align_alns: %llu contributions exceed this build's record count", (u64)M);
	lap(0);
	HIPCHK(c, c->d_bp_in.ensure_grow(in_bytes));
	HIPCHK(c, hipMemcpyAsync(c->d_bp_in.p, hin, in_bytes, hipMemcpyHostToDevice, c->stream));
	lap(1);
	HIPCHK(c, c->d_bp_keys.ensure_grow(std::max<u64>(M, 1) * 4 * 2));
	HIPCHK(c, c->d_bp_vals.ensure_grow(std::max<u64>(M, 1) * 4 * 2));
	HIPCHK(c, c->d_aln_post.ensure_grow(cells * 4));
	RunBufs rb;
	if (prepare_runs(c, M, cells, &rb)) return 1;
	u32 *keys_in = c->d_bp_keys.as<u32>(), *keys_out = keys_in + std::max<u64>(M, 1);
	float *vals_in = c->d_bp_vals.as<float>(), *vals_out = vals_in + std::max<u64>(M, 1);
	const char *din = c->d_bp_in.as<char>();
	BuildPostParams bp;
	fill_store_params(c, bp.s);
	bp.seq1 = (const u32 *)(din + o_seqs); bp.seq2 = bp.seq1 + n1; bp.n1 = n1; bp.n2 = n2;
	bp.p2c1 = (const u32 *)(din + o_maps); bp.p2c2 = bp.p2c1; // offsets below are into the one concatenated array
	bp.p2c1_off = (const u64 *)(din + o_off); bp.p2c2_off = bp.p2c1_off + n1;
	bp.C2 = C2; bp.coff = (const u64 *)(din + o_coff); bp.keys = keys_in; bp.vals = vals_in;
	bp.w1 = weighted ? (const float *)(din + o_w) : nullptr; bp.w2 = weighted ? bp.w1 + n1 : nullptr;
	bp.post = c->d_aln_post.as<float>(); bp.cells = cells; bp.counters = rb.counters;
	TimedSpan ts_bp;
	if (span_begin(c, 5, &ts_bp)) return 1;
	MPC_LAUNCH(build_post_gen_kernel, (u32)std::min<u64>(npairs12, (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, bp);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &ts_bp)) return 1;
	const u32 *keys_sorted = keys_in;
	const float *vals_sorted = vals_in;
	if (span_begin(c, 6, &ts_bp)) return 1;
	if (M > 1) {
		HIPCHK(c, mpc_sort_pairs([&](size_t bytes) -> void * { return c->d_bp_tmp.ensure_grow(bytes) == hipSuccess ? c->d_bp_tmp.p : nullptr; },
			keys_in, keys_out, vals_in, vals_out, (size_t)M, bc, c->stream));
		keys_sorted = keys_out;
		vals_sorted = vals_out;
	}
	if (span_end(c, &ts_bp)) return 1;
	if (span_begin(c, 7, &ts_bp)) return 1;
	if (reduce_runs(c, rb, keys_sorted, vals_sorted, M, cells)) return 1;
	if (span_end(c, &ts_bp)) return 1;
	lap(2);
	c->last_post_cells = cells;
	if (!path) { // matrix only (mpcgpu_build_post); the staging record is reused by the next call: drain the stream
		HIPCHK(c, hipStreamSynchronize(c->stream));
		return 0;
	}
	// the staging record is reused by the next call: run_calc_aln ends with a wait for the stream
	const int rc_aln = run_calc_aln(c, c->d_aln_post.as<float>(), C1, C2, path, pathlen, score);
	lap(3);
	if (host_trace && (++acc_n % 100) == 0)
		fprintf(stderr, "[mpcgpu] align_alns host seconds after %llu calls: vectors %.3f, uploads %.3f, launches %.3f, calc_aln+syncs %.3f\n",
			(unsigned long long)acc_n, acc_t[0], acc_t[1], acc_t[2], acc_t[3]);
	return rc_aln;
}

int mpcgpu_align_alns_w(mpcgpu_ctx *c, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2, uint32_t C1,
	uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2, const float *w1, const float *w2, char *path,
	uint32_t *pathlen, float *score)
{
	if (!c) return 1;
	if (!path || !pathlen) return fail(c, "mpcgpu_align_alns: NULL argument");
	return build_post_impl(c, n1, seq1, n2, seq2, C1, C2, pos2col1, pos2col2, w1, w2, path, pathlen, score);
}

int mpcgpu_build_post(mpcgpu_ctx *c, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2, uint32_t C1,
	uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2, const float *w1, const float *w2, float *post)
{
	if (!c) return 1;
	if (!post) return fail(c, "mpcgpu_build_post: NULL argument");
	if (build_post_impl(c, n1, seq1, n2, seq2, C1, C2, pos2col1, pos2col2, w1, w2, nullptr, nullptr, nullptr)) return 1;
	return mpcgpu_get_last_post(c, C1, C2, post);
}

int mpcgpu_get_last_post(mpcgpu_ctx *c, uint32_t C1, uint32_t C2, float *post)
{
	if (!c) return 1;
	if (!post) return fail(c, "mpcgpu_get_last_post: NULL argument");
	if (c->last_post_cells == 0 || (u64)C1 * C2 != c->last_post_cells)
		return fail(c, "mpcgpu_get_last_post: the last matrix built on this context has %llu cells, not %u x %u", (u64)c->last_post_cells, C1, C2);
	HIPCHK(c, hipSetDevice(c->device));
	HIPCHK(c, hipMemcpyAsync(post, c->d_aln_post.p, (size_t)c->last_post_cells * 4, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	return 0;
}

int mpcgpu_align_msas(mpcgpu_ctx *c, uint32_t npairs, const uint32_t *seq1, const uint32_t *seq2, uint32_t C1, uint32_t C2,
	const uint32_t *pos2col1, const uint32_t *pos2col2, char *path, uint32_t *pathlen, float *score, float *ea_out)
{
	if (!c) return 1;
	if (c->n == 0) return fail(c, "mpcgpu_align_msas: call mpcgpu_set_seqs / mpcgpu_set_seqs_registry first");
	if (!seq1 || !seq2 || !pos2col1 || !pos2col2 || !path || !pathlen) return fail(c, "mpcgpu_align_msas: NULL argument");
	if (npairs == 0 || C1 == 0 || C2 == 0) return fail(c, "mpcgpu_align_msas: empty input");
	for (u32 q = 0; q < npairs; ++q)
		if (seq1[q] >= c->n || seq2[q] >= c->n) return fail(c, "mpcgpu_align_msas: sequence index out of range in pair %u", q);
	// ---- stage A on the listed pairs (X = the MSA1 sequence, Y = the MSA2 sequence: calcpost.cpp:4-36)
	if (stage_a(c, npairs, seq1, seq2)) return 1;
	if (ea_out) memcpy(ea_out, c->sh_ea.data(), (size_t)npairs * 4);
	// ---- CalcPosteriorFlat3 (buildposterior3flat.cpp:19-85): Flat[col1*C2+col2] += Prob in pair-list order
	std::vector<u64> off1(npairs + 1, 0), off2(npairs + 1, 0), coff(npairs + 1, 0), rbase(npairs + 1, 0);
	rbase[0] = shard_header_bytes(npairs) / 4; // records follow the shard header (words)
	for (u32 q = 0; q < npairs; ++q) {
		off1[q + 1] = off1[q] + c->len[seq1[q]];
		off2[q + 1] = off2[q] + c->len[seq2[q]];
		coff[q + 1] = coff[q] + c->sh_nnz[q];
		rbase[q + 1] = rbase[q] + rec_words(c->len[seq1[q]], c->len[seq2[q]], c->sh_nnz[q]);
	}
	for (u64 x = 0; x < off1[npairs]; ++x) if (pos2col1[x] >= C1) return fail(c, "mpcgpu_align_msas: column map of MSA1 out of range");
	for (u64 x = 0; x < off2[npairs]; ++x) if (pos2col2[x] >= C2) return fail(c, "mpcgpu_align_msas: column map of MSA2 out of range");
	const u64 M = coff[npairs];
	const u64 cells = (u64)C1 * C2;
	if (cells > 0xffffffffull) return fail(c, "mpcgpu_align_msas: %llu cells exceed this build's cell index", (u64)cells);
	const u32 bc = bits_for(cells - 1);
	if (M > 0xffffffffull) return fail(c, "mpcgpu_align_msas: %llu contributions exceed this build's record count", (u64)M);
	std::vector<u32> maps(off1[npairs] + off2[npairs]);
	memcpy(maps.data(), pos2col1, off1[npairs] * 4);
	memcpy(maps.data() + off1[npairs], pos2col2, off2[npairs] * 4);
	std::vector<u64> offs(off1);
	offs.insert(offs.end(), off2.begin(), off2.end());
	std::vector<u32> seqs(seq1, seq1 + npairs);
	seqs.insert(seqs.end(), seq2, seq2 + npairs);
	std::vector<u64> bases(coff);
	bases.insert(bases.end(), rbase.begin(), rbase.end());
	if (upload(c, c->d_bp_seq, seqs) || upload(c, c->d_bp_off, offs) || upload(c, c->d_bp_map, maps) || upload(c, c->d_bp_coff, bases))
		return 1;
	HIPCHK(c, c->d_bp_keys.ensure(std::max<u64>(M, 1) * 4 * 2));
	HIPCHK(c, c->d_bp_vals.ensure(std::max<u64>(M, 1) * 4 * 2));
	HIPCHK(c, c->d_aln_post.ensure(cells * 4));
	RunBufs rb;
	if (prepare_runs(c, M, cells, &rb)) return 1;
	u32 *keys_in = c->d_bp_keys.as<u32>(), *keys_out = keys_in + std::max<u64>(M, 1);
	float *vals_in = c->d_bp_vals.as<float>(), *vals_out = vals_in + std::max<u64>(M, 1);
	BuildPostListParams lp;
	lp.seq_len = c->d_seq_len.as<u32>();
	lp.packed = c->d_shard.as<u32>();
	lp.seq1 = c->d_bp_seq.as<u32>(); lp.seq2 = lp.seq1 + npairs; lp.npairs = npairs;
	lp.p2c1 = c->d_bp_map.as<u32>(); lp.p2c2 = lp.p2c1 + off1[npairs];
	lp.off1 = c->d_bp_off.as<u64>(); lp.off2 = lp.off1 + (npairs + 1);
	lp.coff = c->d_bp_coff.as<u64>(); lp.rbase = lp.coff + (npairs + 1);
	lp.nnz = nullptr; // counts come from coff
	lp.C2 = C2; lp.keys = keys_in; lp.vals = vals_in;
	lp.post = c->d_aln_post.as<float>(); lp.cells = cells; lp.counters = rb.counters;
	MPC_LAUNCH(build_post_list_gen_kernel, (u32)std::min<u64>(npairs, (u64)c->prop.multiProcessorCount * 32), 64, 0, c->stream, lp);
	HIPCHK(c, hipGetLastError());
	const u32 *keys_sorted = keys_in;
	const float *vals_sorted = vals_in;
	if (M > 1) {
		HIPCHK(c, mpc_sort_pairs([&](size_t bytes) -> void * { return c->d_bp_tmp.ensure_grow(bytes) == hipSuccess ? c->d_bp_tmp.p : nullptr; },
			keys_in, keys_out, vals_in, vals_out, (size_t)M, bc, c->stream));
		keys_sorted = keys_out;
		vals_sorted = vals_out;
	}
	if (reduce_runs(c, rb, keys_sorted, vals_sorted, M, cells)) return 1;
	c->last_post_cells = cells;
	return run_calc_aln(c, c->d_aln_post.as<float>(), C1, C2, path, pathlen, score); // syncs before the vectors above die
}

// mpcgpu_align_pairs for a SHORT list (what UClust::Search and single AlignPairFlat calls send: 1..8 pairs): the kernels of the
// general path, driven with one wait. Everything the kernels read from the host (pair list, launch order, alignment parameters)
// and everything the host reads back (candidate-overflow flags, path records) lives in ONE page-locked record that the device
// addresses directly; nothing is packed into a shard (mpcgpu_get_list_sparse re-runs the general stage when somebody asks).
// 0 = done, 1 = error, 2 = not applicable (the caller takes the general path).
static int align_pairs_small(mpcgpu_ctx *c, u32 np, const u32 *px, const u32 *py, u32 path_stride, char *paths, u32 *pathlens,
	float *scores, float *ea)
{
	const u32 long_min = (u32)std::min(std::max(env_int("MPCGPU_FB_LONG_MIN", 64 * 12 + 1), 2), 64 * MPC_HMAX + 1);
	u32 LXmax = 0, LYmax = 0, Lsum_max = 0;
	for (u32 q = 0; q < np; ++q) {
		const u32 LX = c->len[px[q]], LY = c->len[py[q]];
		if (LX >= long_min) return 2; // row-block pairs: general path
		if ((u64)LY + 1 > MPC_ALNW_MAXW || (size_t)(LX + 1) * MPC_ALNW_ROWBYTES + 16 > 160u * 1024u) return 2; // not a one-wave alignment
		LXmax = std::max(LXmax, LX); LYmax = std::max(LYmax, LY); Lsum_max = std::max(Lsum_max, LX + LY);
	}
	if (((size_t)LXmax + 2 + 2 * ((size_t)LYmax + 2)) * 4 + 8 + 8 * 1024 > 150 * 1024) return 2;
	const u32 Lmax = std::max(LXmax, LYmax);
	const u32 capc = std::max((u32)std::max(env_int("MPCGPU_CAND_PER_ROW", 12), 1) * Lmax, 1024u);
	const bool mega = c->have_mega;
	c->have_shard = c->have_store = false;
	c->shard_is_list = true;
	c->list_x.assign(px, px + np); c->list_y.assign(py, py + np);
	c->list_q0 = 0; c->ap_x.clear(); c->ap_y.clear();
	c->sh_k0 = 0; c->sh_k1 = np;
	// ---- the page-locked record
	std::vector<u64> off(np + 1, 0);
	for (u32 q = 0; q < np; ++q) off[q + 1] = off[q] + (u64)c->len[px[q]] * c->len[py[q]];
	const u64 rstride = ((u64)8 + Lsum_max + 7) & ~7ull;
	const u64 o_bx = 0, o_by = o_bx + 4 * (u64)np, o_order = o_by + 4 * (u64)np, o_off = (o_order + 4 * (u64)np + 7) & ~7ull,
		o_par = o_off + 8 * ((u64)np + 1), o_flags = o_par + (u64)np * sizeof(AlnParams), o_nnz = o_flags + 4 * (u64)np,
		o_ea = o_nnz + 4 * (u64)np, o_res = (o_ea + 4 * (u64)np + 7) & ~7ull, bytes = o_res + (u64)np * rstride;
	HIPCHK(c, c->h_ap.ensure(bytes));
	char *h = c->h_ap.as<char>();
	u32 *bx = (u32 *)(h + o_bx), *by = (u32 *)(h + o_by), *order = (u32 *)(h + o_order);
	memcpy(bx, px, 4 * (size_t)np);
	memcpy(by, py, 4 * (size_t)np);
	memcpy(h + o_off, off.data(), 8 * ((size_t)np + 1));
	u32 hcount[MPC_HMAX + 1] = {0};
	{
		std::vector<u32> keys(np);
		for (u32 q = 0; q < np; ++q) { const u32 H = (c->len[px[q]] + 63) / 64; hcount[H]++; keys[q] = (H << 16) | q; }
		std::sort(keys.begin(), keys.end());
		for (u32 q = 0; q < np; ++q) order[q] = keys[q] & 0xffffu;
	}
	// ---- scratch
	const u64 res_stride = (u64)LXmax + LYmax + 4 * (u64)capc;
	const int waves_per_block = 4, block = 64 * waves_per_block;
	const size_t fb_smem = (mega ? (size_t)c->mg_tab_floats : (size_t)c->A * c->A + c->A) * sizeof(float);
	HIPCHK(c, c->d_cand.ensure((u64)np * capc * 8));
	HIPCHK(c, c->d_cand_cnt.ensure((u64)np * 4));
	HIPCHK(c, c->d_total.ensure((u64)np * 4));
	HIPCHK(c, c->d_res.ensure((u64)np * res_stride * 4));
	HIPCHK(c, c->d_queue.ensure(4 * (MPC_HMAX + 2)));
	HIPCHK(c, c->d_aln_post.ensure_grow(off[np] * 4));
	HIPCHK(c, c->d_aln_rev.ensure_grow((u64)np * Lsum_max + 16));
	HIPCHK(c, hipMemsetAsync(c->d_queue.p, 0, 4 * (MPC_HMAX + 2), c->stream));
	// ---- forward / backward, one launch per rows-per-lane bin
	FbParams fp;
	fill_fb_params(c, fp, bx, by, capc, mega);
	TimedSpan sp;
	u32 pos = 0;
	for (u32 H = 1; H <= MPC_HMAX; ++H) {
		if (!hcount[H]) continue;
		const u32 cnt = hcount[H];
		const u32 grid = (cnt + waves_per_block - 1) / waves_per_block;
		const u64 fm_stride = (u64)(LYmax + 64) * H * 64;
		HIPCHK(c, c->d_fm.ensure((u64)grid * waves_per_block * fm_stride * 4));
		fp.order = order + pos; fp.count = cnt;
		fp.queue = c->d_queue.as<u32>() + H;
		fp.fm_scratch = c->d_fm.as<float>(); fp.fm_stride = fm_stride;
		if (span_begin(c, 0, &sp)) return 1;
		launch_fb_h((int)H, mega, fp, grid, block, fb_smem, c->stream);
		HIPCHK(c, hipGetLastError());
		if (span_end(c, &sp)) return 1;
		pos += cnt;
	}
	// ---- probabilities, EA, sparsify (the candidate lists keep the probabilities)
	PostRowsParams pr;
	pr.pair_x = bx; pr.pair_y = by; pr.seq_len = c->d_seq_len.as<u32>();
	pr.cand = c->d_cand.as<u64>(); pr.capc = capc; pr.cand_cnt = c->d_cand_cnt.as<u32>();
	pr.use_fma = c->use_fma;
	pr.lx_cap = LXmax + 2; pr.ly_cap = LYmax + 2;
	pr.sort_cap = std::min<u32>(capc, 1024u); pr.sort_stride = capc;
	pr.batch = 64;
	const size_t fixed_lds = ((((size_t)pr.lx_cap + 2 * (size_t)pr.ly_cap) * 4 + 7) & ~(size_t)7);
	const size_t smem = fixed_lds + (size_t)pr.sort_cap * 8;
	if (smem > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void *)post_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
	HIPCHK(c, c->d_sort_scratch.ensure(capc > pr.sort_cap ? (u64)np * pr.sort_stride * 8 : 8));
	pr.sort_scratch = c->d_sort_scratch.as<u64>();
	pr.res = c->d_res.as<u32>(); pr.res_stride = res_stride;
	pr.nnz = (u32 *)(h + o_nnz); pr.ea = (float *)(h + o_ea); pr.flags = (u32 *)(h + o_flags);
	pr.count = np; pr.long_min = long_min; pr.prof = nullptr;
	if (span_begin(c, 1, &sp)) return 1;
	MPC_LAUNCH(post_rows_kernel, np, 64, smem, c->stream, pr);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &sp)) return 1;
	// ---- dense thresholded posteriors, alignments
	DensePostParams dp;
	dp.pair_x = bx; dp.pair_y = by; dp.seq_len = c->d_seq_len.as<u32>();
	dp.cand = c->d_cand.as<u64>(); dp.capc = capc; dp.cand_cnt = c->d_cand_cnt.as<u32>(); dp.long_min = long_min;
	dp.out_off = (const u64 *)(h + o_off); dp.out = c->d_aln_post.as<float>();
	MPC_LAUNCH(dense_post_kernel, np, 256, 0, c->stream, dp);
	HIPCHK(c, hipGetLastError());
	c->last_post_cells = 0;
	AlnParams *ap = (AlnParams *)(h + o_par);
	for (u32 q = 0; q < np; ++q) {
		char *r = h + o_res + (u64)q * rstride;
		ap[q].post = c->d_aln_post.as<float>() + off[q];
		ap[q].LX = c->len[px[q]]; ap[q].LY = c->len[py[q]];
		ap[q].tb = nullptr; ap[q].rev = c->d_aln_rev.as<char>() + (u64)q * Lsum_max;
		ap[q].pathlen = (u32 *)r; ap[q].score = (float *)(r + 4); ap[q].path = r + 8;
	}
	const size_t asmem = (size_t)(LXmax + 1) * MPC_ALNW_ROWBYTES + 16;
	if (asmem > c->aln_smem_set[3]) {
		(void)hipFuncSetAttribute((const void *)calc_aln_wave_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)asmem);
		c->aln_smem_set[3] = asmem;
	}
	if (span_begin(c, 8, &sp)) return 1;
	MPC_LAUNCH(calc_aln_wave_batch_kernel, np, 64, asmem, c->stream, (const AlnParams *)ap);
	HIPCHK(c, hipGetLastError());
	if (span_end(c, &sp)) return 1;
	HIPCHK(c, hipStreamSynchronize(c->stream)); // the one wait
	for (u32 q = 0; q < np; ++q)
		if (((const u32 *)(h + o_flags))[q] & 1u) return 2; // a candidate list overflowed: the general path grows it and retries
	c->sh_nnz.assign((const u32 *)(h + o_nnz), (const u32 *)(h + o_nnz) + np);
	c->sh_ea.assign((const float *)(h + o_ea), (const float *)(h + o_ea) + np);
	for (u32 q = 0; q < np; ++q) {
		const char *r = h + o_res + (u64)q * rstride;
		u32 n_path;
		memcpy(&n_path, r, 4);
		if (n_path > ap[q].LX + ap[q].LY) return fail(c, "mpcgpu_align_pairs: path length %u out of range (internal error)", n_path);
		pathlens[q] = n_path;
		float sc;
		memcpy(&sc, r + 4, 4);
		if (scores) scores[q] = sc;
		if (ea) ea[q] = sc / (float)std::min(ap[q].LX, ap[q].LY); // alignpairflat.cpp:18 (uint -> float, IEEE divide)
		memcpy(paths + (u64)q * path_stride, r + 8, n_path);
	}
	return 0;
}

// AlignPairFlat (alignpairflat.cpp:3-27) for a list of pairs: CalcPost (fwd + bwd + CalcPostFlat, calcpost.cpp:4-36) -> CalcAlnFlat on
// the DENSE thresholded posterior -> path; EA = Score / min(L1, L2). Stage A runs on the list (the kernels of
// mpcgpu_calc_posteriors), the dense matrices are rebuilt from the candidate lists (every cell with Score >= MIN_SPARSE_SCORE, also
// those FromPost drops), the alignments run one wavefront each in ONE launch when they fit (else one after the other).
int mpcgpu_align_pairs(mpcgpu_ctx *c, uint32_t npairs, const uint32_t *seq1, const uint32_t *seq2, uint32_t path_stride, char *paths,
	uint32_t *pathlens, float *scores, float *ea)
{
	if (!c) return 1;
	if (c->n == 0) return fail(c, "mpcgpu_align_pairs: call mpcgpu_set_seqs / mpcgpu_set_seqs_registry first");
	if (!seq1 || !seq2 || !paths || !pathlens) return fail(c, "mpcgpu_align_pairs: NULL argument");
	for (u32 q = 0; q < npairs; ++q) {
		if (seq1[q] >= c->n || seq2[q] >= c->n) return fail(c, "mpcgpu_align_pairs: sequence index out of range in pair %u", q);
		if ((u64)c->len[seq1[q]] + c->len[seq2[q]] > path_stride) return fail(c, "mpcgpu_align_pairs: path_stride %u too small for pair %u", path_stride, q);
	}
	HIPCHK(c, hipSetDevice(c->device));
	if (npairs >= 1 && npairs <= 64 && env_int("MPCGPU_PAIRS_SMALL", 1)) {
		const int rc = align_pairs_small(c, npairs, seq1, seq2, path_stride, paths, pathlens, scores, ea);
		if (rc != 2) return rc;
	}
	u32 chunk = 256; // pairs per stage-A call: their dense matrices (LX*LY floats each) live together
	struct KeepList { mpcgpu_ctx *c; ~KeepList() { c->ap_keep = false; } } keep_guard{c};
	c->ap_keep = true;
	c->ap_x.assign(seq1, seq1 + npairs); c->ap_y.assign(seq2, seq2 + npairs);
	for (u32 q0 = 0; q0 < npairs;) {
		u32 nq = std::min<u32>(chunk, npairs - q0);
		// the dense matrices below are rebuilt from the candidate lists ONE stage-A batch leaves behind: a chunk that stage A had to
		// cut into several batches (long sequences, little free memory) is halved and run again, down to a single pair
		for (;;) {
			if (stage_a(c, nq, seq1 + q0, seq2 + q0)) return 1;
			c->list_q0 = q0; // what the last stage holds is pairs [q0, q0 + nq) of the caller's list
			if (c->sa_b0 == 0 && c->sa_B == nq) break;
			if (nq == 1) return fail(c, "mpcgpu_align_pairs: pair %u (%u x %u residues) does not fit one stage-A batch", q0, c->len[seq1[q0]], c->len[seq2[q0]]);
			nq = (nq + 1) / 2;
			chunk = nq;
		}
		if (!c->sa_post_rows)
			return fail(c, "mpcgpu_align_pairs: pair list with a sequence of more than ~12 000 residues (or MPCGPU_POST=sort): the row-list finishing kernel "
				"whose candidate lists this entry point rebuilds the dense posteriors from does not take them");
		// dense matrices
		std::vector<u64> off(nq + 1, 0);
		u32 Lsum_max = 0, LXmax = 0;
		bool all_wave = true;
		for (u32 q = 0; q < nq; ++q) {
			const u32 LX = c->len[seq1[q0 + q]], LY = c->len[seq2[q0 + q]];
			off[q + 1] = off[q] + (u64)LX * LY;
			Lsum_max = std::max(Lsum_max, LX + LY);
			LXmax = std::max(LXmax, LX);
			all_wave = all_wave && (u64)LY + 1 <= MPC_ALNW_MAXW && (size_t)(LX + 1) * MPC_ALNW_ROWBYTES + 16 <= 160u * 1024u;
		}
		HIPCHK(c, c->d_aln_post.ensure_grow(off[nq] * 4));
		if (upload(c, c->d_ap_off, off)) return 1;
		DensePostParams dp;
		dp.pair_x = c->d_bx.as<u32>(); dp.pair_y = c->d_by.as<u32>(); dp.seq_len = c->d_seq_len.as<u32>();
		dp.cand = c->d_cand.as<u64>(); dp.capc = c->sa_capc; dp.cand_cnt = c->d_cand_cnt.as<u32>(); dp.long_min = c->sa_long_min;
		dp.out_off = c->d_ap_off.as<u64>(); dp.out = c->d_aln_post.as<float>();
		MPC_LAUNCH(dense_post_kernel, nq, 256, 0, c->stream, dp);
		HIPCHK(c, hipGetLastError());
		c->last_post_cells = 0; // several matrices: not what mpcgpu_get_last_post hands out
		if (all_wave) {
			// one launch: parameters in, {length, score, path} out through page-locked memory
			const u64 rstride = ((u64)8 + Lsum_max + 7) & ~7ull;
			const u64 o_par = 0, o_res = o_par + (u64)nq * sizeof(AlnParams), bytes = o_res + (u64)nq * rstride;
			HIPCHK(c, c->h_ap.ensure(bytes));
			HIPCHK(c, c->d_aln_rev.ensure_grow((u64)nq * Lsum_max + 16));
			char *h = c->h_ap.as<char>();
			AlnParams *ap = (AlnParams *)(h + o_par);
			for (u32 q = 0; q < nq; ++q) {
				char *r = h + o_res + (u64)q * rstride;
				ap[q].post = c->d_aln_post.as<float>() + off[q];
				ap[q].LX = c->len[seq1[q0 + q]]; ap[q].LY = c->len[seq2[q0 + q]];
				ap[q].tb = nullptr; ap[q].rev = c->d_aln_rev.as<char>() + (u64)q * Lsum_max;
				ap[q].pathlen = (u32 *)r; ap[q].score = (float *)(r + 4); ap[q].path = r + 8;
			}
			const size_t smem = (size_t)(LXmax + 1) * MPC_ALNW_ROWBYTES + 16;
			if (smem > c->aln_smem_set[3]) {
				(void)hipFuncSetAttribute((const void *)calc_aln_wave_batch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
				c->aln_smem_set[3] = smem;
			}
			TimedSpan ts;
			if (span_begin(c, 8, &ts)) return 1;
			MPC_LAUNCH(calc_aln_wave_batch_kernel, nq, 64, smem, c->stream, (const AlnParams *)ap);
			HIPCHK(c, hipGetLastError());
			if (span_end(c, &ts)) return 1;
			HIPCHK(c, hipStreamSynchronize(c->stream));
			for (u32 q = 0; q < nq; ++q) {
				const char *r = h + o_res + (u64)q * rstride;
				u32 n_path;
				memcpy(&n_path, r, 4);
				if (n_path > ap[q].LX + ap[q].LY) return fail(c, "mpcgpu_align_pairs: path length %u out of range (internal error)", n_path);
				pathlens[q0 + q] = n_path;
				float sc;
				memcpy(&sc, r + 4, 4);
				if (scores) scores[q0 + q] = sc;
				if (ea) ea[q0 + q] = sc / (float)std::min(ap[q].LX, ap[q].LY); // alignpairflat.cpp:18 (uint -> float, IEEE divide)
				memcpy(paths + (u64)(q0 + q) * path_stride, r + 8, n_path);
			}
		} else {
			for (u32 q = 0; q < nq; ++q) {
				const u32 LX = c->len[seq1[q0 + q]], LY = c->len[seq2[q0 + q]];
				float sc = 0;
				if (run_calc_aln(c, c->d_aln_post.as<float>() + off[q], LX, LY, paths + (u64)(q0 + q) * path_stride, &pathlens[q0 + q], &sc)) return 1;
				if (scores) scores[q0 + q] = sc;
				if (ea) ea[q0 + q] = sc / (float)std::min(LX, LY);
			}
		}
		q0 += nq;
	}
	return 0;
}

int mpcgpu_get_list_sparse(mpcgpu_ctx *c, uint32_t q, uint32_t *nnz, uint32_t *offsets, void *values)
{
	if (!c) return 1;
	if (!c->shard_is_list) return fail(c, "mpcgpu_get_list_sparse: no list stage holds pair %u", q);
	HIPCHK(c, hipSetDevice(c->device));
	// q indexes the list the CALLER passed. mpcgpu_align_pairs may have run that list in chunks (256 pairs, halved when stage A had to
	// split one): the last stage then holds pairs [list_q0, list_q0 + list_x.size()) of it. A pair outside that window gets a stage
	// of its own (same kernels, same bits) — never another pair's record.
	const bool want_record = offsets && values;
	if (c->ap_x.empty() && !c->have_shard && want_record) { c->ap_x = c->list_x; c->ap_y = c->list_y; c->list_q0 = 0; } // the short-list path packed nothing
	u32 ql = q;
	if (!c->ap_x.empty()) {
		if (q >= c->ap_x.size()) return fail(c, "mpcgpu_get_list_sparse: no list stage holds pair %u", q);
		struct KeepList { mpcgpu_ctx *c; ~KeepList() { c->ap_keep = false; } } keep_guard{c};
		c->ap_keep = true;
		bool inside = q >= c->list_q0 && q - c->list_q0 < c->list_x.size();
		if (inside && !c->have_shard && want_record) { // the general stage on the window's list packs the records
			const std::vector<u32> lx = c->list_x, ly = c->list_y;
			const u32 q0 = c->list_q0;
			if (stage_a(c, lx.size(), lx.data(), ly.data())) return 1;
			c->list_q0 = q0;
			inside = c->sa_b0 == 0 && c->sa_B == lx.size(); // (split into batches: the shard holds the last batch only)
		}
		if (!inside) {
			// the whole chunk that holds q is staged again and stays resident (a caller that reads q = 0, 1, 2 ... in order pays one stage
			// per 256 pairs, not one per pair: round-5 advisor finding); a chunk that stage A has to split falls back to the single pair
			const u32 q0 = (q / 256u) * 256u, nq = (u32)std::min<size_t>(256u, c->ap_x.size() - q0);
			const std::vector<u32> lx(c->ap_x.begin() + q0, c->ap_x.begin() + q0 + nq), ly(c->ap_y.begin() + q0, c->ap_y.begin() + q0 + nq);
			if (stage_a(c, nq, lx.data(), ly.data())) return 1;
			c->list_q0 = q0;
			if (!(c->sa_b0 == 0 && c->sa_B == nq)) {
				const u32 x = c->ap_x[q], y = c->ap_y[q];
				if (stage_a(c, 1, &x, &y)) return 1;
				c->list_q0 = q;
			}
		}
		ql = q - c->list_q0;
	}
	if (ql >= c->list_x.size() || ql >= c->sh_nnz.size()) return fail(c, "mpcgpu_get_list_sparse: no list stage holds pair %u", q);
	const u64 np = c->list_x.size();
	u64 w = shard_header_bytes(np) / 4;
	for (u32 k = 0; k < ql; ++k) w += rec_words(c->len[c->list_x[k]], c->len[c->list_y[k]], c->sh_nnz[k]);
	const u32 LX = c->len[c->list_x[ql]], LY = c->len[c->list_y[ql]], nz = c->sh_nnz[ql];
	if (nnz) *nnz = nz;
	if (!want_record) return 0;
	if (!c->have_shard) return fail(c, "mpcgpu_get_list_sparse: the list stage of pair %u left no packed records (internal error)", q);
	std::vector<u32> rowcnt(LX);
	HIPCHK(c, hipMemcpyAsync(rowcnt.data(), c->d_shard.as<u32>() + w, (size_t)LX * 4, hipMemcpyDeviceToHost, c->stream));
	if (nz) HIPCHK(c, hipMemcpyAsync(values, c->d_shard.as<u32>() + w + LX + LY, (size_t)nz * 8, hipMemcpyDeviceToHost, c->stream));
	HIPCHK(c, hipStreamSynchronize(c->stream));
	u32 acc = 0;
	for (u32 i = 0; i < LX; ++i) { offsets[i] = acc; acc += rowcnt[i]; }
	offsets[LX] = acc;
	if (acc != nz) return fail(c, "mpcgpu_get_list_sparse: record of pair %u is inconsistent (internal error)", q);
	return 0;
}

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
