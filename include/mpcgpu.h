/*
 * mpcgpu.h — C ABI of libmpcgpu.so: the MI355X (gfx950) implementation of MUSCLE5's MPCFlat
 * all-pairs posterior stage. Plain pointers and sizes only; no C++/torch types.
 *
 * The reference (rcedgar/muscle 5.3) has no plugin/FFI API: its seam is C++ link time, one
 * numeric function per translation unit. Each entry point below names the
 * reference interface it replaces (paths relative to the reference's src/). The C++ drop-in
 * translation units that bind these into an unmodified mpcflat.{h,cpp} are in hostcxx/ and
 * described in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; mpcgpu_last_error() then holds a
 *    message (the reference has no error codes: fatal -> Die(), myutils.cpp:883; the C++ shim
 *    converts non-zero into Die()).
 *  - pair index k enumerates (i<j) row-major over the n sequences, exactly MPCFlat::InitPairs
 *    (mpcflat.cpp:139-159); "pair range [k0,k1)" arguments select a contiguous shard of it.
 *  - sparse posterior matrices cross the boundary in MySparseMx's own layout
 *    (mysparsemx.h:6-98): uint32 offsets[LX+1] and nnz 8-byte entries {float P; uint32 col},
 *    rows ascending, columns ascending.
 *  - the library never falls back to a CPU implementation: without a usable gfx950 device
 *    mpcgpu_create() fails.
 */
#ifndef MPCGPU_H
#define MPCGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpcgpu_ctx mpcgpu_ctx;

/* Library / device lifetime. One context = one GPU = one MPCFlat::Run at a time
 * (device buffers live for one Run: mpcflat.cpp:285-337). */
int mpcgpu_create(mpcgpu_ctx **out, int device_ordinal);
void mpcgpu_destroy(mpcgpu_ctx *ctx);
const char *mpcgpu_last_error(const mpcgpu_ctx *ctx); /* ctx may be NULL: last create() error */
const char *mpcgpu_version(void);

/* Kernel constants. Replaces the process-global PairHMM::m_StartScore[5] / m_TransScore[5][5] /
 * m_MatchScore[256][256] / m_InsScore[256] (pairhmm.h:23-29; filled by HMMParams::ToPairHMM,
 * hmmparams.cpp:298-409; bound inside every DP function by hmmscores.h:1-16). The tables can
 * change between replicates (align.cpp:35-40) so this is called once per Run.
 * min_sparse_score = MIN_SPARSE_SCORE = logf(0.01f) as evaluated by the host (mysparsemx.h:4).
 * expf_variant selects which glibc expf build the device reproduces for calcposteriorflat.cpp:20:
 *   0 = baseline (separate mul/add), 1 = FMA build, -1 = whatever this host's libm resolves to. */
int mpcgpu_set_hmm(mpcgpu_ctx *ctx, const float start[5], const float trans[25],
                   const float match[256 * 256], const float ins[256],
                   float min_sparse_score, int expf_variant);

/* Input sequences. Replaces MPCFlat::InitSeqs + GetBytePtr/GetSeqLength (mpcflat.cpp:41-57,
 * 121-137): n raw ASCII byte strings (already upper-cased by the loader, sequence.cpp:87-88).
 * The bytes are copied; every seven-bit byte value is taken (the emission tables are compacted to the A distinct values that
 * occur, A <= 128, and live in LDS: (A*A + A) floats — beyond the 64 KB default the library asks for the larger dynamic LDS).
 * Bytes >= 128 are refused: the reference indexes m_MatchScore[256][256] by a plain `char` (fwdflat3.cpp:102-109), a negative
 * index there.
 * Also performs MPCFlat::InitPairs (all i<j pairs). Length overflow check of
 * calcposteriorflat.cpp:54-61 (LX*LY*5+100 > INT_MAX) is preserved as an error. Build limits beyond
 * the reference's: a pair whose row sequence is longer than 768 takes the row-block kernels (a choice up to
 * 1024, the only way beyond), which need both lengths <= 65535 (the reference's own limit is ~21k x 21k); the
 * consistency relax runs LDS-tiled while the records of a pair fit the CU's LDS (sequences of up to ~1000-2000 residues,
 * depending on how many cells a row stores; never beyond 4095) and through the gather kernel beyond —
 * mpcgpu_relax_info says which. */
int mpcgpu_set_seqs(mpcgpu_ctx *ctx, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens);

/* Structure-profile ("mega") emissions for stage A. Replaces the Mega statics a .mega input fills
 * (Mega::FromFile, mega.cpp:119-270; mega.h:9-33) as CalcPost uses them when Mega::m_Loaded
 * (calcpost.cpp:14-22): Mega::CalcFwdFlat_mega / CalcBwdFlat_mega (fwdflat_mega.cpp:14-165,
 * bwdflat_mega.cpp:13-193) with Mega::GetInsScore (mega.cpp:273-285) and Mega::GetMatchScore
 * (mega.cpp:341-363) in place of the PairHMM emission tables; transitions still come from set_hmm.
 *   nfeat                 Mega::m_FeatureCount (<= 8 in this build); 0 switches back to byte sequences
 *   alpha[f], weight[f]   Mega::m_AlphaSizes, Mega::m_Weights
 *   logprobs[f]           Mega::m_LogProbsVec[f]: alpha[f] floats
 *   logprob_mx[f]         Mega::m_LogProbMxVec[f]: alpha[f] x alpha[f] floats, row-major
 *   profiles[i]           profile of sequence i of the last set_seqs/set_seqs_registry (Mega::m_Profiles
 *                         via GetProfileByLabel): len[i] positions x nfeat letters, position-major
 * Call after every set_seqs (which drops the previous profiles). Everything downstream of the
 * emissions (posterior, sparsify, EA, relax, joins) is unchanged. */
int mpcgpu_set_mega(mpcgpu_ctx *ctx, uint32_t nfeat, const uint32_t *alpha, const float *weight,
                    const float *const *logprobs, const float *const *logprob_mx,
                    const uint8_t *const *profiles);

uint64_t mpcgpu_pair_count(const mpcgpu_ctx *ctx); /* n(n-1)/2 */

/* ---- pair order of a block-partitioned multi-GPU run -------------------------------
 * MPCFlat::InitPairs (mpcflat.cpp:139-159) enumerates the pairs row-major, and a contiguous range of that order holds pairs
 * (X, Y) with Y anywhere behind X: a rank that owns such a range needs the sparse matrices of nearly every sequence for
 * MPCFlat::ConsPair (conspairflat.cpp:37-92). A rank that owns BLOCKS of the pair triangle (sequence group i x group j) needs
 * those of a few groups only. mpcgpu_set_pair_order makes the context enumerate its pairs rectangle by rectangle (row-major
 * inside each): rects[4 r ..] = {xa, xb, ya, yb} is either off the diagonal (ya >= xb: every (x, y) of [xa,xb) x [ya,yb)) or a
 * triangle (xa == ya, xb == yb: the pairs x < y inside [xa,xb)); together they must hold every pair exactly once. nrects == 0:
 * back to InitPairs order. Call after mpcgpu_set_seqs (which resets it), before stage A.
 * From then on the SHARDING calls — mpcgpu_calc_posteriors, mpcgpu_store_import(_part), mpcgpu_values_slice, mpcgpu_cons_iter —
 * take POSITIONS in this order ("[k0,k1)" = the k0-th to k1-th pair of the enumeration), and the values array is in position
 * order; the RESULT getters (mpcgpu_get_ea, _get_nnz, _get_sparse, _get_sparse_range) and everything that names sequences
 * (mpcgpu_align_alns ...) keep speaking InitPairs pair numbers / sequence indices. */
int mpcgpu_set_pair_order(mpcgpu_ctx *ctx, uint32_t nrects, const uint32_t *rects);
/* position of pair (x, y), x < y, in the context's pair order */
int mpcgpu_pair_position(mpcgpu_ctx *ctx, uint32_t x, uint32_t y, uint64_t *pos);
/* The partition itself (host only, no context): n sequences of the given lengths over `world` ranks. Sequences fall into g groups
 * of equal weight (sum of L + 1), the pair triangle into g (g + 1) / 2 blocks, a rank owns whole blocks: world = g (g - 1) / 2 +
 * g / 2 with g even (2, 8, 18 ...) — one off-diagonal block per rank and one rank per two diagonal blocks (8 ranks: every rank
 * touches 2 of 4 groups, half of the store); any other world — g = world, rank i owns the triangle of group i and the blocks
 * {i, i + d}, d < g / 2, the antipodal blocks of an even g cut in two by rows. rects (capacity max_rects x 4 words) receives the
 * rectangles in rank order, rank_pos[world + 1] the positions where each rank's pairs begin: rank r owns [rank_pos[r],
 * rank_pos[r + 1]). *nrects == 0: no block cut (one rank, too few sequences): InitPairs order,
 * contiguous ranges balanced by DP cells — the partition of rounds 1-5. Returns 0, or 2 when max_rects is too small. */
int mpcgpu_plan_partition(uint32_t n, const uint32_t *lens, uint32_t world, uint32_t max_rects, uint32_t *rects,
                          uint32_t *nrects, uint64_t *rank_pos);

/* Stage A for the pair range [k0,k1): replaces MPCFlat::CalcPosteriors' OpenMP loop
 * (mpcflat.cpp:239-251) over MPCFlat::CalcPosterior (calcposteriorflat.cpp:45-92), i.e.
 * CalcFwdFlat (fwdflat3.cpp:12) + CalcBwdFlat (bwdflat3.cpp:10) + CalcTotalProbFlat
 * (totalprobflat.cpp:3) + CalcPostFlat (calcposteriorflat.cpp:4) + MySparseMx::FromPost
 * (mysparsemx.cpp:115) + CalcAlnScoreFlat (calcalnscoreflat.cpp:4) + EA = Score/min(LX,LY).
 * Results stay on the device ("packed shard"). */
int mpcgpu_calc_posteriors(mpcgpu_ctx *ctx, uint64_t k0, uint64_t k1);

/* Builds the device-resident all-pairs store the relax kernel reads, from this context's own
 * shard. Only valid when the shard is the full range [0, pair_count) (single GPU). */
int mpcgpu_build_store(mpcgpu_ctx *ctx);

/* ---- multi-GPU exchange (one process per GPU; the collective itself is the caller's RCCL
 * all-gather over xGMI on these device pointers — SURVEY.md §8e) --------------------------- */
/* Size in bytes and device address of this context's packed shard (valid after calc_posteriors). */
int mpcgpu_shard_info(mpcgpu_ctx *ctx, uint64_t *bytes, void **dev_ptr);
/* Stored cells (sparse posterior entries) of this context's shard: this rank's share of the values a relax iteration exchanges
 * (mpcgpu_values_slice gives the same count once the store exists; this one is known right after stage A, so that a caller sizes
 * both exchanges — the shards' bytes and the values' counts — with ONE size exchange). */
int mpcgpu_shard_entries(mpcgpu_ctx *ctx, uint64_t *entries);
/* Copy the packed shard (bytes from mpcgpu_shard_info) into caller-owned device memory. */
int mpcgpu_shard_export(mpcgpu_ctx *ctx, void *dev_dst);
/* Build the all-pairs store from nshards packed shards laid out back to back in device memory
 * (dev_all; shard s occupies bytes[s] bytes, covers pairs [k0[s],k1[s]), ascending, contiguous,
 * covering [0,pair_count)). The library ADOPTS dev_all as the packed half of its store: it must stay
 * valid and otherwise untouched until the next set_seqs/destroy, and mpcgpu_cons_commit writes the
 * new probabilities into it (the swap of consflat.cpp:22) — hence not const. */
int mpcgpu_store_import(mpcgpu_ctx *ctx, uint32_t nshards, const uint64_t *k0, const uint64_t *k1,
                        const uint64_t *bytes, void *dev_all);
/* The same for a rank of a pair-sharded run whose stage A ran in pieces: the shards may come in any order and lie anywhere in
 * dev_all (offsets[s], multiples of 4; NULL: back to back in the order given); sorted by k0 they must tile [0, pair_count).
 * [own_k0, own_k1) = the positions this context will relax (mpcgpu_cons_iter): the store is PARTIAL — it holds the row-indexed
 * matrices (A, Z) of the sequences A that those pairs touch and of no other (the block partition: half of the store at 8 ranks;
 * what a ConsPair of the range reads, conspairflat.cpp:49-89), while everything per pair (packed matrices, values, EA) is
 * complete. mpcgpu_cons_iter outside the range is refused; mpcgpu_cons_commit(_range) commits every entry into what exists.
 * own = [0, pair_count) is mpcgpu_store_import. */
int mpcgpu_store_import_part(mpcgpu_ctx *ctx, uint32_t nshards, const uint64_t *k0, const uint64_t *k1, const uint64_t *bytes,
                             const uint64_t *offsets, void *dev_all, uint64_t own_k0, uint64_t own_k1);
/* Makes a partial store whole (the matrices of every sequence, from the packed ones, which hold the current values): what the
 * host that reads results — progressive alignment, MPCFlat::BuildPost — needs after the last mpcgpu_cons_iter. mpcgpu_align_alns
 * and mpcgpu_build_post call it themselves; a no-op on a complete store. */
int mpcgpu_store_complete(mpcgpu_ctx *ctx);
/* Device address + element count of the float array holding the next-iteration probabilities of
 * ALL pairs in canonical order (pair ascending, entries row-major), and the [first, first+count)
 * slice that cons_iter(k0,k1) writes. The caller all-gathers the slices in place, then commits. */
int mpcgpu_values_info(mpcgpu_ctx *ctx, void **dev_ptr, uint64_t *total_count);
int mpcgpu_values_slice(mpcgpu_ctx *ctx, uint64_t k0, uint64_t k1, uint64_t *first, uint64_t *count);
/* Copy count floats starting at canonical entry index `first` out of / into the values array
 * (caller-owned device memory on the other side) — the staging form of the same exchange. */
int mpcgpu_values_export(mpcgpu_ctx *ctx, uint64_t first, uint64_t count, void *dev_dst);
int mpcgpu_values_import(mpcgpu_ctx *ctx, uint64_t first, uint64_t count, const void *dev_src);

/* One consistency iteration for the pair range [k0,k1): replaces MPCFlat::ConsIter
 * (consflat.cpp:5-23) -> ConsPair (conspairflat.cpp:10-110) -> RelaxFlat_{XZ_ZY,ZX_ZY,XZ_YZ}
 * (relaxflat.cpp:4-94) -> MySparseMx::UpdateFromPost (mysparsemx.cpp:87-113). Jacobi: reads the
 * current store, writes the "values" array; mpcgpu_cons_commit makes them current (the buffer
 * swap of consflat.cpp:22). */
int mpcgpu_cons_iter(mpcgpu_ctx *ctx, uint64_t k0, uint64_t k1);
int mpcgpu_cons_commit(mpcgpu_ctx *ctx);
/* The same swap for the canonical entries [first, first+count) only (see mpcgpu_values_slice): a rank of a pair-sharded run
 * commits the slice it relaxed itself while the other ranks' values are still arriving, and the rest after them. Committing
 * every entry exactly once, in any split, equals mpcgpu_cons_commit (consflat.cpp:22). */
int mpcgpu_cons_commit_range(mpcgpu_ctx *ctx, uint64_t first, uint64_t count);

/* ---- results back to the host (the reference's in-memory formats) ------------------------- */
/* EA per pair: what calcposteriorflat.cpp:89-91 stores in m_DistMx[i][j]. Valid for own shard
 * after calc_posteriors, for all pairs after build_store/store_import. */
int mpcgpu_get_ea(mpcgpu_ctx *ctx, uint64_t k0, uint64_t k1, float *ea);
/* nnz per pair (MySparseMx::m_VecSize). */
int mpcgpu_get_nnz(mpcgpu_ctx *ctx, uint64_t k0, uint64_t k1, uint32_t *nnz);
/* Current sparse matrix of pair k in MySparseMx layout: offsets[LX+1], values[8*nnz bytes]. */
int mpcgpu_get_sparse(mpcgpu_ctx *ctx, uint64_t k, uint32_t *offsets, void *values);
/* Bulk form for [k0,k1): offsets concatenated (sum of LX+1 uint32), values concatenated
 * (8*sum nnz bytes). Sizes via mpcgpu_get_nnz and the sequence lengths. */
int mpcgpu_get_sparse_range(mpcgpu_ctx *ctx, uint64_t k0, uint64_t k1, uint32_t *offsets, void *values);

/* The finishing kernels on one caller-supplied candidate list (cells (rows[q], cols[q]) with log-space Score scores[q] >=
 * MIN_SPARSE_SCORE, any order, no duplicates): the expf half of CalcPostFlat (calcposteriorflat.cpp:16-22),
 * MySparseMx::FromPost (mysparsemx.cpp:115-152) and CalcAlnScoreFlat / EA (calcalnscoreflat.cpp:4-32,
 * calcposteriorflat.cpp:89) for ONE pair of lengths LX, LY. kernel: 0 = row-list kernel, 1 = general (sort) kernel;
 * batch: cells of a row per EA pass (64; smaller values exercise the multi-pass path). offsets[LX+1] / values (8 bytes per
 * kept entry, capacity ncand) may be NULL. A unit-test reach into inputs the pair-HMM itself never produces. */
int mpcgpu_post_scores(mpcgpu_ctx *ctx, uint32_t LX, uint32_t LY, uint32_t ncand, const uint32_t *rows,
                       const uint32_t *cols, const float *scores, int kernel, uint32_t batch, float *ea, uint32_t *nnz,
                       uint32_t *offsets, void *values);

/* Posterior-DP alignment of one dense LX x LY matrix resident in HOST memory: replaces
 * CalcAlnFlat (calcalnflat.cpp:6-46) + TraceBackFlat (tracebackflat.cpp:3-37), tie order of
 * best3.h:5-28. path receives the B/X/Y string (capacity >= LX+LY), not NUL-terminated. */
int mpcgpu_calc_aln(mpcgpu_ctx *ctx, const float *post, uint32_t LX, uint32_t LY,
                    char *path, uint32_t *pathlen, float *score);

/* Alignment of two alignments on the device: replaces MPCFlat::AlignAlns' numeric part
 * (alnalnsflat.cpp:7-52) = MPCFlat::BuildPost (buildpostflat.cpp:18-106: dense C1 x C2 matrix, sum over
 * s in MSA1 (outer), t in MSA2 (inner) of the CURRENT pairwise posteriors of the device store scattered
 * through the position->column maps, weights 1.0f as set at mpcflat.cpp:324) followed by CalcAlnFlat +
 * TraceBackFlat (calcalnflat.cpp:6-46, tracebackflat.cpp:3-37) on it. Bit-exact: every cell receives
 * its contributions in the reference's (s,t) order.
 * seq1[n1] / seq2[n2]: sequence indices (InitPairs numbering) of the rows of MSA1 / MSA2, in row order;
 * pos2col1 / pos2col2: Sequence::GetPosToCol (sequence.cpp:144-154) of every row, concatenated
 * (row a contributes len(seq a) entries); C1, C2: column counts. path: B/X/Y string, capacity C1+C2. */
int mpcgpu_align_alns(mpcgpu_ctx *ctx, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2,
                      uint32_t C1, uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2,
                      char *path, uint32_t *pathlen, float *score);
/* The same with sequence weights: every contribution is (w1[a] * w2[b]) * P, the product of the two weights rounded first,
 * as at buildpostflat.cpp:41,52,74,96 (w1[a] / w2[b] = the reference's m_Weights[SeqIndex1] / m_Weights[SeqIndex2] of row a of
 * MSA1 / row b of MSA2; NULL = all 1.0f, which is what MPCFlat::Run sets at mpcflat.cpp:324). */
int mpcgpu_align_alns_w(mpcgpu_ctx *ctx, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2,
                        uint32_t C1, uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2,
                        const float *w1, const float *w2, char *path, uint32_t *pathlen, float *score);

/* mpcgpu_align_alns for a LIST of independent joins: the joins of one level of the guide tree MPCFlat::ProgressiveAlign walks
 * (progalnflat.cpp:72-100 runs its N - 1 joins one after the other; a join needs only its two children). Join j aligns n1[j] rows
 * with n2[j] rows, C1[j] x C2[j] columns; seqs holds the sequence indices of its rows, MSA1's then MSA2's, the joins back to back;
 * pos2col likewise the rows' position -> column maps. All weights are 1.0f (what MPCFlat::Run sets, mpcflat.cpp:324).
 * paths: njoins slots of path_stride bytes (>= C1[j] + C2[j]); scores may be NULL. The small joins — nearly all of a tree's — run
 * in two launches together, the few large ones as mpcgpu_align_alns does; every matrix and path is the single call's. */
int mpcgpu_align_alns_batch(mpcgpu_ctx *ctx, uint32_t njoins, const uint32_t *n1, const uint32_t *n2, const uint32_t *C1,
                            const uint32_t *C2, const uint32_t *seqs, const uint32_t *pos2col, uint32_t path_stride,
                            char *paths, uint32_t *pathlens, float *scores);

/* MPCFlat::BuildPost alone (buildpostflat.cpp:18-106): the C1 x C2 matrix (row-major floats, host memory) that
 * mpcgpu_align_alns_w would align — for the callers of BuildPost outside MPCFlat::Run (profseq.cpp:33-49). Arguments as
 * for mpcgpu_align_alns_w (w1 / w2 may be NULL: all weights 1.0f). */
int mpcgpu_build_post(mpcgpu_ctx *ctx, uint32_t n1, const uint32_t *seq1, uint32_t n2, const uint32_t *seq2, uint32_t C1,
                      uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2, const float *w1, const float *w2,
                      float *post);
/* The dense matrix of the last mpcgpu_align_alns(_w) / mpcgpu_build_post / mpcgpu_align_msas / mpcgpu_calc_aln call on this
 * context (C1 x C2 must be that call's shape): what BuildPost (buildpostflat.cpp:18-106) resp. CalcPosteriorFlat3
 * (buildposterior3flat.cpp:19-85) left for CalcAlnFlat. */
int mpcgpu_get_last_post(mpcgpu_ctx *ctx, uint32_t C1, uint32_t C2, float *post);

/* The MSA x MSA join of PProg (pprog2.cpp:7-56 -> PProg::AlignMSAsFlat, alnmsasflat.cpp:4-50) for an
 * explicit list of cross pairs (getpairs.cpp:33-69 samples at most 2000): per pair
 * PProg::GetPostPairsAlignedFlat (getpostpairsalignedflat.cpp:5-98) = CalcPost + MySparseMx::FromPost
 * + the alignment score (EA = Score / min(L1,L2)), then CalcPosteriorFlat3
 * (buildposterior3flat.cpp:19-85, contributions in pair-list order) and CalcAlnFlat + TraceBackFlat.
 * Sequences are indices into the set given to mpcgpu_set_seqs_registry (the reference's global input
 * registry, globalinputms.cpp:11-143: any sequence may pair with any other, no all-pairs tables);
 * pair q aligns seq1[q] (a row of MSA1, X) with seq2[q] (a row of MSA2, Y); pos2col1 / pos2col2 hold
 * the position->column map of seq1[q] / seq2[q] for q = 0..npairs-1, concatenated. ea_out (may be
 * NULL) receives EA per pair. */
int mpcgpu_set_seqs_registry(mpcgpu_ctx *ctx, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens);
int mpcgpu_align_msas(mpcgpu_ctx *ctx, uint32_t npairs, const uint32_t *seq1, const uint32_t *seq2, uint32_t C1,
                      uint32_t C2, const uint32_t *pos2col1, const uint32_t *pos2col2, char *path, uint32_t *pathlen,
                      float *score, float *ea_out);

/* AlignPairFlat (alignpairflat.cpp:3-27; callers uclust.cpp:14, transaln.cpp:787, eadistmx.cpp:54, eacluster.cpp:209) for a
 * LIST of pairs of registered sequences: CalcPost (calcpost.cpp:4-36: fwd + bwd + CalcPostFlat), CalcAlnFlat + TraceBackFlat on the
 * dense thresholded posterior. paths: npairs slots of path_stride bytes (>= LX+LY of every pair), B/X/Y strings of pathlens[q]
 * characters; scores[q] = CalcAlnFlat's score, ea[q] = score / min(LX, LY) — what AlignPairFlat returns (either may be NULL).
 * The sparse matrices of the same pairs (AlignPairFlat_SparsePost) are then available through mpcgpu_get_list_sparse.
 * Like every stage-A call on a context, this one takes the context's scratch and INVALIDATES the all-pairs store the context may
 * hold (mpcgpu_build_store / mpcgpu_store_import): callers that interleave pair lists with a consistency run use a context of
 * their own for the lists (the drop-in's join contexts). A list is cut into chunks that one stage-A batch serves (halved as
 * often as needed); sequences beyond ~12 000 residues are refused (error, not a fallback). */
int mpcgpu_align_pairs(mpcgpu_ctx *ctx, uint32_t npairs, const uint32_t *seq1, const uint32_t *seq2, uint32_t path_stride,
                       char *paths, uint32_t *pathlens, float *scores, float *ea);
/* MySparseMx::FromPost (mysparsemx.cpp:115-152) of pair q of the LAST list stage on this context (mpcgpu_align_pairs of at most 256
 * pairs, mpcgpu_align_msas): nnz, offsets[LX+1], values (8 bytes per entry; capacity from a first call with values == NULL). */
int mpcgpu_get_list_sparse(mpcgpu_ctx *ctx, uint32_t q, uint32_t *nnz, uint32_t *offsets, void *values);

/* ---- several GPUs of one node inside ONE process (muscle_amd/csrc/mpcgpu_group.cpp) --------------------------
 * The drop-in binary is one process; it consumes the multi-GPU path through these calls. A group owns one context per
 * listed device (an ordinal may repeat: two contexts on one device is how the tests run this on a one-GPU box), shards
 * the pair loops of MPCFlat::CalcPosteriors (mpcflat.cpp:239-251) and MPCFlat::ConsIter (consflat.cpp:5-23) over them
 * (contiguous pair ranges balanced by DP cells) and performs the two exchanges itself: the all-gather of the packed
 * sparse posteriors before relax and the all-gather of the new probabilities after each iteration — RCCL point-to-point
 * sends/receives over xGMI with exact sizes when the devices are distinct and librccl loads (transport "rccl"), peer
 * copies otherwise ("peer"; MPCGPU_GROUP_TRANSPORT=peer|rccl forces one). After group_calc_posteriors every context
 * holds the full store, so the host reads results (EA, matrices, mpcgpu_align_alns ...) from mpcgpu_group_ctx(g, 0). */
typedef struct mpcgpu_group mpcgpu_group;
int mpcgpu_group_create(mpcgpu_group **out, uint32_t ndev, const int *device_ordinals);
void mpcgpu_group_destroy(mpcgpu_group *g);
const char *mpcgpu_group_last_error(const mpcgpu_group *g); /* g may be NULL: last create() error */
uint32_t mpcgpu_group_size(const mpcgpu_group *g);
mpcgpu_ctx *mpcgpu_group_ctx(mpcgpu_group *g, uint32_t rank);
const char *mpcgpu_group_transport(const mpcgpu_group *g); /* "rccl" or "peer" */
/* mpcgpu_set_hmm / mpcgpu_set_seqs / mpcgpu_set_mega on every context */
int mpcgpu_group_set_hmm(mpcgpu_group *g, const float start[5], const float trans[25], const float match[256 * 256],
                         const float ins[256], float min_sparse_score, int expf_variant);
int mpcgpu_group_set_seqs(mpcgpu_group *g, uint32_t n, const uint8_t *const *seqs, const uint32_t *lens);
int mpcgpu_group_set_mega(mpcgpu_group *g, uint32_t nfeat, const uint32_t *alpha, const float *weight,
                          const float *const *logprobs, const float *const *logprob_mx, const uint8_t *const *profiles);
/* MPCFlat::CalcPosteriors for all pairs: stage A sharded, all-gather, store built on every device. */
int mpcgpu_group_calc_posteriors(mpcgpu_group *g);
/* One MPCFlat::ConsIter: relax sharded, all-gather of the values, commit on every device. */
int mpcgpu_group_cons_iter(mpcgpu_group *g);

/* ---- measurement hooks (bench.py) --------------------------------------------------------- */
/* Kernel time in ms measured with hipEvents on the library's own stream, accumulated since the
 * last reset, per kernel family: 0 = fwd/bwd (fb), 1 = posterior finish (sort/EA/pack),
 * 2 = store build, 3 = relax, 4 = commit/scatter; and launches counted per family. */
#define MPCGPU_NKERNELS 9 /* ... 5 = BuildPost record generation, 6 = BuildPost grouping (sort), 7 = BuildPost in-order reduction,
                             8 = CalcAlnFlat + traceback (families 5-8: mpcgpu_align_alns / mpcgpu_align_msas / mpcgpu_calc_aln) */
int mpcgpu_timers_reset(mpcgpu_ctx *ctx);
/* Timing on (default) / off. Off: no hipEvents around the launches (two events per launch family and call are a measurable share
 * of a small join: the drop-in switches them off unless MUSCLE_GPU_TIMING asks for the report); mpcgpu_timers_get then reports
 * what was measured while it was on. */
int mpcgpu_timers_enable(mpcgpu_ctx *ctx, int on);
int mpcgpu_timers_get(mpcgpu_ctx *ctx, float ms[MPCGPU_NKERNELS], uint64_t launches[MPCGPU_NKERNELS]);
/* Algorithmic work of the last calc_posteriors call: DP cells summed over pairs
 * (sum (LX+1)(LY+1)) and of the last cons_iter: (pair,Z) triples and stored entries. */
int mpcgpu_work_get(mpcgpu_ctx *ctx, uint64_t *dp_cells, uint64_t *relax_entry_z, uint64_t *store_entries);
/* How the last stage A (mpcgpu_calc_posteriors / a pair-list call) ran its forward/backward sweeps: pairs in all, pairs that
 * ran as members of chains (consecutive pairs with the same row sequence swept back to back by one wavefront, no systolic
 * fill/drain in between: kernels_fbc.h), and the number of chains. Same cells, same results; MPCGPU_FB_CHAIN=0 turns chains off. */
int mpcgpu_stage_a_info(mpcgpu_ctx *ctx, uint64_t *pairs, uint64_t *chained_pairs, uint64_t *chains);
/* Sizes of the current store, for the measurement's lower bounds (bench.py: min_bytes_per_launch): out[0] bytes of the row-indexed
 * block records, out[1] of the window records (0: not built), out[2] of the packed matrices of all pairs, out[3] stored posteriors
 * of all pairs, out[4] of the pairs this context relaxes, out[5] sequences whose records the store holds (all of them unless the
 * store is partial: mpcgpu_store_import_part). */
int mpcgpu_store_info(mpcgpu_ctx *ctx, uint64_t out[6]);
int mpcgpu_synchronize(mpcgpu_ctx *ctx);
/* Which store layout and relax kernel the current store uses, in words (record sizes, workgroup geometry, the tile shapes
 * once a relax iteration has built them), and whether that is a FALLBACK this build chose because the run exceeds the default
 * layout's limits (sequences longer than 4095, records beyond the CU's LDS ...): bench.py prints it, the drop-in warns. */
int mpcgpu_relax_info(mpcgpu_ctx *ctx, char *buf, uint32_t buflen, int *is_fallback);

#ifdef __cplusplus
}
#endif
#endif /* MPCGPU_H */
