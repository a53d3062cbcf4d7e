// TEST INFRASTRUCTURE ONLY — never linked into the product library.
//
// C-callable wrappers around the *reference's own* compiled functions (rcedgar/muscle, built
// from /root/reference/src by oracle/build_ref.sh into oracle/_ref/libmuscle_ref.so).
// Used (a) to pin oracle/mpc_oracle.c against the real reference, (b) to generate the golden
// fixtures under tests/golden/ (tests/golden/make_golden.py), (c) optionally as bench.py's
// cpu_baseline kind="reference".  This file contains NO restated numerics: every number comes
// out of a reference function, cited below.
#include "muscle.h"      // reference header (-I/root/reference/src)
#include "mpcflat.h"
#include "pprog.h"
#include "pairhmm.h"
#include "hmmparams.h"
#include "mega.h"
#include <cstring>

string g_Arg1; // normally defined in main.cpp:4 (main.o is not linked)

float CalcAlnScoreFlat(const float *Post, uint LX, uint LY, float *DPRows); // calcalnscoreflat.cpp:4
void CalcPostFlat(const float *FlatFwd, const float *FlatBwd, uint LX, uint LY, float *Post); // calcposteriorflat.cpp:4
void CalcPosteriorFlat3(const MultiSequence &MSA1, const MultiSequence &MSA2, const vector<uint> &SeqIndexes1,
  const vector<uint> &SeqIndexes2, const vector<MySparseMx *> &SparseMxs, float *Flat); // buildposterior3flat.cpp:19

// AlignPairFlat_SparsePost (alignpairflat.cpp:3-21) on sequences i, j of the global input of ref_mpc_begin (labels s<i>): the
// path of CalcAlnFlat on the dense posterior, EA = Score / min(L1, L2), and the FromPost matrix.
float AlignPairFlat_SparsePost(const string &Label1, const string &Label2, string &Path, MySparseMx *SparsePost);

extern "C" {

// HMMParams::FromDefaults/PerturbProbs/ToPairHMM (hmmparams.cpp:273,298; perturbhmm.cpp:15)
// -> fills the process-global PairHMM tables (pairhmm.h:23-29).
void ref_init_hmm(int nucleo, unsigned perturb_seed)
	{
	opt_quiet = true; optset_quiet = true;
	HMMParams HP;
	HP.FromDefaults(nucleo != 0);
	if (perturb_seed != 0)
		HP.PerturbProbs(perturb_seed);
	HP.ToPairHMM();
	}

void ref_set_threads(unsigned n)
	{
	opt_threads = n; optset_threads = true;
	}

// Copies of the PairHMM statics (pairhmm.h:23-29): start[5], trans[25], match[256*256], ins[256]
void ref_get_hmm(float *start, float *trans, float *match, float *ins)
	{
	memcpy(start, PairHMM::m_StartScore, sizeof(float)*5);
	memcpy(trans, PairHMM::m_TransScore, sizeof(float)*25);
	memcpy(match, PairHMM::m_MatchScore, sizeof(float)*256*256);
	memcpy(ins, PairHMM::m_InsScore, sizeof(float)*256);
	}

float ref_min_sparse_score() { return MIN_SPARSE_SCORE; } // mysparsemx.h:4 (host logf)

// fwdflat3.cpp:12 / bwdflat3.cpp:10 ; Flat has 5*(LX+1)*(LY+1) floats
void ref_fwd(const byte *X, uint LX, const byte *Y, uint LY, float *Flat) { CalcFwdFlat(X, LX, Y, LY, Flat); }
void ref_bwd(const byte *X, uint LX, const byte *Y, uint LY, float *Flat) { CalcBwdFlat(X, LX, Y, LY, Flat); }
// totalprobflat.cpp:3
float ref_total(const float *Fwd, const float *Bwd, uint LX, uint LY) { return CalcTotalProbFlat(Fwd, Bwd, LX, LY); }
// calcposteriorflat.cpp:4 ; Post has LX*LY floats
void ref_post(const float *Fwd, const float *Bwd, uint LX, uint LY, float *Post) { CalcPostFlat(Fwd, Bwd, LX, LY, Post); }
// calcalnscoreflat.cpp:4
float ref_aln_score(const float *Post, uint LX, uint LY)
	{
	float *DPRows = AllocDPRows(LX, LY);
	float s = CalcAlnScoreFlat(Post, LX, LY, DPRows);
	myfree(DPRows);
	return s;
	}
// calcalnflat.cpp:6 + tracebackflat.cpp:3 ; path must hold LX+LY+1 chars; returns score
float ref_calc_aln(const float *Post, uint LX, uint LY, char *path, uint *pathlen)
	{
	float *DPRows = AllocDPRows(LX, LY);
	char *TB = AllocTB(LX, LY);
	string Path;
	float s = CalcAlnFlat(Post, LX, LY, DPRows, TB, Path);
	myfree(DPRows); myfree(TB);
	memcpy(path, Path.c_str(), Path.size());
	*pathlen = (uint) Path.size();
	return s;
	}

// MySparseMx::FromPost (mysparsemx.cpp:115). offsets: LX+1 uints; values: LX*LY*8 bytes max.
uint ref_sparse_from_post(const float *Post, uint LX, uint LY, uint *offsets, byte *values)
	{
	MySparseMx M;
	M.FromPost(Post, LX, LY);
	memcpy(offsets, M.m_Offsets, sizeof(uint)*(LX+1));
	memcpy(values, M.m_ValueVec, 8*(size_t)M.m_VecSize);
	return M.m_VecSize;
	}

// ---------------------------------------------------------------------------------------
// Whole stage through the reference's own orchestrator. ONE call per process: the global
// input registry can only be set once (globalinputms.cpp:70 asserta(g_GlobalMS == 0)).
// Follows MPCFlat::Run_Super4 (mpcflat.cpp:266-283) minus CalcGuideTree:
//   AllocPairCount, InitSeqs, InitPairs, InitDistMx, CalcPosteriors, ConsIter x iters.
// Snapshots of the sparse store are copied out after CalcPosteriors (stage 0) and after each
// ConsIter (stage 1..iters) through the callback-free "query" functions below.
static MPCFlat *g_M = 0;
static MultiSequence *g_MS = 0;

int ref_mpc_begin(uint n, const char **seqs, uint threads)
	{
	if (g_M != 0)
		return -1;
	opt_quiet = true; optset_quiet = true;
	if (threads > 0)
		ref_set_threads(threads);
	vector<string> Labels, Seqs;
	for (uint i = 0; i < n; ++i)
		{
		char tmp[32];
		snprintf(tmp, sizeof(tmp), "s%u", i);
		Labels.push_back(tmp);
		Seqs.push_back(seqs[i]);
		}
	g_MS = new MultiSequence;
	g_MS->FromStrings(Labels, Seqs);
	SetGlobalInputMS(*g_MS);
	g_M = new MPCFlat;
	MPCFlat &M = *g_M;
	M.Clear();
	M.AllocPairCount((n*(n-1))/2);
	M.InitSeqs(g_MS);
	M.InitPairs();
	M.InitDistMx();
	M.m_Weights.assign(n, 1.0f); // read (then overwritten by 1.0f) at conspairflat.cpp:41-42
	return 0;
	}

// The same, for a .mega input (structure profiles): Mega::FromFile (mega.cpp:119-270) + the label/sequence
// registry LoadInput builds from it (loadinput.cpp:5-9); CalcPost then takes the profile branch
// (calcpost.cpp:14-22). ONE call per process (mega.cpp:125-127 asserts an empty state).
int ref_mpc_begin_mega(const char *path, uint threads)
	{
	if (g_M != 0)
		return -1;
	opt_quiet = true; optset_quiet = true;
	if (threads > 0)
		ref_set_threads(threads);
	Mega::FromFile(path);
	g_MS = new MultiSequence;
	g_MS->FromStrings(Mega::m_Labels, Mega::m_Seqs);
	SetGlobalInputMS(*g_MS);
	const uint n = g_MS->GetSeqCount();
	g_M = new MPCFlat;
	MPCFlat &M = *g_M;
	M.Clear();
	M.AllocPairCount((n*(n-1))/2);
	M.InitSeqs(g_MS);
	M.InitPairs();
	M.InitDistMx();
	M.m_Weights.assign(n, 1.0f);
	return 0;
	}

// Copies of the Mega statics (mega.h:9-33)
uint ref_mega_feature_count() { return Mega::m_FeatureCount; }
uint ref_mega_profile_count() { return Mega::GetProfileCount(); }
uint ref_mega_alpha(uint f) { return Mega::GetAlphaSize(f); }
float ref_mega_weight(uint f) { return Mega::GetWeight(f); }
void ref_mega_logprobs(uint f, float *out)
	{
	const vector<float> &v = Mega::m_LogProbsVec[f];
	memcpy(out, v.data(), sizeof(float)*v.size());
	}
void ref_mega_logprobmx(uint f, float *out) // A x A row-major
	{
	const vector<vector<float> > &m = Mega::m_LogProbMxVec[f];
	const uint A = SIZE(m);
	for (uint a = 0; a < A; ++a)
		memcpy(out + (size_t) a*A, m[a].data(), sizeof(float)*A);
	}
uint ref_mega_length(uint idx) { return SIZE(Mega::GetProfile(idx)); }
void ref_mega_profile(uint idx, byte *out) // position-major, feature_count letters per position
	{
	const vector<vector<byte> > &P = Mega::GetProfile(idx);
	const uint F = Mega::m_FeatureCount;
	for (uint Pos = 0; Pos < SIZE(P); ++Pos)
		for (uint f = 0; f < F; ++f)
			out[(size_t) Pos*F + f] = P[Pos][f];
	}
void ref_mega_seq(uint idx, char *out) { memcpy(out, Mega::m_Seqs[idx].data(), Mega::m_Seqs[idx].size()); }
// Mega::GetInsScore / GetMatchScore (mega.cpp:273, :341) on loaded profiles
float ref_mega_ins(uint idx, uint pos) { return Mega::GetInsScore(Mega::GetProfile(idx), pos); }
float ref_mega_match(uint idx1, uint pos1, uint idx2, uint pos2)
	{
	return Mega::GetMatchScore(Mega::GetProfile(idx1), pos1, Mega::GetProfile(idx2), pos2);
	}
// Mega::CalcFwdFlat_mega / CalcBwdFlat_mega (fwdflat_mega.cpp:14, bwdflat_mega.cpp:13)
void ref_mega_fwd(uint idx1, uint idx2, float *Flat) { Mega::CalcFwdFlat_mega(Mega::GetProfile(idx1), Mega::GetProfile(idx2), Flat); }
void ref_mega_bwd(uint idx1, uint idx2, float *Flat) { Mega::CalcBwdFlat_mega(Mega::GetProfile(idx1), Mega::GetProfile(idx2), Flat); }

void ref_mpc_calc_posteriors() { g_M->CalcPosteriors(); }     // mpcflat.cpp:214
void ref_mpc_cons_iter(uint iter) { g_M->ConsIter(iter); }     // consflat.cpp:5
// MPCFlat::ConsPair (conspairflat.cpp:10-110) of SOME pairs of the current iteration: reads *m_ptrSparsePosts of all pairs, writes the
// listed pairs' matrices of *m_ptrUpdatedSparsePosts; NO swap (consflat.cpp:22) — the store stays at the stage it was. For sampled pins
// of stores whose full ConsIter takes days (rdrp, N = 1000: tests/golden/make_golden.py big-sampled). OpenMP over the list as
// consflat.cpp:13-20 runs it over all pairs.
void ref_mpc_cons_pairs(const uint *ks, uint count)
	{
	const unsigned ThreadCount = GetRequestedThreadCount();
#pragma omp parallel for num_threads(ThreadCount) schedule(dynamic, 1)
	for (int i = 0; i < (int) count; ++i)
		g_M->ConsPair(ks[i]);
	}
// the buffer swap of consflat.cpp:22 alone: what ref_mpc_cons_pairs wrote becomes the current store. For a STAGE-2 pin of a few pairs of
// a store whose full ConsIter takes days: after ConsPair of iteration 1 for every pair that touches a small clique of sequences, the
// swapped store holds the stage-1 matrices of exactly the pairs a ConsPair of two clique members reads (conspairflat.cpp:49-89: (X,Z)
// and (Y,Z) for all Z) — the other pairs' objects are never looked at (tests/golden/make_golden.py big-stage2).
void ref_mpc_swap_stores() { std::swap(g_M->m_ptrSparsePosts, g_M->m_ptrUpdatedSparsePosts); }
uint ref_mpc_updated_nnz(uint k) { const MySparseMx &S = g_M->GetUpdatedSparsePost(k); return S.m_Offsets[S.m_LX]; }
void ref_mpc_updated_sparse(uint k, uint *offsets, byte *values)
	{
	const MySparseMx &S = g_M->GetUpdatedSparsePost(k);
	memcpy(offsets, S.m_Offsets, sizeof(uint)*(S.m_LX+1));
	memcpy(values, S.m_ValueVec, 8*(size_t)S.m_Offsets[S.m_LX]);
	}
uint ref_mpc_pair_count() { return (uint) g_M->m_Pairs.size(); }
void ref_mpc_pair(uint k, uint *i, uint *j) { *i = g_M->m_Pairs[k].first; *j = g_M->m_Pairs[k].second; }
float ref_mpc_ea(uint i, uint j) { return g_M->m_DistMx[i][j]; }
// NB: MySparseMx::UpdateFromPost (mysparsemx.cpp:87-113) never sets m_VecSize on the updated matrix,
// so the entry count is read from m_Offsets[m_LX] instead.
uint ref_mpc_nnz(uint k) { const MySparseMx &S = g_M->GetSparsePost(k); return S.m_Offsets[S.m_LX]; }
void ref_mpc_sparse(uint k, uint *offsets, byte *values)
	{
	const MySparseMx &S = g_M->GetSparsePost(k);
	memcpy(offsets, S.m_Offsets, sizeof(uint)*(S.m_LX+1));
	memcpy(values, S.m_ValueVec, 8*(size_t)S.m_Offsets[S.m_LX]);
	}


// ---- alignments of alignments on the store of g_M (after ref_mpc_calc_posteriors / ref_mpc_cons_iter) --------------------
// rows = gapped rows of one alignment (all of one width), idx = input sequence index of each row (its label is s<idx>).
static MultiSequence *MakeMSA(uint n, const char **rows, const uint *idx)
	{
	vector<string> Labels, Seqs;
	for (uint i = 0; i < n; ++i)
		{
		char tmp[32];
		snprintf(tmp, sizeof(tmp), "s%u", idx[i]);
		Labels.push_back(tmp);
		Seqs.push_back(rows[i]);
		}
	MultiSequence *MSA = new MultiSequence;
	MSA->FromStrings(Labels, Seqs);
	return MSA;
	}

// MPCFlat::BuildPost (buildpostflat.cpp:18-106). weights: what m_Weights holds (indexed by the ROW number inside each
// alignment at buildpostflat.cpp:42,52), n_weights entries; NULL = all 1.0f (mpcflat.cpp:324). post: C1 x C2 floats.
int ref_mpc_build_post(uint n1, const char **rows1, const uint *idx1, uint n2, const char **rows2, const uint *idx2,
  const float *weights, uint n_weights, float *post)
	{
	if (g_M == 0)
		return -1;
	MultiSequence *MSA1 = MakeMSA(n1, rows1, idx1);
	MultiSequence *MSA2 = MakeMSA(n2, rows2, idx2);
	const uint SeqCount = g_M->GetSeqCount();
	g_M->m_Weights.assign(SeqCount, 1.0f);
	if (weights != 0)
		for (uint i = 0; i < n_weights && i < SeqCount; ++i)
			g_M->m_Weights[i] = weights[i];
	g_M->BuildPost(*MSA1, *MSA2, post);
	g_M->m_Weights.assign(SeqCount, 1.0f);
	delete MSA1;
	delete MSA2;
	return 0;
	}

// MPCFlat::AlignAlns (alnalnsflat.cpp:7-52): BuildPost + CalcAlnFlat; path gets the B/X/Y string (capacity C1+C2)
int ref_mpc_align_alns(uint n1, const char **rows1, const uint *idx1, uint n2, const char **rows2, const uint *idx2,
  char *path, uint *pathlen, float *score)
	{
	if (g_M == 0)
		return -1;
	MultiSequence *MSA1 = MakeMSA(n1, rows1, idx1);
	MultiSequence *MSA2 = MakeMSA(n2, rows2, idx2);
	const uint C1 = MSA1->GetColCount(), C2 = MSA2->GetColCount();
	g_M->m_Weights.assign(g_M->GetSeqCount(), 1.0f);
	float *Post = AllocPost(C1, C2);
	g_M->BuildPost(*MSA1, *MSA2, Post);
	float *DPRows = AllocDPRows(C1, C2);
	char *TB = AllocTB(C1, C2);
	string Path;
	*score = CalcAlnFlat(Post, C1, C2, DPRows, TB, Path);
	memcpy(path, Path.data(), Path.size());
	*pathlen = (uint) Path.size();
	myfree(Post); myfree(DPRows); myfree(TB);
	delete MSA1;
	delete MSA2;
	return 0;
	}

// The pieces of PProg::AlignMSAsFlat (alnmsasflat.cpp:4-50) on an explicit pair list (seq1[k] = row of MSA1, seq2[k] = row
// of MSA2): GetPostPairsAlignedFlat (getpostpairsalignedflat.cpp:5-98: fresh posteriors per pair, no consistency) ->
// CalcPosteriorFlat3 (buildposterior3flat.cpp:19-85) -> CalcAlnFlat. Needs the global input of ref_mpc_begin (labels s<i>).
// post: C1 x C2 floats out; ea_avg: the function's return value (sum in pair order when run with one thread).
int ref_align_msas(uint n1, const char **rows1, const uint *idx1, uint n2, const char **rows2, const uint *idx2,
  uint npairs, const uint *seq1, const uint *seq2, float *post, char *path, uint *pathlen, float *ea_avg)
	{
	if (g_MS == 0)
		return -1;
	MultiSequence *MSA1 = MakeMSA(n1, rows1, idx1);
	MultiSequence *MSA2 = MakeMSA(n2, rows2, idx2);
	const uint C1 = MSA1->GetColCount(), C2 = MSA2->GetColCount();
	vector<uint> S1(seq1, seq1 + npairs), S2(seq2, seq2 + npairs);
	vector<MySparseMx *> SparseMxs;
	PProg PP;
	*ea_avg = PP.GetPostPairsAlignedFlat("golden", *MSA1, *MSA2, S1, S2, SparseMxs);
	CalcPosteriorFlat3(*MSA1, *MSA2, S1, S2, SparseMxs, post);
	for (uint i = 0; i < npairs; ++i)
		delete SparseMxs[i];
	float *DPRows = AllocDPRows(C1, C2);
	char *TB = AllocTB(C1, C2);
	string Path;
	CalcAlnFlat(post, C1, C2, DPRows, TB, Path);
	memcpy(path, Path.data(), Path.size());
	*pathlen = (uint) Path.size();
	myfree(DPRows); myfree(TB);
	delete MSA1;
	delete MSA2;
	return 0;
	}


int ref_align_pair(uint i, uint j, char *path, uint *pathlen, float *ea, uint *nnz, uint *offsets, byte *values, uint values_cap)
	{
	if (g_MS == 0)
		return -1;
	char l1[32], l2[32];
	snprintf(l1, sizeof(l1), "s%u", i);
	snprintf(l2, sizeof(l2), "s%u", j);
	string Path;
	MySparseMx S;
	*ea = AlignPairFlat_SparsePost(l1, l2, Path, &S);
	memcpy(path, Path.data(), Path.size());
	*pathlen = (uint) Path.size();
	const uint n = S.m_Offsets[S.m_LX];
	*nnz = n;
	if (offsets != 0)
		memcpy(offsets, S.m_Offsets, sizeof(uint)*(S.m_LX+1));
	if (values != 0 && n <= values_cap)
		memcpy(values, S.m_ValueVec, 8*(size_t)n);
	return 0;
	}

} // extern "C"
