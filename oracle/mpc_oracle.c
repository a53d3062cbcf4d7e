/*
 * oracle/mpc_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar float32) of the MUSCLE5 MPCFlat all-pairs posterior stage.
 * It is the checker for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it. The product library (muscle_amd/csrc) never links, calls or
 * falls back to anything in this directory.
 *
 * Parity status: PINNED. tests/test_oracle_vs_ref.py checks every function below bit-for-bit
 * against the compiled reference (oracle/_ref/libmuscle_ref.so, built from /root/reference/src by
 * oracle/build_ref.sh) when that library is present, and tests/test_oracle_golden.py checks it
 * against the committed fixtures in tests/golden/ (generated from the compiled reference by
 * tests/golden/make_golden.py) everywhere else. The reference's own tests hold no golden vectors
 * for this path, so "outputs of the reference itself run here" is the anchor.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fopenmp (oracle/Makefile). -ffp-contract=off and
 * no -march=native / -ffast-math: the reference build has no FMA, every float op rounds on its own.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/src).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned int uint;
typedef unsigned char byte;
static inline size_t nz1(size_t n) { return n ? n : 1; }

/* PairHMM tables (pairhmm.h:23-29), as filled by HMMParams::ToPairHMM (hmmparams.cpp:298-409). */
typedef struct {
	float start[5];        /* m_StartScore[HMMSTATE_*]   */
	float trans[5][5];     /* m_TransScore               */
	float match[256][256]; /* m_MatchScore               */
	float ins[256];        /* m_InsScore                 */
} orc_hmm;

/* State order pairhmm.h:11-19 */
enum { S_M = 0, S_IX = 1, S_IY = 2, S_JX = 3, S_JY = 4, NS = 5 };

/* scoretype.h:89-98 */
#define LOG_ZERO (-2e20f)
#define LOG_UNDERFLOW_THRESHOLD 7.5f

/* scoretype.h:100-109: log(exp(x)+1) on [0,7.5], four cubics in Horner form */
static inline float logexp1(float x)
{
	if (x <= 1.00f)
		return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x + 0.693203116424741f;
	if (x <= 2.50f)
		return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x + 0.692140569840976f;
	if (x <= 4.50f)
		return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x + 0.514272634594009f;
	return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x + 0.168037164329057f;
}

/* scoretype.h:119-124 (and LOG_PLUS_EQUALS :111-117, same expression) */
static inline float la2(float x, float y)
{
	if (x < y)
		return (x == LOG_ZERO || y - x >= LOG_UNDERFLOW_THRESHOLD) ? y : logexp1(y - x) + x;
	return (y == LOG_ZERO || x - y >= LOG_UNDERFLOW_THRESHOLD) ? x : logexp1(x - y) + y;
}

/* scoretype.h:136-139: 5-ary form nests to the right */
static inline float la5(float a, float b, float c, float d, float e)
{
	return la2(a, la2(b, la2(c, la2(d, e))));
}

float orc_log_add(float x, float y) { return la2(x, y); }

/* flatmx.h:11-15 */
#define FIX(s, i, j, LY) (NS * ((size_t)(i) * ((LY) + 1) + (j)) + (s))

/* hmmscores.h:1-16 */
#define BIND_T(h)                                       \
	const float tSM = (h)->start[S_M];                  \
	const float tSI = (h)->start[S_IX];                 \
	const float tSJ = (h)->start[S_JX];                 \
	const float tMM = (h)->trans[S_M][S_M];             \
	const float tMI = (h)->trans[S_M][S_IX];            \
	const float tMJ = (h)->trans[S_M][S_JX];            \
	const float tII = (h)->trans[S_IX][S_IX];           \
	const float tIM = (h)->trans[S_IX][S_M];            \
	const float tJJ = (h)->trans[S_JX][S_JX];           \
	const float tJM = (h)->trans[S_JX][S_M];

/*
 * Emission scores. Two sources, selected exactly where the reference selects them (calcpost.cpp:14-29):
 *  - byte sequences: PairHMM::m_InsScore[c] / m_MatchScore[x][y] (fwdflat3.cpp:102-109);
 *  - "mega" structure profiles (a .mega input, loadinput.cpp:5-9): per position one letter for each of
 *    nfeat features; Mega::GetInsScore (mega.cpp:273-285) = sum over features, in feature order and
 *    starting from 0, of LogProbs_f[letter_f] * Weight_f; Mega::GetMatchScore (mega.cpp:341-363) = the
 *    same fold over LogProbMx_f[letterX_f][letterY_f] * Weight_f.
 */
typedef struct {
	uint nfeat;
	const uint *alpha;   /* [nfeat] alphabet size of each feature (Mega::m_AlphaSizes) */
	const float *weight; /* [nfeat] Mega::m_Weights */
	const float *lp;     /* Mega::m_LogProbsVec, features back to back */
	const uint *lp_off;  /* [nfeat] start of feature f in lp */
	const float *mx;     /* Mega::m_LogProbMxVec, A_f x A_f row-major, features back to back */
	const uint *mx_off;  /* [nfeat] start of feature f in mx */
} orc_mega;

typedef struct {
	const orc_hmm *h;
	const byte *X, *Y;   /* byte sequences, or (mega) profiles: position-major, nfeat letters per position */
	const orc_mega *g;   /* NULL: byte sequences */
} emit_t;

/* mega.cpp:273-285 */
static float mega_ins(const orc_mega *g, const byte *prof, uint pos)
{
	float score = 0;
	for (uint f = 0; f < g->nfeat; ++f)
		score += g->lp[g->lp_off[f] + prof[(size_t)pos * g->nfeat + f]] * g->weight[f];
	return score;
}

/* mega.cpp:341-363 */
static float mega_match(const orc_mega *g, const byte *px, uint posx, const byte *py, uint posy)
{
	float score = 0;
	for (uint f = 0; f < g->nfeat; ++f) {
		const uint lx = px[(size_t)posx * g->nfeat + f], ly = py[(size_t)posy * g->nfeat + f];
		score += g->mx[g->mx_off[f] + (size_t)lx * g->alpha[f] + ly] * g->weight[f];
	}
	return score;
}

float orc_mega_ins(const orc_mega *g, const byte *prof, uint pos) { return mega_ins(g, prof, pos); }
float orc_mega_match(const orc_mega *g, const byte *px, uint posx, const byte *py, uint posy) { return mega_match(g, px, posx, py, posy); }

/* 0-based positions. "past the end" (bwdflat3.cpp:46,64 use the letter 0; bwdflat_mega.cpp:55,78-80 use a 0 score):
 * the value never reaches a result (it is only ever added to LOG_ZERO), kept per source anyway. */
static inline float e_ins_x(const emit_t *e, uint i, uint LX)
{
	if (e->g) return i >= LX ? 0.0f : mega_ins(e->g, e->X, i);
	return e->h->ins[i >= LX ? 0 : e->X[i]];
}
static inline float e_ins_y(const emit_t *e, uint j, uint LY)
{
	if (e->g) return j >= LY ? 0.0f : mega_ins(e->g, e->Y, j);
	return e->h->ins[j >= LY ? 0 : e->Y[j]];
}
static inline float e_match(const emit_t *e, uint i, uint LX, uint j, uint LY)
{
	if (e->g) return (i >= LX || j >= LY) ? 0.0f : mega_match(e->g, e->X, i, e->Y, j);
	return e->h->match[i >= LX ? 0 : e->X[i]][j >= LY ? 0 : e->Y[j]];
}

/*
 * Forward, fwdflat3.cpp:12-153 (byte sequences) / fwdflat_mega.cpp:14-165 (profiles: the same statements
 * with the emission lookups replaced, fwdflat_mega.cpp:29-31,80,95,113,120-121). F is the flat
 * (LX+1)(LY+1)x5 array. Border initialisation :35-93, interior :100-152. The reference indexes the
 * tables with a (signed) char; inputs here are 7-bit ASCII so byte indexing is identical.
 */
static void fwd_any(const emit_t *e, uint LX, uint LY, float *F)
{
	const orc_hmm *h = e->h;
	BIND_T(h)
	/* (0,0): all five states log-zero (:35-39) */
	for (int s = 0; s < NS; ++s)
		F[FIX(s, 0, 0, LY)] = LOG_ZERO;
	/* column 0 (:48-55, :42-43, :67-79) */
	for (uint i = 1; i <= LX; ++i) {
		float ex = e_ins_x(e, i - 1, LX);
		F[FIX(S_M, i, 0, LY)] = LOG_ZERO;
		F[FIX(S_IY, i, 0, LY)] = LOG_ZERO;
		F[FIX(S_JY, i, 0, LY)] = LOG_ZERO;
		if (i == 1) {
			F[FIX(S_IX, 1, 0, LY)] = tSI + ex;
			F[FIX(S_JX, 1, 0, LY)] = tSJ + ex;
		} else {
			F[FIX(S_IX, i, 0, LY)] = F[FIX(S_IX, i - 1, 0, LY)] + tII + ex;
			F[FIX(S_JX, i, 0, LY)] = F[FIX(S_JX, i - 1, 0, LY)] + tJJ + ex;
		}
	}
	/* row 0 (:57-65, :44-45, :81-93) */
	for (uint j = 1; j <= LY; ++j) {
		float ey = e_ins_y(e, j - 1, LY);
		F[FIX(S_M, 0, j, LY)] = LOG_ZERO;
		F[FIX(S_IX, 0, j, LY)] = LOG_ZERO;
		F[FIX(S_JX, 0, j, LY)] = LOG_ZERO;
		if (j == 1) {
			F[FIX(S_IY, 0, 1, LY)] = tSI + ey;
			F[FIX(S_JY, 0, 1, LY)] = tSJ + ey;
		} else {
			F[FIX(S_IY, 0, j, LY)] = F[FIX(S_IY, 0, j - 1, LY)] + tII + ey;
			F[FIX(S_JY, 0, j, LY)] = F[FIX(S_JY, 0, j - 1, LY)] + tJJ + ey;
		}
	}
	/* interior (:100-152) */
	for (uint i = 1; i <= LX; ++i) {
		const float ex = e_ins_x(e, i - 1, LX);
		for (uint j = 1; j <= LY; ++j) {
			const float ey = e_ins_y(e, j - 1, LY);
			const float exy = e_match(e, i - 1, LX, j - 1, LY);
			const float *D = F + FIX(0, i - 1, j - 1, LY); /* diagonal predecessor */
			const float *U = F + FIX(0, i - 1, j, LY);     /* (i-1, j) */
			const float *L = F + FIX(0, i, j - 1, LY);     /* (i, j-1) */
			float *C = F + FIX(0, i, j, LY);
			if (i == 1 && j == 1)
				C[S_M] = tSM + exy;
			else
				C[S_M] = la5(D[S_M] + tMM, D[S_IX] + tIM, D[S_JX] + tJM, D[S_IY] + tIM, D[S_JY] + tJM) + exy;
			C[S_IX] = la2(U[S_IX] + tII, U[S_M] + tMI) + ex;
			C[S_JX] = la2(U[S_JX] + tJJ, U[S_M] + tMJ) + ex;
			C[S_IY] = la2(L[S_IY] + tII, L[S_M] + tMI) + ey;
			C[S_JY] = la2(L[S_JY] + tJJ, L[S_M] + tMJ) + ey;
		}
	}
}

/*
 * Backward, bwdflat3.cpp:10-184 / bwdflat_mega.cpp:13-193. Corner :53-61, interior :73-130, right
 * column :132-153, bottom row :155-176; all (i,j) in [0,LX]x[0,LY] are written.
 */
static void bwd_any(const emit_t *e, uint LX, uint LY, float *B)
{
	const orc_hmm *h = e->h;
	BIND_T(h)
	for (int i = (int)LX; i >= 0; --i) {
		const float ex = e_ins_x(e, (uint)i, LX); /* x_{i+1} */
		for (int j = (int)LY; j >= 0; --j) {
			float *C = B + FIX(0, i, j, LY);
			if (i == (int)LX && j == (int)LY) {
				C[S_M] = tSM;
				C[S_IX] = tSI;
				C[S_IY] = tSI;
				C[S_JX] = tSJ;
				C[S_JY] = tSJ;
				continue;
			}
			const float ey = e_ins_y(e, (uint)j, LY); /* y_{j+1} */
			if (i < (int)LX && j < (int)LY) {
				const float nM = B[FIX(S_M, i + 1, j + 1, LY)] + e_match(e, (uint)i, LX, (uint)j, LY);
				const float nIX = B[FIX(S_IX, i + 1, j, LY)] + ex;
				const float nJX = B[FIX(S_JX, i + 1, j, LY)] + ex;
				const float nIY = B[FIX(S_IY, i, j + 1, LY)] + ey;
				const float nJY = B[FIX(S_JY, i, j + 1, LY)] + ey;
				if (i > 0 && j > 0)
					C[S_M] = la5(tMM + nM, tMI + nIX, tMJ + nJX, tMI + nIY, tMJ + nJY);
				else
					C[S_M] = LOG_ZERO;
				if (i > 0) {
					C[S_IX] = la2(tII + nIX, tIM + nM);
					C[S_JX] = la2(tJJ + nJX, tJM + nM);
				} else {
					C[S_IX] = LOG_ZERO;
					C[S_JX] = LOG_ZERO;
				}
				if (j > 0) {
					C[S_IY] = la2(tII + nIY, tIM + nM);
					C[S_JY] = la2(tJJ + nJY, tJM + nM);
				} else {
					C[S_IY] = LOG_ZERO;
					C[S_JY] = LOG_ZERO;
				}
				continue;
			}
			if (i < (int)LX) { /* j == LY : right column (:132-153; IY/JY pre-set :25-31) */
				C[S_IY] = LOG_ZERO;
				C[S_JY] = LOG_ZERO;
				if (i > 0) {
					const float nIX = B[FIX(S_IX, i + 1, j, LY)] + ex;
					const float nJX = B[FIX(S_JX, i + 1, j, LY)] + ex;
					C[S_M] = la2(tMI + nIX, tMJ + nJX);
					C[S_IX] = tII + nIX;
					C[S_JX] = tJJ + nJX;
				} else {
					C[S_M] = LOG_ZERO;
					C[S_IX] = LOG_ZERO;
					C[S_JX] = LOG_ZERO;
				}
			} else { /* i == LX, j < LY : bottom row (:155-176; IX/JX pre-set :33-39) */
				C[S_IX] = LOG_ZERO;
				C[S_JX] = LOG_ZERO;
				const float nIY = B[FIX(S_IY, i, j + 1, LY)] + ey;
				const float nJY = B[FIX(S_JY, i, j + 1, LY)] + ey;
				if (j > 0) {
					C[S_M] = la2(tMI + nIY, tMJ + nJY);
					C[S_IY] = tII + nIY;
					C[S_JY] = tJJ + nJY;
				} else {
					C[S_M] = LOG_ZERO;
					C[S_IY] = LOG_ZERO;
					C[S_JY] = LOG_ZERO;
				}
			}
		}
	}
}

void orc_fwd(const orc_hmm *h, const byte *X, uint LX, const byte *Y, uint LY, float *F)
{
	const emit_t e = {h, X, Y, NULL};
	fwd_any(&e, LX, LY, F);
}
void orc_bwd(const orc_hmm *h, const byte *X, uint LX, const byte *Y, uint LY, float *B)
{
	const emit_t e = {h, X, Y, NULL};
	bwd_any(&e, LX, LY, B);
}
/* Mega::CalcFwdFlat_mega / CalcBwdFlat_mega (fwdflat_mega.cpp:14, bwdflat_mega.cpp:13); PX/PY are profiles */
void orc_fwd_mega(const orc_hmm *h, const orc_mega *g, const byte *PX, uint LX, const byte *PY, uint LY, float *F)
{
	const emit_t e = {h, PX, PY, g};
	fwd_any(&e, LX, LY, F);
}
void orc_bwd_mega(const orc_hmm *h, const orc_mega *g, const byte *PX, uint LX, const byte *PY, uint LY, float *B)
{
	const emit_t e = {h, PX, PY, g};
	bwd_any(&e, LX, LY, B);
}

/* totalprobflat.cpp:3-16: left fold over the five end states */
float orc_total(const float *F, const float *B, uint LX, uint LY)
{
	float sum = LOG_ZERO;
	for (int s = 0; s < NS; ++s)
		sum = la2(sum, F[FIX(s, LX, LY, LY)] + B[FIX(s, LX, LY, LY)]);
	return sum;
}

/* mysparsemx.h:3-4 */
#define MIN_SPARSE_PROB 0.01f
float orc_min_sparse_score(void) { return logf(MIN_SPARSE_PROB); }

/* calcposteriorflat.cpp:4-27: dense posterior LX x LY */
void orc_post(const float *F, const float *B, uint LX, uint LY, float *Post)
{
	const float total = orc_total(F, B, LX, LY);
	const float thr = orc_min_sparse_score();
	for (uint i = 0; i < LX; ++i)
		for (uint j = 0; j < LY; ++j) {
			const float sc = F[FIX(S_M, i + 1, j + 1, LY)] + B[FIX(S_M, i + 1, j + 1, LY)] - total;
			float p;
			if (sc < thr)
				p = 0;
			else
				p = (sc >= 0.0f) ? 1.0f : expf(sc);
			Post[(size_t)i * LY + j] = p;
		}
}

/* mysparsemx.cpp:115-152 (FromPost): CSR with {float P, uint col} 8-byte entries. Returns nnz. */
uint orc_sparse_from_post(const float *Post, uint LX, uint LY, uint *offsets, byte *values)
{
	uint n = 0;
	for (uint i = 0; i < LX; ++i) {
		offsets[i] = n;
		for (uint j = 0; j < LY; ++j) {
			const float p = Post[(size_t)i * LY + j];
			if (p >= MIN_SPARSE_PROB) {
				if (values) {
					memcpy(values + 8 * (size_t)n, &p, 4);
					memcpy(values + 8 * (size_t)n + 4, &j, 4);
				}
				++n;
			}
		}
	}
	offsets[LX] = n;
	return n;
}

/* best3.h:31-49 */
static inline float best3(float b, float x, float y)
{
	if (b >= x)
		return (b >= y) ? b : y;
	return (x >= y) ? x : y;
}

/* calcalnscoreflat.cpp:4-32: max-sum DP over the dense posterior, score only */
float orc_aln_score(const float *Post, uint LX, uint LY)
{
	float *row = (float *)calloc((size_t)LY + 1, sizeof(float));
	for (uint i = 1; i <= LX; ++i) {
		float diag = row[0]; /* S(i-1, j-1) */
		float left = 0;      /* S(i, j-1)   */
		row[0] = 0;
		for (uint j = 1; j <= LY; ++j) {
			const float up = row[j];
			const float v = best3(diag + Post[(size_t)(i - 1) * LY + (j - 1)], up, left);
			diag = up;
			left = v;
			row[j] = v;
		}
	}
	const float s = row[LY];
	free(row);
	return s;
}

/* calcposteriorflat.cpp:89: EA = Score/min(LX,LY) (uint -> float, IEEE divide) */
float orc_ea(float score, uint LX, uint LY)
{
	uint m = LX < LY ? LX : LY;
	return score / m;
}

/*
 * calcalnflat.cpp:6-46 + best3.h:5-28 + tracebackflat.cpp:3-37.
 * path receives the B/X/Y string (capacity >= LX+LY), *pathlen its length. Returns the score.
 */
float orc_calc_aln(const float *Post, uint LX, uint LY, char *path, uint *pathlen)
{
	const size_t W = (size_t)LY + 1;
	char *tb = (char *)malloc(((size_t)LX + 1) * W);
	float *prev = (float *)calloc(W, sizeof(float));
	float *cur = (float *)calloc(W, sizeof(float));
	for (uint j = 0; j <= LY; ++j)
		tb[j] = 'Y';
	for (uint i = 1; i <= LX; ++i) {
		tb[i * W] = 'X';
		cur[0] = 0;
		for (uint j = 1; j <= LY; ++j) {
			const float b = prev[j - 1] + Post[(size_t)(i - 1) * LY + (j - 1)];
			const float x = prev[j];
			const float y = cur[j - 1];
			float best;
			char c;
			if (b >= x) {
				if (b >= y) { best = b; c = 'B'; } else { best = y; c = 'Y'; }
			} else {
				if (x >= y) { best = x; c = 'X'; } else { best = y; c = 'Y'; }
			}
			cur[j] = best;
			tb[i * W + j] = c;
		}
		float *t = prev; prev = cur; cur = t;
	}
	const float score = prev[LY];
	/* traceback from (LX,LY) to (0,0), then reverse */
	uint n = 0;
	int i = (int)LX, j = (int)LY;
	while (i != 0 || j != 0) {
		const char c = tb[(size_t)i * W + j];
		path[n++] = c;
		if (c == 'B') { --i; --j; }
		else if (c == 'X') --i;
		else --j;
	}
	for (uint a = 0; a < n / 2; ++a) {
		char t = path[a]; path[a] = path[n - 1 - a]; path[n - 1 - a] = t;
	}
	*pathlen = n;
	free(tb); free(prev); free(cur);
	return score;
}

/* ------------------------------------------------------------------------------------------
 * Sparse matrix view used by the relax restatement (mysparsemx.h:6-98 layout).
 */
typedef struct {
	uint LX, LY, nnz;
	const uint *off;   /* LX+1 */
	const byte *val;   /* nnz x {float P; uint col} */
} orc_sp;

static inline float sp_p(const orc_sp *m, uint k) { float p; memcpy(&p, m->val + 8 * (size_t)k, 4); return p; }
static inline uint sp_c(const orc_sp *m, uint k) { uint c; memcpy(&c, m->val + 8 * (size_t)k + 4, 4); return c; }

/* mysparsemx.cpp:44-62 (GetProb): linear search in row i, 0 when absent */
static float sp_get(const orc_sp *m, uint i, uint j)
{
	for (uint k = m->off[i]; k < m->off[i + 1]; ++k) {
		uint c = sp_c(m, k);
		if (c == j) return sp_p(m, k);
		if (c > j) return 0;
	}
	return 0;
}

/* relaxflat.cpp:4-31, X<Z<Y */
static void relax_xz_zy(const orc_sp *XZ, const orc_sp *ZY, float w, float *Post)
{
	const uint LY = ZY->LY;
	for (uint x = 0; x < XZ->LX; ++x)
		for (uint k = XZ->off[x]; k < XZ->off[x + 1]; ++k) {
			const float pxz = sp_p(XZ, k);
			const uint z = sp_c(XZ, k);
			for (uint m = ZY->off[z]; m < ZY->off[z + 1]; ++m)
				Post[(size_t)x * LY + sp_c(ZY, m)] += w * pxz * sp_p(ZY, m);
		}
}

/* relaxflat.cpp:33-60, Z<X<Y */
static void relax_zx_zy(const orc_sp *ZX, const orc_sp *ZY, float w, float *Post)
{
	const uint LY = ZY->LY;
	for (uint z = 0; z < ZX->LX; ++z)
		for (uint k = ZX->off[z]; k < ZX->off[z + 1]; ++k) {
			const float pzx = sp_p(ZX, k);
			const uint x = sp_c(ZX, k);
			for (uint m = ZY->off[z]; m < ZY->off[z + 1]; ++m)
				Post[(size_t)x * LY + sp_c(ZY, m)] += w * pzx * sp_p(ZY, m);
		}
}

/* relaxflat.cpp:62-94 with GetColToRowLoHi (mysparsemx.cpp:238-268), X<Y<Z */
static void relax_xz_yz(const orc_sp *XZ, const orc_sp *YZ, float w, float *Post)
{
	const uint LY = YZ->LX, LZ = XZ->LY;
	uint *lo = (uint *)malloc(sizeof(uint) * (LZ ? LZ : 1));
	uint *hi = (uint *)malloc(sizeof(uint) * (LZ ? LZ : 1));
	for (uint z = 0; z < LZ; ++z) lo[z] = hi[z] = UINT_MAX;
	for (uint y = 0; y < YZ->LX; ++y)
		for (uint k = YZ->off[y]; k < YZ->off[y + 1]; ++k) {
			uint z = sp_c(YZ, k);
			if (lo[z] == UINT_MAX) { lo[z] = y; hi[z] = y; }
			else { if (y < lo[z]) lo[z] = y; if (y > hi[z]) hi[z] = y; }
		}
	for (uint x = 0; x < XZ->LX; ++x)
		for (uint k = XZ->off[x]; k < XZ->off[x + 1]; ++k) {
			const float pxz = sp_p(XZ, k);
			const uint z = sp_c(XZ, k);
			if (lo[z] == UINT_MAX) continue;
			for (uint y = lo[z]; y <= hi[z]; ++y)
				Post[(size_t)x * LY + y] += w * pxz * sp_get(YZ, y, z);
		}
	free(lo); free(hi);
}

/*
 * Whole store for N sequences: pair k <-> (i<j) row-major (mpcflat.cpp:139-159).
 */
typedef struct {
	uint n;
	const uint *len;       /* n                               */
	uint npairs;
	uint **off;            /* npairs pointers, LX+1 each      */
	byte **val;            /* npairs pointers, 8*nnz each     */
	uint *nnz;             /* npairs                          */
} orc_store;

static inline uint pair_index(uint n, uint i, uint j) /* i<j */
{
	return i * n - (i * (i + 1)) / 2 + (j - i - 1);
}
uint orc_pair_index(uint n, uint i, uint j) { return pair_index(n, i, j); }

static void sp_view(const orc_store *s, uint i, uint j, orc_sp *m)
{
	uint k = pair_index(s->n, i, j);
	m->LX = s->len[i]; m->LY = s->len[j]; m->nnz = s->nnz[k];
	m->off = s->off[k]; m->val = s->val[k];
}

/*
 * conspairflat.cpp:10-110 (ConsPair) + mysparsemx.cpp:87-113 (UpdateFromPost):
 * new values of pair (X,Y) on its old pattern; out_val receives 8*nnz bytes {P', col}.
 */
void orc_cons_pair(const orc_store *s, uint X, uint Y, byte *out_val)
{
	orc_sp XY; sp_view(s, X, Y, &XY);
	const uint LX = XY.LX, LY = XY.LY;
	float *Post = (float *)calloc(nz1((size_t)LX * LY), sizeof(float));
	/* ToPost (mysparsemx.cpp:220-236) then *2 (conspairflat.cpp:29-30) */
	for (uint x = 0; x < LX; ++x)
		for (uint k = XY.off[x]; k < XY.off[x + 1]; ++k)
			Post[(size_t)x * LY + sp_c(&XY, k)] = sp_p(&XY, k);
	for (size_t k = 0; k < (size_t)LX * LY; ++k)
		Post[k] *= 2;
	const float w = 1.0f; /* conspairflat.cpp:42 */
	for (uint Z = 0; Z < s->n; ++Z) {
		if (Z == X || Z == Y) continue;
		orc_sp A, Bm;
		if (Z < X) { sp_view(s, Z, X, &A); sp_view(s, Z, Y, &Bm); relax_zx_zy(&A, &Bm, w, Post); }
		else if (Z < Y) { sp_view(s, X, Z, &A); sp_view(s, Z, Y, &Bm); relax_xz_zy(&A, &Bm, w, Post); }
		else { sp_view(s, X, Z, &A); sp_view(s, Y, Z, &Bm); relax_xz_yz(&A, &Bm, w, Post); }
	}
	const uint N = s->n;
	for (uint x = 0; x < LX; ++x)
		for (uint k = XY.off[x]; k < XY.off[x + 1]; ++k) {
			const uint c = sp_c(&XY, k);
			const float p = Post[(size_t)x * LY + c] / N; /* mysparsemx.cpp:108 */
			memcpy(out_val + 8 * (size_t)k, &p, 4);
			memcpy(out_val + 8 * (size_t)k + 4, &c, 4);
		}
	free(Post);
}

/* ------------------------------------------------------------------------------------------
 * Stage drivers (OpenMP over pairs, like mpcflat.cpp:243 / consflat.cpp:11). The caller owns
 * all buffers; val/off arrays for stage A are allocated here with malloc and freed by
 * orc_store_free.
 */
/* calcposteriorflat.cpp:45-92 + calcpost.cpp:4-36 for one pair */
static void pair_posterior_any(const emit_t *e, uint LX, uint LY,
	uint **off_out, byte **val_out, uint *nnz_out, float *ea_out)
{
	const size_t fb = (size_t)NS * ((size_t)LX + 1) * ((size_t)LY + 1);
	float *F = (float *)malloc(fb * sizeof(float));
	float *B = (float *)malloc(fb * sizeof(float));
	float *Post = (float *)malloc(nz1((size_t)LX * LY) * sizeof(float));
	fwd_any(e, LX, LY, F);
	bwd_any(e, LX, LY, B);
	orc_post(F, B, LX, LY, Post);
	free(F); free(B);
	uint *off = (uint *)malloc(((size_t)LX + 1) * sizeof(uint));
	uint nnz = orc_sparse_from_post(Post, LX, LY, off, NULL);
	byte *val = (byte *)malloc((size_t)(nnz ? nnz : 1) * 8);
	orc_sparse_from_post(Post, LX, LY, off, val);
	*ea_out = orc_ea(orc_aln_score(Post, LX, LY), LX, LY);
	free(Post);
	*off_out = off; *val_out = val; *nnz_out = nnz;
}

void orc_pair_posterior(const orc_hmm *h, const byte *X, uint LX, const byte *Y, uint LY,
	uint **off_out, byte **val_out, uint *nnz_out, float *ea_out)
{
	const emit_t e = {h, X, Y, NULL};
	pair_posterior_any(&e, LX, LY, off_out, val_out, nnz_out, ea_out);
}

orc_store *orc_store_new(uint n, const uint *len)
{
	orc_store *s = (orc_store *)calloc(1, sizeof(orc_store));
	s->n = n;
	uint *l = (uint *)malloc(sizeof(uint) * n);
	memcpy(l, len, sizeof(uint) * n);
	s->len = l;
	s->npairs = n * (n - 1) / 2;
	s->off = (uint **)calloc(s->npairs ? s->npairs : 1, sizeof(uint *));
	s->val = (byte **)calloc(s->npairs ? s->npairs : 1, sizeof(byte *));
	s->nnz = (uint *)calloc(s->npairs ? s->npairs : 1, sizeof(uint));
	return s;
}

void orc_store_free(orc_store *s)
{
	if (!s) return;
	for (uint k = 0; k < s->npairs; ++k) { free(s->off[k]); free(s->val[k]); }
	free(s->off); free(s->val); free(s->nnz); free((void *)s->len); free(s);
}

uint orc_store_nnz(const orc_store *s, uint k) { return s->nnz[k]; }
void orc_store_get(const orc_store *s, uint k, uint *off, byte *val)
{
	uint i = 0, j = 0, c = 0;
	for (i = 0; i < s->n; ++i) { if (k < c + (s->n - 1 - i)) { j = i + 1 + (k - c); break; } c += s->n - 1 - i; }
	memcpy(off, s->off[k], sizeof(uint) * ((size_t)s->len[i] + 1));
	memcpy(val, s->val[k], 8 * (size_t)s->nnz[k]);
	(void)j;
}

/* Install a pair's sparse matrix from caller data (used to seed relax tests from fixtures). */
void orc_store_set(orc_store *s, uint k, uint LX, const uint *off, const byte *val)
{
	free(s->off[k]); free(s->val[k]);
	uint nnz = off[LX];
	s->off[k] = (uint *)malloc(sizeof(uint) * ((size_t)LX + 1));
	memcpy(s->off[k], off, sizeof(uint) * ((size_t)LX + 1));
	s->val[k] = (byte *)malloc((size_t)(nnz ? nnz : 1) * 8);
	memcpy(s->val[k], val, 8 * (size_t)nnz);
	s->nnz[k] = nnz;
}

/* MPCFlat::CalcPosteriors (mpcflat.cpp:214-252): all pairs (or the sub-range [k0,k1)), ea[k] per pair */
static void calc_posteriors_any(const orc_hmm *h, const orc_mega *g, orc_store *s, const byte *const *seqs, float *ea,
	uint k0, uint k1, int threads)
{
	const uint n = s->n;
	uint *pi = (uint *)malloc(sizeof(uint) * (s->npairs ? s->npairs : 1));
	uint *pj = (uint *)malloc(sizeof(uint) * (s->npairs ? s->npairs : 1));
	uint k = 0;
	for (uint i = 0; i < n; ++i)
		for (uint j = i + 1; j < n; ++j) { pi[k] = i; pj[k] = j; ++k; }
	if (k1 > s->npairs) k1 = s->npairs;
#ifdef _OPENMP
	if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
	for (int q = (int)k0; q < (int)k1; ++q) {
		uint i = pi[q], j = pj[q];
		free(s->off[q]); free(s->val[q]);
		const emit_t e = {h, seqs[i], seqs[j], g};
		pair_posterior_any(&e, s->len[i], s->len[j], &s->off[q], &s->val[q], &s->nnz[q], &ea[q]);
	}
	free(pi); free(pj);
}

void orc_calc_posteriors(const orc_hmm *h, orc_store *s, const byte *const *seqs, float *ea,
	uint k0, uint k1, int threads)
{
	calc_posteriors_any(h, NULL, s, seqs, ea, k0, k1, threads);
}

/* the same loop when a .mega input is loaded: CalcPost takes the profile branch (calcpost.cpp:14-22);
 * profs[i] = profile of sequence i, position-major, g->nfeat letters per position */
void orc_calc_posteriors_mega(const orc_hmm *h, const orc_mega *g, orc_store *s, const byte *const *profs, float *ea,
	uint k0, uint k1, int threads)
{
	calc_posteriors_any(h, g, s, profs, ea, k0, k1, threads);
}

/* MPCFlat::ConsIter (consflat.cpp:5-23): Jacobi update of pairs [k0,k1) into a fresh store `dst`
 * (dst shares the pattern: offsets are copied, values are new). */
void orc_cons_iter(const orc_store *src, orc_store *dst, uint k0, uint k1, int threads)
{
	const uint n = src->n;
	uint *pi = (uint *)malloc(sizeof(uint) * (src->npairs ? src->npairs : 1));
	uint *pj = (uint *)malloc(sizeof(uint) * (src->npairs ? src->npairs : 1));
	uint k = 0;
	for (uint i = 0; i < n; ++i)
		for (uint j = i + 1; j < n; ++j) { pi[k] = i; pj[k] = j; ++k; }
	if (k1 > src->npairs) k1 = src->npairs;
#ifdef _OPENMP
	if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
	for (int q = (int)k0; q < (int)k1; ++q) {
		uint LX = src->len[pi[q]];
		uint nnz = src->nnz[q];
		free(dst->off[q]); free(dst->val[q]);
		dst->off[q] = (uint *)malloc(sizeof(uint) * ((size_t)LX + 1));
		memcpy(dst->off[q], src->off[q], sizeof(uint) * ((size_t)LX + 1));
		dst->val[q] = (byte *)malloc((size_t)(nnz ? nnz : 1) * 8);
		dst->nnz[q] = nnz;
		orc_cons_pair(src, pi[q], pj[q], dst->val[q]);
	}
	free(pi); free(pj);
}

/* ------------------------------------------------------------------------------------------
 * glibc 2.35 expf restated op-for-op (sysdeps/ieee754/flt-32/e_expf.c, e_exp2f_data.c; the
 * algorithm is Szabolcs Nagy's from ARM optimized-routines: N=32 table, cubic in r, double
 * arithmetic, one final rounding to float). This is what `expf` in calcposteriorflat.cpp:20
 * resolves to on this image. glibc ships two x86-64 builds of it selected by an ifunc
 * (sysdeps/x86_64/fpu/multiarch/e_expf.c): the baseline one (separate mul/add) and the FMA one
 * (`use_fma` != 0; contraction pattern read from the disassembly of libm.so.6 __expf_fma:
 * kd = fma(InvLn2N, x, SHIFT); r = fma(InvLn2N, x, -kd); z = fma(r, C0, C1); y = fma(r, C2, 1);
 * y = fma(z, r*r, y); y*s). The device kernel carries the same two variants; this host copy
 * exists so tests can pin the emulation against libm's expf bit-for-bit.
 * Valid for the only range the path uses: logf(0.01f) <= x < 0.
 */
static const uint64_t EXP2F_TAB[32] = {
	0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
	0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
	0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
	0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
	0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
	0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
	0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
	0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline uint64_t d2u(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }

float orc_expf_emul(float x, int use_fma)
{
	const double InvLn2N = u2d(0x40471547652b82feull); /* 0x1.71547652b82fep+0 * 32 */
	const double SHIFT = u2d(0x4338000000000000ull);   /* 0x1.8p+52 */
	const double C0 = u2d(0x3ebc6af84b912394ull);      /* 0x1.c6af84b912394p-5 / 32^3 */
	const double C1 = u2d(0x3f2ebfce50fac4f3ull);      /* 0x1.ebfce50fac4f3p-3 / 32^2 */
	const double C2 = u2d(0x3f962e42ff0c52d6ull);      /* 0x1.62e42ff0c52d6p-1 / 32 */
	const double xd = (double)x;
	double kd, r, z, y, r2, s;
	uint64_t ki, t;
	if (use_fma) {
		kd = fma(InvLn2N, xd, SHIFT);
		ki = d2u(kd);
		kd -= SHIFT;
		r = fma(InvLn2N, xd, -kd);
		z = fma(r, C0, C1);
		r2 = r * r;
		y = fma(r, C2, 1.0);
		y = fma(z, r2, y);
	} else {
		z = InvLn2N * xd;
		kd = z + SHIFT;
		ki = d2u(kd);
		kd -= SHIFT;
		r = z - kd;
		z = C0 * r + C1;
		r2 = r * r;
		y = C2 * r + 1.0;
		y = z * r2 + y;
	}
	t = EXP2F_TAB[ki % 32];
	t += ki << (52 - 5);
	s = u2d(t);
	y = y * s;
	return (float)y;
}

float orc_libm_expf(float x) { return expf(x); }

/* Which glibc expf variant this host resolves to (ifunc-fma.h: FMA and AVX2 usable). */
int orc_host_expf_uses_fma(void)
{
#if defined(__x86_64__)
	__builtin_cpu_init();
	return __builtin_cpu_supports("fma") && __builtin_cpu_supports("avx2");
#else
	return 0;
#endif
}
