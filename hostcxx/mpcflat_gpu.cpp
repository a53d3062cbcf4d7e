// hostcxx/mpcflat_gpu.cpp — link-time drop-in that makes an UNMODIFIED mpcflat.{h,cpp} consume
// the MI355X path (include/mpcgpu.h, libmpcgpu.so).
//
// The reference (rcedgar/muscle 5.3) has no plugin API; its seam is one numeric function per
// translation unit. This file is compiled against the reference's own headers
// (-I<muscle>/src, nothing copied) and defines the two MPCFlat members that own the hot loops:
//
//   MPCFlat::CalcPosterior(uint)   replaces calcposteriorflat.cpp:45-92  (called from the OpenMP loop
//                                  of MPCFlat::CalcPosteriors, mpcflat.cpp:239-251)
//   MPCFlat::ConsIter(uint)        replaces consflat.cpp:5-23            (called per iteration from
//                                  MPCFlat::Consistency, mpcflat.cpp:173-181)
//
//   MPCFlat::AlignAlns(MSA1,MSA2) replaces alnalnsflat.cpp:7-52: BuildPost (buildpostflat.cpp:18-106) +
//                                  CalcAlnFlat/TraceBackFlat (calcalnflat.cpp:6, tracebackflat.cpp:3) run on
//                                  the device store, so the progressive stage and the 100 refinement
//                                  rounds never need the sparse matrices on the host
//
//   PProg::AlignMSAsFlat(...)      replaces alnmsasflat.cpp:4-50: the join of two MSAs in the super5/super7
//                                  drivers — GetPostPairsAlignedFlat (getpostpairsalignedflat.cpp:5-98: fwd/bwd/
//                                  posterior per sampled cross pair) + CalcPosteriorFlat3 + CalcAlnFlat
//
// Build: every reference object except consflat.o, alnalnsflat.o and alnmsasflat.o, calcposteriorflat.o with its CalcPosterior
// symbol weakened (the same object also defines CalcPostFlat and the two vestigial virtuals that
// other translation units / the vtable need), plus this file, plus -lmpcgpu
// (hostcxx/build_muscle_gpu.sh; INTEGRATION.md shows the two-line change a maintainer would make
// in the source tree instead of the objcopy step).
//
// A .mega input (structure profiles, loadinput.cpp:5-9) makes the reference's CalcPost take its profile
// branch (calcpost.cpp:14-22): here the Mega statics are handed to the library (mpcgpu_set_mega) right
// after the sequences, in MPCFlat::CalcPosterior and in PProg::AlignMSAsFlat alike.
//
// There is no CPU path in here: if libmpcgpu cannot create a device context the run Die()s, like
// every other fatal condition in the reference (myutils.cpp:883-927).
#include "muscle.h"
#include "mpcflat.h"
#include "pairhmm.h"
#include "pprog.h"
#include <memory>
#include "mega.h"
#include "mpcflat_mega.h"
#include "super7.h"
#include "uclust.h"
#include "mpcgpu.h"

#include <malloc.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <set>
#include <mutex>
#include <thread>

// hostcxx/rand_isolate.cpp
struct MuscleGpuRandSnapshot { char m_State[128]; int m_FOff; int m_ROff; }; // hostcxx/rand_isolate.cpp
extern "C" void MuscleGpuRandSharedSnapshots(const unsigned long long *Draws, unsigned Count, MuscleGpuRandSnapshot *At);
extern "C" void MuscleGpuRandThreadRestore(const MuscleGpuRandSnapshot *At);
extern "C" void MuscleGpuRandThreadEnd(void);

namespace
{
struct Batch
	{
	const MultiSequence *m_Seqs = 0;
	uint m_PairCount = 0;
	uint m_Served = 0;	// CalcPosterior calls answered from this batch
	bool m_Materialise = false; // host copies of the stage-A matrices are needed (no relax follows)
	bool m_OnHost = false;	// the CURRENT device matrices have been copied into the MySparseMx objects
	uint m_ItersDone = 0;	// ConsIter calls since the batch started
	unsigned long long m_Gen = 0;	// which StartBatch this is (process-wide count): what state derived from the batch is checked against
	vector<float> m_EA;	// what calcposteriorflat.cpp:89 stores in m_DistMx
// what the batch was computed from: a caller that re-runs InitSeqs on the same MPCFlat object (cmd_profseq does, with a
// fresh MPCFlat at the same stack address per query) must get a new batch, not the previous one's numbers
	vector<const byte *> m_SeqPtrs;
	vector<uint> m_SeqLens;
	vector<uint64_t> m_SeqEnds; // 64-bit content hash of every sequence (all bytes)
	};

// 64-bit hash of ALL bytes of a sequence, 8 bytes per step (multiply-xorshift rounds): ~50 ns for 400 residues, so every
// CalcPosterior call can afford to re-check both of its sequences — a caller that rewrites the middle of a buffer in place
// (same pointer, same length, same ends) gets a new batch instead of the previous contents' posteriors.
uint64_t SeqEnds(const byte *p, uint L)
	{
	uint64_t h = 0x9e3779b97f4a7c15ull ^ ((uint64_t) L * 0xff51afd7ed558ccdull);
	uint i = 0;
	for (; i + 8 <= L; i += 8)
		{
		uint64_t w;
		memcpy(&w, p + i, 8);
		h = (h ^ w) * 0x9fb21c651e98df25ull;
		h ^= h >> 29;
		}
	uint64_t w = 0;
	for (uint k = 0; i + k < L; ++k)
		w |= (uint64_t) p[i + k] << (8 * k);
	h = (h ^ w) * 0x9fb21c651e98df25ull;
	h ^= h >> 32;
	return h;
	}

// One Slot = one device context (or one group of contexts) with the state of the MPCFlat run it is serving. Slot 0 serves every
// MPCFlat that was not given a slot of its own (-align, -super5, the sequential shrub loop); slots 1.. belong to the worker
// threads of the parallel shrub loop of -super7 (Super7::IntraAlignShrubs below), each with its own context so that the small
// launches of different shrubs overlap on the device(s).
struct Slot
	{
	std::mutex m_Mu;
	mpcgpu_ctx *m_Ctx = 0;
	mpcgpu_group *m_Group = 0;	// MUSCLE_GPU_DEVICES (slot 0 only): the all-pairs stage sharded over several GPUs; m_Ctx is its rank 0
	const MPCFlat *m_StoreOwner = 0; // the MPCFlat whose all-pairs store the context currently holds
	};
enum { MAX_SLOTS = 65 };
Slot g_Slots[MAX_SLOTS];
std::mutex g_MapMu;	// guards the two maps (not the batches themselves: a batch is used under its slot's mutex)
std::map<const MPCFlat *, Batch> g_Batches;
std::map<const MPCFlat *, int> g_SlotOf;
// PProg joins and pair lists (AlignMSAsFlat, AlignPairFlat, UClust::Search): contexts of their own, so that they never disturb the
// store of an MPCFlat run — ONE PER LISTED DEVICE (MUSCLE_GPU_DEVICES), each under its own mutex. A calling thread keeps the join
// context it was dealt: the worker threads of the parallel shrub loop that of their slot's device (so a shrub's joins run where
// its stage ran), any other thread the next one in turn (the OpenMP loops around AlignPairFlat spread over the devices).
struct JoinCtx
	{
	std::mutex m_Mu;
	mpcgpu_ctx *m_Ctx = 0;
	};
enum { MAX_JOIN_CTX = 64 };
JoinCtx g_Join[MAX_JOIN_CTX];
std::atomic<unsigned> g_JoinNext(0);
thread_local int t_JoinIndex = -1;
// join contexts beyond one per device: the worker threads of PProg::Run2 below run independent joins side by side, each on a context
// of its own (context i lives on device i mod #devices)
std::atomic<unsigned> g_JoinCtxWanted(0);

int SlotIndexOf(const MPCFlat *M)
	{
	std::lock_guard<std::mutex> Guard(g_MapMu);
	std::map<const MPCFlat *, int>::const_iterator p = g_SlotOf.find(M);
	return p == g_SlotOf.end() ? 0 : p->second;
	}

Batch &BatchOf(const MPCFlat *M)
	{
	std::lock_guard<std::mutex> Guard(g_MapMu);
	return g_Batches[M]; // std::map: references stay valid when other keys are inserted
	}

// MUSCLE_GPU_DEVICES="0-7" | "0,1,2,3" (an ordinal may repeat, e.g. "0,0" on a one-GPU box): one context per listed device,
// pair loops sharded over them (include/mpcgpu.h, mpcgpu_group_*). Without it: one context on MUSCLE_GPU_DEVICE (default 0).
vector<int> ParseDevices(const char *s)
	{
	vector<int> Devs;
	while (*s != 0)
		{
		char *End = 0;
		long a = strtol(s, &End, 10);
		if (End == s)
			Die("MUSCLE_GPU_DEVICES: cannot parse '%s'", s);
		long b = a;
		s = End;
		if (*s == '-')
			{
			b = strtol(s + 1, &End, 10);
			if (End == s + 1 || b < a)
				Die("MUSCLE_GPU_DEVICES: bad range");
			s = End;
			}
		for (long d = a; d <= b; ++d)
			Devs.push_back((int) d);
		if (*s == ',')
			++s;
		else if (*s != 0)
			Die("MUSCLE_GPU_DEVICES: unexpected '%c'", *s);
		}
	return Devs;
	}

vector<int> DeviceList()
	{
	const char *List = getenv("MUSCLE_GPU_DEVICES");
	if (List != 0 && *List != 0)
		{
		vector<int> Devs = ParseDevices(List);
		if (Devs.empty())
			Die("MUSCLE_GPU_DEVICES is empty");
		return Devs;
		}
	int Device = 0;
	const char *s = getenv("MUSCLE_GPU_DEVICE");
	if (s != 0 && *s != 0)
		Device = atoi(s);
	return vector<int>(1, Device);
	}

// The join context of the calling thread (see JoinCtx), created on first use; lock its m_Mu around the library calls.
JoinCtx &JoinOfThisThread()
	{
	const vector<int> Devs = DeviceList();
	const unsigned N = (unsigned) std::min<size_t>(std::max<size_t>(Devs.size(), g_JoinCtxWanted.load()), MAX_JOIN_CTX);
	if (t_JoinIndex < 0 || (unsigned) t_JoinIndex >= N)
		t_JoinIndex = (int) (g_JoinNext.fetch_add(1) % N);
	JoinCtx &J = g_Join[t_JoinIndex];
	std::lock_guard<std::mutex> Guard(J.m_Mu);
	if (J.m_Ctx == 0)
		{
		if (mpcgpu_create(&J.m_Ctx, Devs[(size_t) t_JoinIndex % Devs.size()]) != 0)
			Die("GPU posterior stage: %s", mpcgpu_last_error(0));
		}
	return J;
	}

// Context of a slot, created on first use (call with the slot's mutex held). Slot 0: a group when MUSCLE_GPU_DEVICES lists
// several devices, else one context. Worker slots: one context each, dealt round-robin over the listed devices.
bool TimingEnv()
	{
	const char *s = getenv("MUSCLE_GPU_TIMING");
	return s != 0 && *s != 0 && *s != '0';
	}
const std::chrono::steady_clock::time_point g_ProcessStart = std::chrono::steady_clock::now(); // static initialisation of this object file
// Small stores (the <= 32-sequence shrubs of -super7, the clusters of -super5) take whole-record relax tiles: no window records or band
// tables to build and no tiles to cut per store (MPCGPU_RELAX_SMALL_PAIRS, muscle_amd/csrc/mpcgpu.cpp: build_var_store; 4.6 ms of host
// round trips per store otherwise, 412 stores in a 10 000-sequence run). Set here, before any thread exists; the user's own setting wins.
const int g_SmallStoreDefault = setenv("MPCGPU_RELAX_SMALL_PAIRS", "40", 0);
// The allocator keeps freed memory mapped (blocks below 1 GB come from the heap, the heap's top is not trimmed). Host memory that the HIP
// runtime has touched for a copy is registered for DMA; when glibc hands such a block back to the kernel (munmap above its moving mmap
// threshold, or a trim), the kernel driver answers the invalidation by EVICTING the process's device queues and restoring them later:
// every queue of the process stops for 10 - 30 ms (profiles/r10k). One thread's joins lose little to that; six threads
// running joins side by side (PProg::Run2 below) freed such blocks all the time: 11.4 s for the run that takes 7.7 s on one thread and
// 6.4 s with the memory kept (profiles/r11d); no cost to -align (profiles/r11i_align_malloc_ab.log).
const int g_KeepFreedMemoryMapped = []
	{
	mallopt(M_MMAP_THRESHOLD, 1 << 30);
	mallopt(M_TRIM_THRESHOLD, 1 << 30);
	return 1;
	}();
// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise), and streams that share a
// queue run one after the other: the 8 shrub workers and 6 join workers, each with a stream of one-wave kernels, were 4 wide on the
// device (-super7 10 000 x 250: 6.5 -> 6.0 s with 8 queues, 5.9 with 16; profiles/r11f). The runtime reads the variable when its first
// call initialises it — after this object's static initialisation; the user's own setting wins.
const int g_HwQueuesDefault = setenv("GPU_MAX_HW_QUEUES", "16", 0);
double g_CtxSeconds = 0; // MUSCLE_GPU_TIMING: creating contexts (the first one pays for the HIP runtime's start-up)
struct CtxClock
	{
	std::chrono::steady_clock::time_point m_T0 = std::chrono::steady_clock::now();
	~CtxClock() { g_CtxSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - m_T0).count(); }
	};
mpcgpu_ctx *GetCtx(int SlotIndex)
	{
	Slot &S = g_Slots[SlotIndex];
	if (S.m_Ctx != 0)
		return S.m_Ctx;
	CtxClock Clock;
	vector<int> Devs = DeviceList();
	if (SlotIndex == 0 && getenv("MUSCLE_GPU_DEVICES") != 0 && *getenv("MUSCLE_GPU_DEVICES") != 0)
		{
		if (mpcgpu_group_create(&S.m_Group, (uint32_t) Devs.size(), Devs.data()) != 0)
			Die("GPU posterior stage: %s", mpcgpu_group_last_error(0));
		S.m_Ctx = mpcgpu_group_ctx(S.m_Group, 0);
		for (uint32_t r = 0; r < mpcgpu_group_size(S.m_Group); ++r)
			mpcgpu_timers_enable(mpcgpu_group_ctx(S.m_Group, r), TimingEnv() ? 1 : 0);
		if (getenv("MUSCLE_GPU_TIMING") != 0 || getenv("MUSCLE_GPU_DEBUG") != 0)
			fprintf(stderr, "[muscle_gpu] %u GPU contexts, exchange by %s\n", mpcgpu_group_size(S.m_Group), mpcgpu_group_transport(S.m_Group));
		return S.m_Ctx;
		}
	const int Device = Devs[SlotIndex == 0 ? 0 : (SlotIndex - 1) % SIZE(Devs)];
	if (mpcgpu_create(&S.m_Ctx, Device) != 0)
		Die("GPU posterior stage: %s", mpcgpu_last_error(0));
	mpcgpu_timers_enable(S.m_Ctx, TimingEnv() ? 1 : 0); // hipEvents around every launch only when the report is asked for
	return S.m_Ctx;
	}

// MUSCLE_GPU_DEBUG=1: FNV-1a digests of what crosses the boundary, on stderr (diagnostics)
// The sparse matrices are only read by BuildPost, which AlignAlns below keeps on the device; they
// are copied back to MySparseMx objects only on request (MUSCLE_GPU_DOWNLOAD=1, or MUSCLE_GPU_DEBUG=1
// for the digests).
bool DownloadOn()
	{
	static int On = -1;
	if (On < 0)
		{
		const char *s = getenv("MUSCLE_GPU_DOWNLOAD");
		const char *d = getenv("MUSCLE_GPU_DEBUG");
		On = ((s != 0 && *s != 0 && *s != '0') || (d != 0 && *d != 0 && *d != '0')) ? 1 : 0;
		}
	return On == 1;
	}

bool DebugOn()
	{
	static int On = -1;
	if (On < 0)
		{
		const char *s = getenv("MUSCLE_GPU_DEBUG");
		On = (s != 0 && *s != 0 && *s != '0') ? 1 : 0;
		}
	return On == 1;
	}

// MUSCLE_GPU_TIMING=1: wall time spent inside the replaced functions, printed at exit (where does an
// end-to-end run spend its time once the stage itself takes a few seconds).
enum { T_STAGE_A, T_CONS_ITER, T_ALN_PREP, T_ALN_LIB, T_ALN_POST, T_JOIN_PREP, T_JOIN_LIB, T_PAIRS_PREP, T_PAIRS_LIB, T_COUNT };
// (atomics: the shrub and join workers of -super7 stop their watches concurrently — round-5 advisor finding)
std::atomic<unsigned long long> g_Nanos[T_COUNT];
std::atomic<unsigned long long> g_Calls[T_COUNT];
// The timeline of ONE MPCFlat::Run (muscle -align): when the replaced functions were entered first / left last, in ns since process
// start. What lies BETWEEN them is the reference's own host code — the row f4 of SURVEY.md 8 (CalcGuideTree = UPGMA5,
// upgma5.cpp:87-..., ClustalWeights, CalcJoinOrder, the bookkeeping of ProgAln / RefineIter, SortMSA, output) — which this file does
// not replace and could not time from inside.
enum { M_POST_ENTER, M_POST_EXIT, M_CONS_ENTER, M_CONS_EXIT, M_ALN_ENTER, M_ALN_EXIT, M_COUNT };
std::atomic<unsigned long long> g_Mark[M_COUNT];
bool TimingOn();
void PhaseMark(int Which, bool First)
	{
	if (!TimingOn())
		return;
	const unsigned long long t = (unsigned long long) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - g_ProcessStart).count() + 1;
	unsigned long long Old = g_Mark[Which].load();
	if (First)
		{
		while (Old == 0 && !g_Mark[Which].compare_exchange_weak(Old, t)) {}
		}
	else
		g_Mark[Which].store(t);
	}
bool TimingOn()
	{
	static int On = -1;
	if (On < 0)
		{
		const char *s = getenv("MUSCLE_GPU_TIMING");
		On = (s != 0 && *s != 0 && *s != '0') ? 1 : 0;
		if (On == 1)
			atexit([]()
				{
				static const char *Names[T_COUNT] = { "stage A (all pairs)", "ConsIter", "AlignAlns: maps",
				  "AlignAlns: library", "AlignAlns: result MSA", "AlignMSAsFlat: pairs+maps", "AlignMSAsFlat: library",
				  "AlignPairFlat lists: registry", "AlignPairFlat lists: library" };
				fprintf(stderr, "[muscle_gpu] %-28s %10.3f s  (static initialisation to exit handlers; the rows below are parts of it)\n",
				  "process", std::chrono::duration<double>(std::chrono::steady_clock::now() - g_ProcessStart).count());
				fprintf(stderr, "[muscle_gpu] %-28s %10.3f s  (inside the first call below that needed it)\n", "context creation, HIP init", g_CtxSeconds);
				for (int i = 0; i < T_COUNT; ++i)
					fprintf(stderr, "[muscle_gpu] %-28s %10.3f s  %8llu calls\n", Names[i], g_Nanos[i].load()*1e-9, g_Calls[i].load());
// (one MPCFlat::Run only: with the shrub workers of -super7 running many at once first-entered / last-left marks interleave)
				const bool OneRun = g_Mark[M_POST_ENTER].load() != 0 && g_Mark[M_POST_ENTER].load() <= g_Mark[M_POST_EXIT].load() &&
				  (g_Mark[M_CONS_ENTER].load() == 0 || (g_Mark[M_POST_EXIT].load() <= g_Mark[M_CONS_ENTER].load() && g_Mark[M_CONS_EXIT].load() <= std::max(g_Mark[M_ALN_ENTER].load(), g_Mark[M_CONS_EXIT].load()))) &&
				  (g_Mark[M_ALN_ENTER].load() == 0 || g_Mark[M_CONS_EXIT].load() <= g_Mark[M_ALN_ENTER].load());
				if (OneRun)
					{
					const double End = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_ProcessStart).count();
					auto At = [](int m) { return g_Mark[m].load()*1e-9; };
					fprintf(stderr, "[muscle_gpu] timeline of the first MPCFlat::Run (s since process start; the gaps are the REFERENCE's host code, SURVEY.md 8 row f4):\n");
					fprintf(stderr, "[muscle_gpu]   %8.3f  start .. CalcPosteriors entered: options, input, Derep, InitSeqs / InitPairs / InitDistMx   %7.3f s\n", At(M_POST_ENTER), At(M_POST_ENTER));
					fprintf(stderr, "[muscle_gpu]   %8.3f  CalcPosteriors (replaced)                                                                   %7.3f s\n", At(M_POST_EXIT), At(M_POST_EXIT) - At(M_POST_ENTER));
					if (g_Mark[M_CONS_ENTER].load() != 0)
						{
						fprintf(stderr, "[muscle_gpu]   %8.3f  .. first ConsIter: CalcGuideTree = UPGMA5 (upgma5.cpp) + PermTree + ClustalWeights                %7.3f s\n", At(M_CONS_ENTER), At(M_CONS_ENTER) - At(M_POST_EXIT));
						fprintf(stderr, "[muscle_gpu]   %8.3f  ConsIter x n (replaced; queues the relax, does not wait for it)                              %7.3f s\n", At(M_CONS_EXIT), At(M_CONS_EXIT) - At(M_CONS_ENTER));
						}
					if (g_Mark[M_ALN_ENTER].load() != 0)
						{
						const double From = g_Mark[M_CONS_EXIT].load() != 0 ? At(M_CONS_EXIT) : At(M_POST_EXIT);
						fprintf(stderr, "[muscle_gpu]   %8.3f  .. first AlignAlns: CalcJoinOrder, ProgressiveAlign's leaf MSAs                                  %7.3f s\n", At(M_ALN_ENTER), At(M_ALN_ENTER) - From);
						const double In = (g_Nanos[T_ALN_PREP].load() + g_Nanos[T_ALN_LIB].load() + g_Nanos[T_ALN_POST].load())*1e-9;
						fprintf(stderr, "[muscle_gpu]   %8.3f  first .. last AlignAlns / RefineIter (replaced): %.3f s inside them (rows above), between them ProgAln's bookkeeping   %7.3f s\n",
						  At(M_ALN_EXIT), In, At(M_ALN_EXIT) - At(M_ALN_ENTER) - In);
						fprintf(stderr, "[muscle_gpu]   %8.3f  .. exit handlers: SortMSA, InsertDupes, output                                                   %7.3f s\n", End, End - At(M_ALN_EXIT));
						}
					}
// device time per kernel family of the library (hipEvents on its stream): where the library seconds above go
				static const char *Fam[MPCGPU_NKERNELS] = { "fwd/bwd", "posterior finish", "store build", "relax", "commit",
				  "BuildPost: records", "BuildPost: sort", "BuildPost: reduce", "CalcAlnFlat+traceback" };
				float Ms[MPCGPU_NKERNELS];
				uint64_t Launches[MPCGPU_NKERNELS];
				if (g_Slots[0].m_Ctx != 0 && mpcgpu_timers_get(g_Slots[0].m_Ctx, Ms, Launches) == 0)
					for (int i = 0; i < MPCGPU_NKERNELS; ++i)
						fprintf(stderr, "[muscle_gpu]   device: %-24s %10.3f s  %8llu launches\n", Fam[i], Ms[i]*1e-3, (unsigned long long) Launches[i]);
				});
		}
	return On == 1;
	}
const bool g_TimingAtStart = TimingOn(); // (decided during static initialisation: before any worker thread can race for the lazy static)
struct Stopwatch
	{
	int m_Slot;
	std::chrono::steady_clock::time_point m_T0;
	explicit Stopwatch(int Slot) : m_Slot(Slot), m_T0(std::chrono::steady_clock::now()) {}
	void Next(int Slot)
		{
		Stop();
		m_Slot = Slot;
		m_T0 = std::chrono::steady_clock::now();
		}
	void Stop()
		{
		if (m_Slot < 0 || !TimingOn())
			return;
		g_Nanos[m_Slot] += (unsigned long long) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - m_T0).count();
		++g_Calls[m_Slot];
		m_Slot = -1;
		}
	~Stopwatch() { Stop(); }
	};

uint64_t Fnv(uint64_t h, const void *p, size_t n)
	{
	const unsigned char *b = (const unsigned char *) p;
	for (size_t i = 0; i < n; ++i)
		{
		h ^= b[i];
		h *= 1099511628211ull;
		}
	return h;
	}

#define GPUCHK(call)	do { if ((call) != 0) Die("GPU posterior stage: %s", mpcgpu_last_error(Ctx)); } while (0)
#define GRPCHK(call)	do { if ((call) != 0) Die("GPU posterior stage: %s", mpcgpu_group_last_error(Group)); } while (0)

// Copies pairs [k0,k1) of the device store into MySparseMx objects (layout of
// mysparsemx.h:6-98; buffers through AllocLX/AllocVec so ownership stays with myalloc/myfree).
template<class GETMX> void Download(mpcgpu_ctx *Ctx, MPCFlat &M, uint PairCount, GETMX GetMx)
	{
	vector<uint32_t> NNZ(PairCount);
	GPUCHK(mpcgpu_get_nnz(Ctx, 0, PairCount, NNZ.data()));
	const uint64_t MaxEntriesPerChunk = uint64_t(32)*1024*1024; // 256 MB of values per transfer
	unsigned ThreadCount = GetRequestedThreadCount();
	uint k0 = 0;
	vector<uint32_t> Offs;
	vector<uint64_t> Vals;
	vector<uint64_t> OffBase, ValBase;
	while (k0 < PairCount)
		{
		uint k1 = k0;
		uint64_t Entries = 0, OffCount = 0;
		OffBase.clear();
		ValBase.clear();
		while (k1 < PairCount && (k1 == k0 || Entries + NNZ[k1] <= MaxEntriesPerChunk))
			{
			OffBase.push_back(OffCount);
			ValBase.push_back(Entries);
			OffCount += M.GetSeqLength(M.GetPair(k1).first) + 1;
			Entries += NNZ[k1];
			++k1;
			}
		Offs.resize(OffCount);
		Vals.resize(Entries + 1);
		GPUCHK(mpcgpu_get_sparse_range(Ctx, k0, k1, Offs.data(), Vals.data()));
#pragma omp parallel for num_threads(ThreadCount) schedule(dynamic, 64)
		for (int k = (int) k0; k < (int) k1; ++k)
			{
			const pair<uint, uint> &Pair = M.GetPair((uint) k);
			const uint LX = M.GetSeqLength(Pair.first);
			const uint LY = M.GetSeqLength(Pair.second);
			MySparseMx &Mx = GetMx((uint) k);
			Mx.AllocLX(LX);
			Mx.AllocVec(NNZ[k]);
			Mx.m_LX = LX;
			Mx.m_LY = LY;
			Mx.m_VecSize = NNZ[k];
			memcpy(Mx.m_Offsets, Offs.data() + OffBase[k - k0], sizeof(uint)*(LX + 1));
			memcpy(Mx.m_ValueVec, Vals.data() + ValBase[k - k0], 8*size_t(NNZ[k]));
			Mx.m_X = M.GetBytePtr(Pair.first);
			Mx.m_Y = M.GetBytePtr(Pair.second);
			}
		k0 = k1;
		}
	if (DebugOn())
		{
		uint64_t h = 14695981039346656037ull;
		uint64_t Total = 0;
		for (uint k = 0; k < PairCount; ++k)
			{
			MySparseMx &Mx = GetMx(k);
			h = Fnv(h, Mx.m_Offsets, sizeof(uint)*(Mx.m_LX + 1));
			h = Fnv(h, Mx.m_ValueVec, 8*size_t(NNZ[k]));
			Total += NNZ[k];
			}
		fprintf(stderr, "[muscle_gpu] download: %u pairs, %llu entries, fnv %016llx\n", PairCount,
		  (unsigned long long) Total, (unsigned long long) h);
		}
	}

// calcpost.cpp:14-22: with a .mega input loaded the emissions come from the structure profiles of the
// sequences (looked up by label, like CalcPost does), not from the PairHMM letter tables. Labels[i] is
// the label of sequence i of the set just given to mpcgpu_set_seqs / mpcgpu_set_seqs_registry.
void SetMega(mpcgpu_ctx *Ctx, const vector<string> &Labels, const vector<uint32_t> &Lens, mpcgpu_group *Group = 0)
	{
	if (!Mega::m_Loaded)
		return; // set_seqs already switched the context back to letter emissions
	const uint FeatureCount = Mega::GetFeatureCount();
	vector<uint32_t> AlphaSizes(FeatureCount);
	vector<float> Weights(FeatureCount);
	vector<vector<float> > Mxs(FeatureCount);
	vector<const float *> LogProbPtrs(FeatureCount), MxPtrs(FeatureCount);
	for (uint f = 0; f < FeatureCount; ++f)
		{
		const uint A = Mega::GetAlphaSize(f);
		AlphaSizes[f] = A;
		Weights[f] = Mega::GetWeight(f);
		asserta(SIZE(Mega::m_LogProbsVec[f]) == A);
		LogProbPtrs[f] = Mega::m_LogProbsVec[f].data();
		const vector<vector<float> > &Mx = Mega::m_LogProbMxVec[f];
		asserta(SIZE(Mx) == A);
		for (uint a = 0; a < A; ++a)
			{
			asserta(SIZE(Mx[a]) == A);
			Mxs[f].insert(Mxs[f].end(), Mx[a].begin(), Mx[a].end());
			}
		MxPtrs[f] = Mxs[f].data();
		}
	const uint SeqCount = SIZE(Labels);
	vector<vector<uint8_t> > Profs(SeqCount);
	vector<const uint8_t *> ProfPtrs(SeqCount);
	for (uint i = 0; i < SeqCount; ++i)
		{
		const vector<vector<byte> > &Profile = *Mega::GetProfileByLabel(Labels[i]);
		asserta(SIZE(Profile) == Lens[i]); // calcpost.cpp:18-19
		Profs[i].reserve(size_t(Lens[i])*FeatureCount);
		for (uint Pos = 0; Pos < Lens[i]; ++Pos)
			{
			asserta(SIZE(Profile[Pos]) == FeatureCount);
			Profs[i].insert(Profs[i].end(), Profile[Pos].begin(), Profile[Pos].end());
			}
		ProfPtrs[i] = Profs[i].data();
		}
	if (Group != 0)
		GRPCHK(mpcgpu_group_set_mega(Group, FeatureCount, AlphaSizes.data(), Weights.data(), LogProbPtrs.data(),
		  MxPtrs.data(), ProfPtrs.data()));
	else
		GPUCHK(mpcgpu_set_mega(Ctx, FeatureCount, AlphaSizes.data(), Weights.data(), LogProbPtrs.data(),
		  MxPtrs.data(), ProfPtrs.data()));
	}

// First CalcPosterior call of a run: the whole all-pairs stage A on the device.
void StartBatch(MPCFlat &M, Batch &B, int SlotIndex)
	{
	static std::atomic<unsigned long long> s_Gen(0);
	B.m_Gen = ++s_Gen;
	Stopwatch SW(T_STAGE_A);
	mpcgpu_ctx *Ctx = GetCtx(SlotIndex);
	mpcgpu_group *Group = g_Slots[SlotIndex].m_Group;
	const uint SeqCount = M.GetSeqCount();
	const uint PairCount = SIZE(M.m_Pairs);
	asserta(PairCount == (SeqCount*(SeqCount - 1))/2);

// The PairHMM tables are process globals that can change between replicates (align.cpp:35-40)
	if (Group != 0)
		GRPCHK(mpcgpu_group_set_hmm(Group, PairHMM::m_StartScore, &PairHMM::m_TransScore[0][0],
		  &PairHMM::m_MatchScore[0][0], PairHMM::m_InsScore, MIN_SPARSE_SCORE, -1));
	else
		GPUCHK(mpcgpu_set_hmm(Ctx, PairHMM::m_StartScore, &PairHMM::m_TransScore[0][0],
		  &PairHMM::m_MatchScore[0][0], PairHMM::m_InsScore, MIN_SPARSE_SCORE, -1));

	vector<const uint8_t *> Ptrs(SeqCount);
	vector<uint32_t> Lens(SeqCount);
	vector<string> Labels(SeqCount);
	for (uint i = 0; i < SeqCount; ++i)
		{
		Ptrs[i] = M.GetBytePtr(i);
		Lens[i] = M.GetSeqLength(i);
		Labels[i] = string(M.GetLabel(i)); // calcposteriorflat.cpp:63-64
		}
// (the "HMM overflow" length check of calcposteriorflat.cpp:54-61 is made by the library)
	if (Group != 0)
		{
// mpcflat.cpp:239-251 sharded over the devices + the all-gather of the sparse posteriors (mpcgpu_group.cpp)
		GRPCHK(mpcgpu_group_set_seqs(Group, SeqCount, Ptrs.data(), Lens.data()));
		SetMega(Ctx, Labels, Lens, Group);
		GRPCHK(mpcgpu_group_calc_posteriors(Group));
		}
	else
		{
		GPUCHK(mpcgpu_set_seqs(Ctx, SeqCount, Ptrs.data(), Lens.data()));
		SetMega(Ctx, Labels, Lens);
		GPUCHK(mpcgpu_calc_posteriors(Ctx, 0, PairCount));
		GPUCHK(mpcgpu_build_store(Ctx));
		}

	B.m_Seqs = M.m_MyInputSeqs;
	B.m_PairCount = PairCount;
	B.m_Served = 0;
	B.m_OnHost = false;
	B.m_ItersDone = 0;
	B.m_SeqPtrs.assign(Ptrs.begin(), Ptrs.end());
	B.m_SeqLens.assign(Lens.begin(), Lens.end());
	B.m_SeqEnds.resize(SeqCount);
	for (uint i = 0; i < SeqCount; ++i)
		B.m_SeqEnds[i] = SeqEnds(Ptrs[i], Lens[i]);
	g_Slots[SlotIndex].m_StoreOwner = &M;
	B.m_EA.resize(PairCount);
	GPUCHK(mpcgpu_get_ea(Ctx, 0, PairCount, B.m_EA.data()));
		{
// a run outside the default layout's limits takes a slower relax path: say so once instead of just being slow
		char Info[1024];
		int Fallback = 0;
		if (mpcgpu_relax_info(Ctx, Info, sizeof(Info), &Fallback) == 0 && (Fallback != 0 || TimingOn()))
			fprintf(stderr, "[muscle_gpu] %sconsistency store: %s\n", Fallback ? "NOTE (slower path) " : "", Info);
		}
// MPCFlat::Consistency (mpcflat.cpp:173-181) is skipped for < 3 sequences or 0 iterations: then the
// progressive stage reads the stage-A matrices, so they must exist on the host.
	B.m_Materialise = (SeqCount < 3 || M.m_ConsistencyIterCount == 0);
	if (DebugOn())
		{
		uint64_t h = Fnv(14695981039346656037ull, B.m_EA.data(), 4*size_t(PairCount));
		uint64_t hs = 14695981039346656037ull;
		for (uint i = 0; i < SeqCount; ++i)
			hs = Fnv(hs, Ptrs[i], Lens[i]);
		uint64_t hh = Fnv(14695981039346656037ull, PairHMM::m_StartScore, sizeof(PairHMM::m_StartScore));
		hh = Fnv(hh, PairHMM::m_TransScore, sizeof(PairHMM::m_TransScore));
		hh = Fnv(hh, PairHMM::m_MatchScore, sizeof(PairHMM::m_MatchScore));
		hh = Fnv(hh, PairHMM::m_InsScore, sizeof(PairHMM::m_InsScore));
		fprintf(stderr, "[muscle_gpu] stage A: %u seqs fnv(seqs) %016llx fnv(hmm) %016llx fnv(EA) %016llx\n", SeqCount,
		  (unsigned long long) hs, (unsigned long long) hh, (unsigned long long) h);
		}
	}
} // namespace

// MPCFlat::CalcPosteriors (mpcflat.cpp:239-251): the reference's OpenMP loop over the pairs. Here the first CalcPosterior call of a run
// computes every pair on the device and the others copy their EA, so the loop is plain: an OpenMP team per call bought nothing and cost
// the -super7 shrub workers their cores — each worker thread owns a team of its own, libgomp's idle team threads spin after a region,
// and 8 workers x 15 spinners starved the threads that feed the device (AlignAlns 0.22 ms per call with one worker, 0.35 with eight;
// profiles/r11j, r11l). Same calls in the same order as one thread of the reference's loop makes them.
void MPCFlat::CalcPosteriors()
	{
	const uint PairCount = SIZE(m_Pairs);
	asserta(PairCount > 0);
	PhaseMark(M_POST_ENTER, true);
	for (uint PairIndex = 0; PairIndex < PairCount; ++PairIndex)
		{
		ProgressStep(PairIndex, PairCount, "Calc posteriors");
		CalcPosterior(PairIndex);
		}
	PhaseMark(M_POST_EXIT, false);
	}

void MPCFlat::CalcPosterior(uint PairIndex)
	{
	const pair<uint, uint> &Pair = GetPair(PairIndex);
	const uint SeqIndexX = Pair.first;
	const uint SeqIndexY = Pair.second;
	float EA;
// the content hashes of this pair's two sequences, computed BEFORE the slot's lock is taken (the reference calls CalcPosterior
// from an OpenMP loop: N^2 hashes of L bytes under one mutex were seconds at 10 000 sequences)
	const uint64_t HashX = SeqEnds(GetBytePtr(SeqIndexX), GetSeqLength(SeqIndexX));
	const uint64_t HashY = SeqEnds(GetBytePtr(SeqIndexY), GetSeqLength(SeqIndexY));
		{
		const int SlotIndex = SlotIndexOf(this);
		Slot &S = g_Slots[SlotIndex];
		std::lock_guard<std::mutex> Guard(S.m_Mu);
		Batch &B = BatchOf(this);
		bool Fresh = (B.m_Seqs == m_MyInputSeqs && B.m_PairCount == SIZE(m_Pairs) && B.m_Served < B.m_PairCount &&
		  S.m_StoreOwner == this && SIZE(B.m_SeqPtrs) == GetSeqCount());
		if (Fresh)
			{
// the two sequences of this pair are still the ones the batch was computed from
			const uint Idx[2] = { SeqIndexX, SeqIndexY };
			const uint64_t Hash[2] = { HashX, HashY };
			for (int q = 0; q < 2; ++q)
				{
				const uint i = Idx[q];
				if (B.m_SeqPtrs[i] != GetBytePtr(i) || B.m_SeqLens[i] != GetSeqLength(i) || B.m_SeqEnds[i] != Hash[q])
					Fresh = false;
				}
			}
		if (!Fresh)
			{
			StartBatch(*this, B, SlotIndex);
			if (B.m_Materialise && DownloadOn())
				{
				Download(S.m_Ctx, *this, B.m_PairCount, [this](uint k) -> MySparseMx & { return GetSparsePost(k); });
				B.m_OnHost = true;
				}
			}
		asserta(PairIndex < B.m_PairCount);
		EA = B.m_EA[PairIndex];
		++B.m_Served;
		}
	m_DistMx[SeqIndexX][SeqIndexY] = EA; // calcposteriorflat.cpp:89-91
	m_DistMx[SeqIndexY][SeqIndexX] = EA;
	}

void MPCFlat::ConsIter(uint Iter)
	{
	uint PairCount = SIZE(m_Pairs);
	asserta(PairCount > 0);
	ProgressStep(0, 1, "Consistency (%u/%u)", Iter+1, m_ConsistencyIterCount);
	PhaseMark(M_CONS_ENTER, true);
	struct ExitMark { ~ExitMark() { PhaseMark(M_CONS_EXIT, false); } } MarkAtExit;
	Stopwatch SW(T_CONS_ITER);
		{
		const int SlotIndex = SlotIndexOf(this);
		Slot &S = g_Slots[SlotIndex];
		std::lock_guard<std::mutex> Guard(S.m_Mu);
		mpcgpu_ctx *Ctx = GetCtx(SlotIndex);
		mpcgpu_group *Group = S.m_Group;
		if (S.m_StoreOwner != this)
			Die("GPU posterior stage: ConsIter on an MPCFlat whose posteriors are not the ones on the device");
		Batch &B = BatchOf(this);
		++B.m_ItersDone;
		B.m_OnHost = false;
		if (Group != 0)
			GRPCHK(mpcgpu_group_cons_iter(Group)); // consflat.cpp:5-23 sharded + the all-gather of the new values
		else
			{
			GPUCHK(mpcgpu_cons_iter(Ctx, 0, PairCount));
			GPUCHK(mpcgpu_cons_commit(Ctx));
			}
// Nothing on the host reads the matrices any more (ProgressiveAlign/Refine -> AlignAlns below);
// they are downloaded after the last iteration only on request.
		if (Iter + 1 == m_ConsistencyIterCount && DownloadOn())
			{
			Download(Ctx, *this, PairCount, [this](uint k) -> MySparseMx & { return GetUpdatedSparsePost(k); });
			B.m_OnHost = true; // (they become GetSparsePost after the swap below)
			}
		}
	swap(m_ptrSparsePosts, m_ptrUpdatedSparsePosts); // consflat.cpp:22
	}

// MPCFlat::BuildPost (buildpostflat.cpp:18-106) for its callers outside MPCFlat::Run — cmd_profseq (profseq.cpp:33-49) computes
// a few posteriors and goes straight to BuildPost: the matrix is built on the device store (mpcgpu_build_post: the kernels
// MPCFlat::AlignAlns below uses) and copied into the caller's buffer. There is no host implementation behind it.
namespace
{
// sequence indices and position -> column maps of the rows of an alignment, as the library wants them
void RowsOf(MPCFlat &M, const MultiSequence &MSA, vector<uint32_t> &Seqs, vector<uint32_t> &Map)
	{
	const uint SeqCount = MSA.GetSeqCount();
	Seqs.resize(SeqCount);
	Map.clear();
	vector<uint> PosToCol;
	for (uint i = 0; i < SeqCount; ++i)
		{
		const Sequence *Seq = MSA.GetSequence(i);
		uint SMI = M.GetMyInputSeqIndex(Seq->m_Label);
		asserta(SMI != UINT_MAX);
		Seqs[i] = SMI;
		Seq->GetPosToCol(PosToCol);
		asserta(SIZE(PosToCol) == M.GetSeqLength(SMI));
		Map.insert(Map.end(), PosToCol.begin(), PosToCol.end());
		}
	}
}

void MPCFlat::BuildPost(const MultiSequence &MSA1, const MultiSequence &MSA2, float *Post)
	{
	const uint SeqCount1 = MSA1.GetSeqCount();
	const uint SeqCount2 = MSA2.GetSeqCount();
	const uint ColCount1 = MSA1.GetColCount();
	const uint ColCount2 = MSA2.GetColCount();
	const int SlotIndex = SlotIndexOf(this);
	Slot &S = g_Slots[SlotIndex];
// weights by the ROW index in MSA1 / MSA2 (buildpostflat.cpp:41,52,74); callers outside Run may never have sized m_Weights
	vector<float> W1(SeqCount1, 1.0f), W2(SeqCount2, 1.0f);
	for (uint i = 0; i < SeqCount1 && i < SIZE(m_Weights); ++i)
		W1[i] = m_Weights[i];
	for (uint i = 0; i < SeqCount2 && i < SIZE(m_Weights); ++i)
		W2[i] = m_Weights[i];
	vector<uint32_t> Seqs1, Seqs2, Map1, Map2;
	RowsOf(*this, MSA1, Seqs1, Map1);
	RowsOf(*this, MSA2, Seqs2, Map2);
	std::lock_guard<std::mutex> Guard(S.m_Mu);
	if (S.m_StoreOwner != this) // (checked under the slot's lock: another MPCFlat's batch may be taking the context over)
		Die("GPU posterior stage: BuildPost on an MPCFlat whose posteriors are not the ones on the device");
	mpcgpu_ctx *Ctx = GetCtx(SlotIndex);
	GPUCHK(mpcgpu_build_post(Ctx, SeqCount1, Seqs1.data(), SeqCount2, Seqs2.data(), ColCount1, ColCount2,
	  Map1.data(), Map2.data(), W1.data(), W2.data(), Post));
	}

MultiSequence *MPCFlat::AlignAlns(const MultiSequence &MSA1,
  const MultiSequence &MSA2, float *ptrScore)
	{
	const uint SeqCount1 = MSA1.GetSeqCount();
	const uint SeqCount2 = MSA2.GetSeqCount();
	const uint ColCount1 = MSA1.GetColCount();
	const uint ColCount2 = MSA2.GetColCount();
	PhaseMark(M_ALN_ENTER, true);
	struct ExitMark { ~ExitMark() { PhaseMark(M_ALN_EXIT, false); } } MarkAtExit;

// alnalnsflat.cpp:16-20
	const uint SeqCount = GetSeqCount();
	if (SIZE(m_Weights) != SeqCount)
		m_Weights.assign(SeqCount, 1.0f);
// BuildPost multiplies by w1*w2, the weights looked up by the ROW index in MSA1 / MSA2 (buildpostflat.cpp:41,52,74);
// MPCFlat::Run overwrites every weight with 1.0f (mpcflat.cpp:324), other callers may not: the weights go along.
	vector<float> W1(SeqCount1), W2(SeqCount2);
	for (uint i = 0; i < SeqCount1; ++i)
		{
		asserta(i < SIZE(m_Weights));
		W1[i] = m_Weights[i];
		}
	for (uint i = 0; i < SeqCount2; ++i)
		{
		asserta(i < SIZE(m_Weights));
		W2[i] = m_Weights[i];
		}

	const int SlotIndex = SlotIndexOf(this);
	Slot &S = g_Slots[SlotIndex];
	if (S.m_StoreOwner != this)
		Die("GPU posterior stage: AlignAlns on an MPCFlat whose posteriors are not the ones on the device");
	Stopwatch SW(T_ALN_PREP);
	vector<uint32_t> Seqs1(SeqCount1), Seqs2(SeqCount2);
	vector<uint32_t> Map1, Map2;
	vector<uint> PosToCol;
	for (uint i = 0; i < SeqCount1; ++i)
		{
		const Sequence *Seq = MSA1.GetSequence(i);
		uint SMI = GetMyInputSeqIndex(Seq->m_Label);
		asserta(SMI != UINT_MAX);
		Seqs1[i] = SMI;
		Seq->GetPosToCol(PosToCol);
		asserta(SIZE(PosToCol) == GetSeqLength(SMI));
		Map1.insert(Map1.end(), PosToCol.begin(), PosToCol.end());
		}
	for (uint i = 0; i < SeqCount2; ++i)
		{
		const Sequence *Seq = MSA2.GetSequence(i);
		uint SMI = GetMyInputSeqIndex(Seq->m_Label);
		asserta(SMI != UINT_MAX);
		Seqs2[i] = SMI;
		Seq->GetPosToCol(PosToCol);
		asserta(SIZE(PosToCol) == GetSeqLength(SMI));
		Map2.insert(Map2.end(), PosToCol.begin(), PosToCol.end());
		}

	string Path(ColCount1 + ColCount2, '?');
	uint32_t PathLen = 0;
	float Score = 0;
	SW.Next(T_ALN_LIB);
		{
		std::lock_guard<std::mutex> Guard(S.m_Mu);
		mpcgpu_ctx *Ctx = GetCtx(SlotIndex);
		GPUCHK(mpcgpu_align_alns_w(Ctx, SeqCount1, Seqs1.data(), SeqCount2, Seqs2.data(), ColCount1, ColCount2,
		  Map1.data(), Map2.data(), W1.data(), W2.data(), &Path[0], &PathLen, &Score));
		}
	Path.resize(PathLen);
	if (ptrScore != 0)
		*ptrScore = Score;
	SW.Next(T_ALN_POST);

// alnalnsflat.cpp:36-50
	MultiSequence *result = new MultiSequence();
	for (uint SeqIndex1 = 0; SeqIndex1 < SeqCount1; ++SeqIndex1)
		{
		const Sequence *InputRow = MSA1.GetSequence(SeqIndex1);
		Sequence *AlignedRow = InputRow->AddGapsPath(Path, 'X');
		result->AddSequence(AlignedRow, true);
		}
	for (uint SeqIndex2 = 0; SeqIndex2 < SeqCount2; ++SeqIndex2)
		{
		const Sequence *InputRow = MSA2.GetSequence(SeqIndex2);
		Sequence *AlignedRow = InputRow->AddGapsPath(Path, 'Y');
		result->AddSequence(AlignedRow, true);
		}
	return result;
	}

// MPCFlat::ProgressiveAlign (progalnflat.cpp:72-100) with the joins of one guide-tree LEVEL in one library call. The reference walks
// its N - 1 joins one after the other (ProgAln, progalnflat.cpp:41-70: AlignAlns of two MultiSequence objects, a new one for the
// parent); a join needs only its two children, and at 1000 sequences ~990 of the 999 joins are one-wave kernels with a round trip
// each. Here every node of the tree is carried as its rows' sequence indices and position -> column maps (what the device wants
// anyway: see RefineIter below), the joins are grouped by their height in the tree — a join's children lie on lower levels — and a
// level goes to mpcgpu_align_alns_batch: two launches for its small joins together. A parent's rows are MSA1's then MSA2's
// (alnalnsflat.cpp:36-50), its columns come off the path (B: both advance, X: MSA1, Y: MSA2). Only the root becomes a MultiSequence.
// Falls back to the reference's function when a weight differs from 1.0f or an input sequence carries gap characters.
extern "C" void MPCFlat_ProgressiveAlign_ref(MPCFlat *This); // = the reference's MPCFlat::ProgressiveAlign, renamed at link time (a member function takes `this` first)
void MPCFlat::ProgressiveAlign()
	{
	const uint SeqCount = m_MyInputSeqs->GetSeqCount();
	const uint JoinCount = SeqCount - 1;
	const uint NodeCount = SeqCount + JoinCount;
	asserta(SIZE(m_JoinIndexes1) == JoinCount);
	asserta(SIZE(m_JoinIndexes2) == JoinCount);
	ValidateJoinOrder(m_JoinIndexes1, m_JoinIndexes2);

	bool Plain = true;
	for (uint i = 0; i < SIZE(m_Weights); ++i)
		if (m_Weights[i] != 1.0f)
			Plain = false;
	for (uint i = 0; i < SeqCount && Plain; ++i)
		{
		const Sequence *Seq = m_MyInputSeqs->GetSequence(i);
		for (char ch : Seq->m_CharVec)
			if (ch == '-')
				{ Plain = false; break; }
		}
	const int SlotIndex = SlotIndexOf(this);
	Slot &S = g_Slots[SlotIndex];
	if (!Plain || S.m_StoreOwner != this || JoinCount < 2)
		{
// the reference's own function (progalnflat.cpp:72-100), kept in the binary under another name (hostcxx/build_muscle_gpu.sh: objcopy
// --redefine-sym on progalnflat.o): its joins go through MPCFlat::AlignAlns above, one after the other
		MPCFlat_ProgressiveAlign_ref(this);
		return;
		}

	struct Node
		{
		vector<uint32_t> m_Rows;		// sequence indices (m_MyInputSeqs), in row order
		vector<vector<uint32_t> > m_Maps;	// per row: position -> column
		uint m_ColCount = 0;
		uint m_Level = 0;
		};
	vector<Node> Nodes(NodeCount);
	for (uint i = 0; i < SeqCount; ++i)
		{
		const uint L = GetSeqLength(i);
		Nodes[i].m_Rows.assign(1, i);
		Nodes[i].m_Maps.resize(1);
		Nodes[i].m_Maps[0].resize(L);
		for (uint k = 0; k < L; ++k)
			Nodes[i].m_Maps[0][k] = k;
		Nodes[i].m_ColCount = L;
		}
	uint MaxLevel = 0;
	for (uint j = 0; j < JoinCount; ++j)
		{
		const uint a = m_JoinIndexes1[j], b = m_JoinIndexes2[j];
		asserta(a < SeqCount + j && b < SeqCount + j);
		Nodes[SeqCount + j].m_Level = 1 + std::max(Nodes[a].m_Level, Nodes[b].m_Level);
		MaxLevel = std::max(MaxLevel, Nodes[SeqCount + j].m_Level);
		}
	PhaseMark(M_ALN_ENTER, true);
	struct ExitMark { ~ExitMark() { PhaseMark(M_ALN_EXIT, false); } } MarkAtExit;
	for (uint Level = 1; Level <= MaxLevel; ++Level)
		{
		Stopwatch SW(T_ALN_PREP);
		vector<uint> Joins;
		for (uint j = 0; j < JoinCount; ++j)
			if (Nodes[SeqCount + j].m_Level == Level)
				Joins.push_back(j);
		const uint nj = SIZE(Joins);
		vector<uint32_t> N1(nj), N2(nj), C1(nj), C2(nj), Seqs, Maps;
		uint32_t Stride = 1;
		for (uint q = 0; q < nj; ++q)
			{
			const Node &A = Nodes[m_JoinIndexes1[Joins[q]]], &B = Nodes[m_JoinIndexes2[Joins[q]]];
			N1[q] = SIZE(A.m_Rows); N2[q] = SIZE(B.m_Rows); C1[q] = A.m_ColCount; C2[q] = B.m_ColCount;
			Stride = std::max(Stride, C1[q] + C2[q]);
			Seqs.insert(Seqs.end(), A.m_Rows.begin(), A.m_Rows.end());
			Seqs.insert(Seqs.end(), B.m_Rows.begin(), B.m_Rows.end());
			for (const vector<uint32_t> &M : A.m_Maps) Maps.insert(Maps.end(), M.begin(), M.end());
			for (const vector<uint32_t> &M : B.m_Maps) Maps.insert(Maps.end(), M.begin(), M.end());
			}
		vector<char> Paths((size_t) nj*Stride);
		vector<uint32_t> PathLens(nj);
		SW.Next(T_ALN_LIB);
			{
			std::lock_guard<std::mutex> Guard(S.m_Mu);
			if (S.m_StoreOwner != this)
				Die("GPU posterior stage: ProgressiveAlign on an MPCFlat whose posteriors are not the ones on the device");
			mpcgpu_ctx *Ctx = GetCtx(SlotIndex);
			GPUCHK(mpcgpu_align_alns_batch(Ctx, nj, N1.data(), N2.data(), C1.data(), C2.data(), Seqs.data(), Maps.data(), Stride,
			  Paths.data(), PathLens.data(), 0));
			}
		SW.Next(T_ALN_POST);
		for (uint q = 0; q < nj; ++q)
			{
			Node &A = Nodes[m_JoinIndexes1[Joins[q]]], &B = Nodes[m_JoinIndexes2[Joins[q]]], &P = Nodes[SeqCount + Joins[q]];
			const char *Path = Paths.data() + (size_t) q*Stride;
			const uint PathLen = PathLens[q];
			vector<uint32_t> Lut1(A.m_ColCount), Lut2(B.m_ColCount);
			uint i1 = 0, i2 = 0;
			for (uint Col = 0; Col < PathLen; ++Col)
				{
				const char c = Path[Col];
				if (c == 'B' || c == 'X')
					{ asserta(i1 < A.m_ColCount); Lut1[i1++] = Col; }
				if (c == 'B' || c == 'Y')
					{ asserta(i2 < B.m_ColCount); Lut2[i2++] = Col; }
				}
			asserta(i1 == A.m_ColCount && i2 == B.m_ColCount);
			P.m_ColCount = PathLen;
			P.m_Rows = std::move(A.m_Rows);
			P.m_Rows.insert(P.m_Rows.end(), B.m_Rows.begin(), B.m_Rows.end());
			P.m_Maps = std::move(A.m_Maps);
			for (vector<uint32_t> &M : P.m_Maps)
				for (uint32_t &c : M)
					c = Lut1[c];
			for (vector<uint32_t> &M : B.m_Maps)
				{
				for (uint32_t &c : M)
					c = Lut2[c];
				P.m_Maps.push_back(std::move(M));
				}
			vector<uint32_t>().swap(B.m_Rows);
			vector<vector<uint32_t> >().swap(B.m_Maps);
			}
		}
// the root, as the MultiSequence the rest of MPCFlat::Run reads (Sequence::AddGapsPath, sequence.cpp:115-140, row by row in the reference)
	const Node &Root = Nodes[NodeCount - 1];
	asserta(SIZE(Root.m_Rows) == SeqCount);
	MultiSequence *Result = new MultiSequence();
	for (uint r = 0; r < SeqCount; ++r)
		{
		const Sequence *In = m_MyInputSeqs->GetSequence(Root.m_Rows[r]);
		Sequence *Row = NewSequence();
		Row->m_Label = In->m_Label;
		Row->m_CharVec.assign(Root.m_ColCount, '-');
		const vector<uint32_t> &M = Root.m_Maps[r];
		asserta(SIZE(M) == SIZE(In->m_CharVec));
		for (uint k = 0; k < SIZE(M); ++k)
			Row->m_CharVec[M[k]] = In->m_CharVec[k];
		Result->AddSequence(Row, true);
		}
	m_MSA = Result;
	}

// MPCFlat::RefineIter (refineflat.cpp:4-31; its own translation unit in the reference, not linked here) on POSITION -> COLUMN MAPS.
// The reference draws a bipartition of the rows with rand(), projects m_MSA onto either part (MultiSequence::Project, project.cpp:16-67:
// a new character matrix per part, all-gap columns dropped), aligns the two with AlignAlns and joins them along the path
// (Sequence::AddGapsPath per row). At 1000 rows x 600 columns that is ~9 ms of host work around an 8 ms device join, 100 times:
// profiles/r12d_align_timeline.log has 0.64 s between the AlignAlns calls + 0.29 s inside them on maps and the result MSA.
// The alignment is carried here as what the device wants anyway — per row its sequence and Sequence::GetPosToCol (sequence.cpp:144-154):
//   projection    = rank of a column among the columns any row of the part occupies;
//   join          = column of MSA1 / MSA2 column c in the merged alignment, read off the path (B: both advance, X: MSA1, Y: MSA2);
//   rows          = the rows of part 1 in ascending row order, then those of part 2 (refineflat.cpp:13-17, alnalnsflat.cpp:36-50),
// O(residues) per round; m_MSA is rebuilt from the maps after every round (a character matrix filled once), so it is valid whenever
// anybody looks. The rand() calls are the reference's, one per row, in row order; the numeric part is mpcgpu_align_alns_w as in
// MPCFlat::AlignAlns above. State is kept per MPCFlat object and checked against m_MSA's address and the posterior batch it belongs to.
namespace
{
struct RefineState
	{
	const MultiSequence *m_MSA = 0;
	unsigned long long m_BatchGen = 0;
	uint m_ColCount = 0;
	vector<uint32_t> m_Seq;			// row -> index in m_MyInputSeqs
	vector<vector<uint32_t> > m_Map;	// row -> position -> column
	vector<string> m_Letters;		// row -> its residues (the non-gap characters of the row)
	};
std::map<const MPCFlat *, RefineState> g_Refine;	// (guarded by g_MapMu)
}

void MPCFlat::RefineIter()
	{
	const uint SeqCount = GetSeqCount();
	asserta(m_MSA != 0);
	asserta(m_MSA->GetSeqCount() == SeqCount);
	const int SlotIndex = SlotIndexOf(this);
	Slot &S = g_Slots[SlotIndex];
	RefineState *St;
	unsigned long long Gen;
		{
		std::lock_guard<std::mutex> Guard(g_MapMu);
		St = &g_Refine[this];
		}
		{
		std::lock_guard<std::mutex> Guard(S.m_Mu);
		Gen = BatchOf(this).m_Gen;
		}
	if (St->m_MSA != m_MSA || St->m_BatchGen != Gen || SIZE(St->m_Seq) != SeqCount)
		{
// (first round of a run, or somebody else put another alignment into m_MSA: the maps are read off it)
		St->m_MSA = m_MSA;
		St->m_BatchGen = Gen;
		St->m_ColCount = m_MSA->GetColCount();
		St->m_Seq.resize(SeqCount);
		St->m_Map.resize(SeqCount);
		St->m_Letters.resize(SeqCount);
		vector<uint> PosToCol;
		for (uint i = 0; i < SeqCount; ++i)
			{
			const Sequence *Row = m_MSA->GetSequence(i);
			const uint SMI = GetMyInputSeqIndex(Row->m_Label);
			asserta(SMI != UINT_MAX);
			St->m_Seq[i] = SMI;
			Row->GetPosToCol(PosToCol);
			asserta(SIZE(PosToCol) == GetSeqLength(SMI));
			St->m_Map[i].assign(PosToCol.begin(), PosToCol.end());
			string &L = St->m_Letters[i];
			L.clear();
			for (uint k = 0; k < SIZE(PosToCol); ++k)
				L.push_back(Row->m_CharVec[PosToCol[k]]);
			}
		}

// create two separate groups (refineflat.cpp:12-17: one rand() per row, in row order)
	vector<uint> Rows1, Rows2;
	for (uint SeqIndex = 0; SeqIndex < SeqCount; SeqIndex++)
		if (rand()%2 == 0)
			Rows1.push_back(SeqIndex);
		else
			Rows2.push_back(SeqIndex);
	if (Rows1.empty() || Rows2.empty())
		return;

	if (S.m_StoreOwner != this)
		Die("GPU posterior stage: RefineIter on an MPCFlat whose posteriors are not the ones on the device");
	PhaseMark(M_ALN_ENTER, true);
	struct ExitMark { ~ExitMark() { PhaseMark(M_ALN_EXIT, false); } } MarkAtExit;
	Stopwatch SW(T_ALN_PREP);
	const uint OldColCount = St->m_ColCount;
// MultiSequence::Project (project.cpp:40-64): the columns that are not all gaps keep their order
	vector<uint32_t> Rank1(OldColCount, 0), Rank2(OldColCount, 0);
	auto Project = [&](const vector<uint> &Rows, vector<uint32_t> &Rank, vector<uint32_t> &Seqs, vector<uint32_t> &Map) -> uint
		{
		for (uint r : Rows)
			for (uint32_t c : St->m_Map[r])
				Rank[c] = 1;
		uint n = 0;
		for (uint c = 0; c < OldColCount; ++c)
			{
			const uint Occupied = Rank[c];
			Rank[c] = n;
			n += Occupied;
			}
		Seqs.clear(); Map.clear();
		for (uint r : Rows)
			{
			Seqs.push_back(St->m_Seq[r]);
			for (uint32_t c : St->m_Map[r])
				Map.push_back(Rank[c]);
			}
		return n;
		};
	vector<uint32_t> Seqs1, Seqs2, Map1, Map2;
	const uint ColCount1 = Project(Rows1, Rank1, Seqs1, Map1);
	const uint ColCount2 = Project(Rows2, Rank2, Seqs2, Map2);
// alnalnsflat.cpp:16-20, buildpostflat.cpp:41,52,74: the weights by ROW index in MSA1 / MSA2
	if (SIZE(m_Weights) != SeqCount)
		m_Weights.assign(SeqCount, 1.0f);
	vector<float> W1(m_Weights.begin(), m_Weights.begin() + SIZE(Rows1)), W2(m_Weights.begin(), m_Weights.begin() + SIZE(Rows2));

	string Path(ColCount1 + ColCount2, '?');
	uint32_t PathLen = 0;
	float Score = 0;
	SW.Next(T_ALN_LIB);
		{
		std::lock_guard<std::mutex> Guard(S.m_Mu);
		mpcgpu_ctx *Ctx = GetCtx(SlotIndex);
		GPUCHK(mpcgpu_align_alns_w(Ctx, SIZE(Seqs1), Seqs1.data(), SIZE(Seqs2), Seqs2.data(), ColCount1, ColCount2,
		  Map1.data(), Map2.data(), W1.data(), W2.data(), &Path[0], &PathLen, &Score));
		}
	Path.resize(PathLen);
	SW.Next(T_ALN_POST);
// the join (alnalnsflat.cpp:36-50, Sequence::AddGapsPath sequence.cpp:115-140): column of either alignment's column in the merged one
	vector<uint32_t> Lut1(ColCount1), Lut2(ColCount2);
		{
		uint i1 = 0, i2 = 0;
		for (uint Col = 0; Col < PathLen; ++Col)
			{
			const char c = Path[Col];
			if (c == 'B' || c == 'X')
				{ asserta(i1 < ColCount1); Lut1[i1++] = Col; }
			if (c == 'B' || c == 'Y')
				{ asserta(i2 < ColCount2); Lut2[i2++] = Col; }
			}
		asserta(i1 == ColCount1 && i2 == ColCount2);
		}
	RefineState Next;
	Next.m_BatchGen = Gen;
	Next.m_ColCount = PathLen;
	MultiSequence *Result = new MultiSequence();
	auto Emit = [&](const vector<uint> &Rows, const vector<uint32_t> &Rank, const vector<uint32_t> &Lut)
		{
		for (uint r : Rows)
			{
			vector<uint32_t> &M = St->m_Map[r];
			for (uint32_t &c : M)
				c = Lut[Rank[c]];
			Sequence *Row = NewSequence();
			Row->m_Label = m_MSA->GetSequence(r)->m_Label;
			Row->m_CharVec.assign(PathLen, '-');
			const string &L = St->m_Letters[r];
			for (uint k = 0; k < SIZE(M); ++k)
				Row->m_CharVec[M[k]] = L[k];
			Result->AddSequence(Row, true);
			Next.m_Seq.push_back(St->m_Seq[r]);
			Next.m_Map.push_back(std::move(M));
			Next.m_Letters.push_back(std::move(St->m_Letters[r]));
			}
		};
	Emit(Rows1, Rank1, Lut1);
	Emit(Rows2, Rank2, Lut2);
	delete m_MSA;
	m_MSA = Result;
	Next.m_MSA = m_MSA;
	*St = std::move(Next);
	}

namespace
{
// PProg::AlignMSAsFlat (alnmsasflat.cpp:4-50) behind its pair sampling: the pairs are given. Split off so that PProg::Run2 below can
// draw every join's pairs in join order on one thread (GetPairs consumes the process-wide randu32 stream) and run the joins side by side.
float AlignMSAsWithPairs(const string &ProgressStr, const MultiSequence &MSA1, const MultiSequence &MSA2,
  const vector<uint> &SeqIndexes1, const vector<uint> &SeqIndexes2, string &Path, bool Progress)
	{
	const uint SeqCount1 = MSA1.GetNumSequences();
	const uint SeqCount2 = MSA2.GetNumSequences();
	asserta(MSA1.IsAligned());
	asserta(MSA2.IsAligned());
	const uint ColCount1 = MSA1.GetColCount();
	const uint ColCount2 = MSA2.GetColCount();
	const uint PairCount = SIZE(SeqIndexes1);
	asserta(SIZE(SeqIndexes2) == PairCount);
	asserta(PairCount > 0);
	if (Progress) // (ProgressStep keeps its state in unguarded process globals: the worker threads of Run2 run without it)
		ProgressStep(0, 1, "%s [%u x %u, %u pairs]", ProgressStr.substr(0, 20).c_str(),
		  min(SeqCount1, SeqCount2), max(SeqCount1, SeqCount2), PairCount);

// The ungapped sequences come from the global input registry by label, like CalcPost does
// (calcpost.cpp:4-36, getpostpairsalignedflat.cpp:43-46); the ones this join touches are handed to
// the library as its registry for this call.
	Stopwatch SW(T_JOIN_PREP);
	std::map<const Sequence *, uint32_t> SeqToIndex;
	vector<const uint8_t *> Ptrs;
	vector<uint32_t> Lens;
	vector<string> Labels;
	auto Register = [&](const string &Label) -> uint32_t
		{
		const Sequence &Seq = GetGlobalInputSeqByLabel(Label);
		std::map<const Sequence *, uint32_t>::const_iterator p = SeqToIndex.find(&Seq);
		if (p != SeqToIndex.end())
			return p->second;
		uint32_t Index = (uint32_t) Ptrs.size();
		SeqToIndex[&Seq] = Index;
		Ptrs.push_back(Seq.GetBytePtr());
		Lens.push_back(Seq.GetLength());
		Labels.push_back(Label);
		return Index;
		};

	vector<uint32_t> Seqs1(PairCount), Seqs2(PairCount);
	vector<uint32_t> Map1, Map2;
	vector<uint> PosToCol;
	for (uint PairIndex = 0; PairIndex < PairCount; ++PairIndex)
		{
		const uint SeqIndex1 = SeqIndexes1[PairIndex];
		const uint SeqIndex2 = SeqIndexes2[PairIndex];
		asserta(SeqIndex1 < SeqCount1);
		asserta(SeqIndex2 < SeqCount2);
		Seqs1[PairIndex] = Register(MSA1.GetLabelStr(SeqIndex1));
		Seqs2[PairIndex] = Register(MSA2.GetLabelStr(SeqIndex2));
// buildposterior3flat.cpp:46-66
		const Sequence *Row1 = MSA1.GetSequence(SeqIndex1);
		const Sequence *Row2 = MSA2.GetSequence(SeqIndex2);
		asserta(Row1->GetLength() == ColCount1);
		asserta(Row2->GetLength() == ColCount2);
		Row1->GetPosToCol(PosToCol);
		asserta(SIZE(PosToCol) == Lens[Seqs1[PairIndex]]);
		Map1.insert(Map1.end(), PosToCol.begin(), PosToCol.end());
		Row2->GetPosToCol(PosToCol);
		asserta(SIZE(PosToCol) == Lens[Seqs2[PairIndex]]);
		Map2.insert(Map2.end(), PosToCol.begin(), PosToCol.end());
		}

	Path.assign(ColCount1 + ColCount2, '?');
	uint32_t PathLen = 0;
	float Score = 0;
	vector<float> EA(PairCount);
	SW.Next(T_JOIN_LIB);
		{
		JoinCtx &J = JoinOfThisThread();
		std::lock_guard<std::mutex> Guard(J.m_Mu);
		mpcgpu_ctx *Ctx = J.m_Ctx;
		GPUCHK(mpcgpu_set_hmm(Ctx, PairHMM::m_StartScore, &PairHMM::m_TransScore[0][0],
		  &PairHMM::m_MatchScore[0][0], PairHMM::m_InsScore, MIN_SPARSE_SCORE, -1));
		GPUCHK(mpcgpu_set_seqs_registry(Ctx, (uint32_t) Ptrs.size(), Ptrs.data(), Lens.data()));
		SetMega(Ctx, Labels, Lens);
		GPUCHK(mpcgpu_align_msas(Ctx, PairCount, Seqs1.data(), Seqs2.data(), ColCount1, ColCount2,
		  Map1.data(), Map2.data(), &Path[0], &PathLen, &Score, EA.data()));
		}
	Path.resize(PathLen);

// getpostpairsalignedflat.cpp:90-96: the reference adds the EAs in thread-arrival order; here in pair order
	float SumEA = 0;
	for (uint PairIndex = 0; PairIndex < PairCount; ++PairIndex)
		SumEA += EA[PairIndex];
	return SumEA/PairCount;
	}
}

float PProg::AlignMSAsFlat(const string &ProgressStr,
  const MultiSequence &MSA1, const MultiSequence &MSA2,
  uint TargetPairCount, string &Path)
	{
// alnmsasflat.cpp:8-25
	const uint SeqCount1 = MSA1.GetNumSequences();
	const uint SeqCount2 = MSA2.GetNumSequences();
	asserta(SeqCount1 > 0);
	asserta(SeqCount2 > 0);
	vector<uint> SeqIndexes1;
	vector<uint> SeqIndexes2;
	GetPairs(SeqCount1, SeqCount2, TargetPairCount, SeqIndexes1, SeqIndexes2);
	return AlignMSAsWithPairs(ProgressStr, MSA1, MSA2, SeqIndexes1, SeqIndexes2, Path, true);
	}

// PProg::Run2 (pprog2.cpp:58-76): the joins of a guide tree in join order, one after the other in the reference. A join needs its two
// children and nothing else, so the joins of different subtrees are independent — and one join is a few milliseconds of latency
// (stage A on <= 2000 sampled pairs, one BuildPost, one alignment; the 411 joins of a 10 000-sequence -super7 run: 2.2 of its 7.7 s).
// Here MUSCLE_GPU_JOIN_WORKERS threads (default 6; 1 = the reference's loop) take ready joins, lowest join index first, each on a join
// context of its own. What the sequential order decides besides the result is kept: every join's pair sample is drawn on THIS thread
// in join order before any join runs (GetPairs, getpairs.cpp:33-69, draws from the process-wide randu32 stream; the sequence counts it
// needs follow from the tree), m_JoinMSAIndexes1/2 are filled in join order, and a join does what PProg::AlignAndJoin
// (pprog2.cpp:7-56) does. With -savedir (a file per join, named by m_JoinIndex) the reference's loop runs.
void PProg::Run2(const vector<uint> &Indexes1,
  const vector<uint> &Indexes2)
	{
	asserta(m_InputMSACount > 0);
	m_JoinCount = m_InputMSACount - 1;
	m_NodeCount = m_InputMSACount + m_JoinCount;
	asserta(SIZE(Indexes1) == m_JoinCount);
	asserta(SIZE(Indexes2) == m_JoinCount);
	ValidateJoinOrder(Indexes1, Indexes2);

	uint Workers = 6;
	const char *EnvWorkers = getenv("MUSCLE_GPU_JOIN_WORKERS");
	if (EnvWorkers != 0 && *EnvWorkers != 0)
		Workers = (uint) atoi(EnvWorkers);
	if (Workers > MAX_JOIN_CTX)
		Workers = MAX_JOIN_CTX;
	if (Workers <= 1 || m_JoinCount < 2 || optset_savedir)
		{
		for (m_JoinIndex = 0; m_JoinIndex < m_JoinCount; ++m_JoinIndex)
			AlignAndJoin(Indexes1[m_JoinIndex], Indexes2[m_JoinIndex]);
		return;
		}

// sequence counts of every node, and every join's pairs, in join order
	vector<uint> SeqCounts(m_NodeCount, 0);
	for (uint i = 0; i < m_InputMSACount; ++i)
		SeqCounts[i] = GetMSA(i).GetNumSequences();
	vector<vector<uint> > Pairs1(m_JoinCount), Pairs2(m_JoinCount);
	for (uint k = 0; k < m_JoinCount; ++k)
		{
		const uint n1 = SeqCounts[Indexes1[k]], n2 = SeqCounts[Indexes2[k]];
		asserta(n1 > 0 && n2 > 0);
		SeqCounts[m_InputMSACount + k] = n1 + n2;
		GetPairs(n1, n2, m_TargetPairCount, Pairs1[k], Pairs2[k]);
		m_JoinMSAIndexes1.push_back(Indexes1[k]);
		m_JoinMSAIndexes2.push_back(Indexes2[k]);
		}

// which join waits for which: join k produces node m_InputMSACount + k
	vector<int> Waiting(m_JoinCount, 0);
	vector<vector<uint> > Users(m_NodeCount);
	for (uint k = 0; k < m_JoinCount; ++k)
		{
		const uint Ins[2] = { Indexes1[k], Indexes2[k] };
		for (int q = 0; q < 2; ++q)
			if (Ins[q] >= m_InputMSACount)
				{
				++Waiting[k];
				Users[Ins[q]].push_back(k);
				}
		}
	std::mutex Mu;
	std::condition_variable Cv;
	std::set<uint> Ready;
	for (uint k = 0; k < m_JoinCount; ++k)
		if (Waiting[k] == 0)
			Ready.insert(k);
	uint Done = 0;
	const bool SavedQuiet = opt_quiet;
	ProgressLog("Joining %u alignments on %u join contexts\n", m_InputMSACount, Workers);
	opt_quiet = true; // (ProgressStep is not thread-safe: as for the shrub workers of Super7::IntraAlignShrubs)
	g_JoinCtxWanted.store(std::max<unsigned>(g_JoinCtxWanted.load(), Workers));
	vector<std::thread> Threads;
	for (uint w = 0; w < Workers; ++w)
		Threads.emplace_back([&, w]()
			{
			t_JoinIndex = (int) w;
			for (;;)
				{
				uint k;
					{
					std::unique_lock<std::mutex> Lock(Mu);
					Cv.wait(Lock, [&] { return !Ready.empty() || Done == m_JoinCount; });
					if (Ready.empty())
						return;
					k = *Ready.begin();
					Ready.erase(Ready.begin());
					}
// PProg::AlignAndJoin (pprog2.cpp:7-56) for join k
				const uint Index1 = Indexes1[k], Index2 = Indexes2[k];
				const MultiSequence &MSA1 = GetMSA(Index1);
				const MultiSequence &MSA2 = GetMSA(Index2);
				AssertSameLabels(MSA1);
				AssertSameLabels(MSA2);
				string ProgressStr;
				Ps(ProgressStr, "Join %u / %u", k+1, m_JoinCount);
				string Path;
				AlignMSAsWithPairs(ProgressStr, MSA1, MSA2, Pairs1[k], Pairs2[k], Path, false);
				MultiSequence *MSA12 = new MultiSequence;
				AlignMSAsByPath(MSA1, MSA2, Path, *MSA12);
				AssertSeqsEq(MSA1, *MSA12);
				AssertSeqsEq(MSA2, *MSA12);
				AssertSameSeqsJoin(MSA1, MSA2, *MSA12);
				AssertSameLabels(*MSA12);
				vector<uint>().swap(Pairs1[k]);
				vector<uint>().swap(Pairs2[k]);
					{
					std::lock_guard<std::mutex> Lock(Mu);
					if (Index1 >= m_InputMSACount)
						{
						delete &MSA1;
						m_MSAs[Index1] = 0;
						}
					if (Index2 >= m_InputMSACount)
						{
						delete &MSA2;
						m_MSAs[Index2] = 0;
						}
					const uint NewMSAIndex = m_InputMSACount + k;
					SetMSA(NewMSAIndex, *MSA12);
					for (size_t u = 0; u < Users[NewMSAIndex].size(); ++u)
						if (--Waiting[Users[NewMSAIndex][u]] == 0)
							Ready.insert(Users[NewMSAIndex][u]);
					++Done;
					}
				Cv.notify_all();
				}
			});
	for (size_t t = 0; t < Threads.size(); ++t)
		Threads[t].join();
	opt_quiet = SavedQuiet;
	m_JoinIndex = m_JoinCount;
	}

// AlignPairFlat (alignpairflat.cpp:3-27) and its sequential caller in -super5, UClust::Search (uclust.cpp:26-56): the same
// kernels as MPCFlat::CalcPosterior + CalcAlnFlat with a third caller. One pair on the device is one wavefront's work (a few
// milliseconds of latency), so what pays is the LIST: UClust::Search tries up to MAX_REJECTS = 8 word-count hits one after the
// other and stops at the first whose EA reaches the threshold — here all of them are aligned in one library call and the first
// that qualifies, in the same order, is returned with its path: the same answer, 8 alignments of latency folded into one.
namespace
{
// One caller's request: a list of (Label1, Label2), answered with paths and EAs (+ the FromPost matrix of its first pair).
struct PairReq
	{
	const vector<string> *m_Labels1 = 0;
	const vector<string> *m_Labels2 = 0;
	vector<string> m_Paths;
	vector<float> m_EAs;
	MySparseMx *m_SparsePost0 = 0;
	bool m_Done = false;
	};

// Runs the requests of one batch as ONE library call on the calling thread's join context (locked by the caller).
void RunPairBatch(const vector<PairReq *> &Batch, mpcgpu_ctx *Ctx)
	{
	std::map<const Sequence *, uint32_t> SeqToIndex;
	vector<const uint8_t *> Ptrs;
	vector<uint32_t> Lens;
	vector<string> Labels;
	auto Register = [&](const string &Label) -> uint32_t
		{
		const Sequence &Seq = GetGlobalInputSeqByLabel(Label); // calcpost.cpp:6-7,25-26 look the sequences up by global label
		std::map<const Sequence *, uint32_t>::const_iterator p = SeqToIndex.find(&Seq);
		if (p != SeqToIndex.end())
			return p->second;
		uint32_t Index = (uint32_t) Ptrs.size();
		SeqToIndex[&Seq] = Index;
		Ptrs.push_back(Seq.GetBytePtr());
		Lens.push_back(Seq.GetLength());
		Labels.push_back(Label);
		return Index;
		};
	vector<uint32_t> Seqs1, Seqs2;
	vector<uint> First; // first pair of every request in the batch's list
	uint32_t Stride = 1;
	for (size_t r = 0; r < Batch.size(); ++r)
		{
		const PairReq &R = *Batch[r];
		First.push_back(SIZE(Seqs1));
		for (uint i = 0; i < SIZE(*R.m_Labels1); ++i)
			{
			Seqs1.push_back(Register((*R.m_Labels1)[i]));
			Seqs2.push_back(Register((*R.m_Labels2)[i]));
			Stride = std::max(Stride, Lens[Seqs1.back()] + Lens[Seqs2.back()]);
			}
		}
	const uint PairCount = SIZE(Seqs1);
	asserta(PairCount > 0);
	vector<char> PathBuf(size_t(PairCount)*Stride);
	vector<uint32_t> PathLens(PairCount);
	vector<float> EAs(PairCount);
	Stopwatch SW(T_PAIRS_PREP);
	GPUCHK(mpcgpu_set_hmm(Ctx, PairHMM::m_StartScore, &PairHMM::m_TransScore[0][0],
	  &PairHMM::m_MatchScore[0][0], PairHMM::m_InsScore, MIN_SPARSE_SCORE, -1));
	GPUCHK(mpcgpu_set_seqs_registry(Ctx, (uint32_t) Ptrs.size(), Ptrs.data(), Lens.data()));
	SetMega(Ctx, Labels, Lens);
	SW.Next(T_PAIRS_LIB);
	GPUCHK(mpcgpu_align_pairs(Ctx, PairCount, Seqs1.data(), Seqs2.data(), Stride, PathBuf.data(), PathLens.data(), 0, EAs.data()));
	for (size_t r = 0; r < Batch.size(); ++r)
		{
		PairReq &R = *Batch[r];
		const uint n = SIZE(*R.m_Labels1);
		R.m_Paths.resize(n);
		R.m_EAs.resize(n);
		for (uint i = 0; i < n; ++i)
			{
			const uint q = First[r] + i;
			R.m_Paths[i].assign(PathBuf.data() + size_t(q)*Stride, PathLens[q]);
			R.m_EAs[i] = EAs[q];
			}
		if (R.m_SparsePost0 != 0)
			{
// AlignPairFlat_SparsePost: SparsePost->FromPost(Post, L1, L2) (alignpairflat.cpp:12-13)
			const uint q = First[r];
			asserta(PairCount <= 256); // mpcgpu_get_list_sparse addresses the pairs of ONE stage of the library
			uint32_t NNZ = 0;
			GPUCHK(mpcgpu_get_list_sparse(Ctx, q, &NNZ, 0, 0));
			const uint LX = Lens[Seqs1[q]], LY = Lens[Seqs2[q]];
			MySparseMx &Mx = *R.m_SparsePost0;
			Mx.AllocLX(LX);
			Mx.AllocVec(NNZ);
			Mx.m_LX = LX;
			Mx.m_LY = LY;
			Mx.m_VecSize = NNZ;
			vector<uint64_t> Vals(size_t(NNZ) + 1);
			GPUCHK(mpcgpu_get_list_sparse(Ctx, q, &NNZ, Mx.m_Offsets, Vals.data()));
			memcpy(Mx.m_ValueVec, Vals.data(), 8*size_t(NNZ));
			}
		}
	}

// The callers of AlignPairFlat inside OpenMP loops (eadistmx.cpp:32-66, eacluster.cpp:114, eesort.cpp:37) ask for one pair each
// from many threads at once. One pair is one wavefront of work, so the requests are COMBINED: a thread that finds nobody
// serving becomes the server, takes everything queued (its own request included) as one library call, and goes on until the
// queue is empty; the others wait for their answers. A lone caller (uclust.cpp's sequential loop) is served at once.
std::mutex g_PairQueueMu;
std::condition_variable g_PairQueueCv;
vector<PairReq *> g_PairQueue;
bool g_PairServing = false;

void AlignPairsByLabel(const vector<string> &Labels1, const vector<string> &Labels2, vector<string> &Paths, vector<float> &EAs,
  MySparseMx *SparsePost0)
	{
	asserta(SIZE(Labels2) == SIZE(Labels1) && !Labels1.empty());
	PairReq Mine;
	Mine.m_Labels1 = &Labels1;
	Mine.m_Labels2 = &Labels2;
	Mine.m_SparsePost0 = SparsePost0;
	std::unique_lock<std::mutex> Lock(g_PairQueueMu);
	g_PairQueue.push_back(&Mine);
	while (!Mine.m_Done)
		{
		if (g_PairServing)
			{
			g_PairQueueCv.wait(Lock);
			continue;
			}
		g_PairServing = true;
		while (!g_PairQueue.empty())
			{
			vector<PairReq *> Batch;
			uint Pairs = 0;
			while (!g_PairQueue.empty() && Pairs + SIZE(*g_PairQueue.front()->m_Labels1) <= 256)
				{
				Pairs += SIZE(*g_PairQueue.front()->m_Labels1);
				Batch.push_back(g_PairQueue.front());
				g_PairQueue.erase(g_PairQueue.begin());
				}
			if (Batch.empty()) // one request of more than 256 pairs: alone (the library cuts it into stages itself)
				{
				Batch.push_back(g_PairQueue.front());
				g_PairQueue.erase(g_PairQueue.begin());
				}
			Lock.unlock();
				{
				JoinCtx &J = JoinOfThisThread();
				std::lock_guard<std::mutex> Guard(J.m_Mu);
				RunPairBatch(Batch, J.m_Ctx);
				}
			Lock.lock();
			for (size_t r = 0; r < Batch.size(); ++r)
				Batch[r]->m_Done = true;
			g_PairQueueCv.notify_all();
			}
		g_PairServing = false;
		g_PairQueueCv.notify_all();
		}
	Paths.swap(Mine.m_Paths);
	EAs.swap(Mine.m_EAs);
	}
}

float AlignPairFlat_SparsePost(const string &Label1, const string &Label2, string &Path, MySparseMx *SparsePost)
	{
	vector<string> L1(1, Label1), L2(1, Label2), Paths;
	vector<float> EAs;
	AlignPairsByLabel(L1, L2, Paths, EAs, SparsePost);
	Path = Paths[0];
	return EAs[0];
	}

float AlignPairFlat(const string &Label1, const string &Label2, string &Path)
	{
	return AlignPairFlat_SparsePost(Label1, Label2, Path, 0);
	}

uint UClust::Search(uint SeqIndex, string &Path)
	{
// uclust.cpp:26-41
	const Sequence *Seq = m_InputSeqs->GetSequence(SeqIndex);
	const byte *ByteSeq = Seq->GetBytePtr();
	const uint L = Seq->GetLength();
	vector<uint> TopSeqIndexes;
	vector<uint> TopWordCounts;
	m_US.SearchSeq(ByteSeq, L, TopSeqIndexes, TopWordCounts);
	uint TopCount = SIZE(TopSeqIndexes);
	asserta(SIZE(TopWordCounts) == TopCount);
	if (TopCount == 0)
		return UINT_MAX;
	if (TopCount > MAX_REJECTS)
		TopCount = MAX_REJECTS;
// uclust.cpp:44-54, all candidates at once
	vector<string> L1, L2, Paths;
	for (uint TopIndex = 0; TopIndex < TopCount; ++TopIndex)
		{
		L1.push_back(m_InputSeqs->GetSequence(SeqIndex)->m_Label);
		L2.push_back(m_InputSeqs->GetSequence(TopSeqIndexes[TopIndex])->m_Label);
		}
	vector<float> EAs;
	AlignPairsByLabel(L1, L2, Paths, EAs, 0);
	for (uint TopIndex = 0; TopIndex < TopCount; ++TopIndex)
		{
		Path = Paths[TopIndex]; // the sequential loop leaves the last path it computed in Path
		if (EAs[TopIndex] >= m_MinEA)
			return TopSeqIndexes[TopIndex];
		}
	return UINT_MAX;
	}

// Super7::IntraAlignShrubs (super7.cpp:127-137): one MPCFlat::Run per shrub of <= shrub_size sequences. The reference runs
// them one after the other; a shrub is far too small to fill a GPU (496 pairs, then 131 small alignments-of-alignments whose
// cost is launch and synchronisation latency), and the shrubs are independent. Here MUSCLE_GPU_SHRUB_CONTEXTS worker threads
// (default 16 — 8 until MPCFlat::CalcPosteriors above stopped spawning an OpenMP team per worker: profiles/r11m —; 1 = the reference's
// loop) take shrubs from a counter, each with its own MPCFlat object and its own device
// context (dealt round-robin over MUSCLE_GPU_DEVICES), so the small launches of different shrubs overlap and several GPUs
// share the shrubs. Results land in m_ShrubMSAs by shrub index; the one process-wide input of MPCFlat::Run that depends on
// the order of the shrubs, the rand() stream of RefineIter (refineflat.cpp:14: one draw per sequence per refinement round,
// Derep is off in -super7: super7.cpp:11), is positioned per shrub where the sequential loop would have had it.
// (.mega inputs: Super7_mega::IntraAlignShrub differs only in the MPCFlat_mega object — the worker threads make one each.)
void Super7::IntraAlignShrubs()
	{
	asserta(m_ShrubMSAs.empty());
	const uint ShrubCount = GetShrubCount();
	uint Workers = 16;
	const char *EnvWorkers = getenv("MUSCLE_GPU_SHRUB_CONTEXTS");
	if (EnvWorkers != 0 && *EnvWorkers != 0)
		Workers = (uint) atoi(EnvWorkers);
	if (Workers > MAX_SLOTS - 1)
		Workers = MAX_SLOTS - 1;
	if (Workers > ShrubCount)
		Workers = ShrubCount;
	const bool IsMega = dynamic_cast<Super7_mega *>(this) != 0; // Super7_mega::IntraAlignShrub (super7_mega.cpp:9-24) = the same with an MPCFlat_mega
	if (Workers <= 1)
		{
		for (uint ShrubIndex = 0; ShrubIndex < ShrubCount; ++ShrubIndex)
			{
			ProgressLog("Aligning shrub %u / %u\n", ShrubIndex+1, ShrubCount);
			IntraAlignShrub(ShrubIndex);
			}
		return;
		}

// where the sequential loop's rand() stream stands when shrub k starts: the shared generator is advanced ONCE past all
// shrubs (linear in the number of draws) and its state is snapshotted at every shrub's first draw
	vector<unsigned long long> RandDraws(ShrubCount, 0);
	for (uint ShrubIndex = 0; ShrubIndex < ShrubCount; ++ShrubIndex)
		{
		vector<uint> LeafNodes;
		m_GuideTree->GetSubtreeLeafNodes(m_ShrubLCAs[ShrubIndex], LeafNodes);
		const unsigned long long n = SIZE(LeafNodes);
		RandDraws[ShrubIndex] = (n >= 3) ? n*m_MPC->m_RefineIterCount : 0; // mpcflat.cpp:254-264
		}
	vector<MuscleGpuRandSnapshot> RandAt(ShrubCount);
	MuscleGpuRandSharedSnapshots(RandDraws.data(), ShrubCount, RandAt.data());

	const double ShrubsBegin = std::chrono::duration<double>(std::chrono::steady_clock::now() - g_ProcessStart).count();
	m_ShrubMSAs.assign(ShrubCount, (const MultiSequence *) 0);
// MPCFlat::Run reports through ProgressStep / Progress (myutils.cpp:1542-1870), which keep their state in unguarded process
// globals (g_ProgressDesc, g_StepCalls, g_CountsInterval — the latter passes through 0, and another thread's
// `g_StepCalls % g_CountsInterval` would then trap). The worker threads therefore run quiet; the shrub count is reported
// from here, as the sequential loop's "Aligning shrub i / N" lines would have.
	const bool SavedQuiet = opt_quiet;
	ProgressLog("Aligning %u shrubs on %u device contexts\n", ShrubCount, Workers);
	opt_quiet = true;
	std::atomic<uint> Next(0);
	vector<std::thread> Threads;
	for (uint w = 0; w < Workers; ++w)
		Threads.emplace_back([&, w]()
			{
// (.mega inputs: the profiles are process-wide statics found by label, Mega::GetProfileByLabel — read-only here — and which emissions
// a run takes is decided by Mega::m_Loaded (calcpost.cpp:14-22, SetMega above); the object is an MPCFlat_mega as in the reference's loop)
// (two owners of exact type: ~MPCFlat is not virtual, mpcflat.h:52 — deleting an MPCFlat_mega through an MPCFlat pointer is undefined)
			std::unique_ptr<MPCFlat_mega> LocalMega(IsMega ? new MPCFlat_mega : 0);
			std::unique_ptr<MPCFlat> LocalPlain(IsMega ? 0 : new MPCFlat);
			MPCFlat &Local = IsMega ? (MPCFlat &) *LocalMega : *LocalPlain;
			Local.m_ConsistencyIterCount = m_MPC->m_ConsistencyIterCount;
			Local.m_RefineIterCount = m_MPC->m_RefineIterCount;
			Local.m_D.m_Disable = m_MPC->m_D.m_Disable; // super7.cpp:11
				{
				std::lock_guard<std::mutex> Guard(g_MapMu);
				g_SlotOf[&Local] = 1 + (int) w;
				}
			t_JoinIndex = (int) (w % (uint) std::min<size_t>(DeviceList().size(), MAX_JOIN_CTX)); // = the device this worker's slot is dealt (CtxOfSlot)
			for (;;)
				{
				const uint ShrubIndex = Next.fetch_add(1);
				if (ShrubIndex >= ShrubCount)
					break;
				MuscleGpuRandThreadRestore(&RandAt[ShrubIndex]);
				MultiSequence ShrubInput;
				MakeShrubInput(m_ShrubLCAs[ShrubIndex], ShrubInput); // super7.cpp:116-126
				Local.m_TreePerm = TP_None;
				Local.Run(&ShrubInput);
				MultiSequence *ShrubMSA = new MultiSequence;
				ShrubMSA->Copy(*Local.m_MSA);
				m_ShrubMSAs[ShrubIndex] = ShrubMSA;
				}
			MuscleGpuRandThreadEnd();
				{
				std::lock_guard<std::mutex> Guard(g_MapMu);
				g_SlotOf.erase(&Local);
				g_Batches.erase(&Local);
				}
			g_Slots[1 + w].m_StoreOwner = 0; // Local dies with this thread
			});
	for (size_t t = 0; t < Threads.size(); ++t)
		Threads[t].join();
	opt_quiet = SavedQuiet;
	if (TimingOn())
		fprintf(stderr, "[muscle_gpu] %u shrubs on %u worker contexts: entered %.3f s after process start, took %.3f s\n", ShrubCount, Workers,
		  ShrubsBegin, std::chrono::duration<double>(std::chrono::steady_clock::now() - g_ProcessStart).count() - ShrubsBegin);
	}
