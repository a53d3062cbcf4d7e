// hostcxx/rand_isolate.cpp — keeps the reference's libc rand() stream private to the reference.
//
// MPCFlat::RefineIter picks its random bipartitions with `rand()%2` (refineflat.cpp:14) and the
// reference never seeds it, so its 100 refinement rounds are a fixed function of glibc's default
// rand() sequence (srand(1)). The HIP/ROCr runtime that libmpcgpu.so brings into the process also
// draws from that same process-global generator (from its own threads, at unpredictable times —
// observed on MI355X / ROCm 7.2: the final MSA of small inputs changed from run to run although
// every number crossing the GPU boundary was bit-identical), which shifts the sequence the
// refinement sees. muscle_gpu is therefore linked with -Wl,--wrap=rand: calls to rand() from the
// reference's objects land here, shared libraries keep using libc's. The replacement is glibc's
// own generator (random_r on a private 128-byte TYPE_3 state seeded with 1 — exactly what rand()
// is before any srand()), so the reference sees the sequence it would see without a GPU runtime
// in the process.
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

namespace
{
struct random_data g_Data;
char g_State[128];
bool g_Init = false;
std::mutex g_Mu;

// Parallel shrub loop of -super7 (hostcxx/mpcflat_gpu.cpp, Super7::IntraAlignShrubs): the reference aligns the shrubs one after
// the other, so shrub k's refinement sees the stream from position sum_{j<k} calls_j on. A worker thread that runs shrub k
// positions a PRIVATE copy of the same generator there (same seed, that many values discarded) and draws from it until it
// says otherwise. The positions are recorded in ONE pass over the shared stream (MuscleGpuRandSharedSnapshots), which also
// leaves the shared stream where the sequential loop would have left it.
struct ThreadStream
	{
	bool m_On = false;
	struct random_data m_Data;
	char m_State[128];
	};
thread_local ThreadStream t_Stream;
}

// A position of the shared stream: the generator's 128 state bytes and where its two taps stand (in words from `state`).
struct MuscleGpuRandSnapshot
	{
	char m_State[128];
	int m_FOff;
	int m_ROff;
	};

namespace
{
void InitShared()
	{
	if (g_Init)
		return;
	g_Data.state = 0; // initstate_r requires this on first use
	initstate_r(1, g_State, sizeof(g_State), &g_Data);
	g_Init = true;
	}
}

// Advances the SHARED stream past Draws[0] + ... + Draws[Count-1] values, as the sequential shrub loop would, and records in
// At[k] where it stood before shrub k's draws: one pass, linear in the number of draws, from wherever the stream stands now.
extern "C" void MuscleGpuRandSharedSnapshots(const unsigned long long *Draws, unsigned Count, MuscleGpuRandSnapshot *At)
	{
	std::lock_guard<std::mutex> Guard(g_Mu);
	InitShared();
	for (unsigned k = 0; k < Count; ++k)
		{
		memcpy(At[k].m_State, g_State, sizeof(g_State));
		At[k].m_FOff = (int) (g_Data.fptr - g_Data.state);
		At[k].m_ROff = (int) (g_Data.rptr - g_Data.state);
		int32_t r = 0;
		for (unsigned long long i = 0; i < Draws[k]; ++i)
			random_r(&g_Data, &r);
		}
	}

// The calling thread's rand() now continues from a recorded position of the shared stream, privately.
extern "C" void MuscleGpuRandThreadRestore(const MuscleGpuRandSnapshot *At)
	{
	ThreadStream &T = t_Stream;
	T.m_Data.state = 0;
	initstate_r(1, T.m_State, sizeof(T.m_State), &T.m_Data); // type, degree, separation, end pointer
	memcpy(T.m_State, At->m_State, sizeof(T.m_State));
	T.m_Data.fptr = T.m_Data.state + At->m_FOff;
	T.m_Data.rptr = T.m_Data.state + At->m_ROff;
	T.m_On = true;
	}

extern "C" void MuscleGpuRandThreadEnd(void)
	{
	t_Stream.m_On = false;
	}

extern "C" int __wrap_rand(void)
	{
	if (t_Stream.m_On)
		{
		int32_t r = 0;
		random_r(&t_Stream.m_Data, &r);
		return (int) r;
		}
	std::lock_guard<std::mutex> Guard(g_Mu);
	InitShared();
	int32_t r = 0;
	random_r(&g_Data, &r);
	return (int) r;
	}
