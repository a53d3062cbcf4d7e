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

namespace
{
struct random_data g_Data;
char g_State[128];
bool g_Init = false;
std::mutex g_Mu;

// Parallel shrub loop of -super7 (hostcxx/mpcflat_gpu.cpp, Super7::IntraAlignShrubs): the reference aligns the shrubs one after
// the other, so shrub k's refinement sees the stream from position sum_{j<k} calls_j on. A worker thread that runs shrub k
// positions a PRIVATE copy of the same generator there (same seed, that many values discarded) and draws from it until it
// says otherwise; the shared stream is advanced past all shrubs at the end, as the sequential loop would have left it.
struct ThreadStream
	{
	bool m_On = false;
	struct random_data m_Data;
	char m_State[128];
	};
thread_local ThreadStream t_Stream;
}

extern "C" void MuscleGpuRandThreadSeek(unsigned long long Offset)
	{
	ThreadStream &T = t_Stream;
	T.m_Data.state = 0;
	initstate_r(1, T.m_State, sizeof(T.m_State), &T.m_Data);
	int32_t r = 0;
	for (unsigned long long i = 0; i < Offset; ++i)
		random_r(&T.m_Data, &r);
	T.m_On = true;
	}

extern "C" void MuscleGpuRandThreadEnd(void)
	{
	t_Stream.m_On = false;
	}

extern "C" int __wrap_rand(void);
extern "C" void MuscleGpuRandSharedSkip(unsigned long long Count)
	{
	for (unsigned long long i = 0; i < Count; ++i)
		(void) __wrap_rand();
	}

namespace
{
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
	if (!g_Init)
		{
		g_Data.state = 0; // initstate_r requires this on first use
		initstate_r(1, g_State, sizeof(g_State), &g_Data);
		g_Init = true;
		}
	int32_t r = 0;
	random_r(&g_Data, &r);
	return (int) r;
	}
