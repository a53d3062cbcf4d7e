"""Host-side plumbing of the C++ drop-in (hostcxx/mpcflat_gpu.cpp) WITHOUT a GPU: the reference's
own `muscle` is linked with the drop-in and the SIMT-emulator build of libmpcgpu (test
infrastructure) and must write byte-identical final MSAs to the committed golden MD5s of the
unmodified reference (tests/golden/msa_md5.json). Checks lazy batching from the OpenMP loop, the
HMM-table hand-over, Derep/InsertDupes around the stage, the <3-sequence / -consiters 0 paths and
the buffer swap, and MPCFlat::AlignAlns on the device store (progressive joins + refinement rounds,
limited with -refineiters because every round is an emulated kernel), and PProg::AlignMSAsFlat
(the MSA x MSA joins of the -super7 / -super5 drivers) on explicit pair lists. Needs the reference objects (oracle/_ref/obj): skipped where
/root/reference was never available."""
import os
import subprocess

import pytest

import _msa

ROOT = _msa.ROOT
pytestmark = pytest.mark.ref


@pytest.fixture(scope="module")
def emu_muscle():
    if not os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "obj")) or not os.path.isdir("/root/reference/src"):
        if os.path.exists(_msa.EMU_MUSCLE):
            return _msa.EMU_MUSCLE
        pytest.skip("reference objects not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu")], stdout=subprocess.DEVNULL)
    env = dict(os.environ, MPCGPU_LIBDIR=os.path.join(ROOT, "tests", "emu"), MPCGPU_LIBNAME="mpcgpu_emu",
               MPCGPU_BIN="muscle_gpu_emu")
    subprocess.check_call(["bash", os.path.join(ROOT, "hostcxx", "build_muscle_gpu.sh")], env=env, stdout=subprocess.DEVNULL)
    return _msa.EMU_MUSCLE


@pytest.mark.parametrize("name", ["n2_L40+r2", "n3_L30+r2", "synth_6x40_s2+r2", "dupes+r2", "consiters0+r2", "perturb_small+r2", "super7_8x18_b4", "super5_14x20",
                                  # .mega inputs: structure-profile emissions in MPCFlat::CalcPosterior and in the PProg joins
                                  "mega_bb11001+r2", "mega_synth_6x40_s2+r2", "mega_super7_6x16_b3",
                                  # full runs (100 refinement rounds): affordable since the emulator runs GPU threads as fibers
                                  "n8_L60", "bb11001", "mega_bb11001", "perturb",
                                  # sequences longer than 1024: the row-block fb kernels (H = 7, three blocks) and the multi-wave alignment kernel. The
                                  # same paths as synth_5x1300_s3 with its 100 refinement rounds (108 s on the emulator; that set runs on the GPU:
                                  # tests/test_gpu_dropin.py) in 6 s
                                  "synth_3x1100_s3+r2"])
def test_final_msa_identical(emu_muscle, name):
    md5, _ = _msa.run_muscle(emu_muscle, name, threads=3)
    assert md5 == _msa.golden_md5()[name]


@pytest.mark.parametrize("name,devices", [("synth_6x40_s2+r2", "0,0"), ("n8_L60", "0,0,0"), ("mega_synth_6x40_s2+r2", "0,0")])
def test_final_msa_identical_sharded_over_contexts(emu_muscle, name, devices):
    """MUSCLE_GPU_DEVICES: the drop-in shards CalcPosteriors / ConsIter over several contexts (mpcgpu_group_*: pair shards,
    all-gather of the packed posteriors, all-gather of the values) — here contexts of the one emulated device — and the
    final MSA stays the reference's."""
    md5, _ = _msa.run_muscle(emu_muscle, name, threads=3, env={"MUSCLE_GPU_DEVICES": devices})
    assert md5 == _msa.golden_md5()[name]


@pytest.mark.parametrize("workers", ["1", "2", "6"])
def test_super7_joins_side_by_side(emu_muscle, workers):
    """-super7 on 24 sequences in shrubs of 3: eight shrubs, seven PProg joins along the shrub tree (pprog2.cpp:58-76) — run by
    MUSCLE_GPU_JOIN_WORKERS threads on join contexts of their own (hostcxx: PProg::Run2), independent subtrees side by side; 1 = the
    reference's loop. The final MSA is the reference's whatever the number of workers."""
    md5, _ = _msa.run_muscle(emu_muscle, "super7_24x14_b3", threads=2, env={"MUSCLE_GPU_JOIN_WORKERS": workers, "MUSCLE_GPU_SHRUB_CONTEXTS": "3"})
    assert md5 == _msa.golden_md5()["super7_24x14_b3"]


@pytest.mark.parametrize("name,workers", [("super7_8x18_b4", "1"), ("super7_8x18_b4", "2"), ("super7_8x18_b4", "5")])
def test_super7_shrubs_over_worker_contexts(emu_muscle, name, workers):
    """-super7: the shrub loop (super7.cpp:127-137) run by MUSCLE_GPU_SHRUB_CONTEXTS worker threads, each with its own MPCFlat and
    device context and its own position in the rand() stream of the refinement (hostcxx/mpcflat_gpu.cpp, rand_isolate.cpp):
    the final MSA is the reference's whatever the number of workers (1 = the reference's sequential loop)."""
    md5, _ = _msa.run_muscle(emu_muscle, name, threads=2, env={"MUSCLE_GPU_SHRUB_CONTEXTS": workers})
    assert md5 == _msa.golden_md5()[name]


def test_super7_shrub_workers_over_a_device_list(emu_muscle):
    """shrub workers + MUSCLE_GPU_DEVICES: worker contexts dealt over the listed devices, one join context per listed device, a
    worker's joins on the join context of its device (hostcxx/mpcflat_gpu.cpp: JoinCtx)"""
    md5, _ = _msa.run_muscle(emu_muscle, "super7_8x18_b4", threads=2, env={"MUSCLE_GPU_DEVICES": "0,0,0", "MUSCLE_GPU_SHRUB_CONTEXTS": "4"})
    assert md5 == _msa.golden_md5()["super7_8x18_b4"]


def test_super7_parallel_shrubs_with_progress_output(emu_muscle):
    """Default verbosity (no -quiet): MPCFlat::Run's ProgressStep keeps unguarded process globals (myutils.cpp:1453-1870), so the
    worker threads of the parallel shrub loop must not call it concurrently (round-2 advisor finding); same MSA, no crash."""
    md5, _ = _msa.run_muscle(emu_muscle, "super7_8x18_b4", threads=2, env={"MUSCLE_GPU_SHRUB_CONTEXTS": "4"}, quiet=False)
    assert md5 == _msa.golden_md5()["super7_8x18_b4"]


@pytest.mark.parametrize("fixture", ["bp_n9_L40"])
def test_profseq_drives_buildpost_on_the_device(emu_muscle, fixture):
    """`-profseq` is the caller of MPCFlat::BuildPost outside MPCFlat::Run (profseq.cpp:33-49). The drop-in's BuildPost builds the
    matrix on the device (mpcgpu_build_post; no reference code behind it: buildpostflat.o is not linked); the path the command
    logs equals what the compiled reference's BuildPost + CalcAlnFlat give for the same alignment and query
    (tests/golden/bp_*.npz: ps_rows / ps_path)."""
    got, want = _msa.run_profseq(emu_muscle, fixture)
    assert got == want
