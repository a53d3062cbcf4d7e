"""North-star check on the GPU box: the reference's own `muscle -align` with the MI355X posterior
stage linked in as a drop-in (hostcxx/_build/muscle_gpu = unmodified mpcflat.{h,cpp} + every other
reference object + hostcxx/mpcflat_gpu.cpp + libmpcgpu.so) writes the SAME final MSA, byte for
byte, as the unmodified reference — against the committed golden MD5s and, where the compiled
reference travelled along (oracle/_ref/muscle), against a live run of it."""
import os

import pytest

import _msa

pytestmark = pytest.mark.gpu

SETS = ["n2_L40", "n3_L30", "n8_L60", "ragged", "bb11001", "bb11005", "n32_L150", "dupes", "consiters0", "perturb",
        "synth_64x200_s1", "synth_128x300_s1", "super7_200x120_b32", "super7_8x18_b4", "super5_14x20", "super5_120x80",
        "synth_5x1300_s3",  # sequences longer than 1024: row-block fb kernel, gather relax
        # .mega inputs (structure profiles): CalcPost's profile branch in -align and in the -super7 joins
        "mega_bb11001", "mega_synth_6x40_s2", "mega_super7_12x30_b4"]


@pytest.fixture(scope="module")
def gpu_muscle():
    if not os.path.exists(_msa.GPU_MUSCLE):
        pytest.fail("hostcxx/_build/muscle_gpu missing: run __graft_entry__.build() where /root/reference exists")
    return _msa.GPU_MUSCLE


@pytest.mark.parametrize("name", SETS)
def test_final_msa_identical_to_reference(gpu_muscle, name):
    """Primary check: muscle_gpu vs the unmodified reference binary run on THIS box, same input,
    same thread count. Secondary: both equal the MD5 committed from the build container (a host
    whose libm/ifunc selection differs could legitimately move both together; report, don't hide)."""
    from muscle_amd.hostinfo import usable_cores
    th = usable_cores()
    md5_gpu, data_gpu = _msa.run_muscle(gpu_muscle, name, threads=th)
    golden = _msa.golden_md5()[name]
    if os.path.exists(_msa.REF_MUSCLE):
        md5_ref, data_ref = _msa.run_muscle(_msa.REF_MUSCLE, name, threads=th)
        print("%s gpu=%s ref(live)=%s golden=%s" % (name, md5_gpu, md5_ref, golden))
        assert data_gpu == data_ref, "final MSA differs from the live reference's for %s" % name
        if md5_ref != golden:
            pytest.xfail("the unmodified reference itself writes a different MSA on this host than in the build "
                         "container (%s vs %s); muscle_gpu follows the live reference" % (md5_ref, golden))
    assert md5_gpu == golden, "final MSA differs from the reference's committed MD5 for %s" % name


@pytest.mark.parametrize("name,devices", [("n32_L150", "0,0"), ("synth_64x200_s1", "0,0,0"), ("bb11001", "0,0"), ("mega_bb11001", "0,0")])
def test_final_msa_identical_sharded_over_contexts(gpu_muscle, name, devices):
    """The multi-GPU path of the product (MUSCLE_GPU_DEVICES -> mpcgpu_group_*: pairs sharded over contexts, all-gather of
    the sparse posteriors before relax, all-gather of the values after each iteration) with several contexts on the one
    device of this box (peer-copy transport): same final MSA as the reference."""
    from muscle_amd.hostinfo import usable_cores
    md5, _ = _msa.run_muscle(gpu_muscle, name, threads=usable_cores(), env={"MUSCLE_GPU_DEVICES": devices})
    assert md5 == _msa.golden_md5()[name]


@pytest.mark.parametrize("name,workers", [("super7dm_300x100_b16", "1"), ("super7dm_300x100_b16", "4"), ("super7dm_2000x250_b32", "8"),
                                          ("super7_200x120_b32", "3")])
def test_super7_with_distmx_and_parallel_shrubs(gpu_muscle, name, workers):
    """BASELINE config 5 as stated: -super7 with the guide tree from a precomputed distance matrix (-distmxin, reseek format),
    MPCFlat on the shrubs — run by MUSCLE_GPU_SHRUB_CONTEXTS worker threads with a device context each (the reference's loop is
    sequential: super7.cpp:127-137) — then the PProg joins. Final MSA = the reference's (MD5s from the compiled reference:
    2000 x L~250 took it 231 s on 7 threads; muscle_gpu 3.5 s)."""
    from muscle_amd.hostinfo import usable_cores
    md5, _ = _msa.run_muscle(gpu_muscle, name, threads=usable_cores(), env={"MUSCLE_GPU_SHRUB_CONTEXTS": workers})
    assert md5 == _msa.golden_md5()[name]


@pytest.mark.parametrize("name", ["super7dm_300x100_b16", "super7dm_2000x250_b32"])
def test_super7_shrub_workers_over_a_device_list(gpu_muscle, name):
    """BASELINE config 5's second split (shrubs over GPUs): MUSCLE_GPU_DEVICES lists the devices, the shrub workers' contexts
    are dealt over them and every worker's joins run on the join context of ITS device (one join context per listed device,
    hostcxx/mpcflat_gpu.cpp: JoinCtx). With "0,0" on this one-GPU box: two join contexts, four workers. Same MSA."""
    from muscle_amd.hostinfo import usable_cores
    md5, _ = _msa.run_muscle(gpu_muscle, name, threads=usable_cores(), env={"MUSCLE_GPU_DEVICES": "0,0", "MUSCLE_GPU_SHRUB_CONTEXTS": "4"})
    assert md5 == _msa.golden_md5()[name]


def test_super7_parallel_shrubs_with_progress_output(gpu_muscle):
    """The same without -quiet: the reference's progress reporting is not thread-safe (myutils.cpp:1453-1870), the parallel shrub
    loop keeps its workers quiet and reports from the main thread; 40 shrubs on 8 workers, 3 repetitions."""
    from muscle_amd.hostinfo import usable_cores
    for _ in range(3):
        md5, _d = _msa.run_muscle(gpu_muscle, "super7dm_300x100_b16", threads=usable_cores(), env={"MUSCLE_GPU_SHRUB_CONTEXTS": "8"}, quiet=False)
        assert md5 == _msa.golden_md5()["super7dm_300x100_b16"]


@pytest.mark.parametrize("fixture", ["bp_n9_L40", "bp_n12_L70"])
def test_profseq_drives_buildpost_on_the_device(gpu_muscle, fixture):
    """`-profseq` (profseq.cpp:33-49) -> MPCFlat::BuildPost of the drop-in -> mpcgpu_build_post: the logged path equals what the
    compiled reference's BuildPost + CalcAlnFlat give (tests/golden/bp_*.npz); no reference BuildPost is linked into muscle_gpu."""
    import subprocess
    got, want = _msa.run_profseq(gpu_muscle, fixture)
    assert got == want
    syms = subprocess.run(["nm", gpu_muscle], capture_output=True, text=True).stdout
    assert "MPCFlat_BuildPost_ref" not in syms


def test_super5_uclust_on_the_device(gpu_muscle):
    """-super5 on 600 x L~150: UCLUST (uclust.cpp:26-56; the drop-in's UClust::Search aligns all <= 8 word-count hits of a sequence
    in one mpcgpu_align_pairs call), then 214 PProg joins. Final MSA = the reference's (MD5 from a single-thread run of the compiled
    reference, 12.5 CPU-minutes — not repeated here)."""
    md5, _ = _msa.run_muscle(gpu_muscle, "super5_600x150", threads=1)
    assert md5 == _msa.golden_md5()["super5_600x150"]


@pytest.mark.parametrize("name,env,limit_s", [
    ("synth_1000x400_s1", {}, 60.0),                                            # BASELINE config 3 end to end: muscle -align
    ("super7dm_10000x250_b32", {"MUSCLE_GPU_SHRUB_CONTEXTS": "8"}, 90.0),       # BASELINE config 5 end to end: -super7 + distance matrix
])
def test_baseline_configs_end_to_end(gpu_muscle, name, env, limit_s):
    """BASELINE configs 3 and 5 END TO END through the reference's own CLI with the device stage linked in: the final MSA's MD5
    is the one the unmodified compiled reference wrote (1000 x L~400 `-align`: 81 minutes on 6 threads, profiles/r01i_ref1000.log;
    10 000 x L~250 `-super7 -distmxin`: 19 minutes on 7 threads, profiles/r03i_joins_e2e_reftime.log) — the reference is not run
    again here. The time limits are loose guards against a silent slow path, not measurements (those are in profiles/)."""
    import time
    from muscle_amd.hostinfo import usable_cores
    t0 = time.time()
    md5, _ = _msa.run_muscle(gpu_muscle, name, threads=usable_cores(), env=env, timeout=600)
    el = time.time() - t0
    print("%s: %.1f s end to end, md5 %s" % (name, el, md5))
    assert md5 == _msa.golden_md5()[name]
    if el >= limit_s:  # a loaded box must not fail a parity test: the time is a warning, the measurements are in profiles/
        import warnings
        warnings.warn("%s took %.1f s end to end (guard %.0f s): slow path or loaded box?" % (name, el, limit_s))
