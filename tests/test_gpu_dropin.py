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
        "synth_64x200_s1", "synth_128x300_s1"]


@pytest.fixture(scope="module")
def gpu_muscle():
    if not os.path.exists(_msa.GPU_MUSCLE):
        pytest.fail("hostcxx/_build/muscle_gpu missing: run __graft_entry__.build() where /root/reference exists")
    return _msa.GPU_MUSCLE


@pytest.mark.parametrize("name", SETS)
def test_final_msa_identical_to_reference(gpu_muscle, name):
    from muscle_amd.hostinfo import usable_cores
    md5, data = _msa.run_muscle(gpu_muscle, name, threads=usable_cores())
    assert md5 == _msa.golden_md5()[name], "final MSA differs from the reference's for %s" % name


def test_live_reference_agrees(gpu_muscle):
    if not os.path.exists(_msa.REF_MUSCLE):
        pytest.skip("compiled reference not shipped")
    from muscle_amd.hostinfo import usable_cores
    a = _msa.run_muscle(gpu_muscle, "synth_40x120_s7", threads=usable_cores())
    b = _msa.run_muscle(_msa.REF_MUSCLE, "synth_40x120_s7", threads=usable_cores())
    assert a[1] == b[1]
