"""Kernel-logic check WITHOUT a GPU: the product sources (muscle_amd/csrc) compiled against the
SIMT emulator (tests/emu) must agree bit for bit with the oracle on tiny inputs. This validates
indexing, the systolic skew schedule, border cells, sorting/EA/sparsify, slab build, relax and
commit before GPU minutes are spent; the real parity tests are the -m gpu ones."""
import os
import subprocess

import numpy as np
import pytest

import _golden as G
import _parity as P
from muscle_amd.synth import make_family

EMU_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu")
EMU_LIB = os.path.join(EMU_DIR, "libmpcgpu_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", EMU_DIR], stdout=subprocess.DEVNULL)
    return EMU_LIB


@pytest.mark.parametrize("name", ["n2_L40", "n3_L30"])
def test_emu_vs_golden_small(emu, name):
    g = G.mpc(name)
    stages, ea = P.run_lib(g["seqs"], lib_path=emu)
    assert np.array_equal(P.bits(ea), P.bits(g["ea"]))
    for s in range(g["nstages"]):
        assert G.stage_digest(stages[s]) == g["digest"][s]


def test_emu_vs_oracle_ragged(emu):
    seqs = ["M", "MKVLA", make_family(1, 70, seed=9)[0], "ACDEFGHIKLMNPQRSTVWY" * 4, "WWWWWWWW"]
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "ragged")


def test_emu_multirow_lanes(emu):
    # LX > 64 -> H = 2..3 rows per lane, several lanes idle at the tail
    seqs = make_family(3, 150, seed=4)
    P.assert_same(P.run_lib(seqs, lib_path=emu), P.run_oracle(seqs), "H>1")
