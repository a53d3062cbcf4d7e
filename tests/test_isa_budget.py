"""Build-time checks of the compiled gfx950 code of the hot kernels (hipcc cross-compiles without a GPU): what the register
allocator did is a performance property that parity tests cannot see.

relax_band_kernel (and relax_var_kernel before it) walks Z with every accumulator and row offset in VGPRs; a value the compiler keeps in a spill slot instead is
reloaded between the merges of a step, and each reload waits for vmcnt(0). The 14-cells-per-lane instantiation of the default
geometry had five such reloads (1195 ms per two iterations at 1000 x L~400 against 1171 with 13 cells per lane and none:
profiles/r05g). The forward/backward kernels must not spill at all."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "muscle_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    # the listing `make -C muscle_amd/csrc asm` (__graft_entry__.build()) leaves, when it is newer than every source it was made from:
    # same compiler, same flags — a minute of compilation saved
    made = os.path.join(CSRC, "mpcgpu.gfx950.s")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cpp", ".inc"))] + [os.path.join(ROOT, "include", "mpcgpu.h"), os.path.join(CSRC, "Makefile")]
    if os.path.exists(made) and os.path.getmtime(made) >= max(os.path.getmtime(f) for f in srcs):
        return open(made).read()
    out = tmp_path_factory.mktemp("isa") / "mpcgpu.gfx950.s"
    flags = "-O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off".split()  # the Makefile's CXXFLAGS
    subprocess.check_call([HIPCC, "-x", "hip", "--offload-arch=gfx950"] + flags + ["-S", "--cuda-device-only", "mpcgpu.cpp", "-o", str(out)],
                          cwd=CSRC, stderr=subprocess.DEVNULL)
    return open(out).read()


def _body(isa, mangled_prefix):
    i = isa.index(mangled_prefix)
    i = isa.index(":\n", i)
    return isa[i:isa.index(".Lfunc_end", i)].split("\n")


def _scratch(lines):
    return [k for k, l in enumerate(lines) if re.search(r"\bscratch_(load|store)|\bbuffer_(load|store)", l)]


BAND = "_Z17relax_band_kernelILi1024ELi13ELi2ELi0E14MpcRbBlocksAsmEv15RelaxBandParams"   # the two-list walk (wide rows)
BAND_WIN = "_Z17relax_band_kernelILi1024ELi15ELi2ELi0E11MpcRbWinAsmEv15RelaxBandParams"  # the direct-index merge (narrow rows: the bench's default), 15 cells per lane (16: two accumulators reloaded per step)


def _walk(body):
    merges = [k for k, l in enumerate(body) if re.match(r"\.Lrv_step_\d+:", l.strip())]
    return merges


@pytest.mark.parametrize("kernel", [BAND_WIN, BAND])
def test_default_relax_walk_has_no_spill_reloads(isa, kernel):
    """relax_band_kernel, the instantiation relax_band launches by default (kBandSlots cells per lane). A reload inside the walk is
    worse here than in relax_var_kernel: its wait is vmcnt(0), and the prefetch of the next step is in flight on the same counter."""
    src = open(os.path.join(CSRC, "mpcgpu_relax.inc")).read()  # (the host side of the relax: one of the files mpcgpu.cpp includes)
    assert re.search(r"kBandThreads = 1024, kBandSlots = 13;", src) and re.search(r"kBandSlotsWin = 15;", src), "default geometry of relax_band changed: update this test"
    body = _body(isa, kernel)
    merges = _walk(body)
    assert len(merges) == (15 if kernel == BAND_WIN else 13), len(merges)  # one hand-scheduled merge loop per cell slot
    inside = [k for k in _scratch(body) if merges[0] <= k <= merges[-1]]
    assert not inside, "spill code between the merges of a step: " + "; ".join(body[k].strip() for k in inside[:5])
    # no wait for the VMEM counter between the prefetch (the staging block sits between the first and the second merge) and the last
    # merge: the transfers of the next step must stay in flight under the merges
    dma = [k for k, l in enumerate(body) if "global_load_lds_dwordx4" in l and merges[0] <= k <= merges[-1]]
    assert dma, "the prefetch is expected between the merges of a step"
    waits = [k for k, l in enumerate(body) if "vmcnt" in l and dma[0] < k <= merges[-1]]
    assert not waits, "; ".join(body[k].strip() for k in waits[:5])


BAND_512 = "_Z17relax_band_kernelILi512ELi13ELi4ELi0E14MpcRbBlocksAsmEv15RelaxBandParams"  # MPCGPU_RELAX_WG=512: four 512-thread workgroups per CU


@pytest.mark.parametrize("kernel", [BAND_WIN, BAND, BAND_512])
def test_band_merge_registers_are_not_touched_between_statements(isa, kernel):
    """The hand-scheduled merge leaves LDS reads in flight into v24..v40 (the next slot's first blocks, its Y bias) when a
    statement ends; the compiler does not know. Between the end of one merge statement and the opening wait of the next
    (or the drain after the last slot) no compiler-generated instruction may read or write those registers."""
    body = _body(isa, kernel)
    # the walk pins v24..v40; the direct-index merge v24..v28, v32..v36 and v40 (X block + descriptor word per set, the value base)
    # (every hand-scheduled instantiation relax_band can launch is checked: ADVICE r4)
    pinned = set(range(24, 41)) if kernel != BAND_WIN else (set(range(24, 29)) | set(range(32, 37)) | {40})

    def regs(line):
        out = set()
        for m in re.finditer(r"\bv(\d+)\b", line):
            out.add(int(m.group(1)))
        for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        return out

    in_gap, bad = False, []
    for k, l in enumerate(body):
        t = l.strip()
        if re.match(r"\.Lrv_(done|last)_\d+:", t):
            in_gap = "armed"  # the statement's last instruction (restore exec) follows, then ;;#ASMEND
        elif in_gap == "armed" and t.startswith(";;#ASMEND"):
            in_gap = True
        elif in_gap is True and t.startswith("s_waitcnt lgkmcnt(0)"):
            in_gap = False
        elif in_gap is True and t and not t.startswith((";", ".")) and regs(t) & pinned:
            bad.append("%d: %s" % (k, t))
    assert not bad, "\n".join(bad[:8])


def test_forward_backward_kernels_do_not_spill(isa):
    for name in ("_Z15fb_chain_kernelILi7EEv13FbChainParams", "_Z9fb_kernelILi7ELb0ELb0EEv8FbParams"):
        body = _body(isa, name)
        assert not _scratch(body), name
