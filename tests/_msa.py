"""Final-MSA parity helpers: run a `muscle` binary (the compiled reference, or the same reference
with the GPU drop-in linked in) on one of the named input sets and hash the output alignment.
TEST INFRASTRUCTURE."""
import hashlib
import json
import os
import subprocess
import tempfile

import _golden as G
import _mega
from muscle_amd.synth import make_family, write_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MUSCLE = os.path.join(ROOT, "oracle", "_ref", "muscle")
GPU_MUSCLE = os.path.join(ROOT, "hostcxx", "_build", "muscle_gpu")
EMU_MUSCLE = os.path.join(ROOT, "hostcxx", "_build", "muscle_gpu_emu")
GOLDEN = os.path.join(ROOT, "tests", "golden", "msa_md5.json")


def _balanced_newick(lo, hi):
    if hi - lo == 1:
        return "s%d:0.1" % lo
    mid = (lo + hi) // 2
    return "(%s,%s):0.1" % (_balanced_newick(lo, mid), _balanced_newick(mid, hi))


def reseek_distmx(n, band=24):
    """A deterministic sparse similarity matrix in the reseek format -distmxin reads (upgma5.cpp:436-503: "distmx<TAB>N", N lines
    "index<TAB>label", then "i<TAB>j<TAB>similarity"; absent pairs are 0): every sequence is similar to its next `band` neighbours,
    less with distance, plus a little hash noise so that UPGMA has no ties."""
    out = ["distmx\t%d" % n] + ["%d\ts%d" % (i, i) for i in range(n)]
    for i in range(n):
        for j in range(i + 1, min(i + 1 + band, n)):
            out.append("%d\t%d\t%.6f" % (i, j, 1.0 / (1 + j - i) + 0.001 * ((i * 7919 + j * 104729) % 97) / 97.0))
    return "\n".join(out) + "\n"


def command(name):
    """The muscle command of a set: -align (MPCFlat::Run on everything), or for super7_* the
    BASELINE config-5 path (super7.cpp:9-137: guide tree -> shrubs -> MPCFlat::Run per shrub ->
    PProg joins), driven by a balanced guide tree so no distance matrix is needed."""
    name = name.split("+r")[0]
    if name.startswith("mega_"):
        name = name[5:]
    if name.startswith("super7_"):  # super7_<n>x<L>_b<shrub size>
        n = int(name[7:].split("x")[0])
        shrub = name.split("_b")[1]
        return "-super7", ["-guidetreein", "tree.nwk", "-shrub_size", shrub], {"tree.nwk": _balanced_newick(0, n) + ";\n"}
    if name.startswith("super7dm_"):  # super7dm_<n>x<L>_b<shrub size>: BASELINE config 5 as stated — guide tree from a precomputed distance matrix
        n = int(name[9:].split("x")[0])
        shrub = name.split("_b")[1]
        return "-super7", ["-distmxin", "dm.tsv", "-shrub_size", shrub], {"dm.tsv": reseek_distmx(n)}
    if name.startswith("super5_"):  # super5_<n>x<L>: UCLUST split + MPCFlat per cluster + PProg joins (super5.cpp)
        return "-super5", [], {}
    return "-align", [], {}


def input_set(name):
    """-> (seqs, labels, extra command-line args). A "+rK" suffix adds -refineiters K (the emulator
    runs of the drop-in use it: every refinement round is one emulated device alignment)."""
    if "+r" in name:
        base, k = name.split("+r")
        seqs, labels, extra = input_set(base)
        return seqs, labels, extra + ["-refineiters", k]
    if name.startswith("super7_"):
        n, L = name[7:].split("_b")[0].split("x")
        return make_family(int(n), int(L), seed=13), None, []
    if name.startswith("super7dm_"):
        n, L = name[9:].split("_b")[0].split("x")
        return make_family(int(n), int(L), seed=13), None, []
    if name.startswith("super5_"):
        n, L = name[7:].split("x")
        return make_family(int(n), int(L), seed=4), None, []
    if name.startswith("synth"):  # synth_<n>x<L>_s<seed>
        n, rest = name[6:].split("x")
        L, seed = rest.split("_s")
        return make_family(int(n), int(L), seed=int(seed)), None, []
    if name == "dupes":  # exact duplicates exercise Derep + InsertDupes around the stage (mpcflat.cpp:290-336)
        s = make_family(5, 50, seed=9)
        return [s[0], s[1], s[0], s[2], s[1], s[3], s[4]], None, []
    if name == "consiters0":
        return make_family(6, 45, seed=4), None, ["-consiters", "0"]
    if name == "perturb_small":
        return make_family(5, 36, seed=6), None, ["-perturb", "5", "-perm", "bca"]
    if name == "perturb":
        return make_family(7, 60, seed=5), None, ["-perturb", "3", "-perm", "acb"]
    return G.mpc(name)["seqs"], None, []


def mega_input(name):
    """mega_* sets: the input is a .mega file (structure profiles; loadinput.cpp:5-9 keys on the extension), so
    CalcPost runs its profile branch (calcpost.cpp:14-22). -> (file text, extra args)"""
    base, extra = name, []
    if "+r" in name:
        base, k = name.split("+r")
        extra = ["-refineiters", k]
    if base.startswith("mega_super7_"):  # mega_super7_<n>x<L>_b<shrub size>: labels s0.. match the balanced tree
        n, L = base[12:].split("_b")[0].split("x")
        return _mega.synth_mega_text(int(n), int(L), seed=13), extra
    return _mega.mega_text(base), extra


def run_muscle(binary, name, threads=4, timeout=900, env=None, quiet=True):
    mega = name.startswith("mega_")
    if mega:
        text, extra = mega_input(name)
    else:
        seqs, labels, extra = input_set(name)
    cmd, cmd_extra, files = command(name)
    if cmd == "-super5":
        # PProg::Run picks joins by the average EA of AlignMSAsFlat (pprog.cpp:286,394), which the reference
        # sums in thread-arrival order (getpostpairsalignedflat.cpp:92-95): only one thread is deterministic
        threads = 1
    with tempfile.TemporaryDirectory() as d:
        fa, out = os.path.join(d, "in.mega" if mega else "in.fa"), os.path.join(d, "out.afa")
        if mega:
            with open(fa, "w") as f:
                f.write(text)
        else:
            write_fasta(fa, seqs, labels)
        for fn, text in files.items():
            with open(os.path.join(d, fn), "w") as f:
                f.write(text)
        subprocess.run([binary, cmd, fa, "-output", out, "-threads", str(threads)] + (["-quiet"] if quiet else []) + cmd_extra + extra,
                       check=True, timeout=timeout, cwd=d, stdout=subprocess.DEVNULL,
                       stderr=None if os.environ.get("MUSCLE_GPU_TIMING") else subprocess.DEVNULL,
                       env=None if env is None else dict(os.environ, **env))
        with open(out, "rb") as f:
            data = f.read()
    return hashlib.md5(data).hexdigest(), data


def golden_md5():
    with open(GOLDEN) as f:
        return json.load(f)


def run_profseq(binary, fixture, threads=2, timeout=600):
    """`muscle -profseq msa.afa -input2 query.fa` (profseq.cpp:59-100) on the alignment and the query of a bp_* fixture
    (tests/golden/make_golden.py bp): -> the path it logs. The command's hot path is CalcPosterior for the (row, query) pairs and
    MPCFlat::BuildPost + CalcAlnFlat."""
    import re
    z = G.load(fixture)
    seqs = [str(x) for x in z["seqs"]]
    rows = [str(x) for x in z["ps_rows"]]
    n = len(seqs)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "msa.afa"), "w") as f:
            for i, r in enumerate(rows):
                f.write(">s%d\n%s\n" % (i, r))
        write_fasta(os.path.join(d, "q.fa"), [seqs[n - 1]], ["s%d" % (n - 1)])
        subprocess.run([binary, "-profseq", "msa.afa", "-input2", "q.fa", "-log", "ps.log", "-threads", str(threads), "-quiet"],
                       check=True, timeout=timeout, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(os.path.join(d, "ps.log")) as f:
            paths = [ln.strip() for ln in f if re.fullmatch(r"[BXY]+", ln.strip())]
    assert len(paths) == 1, paths
    return paths[0], str(z["ps_path"])
