"""Loaders for the committed fixtures in tests/golden/ (made by tests/golden/make_golden.py)."""
import hashlib
import os

import numpy as np

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GDIR, name + ".npz"), allow_pickle=False)


def hmm_tables(name="hmm_amino"):
    z = load(name)
    return z["start"], z["trans"], z["match"], z["ins"], np.float32(z["min_sparse_score"])


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def stage_digest(stage):
    h = hashlib.sha256()
    for off, val in stage:
        h.update(np.ascontiguousarray(off, np.uint32).tobytes())
        h.update(np.ascontiguousarray(val, np.uint32).tobytes())
    return h.hexdigest()


MPC_SETS = ["n2_L40", "n3_L30", "n8_L60", "ragged", "alpha82", "bb11001", "n32_L150", "bb11005", "n48_L260", "n3_L1100"]
# GPU suite only (minutes of CPU for the oracle): real data, ~7 stored cells per row: records of tens of KB, the 160 KB relax geometry
MPC_SETS_GPU = MPC_SETS + ["rdrp128"]


def mpc(name):
    """-> dict(seqs, ea, nstages, digest[s], nnz[s], and stage[s] = [(off,val)...] when stored in full)"""
    z = load("mpc_" + name)
    seqs = [str(s) for s in z["seqs"]]
    n = len(seqs)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    out = {"seqs": seqs, "ea": z["ea"], "nstages": int(z["nstages"]), "digest": [], "nnz": [], "stage": []}
    for s in range(out["nstages"]):
        out["digest"].append(str(z["digest%d" % s]))
        nnz = z["nnz%d" % s]
        out["nnz"].append(nnz)
        if ("off%d" % s) in z.files:
            offs, vals = z["off%d" % s], z["val%d" % s]
            st, po, pv = [], 0, 0
            for k, (i, j) in enumerate(pairs):
                L = len(seqs[i]) + 1
                st.append((offs[po:po + L], vals[pv:pv + 2 * int(nnz[k])]))
                po += L
                pv += 2 * int(nnz[k])
            out["stage"].append(st)
        else:
            out["stage"].append(None)
    return out


MEGA_SETS = ["mega_bb11001", "mega_synth_6x40_s2", "mega_synth_3x25_s5_f3", "mega_synth_2x70_s7"]


def mega(name):
    """-> dict(seqs, alpha, weight, lp, mx, profs[list of u8 arrays], ea, stage[s], probes...) of a mega_<name>.npz"""
    z = load(name)
    seqs = [str(s) for s in z["seqs"]]
    n, F = len(seqs), len(z["alpha"])
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    profs, po = [], 0
    for s in seqs:
        profs.append(z["profs"][po:po + len(s) * F].copy())
        po += len(s) * F
    out = {"seqs": seqs, "alpha": z["alpha"], "weight": z["weight"], "lp": z["lp"], "mx": z["mx"], "profs": profs,
           "ea": z["ea"], "nstages": int(z["nstages"]), "stage": [], "digest": [], "z": z}
    for s in range(out["nstages"]):
        out["digest"].append(str(z["digest%d" % s]))
        nnz, offs, vals = z["nnz%d" % s], z["off%d" % s], z["val%d" % s]
        st, o, v = [], 0, 0
        for k, (i, j) in enumerate(pairs):
            L = len(seqs[i]) + 1
            st.append((offs[o:o + L], vals[v:v + 2 * int(nnz[k])]))
            o += L
            v += 2 * int(nnz[k])
        out["stage"].append(st)
    return out
