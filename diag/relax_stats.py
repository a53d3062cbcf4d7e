"""Row statistics of the relax operands (oracle store, CPU): entries per row, column spans, and the wave-level step counts of
candidate merge forms. diag only."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _oracle as O, _golden as G
from muscle_amd.synth import make_family, read_fasta

def main():
    kind, n = sys.argv[1], int(sys.argv[2])
    if kind == "synth":
        seqs = make_family(n, 400, 1)
    else:
        seqs = read_fasta(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rdrp_first1000.fa.gz"))[:n]
    s, t, m, i, thr = G.hmm_tables("hmm_amino")
    h = O.make_hmm(s, t, m, i)
    st = O.Store(seqs)
    st.calc_posteriors(h, threads=8)
    L = [len(x) for x in seqs]
    # ordered-pair CSR: rows[A][Z] = (off, cols)
    M = {}
    for k, (a, b) in enumerate(st.pairs()):
        off, val = st.get(k)
        cols = O.val_cols(val).astype(np.int64)
        off = off.astype(np.int64)
        M[(a, b)] = (off, cols)
        # transpose
        rows = np.repeat(np.arange(L[a]), np.diff(off))
        order = np.lexsort((rows, cols))
        tc = rows[order]
        cnt = np.bincount(cols, minlength=L[b])
        toff = np.concatenate([[0], np.cumsum(cnt)])
        M[(b, a)] = (toff, tc)
    cnts, spans = [], []
    for (a, z), (off, cols) in M.items():
        c = np.diff(off)
        cnts.append(c)
        nz = c > 0
        first = cols[off[:-1][nz]]
        last = cols[off[1:][nz] - 1]
        sp = np.zeros(len(c), np.int64)
        sp[nz] = last - first + 1
        spans.append(sp)
    cnts = np.concatenate(cnts); spans = np.concatenate(spans)
    print("seqs", n, "mean len", np.mean(L), "max", max(L))
    print("entries/row mean %.2f; hist" % cnts.mean(), np.bincount(np.minimum(cnts, 16)) / len(cnts))
    print("span hist (0..)", np.round(np.bincount(np.minimum(spans, 40)) / len(spans), 4))
    for w in (4, 6, 8, 12, 16, 24, 32, 64):
        print("span>%d: %.5f" % (w, (spans > w).mean()))
    # per record: max span
    # wave-level steps: cells of pair (X,Y) in stored order, groups of 64, for each Z
    rng = np.random.default_rng(1)
    pairs = st.pairs()
    sel = rng.choice(len(pairs), size=min(40, len(pairs)), replace=False)
    res = {k: [] for k in ("merge2", "x2", "x4", "min2", "min4", "lane_x", "lane_min")}
    for k in sel:
        X, Y = pairs[k]
        off, cols = M[(X, Y)]
        rows = np.repeat(np.arange(L[X]), np.diff(off))
        for Z in rng.choice(n, size=min(12, n), replace=False):
            if Z == X or Z == Y: continue
            cx = np.diff(M[(X, Z)][0])[rows]
            cy = np.diff(M[(Y, Z)][0])[cols]
            pad = (-len(cx)) % 64
            cxp = np.concatenate([cx, np.zeros(pad, np.int64)]).reshape(-1, 64)
            cyp = np.concatenate([cy, np.zeros(pad, np.int64)]).reshape(-1, 64)
            b = lambda v, w: np.maximum((v + w - 1) // w, 1)
            res["x2"].append(b(cxp, 2).max(1)); res["x4"].append(b(cxp, 4).max(1))
            mn = np.minimum(cxp, cyp)
            res["min2"].append(b(mn, 2).max(1)); res["min4"].append(b(mn, 4).max(1))
            res["lane_x"].append(cxp.mean(1)); res["lane_min"].append(mn.mean(1))
    for k2, v in res.items():
        if v: print(k2, "mean per wave: %.3f" % np.concatenate(v).mean())
main()
