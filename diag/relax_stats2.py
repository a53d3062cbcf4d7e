"""Wide-row statistics of the relax operands on real data (oracle store, CPU): spans, 64-column clusters per row, entries per row,
matches per (cell, Z). diag only."""
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
    M = {}
    for k, (a, b) in enumerate(st.pairs()):
        off, val = st.get(k)
        cols = O.val_cols(val).astype(np.int64)
        off = off.astype(np.int64)
        M[(a, b)] = (off, cols)
        rows = np.repeat(np.arange(L[a]), np.diff(off))
        order = np.lexsort((rows, cols))
        tc = rows[order]
        cnt = np.bincount(cols, minlength=L[b])
        toff = np.concatenate([[0], np.cumsum(cnt)])
        M[(b, a)] = (toff, tc)
    cnts, spans, clusters, clusters32 = [], [], [], []
    for (a, z), (off, cols) in M.items():
        c = np.diff(off)
        cnts.append(c)
        for r in range(len(c)):
            cc = cols[off[r]:off[r + 1]]
            if len(cc) == 0:
                spans.append(0); clusters.append(0); clusters32.append(0); continue
            spans.append(cc[-1] - cc[0] + 1)
            for W, dst in ((64, clusters), (32, clusters32)):
                k, start = 0, -10**9
                for x in cc:
                    if x - start >= W: k += 1; start = x
                dst.append(k)
    cnts = np.concatenate(cnts); spans = np.array(spans); clusters = np.array(clusters); clusters32 = np.array(clusters32)
    nz = cnts > 0
    print("seqs", n, "mean len %.1f" % np.mean(L), "rows", len(cnts), "empty rows %.3f" % (1 - nz.mean()))
    print("entries/row mean %.2f (nonempty %.2f)" % (cnts.mean(), cnts[nz].mean()))
    for w in (2, 4, 6, 8, 10, 12, 16, 24): print("  entries>%d: %.4f" % (w, (cnts > w).mean()))
    for w in (8, 16, 32, 48, 64, 96, 128, 192, 256): print("  span>%d: %.4f" % (w, (spans > w).mean()))
    print("64-col clusters per row hist", np.round(np.bincount(np.minimum(clusters, 8)) / len(clusters), 4))
    print("32-col clusters per row hist", np.round(np.bincount(np.minimum(clusters32, 12)) / len(clusters32), 4))
    print("bytes/row: blocks %.1f  bitmap64-per-cluster %.1f  bitmap32-per-cluster %.1f" % (
        16 * np.maximum((cnts + 1) // 2, 1).mean(), (4 + 8 * clusters + 4 * cnts).mean(), (4 + 8 * clusters32 + 4 * cnts).mean()))
    # matches per (cell, Z)
    rng = np.random.default_rng(1)
    pairs = st.pairs()
    sel = rng.choice(len(pairs), size=min(30, len(pairs)), replace=False)
    mt, nx_, ny_ = [], [], []
    for k in sel:
        X, Y = pairs[k]
        off, cols = M[(X, Y)]
        rows = np.repeat(np.arange(L[X]), np.diff(off))
        for Z in rng.choice(n, size=min(6, n), replace=False):
            if Z == X or Z == Y: continue
            ox, cx = M[(X, Z)]; oy, cy = M[(Y, Z)]
            for x, y in zip(rows[::7], cols[::7]):
                a = cx[ox[x]:ox[x + 1]]; b = cy[oy[y]:oy[y + 1]]
                mt.append(len(np.intersect1d(a, b))); nx_.append(len(a)); ny_.append(len(b))
    mt = np.array(mt)
    print("per (cell,Z): X entries %.2f, Y entries %.2f, matches %.2f; matches hist" % (np.mean(nx_), np.mean(ny_), mt.mean()), np.round(np.bincount(np.minimum(mt, 10)) / len(mt), 3))
main()
