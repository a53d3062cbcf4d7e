"""CPU model of the two-list walk's rounds per (wave, slot, Z) on real data under different CELL ORDERS inside an X group
(oracle posteriors): pair order (shipped for the walk), and cells sorted by a static cost proxy (mean entries per row of the two
rows over all Z). diag only.  usage: python diag/relax_order_model.py rdrp 40"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import _oracle as O, _golden as G
from muscle_amd.synth import make_family, read_fasta

def steps_of(a, b):
    """block steps of the two-list walk on column lists a, b (2 entries per block)"""
    na, nb = (len(a) + 1) // 2, (len(b) + 1) // 2
    if na == 0 or nb == 0: return 1
    ia = ib = 0; s = 0
    while True:
        s += 1
        la = a[min(2 * ia + 1, len(a) - 1)]; lb = b[min(2 * ib + 1, len(b) - 1)]
        adv_a, adv_b = la <= lb, lb <= la
        if (adv_a and ia + 1 >= na) or (adv_b and ib + 1 >= nb): return s
        if adv_a: ia += 1
        if adv_b: ib += 1

def main():
    kind, n = sys.argv[1], int(sys.argv[2])
    seqs = make_family(n, 400, 1) if kind == "synth" else read_fasta(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "rdrp_first1000.fa.gz"))[:n]
    s, t, m, i, thr = G.hmm_tables("hmm_amino")
    st = O.Store(seqs); st.calc_posteriors(O.make_hmm(s, t, m, i), threads=8)
    L = [len(x) for x in seqs]
    M = {}
    for k, (a, b) in enumerate(st.pairs()):
        off, val = st.get(k); cols = O.val_cols(val).astype(np.int64); off = off.astype(np.int64)
        M[(a, b)] = (off, cols)
        rows = np.repeat(np.arange(L[a]), np.diff(off)); order = np.lexsort((rows, cols))
        cnt = np.bincount(cols, minlength=L[b]); M[(b, a)] = (np.concatenate([[0], np.cumsum(cnt)]), rows[order])
    # static proxy: mean entries of row a of sequence A over all partners
    rowcost = [np.zeros(L[A]) for A in range(n)]
    for (A, Z), (off, cols) in M.items(): rowcost[A] += np.diff(off)
    rng = np.random.default_rng(3)
    res = {"pairs": [], "sorted": [], "lane_mean": []}
    # an X group of a 4x2 tile: X fixed, two Y's, a band of 50 rows -> cells; cut in waves of 64
    for trial in range(24):
        X = int(rng.integers(0, n - 2)); Ys = [y for y in rng.choice(np.arange(X + 1, n), size=2, replace=False)]
        r0 = int(rng.integers(0, max(L[X] - 50, 1)))
        cells = []
        for Y in Ys:
            off, cols = M[(X, Y)]
            for x in range(r0, min(r0 + 50, L[X])):
                for y in cols[off[x]:off[x + 1]]: cells.append((x, Y, int(y)))
        if len(cells) < 64: continue
        Zs = [z for z in rng.choice(n, size=10, replace=False) if z != X and z not in Ys]
        S = np.zeros((len(cells), len(Zs)), np.int64)
        for zi, Z in enumerate(Zs):
            ox, cx = M[(X, Z)]
            for ci, (x, Y, y) in enumerate(cells):
                oy, cy = M[(Y, Z)]
                S[ci, zi] = steps_of(cx[ox[x]:ox[x + 1]], cy[oy[y]:oy[y + 1]])
        proxy = np.array([rowcost[X][x] + rowcost[Y][y] for (x, Y, y) in cells])
        for name, order in (("pairs", np.arange(len(cells))), ("sorted", np.argsort(-proxy, kind="stable"))):
            T = S[order]
            nw = len(cells) // 64
            W = T[:nw * 64].reshape(nw, 64, -1).max(1)  # rounds per (wave-slot, Z)
            res[name].append(W.reshape(-1))
        res["lane_mean"].append(S.reshape(-1))
        # a lane walks its cells of a step as a QUEUE (no wave-wide slot boundaries): rounds of a step = max over lanes of the SUM over its slots
        for ns in (2, 3, 4, 6):
            nw = len(cells) // (64 * ns)
            if nw == 0: continue
            T = S[:nw * 64 * ns].reshape(nw, ns, 64, -1)
            res.setdefault("slots%d sum of max" % ns, []).append(T.max(2).sum(1).reshape(-1) / ns)
            res.setdefault("slots%d max of sum" % ns, []).append(T.sum(1).max(1).reshape(-1) / ns)
    for k, v in res.items():
        v = np.concatenate(v); print("%-10s mean %.2f  (n %d)" % (k, v.mean(), len(v)))
main()
