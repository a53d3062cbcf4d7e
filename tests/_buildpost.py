"""Test-side restatement of MPCFlat::BuildPost (buildpostflat.cpp:18-106) in numpy float32, in the
reference's loop order (s in MSA1 outer, t in MSA2 inner, rows ascending, entries ascending; weights
1.0f as at mpcflat.cpp:324), for small cases; and helpers to make aligned rows. TEST INFRASTRUCTURE."""
import numpy as np


def pos_to_col(aligned_row):
    """Sequence::GetPosToCol (sequence.cpp:144-154): column of every residue of a gapped row."""
    return np.array([c for c, ch in enumerate(aligned_row) if ch != "-"], np.uint32)


def build_post(store_stage, pairs_index, seq1, seq2, p2c1, p2c2, C1, C2, w1=None, w2=None):
    """store_stage: list over pair index of (offsets, values u32 interleaved {P bits, col});
    pairs_index: dict (i,j)->k for i<j; w1/w2: sequence weights per row (default 1.0f)."""
    post = np.zeros((C1, C2), np.float32)
    for a, S in enumerate(seq1):
        for b, T in enumerate(seq2):
            w = np.float32(1.0 if w1 is None else w1[a]) * np.float32(1.0 if w2 is None else w2[b])  # w1*w2, rounded first
            if S < T:  # buildpostflat.cpp:56-77
                off, val = store_stage[pairs_index[(S, T)]]
                p, col = val[0::2].view(np.float32), val[1::2]
                for i in range(len(off) - 1):
                    for k in range(off[i], off[i + 1]):
                        post[p2c1[a][i], p2c2[b][col[k]]] += w * p[k]
            else:      # buildpostflat.cpp:78-100
                off, val = store_stage[pairs_index[(T, S)]]
                p, col = val[0::2].view(np.float32), val[1::2]
                for i in range(len(off) - 1):
                    for k in range(off[i], off[i + 1]):
                        post[p2c1[a][col[k]], p2c2[b][i]] += w * p[k]
    return post


def random_msa(seqs, idxs, rng, extra=0):
    """A random gapped alignment of the given sequences (same width): rows as strings. extra: more gap columns."""
    width = max(len(seqs[i]) for i in idxs) + int(rng.integers(0, 6)) + extra
    rows = []
    for i in idxs:
        s = seqs[i]
        cols = np.sort(rng.choice(width, size=len(s), replace=False))
        row = ["-"] * width
        for ch, c in zip(s, cols):
            row[c] = ch
        rows.append("".join(row))
    return rows, width
