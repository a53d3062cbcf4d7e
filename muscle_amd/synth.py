"""Synthetic protein-family generator; the bench and the parity tests use it.

random.seed(S); ancestor = L letters uniform over the 20 amino acids; each of the N sequences is
the ancestor with, per site: delete w.p. 0.03, else insert a random letter before it w.p. 0.03,
then substitute w.p. 0.30.  Gives ~50 % pairwise identity and ~2 stored posteriors per row.
Labels are s0..s{N-1} (the reference's MPCFlat needs unique labels, mpcflat.cpp:133-135).
"""
import random

AMINO = "ACDEFGHIKLMNPQRSTVWY"


def make_family(n, length, seed=1, p_del=0.03, p_ins=0.03, p_sub=0.30):
    rng = random.Random(seed)
    anc = [rng.choice(AMINO) for _ in range(length)]
    seqs = []
    for _ in range(n):
        out = []
        for c in anc:
            if rng.random() < p_del:
                continue
            if rng.random() < p_ins:
                out.append(rng.choice(AMINO))
            if rng.random() < p_sub:
                c = rng.choice(AMINO)
            out.append(c)
        if not out:
            out = [rng.choice(AMINO)]
        seqs.append("".join(out))
    return seqs


def write_fasta(path, seqs, labels=None):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            f.write(">%s\n%s\n" % (labels[i] if labels else "s%d" % i, s))


def read_fasta(path):
    """Sequences of a FASTA file (.gz accepted), upper-cased like the reference's loader (sequence.cpp:87-88)."""
    import gzip
    seqs, cur = [], []
    with (gzip.open(path, "rt") if path.endswith(".gz") else open(path)) as f:
        for line in f:
            line = line.strip()
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur))
                cur = []
            elif line:
                cur.append(line.upper())
    if cur:
        seqs.append("".join(cur))
    return seqs
