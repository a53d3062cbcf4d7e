"""Structure-profile (".mega") test inputs — TEST INFRASTRUCTURE.

The reference reads a .mega file with Mega::FromFile (mega.cpp:119-270): a header
`mega <features> <chains> <gapopen> <gapext>`, per feature `idx name alphasize weight`, a `freqs` line,
the lower triangle of the joint letter-pair frequencies, `logoddsmx` and the lower triangle of a score
matrix, then per chain `chain idx label L` and L lines `idx pos <one symbol per feature>` (feature 0 is
the amino-acid letter, the others 'A'+letter with letter < 16). synth_mega_text() writes such a file
for a synthetic family; fixture_text() returns the committed copy of the reference's own
test_data/mega/BB11001.mega (test data, not source)."""
import math
import os
import random

from muscle_amd.synth import AMINO, make_family

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _g(x):
    return "%.4g" % x


def synth_mega_text(n, length, seed=1, nfeat=8, alphas=None, weights=None):
    rng = random.Random(1000 + seed)
    seqs = make_family(n, length, seed=seed)
    alphas = alphas or ([20] + [rng.choice([3, 6, 10, 12, 16]) for _ in range(nfeat - 1)])
    weights = weights or [round(rng.uniform(0.05, 0.45), 4) for _ in range(nfeat)]
    out = ["mega\t%d\t%d\t0.6855\t0.05188" % (nfeat, n)]
    for f in range(nfeat):
        A = alphas[f]
        # symmetric joint frequencies with a heavy diagonal
        w = [[0.0] * A for _ in range(A)]
        for a in range(A):
            for b in range(a + 1):
                v = rng.uniform(0.2, 1.0) * (6.0 if a == b else 1.0)
                w[a][b] = w[b][a] = v
        tot = sum(sum(r) for r in w)
        joint = [[v / tot for v in r] for r in w]
        marg = [sum(r) for r in joint]
        out.append("%d\t%s\t%d\t%s" % (f, "AA" if f == 0 else "F%d" % f, A, _g(weights[f])))
        out.append("freqs\t" + "\t".join(_g(m) for m in marg))
        for a in range(A):
            out.append("%d\t" % a + "\t".join(_g(joint[a][b]) for b in range(a + 1)))
        out.append("logoddsmx")
        for a in range(A):
            name = AMINO[a] if f == 0 else chr(ord("A") + a)
            out.append("%d\t%s\t" % (a, name) +
                       "\t".join(_g(math.log(joint[a][b] / (marg[a] * marg[b]))) for b in range(a + 1)))
    for i, s in enumerate(seqs):
        out.append("chain\t%d\ts%d\t%d" % (i, i, len(s)))
        for pos, c in enumerate(s):
            syms = [c]
            for f in range(1, nfeat):
                # structure letters follow the residue (so homologous positions tend to agree) with noise
                letter = (AMINO.index(c) * 7 + f * 3) % alphas[f]
                if rng.random() < 0.25:
                    letter = rng.randrange(alphas[f])
                syms.append(chr(ord("A") + letter))
            out.append("%d\t%d\t%s" % (i, pos, "".join(syms)))
    return "\n".join(out) + "\n"


def fixture_text(name="bb11001"):
    with open(os.path.join(GDIR, name + ".mega")) as f:
        return f.read()


def mega_text(name):
    """mega_bb11001 | mega_synth_<n>x<L>_s<seed>[_f<features>]"""
    assert name.startswith("mega_")
    name = name[5:]
    if name.startswith("synth_"):
        parts = name[6:].split("_")
        n, L = parts[0].split("x")
        seed, nfeat = 1, 8
        for p in parts[1:]:
            if p[0] == "s":
                seed = int(p[1:])
            if p[0] == "f":
                nfeat = int(p[1:])
        return synth_mega_text(int(n), int(L), seed=seed, nfeat=nfeat)
    return fixture_text(name)
