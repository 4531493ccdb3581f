"""Seeded synthetic DBoW2 vocabularies for the CPU (oracle) and GPU (parity) tests.  ORBvoc.txt is external to the reference
(README: downloaded separately), so the tests build hierarchical k-ary trees of 256-bit descriptors in the same text format
(TemplatedVocabulary::saveToTextFile, TemplatedVocabulary.h:1429-1450): children = parent with random bits flipped, fewer
flips deeper down; optional irregularities: nodes with fewer than k children, leaves above level L, zero-weight words
(stop words) and duplicate sibling descriptors (distance ties)."""
import numpy as np


def make(k=10, L=3, seed=0, irregular=True, weighting=0, scoring=0):
    rng = np.random.default_rng(seed)
    parent, is_leaf, desc, weight = [], [], [], []
    bits_root = rng.integers(0, 2, 256, dtype=np.uint8)

    def add(pid, bits, leaf):
        parent.append(pid); is_leaf.append(1 if leaf else 0)
        desc.append(np.packbits(bits))
        if leaf:
            w = float(rng.uniform(0.5, 9.0)) if weighting in (0, 2) else 1.0
            if irregular and rng.random() < 0.03:
                w = 0.0                                   # a stop word: features landing here are dropped
            weight.append(w)
        else:
            weight.append(0.0)
        return len(parent)                                # node id (root = 0)

    def grow(pid, bits, level):
        nchild = k if not irregular or rng.random() > 0.15 else int(rng.integers(1, k + 1))
        kids = []
        for c in range(nchild):                           # siblings are created together (HKmeansStep), then expanded
            b = bits.copy()
            flip = rng.choice(256, size=max(2, 96 >> level), replace=False)
            b[flip] ^= 1
            if irregular and c > 0 and rng.random() < 0.05:
                b = kids[-1][1].copy()                    # duplicate sibling descriptor -> tie, first wins
            leaf = level == L or (irregular and level >= 2 and rng.random() < 0.05)
            kids.append((add(pid, b, leaf), b, leaf))
        for nid, b, leaf in kids:
            if not leaf:
                grow(nid, b, level + 1)

    grow(0, bits_root, 1)
    return dict(k=k, L=L, scoring=scoring, weighting=weighting, parent=np.array(parent, np.int32),
                is_leaf=np.array(is_leaf, np.uint8), desc=np.array(desc, np.uint8), weight=np.array(weight, np.float64))


def write_text(voc, path, trailing_newline=True):
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (voc["k"], voc["L"], voc["scoring"], voc["weighting"]))
        n = len(voc["parent"])
        for i in range(n):
            line = "%d %d %s %r" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]),
                                    float(voc["weight"][i]))
            f.write(line + ("\n" if (i + 1 < n or trailing_newline) else ""))


def features(voc, n, seed, noise=24):
    """n query descriptors: random leaves' descriptors with `noise` bits flipped (plus a few very sparse ones that sit
    closest to an all-zero node)."""
    rng = np.random.default_rng(seed)
    leaves = np.flatnonzero(voc["is_leaf"])
    pick = rng.choice(leaves, n)
    bits = np.unpackbits(voc["desc"][pick], axis=1)
    for i in range(n):
        bits[i, rng.choice(256, noise, replace=False)] ^= 1
    sparse = rng.choice(n, max(1, n // 50), replace=False)
    bits[sparse] = 0
    bits[sparse, rng.integers(0, 256, len(sparse))] = 1
    return np.packbits(bits, axis=1)
