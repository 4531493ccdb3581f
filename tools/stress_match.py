"""Randomised cross-check of the matching entry points against the oracle (run ON the GPU box): knn2 (all sizes incl. 0 and 1,
clustered descriptors so that ties occur), SearchForInitialization, SearchByProjection modes 0 / 1.
    python tools/stress_match.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from orb_slam2_aruco_amd import binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def descs(m, nclusters):
    """m descriptors around nclusters centres (a few flipped bits each): near-duplicates and exact ties"""
    c = rng.integers(0, 256, (max(nclusters, 1), 32), dtype=np.uint8)
    d = c[rng.integers(0, len(c), m)].copy()
    flips = rng.integers(0, 12, m)
    for i in range(m):
        for b in rng.integers(0, 256, flips[i]):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def kps(m, cols, rows, nlevels=8):
    k = np.zeros(m, binding.KP_DTYPE)
    k["octave"] = np.minimum(rng.geometric(0.35, m) - 1, nlevels - 1)
    k["x"] = rng.uniform(16, cols - 16, m).astype(np.float32)
    k["y"] = rng.uniform(16, rows - 16, m).astype(np.float32)
    k["angle"] = rng.uniform(0, 360, m).astype(np.float32)
    k["size"] = 31.0 * 1.2 ** k["octave"]
    k["response"] = rng.integers(7, 200, m)
    k["class_id"] = -1
    return k


for case in range(n):
    why = []
    try:
        nq, nt = int(rng.integers(0, 2500)), int(rng.integers(0, 2500))
        if case % 9 == 0: nq = int(rng.integers(0, 3))
        if case % 11 == 0: nt = int(rng.integers(0, 3))
        Q, T = descs(nq, int(rng.integers(1, 40))), descs(nt, int(rng.integers(1, 40)))
        init = [256, 256, 100, 50][int(rng.integers(0, 4))]
        g, w = binding.knn2(Q, T, init), O.knn2(Q, T, init)
        if not all(np.array_equal(a, b) for a, b in zip(g, w)): why.append("knn2 %d x %d init %d" % (nq, nt, init))
        cols, rows = int(rng.integers(320, 1400)), int(rng.integers(240, 900))
        n1, n2 = int(rng.integers(0, 2200)), int(rng.integers(0, 2200))
        k1, k2 = kps(n1, cols, rows), kps(n2, cols, rows)
        d1 = descs(n1, int(rng.integers(1, 60)))
        d2 = np.concatenate([d1[rng.integers(0, max(n1, 1), n2 // 2)] if n1 else descs(n2 // 2, 5), descs(n2 - n2 // 2, 20)])[:n2] if n2 else descs(0, 1)
        win = [100, 100, 50, 200][int(rng.integers(0, 4))]
        gn, gm, gp = binding.ORBmatcher(0.9, True).SearchForInitialization(k1, d1, k2, d2, cols, rows, None, win)
        wn, wm, wp = O.search_for_initialization(k1, d1, k2, d2, cols, rows, None, win, 0.9, True)
        if not (gn == wn and np.array_equal(gm, wm) and np.array_equal(gp, wp)): why.append("SearchForInitialization %d vs %d, window %d (%d / %d matches)" % (n1, n2, win, gn, wn))
        nqp = int(rng.integers(0, 1500))
        q = np.zeros(nqp, binding.WINDOW_QUERY_DTYPE)
        src = rng.integers(0, max(n2, 1), nqp)
        q["x"] = (k2["x"][src] if n2 else rng.uniform(0, cols, nqp)) + rng.normal(0, 2, nqp)
        q["y"] = (k2["y"][src] if n2 else rng.uniform(0, rows, nqp)) + rng.normal(0, 2, nqp)
        q["r"] = rng.uniform(2, 30, nqp)
        lv = k2["octave"][src] if n2 else np.zeros(nqp, np.int32)
        q["min_level"] = lv - 1; q["max_level"] = lv
        qd = d2[src] if n2 else descs(nqp, 3)
        taken = (rng.random(n2) < 0.1).astype(np.uint8)
        for mode in (0, 1):
            g = binding.search_by_projection(k2, d2, cols, rows, q, qd, taken.copy(), mode, 100, 0.8)
            w = O.search_by_projection(k2, d2, cols, rows, q, qd, taken.copy(), mode, 100, 0.8)
            for name in w:
                if not np.array_equal(np.asarray(g[name]), np.asarray(w[name])): why.append("search_by_projection mode %d: %s (%d queries, %d keypoints)" % (mode, name, nqp, n2)); break
    except Exception as e:
        why.append("exception %r" % (e,))
    if why:
        bad += 1
        print("case %d: %s" % (case, why))
print("%d cases, %d mismatches" % (n, bad))
