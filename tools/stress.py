"""Randomised cross-check (run ON the GPU box): extractor and detector against the oracle on random frame sizes and seeds.
    python tools/stress.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from orb_slam2_aruco_amd import binding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0; refused = 0
for case in range(n):
    big = case % 7 == 3
    cols, rows = int(rng.integers(200, 2600 if big else 1000)), int(rng.integers(160, 1500 if big else 760))
    rows = max(rows, cols // 12)               # more than 16 roots is a capacity limit of the quadtree kernels
    if rows > cols: cols, rows = rows, cols   # portrait frames give nIni = 0 in DistributeOctTree: undefined in the reference, rejected here
    nf, nl = int(rng.integers(50, 6000)), int(rng.integers(1, 13))
    sc = [1.2, 1.2, 1.1, 1.44, 2.0][int(rng.integers(0, 5))]
    ini, mn = [(20, 7), (20, 7), (30, 10), (12, 5), (7, 7)][int(rng.integers(0, 5))]
    while min(cols, rows) / sc ** (nl - 1) < 70:
        nl -= 1
    dic = ["ARUCO", "ARUCO_MIP_36h12", "ARUCO_MIP_25h7", "TAG36h11"][case % 4]
    img, _ = synth.scene(rows, cols, int(rng.integers(1, 10 ** 6)), dic, int(rng.integers(0, 5)), side_range=(30, max(31, min(rows, cols) // 4)))
    if case % 5 == 0:
        salt = rng.random(img.shape) < 0.1
        img = np.where(salt, rng.integers(0, 256, img.shape), img).astype(np.uint8)
    try:
        ex = binding.ORBextractor(nf, sc, nl, ini, mn); ora = O.OrbOracle(nf, sc, nl, ini, mn)
        k, d = ex(img); ok_, od = ora.extract(img)
        why = []
        if not (len(k) == len(ok_) and all(np.array_equal(k[f], ok_[f]) for f in ("x", "y", "octave", "response", "angle", "size"))): why.append("keypoints %d vs %d" % (len(k), len(ok_)))
        elif not np.array_equal(d, od): why.append("descriptors")
        det = binding.MarkerDetector(dic); oa = O.ArucoOracle(dic)
        g, w = det.detect(img), oa.detect(img)
        if not np.array_equal(det.thresholded(0), oa.stage_image(0)): why.append("threshold")
        if not (np.array_equal(g["id"], w["id"]) and np.allclose(g["corners"], w["corners"], atol=1e-3)): why.append("markers %s vs %s" % (g["id"].tolist(), w["id"].tolist()))
        c = det.counts(0)
        if c["flags"]: why.append("flags %d" % c["flags"])
        good = not why
        if why: print("   ", why)
    except binding.OrbfeError as e:
        if "(-4)" in str(e) or "quadtree roots (supported" in str(e):   # ORBFE_ERR_CAPACITY / more than 16 quadtree roots (aspect ratio above 16.5): documented limits (include/orbfe.h), refused loudly -- not a wrong result
            refused += 1
            print("case %d %dx%d nf %d nl %d sc %.2f: refused (capacity): %s" % (case, cols, rows, nf, nl, sc, str(e)[:90]))
            continue
        good = False
        print("case %d: exception %r" % (case, e))
    except Exception as e:
        good = False
        print("case %d %dx%d nf %d nl %d sc %.2f th %d/%d %s: exception %r" % (case, cols, rows, nf, nl, sc, ini, mn, dic, e))
    if not good:
        bad += 1
        print("case %d %dx%d nf %d nl %d sc %.2f th %d/%d %s: MISMATCH" % (case, cols, rows, nf, nl, sc, ini, mn, dic))
print("%d cases, %d mismatches, %d refused at a documented limit" % (n, bad, refused))
