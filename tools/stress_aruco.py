"""Randomised cross-check of the detector alone (run ON the GPU box): tiny to large frames, noise, flat frames, all dictionaries,
error-correction rates, corner refinement on / off.    python tools/stress_aruco.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from orb_slam2_aruco_amd import binding, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
dics = ["ARUCO", "ARUCO_MIP_36h12", "ARUCO_MIP_25h7", "ARUCO_MIP_16h3", "TAG36h11", "TAG25h9", "TAG16h5"]
for case in range(n):
    kind = case % 6
    if kind == 0: cols, rows = int(rng.integers(40, 200)), int(rng.integers(40, 160))
    elif kind == 1: cols, rows = int(rng.integers(1000, 2000)), int(rng.integers(600, 1100))
    else: cols, rows = int(rng.integers(200, 1000)), int(rng.integers(120, 760))
    dic = dics[int(rng.integers(0, len(dics)))]
    try:
        img, _ = synth.scene(rows, cols, int(rng.integers(1, 10 ** 6)), dic, int(rng.integers(0, 6)), side_range=(24, max(25, min(rows, cols) // 3)))
    except Exception:
        img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
    if kind == 2:
        salt = rng.random(img.shape) < rng.uniform(0.02, 0.5)
        img = np.where(salt, rng.integers(0, 256, img.shape), img).astype(np.uint8)
    if kind == 3 and case % 12 == 3: img[:] = int(rng.integers(0, 256))
    rate = [0.0, 0.0, 0.3, 0.6][int(rng.integers(0, 4))]
    lines = bool(rng.integers(0, 2))
    why = []
    try:
        det = binding.MarkerDetector(dic); oa = O.ArucoOracle(dic)
        det.setDictionary(dic, rate); oa.set_params(rate, lines)
        det.setCornerRefinementMethod(1 if lines else 2)   # CORNER_LINES = 1, CORNER_NONE = 2
        g, w = det.detect(img), oa.detect(img)
        if not np.array_equal(det.thresholded(0), oa.stage_image(0)): why.append("threshold")
        if not (np.array_equal(g["id"], w["id"]) and np.allclose(g["corners"], w["corners"], atol=1e-3)): why.append("markers %s vs %s" % (g["id"].tolist(), w["id"].tolist()))
        c = det.counts(0)
        if c["flags"]: why.append("flags %d" % c["flags"])
    except Exception as e:
        why.append("exception %r" % (e,))
    if why:
        bad += 1
        print("case %d %dx%d %s rate %.1f lines %d: %s" % (case, cols, rows, dic, rate, lines, why))
print("%d cases, %d mismatches" % (n, bad))
