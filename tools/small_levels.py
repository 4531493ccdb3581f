"""Extractor against the oracle on frames whose smallest pyramid level is at the limit the reference itself has (62 pixels a side: one
column of FAST cells), e.g. 320 x 240 with 8 levels (level 7 is 89 x 67).  Run ON the GPU box:  python tools/small_levels.py"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import oracle_lib as O
from orb_slam2_aruco_amd import binding, synth
bad = 0
for (rows, cols, nl, sc, nf) in [(240, 320, 8, 1.2, 500), (160, 409, 5, 1.2, 700), (222, 222, 8, 1.2, 300), (221, 330, 8, 1.2, 400), (62, 62, 1, 1.2, 50), (63, 200, 1, 1.2, 100), (124, 124, 2, 2.0, 200), (100, 150, 3, 1.2, 300), (66, 66, 1, 1.2, 50), (67, 130, 1, 1.5, 80), (64, 64, 1, 1.2, 60), (65, 190, 1, 1.2, 90)]:
    for seed in (1, 2):
        img = synth.scene(480, 640, seed, "ARUCO", 2)[0][100:100 + rows, 120:120 + cols].copy()
        ex = binding.ORBextractor(nf, sc, nl, 20, 7); ora = O.OrbOracle(nf, sc, nl, 20, 7)
        try:
            k, d = ex(img)
        except Exception as e:
            print(rows, cols, nl, "exception", str(e)[:100]); bad += 1; continue
        ok_, od = ora.extract(img)
        same = len(k) == len(ok_) and all(np.array_equal(k[f], ok_[f]) for f in ("x", "y", "octave", "response", "angle", "size")) and np.array_equal(d, od)
        print(rows, cols, nl, sc, "n", len(k), "OK" if same else "MISMATCH")
        bad += not same
print("bad", bad)
