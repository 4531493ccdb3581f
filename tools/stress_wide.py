"""The guided searches of the widened rows over many seeds (run ON the GPU box): the parametrised GPU tests of tests/test_match_gpu.py
called directly with seeds the suite does not use.    python tools/stress_wide.py [n_seeds] [first_seed]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from orb_slam2_aruco_amd import binding as B
import test_match_gpu as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(s0)
bad = 0
for seed in range(s0, s0 + n):
    jobs = [("last_frame", lambda: T.test_search_by_projection_last_frame(B, O, seed, float(rng.choice([7.0, 15.0, 30.0])), bool(rng.integers(0, 2)))),
            ("fuse", lambda: T.test_fuse_search(B, O, seed, float(rng.choice([2.5, 3.0, 4.0, 10.0])), float(rng.choice([5.99, 0.0])))),
            ("keyframe", lambda: T.test_search_by_projection_keyframe(B, O, seed, 10.0, 100)),
            ("sim3", lambda: T.test_search_by_sim3(B, O, seed, float(rng.choice([4.0, 7.5])), float(rng.choice([1.0, 1.03, 0.97])))),
            ("projection_sim3", lambda: T.test_search_by_projection_sim3(B, O, seed, int(rng.choice([4, 10, 15])))),
            ("search_init", lambda: T.test_search_for_initialization(B, O, 1000 + seed, 100, 0.9, True))]
    for name, fn in jobs:
        try:
            fn()
        except AssertionError as e:
            bad += 1
            print("seed %d %s: ASSERT %s" % (seed, name, str(e)[:200]))
        except Exception as e:
            bad += 1
            print("seed %d %s: %r" % (seed, name, e))
print("%d seeds, %d failures" % (n, bad))
