"""The guided searches of the widened rows over many seeds (run ON the GPU box): the parametrised GPU tests of tests/test_match_gpu.py
called directly with seeds the suite does not use.    python tools/stress_wide.py [n_seeds] [first_seed]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from orb_slam2_aruco_amd import binding as B
import test_match_gpu as T
import test_bow_gpu as TB
import test_pose_gpu as TP
import test_keyframe_io_gpu as TK
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
            ("search_init", lambda: T.test_search_for_initialization(B, O, 1000 + seed, 100, 0.9, True)),
            ("bow_transform", lambda: TB.test_transform_matches_oracle(B, O, int(rng.choice([3, 4, 8, 10])), int(rng.integers(2, 6)), seed, int(rng.integers(1, 5)), int(rng.integers(1, 3000)))),
            ("search_by_bow", lambda: TB.test_search_by_bow_keyframe_frame(B, O, seed, int(rng.integers(1, 5)), float(rng.choice([0.7, 0.9])), bool(rng.integers(0, 2)))),
            ("search_by_bow_kf", lambda: TB.test_search_by_bow_keyframe_keyframe(B, O, seed)),
            ("triangulation", lambda: TB.test_search_for_triangulation(B, O, seed, int(rng.integers(1, 5)), bool(rng.integers(0, 2)))),
            ("poses", lambda: TP.test_marker_poses_match_oracle(B, O, seed, float(rng.choice([0.0, 0.3, 1.0])), float(rng.choice([0.05, 0.187, 1.0])))),
            ("keyframe_io", lambda: TK.test_pack_unpack_match_oracle(B, O, int(rng.integers(0, 1001))))]
    for name, fn in jobs:
        try:
            fn()
        except AssertionError as e:
            # (the suite's tests also assert properties of THEIR seeds -- "some match exists" -- that a foreign seed may not have:
            # an assertion without a message from tests/test_bow_gpu.py::test_search_for_triangulation at levelsup 1 is that, seed 106)
            bad += 1
            print("seed %d %s: ASSERT %s" % (seed, name, str(e)[:200]))
        except Exception as e:
            bad += 1
            print("seed %d %s: %r" % (seed, name, e))
print("%d seeds, %d failures" % (n, bad))
