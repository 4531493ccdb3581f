import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
img, _ = synth.scene(480, 640, 1)
ex = binding.ORBextractor(1000, 1.2, 8, 20, 7)
ex.extract_batch(np.stack([img] * 4))
for l in range(8):
    n = C.c_int32(0)
    binding._check(ex.L, ex.L.orbfe_extractor_debug_level_keypoints(ex.h, 0, l, 2, None, 0, C.byref(n)), "dbg")
    v = n.value
    print("level %d: gather %6d  tree %6d  pick %6d cycles; cand %d kept %d" % (
        l, (v & 1023) << 8, ((v >> 10) & 1023) << 8, ((v >> 20) & 1023) << 8, len(ex.level_keypoints(0, l, 0)), len(ex.level_keypoints(0, l, 1))))
