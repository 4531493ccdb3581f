import sys, os, subprocess
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from orb_slam2_aruco_amd import synth, binding
rows, cols, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n = 3 if rows * cols <= 1280 * 720 else 1
imgs = synth.stream(rows, cols, n, 77, "ARUCO", n_markers=2 if rows >= 200 else 0)
if mode == "noise":
    rng = np.random.default_rng(rows)
    imgs[-1] = np.clip(imgs[-1].astype(np.int32) + rng.integers(-40, 40, imgs[-1].shape), 0, 255).astype(np.uint8)
det = binding.MarkerDetector("ARUCO")
if mode == "off": det.set_speck_passes(False)
out = det.detect_batch(imgs)
print("ok", rows, cols, mode, [len(o) for o in out], [det.counts(f) for f in range(n)], "retries", det.contour_retries(), flush=True)
