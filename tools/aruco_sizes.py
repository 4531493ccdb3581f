"""ArUco detector stage times and per-frame contour statistics at a given frame size (run on the GPU box):
    python tools/aruco_sizes.py ROWS COLS DICTIONARY MARKERS [NFRAMES]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
rows, cols, dic, K = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
frames = synth.stream(rows, cols, B, 1000, dic, n_markers=K)
det = binding.MarkerDetector(dic)
det.detect_batch(frames)
det.enable_kernel_timing(True)
for legacy in (False, True):
    det.force_legacy_contours(legacy)
    det.detect_batch(frames)
    t0 = time.perf_counter(); det.detect_batch(frames); dt = time.perf_counter() - t0
    print("legacy" if legacy else "relay ", "stages us:", dict(zip(det.STAGES, np.round(det.kernel_times_us(), 1))), "wall %.1f ms" % (dt * 1e3))
    if not legacy:
        cs = [det.counts(f) for f in range(B)]
        print("  nkept mean %.0f max %d; fell_back %d of %d; flags %s" % (np.mean([c["nkept"] for c in cs]), max(c["nkept"] for c in cs),
              sum(c["fell_back"] for c in cs), B, sorted(set(c["flags"] for c in cs))))
