"""ON THE GPU BOX: HIP-event stage times of single-frame calls (the drop-in path), detector and extractor, tiled vs one-workgroup contours."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from orb_slam2_aruco_amd import binding, synth
frames = synth.stream(480, 640, 16, 1000, "ARUCO", n_markers=4)
det = binding.MarkerDetector("ARUCO"); ex = binding.ORBextractor(1000, 1.2, 8, 20, 7)
for mode in (None, False):
    det.set_tiled_contours(mode)
    det.enable_kernel_timing(True)
    ts, wall = [], []
    for i in range(60):
        t0 = time.perf_counter(); det.detect(frames[i % 16]); wall.append(time.perf_counter() - t0)
        ts.append(det.kernel_times_us())
    t = np.median(np.array(ts[10:]), axis=0)
    print("detect, contours %s: wall %.3f ms (timing on)" % ("tiled" if mode is None else "one workgroup", np.median(wall[10:]) * 1e3), dict(zip(binding.MarkerDetector.STAGES, [round(float(v), 1) for v in t])), "sum %.1f us" % t.sum())
    det.enable_kernel_timing(False)
    wall = []
    for i in range(60):
        t0 = time.perf_counter(); det.detect(frames[i % 16]); wall.append(time.perf_counter() - t0)
    print("   wall without timing %.3f ms" % (np.median(wall[10:]) * 1e3))
ex.enable_kernel_timing(True)
ts = []
for i in range(60):
    ex(frames[i % 16]); ts.append(ex.kernel_times_us())
t = np.median(np.array(ts[10:]), axis=0)
print("extract:", dict(zip(binding.ORBextractor.stage_names(len(t)), [round(float(v), 1) for v in t])), "sum %.1f us" % t.sum())
