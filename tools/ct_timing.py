import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
img, _ = synth.scene(480, 640, 1)
det = binding.MarkerDetector("ARUCO")
det.detect(img)
imgs = np.stack([img] * 8)
det.detect_batch(imgs)
out = np.zeros(12, np.int64)
binding._check(det.L, det.L.orbfe_aruco_debug_image(det.h, 0, 102, out.ctypes.data_as(C.c_void_p)), "dbg")
print("cycles: load_bits %d, candidates %d, trace %d, sort+retrace %d, approx+compact %d" % tuple(out[:5]))
print(det.counts(0))
