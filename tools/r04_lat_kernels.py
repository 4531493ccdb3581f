"""ON THE GPU BOX (under rocprofv3 --kernel-trace --stats): 40 single-frame detector calls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_aruco_amd import binding, synth
frames = synth.stream(480, 640, 8, 1000, "ARUCO", n_markers=4)
det = binding.MarkerDetector("ARUCO")
for i in range(40):
    det.detect(frames[i % 8])
