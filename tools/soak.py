"""Determinism soak (run ON the GPU box): 30 runs of the 300-frame batch through the extractor and the detector must hash identically."""
import sys, hashlib, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_aruco_amd import binding, synth
fr = synth.stream(480, 640, 300, 1000)
ex = binding.ORBextractor(1000, 1.2, 8, 20, 7); det = binding.MarkerDetector("ARUCO")
hs = set()
for it in range(30):
    h = hashlib.sha256()
    for k, d in ex.extract_batch(fr):
        h.update(k.tobytes()); h.update(d.tobytes())
    for m in det.detect_batch(fr):
        h.update(m.tobytes())
    hs.add(h.hexdigest())
print("distinct results over 30 runs:", len(hs))
