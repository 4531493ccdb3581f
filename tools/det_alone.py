"""The marker detector alone on the C2 batch (300 frames of 640 x 480): stage times with nothing else on the GPU, for every setting of
the speck passes.  python tools/det_alone.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
imgs = synth.stream(480, 640, 300, 1000, "ARUCO")
for name, inside, launch in (("none", False, False), ("inside", True, False), ("launch", False, True)):
    det = binding.MarkerDetector("ARUCO")
    det.set_speck_passes_in_kernel(inside)
    det.set_speck_passes(launch)
    det.detect_batch(imgs)
    det.enable_kernel_timing(True)
    for _ in range(reps):
        det.detect_batch(imgs)
    t = det.kernel_times_us(median=True)
    print(name, dict(zip(det.STAGES, np.round(t).astype(int).tolist())), "ncand frame 0:", det.counts(0)["ncand"])
