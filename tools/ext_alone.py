"""The ORB extractor alone on the C2 batch (300 frames of 640 x 480, 2000 features, 8 levels): stage times with nothing else on the GPU.
ORBFE_LIB=build/liborbfe_x.so python tools/ext_alone.py [reps]      (A/B of two builds of one kernel without the pipeline around it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
imgs = synth.stream(480, 640, 300, 1000, "ARUCO")
ex = binding.ORBextractor(2000, 1.2, 8, 20, 7)
ex.extract_batch(imgs)
ex.enable_kernel_timing(True)
for _ in range(reps):
    ex.extract_batch(imgs)
t = ex.kernel_times_us(median=True)
print(os.environ.get("ORBFE_LIB", "default"), dict(zip(ex.stage_names(len(t)), np.round(t).astype(int).tolist())))
