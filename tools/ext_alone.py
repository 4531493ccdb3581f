"""The ORB extractor alone on the C2 batch (300 frames of 640 x 480, 2000 features, 8 levels): stage times with nothing else on the GPU.
ORBFE_LIB=build/liborbfe_x.so python tools/ext_alone.py [reps [rows cols frames nfeatures nlevels]]      (A/B of two builds of one kernel
without the pipeline around it; C5 = 10 1080 1920 100 4000 12)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rows, cols, nfr, nfeat, nlev = (int(a) for a in sys.argv[2:7]) if len(sys.argv) > 6 else (480, 640, 300, 2000, 8)
imgs = synth.stream(rows, cols, nfr, 1000, "ARUCO")
ex = binding.ORBextractor(nfeat, 1.2, nlev, 20, 7)
ex.extract_batch(imgs)
ex.enable_kernel_timing(True)
for _ in range(reps):
    ex.extract_batch(imgs)
t = ex.kernel_times_us(median=True)
print(os.environ.get("ORBFE_LIB", "default"), dict(zip(ex.stage_names(len(t)), np.round(t).astype(int).tolist())))
