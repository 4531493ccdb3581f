"""Run by tests/test_aruco_modes_gpu.py in its own process (torch first): orbfe_aruco_detect_batch_device with the detector modes of
csrc/aruco_modes.hip -- a frame-sequential handle (DM_FAST) is refused, a reduced working image + cornerUpsample (stateless) runs
on device pointers and equals the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib  # noqa: E402
from orb_slam2_aruco_amd import binding as orbfe  # noqa: E402
from test_aruco_modes_gpu import CORNER_TOL, frames  # noqa: E402

seq = frames(5)
d = torch.from_numpy(np.stack(seq)).cuda()
b = orbfe.MarkerDetector("ARUCO")
out = torch.zeros((len(seq), b.capacity, 36), dtype=torch.uint8, device="cuda")
n = torch.zeros(len(seq), dtype=torch.int32, device="cuda")
L = b.L
args = (d.data_ptr(), len(seq), d.stride(0), 480, 640, 640, out.data_ptr(), b.capacity, n.data_ptr(), None)
b.setDetectionMode(b.DM_FAST)
assert L.orbfe_aruco_detect_batch_device(b.h, *args) != 0 and b"frame-sequential" in L.orbfe_last_error()
c, ora = orbfe.MarkerDetector("ARUCO"), oracle_lib.ArucoOracle("ARUCO")
c.setCornerRefinementMethod(c.CORNER_SUBPIX); c.setDetectionMode(c.DM_NORMAL, 0.06)
ora.set_corner_method(0); ora.set_detection_mode(0, 0.06)
assert L.orbfe_aruco_detect_batch_device(c.h, *args) == 0, L.orbfe_last_error()
torch.cuda.synchronize()
rec = out.cpu().numpy().reshape(len(seq), -1).view(orbfe.MARKER_DTYPE).reshape(len(seq), -1)
nn = n.cpu().numpy()
total = 0
for i, im in enumerate(seq):
    want = ora.detect(im)
    assert nn[i] == len(want) and np.array_equal(rec[i, :len(want)]["id"], want["id"]), (i, nn[i], len(want))
    assert np.allclose(rec[i, :len(want)]["corners"], want["corners"], atol=CORNER_TOL)
    total += len(want)
assert total >= 8
print("device case ok", total)
