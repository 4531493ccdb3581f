"""Phase timers of the contour kernels (instrumented build: -DORBFE_CT_TIMING, ORBFE_LIB=build/liborbfe_timing.so).
clock64() ticks at 100 MHz on gfx950 (s_memtime), so 1 tick = 10 ns."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_aruco_amd import binding, synth
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
dic = sys.argv[3] if len(sys.argv) > 3 else "ARUCO"
imgs = synth.stream(rows, cols, 300 if rows == 480 else 40, 1000, dic)[::38 if rows == 480 else 5][:8].copy()
det = binding.MarkerDetector(dic)
det.set_tiled_contours(False)      # the one-workgroup relay kernel (a batch of eight frames would take the tiled path)
if os.environ.get("CT_SPECKS") == "0":
    det.set_speck_passes(False)
for legacy in (False, True):
    det.force_legacy_contours(legacy)
    det.detect_batch(imgs)
    for f in (0, 5):
        out = np.zeros(12, np.int64)
        binding._check(det.L, det.L.orbfe_aruco_debug_image(det.h, f, 102, out.ctypes.data_as(C.c_void_p)), "dbg")
        if legacy:
            print("legacy frame %d ticks: load_bits %d, probe %d, long %d, - %d, sort+approx+compact %d" % ((f,) + tuple(out[:5])))
        else:
            print("relay frame %d ticks: load_bits %d, markers %d, small %d, segments %d, lists %d, points %d"
                  % ((f,) + tuple(out[:6])))   # (the tail is three kernels of its own: k_tail_prep / _approx / _finish in the kernel trace)
            print("    phase (c) loop iterations of a wave: max %d, mean %.0f; longest closed small border %d" % (out[6] >> 40, ((out[6] >> 16) & 0xffffff) / 16.0, out[6] & 0xffff))
            print("    phase (c) steps: closed small borders %d, stopped at a grid marker %d, stopped as not canonical %d" % (out[7], out[10], out[11]))
        print("   ", det.counts(f))

det.force_legacy_contours(False)
det.detect_batch(imgs)
