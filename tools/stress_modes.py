"""Randomised cross-check of the detector's modes outside Frame.cc:135-137 (run ON the GPU box): sequences of frames through one
detector per case -- DM_NORMAL / DM_FAST / DM_VIDEO_FAST, minMarkerSize, all three corner methods, grey and BGR frames, odd
sizes, dark / bright / empty / noisy frames -- against the oracle behind the same srand().
    python tools/stress_modes.py [n_cases] [seed]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from orb_slam2_aruco_amd import binding, synth
LIBC = ctypes.CDLL(None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dics = ["ARUCO", "ARUCO_MIP_36h12", "ARUCO_MIP_25h7", "ARUCO_MIP_16h3", "TAG36h11"]
bad = refused = frames = markers = retries = reduced = tracked = 0
for case in range(n):
    cols, rows = int(rng.integers(200, 1400)), int(rng.integers(150, 800))
    if case % 5 == 0: cols, rows = 640, 480
    dic = dics[int(rng.integers(0, len(dics)))]
    mode = int(rng.integers(0, 3))
    corner = int(rng.integers(0, 3))
    ms = [0.0, 0.0, 0.03, 0.05, 0.08, 0.15][int(rng.integers(0, 6))]
    bgr = bool(rng.integers(0, 4) == 0)
    bits = 14 if rng.integers(0, 2) else 15
    enclosed = bool(rng.integers(0, 3) == 0)
    track = int(rng.integers(0, 3))                # trackingMinDetections
    video = bool(rng.integers(0, 3) == 0)          # one scene, markers painted over at random: what the tracking block is for
    seq = []
    if video:
        from test_aruco_modes_gpu import _damaged
        base, truth = synth.scene(rows, cols, int(rng.integers(1, 10 ** 6)), dic, int(rng.integers(2, 6)), side_range=(40, max(41, min(rows, cols) // 3)))
        for i in range(int(rng.integers(4, 9))):
            img = base
            for t in truth:
                if i >= 2 and rng.random() < 0.4: img = _damaged(img, t[1])
            seq.append(np.roll(img, int(rng.integers(-2, 3)), axis=1))
    for i in range(0 if video else int(rng.integers(2, 6))):
        try:
            img, _ = synth.scene(rows, cols, int(rng.integers(1, 10 ** 6)), dic, int(rng.integers(0, 5)), side_range=(30, max(31, min(rows, cols) // 3)))
        except Exception:
            img = rng.integers(0, 256, (rows, cols), dtype=np.uint8)
        k = int(rng.integers(0, 6))
        if k == 0: img = (img.astype(np.float32) * rng.uniform(0.15, 0.5)).astype(np.uint8)
        elif k == 1: img = np.clip(img.astype(np.int32) + int(rng.integers(30, 100)), 0, 255).astype(np.uint8)
        elif k == 2:
            salt = rng.random(img.shape) < rng.uniform(0.01, 0.1)
            img = np.where(salt, rng.integers(0, 256, img.shape), img).astype(np.uint8)
        if bgr:
            img = np.stack([np.clip(img.astype(np.int32) + rng.integers(-9, 10, img.shape), 0, 255).astype(np.uint8) for _ in range(3)], axis=2)
        seq.append(img)
    why = []
    try:
        det, ora = binding.MarkerDetector(dic), O.ArucoOracle(dic)
        det.setGrayConversion(bits)
        det.detectEnclosedMarkers(enclosed); ora.detect_enclosed_markers(enclosed)
        det.setTracking(track); ora.set_tracking(track)
        det.setCornerRefinementMethod(corner); ora.set_corner_method(corner)
        det.setDetectionMode(mode, ms); ora.set_detection_mode(mode, ms)
        seed = int(rng.integers(1, 10 ** 6))
        LIBC.srand(seed)
        want, ora_tracked = [], []
        for im in seq:
            want.append((ora.detect(im, bits15=int(bits == 15)), ora.state())); ora_tracked.append(ora.tracked())
        LIBC.srand(seed)
        for i, im in enumerate(seq):
            try:
                g = det.detect(im)
            except binding.OrbfeError as e:
                if "below 64 x 48" in str(e): refused += 1; break      # documented refusal of tiny working images
                raise
            gs, (w, ws) = det.state(), want[i]
            frames += 1; markers += len(w); retries += ws["attempts"] > 1; reduced += ws["work_shape"][1] != cols
            if (gs["attempts"], gs["threshold"], tuple(gs["work_shape"])) != (ws["attempts"], ws["threshold"], tuple(ws["work_shape"])) or \
                    np.float32(gs["min_size"]) != np.float32(ws["min_size"]):
                why.append("frame %d state %s vs %s" % (i, gs, ws)); break
            tracked += ora_tracked[i]
            if det.tracked() != ora_tracked[i]:
                why.append("frame %d tracked %d vs %d" % (i, det.tracked(), ora_tracked[i])); break
            if not (np.array_equal(g["id"], w["id"]) and np.allclose(g["corners"], w["corners"], atol=1e-3)):
                why.append("frame %d markers %s vs %s (max corner diff %s)" % (i, g["id"].tolist(), w["id"].tolist(),
                                                                          np.abs(g["corners"] - w["corners"]).max() if len(g) == len(w) and len(g) else None)); break
    except Exception as e:
        why.append("exception %r" % (e,))
    if why:
        bad += 1
        print("case %d %dx%d %s mode %d minSize %.2f corner %d bgr %d/%d enclosed %d track %d video %d: %s" % (case, cols, rows, dic, mode, ms, corner, bgr, bits, enclosed, track, video, why))
print("%d cases, %d frames (%d markers, %d recovered by tracking, %d frames with retries, %d on a reduced image, %d sequences refused as documented), %d mismatches" %
      (n, frames, markers, tracked, retries, reduced, refused, bad))
