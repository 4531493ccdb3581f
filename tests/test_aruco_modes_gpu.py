"""GPU parity of the detector's parameter surface outside src/Frame.cc:135-137 (csrc/aruco_modes.hip) against the CPU oracle:
DM_FAST / DM_VIDEO_FAST (THRES_AUTO_FIXED with its rand() retries and frame-to-frame threshold, the automatic size estimation),
Params::minSize > 0 (reduced working image + cornerUpsample), CORNER_SUBPIX, CV_8UC3 input, cv::cornerSubPix by itself.
rand() is the process's sequence on both sides: every sequence is run once per side behind the same srand()."""
import ctypes

import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

pytestmark = pytest.mark.gpu

LIBC = ctypes.CDLL(None)
CORNER_TOL = 1e-3   # px; ids, thresholds, attempts, working sizes and minSize are compared exactly


def frames(n, rows=480, cols=640, dic="ARUCO", seed0=70):
    """A video-like sequence: scenes with markers, a dark one (every marker pixel below the start threshold: THRES_AUTO_FIXED has
    to retry with random thresholds), one without markers, a bright one."""
    out = []
    for i in range(n):
        img, _ = synth.scene(rows, cols, seed0 + i, dic, 3 + i % 3, side_range=(60, 130))
        if i % 5 == 2:
            img = (img.astype(np.float32) * 0.22).astype(np.uint8)          # max ~56: nothing at threshold 100+
        elif i % 5 == 3:
            img, _ = synth.scene(rows, cols, seed0 + i, dic, 0)
        elif i % 5 == 4:
            img = np.clip(img.astype(np.int32) + 70, 0, 255).astype(np.uint8)
        out.append(img)
    return out


# tests/golden/aruco_modes_seq.npz (tests/gen_golden.py modes): name -> (mode, minMarkerSize, corner method, enclosed, trackingMinDetections)
GOLDEN_CONFIGS = {"fast_lines": (1, 0.0, 1, False, 0), "video_subpix": (2, 0.0, 0, False, 0), "normal_min06_subpix": (0, 0.06, 0, False, 0),
                  "fast_enclosed_track": (1, 0.0, 1, True, 1), "normal_track2_none": (0, 0.0, 2, False, 2)}


def golden_sequence():
    """Ten frames: one scene, two of its markers painted over in some frames, a dark, an empty and a shifted frame."""
    img, truth = synth.scene(480, 640, 72, "ARUCO", 4, side_range=(60, 130))
    d0 = _damaged(img, truth[0][1]); d1 = _damaged(d0, truth[1][1])
    dark = (img.astype(np.float32) * 0.22).astype(np.uint8)
    return [img, img, d0, np.roll(d0, 2, axis=1), dark, img, d1, np.full_like(img, 90), img, d0]


def test_golden_sequences_without_the_oracle(orbfe):
    """The library against the committed fixture (generated from the oracle, tests/gen_golden.py modes): no oracle at run time."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "aruco_modes_seq.npz"))
    seq = golden_sequence()
    assert int(sum(int(f.astype(np.int64).sum()) for f in seq)) == int(g["frames_sum"][0])
    for name, (mode, ms, corner, enclosed, track) in GOLDEN_CONFIGS.items():
        det = orbfe.MarkerDetector("ARUCO")
        det.detectEnclosedMarkers(enclosed); det.setCornerRefinementMethod(corner); det.setDetectionMode(mode, ms); det.setTracking(track)
        LIBC.srand(5)
        for i, im in enumerate(seq):
            m = det.detect(im)
            st = det.state()
            want = g[name + "_state"][i]
            assert [st["threshold"], st["attempts"], st["work_shape"][0], st["work_shape"][1], det.tracked(), len(m)] == want.tolist(), (name, i)
            n = min(len(m), 8)
            assert np.array_equal(m["id"][:n], g[name + "_ids"][i][:n]) and np.all(g[name + "_ids"][i][n:] == -1), (name, i)
            assert np.allclose(m["corners"][:n], g[name + "_corners"][i][:n], atol=CORNER_TOL), (name, i)


def run_both(orbfe, oracle, seq, dic, mode, min_size, corner, seed=11, enclosed=False):
    det, ora = orbfe.MarkerDetector(dic), oracle.ArucoOracle(dic)
    det.detectEnclosedMarkers(enclosed); ora.detect_enclosed_markers(enclosed)
    # the order a caller of the reference uses with CORNER_SUBPIX: the corner method first, else it resets minSize (markerdetector.cpp:392-395)
    det.setCornerRefinementMethod(corner); ora.set_corner_method(corner)
    det.setDetectionMode(mode, min_size); ora.set_detection_mode(mode, min_size)
    LIBC.srand(seed)
    want = [(ora.detect(im), ora.state()) for im in seq]
    LIBC.srand(seed)
    got = [(det.detect(im), det.state()) for im in seq]
    return got, want


def compare(got, want):
    nretry = 0
    for i, ((g, gs), (w, ws)) in enumerate(zip(got, want)):
        assert gs["attempts"] == ws["attempts"] and gs["threshold"] == ws["threshold"], (i, gs, ws)
        assert tuple(gs["work_shape"]) == tuple(ws["work_shape"]), (i, gs, ws)
        assert np.float32(gs["min_size"]) == np.float32(ws["min_size"]), (i, gs, ws)
        assert np.array_equal(g["id"], w["id"]), (i, g["id"], w["id"])
        assert np.allclose(g["corners"], w["corners"], atol=CORNER_TOL), (i, np.abs(g["corners"] - w["corners"]).max())
        nretry += ws["attempts"] > 1
    return nretry


def test_dm_fast_sequence_with_random_retries(orbfe, oracle):
    seq = frames(10)
    got, want = run_both(orbfe, oracle, seq, "ARUCO", 1, 0.0, 1)
    assert compare(got, want) >= 2                                  # the dark and the empty frames went through the retries
    assert sum(len(w) for w, _ in want) >= 15
    assert len({ws["threshold"] for _, ws in want}) >= 3            # the threshold follows the markers' pixels
    # same sequence, another seed: the retries draw other thresholds
    got2, want2 = run_both(orbfe, oracle, seq, "ARUCO", 1, 0.0, 1, seed=12345)
    compare(got2, want2)


@pytest.mark.parametrize("dic,rows,cols", [("ARUCO", 480, 640), ("ARUCO_MIP_36h12", 540, 960)])
def test_corner_subpix_mode(orbfe, oracle, dic, rows, cols):
    seq = frames(4, rows, cols, dic)
    got, want = run_both(orbfe, oracle, seq, dic, 0, 0.0, 0)
    compare(got, want)
    lines = orbfe.MarkerDetector(dic).detect(seq[0])
    assert len(lines) == len(got[0][0]) and not np.array_equal(lines["corners"], got[0][0]["corners"])
    # the batch entry point runs the same kernel
    det = orbfe.MarkerDetector(dic)
    det.setCornerRefinementMethod(det.CORNER_SUBPIX)
    batch = det.detect_batch(np.stack(seq))
    for i in range(len(seq)):
        assert np.array_equal(batch[i], got[i][0])


@pytest.mark.parametrize("mode,min_size", [(0, 0.05), (0, 0.08), (1, 0.05), (0, 0.12)])
def test_min_marker_size_reduced_working_image(orbfe, oracle, mode, min_size):
    seq = frames(5)
    got, want = run_both(orbfe, oracle, seq, "ARUCO", mode, min_size, 0)
    compare(got, want)
    assert all(ws["work_shape"][1] < 640 for _, ws in want)
    assert sum(len(w) for w, _ in want) >= 4


def test_min_marker_size_other_frame_sizes(orbfe, oracle):
    for rows, cols, ms in [(720, 1280, 0.04), (540, 960, 0.06), (480, 752, 0.1)]:
        seq = frames(2, rows, cols, "ARUCO_MIP_25h7", seed0=31)
        got, want = run_both(orbfe, oracle, seq, "ARUCO_MIP_25h7", 0, ms, 0)
        compare(got, want)
        assert want[0][1]["work_shape"][1] < cols


def test_dm_video_fast_follows_the_marker_size(orbfe, oracle):
    seq = frames(10, seed0=120)
    got, want = run_both(orbfe, oracle, seq, "ARUCO", 2, 0.0, 0)
    compare(got, want)
    shapes = {tuple(ws["work_shape"]) for _, ws in want}
    assert len(shapes) >= 3 and (480, 640) in shapes               # reduced after frames with markers, full size after empty ones
    # with CORNER_LINES set AFTER the mode the reference resets minSize to 0: only the automatic size is left
    det, ora = orbfe.MarkerDetector("ARUCO"), oracle.ArucoOracle("ARUCO")
    det.setDetectionMode(det.DM_VIDEO_FAST, 0.1); ora.set_detection_mode(2, 0.1)
    det.setCornerRefinementMethod(det.CORNER_LINES); ora.set_corner_method(1)
    assert det.state()["min_size"] == 0.0 and ora.state()["min_size"] == 0.0
    LIBC.srand(3)
    want = [(ora.detect(im), ora.state()) for im in seq[:4]]
    LIBC.srand(3)
    got = [(det.detect(im), det.state()) for im in seq[:4]]
    compare(got, want)


@pytest.mark.parametrize("mode,min_size,corner", [(0, 0.0, 1), (1, 0.0, 1), (1, 0.0, 0), (2, 0.0, 0), (0, 0.06, 0)])
def test_detect_enclosed_markers(orbfe, oracle, mode, min_size, corner):
    """Params::detectEnclosedMarkers: every rectangle candidate enlarged along its diagonals; with THRES_AUTO_FIXED the thresholded
    image is its own inner edge band (erode with a cross + xor)."""
    seq = frames(6, seed0=210)
    got, want = run_both(orbfe, oracle, seq, "ARUCO", mode, min_size, corner, enclosed=True)
    compare(got, want)
    assert sum(len(w) for w, _ in want) >= 6
    det, ora = orbfe.MarkerDetector("ARUCO"), oracle.ArucoOracle("ARUCO")
    det.detectEnclosedMarkers(True); ora.detect_enclosed_markers(True)
    det.setDetectionMode(mode, 0.0); ora.set_detection_mode(mode, 0.0)
    LIBC.srand(1); w = ora.detect(seq[0]); LIBC.srand(1); g = det.detect(seq[0])
    assert np.array_equal(det.thresholded(0), ora.stage_image(0))
    orects, grects = ora.candidates(0), det.rects(0)
    assert len(grects) == len(orects) and np.array_equal(grects["corners"].reshape(-1, 8), orects[:, :8])   # enlarged, integer-valued
    plain = orbfe.MarkerDetector("ARUCO")
    plain.setDetectionMode(mode, 0.0)
    LIBC.srand(1); plain.detect(seq[0])
    assert not np.array_equal(plain.rects(0)["corners"][:len(grects)], grects["corners"][:len(plain.rects(0))])


def _damaged(img, quad, frac=0.45, val=128):
    """The marker's code painted over with a flat grey (its black border stays): the rectangle is still found, the dictionary rejects it."""
    q = np.asarray(quad, np.float64); c = q.mean(0)
    inner = c + (q - c) * frac
    ys, xs = np.mgrid[0:img.shape[0], 0:img.shape[1]]
    inside = np.ones(img.shape, bool)
    sign = None
    for k in range(4):
        a, b = inner[k], inner[(k + 1) % 4]
        cr = (b[0] - a[0]) * (ys - a[1]) - (b[1] - a[1]) * (xs - a[0])
        sign = np.sign(cr[int(c[1]), int(c[0])]) if sign is None else sign
        inside &= (cr * sign) >= 0
    out = img.copy(); out[inside] = val
    return out


@pytest.mark.parametrize("mode,min_size,corner,nmin", [(0, 0.0, 1, 1), (0, 0.0, 1, 2), (1, 0.0, 0, 1), (2, 0.0, 0, 1), (0, 0.05, 0, 1)])
def test_marker_tracking_adopts_rejected_candidates(orbfe, oracle, mode, min_size, corner, nmin):
    """Params::trackingMinDetections: a marker found in enough calls and missing now is recovered from the candidates the dictionary
    rejected (centre inside its last outline, similar area), with its id and the corner order of its previous orientation."""
    img, truth = synth.scene(480, 640, 72, "ARUCO", 4, side_range=(60, 130))
    d0, d1 = _damaged(img, truth[0][1]), _damaged(_damaged(img, truth[0][1]), truth[1][1])
    shifted = np.roll(d0, 3, axis=1)                         # the scene moved by three pixels: still inside the last outline
    seq = [img, img, img, d0, shifted, img, d1, d1, np.full_like(img, 90), img]
    det, ora = orbfe.MarkerDetector("ARUCO"), oracle.ArucoOracle("ARUCO")
    det.setCornerRefinementMethod(corner); ora.set_corner_method(corner)
    det.setDetectionMode(mode, min_size); ora.set_detection_mode(mode, min_size)
    det.setTracking(nmin); ora.set_tracking(nmin)
    LIBC.srand(5)
    want = [(ora.detect(im), ora.state(), ora.tracked()) for im in seq]
    LIBC.srand(5)
    got = [(det.detect(im), det.state(), det.tracked()) for im in seq]
    compare([(g, s) for g, s, _ in got], [(w, s) for w, s, _ in want])
    assert [t for _, _, t in got] == [t for _, _, t in want]
    if mode != 2 and min_size == 0.0:
        assert sum(t for _, _, t in want) >= 2                 # something was recovered
    else:                                                      # on a reduced working image the rejected candidates keep ITS coordinates
        assert sum(t for _, _, t in want) == 0                 # (the reference compares them with full-size outlines: nothing matches)
    if mode == 0 and min_size == 0.0:
        ids = want[3][0]["id"].tolist()
        assert truth[0][0] in ids and want[3][2] == 1            # the painted-over marker is back, under its id
    det.setTracking(0)                                        # off: the history is gone, plain detection again
    LIBC.srand(5)
    assert det.tracked() == 0 and truth[0][0] not in det.detect(d0)["id"].tolist()
    with pytest.raises(orbfe.OrbfeError):
        det.setTracking(-1)


def test_bgr_input(orbfe, oracle):
    rng = np.random.default_rng(5)
    img, truth = synth.scene(480, 640, 3, "ARUCO", 4)
    bgr = np.stack([np.clip(img.astype(np.int32) + rng.integers(-12, 13, img.shape), 0, 255).astype(np.uint8) for _ in range(3)], axis=2)
    for bits in (14, 15):
        det, ora = orbfe.MarkerDetector("ARUCO"), oracle.ArucoOracle("ARUCO")
        det.setGrayConversion(bits)
        got, want = det.detect(bgr), ora.detect(bgr, bits15=int(bits == 15))
        assert np.array_equal(got["id"], want["id"]) and len(got) == len(truth)
        assert np.allclose(got["corners"], want["corners"], atol=CORNER_TOL)
        assert np.array_equal(det.thresholded(0), ora.stage_image(0))       # the grey image itself was the same
    # with a camera: the poses entry point takes BGR too
    K = np.array([[520., 0, 320], [0, 520., 240], [0, 0, 1]], np.float32)
    det = orbfe.MarkerDetector("ARUCO")
    mk, poses = det.detect(bgr, (K, np.zeros(5, np.float32), (640, 480)), 0.187)
    mk_g, poses_g = det.detect(oracle.bgr_to_gray(bgr), (K, np.zeros(5, np.float32), (640, 480)), 0.187)
    assert np.array_equal(mk, mk_g) and np.array_equal(poses, poses_g)
    # a strided view (a ROI of a larger frame)
    big = np.zeros((500, 700, 3), np.uint8)
    big[10:490, 20:660] = bgr
    assert np.array_equal(det.detect(big[10:490, 20:660]), det.detect(bgr))


@pytest.mark.parametrize("win,iters,eps", [(4, 12, 0.005), (3, 4, 0.0), (5, 4, 0.0), (2, 30, 0.001), (8, 6, 0.01)])
def test_corner_subpix_primitive(orbfe, oracle, win, iters, eps):
    img, truth = synth.scene(480, 640, 8, "ARUCO", 5)
    rng = np.random.default_rng(win)
    pts = [c for t in truth for c in np.asarray(t[1], np.float32).reshape(4, 2)]
    pts = np.array(pts, np.float32) + rng.uniform(-1.5, 1.5, (len(pts), 2)).astype(np.float32)
    # corners whose window leaves the image on every side, and one in a flat region (singular system: stays)
    edge = np.array([[1.2, 1.7], [638.4, 2.2], [2.5, 477.9], [637.1, 478.3], [320.3, 0.4], [0.3, 240.6], [639.0, 200.0], [100.0, 479.0]], np.float32)
    pts = np.vstack([pts, edge])
    got = orbfe.corner_subpix(img, pts, win, iters, eps)
    want = oracle.corner_subpix(img, pts, win, iters, eps)
    assert np.array_equal(got, want), np.abs(got - want).max()
    assert np.abs(got[:len(pts) - len(edge)] - pts[:len(pts) - len(edge)]).max() > 0.05       # something moved
    flat = np.full((64, 64), 77, np.uint8)
    p = np.array([[30.5, 31.25]], np.float32)
    assert np.array_equal(orbfe.corner_subpix(flat, p, win, iters, eps), p)


def test_mode_setters_and_the_batch_entry_points(orbfe, oracle):
    det = orbfe.MarkerDetector("ARUCO")
    for bad in (lambda: det.setDetectionMode(3), lambda: det.setDetectionMode(det.DM_FAST, 1.5), lambda: det.setDetectionMode(det.DM_NORMAL, -0.1),
                lambda: det.setCornerRefinementMethod(3), lambda: det.setGrayConversion(16)):
        with pytest.raises(orbfe.OrbfeError):
            bad()
    # a reduction below 64 x 48 is refused at detect time, never ignored
    det.setCornerRefinementMethod(det.CORNER_SUBPIX); det.setDetectionMode(det.DM_NORMAL, 0.5)
    with pytest.raises(orbfe.OrbfeError):
        det.detect(synth.scene(480, 640, 1, "ARUCO", 2)[0])
    # host batches of a frame-sequential handle are taken frame by frame, in order
    seq = frames(5)
    a, b = orbfe.MarkerDetector("ARUCO"), orbfe.MarkerDetector("ARUCO")
    a.setDetectionMode(a.DM_FAST); b.setDetectionMode(b.DM_FAST)
    LIBC.srand(9)
    one = [a.detect(im) for im in seq]
    LIBC.srand(9)
    batch = b.detect_batch(np.stack(seq))
    for i in range(len(seq)):
        assert np.array_equal(one[i], batch[i])
    assert a.state() == b.state()
    # the device-pointer batch entry point refuses such a handle and takes a reduced working image (stateless) as it is: in its own
    # process, torch (device memory) has to initialise HIP before the library does
    import os, subprocess, sys
    case = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aruco_modes_device_case.py")
    r = subprocess.run([sys.executable, case], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "device case ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
