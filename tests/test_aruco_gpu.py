"""GPU parity: HIP ArUco detector (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    # rows, cols, seed, dictionary, markers
    (480, 640, 1, "ARUCO", 4),
    (480, 640, 2, "ARUCO", 4),
    (480, 640, 9, "ARUCO_MIP_25h7", 5),
    (720, 1280, 3, "ARUCO_MIP_25h7", 6),
    (540, 960, 6, "ARUCO_MIP_36h12", 5),   # what mono_cvcam really feeds (mono_cvcam.cc:124); 40x40 warps
]


@pytest.mark.parametrize("rows,cols,seed,dic,K", CASES)
def test_detect_matches_oracle(orbfe, oracle, rows, cols, seed, dic, K):
    img, truth = synth.scene(rows, cols, seed, dic, K)
    det = orbfe.MarkerDetector(dic)
    got = det.detect(img)
    ora = oracle.ArucoOracle(dic)
    want = ora.detect(img)
    # stage: thresholded image is integer work -> bit exact
    assert np.array_equal(det.thresholded(0), ora.stage_image(0)), "adaptive threshold differs"
    c = det.counts(0)
    assert c["flags"] == 0, c
    # stage: rectangle candidates (corners are integer-valued contour vertices) in the reference's order
    orects = ora.candidates(0)
    grects = det.rects(0)
    assert len(grects) == len(orects), (len(grects), len(orects))
    assert np.array_equal(grects["corners"].reshape(-1, 8), orects[:, :8])
    assert np.array_equal(grects["len"], orects[:, 8].astype(np.int32))
    # final: ids exact (and every planted marker found), corners within 1e-3 px of the oracle
    assert np.array_equal(got["id"], want["id"])
    assert sorted(got["id"].tolist()) == sorted(t[0] for t in truth)
    assert np.allclose(got["corners"], want["corners"], atol=1e-3), np.abs(got["corners"] - want["corners"]).max()


def test_batch_equals_single(orbfe):
    imgs = np.stack([synth.scene(480, 640, 40 + i, "ARUCO", 3)[0] for i in range(4)])
    det = orbfe.MarkerDetector("ARUCO")
    batch = det.detect_batch(imgs)
    for i in range(4):
        one = det.detect(imgs[i])
        assert np.array_equal(one, batch[i])


def test_no_markers_and_flat(orbfe, oracle):
    det = orbfe.MarkerDetector("ARUCO")
    assert len(det.detect(np.full((480, 640), 90, np.uint8))) == 0
    img, _ = synth.scene(480, 640, 5, "ARUCO", 0)
    assert np.array_equal(det.detect(img)["id"], oracle.ArucoOracle("ARUCO").detect(img)["id"])


def test_unknown_dictionary_fails_loudly(orbfe):
    with pytest.raises(orbfe.OrbfeError):
        orbfe.MarkerDetector("NOT_A_DICT")


def test_rotated_markers_all_four_rotations(orbfe, oracle):
    """A marker pasted at 0/90/180/270 degrees decodes to the same id (dictionary_based.cpp:2501-2645)."""
    base = np.full((480, 640), 200, np.uint8)
    m = synth.render_marker("ARUCO", 77, 12, quiet=1)
    det = orbfe.MarkerDetector("ARUCO")
    ora = oracle.ArucoOracle("ARUCO")
    for k in range(4):
        img = base.copy()
        mk = np.rot90(m, k)
        img[100:100 + mk.shape[0], 200:200 + mk.shape[1]] = mk
        got, want = det.detect(img), ora.detect(img)
        assert list(got["id"]) == [77] and list(want["id"]) == [77]
        assert np.allclose(got["corners"], want["corners"], atol=1e-3)


def _rects_key(det, f=0):
    r = det.rects(f)
    return r["corners"].copy(), r["len"].copy()


def test_relay_and_legacy_contour_kernels_agree(orbfe, oracle):
    """The relay-segment kernel (default) and the single-walker kernel (its fallback) give the same rectangles,
    markers and border counts on video-like frames."""
    imgs = synth.stream(480, 640, 6, 321)
    det = orbfe.MarkerDetector("ARUCO")
    a = det.detect_batch(imgs)
    ra = [(_rects_key(det, f), det.counts(f)) for f in range(len(imgs))]
    assert not any(c["fell_back"] for _, c in ra)
    det.force_legacy_contours(True)
    b = det.detect_batch(imgs)
    for f in range(len(imgs)):
        (ca, la), cnt = ra[f]
        cb, lb = _rects_key(det, f)
        c2 = det.counts(f)
        assert np.array_equal(ca, cb) and np.array_equal(la, lb)
        assert (cnt["nkept"], cnt["nrect"], cnt["ncand"]) == (c2["nkept"], c2["nrect"], c2["ncand"])
        assert np.array_equal(a[f], b[f])
    ora = oracle.ArucoOracle("ARUCO")
    assert np.array_equal(a[0]["id"], ora.detect(imgs[0])["id"])


def test_dense_frame_coarsens_the_relay_grid(orbfe, oracle):
    """A frame with more grid markers than the relay kernel's table holds is handled on a coarser grid -- for salt noise
    with no grid at all, i.e. every border followed whole; a batch may mix both kinds."""
    rng = np.random.default_rng(5)
    noisy, _ = synth.scene(480, 640, 11, "ARUCO", 3)
    salt = rng.random(noisy.shape) < 0.35
    noisy = np.where(salt, rng.integers(0, 256, noisy.shape), noisy).astype(np.uint8)
    clean, _ = synth.scene(480, 640, 12, "ARUCO", 3)
    det = orbfe.MarkerDetector("ARUCO")
    got = det.detect_batch(np.stack([clean, noisy, clean]))
    ora = oracle.ArucoOracle("ARUCO")
    for f, img in enumerate([clean, noisy, clean]):
        want = ora.detect(img)
        orects = ora.candidates(0)
        grects = det.rects(f)
        assert det.counts(f)["flags"] == 0
        assert np.array_equal(grects["corners"].reshape(-1, 8), orects[:, :8])
        assert np.array_equal(got[f]["id"], want["id"])


@pytest.mark.parametrize("pattern", ["spiral", "comb", "serpent", "checker", "frame"])
def test_structured_binary_patterns(orbfe, oracle, pattern):
    """Long thin borders, borders between grid lines and one-pixel structures (the relay kernel's corner cases)."""
    img = np.full((240, 320), 200, np.uint8)
    if pattern == "spiral":        # one very long border
        for k in range(0, 100, 8):
            img[20 + k:220 - k, 20 + k:24 + k] = 20
            img[20 + k:24 + k, 20 + k:300 - k] = 20
            img[216 - k:220 - k, 28 + k:300 - k] = 20
            img[28 + k:220 - k, 296 - k:300 - k] = 20
    elif pattern == "comb":        # > 70 border points inside one 32 x 32 grid cell
        for x in range(34, 62, 4):
            img[35:62, x:x + 2] = 20
        img[60:62, 34:60] = 20
    elif pattern == "serpent":     # the comb with a stem across a grid column: one segment of several hundred states between two grid
        for x in range(34, 62, 4):  # markers (the tiled path records a segment's directions in slots of 160 steps and cuts it when one is full)
            img[35:62, x:x + 2] = 20
        img[60:62, 20:60] = 20
        for y in range(100, 128, 4):  # and one across a grid row
            img[y:y + 2, 100:126] = 20
        img[90:126, 124:126] = 20
    elif pattern == "checker":
        yy, xx = np.mgrid[0:240, 0:320]
        img[((yy // 6 + xx // 6) % 2 == 0) & (yy > 30) & (yy < 200) & (xx > 40) & (xx < 280)] = 20
    else:                          # border along the image frame
        img[:] = 20
        img[3:-3, 3:-3] = 200
    det = orbfe.MarkerDetector("ARUCO")
    ora = oracle.ArucoOracle("ARUCO")
    got, want = det.detect(img), ora.detect(img)
    assert np.array_equal(det.thresholded(0), ora.stage_image(0))
    c = det.counts(0)
    assert c["flags"] == 0 and not c["fell_back"], c
    assert c["nkept"] == sum(len(b) > 70 for b in oracle.find_contours(ora.stage_image(0)))
    orects, grects = ora.candidates(0), det.rects(0)
    assert np.array_equal(grects["corners"].reshape(-1, 8), orects[:, :8])
    assert np.array_equal(grects["len"], orects[:, 8].astype(np.int32))
    assert np.array_equal(got["id"], want["id"])


@pytest.mark.parametrize("rows,cols", [(480, 640), (472, 632), (240, 320)])
def test_detector_pyramid_is_the_2x2_mean(orbfe, rows, cols):
    """buildPyramid's exact /2 levels (markerdetector_impl.cpp:1299-1488: resize by exactly 1/2 = 2 x 2 mean, rounded): the
    four-pixels-per-thread kernel against numpy, for row pitches with and without slack behind the last pixel."""
    img = np.random.default_rng(rows).integers(0, 256, (rows, cols), dtype=np.uint8)
    det = orbfe.MarkerDetector("ARUCO")
    det.detect(img)
    ref, level = img, 1
    while ref.shape[1] % 2 == 0 and ref.shape[0] % 2 == 0 and ref.shape[1] // 2 > 70:
        a = ref.astype(np.uint16)
        ref = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        got = det.pyramid_level(level)
        assert got is not None and got.shape == ref.shape and np.array_equal(got, ref), (level, ref.shape)
        level += 1
    assert level >= 3


def test_big_frame_kernel_same_results(orbfe, oracle):
    """The big-frame contour kernel (bit image in HBM, 4096 kept contours) gives the same markers at an ordinary size."""
    img, _ = synth.scene(480, 640, 2, "ARUCO", 4)
    det = orbfe.MarkerDetector("ARUCO")
    want = det.detect(img)
    det.set_big_frames(True)
    got = det.detect(img)
    det.set_big_frames(False)
    assert len(want) > 0 and np.array_equal(got["id"], want["id"]) and np.array_equal(got["corners"], want["corners"])


def test_full_hd_frame_with_many_contours(orbfe, oracle):
    """1920x1080 (config C5): more than 1024 contours longer than 70 points in a frame -> the host entry point redoes the batch
    with the big-frame kernel instead of failing; markers equal the oracle's."""
    img = synth.stream(1080, 1920, 1, 1000, "ARUCO", n_markers=4)[0]
    det = orbfe.MarkerDetector("ARUCO")
    got = det.detect(img)
    want = oracle.ArucoOracle("ARUCO").detect(img)
    assert len(want) > 0 and np.array_equal(got["id"], want["id"])
    assert np.allclose(got["corners"], want["corners"], atol=1e-3)
    assert det.counts(0)["nkept"] > 1024


def test_full_hd_structured_borders_between_grid_lines(orbfe, oracle):
    """1920x1080: the bit image does not fit LDS, so the relay kernel walks it in HBM (k_contours_relay8g) and the borders that touch
    no grid line are k_contours_small's.  Hollow and filled squares and a comb, each with more than 70 border points, placed strictly
    inside 32-pixel grid cells (kept by the small-border kernel), at block and band seams, next to long borders that cross many grid
    lines and rendered markers: kept-contour count, rectangle candidates (order and corners) and markers equal the oracle's, and equal
    the single-walker big-frame kernel's."""
    img = np.full((1080, 1920), 200, np.uint8)
    rng = np.random.default_rng(11)
    for (y0, x0) in [(35, 35), (35, 259), (227, 1027), (515, 1859), (1027, 35), (995, 1795), (259, 515)]:   # cells of the 32-px grid
        img[y0:y0 + 25, x0:x0 + 25] = 20                      # outer border 96 points
    for (y0, x0) in [(99, 99), (547, 771), (803, 1283)]:
        img[y0:y0 + 26, x0:x0 + 26] = 20
        img[y0 + 2:y0 + 24, x0 + 2:x0 + 24] = 200             # hollow: hole border > 70 points as well
    for x in range(1573, 1597, 4):                            # comb inside one cell
        img[163:188, x:x + 2] = 20
    img[186:188, 1573:1595] = 20
    img[300:304, 100:1800] = 20                               # long thin borders across many grid lines
    img[300:900, 100:104] = 20
    img[400:700, 900:1300] = 20
    img[450:650, 950:1250] = 200
    for k, (y0, x0) in enumerate([(720, 300), (60, 1300), (760, 1500)]):
        m = synth.render_marker("ARUCO", 7 + 31 * k, 14, quiet=1)
        img[y0:y0 + m.shape[0], x0:x0 + m.shape[1]] = m
    img = np.clip(img.astype(np.int32) + rng.integers(-3, 4, img.shape), 0, 255).astype(np.uint8)
    det = orbfe.MarkerDetector("ARUCO")
    ora = oracle.ArucoOracle("ARUCO")
    got, want = det.detect(img), ora.detect(img)
    assert np.array_equal(det.thresholded(0), ora.stage_image(0))
    c = det.counts(0)
    assert c["flags"] == 0 and not c["fell_back"], c
    assert c["nkept"] == sum(len(b) > 70 for b in oracle.find_contours(ora.stage_image(0)))
    orects, grects = ora.candidates(0), det.rects(0)
    assert len(orects) >= 10
    assert np.array_equal(grects["corners"].reshape(-1, 8), orects[:, :8])
    assert np.array_equal(grects["len"], orects[:, 8].astype(np.int32))
    assert len(want) == 3 and np.array_equal(got["id"], want["id"]) and np.allclose(got["corners"], want["corners"], atol=1e-3)
    relay_rects = _rects_key(det)
    det.set_big_frames(True)
    got_big = det.detect(img)
    det.set_big_frames(False)
    big_rects = _rects_key(det)                     # (rects of the last run: the big-frame kernel's)
    assert np.array_equal(big_rects[0], relay_rects[0]) and np.array_equal(big_rects[1], relay_rects[1])
    assert np.array_equal(got_big["id"], got["id"]) and np.array_equal(got_big["corners"], got["corners"])


@pytest.mark.parametrize("rows,cols", [(720, 1280), (1080, 1920)])
def test_border_lengths_around_the_tail_kernel_thresholds(orbfe, oracle, rows, cols):
    """k_tail_approx does borders below 256 points four to a wave on 16-lane rows, longer ones one per wave out of a 1024-point LDS
    buffer and still longer ones out of the pool: rectangles (some with a corner cut off: five vertices) whose border lengths step
    through 240 .. 272 and 1000 .. 1050 points, next to a frame-sized border, give the oracle's rectangle candidates in its order."""
    img = np.full((rows, cols), 210, np.uint8)
    k = 0
    for i in range(16):                                   # outer border of a w x h block: 2 (w + h) - 4 points
        w, h = 60 + i, 64
        y0, x0 = 20 + 90 * (i // 8), 20 + 150 * (i % 8)
        img[y0:y0 + h, x0:x0 + w] = 30
        if i % 3 == 2:
            for d in range(14): img[y0 + d, x0 + w - 14 + d:x0 + w] = 210
    for i in range(6):
        w, h = 252 + i * 4, 256
        y0, x0 = 220 + 270 * (i // 4), 20 + 300 * (i % 4)
        if y0 + h + 10 > rows: break
        img[y0:y0 + h, x0:x0 + w] = 30
        img[y0 + 20:y0 + h - 20, x0 + 20:x0 + w - 20] = 210
        if i % 2 == 0:
            for d in range(40): img[y0 + d, x0:x0 + 40 - d] = 210
        k += 1
    img[4:8, 4:cols - 4] = 30; img[rows - 8:rows - 4, 4:cols - 4] = 30       # a border of ~2 (rows + cols) points
    img[4:rows - 4, 4:8] = 30; img[4:rows - 4, cols - 8:cols - 4] = 30
    rng = np.random.default_rng(3)
    img = np.clip(img.astype(np.int32) + rng.integers(-2, 3, img.shape), 0, 255).astype(np.uint8)
    det, ora = orbfe.MarkerDetector("ARUCO"), oracle.ArucoOracle("ARUCO")
    det.detect(img); ora.detect(img)
    assert np.array_equal(det.thresholded(0), ora.stage_image(0))
    c = det.counts(0)
    assert c["flags"] == 0 and not c["fell_back"], c
    lens = np.array([len(b) for b in oracle.find_contours(ora.stage_image(0))])
    assert ((lens >= 230) & (lens < 256)).any() and ((lens >= 256) & (lens < 290)).any(), sorted(lens)
    assert ((lens > 900) & (lens <= 1024)).any() and ((lens > 1024) & (lens < 1200)).any() and (lens > 3000).any(), sorted(lens)
    assert c["nkept"] == (lens > 70).sum()
    orects, grects = ora.candidates(0), det.rects(0)
    assert len(orects) >= 12
    assert np.array_equal(grects["corners"].reshape(-1, 8), orects[:, :8])
    assert np.array_equal(grects["len"], orects[:, 8].astype(np.int32))


def _damaged_markers_image(dic, ids, flips, bit=8):
    """White 480x640 frame with axis-aligned markers; marker k has flips[k] inner cells inverted."""
    img = np.full((480, 640), 235, np.uint8)
    rng = np.random.default_rng(3)
    x = 30
    for mid, nf in zip(ids, flips):
        m = synth.render_marker(dic, mid, bit, quiet=1).copy()
        nb = m.shape[0] // bit - 4
        cells = rng.permutation(nb * nb)[:nf]
        for c in cells:
            cy, cx = 2 + c // nb, 2 + c % nb
            blk = m[cy * bit:(cy + 1) * bit, cx * bit:(cx + 1) * bit]
            m[cy * bit:(cy + 1) * bit, cx * bit:(cx + 1) * bit] = 255 - blk
        s = m.shape[0]
        img[200:200 + s, x:x + s] = m
        x += s + 20
    noise = np.random.default_rng(4).normal(0, 2.0, img.shape)
    return np.clip(np.rint(img + noise), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("dic,rate", [("ARUCO_MIP_36h12", 0.5), ("ARUCO_MIP_36h12", 0.0), ("TAG36h11", 0.3), ("ARUCO_MIP_25h7", 1.0)])
def test_error_correction_rate(orbfe, oracle, dic, rate):
    """MarkerDetector(dict, error_correction_rate) (markerdetector.h:222-229 -> dictionary_based.cpp:1423-1560): damaged markers are
    accepted when closer than int(tau * rate) bits to a dictionary code; ids, order and corners equal the oracle's."""
    ids, flips = [3, 17, 42, 60], [0, 2, 5, 9]
    img = _damaged_markers_image(dic, ids, flips)
    det = orbfe.MarkerDetector(dic)
    det.setDictionary(dic, rate)
    ora = oracle.ArucoOracle(dic)
    ora.set_params(rate, True)
    got, want = det.detect(img), ora.detect(img)
    assert np.array_equal(got["id"], want["id"]) and np.allclose(got["corners"], want["corners"], atol=1e-3)
    tau = {"ARUCO_MIP_36h12": 12, "TAG36h11": 11, "ARUCO_MIP_25h7": 7}[dic]
    maxc = int(np.float32(tau) * np.float32(rate))
    expect = sorted(i for i, f in zip(ids, flips) if f == 0 or f < maxc)
    if 2 * (maxc - 1) < tau:    # correction radius below half the dictionary distance: the accepted code is the damaged marker's own
        assert sorted(got["id"].tolist()) == expect, (got["id"], expect)
    else:                       # beyond it the FIRST close entry in code order wins, not the nearest (as in the reference)
        assert len(got) >= len(expect) - 1


def test_corner_none_modes_and_contours(orbfe, oracle):
    """setCornerRefinementMethod(CORNER_NONE) returns the rotated approxPolyDP corners; aruco::Marker::contourPoints are the border
    the rectangle came from; parameters outside their range are refused loudly (the other modes: tests/test_aruco_modes_gpu.py)."""
    img, truth = synth.scene(480, 640, 2, "ARUCO", 4)
    det = orbfe.MarkerDetector("ARUCO")
    ora = oracle.ArucoOracle("ARUCO")
    lines = det.detect(img)
    det.setCornerRefinementMethod(det.CORNER_NONE); ora.set_params(0.0, False)
    got, want = det.detect(img), ora.detect(img)
    assert np.array_equal(got["id"], want["id"]) and np.array_equal(got["corners"], want["corners"])
    assert np.array_equal(got["corners"], np.rint(got["corners"])) and not np.array_equal(got["corners"], lines["corners"])
    for i in range(len(got)):
        c = det.contour(i)
        assert len(c) > 70                                     # only borders longer than 70 points become candidates
        # every corner of the unrefined marker is a point of its contour, and the contour is a closed 8-connected chain
        for k in range(4):
            assert np.any(np.all(c == got["corners"][i, k].astype(np.int32), axis=1))
        step = np.abs(np.diff(np.vstack([c, c[:1]]), axis=0)).max(axis=1)
        assert step.max() == 1 and step.min() == 1
    allc = det.contours(len(got))                              # the one-round-trip variant the MarkerDetector shim uses
    assert len(allc) == len(got) and all(np.array_equal(a, det.contour(i)) for i, a in enumerate(allc))
    assert det.contours(0) == []
    with pytest.raises(orbfe.OrbfeError):
        det.contour(len(got))
    with pytest.raises(orbfe.OrbfeError):
        det.contours(len(got) + 1)
    det.setCornerRefinementMethod(det.CORNER_LINES)
    assert np.array_equal(det.detect(img)["corners"], lines["corners"])
    det.setDetectionMode(det.DM_NORMAL)
    for bad in (lambda: det.setDetectionMode(7), lambda: det.setDetectionMode(det.DM_VIDEO_FAST, 1.1),
                lambda: det.setCornerRefinementMethod(5), lambda: det.setDictionary("ARUCO", 1.5)):
        with pytest.raises(orbfe.OrbfeError):
            bad()


@pytest.mark.parametrize("rows,cols", [(619, 1582), (700, 1400), (760, 1500), (655, 1599), (820, 2341), (603, 2172)])
def test_frames_at_the_lds_boundary_of_the_relay_kernels(orbfe, oracle, rows, cols):
    """Frame sizes whose bit image only just fits (or just does not fit) LDS next to the marker table: the kernel variant is chosen
    from the kernels' real static LDS (a constant once fell behind: 1582 x 619 and 1400 x 700 failed at launch); frames wider than
    2047 pixels have threshold windows of 17 and more (k_adaptive_threshold<15>).  Found by tools/stress_aruco.py / stress.py."""
    img, _ = synth.scene(rows, cols, rows + cols, "ARUCO", 4, side_range=(50, 150))
    det = orbfe.MarkerDetector("ARUCO")
    ora = oracle.ArucoOracle("ARUCO")
    got, want = det.detect(img), ora.detect(img)
    assert det.counts(0)["flags"] == 0
    assert np.array_equal(det.thresholded(0), ora.stage_image(0))
    assert np.array_equal(got["id"], want["id"]) and np.allclose(got["corners"], want["corners"], atol=1e-3)
    assert len(want) > 0


@pytest.mark.parametrize("mode", ["0", "1"])
def test_small_border_phase_inside_and_outside_the_relay_kernel(mode):
    """The borders between grid lines are followed inside the relay kernel (phase (c): what a full batch of LDS-resident frames
    runs) or by k_contours_small as its own launch (frames whose bit image lives in HBM, and batches of up to 32 frames, where it
    shortens the call).  ORBFE_ARUCO_SMALL_SEPARATE=0 / 1 forces one or the other for every batch size: the contour tests must
    pass both ways, so they are run in a process with the switch set."""
    import os, subprocess, sys
    env = dict(os.environ, ORBFE_ARUCO_SMALL_SEPARATE=mode)
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-k",
                        "structured_binary or relay_and_legacy or dense_frame or detect_matches_oracle or lds_boundary or tail_kernel"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("mode,tile_w,tpw,banded,band_rows", [("0", "0", "0", "0", "0"), ("1", "0", "0", "0", "0"), ("1", "64", "4", "0", "0"),
                                                              ("1", "480", "1", "0", "0"), ("1", "0", "0", "1", "0"), ("1", "96", "0", "1", "1"),
                                                              ("1", "0", "0", "1", "3")])
def test_tiled_and_one_workgroup_contour_paths(mode, tile_w, tpw, banded, band_rows):
    """The contour stage has two formulations of the same relay segments: one workgroup per frame out of one LDS image
    (k_contours_relay*: what full batches of frames up to 1280 x 720 run) and tiles of whole grid cells, a wave each
    (aruco_tiles.hip: 1920 x 1080 and batches of up to 32 frames by default).  ORBFE_ARUCO_TILED=0 / 1 forces one or the other for
    every batch; narrow tiles with four tiles per wave (the waves are persistent and overlap their tiles) and the widest tiles are run
    too, and the walks by bands of cell rows, a workgroup each (k_ct_band: what full batches of the tiled path run), with the default
    band height, with one cell row per band (every relay row is a band boundary) and with three.  The contour, detector-mode and
    full-HD tests must pass every way."""
    import os, subprocess, sys
    env = dict(os.environ, ORBFE_ARUCO_TILED=mode, ORBFE_ARUCO_TILE_W=tile_w, ORBFE_ARUCO_TPW=tpw, ORBFE_ARUCO_BANDED=banded,
               ORBFE_ARUCO_BAND_ROWS=band_rows)
    here = os.path.abspath(__file__)
    contour_tests = "structured_binary or relay_and_legacy or dense_frame or detect_matches_oracle or lds_boundary or tail_kernel or full_hd"
    plain = tile_w == "0" and band_rows == "0"          # the detector-mode sequences with the default geometry of each formulation only
    files = [here] + ([os.path.join(os.path.dirname(here), "test_aruco_modes_gpu.py")] if plain else [])
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-q", "-x", "-k", contour_tests + (" or sequence or enclosed or min_marker" if plain else "")],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("rows,cols,dict_name", [(480, 640, "ARUCO"), (720, 1280, "ARUCO_MIP_25h7"), (1080, 1920, "ARUCO")])
def test_tiled_path_completes_on_stream_frames(orbfe, rows, cols, dict_name):
    """A frame that exceeds a capacity of the tiled contour path is done again on the one-workgroup kernels -- silently, with the same
    answer.  Ordinary frames must not take that road (a walk bug that flags frames would hide behind it): forty frames (bands of cell
    rows), eight (a wave per tile) and one (one-row bands) with the tiled path forced, no retries, and the markers, border counts and
    candidate rectangles of the default path (one-workgroup kernels for the 640 x 480 batch of forty)."""
    nf = 40 if rows < 1080 else 34
    imgs = synth.stream(rows, cols, nf, 4242, dict_name, n_markers=4)
    ref = orbfe.MarkerDetector(dict_name)
    ref.set_speck_passes(False)          # (the start-candidate counts below are those of the thresholded image as it is)
    want = ref.detect_batch(imgs)
    wkeys = [(_rects_key(ref, f), ref.counts(f)) for f in range(nf)]
    assert ref.contour_retries() == 0
    dflt = orbfe.MarkerDetector(dict_name)   # the shipped default: the speck passes run in front of the one-workgroup kernels of a full batch
    got_dflt = dflt.detect_batch(imgs)
    for f in range(nf):
        assert np.array_equal(got_dflt[f], want[f]) and (dflt.counts(f)["nkept"], dflt.counts(f)["nrect"]) == (wkeys[f][1]["nkept"], wkeys[f][1]["nrect"])
    det = orbfe.MarkerDetector(dict_name)
    det.set_tiled_contours(True)
    for lo, hi in ((0, nf), (3, 11), (5, 6)):
        got = det.detect_batch(imgs[lo:hi])
        for f in range(lo, hi):
            (ca, la), cnt = wkeys[f]
            cb, lb = _rects_key(det, f - lo)
            c2 = det.counts(f - lo)
            assert np.array_equal(got[f - lo], want[f]) and np.array_equal(ca, cb) and np.array_equal(la, lb)
            assert (cnt["nkept"], cnt["nrect"], cnt["ncand"]) == (c2["nkept"], c2["nrect"], c2["ncand"])
    assert det.contour_retries() == 0
    assert sum(len(w) for w in want) > 0


def test_detector_paired_with_an_extractor(orbfe, oracle):
    """orbfe_extractor_pair_detector: the extractor's one-frame call starts the paired detector on the image it uploads; the detector's
    call takes that work if it is handed the same image and runs normally otherwise.  Same results in every case: same image (with
    and without poses, with the camera of the previous call and with another one), another image, detect without a preceding extract,
    a batch call in between, unpairing."""
    K = np.array([517.3, 516.5, 318.6, 255.3], np.float32); D = np.array([0.26, -0.95, -0.005, 0.003, 1.16], np.float32)
    K2 = K * np.float32(1.01)
    imgs = synth.stream(480, 640, 5, 77, "ARUCO", n_markers=4)
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    det = orbfe.MarkerDetector("ARUCO")
    ref = orbfe.MarkerDetector("ARUCO")                                  # never paired
    ora = oracle.ArucoOracle("ARUCO")
    want = [ref.detect(im, (K, D, (640, 480)), 0.187) for im in imgs]
    want2 = ref.detect(imgs[1], (K2, D, (640, 480)), 0.187)
    okps = [oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(im) for im in imgs[:2]]
    ex.pair_detector(det)
    same = lambda a, b: np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for rep in range(2):                                                 # second round: the speculation knows the camera
        for i, im in enumerate(imgs):
            k, d = ex(im)
            if i < 2:
                assert np.array_equal(k, okps[i][0]) and np.array_equal(d, okps[i][1])
            assert same(det.detect(im, (K, D, (640, 480)), 0.187), want[i]), (rep, i)
    assert np.array_equal(det.detect(imgs[0])["id"], ora.detect(imgs[0])["id"])
    ex(imgs[1]); assert same(det.detect(imgs[1], (K2, D, (640, 480)), 0.187), want2)         # same image, another camera
    ex(imgs[2]); assert same(det.detect(imgs[3], (K, D, (640, 480)), 0.187), want[3])         # another image than the extractor saw
    assert same(det.detect(imgs[4], (K, D, (640, 480)), 0.187), want[4])                      # no extract in between
    ex(imgs[0]); b = det.detect_batch(imgs[:3])                                               # a batch while work is pending
    assert all(np.array_equal(b[f]["id"], want[f][0]["id"]) for f in range(3))
    ex(imgs[1]); ex(imgs[2]); assert same(det.detect(imgs[2], (K, D, (640, 480)), 0.187), want[2])   # two extracts in a row
    c = det.contours(len(want[2][0]))
    assert len(c) == len(want[2][0]) and all(len(x) > 70 for x in c)
    # a setter between the extractor's call and the detector's ends the work started with the old parameters: the
    # detector's call must answer with the new dictionary / corner method / error-correction rate, not with what was speculated
    other = orbfe.MarkerDetector("ARUCO_MIP_36h12")
    mip = synth.stream(480, 640, 2, 78, "ARUCO_MIP_36h12", n_markers=4)[1]
    want_mip = other.detect(mip, (K, D, (640, 480)), 0.187)
    assert len(want_mip[0]) > 0
    ex(mip); det.setDictionary("ARUCO_MIP_36h12"); assert same(det.detect(mip, (K, D, (640, 480)), 0.187), want_mip)
    ex(imgs[1]); det.setDictionary("ARUCO"); assert same(det.detect(imgs[1], (K, D, (640, 480)), 0.187), want[1])
    ref.setCornerRefinementMethod(2); want_none = ref.detect(imgs[2], (K, D, (640, 480)), 0.187); ref.setCornerRefinementMethod(1)
    ex(imgs[2]); det.setCornerRefinementMethod(2); assert same(det.detect(imgs[2], (K, D, (640, 480)), 0.187), want_none)
    det.setCornerRefinementMethod(1)
    ex(imgs[2]); assert same(det.detect(imgs[2], (K, D, (640, 480)), 0.187), want[2])
    ex.pair_detector(None)
    ex(imgs[0]); assert same(det.detect(imgs[0], (K, D, (640, 480)), 0.187), want[0])


# ---- the speck passes between threshold and contours (csrc/aruco_kernels.hip k_speck_clean; aruco_trace.hpp "FEWER WALKS")
def _proto_speck_clean(tmp_path_factory):
    import ctypes as C
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path_factory.mktemp("proto") / "libproto.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(root, "tests", "proto_contours.cpp")])
    P = C.CDLL(so)
    P.proto_speck_clean.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    P.proto_speck_clean.restype = None

    def clean(b):
        b = np.ascontiguousarray(b, np.uint8)
        out = np.zeros_like(b)
        P.proto_speck_clean(b.ctypes.data, b.shape[1], b.shape[0], out.ctypes.data)
        return out * 255
    return clean


@pytest.mark.parametrize("rows,cols", [(480, 640), (427, 641), (97, 129), (720, 1280), (1080, 1920), (48, 64), (49, 95)])
def test_speck_passes_equal_the_cpu_twin(orbfe, tmp_path_factory, rows, cols):
    """The bit image the contour kernels read = the CPU twin's passes (tests/proto_contours.cpp, itself checked against the pixel-wise
    definition and the oracle in test_contour_logic_cpu.py) on the thresholded image, for frame sizes whose band / word geometry differs
    (rows not a multiple of the band height, widths of 32 k - 1 / + 1 pixels, one band), on a batch (every frame)."""
    clean = _proto_speck_clean(tmp_path_factory)
    n = 3 if rows * cols <= 1280 * 720 else 1
    imgs = synth.stream(rows, cols, n, 77, "ARUCO", n_markers=2 if rows >= 200 else 0)
    det = orbfe.MarkerDetector("ARUCO")
    det.set_speck_passes(True)
    det.detect_batch(imgs)
    removed = 0
    for f in range(n):
        th, got = det.thresholded(f), det.contour_image(f)
        assert np.array_equal(got, clean(th)), (rows, cols, f)
        removed += int(np.count_nonzero(th) - np.count_nonzero(got))
    assert removed > 0
    det.set_speck_passes(False)
    det.detect_batch(imgs)
    assert np.array_equal(det.contour_image(0), det.thresholded(0))


def test_speck_passes_change_no_result(orbfe):
    """Rectangle candidates (order, corners, contour lengths), kept-border counts and markers with and without the passes, on every
    contour path; with them far fewer start candidates are walked."""
    imgs = synth.stream(480, 640, 6, 4321, "ARUCO", n_markers=4)
    ref = orbfe.MarkerDetector("ARUCO")
    want = ref.detect_batch(imgs)
    wkeys = [(_rects_key(ref, f), ref.counts(f)) for f in range(len(imgs))]
    for mode in ("default", "tiled", "legacy"):
        det = orbfe.MarkerDetector("ARUCO")
        det.set_speck_passes(True)
        if mode == "tiled":
            det.set_tiled_contours(True)
        if mode == "legacy":
            det.force_legacy_contours(True)
        got = det.detect_batch(imgs)
        for f in range(len(imgs)):
            (ca, la), cnt = wkeys[f]
            cb, lb = _rects_key(det, f)
            c2 = det.counts(f)
            assert np.array_equal(got[f], want[f]) and np.array_equal(ca, cb) and np.array_equal(la, lb), (mode, f)
            assert (cnt["nkept"], cnt["nrect"]) == (c2["nkept"], c2["nrect"]) and c2["flags"] == 0
            assert c2["ncand"] * 2 < cnt["ncand"], (mode, f, c2["ncand"], cnt["ncand"])     # (both counts are after the run tests)
    assert sum(len(w) for w in want) > 0


def test_speck_passes_inside_the_relay_kernels(orbfe):
    """Full batches of frames whose bit image fits LDS go to the one-workgroup relay kernels, which can run the speck passes on the image
    they hold (speck_pass_frame; ORBFE_ARUCO_SPECKS=2).  Same function of the image as the launch of its own: the same number of start
    candidates per frame as with k_speck_clean in front of the same kernels, fewer than half of what they walk without the passes
    (clean frames; fewer on noisy ones),
    and rectangles, kept borders and markers are those of the run without."""
    imgs = synth.stream(480, 640, 40, 2468, "ARUCO", n_markers=4)
    rng = np.random.default_rng(2468)
    for f in range(1, 40, 2):      # every other frame with sensor noise: specks everywhere, also touching the markers' borders
        imgs[f] = np.clip(imgs[f].astype(np.int32) + rng.integers(-12, 13, imgs[f].shape), 0, 255).astype(np.uint8)
    runs = {}
    for mode in ("inside", "launch", "none"):
        det = orbfe.MarkerDetector("ARUCO")
        det.set_speck_passes_in_kernel(mode == "inside")
        det.set_speck_passes(mode == "launch")
        out = det.detect_batch(imgs)
        runs[mode] = (out, [(_rects_key(det, f), det.counts(f)) for f in range(len(imgs))])
        assert det.contour_retries() == 0
    for f in range(len(imgs)):
        (c0, l0), n0 = runs["none"][1][f]
        for mode in ("inside", "launch"):
            (c1, l1), n1 = runs[mode][1][f]
            assert np.array_equal(runs[mode][0][f], runs["none"][0][f]) and np.array_equal(c0, c1) and np.array_equal(l0, l1), (mode, f)
            assert (n0["nkept"], n0["nrect"]) == (n1["nkept"], n1["nrect"]) and n1["flags"] == 0
        assert runs["inside"][1][f][1]["ncand"] == runs["launch"][1][f][1]["ncand"], f
        assert runs["inside"][1][f][1]["ncand"] * (2 if f % 2 == 0 else 1) < n0["ncand"], f     # (noise makes blobs larger than a window too)
    assert sum(len(w) for w in runs["none"][0]) > 0


def test_dense_noise_frame_in_a_small_batch(orbfe, oracle):
    """A frame with +-40 grey levels of noise in a three-frame call: the tiled path's segment lists overflow, the call is redone on the
    one-workgroup relay kernels (which get through it) and the markers of all three frames equal the oracle's.  Forced onto the
    single-walker kernel in big-frame mode the same frame exceeds the per-lane arenas: a capacity error, loudly -- until round 5 a
    kept-border slot that was counted but never written sent the tail to an address made of stale LDS (GPU memory fault)."""
    imgs = synth.stream(480, 640, 3, 77, "ARUCO", n_markers=2)
    rng = np.random.default_rng(480)
    imgs[-1] = np.clip(imgs[-1].astype(np.int32) + rng.integers(-40, 40, imgs[-1].shape), 0, 255).astype(np.uint8)
    det = orbfe.MarkerDetector("ARUCO")
    got = det.detect_batch(imgs)
    assert det.contour_retries() >= 1
    for f in range(3):
        want = oracle.ArucoOracle("ARUCO").detect(imgs[f])
        assert np.array_equal(got[f]["id"], want["id"]) and np.allclose(got[f]["corners"], want["corners"], atol=1e-3), f
    big = orbfe.MarkerDetector("ARUCO")
    big.L.orbfe_aruco_set_big_frames(big.h, 1)
    try:
        out = big.detect_batch(imgs)
        assert all(np.array_equal(out[f]["id"], got[f]["id"]) for f in range(3))
    except orbfe.OrbfeError as e:
        assert "capacity" in str(e)


@pytest.mark.parametrize("rows,cols", [(480, 640), (427, 641), (720, 1280), (1080, 1920), (96, 130), (64, 64)])
def test_threshold_kernels_agree_and_the_fused_pyramid_is_the_pyramid(orbfe, oracle, rows, cols):
    """k_threshold_pyr (round 6: packed 16-bit vertical pass, ballots through v_writelane, the /2 pyramid levels a 64 x 64 tile holds
    written by the same launch) against k_adaptive_threshold_t + k_half_area4: the bit image, every pyramid level and the markers are
    the same, on frame sizes with partial tiles, odd widths, inexact deeper levels and all four window sizes' neighbours."""
    n = 2 if rows * cols <= 1280 * 720 else 1
    imgs = synth.stream(rows, cols, n, 99, "ARUCO", n_markers=3 if rows >= 200 else 0)
    a, b, c = orbfe.MarkerDetector("ARUCO"), orbfe.MarkerDetector("ARUCO"), orbfe.MarkerDetector("ARUCO")
    a.set_threshold_on_matrix_cores(False)                                        # a: k_threshold_pyr
    b.set_threshold_on_matrix_cores(False); b.set_threshold_pyramid_kernel(False)  # b: k_adaptive_threshold_t + k_half_area4 a level
    b.set_half_pyramid_kernel(False)                                              #    (c's pyramid: k_half_pyr, the leading exact levels in one launch)
    c.set_threshold_on_matrix_cores(True)                                         # c: k_threshold_mfma (a batch's default; calls of fewer than 8 frames
    ma, mb, mc = a.detect_batch(imgs), b.detect_batch(imgs), c.detect_batch(imgs)  #    take k_threshold_pyr unless told otherwise) + k_half_area4
    for f in range(n):
        assert np.array_equal(a.thresholded(f), b.thresholded(f)), (rows, cols, f)
        assert np.array_equal(c.thresholded(f), b.thresholded(f)), (rows, cols, f)
        assert np.array_equal(mc[f], mb[f])
        win = max(3, int(15 * float(cols) / 1920.)) | 1
        assert np.array_equal(a.thresholded(f), oracle.adaptive_threshold(imgs[f], win, 7)), (rows, cols, f)   # (window: markerdetector_impl.cpp:3765-3809)
        lvl = 1
        while True:
            la, lb, lc = a.pyramid_level(lvl, f), b.pyramid_level(lvl, f), c.pyramid_level(lvl, f)
            if la is None or lb is None:
                assert la is None and lb is None and lc is None
                break
            assert np.array_equal(la, lb) and np.array_equal(lc, lb), (rows, cols, f, lvl)
            lvl += 1
        assert np.array_equal(ma[f], mb[f])
