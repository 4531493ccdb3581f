"""Independent second opinions on the oracle's from-memory OpenCV primitives, with what the image offers:
torch's bilinear resize, scipy's box filter and connected-component labelling, a by-definition numpy Otsu.  These are NOT pins
(no OpenCV exists in this image): they bound how far a restated primitive could be from its textbook definition -- resize within
1 grey level of float bilinear interpolation, the adaptive threshold's box mean exact after rounding, one outer border per
8-connected foreground component and one hole border per hole, Otsu equal to the between-class-variance definition."""
import numpy as np
import pytest

from orb_slam2_aruco_amd import synth


def test_resize_is_bilinear_within_one_level(oracle):
    torch = pytest.importorskip("torch")
    img, _ = synth.scene(240, 320, 3, n_markers=1, side_range=(40, 60))
    for dw, dh in ((267, 200), (222, 167), (160, 120)):          # the 1/1.2 steps of the pyramid and a plain /2
        got = oracle.resize_linear_u8(img, dw, dh).astype(np.float64)
        t = torch.from_numpy(img.astype(np.float64))[None, None]
        ref = torch.nn.functional.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
        err = np.abs(got - ref)
        assert err.max() <= 1.0, (dw, dh, err.max())               # 11-bit fixed-point coefficients, two roundings
        assert err.mean() < 0.3


def test_adaptive_threshold_box_mean_matches_scipy(oracle):
    ndi = pytest.importorskip("scipy.ndimage")
    img, _ = synth.scene(200, 260, 5, n_markers=2, side_range=(40, 60))
    for win in (3, 5, 7, 11):
        got = oracle.adaptive_threshold(img, win, 7)
        # ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV: 255 where src - mean <= -C; the mean is boxFilter with BORDER_REPLICATE, rounded
        s = ndi.uniform_filter(img.astype(np.float64), size=win, mode="nearest") * win * win
        mean = np.rint(np.rint(s) / (win * win))                  # exact integer sums, then round half to even
        want = np.where(img.astype(np.float64) - mean <= -7, 255, 0).astype(np.uint8)
        assert np.array_equal(got, want), win


def test_contour_counts_match_component_labelling(oracle):
    ndi = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(4)
    for trial in range(3):
        img, _ = synth.scene(160, 200, 20 + trial, n_markers=2, side_range=(30, 50))
        b = oracle.adaptive_threshold(img, 5, 7)
        b[0, :] = b[-1, :] = 0; b[:, 0] = b[:, -1] = 0           # keep every component off the frame
        cs = oracle.find_contours(b)
        # RETR_LIST returns one outer border per 8-connected foreground component and one hole border per hole
        # (a hole = a 4-connected background component that does not touch the frame)
        fg, nfg = ndi.label(b > 0, structure=np.ones((3, 3), int))
        bg, nbg = ndi.label(b == 0)
        holes = nbg - 1
        assert len(cs) == nfg + holes, (trial, len(cs), nfg, holes)
        # every contour point is a foreground pixel, and every component's pixel set is touched by exactly its own borders
        pts = np.concatenate(cs)
        assert np.all(b[pts[:, 1], pts[:, 0]] > 0)
        total = sum(len(c) for c in cs)
        assert total >= nfg + holes


def test_otsu_is_the_between_class_variance_maximum(oracle):
    rng = np.random.default_rng(9)
    for trial in range(20):
        n = 35 * 35
        a = np.clip(rng.normal(rng.uniform(30, 110), rng.uniform(5, 30), n // 2), 0, 255)
        b = np.clip(rng.normal(rng.uniform(140, 230), rng.uniform(5, 30), n - n // 2), 0, 255)
        v = np.concatenate([a, b]).astype(np.uint8)
        h = np.bincount(v, minlength=256).astype(np.float64)
        p = h / h.sum()
        i = np.arange(256)
        q1 = np.cumsum(p); q2 = 1.0 - q1
        m1 = np.cumsum(i * p)
        mu = m1[-1]
        with np.errstate(divide="ignore", invalid="ignore"):
            sigma = np.where((q1 > 1e-7) & (q2 > 1e-7), (mu * q1 - m1) ** 2 / (q1 * q2), 0.0)
        # the first maximiser (ties within floating noise are taken by the smallest threshold, as the running loop does)
        best = sigma.max()
        want = int(np.nonzero(sigma >= best * (1 - 1e-12))[0][0])
        assert oracle.otsu_threshold(v) == want, trial


def test_gaussian_tap_variants(oracle):
    assert oracle.gaussian7_taps(0).tolist() == [18, 34, 49, 55, 49, 34, 18]       # sum 257: 2.4 / 3.2 / early 3.4
    assert oracle.gaussian7_taps(1).tolist() == [18, 34, 48, 56, 48, 34, 18]       # sum 256: late 3.4.x / 4.x
    # the choice changes a small share of descriptor bits and no keypoint
    img, _ = synth.scene(240, 320, 7, n_markers=2, side_range=(40, 70))
    o = oracle.OrbOracle(500, 1.2, 4, 20, 7)
    k0, d0 = o.extract(img)
    o.set_gaussian_taps(1)
    k1, d1 = o.extract(img)
    assert np.array_equal(k0, k1)
    flipped = np.unpackbits(d0 ^ d1).mean()
    assert 0 < flipped < 0.02, flipped


# ---------------------------------------------------------------------------------------------------------------------------
# Round 3: second opinions for the primitives that had none -- the FAST segment test and corner score, approxPolyDP,
# warpPerspective's fixed-point interpolation, fastAtan2.  Tolerance checks against textbook definitions, not pins.
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _segment_test(img, t):
    """Rosten's definition, vectorised: pixel p is a corner at threshold t iff 9 CONTIGUOUS pixels of the 16-ring of radius 3
    are all brighter than I(p) + t or all darker than I(p) - t.  Boolean planes and circular runs; no minimum / maximum network."""
    I = img.astype(np.int32)
    h, w = I.shape
    c = I[3:h - 3, 3:w - 3]
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])
    out = np.zeros(c.shape, bool)
    for plane in (ring > c + t, ring < c - t):
        p2 = np.concatenate([plane, plane[:8]])
        run = np.ones((16,) + c.shape, bool)
        for j in range(9):
            run &= p2[j:j + 16]
        out |= run.any(0)
    res = np.zeros((h, w), bool)
    res[3:h - 3, 3:w - 3] = out
    return res


def test_fast_score_is_the_largest_threshold_of_the_segment_test(oracle):
    """cornerScore (the oracle's max-over-arcs-of-min network, as OpenCV's cornerScore<16>) against the definition: the score of a
    pixel is the LARGEST t at which the 9-of-16 segment test still calls it a corner.  All 255 thresholds on small-alphabet images
    (ties, equal neighbours, saturated values) and a textured scene; then the detector with its 3 x 3 strict maximum."""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(11)
    imgs = [rng.choice(np.array([0, 19, 20, 21, 40, 41, 128, 255], np.uint8), size=(40, 48)),
            rng.integers(0, 256, (40, 48)).astype(np.uint8),
            synth.scene(96, 128, 4, n_markers=1, side_range=(30, 40))[0]]
    for img in imgs:
        img = np.ascontiguousarray(img)
        h, w = img.shape
        score = np.zeros((h, w), np.int32)
        L.oracle_fast_score_map(img.ctypes.data_as(C.c_void_p), w, h, score.ctypes.data_as(C.c_void_p))
        by_def = np.full((h, w), -1, np.int32)            # largest t with a positive segment test; -1: never a corner
        for t in range(0, 255):
            by_def[_segment_test(img, t)] = t
        inner = (slice(3, h - 3), slice(3, w - 3))
        s_in, d_in = score[inner], by_def[inner]
        assert np.array_equal(np.maximum(s_in, -1), d_in), "cornerScore differs from the segment-test definition"
        # cv::FAST(t, nonmaxSuppression): corners at t whose score exceeds all 8 neighbours' scores (non-corners count as 0)
        for t in (7, 20):
            sc = np.where(by_def >= t, by_def, 0)
            keep = np.zeros((h, w), bool)
            for y in range(3, h - 3):
                for x in range(3, w - 3):
                    if sc[y, x] > 0:
                        nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
                        nb[1, 1] = -1
                        keep[y, x] = sc[y, x] > nb.max()
            want = [(x, y, int(by_def[y, x])) for y in range(h) for x in range(w) if keep[y, x]]     # raster order
            kps = np.zeros(4096, oracle.KP_DTYPE)
            n = L.oracle_fast_detect(img.ctypes.data_as(C.c_void_p), w, h, t, kps.ctypes.data_as(C.c_void_p), len(kps))
            got = [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps[:n]]
            assert got == want, t


def _point_segment_distance(p, a, b):
    p, a, b = (np.asarray(v, np.float64) for v in (p, a, b))
    ab = b - a
    den = float(ab @ ab)
    if den == 0:
        return float(np.hypot(*(p - a)))
    return abs(float(ab[0] * (p - a)[1] - ab[1] * (p - a)[0])) / np.sqrt(den)      # distance to the LINE, as Douglas-Peucker uses


def test_approx_poly_is_a_douglas_peucker_polygon(oracle):
    """approxPolyDP(closed, eps = 0.05 * length) on marker borders: (1) the output is a cyclic subsequence of the contour; (2) the
    Douglas-Peucker guarantee -- every contour point between two consecutive output vertices lies within eps of the line through
    them; (3) a plain recursive Douglas-Peucker, started from the two mutually farthest points, finds the same number of vertices
    on rendered marker quadrilaterals and its vertices lie within 2 pixels of OpenCV-style ones."""
    import ctypes as C
    L = oracle.lib()
    L.oracle_approx_poly.restype = C.c_int
    L.oracle_approx_poly.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int]

    def dp(points, i, j, eps, keep):
        if j <= i + 1:
            return
        d = [_point_segment_distance(points[k], points[i], points[j]) for k in range(i + 1, j)]
        k = int(np.argmax(d))
        if d[k] > eps:
            keep.add(i + 1 + k)
            dp(points, i, i + 1 + k, eps, keep); dp(points, i + 1 + k, j, eps, keep)

    checked = 0
    for seed in (2, 3, 5):
        img, _ = synth.scene(240, 320, seed, "ARUCO", 3, side_range=(40, 80))
        b = oracle.adaptive_threshold(img, 3, 7)
        for c in oracle.find_contours(b):
            n = len(c)
            if n <= 70 or len(np.unique(c, axis=0)) != n:      # borders that pass a pixel twice have no unique vertex positions
                continue
            eps = 0.05 * n
            out = np.zeros((256, 2), np.int32)
            m = L.oracle_approx_poly(np.ascontiguousarray(c, np.int32).ctypes.data_as(C.c_void_p), n, eps, out.ctypes.data_as(C.c_void_p), 256)
            v = out[:m]
            # (1) cyclic subsequence: the vertices' (first) positions in the contour increase once round the cycle
            pos = [int(np.nonzero((c == p).all(1))[0][0]) for p in v]
            rot = int(np.argmin(pos))
            cyc = pos[rot:] + pos[:rot]
            assert cyc == sorted(cyc), (pos,)
            # (2) the guarantee
            for a in range(m):
                i, j = cyc[a], cyc[(a + 1) % m] if a + 1 < m else cyc[0] + n
                for k in range(i + 1, j):
                    assert _point_segment_distance(c[k % n], c[i % n], c[j % n]) <= eps + 1e-9
            # (3) textbook Douglas-Peucker on quadrilateral-looking borders
            if m == 4:
                D = ((c[:, None, :] - c[None, :, :]) ** 2).sum(2)
                i0, j0 = np.unravel_index(int(np.argmax(D)), D.shape)
                i0, j0 = min(i0, j0), max(i0, j0)
                keep = {0, j0 - i0}
                cyc_pts = np.concatenate([c[i0:], c[:i0], c[i0:i0 + 1]])
                dp(cyc_pts, 0, j0 - i0, eps, keep); dp(cyc_pts, j0 - i0, n, eps, keep)
                tv = cyc_pts[sorted(keep - {n})]
                assert len(tv) == 4, len(tv)
                for p in v:
                    assert np.min(np.hypot(*(tv - p).T)) <= 2.0
                checked += 1
    assert checked >= 3


def test_warp_perspective_is_bilinear_sampling_within_rounding(oracle):
    """warpPerspective(INTER_LINEAR) of the oracle (1/32-pixel coordinates, 5-bit weights) against float bilinear sampling through the
    same homography (torch.nn.functional.grid_sample, align_corners=True == pixel-centre coordinates): the fixed-point rounding moves
    a sample by at most 1/64 pixel, i.e. by gradient / 64 grey levels, plus one level of final rounding."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(6)
    yy, xx = np.mgrid[0:200, 0:260].astype(np.float64)
    img = (127 + 60 * np.sin(xx / 17.0) * np.cos(yy / 23.0) + 40 * np.sin((xx + yy) / 31.0)).clip(0, 255).astype(np.uint8)   # smooth: |grad| < 8 / px
    for trial in range(4):
        c = np.array([60 + 100 * rng.random(), 50 + 80 * rng.random()])
        s = 25 + 30 * rng.random()
        ang = rng.uniform(0, 2 * np.pi)
        quad = np.array([c + s * np.array([np.cos(ang + k * np.pi / 2 + rng.uniform(-0.1, 0.1)), np.sin(ang + k * np.pi / 2 + rng.uniform(-0.1, 0.1))])
                         for k in range(4)], np.float32)
        S = 35
        got = oracle.warp35(img, quad, S).astype(np.float64)
        # homography dst -> src from the four correspondences (float64 DLT), as getPerspectiveTransform defines it
        dst = np.array([[0, 0], [S - 1, 0], [S - 1, S - 1], [0, S - 1]], np.float64)
        A, bvec = [], []
        for (X, Y), (x, y) in zip(dst, quad.astype(np.float64)):
            A.append([X, Y, 1, 0, 0, 0, -X * x, -Y * x]); bvec.append(x)
            A.append([0, 0, 0, X, Y, 1, -X * y, -Y * y]); bvec.append(y)
        hvec = np.linalg.solve(np.array(A), np.array(bvec))
        Hm = np.append(hvec, 1.0).reshape(3, 3)
        gy, gx = np.mgrid[0:S, 0:S].astype(np.float64)
        den = Hm[2, 0] * gx + Hm[2, 1] * gy + Hm[2, 2]
        sx = (Hm[0, 0] * gx + Hm[0, 1] * gy + Hm[0, 2]) / den
        sy = (Hm[1, 0] * gx + Hm[1, 1] * gy + Hm[1, 2]) / den
        grid = torch.from_numpy(np.stack([2 * sx / (img.shape[1] - 1) - 1, 2 * sy / (img.shape[0] - 1) - 1], -1))[None]
        ref = torch.nn.functional.grid_sample(torch.from_numpy(img.astype(np.float64))[None, None], grid, mode="bilinear",
                                              padding_mode="zeros", align_corners=True)[0, 0].numpy()
        inside = (sx > 1) & (sx < img.shape[1] - 2) & (sy > 1) & (sy < img.shape[0] - 2)
        err = np.abs(got - ref)[inside]
        assert inside.mean() > 0.9 and err.max() <= 2.0 and err.mean() < 0.5, (trial, err.max(), err.mean())


def test_fast_atan2_is_atan2_within_a_third_of_a_degree(oracle):
    """cv::fastAtan2 (the 7th-order odd polynomial on the octant-reduced ratio, degrees in [0, 360)) against numpy.arctan2."""
    import ctypes as C
    L = oracle.lib()
    rng = np.random.default_rng(12)
    xy = np.concatenate([rng.normal(0, 3000, (4000, 2)), rng.integers(-20, 21, (2000, 2)).astype(np.float64),
                         np.array([[1, 0], [0, 1], [-1, 0], [0, -1], [1, 1], [-1, 1], [-1, -1], [1, -1], [1e-3, 1], [1, 1e-3]], np.float64)])
    worst = 0.0
    for x, y in xy:
        if x == 0 and y == 0:
            continue
        got = float(L.oracle_fast_atan2(C.c_float(y), C.c_float(x)))
        want = np.degrees(np.arctan2(np.float32(y), np.float32(x))) % 360.0
        d = abs(got - want)
        d = min(d, 360.0 - d)
        worst = max(worst, d)
        assert 0.0 <= got < 360.0 + 1e-3
    assert worst <= 0.3, worst
    assert float(L.oracle_fast_atan2(C.c_float(0.0), C.c_float(0.0))) == 0.0
