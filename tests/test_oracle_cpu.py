"""CPU tests (no GPU): the oracle against the reference's source-embedded known answers and the
committed golden vectors."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- ctor tables (ORBextractor.cc:410-470)
def test_ctor_tables_match_reference_constants(oracle):
    t = oracle.OrbOracle(1000, 1.2, 8, 20, 7).tables()
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert t["per_level"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.OrbOracle(2000, 1.2, 8, 20, 7).tables()["per_level"].tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert oracle.OrbOracle(4000, 1.2, 12, 20, 7).tables()["per_level"].tolist() == \
        [751, 626, 521, 435, 362, 302, 251, 210, 175, 146, 121, 100]
    # float recurrence mvScaleFactor[i] = mvScaleFactor[i-1] * (double)1.2f, rounded to float each step
    s = np.float32(1.0)
    for i in range(1, 8):
        s = np.float32(np.float64(s) * np.float64(np.float32(1.2)))
        assert t["scale"][i] == s
    assert np.all(t["inv_scale"] == np.float32(1.0) / t["scale"])


def test_level_sizes_match_survey_appendix_e(oracle):
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    o.extract(np.zeros((480, 640), np.uint8))
    got = [o.level_image(l).shape[::-1] for l in range(8)]
    assert got == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]


# ---------------------------------------------------------------- numerics contract (include/orbfe_math.h)
def test_cv_round_half_to_even(oracle):
    L = oracle.lib()
    for v, want in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (3.4999, 3), (-2.5, -2), (1e6 + 0.5, 1000000)]:
        assert L.oracle_cv_round(v) == want


def test_fast_atan2_properties(oracle):
    L = oracle.lib()
    assert L.oracle_fast_atan2(0.0, 0.0) == 0.0
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = (float(v) for v in rng.integers(-3000000, 3000000, 2))
        a = L.oracle_fast_atan2(y, x)
        true = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - true)
        assert min(d, 360 - d) < 0.02            # OpenCV documents ~0.3 deg; the polynomial is much better
        assert 0.0 <= a <= 360.0


def test_sincos_is_correctly_rounded(oracle):
    """orbfe_sincosf == float32(round(sin/cos in double)) on a dense sweep of the angles the extractor can produce."""
    L = oracle.lib()
    s, c = C.c_float(), C.c_float()
    deg = np.arange(0, 360, 0.0137, dtype=np.float32)
    rad = (deg * np.float32(math.pi / 180.0)).astype(np.float32)
    bad = 0
    for r in rad[::3]:
        L.oracle_sincosf(float(r), C.byref(s), C.byref(c))
        if np.float32(math.sin(float(r))) != np.float32(s.value) or np.float32(math.cos(float(r))) != np.float32(c.value):
            bad += 1
    assert bad == 0


# ---------------------------------------------------------------- OpenCV primitives restated (App. B)
def test_gaussian_taps(oracle):
    taps = np.zeros(7, np.int32)
    oracle.lib().oracle_gaussian7_taps(taps.ctypes.data_as(C.c_void_p))
    assert taps.tolist() == [18, 34, 49, 55, 49, 34, 18]


def test_blur_constant_and_saturation(oracle):
    L = oracle.lib()
    for v in (0, 7, 128, 254, 255):
        src = np.full((20, 31), v, np.uint8)
        dst = np.zeros_like(src)
        L.oracle_gaussian_blur7(src.ctypes.data_as(C.c_void_p), 31, 20, dst.ctypes.data_as(C.c_void_p))
        want = min(255, (257 * 257 * v + 32768) >> 16)
        assert np.all(dst == want)


def test_resize_identity_and_constant(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (40, 60), dtype=np.uint8)
    dst = np.zeros_like(src)
    L.oracle_resize_linear_u8(src.ctypes.data_as(C.c_void_p), 60, 40, dst.ctypes.data_as(C.c_void_p), 60, 40)
    assert np.array_equal(src, dst)
    src = np.full((48, 64), 93, np.uint8)
    dst = np.zeros((40, 53), np.uint8)
    L.oracle_resize_linear_u8(src.ctypes.data_as(C.c_void_p), 64, 48, dst.ctypes.data_as(C.c_void_p), 53, 40)
    assert np.all(dst == 93)


def _ring():
    return [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
            (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_fast_score_is_largest_threshold_still_a_corner(oracle):
    """score(p) = max t such that p passes the 9-of-16 segment test at threshold t (definition of cornerScore)."""
    L = oracle.lib()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (24, 24), dtype=np.uint8)
    out = np.zeros((24, 24), np.int32)
    L.oracle_fast_score_map(img.ctypes.data_as(C.c_void_p), 24, 24, out.ctypes.data_as(C.c_void_p))

    def is_corner(y, x, t):
        v = int(img[y, x])
        r = [int(img[y + dy, x + dx]) for dx, dy in _ring()]
        for s in range(16):
            if all(r[(s + j) % 16] > v + t for j in range(9)) or all(r[(s + j) % 16] < v - t for j in range(9)):
                return True
        return False

    for y in range(3, 21, 2):
        for x in range(3, 21, 3):
            s = out[y, x]
            if s >= 0:
                assert is_corner(y, x, s) and not is_corner(y, x, s + 1)
            else:
                assert not is_corner(y, x, 0)


def test_quadtree_small_cases(oracle):
    kp = np.zeros(6, oracle.KP_DTYPE)
    kp["x"] = [10, 300, 20, 500, 310, 305]
    kp["y"] = [10, 20, 300, 400, 25, 22]
    kp["response"] = [5, 50, 7, 9, 50, 60]
    out = np.zeros(16, oracle.KP_DTYPE)
    L = oracle.lib()
    n = L.oracle_distribute(kp.ctypes.data_as(C.c_void_p), 6, 16, 624, 16, 464, 3, out.ctypes.data_as(C.c_void_p), 16)
    assert 3 <= n <= 6
    # N larger than the candidates: splitting stops as soon as one pass leaves the node count unchanged (:661), which
    # happens here while the three clustered points still share a node -> its best response (60) represents them
    n = L.oracle_distribute(kp.ctypes.data_as(C.c_void_p), 6, 16, 624, 16, 464, 100, out.ctypes.data_as(C.c_void_p), 16)
    assert n in (4, 5, 6) and 60 in out["response"][:n].tolist() and 5 in out["response"][:n].tolist()
    # well separated points: every point ends up alone
    kp["x"] = [10, 300, 20, 500, 100, 400]
    kp["y"] = [10, 20, 300, 400, 200, 100]
    n = L.oracle_distribute(kp.ctypes.data_as(C.c_void_p), 6, 16, 624, 16, 464, 100, out.ctypes.data_as(C.c_void_p), 16)
    assert n == 6 and sorted(out["response"][:6].tolist()) == [5, 7, 9, 50, 50, 60]


# ---------------------------------------------------------------- matching (App. D)
def test_descriptor_distance_is_popcount(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    for _ in range(500):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        assert L.oracle_descriptor_distance(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)) == want
    z = np.zeros(32, np.uint8)
    f = np.full(32, 255, np.uint8)
    assert L.oracle_descriptor_distance(z.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)) == 256


def test_three_maxima(oracle):
    L = oracle.lib()

    def tm(h):
        h = np.asarray(h, np.int32)
        out = np.zeros(3, np.int32)
        L.oracle_three_maxima(h.ctypes.data_as(C.c_void_p), len(h), out.ctypes.data_as(C.c_void_p))
        return out.tolist()

    h = [0] * 30
    h[3], h[7], h[20] = 100, 50, 20
    assert tm(h) == [3, 7, 20]
    h[20] = 9                      # third < 10 % of first
    assert tm(h) == [3, 7, -1]
    h[7] = 9
    assert tm(h) == [3, -1, -1]
    assert tm([0] * 30) == [-1, -1, -1]


def test_knn2_tie_rules(oracle):
    T = np.zeros((5, 32), np.uint8)
    T[1, 0] = 1; T[2, 0] = 1; T[3, 0] = 3
    Q = np.zeros((1, 32), np.uint8); Q[0, 0] = 1
    bi, bd, sd = oracle.knn2(Q, T, 256)
    assert (bi[0], bd[0], sd[0]) == (1, 0, 0)      # first of the two exact matches wins; the duplicate is the runner-up
    bi, bd, sd = oracle.knn2(Q, T[:1], 256)
    assert (bi[0], bd[0], sd[0]) == (0, 1, 256)


# ---------------------------------------------------------------- ArUco
def test_marker_render_decode_roundtrip_all_rotations(oracle):
    """getMarkerImage_id (dictionary.cpp:254-342) is the inverse of the decoder (dictionary_based.cpp:2372-2645)."""
    for dic, ids in (("ARUCO", [0, 1, 77, 500, 1022]), ("ARUCO_MIP_25h7", [0, 42, 99]), ("ARUCO_MIP_36h12", [0, 249]),
                     ("TAG16h5", [0, 29])):
        a = oracle.ArucoOracle(dic)
        nbits, _ = synth.dictionary_codes(dic)
        n = int(round(nbits ** 0.5)) + 2
        for i in ids:
            m = synth.render_marker(dic, i, 5, quiet=0)      # exactly the 5 px/bit patch the detector warps to
            assert m.shape == (5 * n, 5 * n)
            for k in range(4):
                got, rot = a.decode(np.ascontiguousarray(np.rot90(m, -k)))
                assert got == i, (dic, i, k, got)
                assert rot == (4 - k) % 4     # k clockwise quarter turns are undone by 4-k decoder rotations
        assert a.decode(np.zeros((5 * n, 5 * n), np.uint8))[0] == -1          # all-black: code 0 is rejected
        assert a.decode(np.full((5 * n, 5 * n), 255, np.uint8))[0] == -1      # white border cells: rejected


def test_adaptive_threshold_definition(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (30, 41), dtype=np.uint8)
    for win in (3, 5, 11):
        got = oracle.adaptive_threshold(img, win, 7)
        r = win // 2
        p = np.pad(img.astype(np.int64), r, mode="edge")
        s = sum(p[dy:dy + 30, dx:dx + 41] for dy in range(win) for dx in range(win))
        mean = np.floor(s / (win * win) + 0.5).astype(np.int64)     # no exact ties: win*win is odd
        want = np.where(img.astype(np.int64) - mean <= -7, 255, 0).astype(np.uint8)
        assert np.array_equal(got, want)


def test_find_contours_small_known_answers(oracle):
    b = np.zeros((8, 8), np.uint8)
    b[2:6, 2:6] = 255
    c = oracle.find_contours(b)
    assert len(c) == 1 and len(c[0]) == 12 and c[0][0].tolist() == [2, 2]
    assert c[0][1].tolist() == [2, 3]                       # outer borders run counter-clockwise in image coordinates
    b[3:5, 3:5] = 0
    c = oracle.find_contours(b)
    assert len(c) == 2
    assert c[0][0].tolist() == [2, 3] and len(c[0]) == 8    # hole border found later, returned first (reverse order)
    b = np.zeros((5, 5), np.uint8); b[2, 2] = 255
    c = oracle.find_contours(b)
    assert len(c) == 1 and c[0].tolist() == [[2, 2]]


# ---------------------------------------------------------------- golden vectors
def test_golden_orb(oracle):
    g = np.load(os.path.join(GOLD, "orb_240x320.npz"))
    nf, nl, ini, mn = g["params"]
    k, d = oracle.OrbOracle(int(nf), 1.2, int(nl), int(ini), int(mn)).extract(g["image"])
    assert np.array_equal(k, g["kps"]) and np.array_equal(d, g["desc"])
    g = np.load(os.path.join(GOLD, "orb_640x480_seed1.npz"))
    img, _ = synth.scene(480, 640, 1, "ARUCO", 4)
    assert int(img.astype(np.int64).sum()) == int(g["image_sum"][0]), "synthetic generator changed"
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    k, d = o.extract(img)
    assert np.array_equal(k, g["kps"]) and np.array_equal(d, g["desc"])
    assert [len(o.level_keypoints(l, 0)) for l in range(8)] == g["ncand"].tolist()


def test_golden_aruco(oracle):
    for name, (h, w, seed, dic, K) in {"aruco_640x480_seed1": (480, 640, 1, "ARUCO", 4),
                                       "aruco_540x960_seed6": (540, 960, 6, "ARUCO_MIP_36h12", 5)}.items():
        g = np.load(os.path.join(GOLD, name + ".npz"))
        img, truth = synth.scene(h, w, seed, dic, K)
        a = oracle.ArucoOracle(dic)
        m = a.detect(img)
        assert np.array_equal(m["id"], g["markers"]["id"])
        assert np.array_equal(m["corners"], g["markers"]["corners"])
        assert np.array_equal(a.candidates(0), g["rects"])
        assert m["id"].tolist() == g["truth_ids"].tolist()      # every planted marker is found, nothing else


def test_golden_match(oracle):
    g = np.load(os.path.join(GOLD, "match_stream1000.npz"))
    bi, bd, sd = oracle.knn2(g["d1"], g["d2"], 256)
    assert np.array_equal(bi, g["best_idx"]) and np.array_equal(bd, g["best_dist"]) and np.array_equal(sd, g["second_dist"])
    n, m12, prev = oracle.search_for_initialization(g["k1"], g["d1"], g["k2"], g["d2"], 640, 480, None, 100, 0.9, True)
    assert n == int(g["nmatches"][0]) and np.array_equal(m12, g["matches12"]) and np.array_equal(prev, g["prev"])
    assert n > 50


def _projection_golden(mod):
    g = np.load(os.path.join(GOLD, "match_stream1000.npz"))
    p = np.load(os.path.join(GOLD, "projection_stream1000.npz"))
    q = p["queries"].view(mod.WINDOW_QUERY_DTYPE).reshape(-1)
    return g, p, q


def test_golden_search_by_projection(oracle):
    g, p, q = _projection_golden(oracle)
    r0 = oracle.search_by_projection(g["k2"], g["d2"], 640, 480, q, g["d1"][p["sel"]], p["taken"], 0, 100, 0.8)
    for f in ("best_idx", "best_dist", "best_level", "second_dist", "second_level"):
        assert np.array_equal(r0[f], p[f]), f
    r1 = oracle.search_by_projection(g["k2"], g["d2"], 640, 480, q, g["d1"][p["sel"]], p["taken"], 1, 100, 0.8)
    assert r1["nmatches"] == int(p["nmatches"][0]) > 20
    assert np.array_equal(r1["match"], p["match"]) and np.array_equal(r1["taken"], p["taken_after"])


def test_libm_trig_sensitivity_is_small(oracle):
    """The reference calls libm cos/sin (ORBextractor.cc:112-113); the oracle uses correctly rounded values.
    Quantify what that choice can change: a handful of descriptor BITS per frame at most."""
    img, _ = synth.scene(240, 320, 7, "ARUCO", 2, side_range=(40, 70))
    o = oracle.OrbOracle(500, 1.2, 4, 20, 7)
    k1, d1 = o.extract(img)
    oracle.lib().oracle_orb_set_trig_libm(o.h, 1)
    k2, d2 = o.extract(img)
    assert np.array_equal(k1, k2)
    flipped = int(np.unpackbits(d1 ^ d2).sum())
    assert flipped <= max(8, d1.size * 8 // 20000)


def test_search_by_projection_oracle_against_brute_force(oracle):
    """The oracle's SearchByProjection loop, mode 0, equals a plain numpy restatement of window + octave + Hamming."""
    from orb_slam2_aruco_amd import synth
    rng = np.random.default_rng(3)
    img, _ = synth.scene(240, 320, 4, n_markers=2, side_range=(30, 60))
    kps, desc = oracle.OrbOracle(500, 1.2, 4, 20, 7).extract(img)
    nq = 60
    pick = rng.integers(0, len(kps), nq)
    q = np.zeros(nq, oracle.WINDOW_QUERY_DTYPE)
    q["x"] = kps["x"][pick] + 1.5; q["y"] = kps["y"][pick] - 0.5; q["r"] = 9.0
    q["min_level"] = kps["octave"][pick] - 1; q["max_level"] = kps["octave"][pick]
    qd = desc[pick] ^ np.uint8(1)
    got = oracle.search_by_projection(kps, desc, 320, 240, q, qd, None, 0)
    bits = np.unpackbits(desc, axis=1).astype(np.int32)
    for i in range(nq):
        ok = (np.abs(kps["x"] - q["x"][i]) < 9.0) & (np.abs(kps["y"] - q["y"][i]) < 9.0) & \
             (kps["octave"] >= q["min_level"][i]) & (kps["octave"] <= q["max_level"][i])
        d = np.abs(bits[ok] - np.unpackbits(qd[i]).astype(np.int32)).sum(1)
        assert got["best_dist"][i] == (d.min() if len(d) else 256)
        if len(d) > 1:
            assert got["second_dist"][i] == np.sort(d)[1]


# ---------------------------------------------------------------- Frame glue: undistortion (Frame.cc:357-451)
TUM1_K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
TUM1_DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


def _distort(xy, K, d):
    """Forward Brown model in float64 (the model cv::undistortPoints inverts)."""
    d = np.concatenate([d.astype(np.float64), np.zeros(12 - len(d))])
    fx, fy, cx, cy = K.astype(np.float64)
    x = (xy[:, 0].astype(np.float64) - cx) / fx; y = (xy[:, 1].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    cd = (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2) / (1 + ((d[7] * r2 + d[6]) * r2 + d[5]) * r2)
    xd = x * cd + 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x) + d[8] * r2 + d[9] * r2 * r2
    yd = y * cd + d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y + d[10] * r2 + d[11] * r2 * r2
    return np.stack([xd * fx + cx, yd * fy + cy], 1)


def test_undistort_points_inverts_the_brown_model(oracle):
    rng = np.random.default_rng(0)
    pts = rng.uniform([0, 0], [640, 480], (2000, 2)).astype(np.float32)
    un = oracle.undistort_points(pts, TUM1_K, TUM1_DIST)
    back = _distort(un, TUM1_K, TUM1_DIST)
    # five iterations converge to well under a pixel over the TUM1 image (OpenCV's fixed iteration count)
    assert np.abs(back - pts).max() < 0.05
    assert np.abs(un - pts).max() > 1.0                       # the distortion is not a no-op on this camera
    # no coefficients: normalise + re-project only
    same = oracle.undistort_points(pts, TUM1_K, np.zeros(0, np.float32))
    assert np.abs(same - pts).max() < 1e-4
    # the principal point is a fixed point
    pp = oracle.undistort_points(TUM1_K[None, 2:4], TUM1_K, TUM1_DIST)
    assert np.allclose(pp, TUM1_K[None, 2:4], atol=1e-4)


def test_compute_image_bounds(oracle):
    b = oracle.compute_image_bounds(640, 480, TUM1_K, TUM1_DIST)
    c = oracle.undistort_points(np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32), TUM1_K, TUM1_DIST)
    assert b[0] == min(c[0, 0], c[2, 0]) and b[2] == max(c[1, 0], c[3, 0])
    assert b[1] == min(c[0, 1], c[1, 1]) and b[3] == max(c[2, 1], c[3, 1])
    assert np.array_equal(oracle.compute_image_bounds(640, 480, TUM1_K, np.zeros(5, np.float32)), [0, 0, 640, 480])
    assert np.array_equal(oracle.compute_image_bounds(640, 480, TUM1_K, np.zeros(0, np.float32)), [0, 0, 640, 480])


# ---------------------------------------------------------------- marker pose (IPPE; marker.cpp:322-344, ippe.cpp)
def test_marker_pose_recovers_synthetic_poses(oracle):
    """The restated solver is pinned by geometry: corners projected from a known pose give that pose back (to the accuracy
    of five undistortion iterations and float corners), and the second IPPE solution reprojects worse."""
    import pose_cases as pc
    for R, t, c in pc.random_cases(200, 7):
        r1, t1, r2, t2, err = oracle.marker_pose(c, 0.187, pc.K4, pc.DIST)
        assert err[0] <= err[1]
        assert err[0] < 0.05, err
        assert np.abs(pc.rodrigues(r1) - R).max() < 2e-2 and np.abs(t1 - t).max() < 5e-3 * t[2] + 1e-3
        # both solutions are rotations and keep the marker in front of the camera
        for r, tt in ((r1, t1), (r2, t2)):
            Rm = pc.rodrigues(r)
            assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-9) and tt[2] > 0
        # reported errors are what the poses reproject to
        for r, tt, e in ((r1, t1, err[0]), (r2, t2, err[1])):
            d = pc.project(pc.object_points(0.187), pc.rodrigues(r), tt).astype(np.float32) - c
            assert abs(np.sqrt((d.astype(np.float64) ** 2).sum() / 8) - e) < 1e-3 + 1e-3 * e


def test_marker_pose_scale_and_camera_resize(oracle):
    import pose_cases as pc
    R, t, c = pc.random_cases(1, 11)[0]
    _, t1, _, _, _ = oracle.marker_pose(c, 0.187, pc.K4, pc.DIST)
    _, t2, _, _, _ = oracle.marker_pose(c, 0.374, pc.K4, pc.DIST)
    assert np.allclose(t2, 2 * t1, rtol=1e-6)                        # translation scales with the marker size
    assert np.array_equal(oracle.camera_resize(pc.K4, (640, 480), (640, 480)), pc.K4)
    k = oracle.camera_resize(pc.K4, (1280, 720), (640, 480))        # Frame.cc:132 hard-codes CamSize 1280x720
    ax, ay = np.float32(640) / np.float32(1280), np.float32(480) / np.float32(720)
    assert np.array_equal(k, np.array([pc.K4[0] * ax, pc.K4[1] * ay, pc.K4[2] * ax, pc.K4[3] * ay], np.float32))


# ---------------------------------------------------------------- DBoW2 vocabulary transform (TemplatedVocabulary.h)
def _brute_transform(voc, feats, levelsup):
    """Independent numpy restatement of the tree descent (no shared code with oracle/bow_oracle.cpp)."""
    parent = np.concatenate([[0], voc["parent"]]); n = len(parent)
    children = [[] for _ in range(n)]
    for i in range(1, n):
        children[parent[i]].append(i)
    desc = np.concatenate([np.zeros((1, 32), np.uint8), voc["desc"]])
    wid = np.zeros(n, np.int64); wid[1:][voc["is_leaf"] > 0] = np.arange(int((voc["is_leaf"] > 0).sum()))
    weight = np.concatenate([[0.0], voc["weight"]])
    out = []
    for f in feats:
        node, level, nid, nid_level = 0, 0, None, voc["L"] - levelsup
        if nid_level <= 0:
            nid = 0
        while children[node]:
            ch = children[node]
            d = np.unpackbits(desc[ch] ^ f, axis=1).sum(1)
            node = ch[int(np.argmin(d))]                  # argmin = first minimum, like the strict '<'
            level += 1
            if level == nid_level:
                nid = node
        out.append((wid[node], node if nid is None else nid, weight[node]))
    return out


@pytest.mark.parametrize("k,L,seed,levelsup", [(10, 3, 1, 1), (4, 5, 2, 4), (10, 4, 3, 2), (3, 2, 4, 4)])
def test_vocabulary_transform_oracle(oracle, k, L, seed, levelsup, tmp_path):
    import voc_cases as vc
    voc = vc.make(k, L, seed)
    o = oracle.VocabularyOracle.from_arrays(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    feats = vc.features(voc, 400, seed)
    got = o.transform(feats, levelsup)
    want = _brute_transform(voc, feats, levelsup)
    assert np.array_equal(got["word"], [w[0] for w in want])
    assert np.array_equal(got["node"], [w[1] for w in want])
    assert np.array_equal(got["weight"], [w[2] for w in want])
    # BowVector = per-word sums, L1-normalised; FeatureVector = features grouped by node, both over weight > 0 only
    keep = got["weight"] > 0
    assert 0 < keep.sum() < len(feats)                    # the stop words drop some
    words, vals = got["bow"]
    assert np.array_equal(words, np.unique(got["word"][keep])) and abs(vals.sum() - 1.0) < 1e-12
    raw = np.array([got["weight"][keep][got["word"][keep] == w].sum() for w in words])
    assert np.allclose(vals, raw / raw.sum(), rtol=1e-12)
    nodes, off, feat = got["fv"]
    assert np.array_equal(nodes, np.unique(got["node"][keep])) and off[0] == 0 and off[-1] == keep.sum()
    for j, nd in enumerate(nodes):
        assert np.array_equal(feat[off[j]:off[j + 1]], np.flatnonzero(keep & (got["node"] == nd)))
    # the text loader builds the same tree; a trailing newline adds the reference's phantom root child
    p = tmp_path / "voc.txt"
    vc.write_text(voc, p, trailing_newline=False)
    t = oracle.VocabularyOracle.load_text(str(p))
    assert t.info() == o.info()
    g2 = t.transform(feats, levelsup)
    assert all(np.array_equal(g2[f], got[f]) for f in ("word", "node", "weight"))
    vc.write_text(voc, p, trailing_newline=True)
    t2 = oracle.VocabularyOracle.load_text(str(p))
    assert t2.info()["nodes"] == o.info()["nodes"] + 1 and t2.info()["words"] == o.info()["words"]
    g3 = t2.transform(feats, levelsup)
    sparse = np.unpackbits(feats, axis=1).sum(1) == 1
    assert np.all(g3["weight"][sparse] == 0)              # one-bit descriptors sit closest to the all-zero phantom: dropped
    assert np.array_equal(g3["word"][~sparse], got["word"][~sparse])


# ---------------------------------------------------------------- on-disk keyframe features (Map.cc:297-321, :478-511)
KF_REC = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                   ("cols", "<i4"), ("desc", "u1", 32), ("mp", "<u8")])


def test_keyframe_feature_records(oracle):
    assert KF_REC.itemsize == 68
    k, d = oracle.OrbOracle(500, 1.2, 4, 20, 7).extract(synth.scene(240, 320, 7, n_markers=2, side_range=(40, 70))[0])
    rng = np.random.default_rng(0)
    mp = rng.integers(0, 10000, len(k)).astype(np.uint64); mp[rng.random(len(k)) < 0.3] = np.uint64(2**64 - 1)
    buf = oracle.keyframe_features_pack(k, d, mp)
    rec = buf.view(KF_REC)                               # the packed layout, field by field
    assert len(rec) == len(k) and np.all(rec["cols"] == 32)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(rec[f], k[f])
    assert np.array_equal(rec["desc"], d) and np.array_equal(rec["mp"], mp)
    k2, d2, m2 = oracle.keyframe_features_unpack(buf, len(k))
    assert np.array_equal(d2, d) and np.array_equal(m2, mp) and np.all(k2["class_id"] == -1)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        assert np.array_equal(k2[f], k[f])
    none = oracle.keyframe_features_pack(k, d, None).view(KF_REC)
    assert np.all(none["mp"] == np.uint64(2**64 - 1))    # ULONG_MAX = no map point
    bad = buf.copy(); bad.view(KF_REC)["cols"][3] = 31
    with pytest.raises(ValueError):
        oracle.keyframe_features_unpack(bad, len(k))


def _extras():
    g = np.load(os.path.join(GOLD, "extras_stream1000.npz"))
    m = np.load(os.path.join(GOLD, "match_stream1000.npz"))
    return g, m


def test_golden_extras(oracle):
    """Pose, undistortion, vocabulary transform, SearchByBoW, last-frame projection search and keyframe records against the
    committed vectors (tests/golden/extras_stream1000.npz; inputs: tests/pose_cases.py, tests/voc_cases.py, match_stream1000)."""
    import hashlib
    import pose_cases as pc
    import voc_cases as vc
    g, m = _extras()
    for c, want, e in zip(g["corners"], g["poses"], g["pose_err"]):
        r1, t1, r2, t2, err = oracle.marker_pose(c, 0.187, pc.K4, pc.DIST)
        assert np.allclose(np.concatenate([r1, t1, r2, t2]), want, rtol=1e-9, atol=1e-12) and np.allclose(err, e, atol=1e-5)
    assert np.array_equal(oracle.undistort_points(g["pts"], pc.K4, pc.DIST), g["undistorted"])
    assert np.array_equal(oracle.compute_image_bounds(640, 480, pc.K4, pc.DIST), g["bounds"])
    voc = vc.make(10, 4, 41, irregular=False)
    ov = oracle.VocabularyOracle.from_arrays(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    t1 = ov.transform(m["d1"], 2)
    assert np.array_equal(t1["word"], g["word1"]) and np.array_equal(t1["node"], g["node1"])
    assert np.array_equal(t1["bow"][0], g["bow1_words"]) and np.array_equal(t1["bow"][1], g["bow1_values"])
    fv1 = (g["fv1_nodes"], g["fv1_offsets"], g["fv1_features"]); fv2 = (g["fv2_nodes"], g["fv2_offsets"], g["fv2_features"])
    assert all(np.array_equal(a, b) for a, b in zip(t1["fv"], fv1))
    nb, b12, _ = oracle.search_by_bow(m["k1"], m["d1"], fv1, m["k2"], m["d2"], fv2, g["valid1"], None, 0.7, True, 50, 30 / 360.0)
    assert nb == int(g["bow_nmatches"][0]) and np.array_equal(b12, g["bow_match12"])
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32); sf = np.array([1.2 ** i for i in range(8)], np.float32)
    nl, ml = oracle.search_by_projection_last_frame(m["k2"], m["d2"], 640, 480, m["k1"], g["valid1"], g["x3Dw"], m["d1"], g["Tcw"], K4, sf, 15.0)
    assert nl == int(g["last_nmatches"][0]) and np.array_equal(ml, g["last_match_cur"])
    rec = oracle.keyframe_features_pack(m["k1"], m["d1"], np.arange(len(m["k1"]), dtype=np.uint64))
    assert np.array_equal(np.frombuffer(hashlib.sha256(rec.tobytes()).digest(), np.uint8), g["kf_sha256"])


# ---------------------------------------------------------------- PredictScale / relocalisation search (a13)
def test_predict_scale_known_answers(oracle):
    """MapPoint::PredictScale (MapPoint.cc:414-446): ceil(logf(mfMaxDistance / dist) / mfLogScaleFactor), clamped to [0, nlevels - 1]."""
    L = float(np.log(np.float32(1.2)))
    assert oracle.predict_scale(1.5, 1.0, L, 8) == 3          # log(1.5) / log(1.2) = 2.22
    assert oracle.predict_scale(1.0, 1.0, L, 8) == 0          # ratio 1
    assert oracle.predict_scale(1.0, 2.0, L, 8) == 0          # ratio < 1: negative -> 0
    assert oracle.predict_scale(100.0, 1.0, L, 8) == 7        # 25.3 -> nlevels - 1
    assert oracle.predict_scale(1.25, 1.0, L, 8) == 2         # just above one level
    assert oracle.predict_scale(2.0, 1.0, L, 12) == 4         # 3.80


def test_float_log_choice_is_immaterial():
    """The reference evaluates PredictScale's log as logf (float overload); the device uses the correctly rounded float log.
    glibc's logf differs from it on a small share of inputs, and the predicted level never changes on 10^6 random ratios."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.logf.restype = ctypes.c_float; libm.logf.argtypes = [ctypes.c_float]
    r = np.random.default_rng(3).uniform(0.3, 6.0, 1_000_000).astype(np.float32)
    cr = np.log(r.astype(np.float64)).astype(np.float32)
    sample = r[:20000]
    lm = np.array([libm.logf(float(x)) for x in sample], np.float32)
    differ = float((lm != cr[:20000]).mean())
    assert differ < 0.02, differ
    L = np.float32(np.log(np.float32(1.2)))
    lv_cr = np.ceil(cr[:20000] / L)
    lv_lm = np.ceil(lm / L)
    assert np.array_equal(lv_cr, lv_lm)


def test_search_by_projection_keyframe_hand_case(oracle):
    """ORBmatcher.cc:1476-1603 on a hand-built frame: a point straight ahead is matched at its keypoint; a point BEHIND the
    camera is projected all the same (this variant has no depth gate, :1503-1512) and takes the keypoint at its mirror image;
    a point outside [0.8 mfMin, 1.2 mfMax] is skipped; ORBdist is honoured; a taken keypoint is not reused."""
    K4 = np.array([500.0, 500.0, 320.0, 240.0], np.float32)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    L = np.float32(np.log(np.float32(1.2)))
    kps = np.zeros(3, oracle.KP_DTYPE)
    kps["x"] = [320.0, 420.0, 100.0]; kps["y"] = [240.0, 240.0, 100.0]; kps["octave"] = [0, 1, 0]; kps["angle"] = [10.0, 10.0, 10.0]
    d = np.zeros((3, 32), np.uint8); d[1, 0] = 0xFF; d[2, :4] = 0xFF
    T = np.eye(3, 4, dtype=np.float32); Ow = np.zeros(3, np.float32)
    p = np.array([[0, 0, 2.0], [-0.4, 0, -2.0], [0, 0, 2.0], [0, 0, 50.0]], np.float32)   # ahead; behind -> u = 420; duplicate; too far
    mfmax = np.array([2.0, 2.3, 2.0, 2.0], np.float32); mfmin = mfmax / sf[-1]
    md = np.zeros((4, 32), np.uint8); md[1, 0] = 0xFF
    ang = np.array([10.0, 10.0, 10.0, 10.0], np.float32)
    n, m = oracle.search_by_projection_keyframe(kps, d, 640, 480, ang, None, p, mfmin, mfmax, md, T, Ow, K4, sf, L, 10.0, 100, check_orientation=False)
    assert n == 2 and m.tolist() == [0, 1, -1]
    # ORBdist: keypoint 1's descriptor is 8 bits away from a zero descriptor
    md2 = md.copy(); md2[1, 0] = 0
    assert oracle.search_by_projection_keyframe(kps, d, 640, 480, ang, None, p, mfmin, mfmax, md2, T, Ow, K4, sf, L, 10.0, 7, check_orientation=False)[0] == 1
    assert oracle.search_by_projection_keyframe(kps, d, 640, 480, ang, None, p, mfmin, mfmax, md2, T, Ow, K4, sf, L, 10.0, 8, check_orientation=False)[0] == 2
    # taken keypoint 0: nothing else lies in the window of the points ahead
    tk = np.array([1, 0, 0], np.uint8)
    n, m = oracle.search_by_projection_keyframe(kps, d, 640, 480, ang, None, p, mfmin, mfmax, md, T, Ow, K4, sf, L, 10.0, 100, taken_cur=tk,
                                                check_orientation=False)
    assert n == 1 and m.tolist() == [-1, 1, -1]


# ---------------------------------------------------------------- the reference's own BowVector / FeatureVector (oracle/_ref)
def _dbow2_ref():
    import ctypes as C
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libdbow2_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libdbow2_ref.so not built (make -C oracle ref needs /root/reference)")
    L = C.CDLL(so)
    L.dbow2_ref_bowvector.restype = C.c_int
    L.dbow2_ref_bowvector.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.dbow2_ref_featurevector.restype = C.c_int
    L.dbow2_ref_featurevector.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3
    return L


@pytest.mark.parametrize("weighting,scoring", [(0, 0), (1, 0), (2, 0), (3, 0), (0, 1), (0, 5), (1, 5), (0, 2)])
def test_bow_vectors_pinned_by_reference_classes(oracle, weighting, scoring):
    """The oracle's BowVector / FeatureVector (oracle/bow_oracle.cpp restating BowVector.cpp:32-89, FeatureVector.cpp:30-45 and the
    accumulation / normalisation switch of TemplatedVocabulary.h:1143-1191, ScoringObject.h:74-91) against the REFERENCE's own
    classes compiled from /root/reference (oracle/_ref/libdbow2_ref.so): the per-feature (word, weight, node) stream of the
    oracle's tree descent goes through the real addWeight / addIfNotExist / normalize / addFeature -- doubles bit-exact."""
    import voc_cases as vc
    R = _dbow2_ref()
    voc = vc.make(9, 3, 17 + weighting + 10 * scoring, irregular=True)
    ov = oracle.VocabularyOracle.from_arrays(9, 3, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    desc = np.random.default_rng(weighting * 7 + scoring).integers(0, 256, (700, 32), dtype=np.uint8)
    desc[100:140] = desc[0:40]                     # repeated words: accumulation order matters
    t = ov.transform(desc, 2)
    keep = t["weight"] > 0                          # "not stopped" (TemplatedVocabulary.h:1159,1183)
    word = np.ascontiguousarray(t["word"][keep], np.uint32); w = np.ascontiguousarray(t["weight"][keep], np.float64)
    n = len(word)
    assert n > 300
    mode = 0 if weighting in (0, 1) else 1
    norm = {0: 1, 1: 2, 2: 1, 3: 1, 4: 1, 5: 0}[scoring]     # ScoringObject.h:74-91: L1 / L2 / none (DOT_PRODUCT)
    ow = np.zeros(n, np.uint32); ov_ = np.zeros(n, np.float64)
    k = R.dbow2_ref_bowvector(word.ctypes.data, w.ctypes.data, n, mode, norm, ow.ctypes.data, ov_.ctypes.data)
    if mode == 0 and norm == 0:                     # TF / TF_IDF without normalisation: divided by the vector size (:1168-1174)
        ov_[:k] = ov_[:k] / float(k)
    assert np.array_equal(ow[:k], t["bow"][0])
    assert np.array_equal(ov_[:k].view(np.uint64), t["bow"][1].view(np.uint64))
    node = np.ascontiguousarray(t["node"][keep], np.uint32); feat = np.ascontiguousarray(np.nonzero(keep)[0], np.uint32)
    on = np.zeros(n, np.uint32); oo = np.zeros(n + 1, np.int32); of = np.zeros(n, np.uint32)
    kf = R.dbow2_ref_featurevector(node.ctypes.data, feat.ctypes.data, n, on.ctypes.data, oo.ctypes.data, of.ctypes.data)
    assert np.array_equal(on[:kf], t["fv"][0]) and np.array_equal(oo[:kf + 1], t["fv"][1]) and np.array_equal(of[:oo[kf]], t["fv"][2])


def test_distinctive_descriptor_known_answers(oracle):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:270-333): least median distance, median = sorted[(int)(0.5 (N - 1))],
    first row wins ties; a point without observations is left alone (-1)."""
    z = np.zeros((1, 32), np.uint8)
    a = z.copy(); a[0, 0] = 0x0F            # 4 bits from z
    b = z.copy(); b[0, :2] = 0xFF           # 16 bits from z, 12 from a
    # N = 3: rows sorted (0,4,16) (0,4,12) (0,12,16): medians 4, 4, 12 -> first of the tie = 0
    d = np.concatenate([z, a, b])
    assert oracle.distinctive_descriptors(d, [0, 3]).tolist() == [0]
    # N = 2: median index (int)(0.5) = 0 -> every row's median is its own 0 distance -> index 0
    assert oracle.distinctive_descriptors(np.concatenate([b, z]), [0, 2]).tolist() == [0]
    # N = 4, index (int)1.5 = 1: rows z:(0,0,4,16)->0  z:(0,0,4,16)->0  a:(0,4,4,12)->4  b -> 12
    d = np.concatenate([b, a, z, z])
    assert oracle.distinctive_descriptors(d, [0, 4]).tolist() == [2]
    # several points, one of them empty, one with a single observation
    d = np.concatenate([z, a, b, a])
    assert oracle.distinctive_descriptors(d, [0, 3, 3, 4]).tolist() == [0, -1, 0]
