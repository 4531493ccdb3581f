"""GPU parity: Hamming matching kernels vs the CPU oracle (bit-exact: integer work)."""
import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nq,nt,init", [(1, 1, 256), (7, 300, 256), (1000, 1000, 256), (1030, 999, 2**31 - 1),
                                        (513, 4100, 256), (300, 0, 256)])
def test_knn2_all_pairs(orbfe, oracle, nq, nt, init):
    Q = synth.random_descriptors(nq, 5)
    T = synth.random_descriptors(max(nt, 1), 6)[:nt]
    # duplicates and near-duplicates exercise the tie rules (first candidate wins)
    if nt > 20:
        T[10] = T[3]; T[11] = T[3]; Q[0] = T[3]
    got = orbfe.knn2(Q, T, init)
    want = oracle.knn2(Q, T, init)
    for g, w, nm in zip(got, want, ("best_idx", "best_dist", "second_dist")):
        assert np.array_equal(g, w), nm


@pytest.mark.parametrize("path", [1, 2])
@pytest.mark.parametrize("nq,nt,init", [(1, 1, 256), (33, 65, 256), (1000, 1000, 256), (1030, 999, 2**31 - 1),
                                        (300, 257, 60), (5, 0, 256)])
def test_knn2_both_kernels(orbfe, oracle, path, nq, nt, init):
    """The VALU tile kernel (1) and the matrix-core kernel (2) implement the same rule, ties and `init` included."""
    Q = synth.random_descriptors(nq, nq * 1000 + nt)
    T = synth.random_descriptors(max(nt, 1), nq * 1000 + nt + 1)[:nt]
    if nt > 40:   # duplicates and near-duplicates: ties in best and in second
        T[7] = T[3]; T[40] = T[3]; Q[0] = T[3]
        if nq > 2: Q[2] = T[3] ^ np.uint8(1)
    orbfe.debug_control("knn2_path", path)
    try:
        got = orbfe.knn2(Q, T, init)
    finally:
        orbfe.debug_control("knn2_path", 0)
    want = oracle.knn2(Q, T, init)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_knn2_large_property(orbfe):
    """10k x 10k (config C5): checked through size-independent properties instead of the (slow) oracle."""
    n = 10000
    D = synth.random_descriptors(n, 5)
    bi, bd, sd = orbfe.knn2(D, D, 256)
    assert np.array_equal(bi, np.arange(n))            # every descriptor is its own nearest neighbour ...
    assert np.all(bd == 0) and np.all(sd > 0)          # ... at distance 0, the runner-up is farther
    # runner-up distance = min over t != q of d(q, t), spot-checked with numpy popcounts
    idx = np.array([0, 17, 4999, 9999])
    x = np.unpackbits(D[idx][:, None, :] ^ D[None, :, :], axis=2).sum(2)
    x[np.arange(len(idx)), idx] = 10**6
    assert np.array_equal(sd[idx], x.min(1))


def test_knn2_c5_oracle_sample_both_kernels():
    """bench.py's C5 matching leg (10k x 10k, seed 5, resident): 192 queries of the full-size problem against the oracle, for the
    VALU kernel and the matrix-core kernel, plus the self-match property over all 10^4 queries -- so the driver's GPU test
    record holds an oracle check of the full-size configuration, not only `bench.py --config C5`.  In its own process, torch
    imported first (torch brings its own HIP runtime; a process that has already initialised the system one through liborbfe.so
    may find no device through torch afterwards)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, torch; sys.path[:0] = [%r, %r]; import bench, oracle_lib; from orb_slam2_aruco_amd import binding; "
            "print(json.dumps(bench.c5_match_leg(binding, torch, torch.device('cuda', 0), oracle_lib)))" % (root, os.path.join(root, "tests")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    import json
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["verified_queries"] == 192 and res["pairs"] == 10**8
    assert res["valu"]["launch_us"] > 0 and res["mfma_i8"]["launch_us"] > 0


def _frame_pair(orbfe, seed):
    s = synth.stream(480, 640, 2, seed)
    ex = orbfe.ORBextractor(1000, 1.2, 8, 20, 7)
    return ex(s[0]), ex(s[1])


@pytest.mark.parametrize("seed,window,ratio,ori", [(1000, 100, 0.9, True), (1234, 100, 0.9, False),
                                                   (77, 30, 0.8, True), (5, 400, 0.95, True)])
def test_search_for_initialization(orbfe, oracle, seed, window, ratio, ori):
    (k1, d1), (k2, d2) = _frame_pair(orbfe, seed)
    m = orbfe.ORBmatcher(ratio, ori)
    n, m12, prev = m.SearchForInitialization(k1, d1, k2, d2, 640, 480, None, window)
    on, om12, oprev = oracle.search_for_initialization(k1, d1, k2, d2, 640, 480, None, window, ratio, ori)
    assert n == on and n > 0
    assert np.array_equal(m12, om12)
    assert np.array_equal(prev, oprev)


def test_search_for_initialization_second_call_uses_prev(orbfe, oracle):
    (k1, d1), (k2, d2) = _frame_pair(orbfe, 31)
    m = orbfe.ORBmatcher(0.9, True)
    n, m12, prev = m.SearchForInitialization(k1, d1, k2, d2, 640, 480, None, 100)
    n2, m12b, prev2 = m.SearchForInitialization(k1, d1, k2, d2, 640, 480, prev, 100)
    on2, om12b, oprev2 = oracle.search_for_initialization(k1, d1, k2, d2, 640, 480, prev, 100, 0.9, True)
    assert n2 == on2 and np.array_equal(m12b, om12b) and np.array_equal(prev2, oprev2)


def _projection_case(oracle_mod, seed, nq, jitter):
    """A frame's keypoints + queries placed near real keypoints (as projected map points are), descriptors = the
    keypoint's with a few flipped bits, so that best / second-best, octave ties and taken keypoints all occur."""
    rng = np.random.default_rng(seed)
    img, _ = synth.scene(480, 640, seed % 7 + 1, n_markers=3, side_range=(40, 90))
    kps, desc = oracle_mod.OrbOracle(1000, 1.2, 8, 20, 7).extract(img)
    pick = rng.integers(0, len(kps), nq)
    q = np.zeros(nq, oracle_mod.WINDOW_QUERY_DTYPE)
    q["x"] = kps["x"][pick] + rng.normal(0, jitter, nq).astype(np.float32)
    q["y"] = kps["y"][pick] + rng.normal(0, jitter, nq).astype(np.float32)
    lvl = kps["octave"][pick]
    q["r"] = (rng.choice([2.5, 4.0], nq) * 1.2 ** lvl).astype(np.float32) * rng.choice([1.0, 3.0], nq).astype(np.float32)
    q["min_level"] = lvl - 1
    q["max_level"] = lvl
    special = rng.random(nq)
    q["min_level"][special < 0.05] = 0; q["max_level"][special < 0.05] = -1      # no octave test
    q["x"][special > 0.97] = -500.0                                                  # window outside the image
    qd = desc[pick].copy()
    flips = rng.integers(0, 256, (nq, 6))
    for i in range(nq):
        for b in flips[i][: rng.integers(0, 7)]:
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    taken = (rng.random(len(kps)) < 0.15).astype(np.uint8)
    return kps, desc, q, qd, taken


@pytest.mark.parametrize("seed,nq,jitter", [(1, 600, 1.0), (2, 1500, 3.0), (3, 64, 0.5)])
@pytest.mark.parametrize("mode", [0, 1])
def test_search_by_projection(orbfe, oracle, seed, nq, jitter, mode):
    kps, desc, q, qd, taken = _projection_case(oracle, seed, nq, jitter)
    want = oracle.search_by_projection(kps, desc, 640, 480, q, qd, taken, mode, 100, 0.8)
    got = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, taken, mode, 100, 0.8)
    for f in ("best_idx", "best_dist", "best_level", "second_dist", "second_level"):
        assert np.array_equal(got[f], want[f]), f
    if mode == 1:
        assert got["nmatches"] == want["nmatches"] and got["nmatches"] > nq // 4
        assert np.array_equal(got["match"], want["match"])
        assert np.array_equal(got["taken"], want["taken"])
        m = got["match"][got["match"] >= 0]
        assert len(np.unique(m)) == len(m)          # a keypoint is matched at most once
        # map points without observations do not block the keypoint they receive (:91-93): it can be assigned again
        obs = (np.random.default_rng(seed).random(len(q)) < 0.5).astype(np.uint8)
        want = oracle.search_by_projection(kps, desc, 640, 480, q, qd, taken, 1, 100, 0.8, q_observed=obs)
        got = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, taken, 1, 100, 0.8, q_observed=obs)
        assert got["nmatches"] == want["nmatches"] and np.array_equal(got["match"], want["match"])
        assert np.array_equal(got["taken"], want["taken"])


def test_search_by_projection_edge_cases(orbfe, oracle):
    kps, desc, q, qd, taken = _projection_case(oracle, 5, 40, 1.0)
    # no queries, no keypoints, no taken array
    assert orbfe.search_by_projection(kps, desc, 640, 480, q[:0], qd[:0], None, 1)["nmatches"] == 0
    got = orbfe.search_by_projection(kps[:0], desc[:0], 640, 480, q, qd, None, 0)
    assert (got["best_idx"] == -1).all() and (got["best_dist"] == 256).all() and (got["second_level"] == -1).all()
    want = oracle.search_by_projection(kps, desc, 640, 480, q, qd, None, 1, 100, 0.8)
    got = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, None, 1, 100, 0.8)
    assert np.array_equal(got["match"], want["match"])
    # a huge window: every keypoint is a candidate of every query (row stride grows)
    q["r"] = 2000.0; q["min_level"] = 0; q["max_level"] = -1
    want = oracle.search_by_projection(kps, desc, 640, 480, q, qd, None, 0)
    got = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, None, 0)
    for f in ("best_idx", "best_dist", "second_dist", "best_level", "second_level"):
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("init", [256, 60, 2**31 - 1])
def test_knn2_csr_guided(orbfe, oracle, init):
    """Guided best / second-best over candidate lists built by GetFeaturesInArea (what SearchByBoW / Fuse feed it)."""
    img, _ = synth.scene(480, 640, 3, n_markers=3, side_range=(40, 90))
    kps, desc = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(img)
    rng = np.random.default_rng(8)
    nq = 700
    pick = rng.integers(0, len(kps), nq)
    off, idx = oracle.features_in_area(kps, 640, 480, kps["x"][pick] + 2, kps["y"][pick] - 1, 25.0, 0, -1)
    Q = desc[pick] ^ np.uint8(3)
    off[5] = off[4]                      # an empty list in the middle stays legal
    got = orbfe.knn2_csr(Q, desc, off, idx, init)
    want = oracle.knn2_csr(Q, desc, off, idx, init)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    with pytest.raises(orbfe.OrbfeError):
        orbfe.knn2_csr(Q, desc, off, np.where(idx == idx[0], len(desc), idx), init)   # candidate out of range


def test_search_by_projection_golden(orbfe):
    """Against the committed vectors (no oracle needed): tests/golden/projection_stream1000.npz."""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(gold, "match_stream1000.npz"))
    p = np.load(os.path.join(gold, "projection_stream1000.npz"))
    q = p["queries"].view(orbfe.WINDOW_QUERY_DTYPE).reshape(-1)
    r0 = orbfe.search_by_projection(g["k2"], g["d2"], 640, 480, q, g["d1"][p["sel"]], p["taken"], 0, 100, 0.8)
    for f in ("best_idx", "best_dist", "best_level", "second_dist", "second_level"):
        assert np.array_equal(r0[f], p[f]), f
    r1 = orbfe.search_by_projection(g["k2"], g["d2"], 640, 480, q, g["d1"][p["sel"]], p["taken"], 1, 100, 0.8)
    assert r1["nmatches"] == int(p["nmatches"][0])
    assert np.array_equal(r1["match"], p["match"]) and np.array_equal(r1["taken"], p["taken_after"])


def test_grid_bounds_of_a_distorted_camera(orbfe, oracle):
    """With lens distortion the Frame grid spans the undistorted image bounds (Frame.cc:418-451), not 0..cols."""
    bounds = np.array([-14.25, -9.5, 655.75, 489.0], np.float32)
    kps, desc, q, qd, taken = _projection_case(oracle, 4, 500, 2.0)
    kps = kps.copy()
    kps["x"] = kps["x"] * 1.04 - 13.0          # "undistorted" positions reaching beyond the sensor
    kps["y"] = kps["y"] * 1.03 - 8.0
    q = q.copy(); q["x"] = q["x"] * 1.04 - 13.0; q["y"] = q["y"] * 1.03 - 8.0
    for mode in (0, 1):
        want = oracle.search_by_projection(kps, desc, 640, 480, q, qd, taken, mode, 100, 0.8, bounds)
        got = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, taken, mode, 100, 0.8, bounds=bounds)
        for f in ("best_idx", "best_dist", "best_level", "second_dist", "second_level") + (("match",) if mode else ()):
            assert np.array_equal(got[f], want[f]), (mode, f)
    s = synth.stream(480, 640, 2, 1000)
    o = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = o.extract(s[0]), o.extract(s[1])
    for k in (k1, k2):
        k["x"] = k["x"] * 1.04 - 13.0; k["y"] = k["y"] * 1.03 - 8.0
    wn, wm, wp = oracle.search_for_initialization(k1, d1, k2, d2, 640, 480, None, 100, 0.9, True, bounds)
    gn, gm, gp = orbfe.ORBmatcher(0.9, True).SearchForInitialization(k1, d1, k2, d2, 640, 480, None, 100, bounds=bounds)
    assert gn == wn and np.array_equal(gm, wm) and np.array_equal(gp, wp)


# TUM1.yaml of the reference (Examples/Monocular/TUM1.yaml): the camera the reference's monocular example is run with
TUM1_K = np.array([517.306408, 516.469215, 318.643040, 255.313989], np.float32)
TUM1_DIST = np.array([0.262383, -0.953104, -0.005358, 0.002628, 1.163314], np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("ndist", [4, 5, 8, 12])
def test_undistort_points(orbfe, oracle, ndist):
    """cv::undistortPoints with P = K (Frame.cc:357-416): bit-exact float results (double arithmetic on both sides)."""
    rng = np.random.default_rng(ndist)
    pts = np.concatenate([rng.uniform([-5, -5], [645, 485], (5000, 2)),
                          [[0, 0], [640, 0], [0, 480], [640, 480], [318.643040, 255.313989]]]).astype(np.float32)
    dist = np.concatenate([TUM1_DIST, [0.01, -0.02, 0.005, 1e-3, -2e-3, 5e-4, 1e-4]]).astype(np.float32)[:ndist]
    want = oracle.undistort_points(pts, TUM1_K, dist)
    got = orbfe.undistort_points(pts, TUM1_K, dist)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert orbfe.undistort_points(np.zeros((0, 2), np.float32), TUM1_K, dist).shape == (0, 2)
    Kmat = np.array([[TUM1_K[0], 0, TUM1_K[2]], [0, TUM1_K[1], TUM1_K[3]], [0, 0, 1]], np.float32)
    assert np.array_equal(orbfe.undistort_points(pts, Kmat, dist), got)      # 3x3 K accepted like mK


@pytest.mark.gpu
def test_compute_image_bounds_and_keypoints(orbfe, oracle):
    want = oracle.compute_image_bounds(640, 480, TUM1_K, TUM1_DIST)
    got = orbfe.ComputeImageBounds(640, 480, TUM1_K, TUM1_DIST)
    assert np.array_equal(got, want)
    assert not np.array_equal(got, [0, 0, 640, 480])
    assert np.array_equal(orbfe.ComputeImageBounds(640, 480, TUM1_K, np.zeros(5, np.float32)), [0, 0, 640, 480])
    assert np.array_equal(orbfe.ComputeImageBounds(640, 480, TUM1_K, None), [0, 0, 640, 480])
    s = synth.stream(480, 640, 1, 1000)
    k, d = oracle.OrbOracle(1000, 1.2, 8, 20, 7).extract(s[0])
    un = orbfe.UndistortKeyPoints(k, TUM1_K, TUM1_DIST)
    xy = oracle.undistort_points(np.stack([k["x"], k["y"]], 1), TUM1_K, TUM1_DIST)
    assert np.array_equal(un["x"], xy[:, 0]) and np.array_equal(un["y"], xy[:, 1])
    for f in ("size", "angle", "response", "octave"):
        assert np.array_equal(un[f], k[f])
    same = orbfe.UndistortKeyPoints(k, TUM1_K, np.zeros(5, np.float32))       # Frame.cc:359-363
    assert np.array_equal(same, k)
    with pytest.raises(RuntimeError):
        orbfe.undistort_points(np.zeros((3, 2), np.float32), [0, 1, 2, 3], TUM1_DIST)


@pytest.mark.gpu
def test_undistort_keypoints_batch_device():
    """Extractor records undistorted in place on the device, then matched over the undistorted bounds: the Frame
    constructor's order (Frame.cc:204-233) for a distorted camera, against the oracle end to end.  Runs in its own
    process because torch (device memory) has to initialise HIP before the library does."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "device_pipeline_case.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def _motion_case(oracle, seed, depth_lo=1.0, depth_hi=6.0):
    """Two consecutive stream frames; the last frame's keypoints get map points by back-projection at random depths (last
    pose = identity) and the current pose is a small motion, so the projections land near the keypoints' new positions."""
    s = synth.stream(480, 640, 2, 1000 + seed)
    ex = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    (kl, dl), (kc, dc) = ex.extract(s[0]), ex.extract(s[1])
    rng = np.random.default_rng(seed)
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
    z = rng.uniform(depth_lo, depth_hi, len(kl)).astype(np.float32)
    x3 = np.stack([(kl["x"] - K4[2]) / K4[0] * z, (kl["y"] - K4[3]) / K4[1] * z, z], 1).astype(np.float32)
    a = 0.004 * rng.normal(size=3)
    Rx = np.array([[1, -a[2], a[1]], [a[2], 1, -a[0]], [-a[1], a[0], 1]])
    U, _, Vt = np.linalg.svd(Rx)
    Tcw = np.concatenate([U @ Vt, (0.01 * rng.normal(size=3))[:, None]], 1).astype(np.float32)
    x3[rng.random(len(kl)) < 0.02, 2] *= -1                           # a few points behind the camera (invzc < 0)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    return kc, dc, kl, dl, x3, Tcw, K4, sf, rng


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,ori", [(1, 15.0, True), (2, 7.0, True), (3, 15.0, False), (4, 30.0, True)])
def test_search_by_projection_last_frame(orbfe, oracle, seed, th, ori):
    """SearchByProjection(CurrentFrame, LastFrame, th, mono) (ORBmatcher.cc:1332-1474), projection included: bit-exact."""
    kc, dc, kl, dl, x3, Tcw, K4, sf, rng = _motion_case(oracle, seed)
    valid = (rng.random(len(kl)) < 0.85).astype(np.uint8)
    taken = (rng.random(len(kc)) < 0.05).astype(np.uint8)
    observed = (rng.random(len(kl)) < 0.9).astype(np.uint8)              # unobserved map points do not block their keypoint
    for tk, ob in ((None, None), (taken, observed)):
        want = oracle.search_by_projection_last_frame(kc, dc, 640, 480, kl, valid, x3, dl, Tcw, K4, sf, th, tk, ob, 100, ori)
        got = orbfe.search_by_projection_last_frame(kc, dc, 640, 480, kl, valid, x3, dl, Tcw, K4, sf, th, tk, ob, 100, ori)
        assert got[0] == want[0] and np.array_equal(got[1], want[1])
        assert got[0] > 10
        m = got[1]
        assert np.all(valid[m[m >= 0]] == 1)
        if tk is not None:
            assert np.all(m[taken == 1] == -1)


@pytest.mark.gpu
def test_search_by_projection_last_frame_edges(orbfe, oracle):
    kc, dc, kl, dl, x3, Tcw, K4, sf, rng = _motion_case(oracle, 9)
    nm, m = orbfe.search_by_projection_last_frame(kc, dc, 640, 480, kl[:0], None, x3[:0], dl[:0], Tcw, K4, sf, 15.0)
    assert nm == 0 and np.all(m == -1)
    nm, m = orbfe.search_by_projection_last_frame(kc[:0], dc[:0], 640, 480, kl, None, x3, dl, Tcw, K4, sf, 15.0)
    assert nm == 0 and len(m) == 0
    none = np.zeros(len(kl), np.uint8)
    nm, m = orbfe.search_by_projection_last_frame(kc, dc, 640, 480, kl, none, x3, dl, Tcw, K4, sf, 15.0)
    assert nm == 0 and np.all(m == -1)
    # distorted-camera bounds are honoured by the projection test and the grid
    b = np.array([-14.25, -9.5, 655.75, 489.0], np.float32)
    want = oracle.search_by_projection_last_frame(kc, dc, 640, 480, kl, None, x3, dl, Tcw, K4, sf, 15.0, bounds=b)
    got = orbfe.search_by_projection_last_frame(kc, dc, 640, 480, kl, None, x3, dl, Tcw, K4, sf, 15.0, bounds=b)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    # the caller-projected variant with the same queries gives the same result
    X = x3.astype(np.float32)
    t0 = (Tcw[:, 0] * X[:, 0:1] + Tcw[:, 1] * X[:, 1:2]).astype(np.float32)
    pc = ((t0 + Tcw[:, 2] * X[:, 2:3]).astype(np.float32) + Tcw[:, 3]).astype(np.float32)
    inv = (1.0 / pc[:, 2].astype(np.float64)).astype(np.float32)
    u = ((K4[0] * pc[:, 0]).astype(np.float32) * inv).astype(np.float32) + K4[2]
    v = ((K4[1] * pc[:, 1]).astype(np.float32) * inv).astype(np.float32) + K4[3]
    q = np.zeros(len(kl), orbfe.WINDOW_QUERY_DTYPE)
    ok = ~(inv < 0) & (u >= 0) & (u <= 640) & (v >= 0) & (v <= 480)
    q["x"], q["y"] = u, v
    q["r"] = np.where(ok, np.float32(15.0) * sf[kl["octave"]], np.float32(-1))
    q["min_level"], q["max_level"] = kl["octave"] - 1, kl["octave"] + 1
    want = oracle.search_by_projection_last_frame(kc, dc, 640, 480, kl, None, x3, dl, Tcw, K4, sf, 15.0)
    got = orbfe.search_by_projection_best(kc, dc, 640, 480, q, kl["angle"], dl, 100, 1.0 / 30)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])


def _map_points_for(kl, x3, Tcw, rng, sf):
    """Scale-invariance range, normal and camera centre for map points observed at octave kl.octave from the identity pose."""
    d0 = np.linalg.norm(x3.astype(np.float64), axis=1)
    # mfMaxDistance / mfMinDistance as MapPoint::UpdateNormalAndDepth sets them (MapPoint.cc:395-397), with some spread
    max_d = (d0 * sf[kl["octave"]] * rng.uniform(0.8, 1.15, len(kl))).astype(np.float32)
    min_d = (max_d / sf[-1] * rng.uniform(0.95, 1.2, len(kl))).astype(np.float32)
    nrm = x3 / np.maximum(d0, 1e-9)[:, None] + 0.3 * rng.normal(size=x3.shape)
    nrm = (nrm / np.linalg.norm(nrm, axis=1)[:, None]).astype(np.float32)
    flip = rng.random(len(kl)) < 0.05
    nrm[flip] *= -1                                                   # seen from behind: viewing-angle gate
    R, t = Tcw[:, :3].astype(np.float64), Tcw[:, 3].astype(np.float64)
    Ow = (-R.T @ t).astype(np.float32)
    return min_d, max_d, nrm, Ow


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,chi2", [(1, 3.0, 5.99), (2, 2.5, 5.99), (3, 4.0, 0.0), (4, 10.0, 0.0)])
def test_fuse_search(orbfe, oracle, seed, th, chi2):
    """Matching part of Fuse (ORBmatcher.cc:829-970; chi2 = 0: :972-1104): projection, gates, PredictScale, window, levels,
    reprojection gate, best distance -- bit-exact."""
    kc, dc, kl, dl, x3, Tcw, K4, sf, rng = _motion_case(oracle, seed)
    # the keyframe the points are fused into sees them almost where the source keyframe did: same keypoints, a fifth of the motion
    kc, dc = kl, dl
    Tcw = (np.eye(3, 4) + 0.2 * (Tcw.astype(np.float64) - np.eye(3, 4))).astype(np.float32)
    min_d, max_d, nrm, Ow = _map_points_for(kl, x3, Tcw, rng, sf)
    valid = (rng.random(len(kl)) < 0.9).astype(np.uint8)
    isg = (1.0 / (sf * sf)).astype(np.float32)
    logsf = np.float32(np.log(np.float32(1.2)))
    want = oracle.fuse_search(kc, dc, 640, 480, x3, valid, min_d, max_d, nrm, dl, Tcw, Ow, K4, sf, isg, logsf, th, chi2)
    got = orbfe.fuse_search(kc, dc, 640, 480, x3, valid, min_d, max_d, nrm, dl, Tcw, Ow, K4, sf, isg, logsf, th, chi2)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    fused = (got[1] <= 50).sum()
    assert fused > 20 and np.all(got[0][valid == 0] == -1)
    # the projection alone, as window queries: searched points are exactly the ones Fuse searched
    q = orbfe.project_map_points(x3, valid, min_d, max_d, nrm, Tcw, Ow, K4, 640, 480, sf, logsf, th, 1, 0)
    assert np.all(q["r"][got[0] >= 0] > 0) and np.all(q["r"][valid == 0] < 0)
    assert np.all(q["max_level"][q["r"] > 0] - q["min_level"][q["r"] > 0] == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,orb_dist", [(1, 10.0, 100), (2, 3.0, 64), (3, 10.0, 64), (4, 3.0, 100)])
def test_search_by_projection_keyframe(orbfe, oracle, seed, th, orb_dist):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1476-1603; Relocalization calls it with
    (10, 100) and (3, 64), Tracking.cc:1858, :1875) against its restatement: the dist3D range gate, PredictScale on mfMaxDistance,
    radius th * scale[level], levels +-1, bestDist <= ORBdist, the taken-keypoint coupling, the rotation histogram -- bit-exact,
    through the dedicated entry point AND as orbfe_project_map_points(keyframe_variant = 0, normal = NULL, levels +-1) +
    orbfe_search_by_projection_best."""
    kc, dc, kl, dl, x3, Tcw, K4, sf, rng = _motion_case(oracle, seed)
    if th < 5 or orb_dist < 100:  # the narrow second pass runs on an optimised pose (Tracking.cc:1875): the keyframe's own keypoints, a tenth of the motion
        kc, dc = kl, dl
        Tcw = (np.eye(3, 4) + 0.1 * (Tcw.astype(np.float64) - np.eye(3, 4))).astype(np.float32)
    min_d, max_d, _, Ow = _map_points_for(kl, x3, Tcw, rng, sf)
    # some points lie BEHIND the camera: this variant has no depth gate (:1503-1512) and still projects them
    behind = rng.random(len(kl)) < 0.05
    x3 = x3.copy(); x3[behind, 2] *= -1
    valid = (rng.random(len(kl)) < 0.85).astype(np.uint8)                # has a map point, not bad, not in sAlreadyFound
    taken = (rng.random(len(kc)) < 0.15).astype(np.uint8)                # CurrentFrame.mvpMapPoints[i2] != NULL
    logsf = np.float32(np.log(np.float32(1.2)))
    want = oracle.search_by_projection_keyframe(kc, dc, 640, 480, kl["angle"], valid, x3, min_d, max_d, dl, Tcw, Ow, K4, sf, logsf, th, orb_dist,
                                                taken_cur=taken)
    got = orbfe.search_by_projection_keyframe(kc, dc, 640, 480, kl["angle"], valid, x3, min_d, max_d, dl, Tcw, Ow, K4, sf, logsf, th, orb_dist,
                                              taken_cur=taken)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    assert got[0] > 20 and got[0] == (got[1] >= 0).sum() and np.all(got[1][taken == 1] == -1)
    assert np.all(valid[got[1][got[1] >= 0]] == 1)
    # the composition the header documents
    q = orbfe.project_map_points(x3, valid, min_d, max_d, None, Tcw, Ow, K4, 640, 480, sf, logsf, th, 1, 1, strict_max=False)
    nm, mc = orbfe.search_by_projection_best(kc, dc, 640, 480, q, kl["angle"], dl, orb_dist, 1.0 / 30, taken=taken)
    assert nm == want[0] and np.array_equal(mc, want[1])
    assert np.all(q["max_level"][q["r"] > 0] - q["min_level"][q["r"] > 0] == 2)
    # without the orientation check and with distorted-camera bounds
    b = np.array([-14.25, -9.5, 655.75, 489.0], np.float32)
    want = oracle.search_by_projection_keyframe(kc, dc, 640, 480, kl["angle"], None, x3, min_d, max_d, dl, Tcw, Ow, K4, sf, logsf, th, orb_dist,
                                                check_orientation=False, bounds=b)
    got = orbfe.search_by_projection_keyframe(kc, dc, 640, 480, kl["angle"], None, x3, min_d, max_d, dl, Tcw, Ow, K4, sf, logsf, th, orb_dist,
                                              check_orientation=False, bounds=b)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    # nothing valid, empty sides
    none = np.zeros(len(kl), np.uint8)
    assert orbfe.search_by_projection_keyframe(kc, dc, 640, 480, kl["angle"], none, x3, min_d, max_d, dl, Tcw, Ow, K4, sf, logsf, th, orb_dist)[0] == 0
    assert orbfe.search_by_projection_keyframe(kc[:0], dc[:0], 640, 480, kl["angle"], None, x3, min_d, max_d, dl, Tcw, Ow, K4, sf, logsf, th,
                                               orb_dist)[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,s12", [(1, 7.5, 1.0), (2, 7.5, 1.03), (3, 4.0, 0.97)])
def test_search_by_sim3(orbfe, oracle, seed, th, s12):
    """SearchBySim3 (ORBmatcher.cc:1106-1330): two-stage transforms, gates, PredictScale, both directions and the agreement pass."""
    _, _, kl, dl, x1, Tm, K4, sf, rng = _motion_case(oracle, seed)
    T2w = (np.eye(3, 4) + 0.2 * (Tm.astype(np.float64) - np.eye(3, 4)))
    T1w = np.eye(3, 4)
    R2, t2 = T2w[:, :3], T2w[:, 3]
    z2 = rng.uniform(1.0, 6.0, len(kl))
    c2 = np.stack([(kl["x"] - K4[2]) / K4[0] * z2, (kl["y"] - K4[3]) / K4[1] * z2, z2], 1)
    x2 = ((c2 - t2) @ R2).astype(np.float32)                         # world points seen by keyframe 2 at its keypoints
    # similarity camera 2 -> camera 1: the true relative pose (T1w = I) with a scale and a small error
    R12 = R2.T; t12 = -R2.T @ t2 + 0.003 * rng.normal(size=3)
    sR12 = (np.float32(s12) * R12.astype(np.float32)).astype(np.float32)
    sR21 = ((1.0 / np.float64(np.float32(s12))) * R12.T.astype(np.float32).astype(np.float64)).astype(np.float32)
    t12f = t12.astype(np.float32)
    t21 = -(sR21 @ t12f).astype(np.float32)
    sT12 = np.concatenate([sR12, t12f[:, None]], 1); sT21 = np.concatenate([sR21, t21[:, None]], 1)
    logsf = np.float32(np.log(np.float32(1.2)))

    def kf(x3, cam_T):
        d0 = np.linalg.norm((x3.astype(np.float64) @ cam_T[:, :3].T + cam_T[:, 3]), axis=1)
        mx = (d0 * sf[kl["octave"]] * rng.uniform(0.8, 1.15, len(kl))).astype(np.float32)
        mn = (mx / sf[-1] * rng.uniform(0.95, 1.2, len(kl))).astype(np.float32)
        return dict(kps=kl, desc=dl, p3Dw=x3, valid=(rng.random(len(kl)) < 0.85).astype(np.uint8), min_dist=mn, max_dist=mx, mp_desc=dl)
    kf1, kf2 = kf(x1, T1w), kf(x2, T2w)
    want = oracle.search_by_sim3(kf1, kf2, 640, 480, T1w, T2w, sT12, sT21, K4, sf, logsf, th)
    got = orbfe.search_by_sim3(kf1, kf2, 640, 480, T1w, T2w, sT12, sT21, K4, sf, logsf, th)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    assert got[0] > 30 and got[0] == (got[1] >= 0).sum()
    m = got[1]
    assert np.all(kf1["valid"][m >= 0] == 1) and np.all(kf2["valid"][m[m >= 0]] == 1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th", [(1, 10), (2, 4), (3, 15)])
def test_search_by_projection_sim3(orbfe, oracle, seed, th):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:294-407): the loop-closing variant, bit-exact incl. the
    matched-keypoint coupling."""
    _, _, kl, dl, x3, Tcw, K4, sf, rng = _motion_case(oracle, seed)
    Tcw = (np.eye(3, 4) + 0.2 * (Tcw.astype(np.float64) - np.eye(3, 4))).astype(np.float32)
    min_d, max_d, nrm, Ow = _map_points_for(kl, x3, Tcw, rng, sf)
    valid = (rng.random(len(kl)) < 0.9).astype(np.uint8)
    matched = (rng.random(len(kl)) < 0.2).astype(np.uint8)
    logsf = np.float32(np.log(np.float32(1.2)))
    # the candidate points come in another order than the keypoints, and every point twice: the second copy finds its keypoint taken
    order = np.concatenate([rng.permutation(len(kl)), rng.permutation(len(kl))])
    args = (x3[order], valid[order], min_d[order], max_d[order], nrm[order], dl[order], Tcw, Ow, K4, sf, logsf, th)
    want = oracle.search_by_projection_sim3(kl, dl, 640, 480, matched, *args)
    got = orbfe.search_by_projection_sim3(kl, dl, 640, 480, matched, *args)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    assert got[0] > 50 and np.all(got[1][matched == 1] == -1)


@pytest.mark.gpu
def test_matching_entry_points_are_thread_safe(orbfe, oracle):
    """ORBmatcher members and ComputeBoW are called from the Tracking, LocalMapping and LoopClosing threads at once:
    three threads hammer knn2, the projection search and one shared vocabulary; every result equals the serial one."""
    import threading
    import voc_cases as vc
    kps, desc, q, qd, taken = _projection_case(oracle, 5, 600, 2.0)
    voc = vc.make(10, 3, 3, irregular=False)
    v = orbfe.ORBVocabulary.from_arrays(10, 3, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    rng = np.random.default_rng(0)
    Q = rng.integers(0, 256, (700, 32), dtype=np.uint8); T = rng.integers(0, 256, (900, 32), dtype=np.uint8)
    ref_knn = orbfe.knn2(Q, T)
    ref_sbp = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, taken, 1, 100, 0.8)
    ref_bow = v.transform(desc, 2)
    errors = []

    def work(kind):
        try:
            for _ in range(25):
                if kind == 0:
                    got = orbfe.knn2(Q, T)
                    assert all(np.array_equal(a, b) for a, b in zip(got, ref_knn))
                elif kind == 1:
                    got = orbfe.search_by_projection(kps, desc, 640, 480, q, qd, taken, 1, 100, 0.8)
                    assert got["nmatches"] == ref_sbp["nmatches"] and np.array_equal(got["match"], ref_sbp["match"])
                else:
                    got = v.transform(desc, 2)
                    assert np.array_equal(got["bow"][1].view(np.uint64), ref_bow["bow"][1].view(np.uint64))
                    assert all(np.array_equal(a, b) for a, b in zip(got["fv"], ref_bow["fv"]))
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    threads = [threading.Thread(target=work, args=(k % 3,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.gpu
def test_distinctive_descriptors(orbfe, oracle):
    """MapPoint::ComputeDistinctiveDescriptors over a whole map at once: bit-exact against the oracle, N from 0 to 256, clusters of
    near-duplicates (ties in the medians), chosen descriptors returned."""
    rng = np.random.default_rng(11)
    sizes = [0, 1, 2, 3, 4, 5, 7, 16, 33, 63, 64, 65, 127, 128, 129, 200, 256] + rng.integers(1, 40, 400).tolist()
    rng.shuffle(sizes)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = np.zeros((offsets[-1], 32), np.uint8)
    for p, n in enumerate(sizes):            # observations of one point: a base descriptor with a few bits flipped
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        blk = np.tile(base, (n, 1))
        flips = rng.integers(0, 256, (n, 3))
        for i in range(n):
            for b in flips[i][:rng.integers(0, 4)]:
                blk[i, b >> 3] ^= np.uint8(1 << (b & 7))
        desc[offsets[p]:offsets[p + 1]] = blk
    want = oracle.distinctive_descriptors(desc, offsets)
    got, chosen = orbfe.distinctive_descriptors(desc, offsets)
    assert np.array_equal(got, want)
    for p, n in enumerate(sizes):
        if n:
            assert np.array_equal(chosen[p], desc[offsets[p] + got[p]])
    assert len(orbfe.distinctive_descriptors(desc[:0], np.zeros(1, np.int32))[0]) == 0
    with pytest.raises(orbfe.OrbfeError):
        orbfe.distinctive_descriptors(np.zeros((300, 32), np.uint8), np.array([0, 300], np.int32))
