"""Checker shared by bench.py's post-run verification and the GPU tests of the batched pipeline (test infrastructure: this
is the only place where the pipeline's outputs meet the oracle; nothing under orb_slam2_aruco_amd/ imports it)."""
import numpy as np

MARKER_SIZE = 0.187  # Frame.cc:131


def check_against_oracle(oracle, frames_u8, frame_ids, rec, matches, nfeatures, nlevels, dictionary, cols, rows, cam_K, cam_D,
                         use_orb=True, use_aruco=True, pairs=None, prev_last=None):
    """The checker bench.py and the GPU tests share: records of frames `frame_ids` (indices into the batch whose host
    images are frames_u8) and matching results of `pairs` (p = frame p vs p + 1) against the CPU oracle.  Bit-exact for
    keypoints (angle: 1e-4, north_star), descriptors, marker ids, knn2 and SearchForInitialization outputs; marker
    corners 1e-3 px; poses 1e-4 relative.  Returns a summary dict; raises AssertionError on any difference.

    The pipeline's matching rows carry one pair in front of the batch (FrontEndPipeline.read_matches): row 0 = the last frame of
    the stream's previous batch against frame 0, row p + 1 = frame p against p + 1.  prev_last = the host image of that previous
    frame: the pair across the batch boundary is then checked too, together with the halo slot of the record set."""
    orb = oracle.OrbOracle(nfeatures, 1.2, nlevels, 20, 7) if use_orb else None
    aru = oracle.ArucoOracle(dictionary) if use_aruco else None
    ext = {}

    def extract(f):
        if f not in ext:
            ext[f] = orb.extract(frames_u8[f])
        return ext[f]

    nk = nmk = 0
    for f in frame_ids:
        if use_orb:
            k, d = extract(f)
            n = int(rec["n"][f])
            assert n == len(k), ("keypoint count", f, n, len(k))
            g = rec["kps"][f, :n]
            for fld in ("x", "y", "size", "response", "octave"):
                assert np.array_equal(g[fld], k[fld]), ("keypoint " + fld, f)
            assert np.allclose(g["angle"], k["angle"], atol=1e-4), ("keypoint angle", f)
            assert np.array_equal(rec["desc"][f, :n], d), ("descriptors", f)
            nk += n
        if use_aruco:
            want = aru.detect(frames_u8[f])
            m = int(rec["nmk"][f])
            assert m == len(want), ("marker count", f, m, len(want))
            assert np.array_equal(rec["markers"][f, :m]["id"], want["id"]), ("marker ids", f)
            assert np.allclose(rec["markers"][f, :m]["corners"], want["corners"], atol=1e-3), ("marker corners", f)
            for j, w in enumerate(want):
                r1, t1, _, _, _ = oracle.marker_pose(w["corners"], MARKER_SIZE, cam_K, cam_D)
                p = rec["poses"][f, j]
                assert np.allclose(p["rvec"], r1, rtol=1e-4, atol=1e-5) and np.allclose(p["tvec"], t1, rtol=1e-4, atol=1e-5), \
                    ("marker pose", f, j)
            nmk += m
    npairs = 0
    boundary = False
    if use_orb and matches is not None:
        def check_pair(row, a, b, what):
            (k1, d1), (k2, d2) = a, b
            bi, bd, sd = oracle.knn2(d1, d2, 256)
            n1 = len(k1)
            assert np.array_equal(matches["best_idx"][row, :n1], bi), ("knn2 best_idx", what)
            assert np.array_equal(matches["best_dist"][row, :n1], bd), ("knn2 best_dist", what)
            assert np.array_equal(matches["second_dist"][row, :n1], sd), ("knn2 second_dist", what)
            wn, wm, _ = oracle.search_for_initialization(k1, d1, k2, d2, cols, rows, None, 100, 0.9, True)
            assert int(matches["nmatches"][row]) == wn, ("SearchForInitialization nmatches", what, int(matches["nmatches"][row]), wn)
            assert np.array_equal(matches["matches12"][row, :n1], wm), ("SearchForInitialization matches12", what)
        for p in (pairs or []):
            check_pair(p + 1, extract(p), extract(p + 1), p)
            npairs += 1
        if prev_last is not None:   # the pair across the batch boundary: the previous batch's last frame (the halo slot) against frame 0
            kp, dp = orb.extract(prev_last)
            n = int(rec["halo_n"][0])
            assert n == len(kp) and np.array_equal(rec["halo_desc"][0, :n], dp), ("halo slot", n, len(kp))
            for fld in ("x", "y", "size", "response", "octave"):
                assert np.array_equal(rec["halo_kps"][0, :n][fld], kp[fld]), ("halo keypoint " + fld)
            check_pair(0, (kp, dp), extract(0), "previous batch's last frame -> 0")
            boundary = True
    return {"frames": [int(f) for f in frame_ids], "pairs": [int(p) for p in (pairs or [])] if use_orb and matches is not None else [],
            "keypoints_checked": nk, "markers_checked": nmk, "pairs_checked": npairs, "boundary_pair_checked": boundary}
