#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (committed, small).

The reference ships no golden vectors and cannot be compiled here, so these fixtures are produced by
the oracle on seeded synthetic inputs; they freeze the oracle's behaviour (any later edit of oracle/ that changes
results is caught by tests/test_oracle_cpu.py) and give the GPU tests data that does not need the oracle at all.

    python tests/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from orb_slam2_aruco_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def modes():
    """The detector's other modes (oracle/aruco_oracle.cpp, csrc/aruco_modes.hip): one ten-frame sequence per configuration through ONE
    detector behind srand(5) -- ids, corners, Params::ThresHold, threshold passes, working sizes, minSize and tracked markers per frame.
    The frames are regenerated from their seeds by tests/test_aruco_modes_gpu.py::golden_sequence (checksum stored)."""
    import ctypes
    import test_aruco_modes_gpu as T
    libc = ctypes.CDLL(None)
    seq = T.golden_sequence()
    out = {"frames_sum": np.array([int(sum(int(f.astype(np.int64).sum()) for f in seq))])}
    for name, (mode, ms, corner, enclosed, track) in T.GOLDEN_CONFIGS.items():
        o = O.ArucoOracle("ARUCO")
        o.detect_enclosed_markers(enclosed); o.set_corner_method(corner); o.set_detection_mode(mode, ms); o.set_tracking(track)
        libc.srand(5)
        ids, corners, state = [], [], []
        for im in seq:
            m = o.detect(im)
            st = o.state()
            ids.append(np.pad(m["id"], (0, 8 - len(m)), constant_values=-1)[:8])
            c = np.zeros((8, 4, 2), np.float32); c[:min(len(m), 8)] = m["corners"][:8]; corners.append(c)
            state.append([st["threshold"], st["attempts"], st["work_shape"][0], st["work_shape"][1], o.tracked(), len(m)])
        out[name + "_ids"] = np.array(ids, np.int32); out[name + "_corners"] = np.array(corners)
        out[name + "_state"] = np.array(state, np.int32)
    np.savez_compressed(os.path.join(OUT, "aruco_modes_seq.npz"), **out)
    print("wrote aruco_modes_seq.npz")


def natural():
    """Photographs (tests/natural_cases.py): the oracle's keypoints / descriptors / markers per case as counts and SHA-256."""
    import json
    import natural_cases as N
    out = {}
    for case in N.CASES:
        img, ids = N.build(case)
        k, d = O.OrbOracle(case[3], 1.2, 8, 20, 7).extract(img)
        m = O.ArucoOracle(case[4]).detect(img)
        out[case[0]] = dict(N.digest(k, d, m), image_sum=int(img.astype(np.int64).sum()), pasted_ids=sorted(int(i) for i in ids),
                            detected_ids=[int(i) for i in m["id"]])
        print(case[0], out[case[0]])
    json.dump(out, open(os.path.join(OUT, "natural_images.json"), "w"), indent=1, sort_keys=True)
    print("wrote natural_images.json")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "natural":
        os.makedirs(OUT, exist_ok=True)
        return natural()
    if len(sys.argv) > 1 and sys.argv[1] == "modes":   # only the fixture of the detector modes (the others stay byte for byte)
        os.makedirs(OUT, exist_ok=True)
        return modes()
    os.makedirs(OUT, exist_ok=True)
    # ORB: 240x320 crop-sized scene (image stored, ~77 KB) + 640x480 scene (only the seed is stored)
    img, _ = synth.scene(240, 320, 7, "ARUCO", 2, side_range=(40, 70))
    k, d = O.OrbOracle(500, 1.2, 4, 20, 7).extract(img)
    np.savez_compressed(os.path.join(OUT, "orb_240x320.npz"), image=img, kps=k, desc=d,
                        params=np.array([500, 4, 20, 7]))
    img, _ = synth.scene(480, 640, 1, "ARUCO", 4)
    o = O.OrbOracle(1000, 1.2, 8, 20, 7)
    k, d = o.extract(img)
    np.savez_compressed(os.path.join(OUT, "orb_640x480_seed1.npz"), kps=k, desc=d,
                        image_sum=np.array([int(img.astype(np.int64).sum()), int((img.astype(np.int64) * np.arange(img.size).reshape(img.shape) % 65521).sum())]),
                        ncand=np.array([len(o.level_keypoints(l, 0)) for l in range(8)]),
                        nkept=np.array([len(o.level_keypoints(l, 1)) for l in range(8)]))
    # ArUco
    for name, (h, w, seed, dic, K) in {"aruco_640x480_seed1": (480, 640, 1, "ARUCO", 4),
                                       "aruco_540x960_seed6": (540, 960, 6, "ARUCO_MIP_36h12", 5)}.items():
        img, truth = synth.scene(h, w, seed, dic, K)
        a = O.ArucoOracle(dic)
        m = a.detect(img)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), markers=m, rects=a.candidates(0),
                            thr_on=np.array([int((a.stage_image(0) > 0).sum())]),
                            truth_ids=np.array(sorted(t[0] for t in truth)))
    # matching
    s = synth.stream(480, 640, 2, 1000)
    o = O.OrbOracle(1000, 1.2, 8, 20, 7)
    (k1, d1), (k2, d2) = o.extract(s[0]), o.extract(s[1])
    bi, bd, sd = O.knn2(d1, d2, 256)
    n, m12, prev = O.search_for_initialization(k1, d1, k2, d2, 640, 480, None, 100, 0.9, True)
    np.savez_compressed(os.path.join(OUT, "match_stream1000.npz"), k1=k1, d1=d1, k2=k2, d2=d2, best_idx=bi,
                        best_dist=bd, second_dist=sd, nmatches=np.array([n]), matches12=m12, prev=prev)
    # SearchByProjection matching loop: map points = the keypoints of frame 1 "projected" into frame 2 at their matched
    # position (or their own position), predicted octave = their octave; descriptors = frame 1's
    rng = np.random.default_rng(77)
    sel = rng.permutation(len(k1))[:600]
    q = np.zeros(len(sel), O.WINDOW_QUERY_DTYPE)
    q["x"] = np.where(m12[sel] >= 0, k2["x"][np.maximum(m12[sel], 0)], k1["x"][sel]) + rng.normal(0, 1.0, len(sel)).astype(np.float32)
    q["y"] = np.where(m12[sel] >= 0, k2["y"][np.maximum(m12[sel], 0)], k1["y"][sel]) + rng.normal(0, 1.0, len(sel)).astype(np.float32)
    q["r"] = (4.0 * 1.2 ** k1["octave"][sel]).astype(np.float32)
    q["min_level"] = k1["octave"][sel] - 1
    q["max_level"] = k1["octave"][sel]
    taken = (rng.random(len(k2)) < 0.1).astype(np.uint8)
    r0 = O.search_by_projection(k2, d2, 640, 480, q, d1[sel], taken, 0, 100, 0.8)
    r1 = O.search_by_projection(k2, d2, 640, 480, q, d1[sel], taken, 1, 100, 0.8)
    np.savez_compressed(os.path.join(OUT, "projection_stream1000.npz"), sel=sel.astype(np.int32), queries=q, taken=taken,
                        best_idx=r0["best_idx"], best_dist=r0["best_dist"], best_level=r0["best_level"],
                        second_dist=r0["second_dist"], second_level=r0["second_level"], match=r1["match"],
                        nmatches=np.array([r1["nmatches"]]), taken_after=r1["taken"])
    # the rows built after the first three: pose, undistortion, vocabulary transform, SearchByBoW, last-frame projection
    # search, on-disk keyframe records -- inputs are the seeded generators in tests/ and match_stream1000's two frames
    import hashlib
    import pose_cases as pc
    import voc_cases as vc
    cases = pc.random_cases(30, 123, 0.187, 0.2)
    corners = np.stack([c for _, _, c in cases])
    poses = np.zeros((len(cases), 12)); errs = np.zeros((len(cases), 2), np.float32)
    for i, c in enumerate(corners):
        r1, t1, r2, t2, e = O.marker_pose(c, 0.187, pc.K4, pc.DIST)
        poses[i] = np.concatenate([r1, t1, r2, t2]); errs[i] = e
    pts = np.random.default_rng(5).uniform([0, 0], [640, 480], (200, 2)).astype(np.float32)
    und = O.undistort_points(pts, pc.K4, pc.DIST)
    bnd = O.compute_image_bounds(640, 480, pc.K4, pc.DIST)
    voc = vc.make(10, 4, 41, irregular=False)
    ov = O.VocabularyOracle.from_arrays(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    t1, t2 = ov.transform(d1, 2), ov.transform(d2, 2)
    valid1 = (np.random.default_rng(6).random(len(k1)) < 0.8).astype(np.uint8)
    nb, b12, b21 = O.search_by_bow(k1, d1, t1["fv"], k2, d2, t2["fv"], valid1, None, 0.7, True, 50, 30 / 360.0)
    rng = np.random.default_rng(8)
    z = rng.uniform(1.0, 6.0, len(k1)).astype(np.float32)
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32)
    x3 = np.stack([(k1["x"] - K4[2]) / K4[0] * z, (k1["y"] - K4[3]) / K4[1] * z, z], 1).astype(np.float32)
    Tcw = np.array([[1, -0.002, 0.001, 0.004], [0.002, 1, -0.003, -0.006], [-0.001, 0.003, 1, 0.01]], np.float32)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    nl, ml = O.search_by_projection_last_frame(k2, d2, 640, 480, k1, valid1, x3, d1, Tcw, K4, sf, 15.0)
    rec = O.keyframe_features_pack(k1, d1, np.arange(len(k1), dtype=np.uint64))
    np.savez_compressed(os.path.join(OUT, "extras_stream1000.npz"), corners=corners, poses=poses, pose_err=errs, pts=pts, undistorted=und,
                        bounds=bnd, word1=t1["word"], node1=t1["node"], bow1_words=t1["bow"][0], bow1_values=t1["bow"][1],
                        fv1_nodes=t1["fv"][0], fv1_offsets=t1["fv"][1], fv1_features=t1["fv"][2], fv2_nodes=t2["fv"][0],
                        fv2_offsets=t2["fv"][1], fv2_features=t2["fv"][2], valid1=valid1, bow_nmatches=np.array([nb]), bow_match12=b12,
                        x3Dw=x3, Tcw=Tcw, last_nmatches=np.array([nl]), last_match_cur=ml,
                        kf_sha256=np.frombuffer(hashlib.sha256(rec.tobytes()).digest(), np.uint8))
    modes()
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
