"""GPU vs the committed golden vectors only -- no oracle involved (tests/golden/extras_stream1000.npz, match_stream1000.npz;
the generating script is tests/gen_golden.py).  Integer / byte / index results bit-exact, doubles of the BowVector bit-exact,
pose within 1e-5 relative."""
import hashlib
import os

import numpy as np
import pytest

import pose_cases as pc
import voc_cases as vc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load():
    return np.load(os.path.join(GOLD, "extras_stream1000.npz")), np.load(os.path.join(GOLD, "match_stream1000.npz"))


def test_pose_and_undistortion(orbfe):
    g, _ = _load()
    mk = np.zeros(len(g["corners"]), orbfe.MARKER_DTYPE); mk["corners"] = g["corners"]
    p = orbfe.marker_poses(mk, 0.187, pc.K4, pc.DIST)
    got = np.concatenate([p["rvec"], p["tvec"], p["rvec2"], p["tvec2"]], 1)
    assert np.allclose(got, g["poses"], rtol=1e-5, atol=1e-6) and np.allclose(p["err"], g["pose_err"], atol=1e-3)
    assert np.array_equal(orbfe.undistort_points(g["pts"], pc.K4, pc.DIST), g["undistorted"])
    assert np.array_equal(orbfe.ComputeImageBounds(640, 480, pc.K4, pc.DIST), g["bounds"])


def test_vocabulary_and_search_by_bow(orbfe):
    g, m = _load()
    voc = vc.make(10, 4, 41, irregular=False)
    v = orbfe.ORBVocabulary.from_arrays(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    t1, t2 = v.transform(m["d1"], 2), v.transform(m["d2"], 2)
    assert np.array_equal(t1["word"], g["word1"]) and np.array_equal(t1["node"], g["node1"])
    assert np.array_equal(t1["bow"][0], g["bow1_words"])
    assert np.array_equal(t1["bow"][1].view(np.uint64), g["bow1_values"].view(np.uint64))
    assert all(np.array_equal(a, g[k]) for a, k in zip(t1["fv"], ("fv1_nodes", "fv1_offsets", "fv1_features")))
    assert all(np.array_equal(a, g[k]) for a, k in zip(t2["fv"], ("fv2_nodes", "fv2_offsets", "fv2_features")))
    nb, b12, b21 = orbfe.search_by_bow(m["k1"], m["d1"], t1["fv"], m["k2"], m["d2"], t2["fv"], g["valid1"], None, 0.7, True, 50, 30 / 360.0)
    assert nb == int(g["bow_nmatches"][0]) and np.array_equal(b12, g["bow_match12"])


def test_last_frame_projection_and_keyframe_records(orbfe):
    g, m = _load()
    K4 = np.array([517.3, 516.5, 318.6, 255.3], np.float32); sf = np.array([1.2 ** i for i in range(8)], np.float32)
    nl, ml = orbfe.search_by_projection_last_frame(m["k2"], m["d2"], 640, 480, m["k1"], g["valid1"], g["x3Dw"], m["d1"], g["Tcw"], K4, sf, 15.0)
    assert nl == int(g["last_nmatches"][0]) and np.array_equal(ml, g["last_match_cur"])
    rec = orbfe.keyframe_features_pack(m["k1"], m["d1"], np.arange(len(m["k1"]), dtype=np.uint64))
    assert np.array_equal(np.frombuffer(hashlib.sha256(rec.tobytes()).digest(), np.uint8), g["kf_sha256"])
