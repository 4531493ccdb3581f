"""Photographs through the front-end (tests/natural_cases.py): the oracle's answers are frozen in tests/golden/natural_images.json
(tests/gen_golden.py natural); the CPU test checks that the oracle still gives them, the GPU tests that the extractor and the
detector give the oracle's keypoints, descriptors and markers bit for bit -- single frames through the host-pointer ABI, and a batch
through the resident pipeline incl. the frame-to-frame matching between two crops of one photograph."""
import json
import os

import numpy as np
import pytest

import natural_cases as N

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "natural_images.json")))
needs_photos = pytest.mark.skipif(not N.available(), reason="scikit-learn's sample photographs are not installed")


@needs_photos
@pytest.mark.parametrize("case", N.CASES, ids=[c[0] for c in N.CASES])
def test_oracle_reproduces_the_frozen_answers(oracle, case):
    img, ids = N.build(case)
    g = GOLD[case[0]]
    assert int(img.astype(np.int64).sum()) == g["image_sum"]            # the same pixels as when the fixture was made
    k, d = oracle.OrbOracle(case[3], 1.2, 8, 20, 7).extract(img)
    m = oracle.ArucoOracle(case[4]).detect(img)
    got = N.digest(k, d, m)
    assert {k_: got[k_] for k_ in ("n", "nm", "kp_sha", "mk_sha")} == {k_: g[k_] for k_ in ("n", "nm", "kp_sha", "mk_sha")}
    assert sorted(int(i) for i in m["id"]) == sorted(ids)                 # every pasted marker is found on the photograph


@needs_photos
@pytest.mark.gpu
@pytest.mark.parametrize("case", N.CASES, ids=[c[0] for c in N.CASES])
def test_gpu_equals_the_oracle_on_photographs(orbfe, oracle, case):
    img, ids = N.build(case)
    ex, ora = orbfe.ORBextractor(case[3], 1.2, 8, 20, 7), oracle.OrbOracle(case[3], 1.2, 8, 20, 7)
    k, d = ex(img)
    ok, od = ora.extract(img)
    assert len(k) == len(ok)
    for f in ("x", "y", "size", "response", "octave"):
        assert np.array_equal(k[f], ok[f]), f
    assert np.allclose(k["angle"], ok["angle"], atol=1e-4) and np.array_equal(d, od)
    for lvl in range(8):                                                    # the pyramid and the FAST candidates of every level too
        assert np.array_equal(ex.level_image(0, lvl), ora.level_image(lvl)), lvl
        assert np.array_equal(ex.level_keypoints(0, lvl, 0), ora.level_keypoints(lvl, 0)), lvl
    det, oa = orbfe.MarkerDetector(case[4]), oracle.ArucoOracle(case[4])
    m, om = det.detect(img), oa.detect(img)
    assert np.array_equal(det.thresholded(0), oa.stage_image(0))
    assert np.array_equal(m["id"], om["id"]) and np.allclose(m["corners"], om["corners"], atol=1e-3)
    assert det.counts(0)["flags"] == 0
    # and the frozen answers, without the oracle in between
    g = GOLD[case[0]]
    got = N.digest(k, d, m)
    assert (got["n"], got["nm"], got["kp_sha"], got["mk_sha"]) == (g["n"], g["nm"], g["kp_sha"], g["mk_sha"])
    # both contour formulations on real edges
    for mode in (True, False):
        det.set_tiled_contours(mode)
        m2 = det.detect(img)
        assert np.array_equal(m2["id"], om["id"]) and np.allclose(m2["corners"], om["corners"], atol=1e-3), mode
    det.set_tiled_contours(None)


@needs_photos
@pytest.mark.gpu
def test_pipeline_on_a_panning_photograph(orbfe, oracle):
    """Twelve 640 x 400 windows sliding over the enlarged photograph (a camera pan: consecutive frames share most of their content,
    unlike the synthetic stream's independent noise) through the resident pipeline: every frame's records and every pair's knn2 /
    SearchForInitialization output against the oracle."""
    import pipeline_check
    from orb_slam2_aruco_amd.pipeline import FrontEndPipeline
    big, _ = N.build(("pan", "china.jpg", (1200, 560), 1000, "ARUCO", 21))
    B, rows, cols = 12, 400, 640
    frames = np.stack([big[20 + 6 * i:20 + 6 * i + rows, 10 + 40 * i:10 + 40 * i + cols] for i in range(B)])
    pipe = FrontEndPipeline(B, rows, cols, 1000, 8, "ARUCO")
    d = pipe.upload(frames)
    pipe.warmup(d, 1)
    cur = pipe.step(d)
    rec, matches = pipe.read_records(cur), pipe.read_matches()
    res = pipeline_check.check_against_oracle(oracle, frames, list(range(B)), rec, matches, 1000, 8, "ARUCO", cols, rows, pipe.cam_K, pipe.cam_D,
                                              pairs=list(range(B - 1)), prev_last=frames[B - 1])
    assert res["pairs_checked"] == B - 1 and res["boundary_pair_checked"] and res["keypoints_checked"] > 900 * B
    assert int(matches["nmatches"][1:].min()) > 100          # a pan: most keypoints find their match in the next frame
