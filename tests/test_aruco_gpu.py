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
