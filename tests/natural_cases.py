"""Photographs through the front-end (every input so far came from orb_slam2_aruco_amd/synth.py -- box-blurred noise,
rectangles, pasted markers).  The reference is run on camera frames (Examples/Monocular/mono_cvcam.cc:110,128: cv::VideoCapture,
resized to 960 x 540); the two photographs scikit-learn ships (sklearn/datasets/images/{china,flower}.jpg, present in the build
container and on the GPU box alike) stand in: large smooth regions (sky, petals), real edges, foliage texture, JPEG blur -- where the
minThFAST retry, sparse quadtrees and long thin contours live.  TEST INFRASTRUCTURE: builds the cases, shared by
tests/gen_golden.py (freezes the oracle's answers as hashes), tests/test_oracle_cpu.py-style CPU check and tests/test_natural_gpu.py."""
import hashlib

import numpy as np

# name, photograph, output size (cols, rows), nfeatures, dictionary, marker seed
CASES = [("china_640x427", "china.jpg", (640, 427), 1000, "ARUCO", 11),
         ("flower_640x427", "flower.jpg", (640, 427), 1000, "ARUCO_MIP_36h12", 12),
         ("china_960x540", "china.jpg", (960, 540), 1500, "ARUCO", 13),
         ("flower_960x540", "flower.jpg", (960, 540), 1500, "ARUCO_MIP_25h7", 14)]


def available():
    try:
        from sklearn.datasets import load_sample_image
        load_sample_image("china.jpg")
        return True
    except Exception:
        return False


def grey(rgb):
    """ITU-R 601 luma in integers (what cvtColor(RGB2GRAY) computes with 14 fractional bits)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def build(case):
    """-> uint8 grey image of the case's size with its markers pasted in."""
    from sklearn.datasets import load_sample_image
    from orb_slam2_aruco_amd import synth
    name, photo, (cols, rows), _, dic, seed = case
    g = grey(load_sample_image(photo))
    if cols > g.shape[1] or rows > g.shape[0]:      # twice the size (pixel doubling + a 2 x 2 mean: a mildly blurred enlargement), centre crop
        up = np.kron(g, np.ones((2, 2), np.uint8)).astype(np.int64)
        up = ((up[:-1, :-1] + up[1:, :-1] + up[:-1, 1:] + up[1:, 1:] + 2) >> 2).astype(np.uint8)
        y0, x0 = (up.shape[0] - rows) // 2, (up.shape[1] - cols) // 2
        g = up[y0:y0 + rows, x0:x0 + cols]
    else:
        g = g[:rows, :cols]
    img, truth = synth.paste_markers(np.ascontiguousarray(g), seed, dic, n_markers=3, side_range=(50, 110))
    return img, [t[0] for t in truth]


def digest(kps, desc, markers):
    """What is frozen per case: counts and SHA-256 of keypoints (x, y, octave, response as the oracle returns them), descriptors,
    marker ids and corners (rounded to 1e-2 px)."""
    h = hashlib.sha256()
    for f in ("x", "y", "octave", "response", "size"):
        h.update(np.ascontiguousarray(kps[f]).tobytes())
    h.update(np.ascontiguousarray(desc).tobytes())
    hm = hashlib.sha256()
    hm.update(np.ascontiguousarray(markers["id"]).tobytes())
    hm.update(np.round(np.asarray(markers["corners"], np.float64) * 100).astype(np.int64).tobytes())
    return dict(n=len(kps), nm=len(markers), kp_sha=h.hexdigest(), mk_sha=hm.hexdigest())
