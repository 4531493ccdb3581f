"""CPU test of the border-following logic shared with the gfx950 kernel (csrc/aruco_trace.hpp): independent
read-only traces from candidate starts must reproduce the oracle's sequential Suzuki-Abe scan exactly."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from orb_slam2_aruco_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def protos(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("proto") / "libproto.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests", "proto_contours.cpp")])
    P = C.CDLL(so)
    P.proto_find_contours.restype = C.c_int
    P.proto_find_contours.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]

    P.proto_find_contours_relay.restype = C.c_int
    P.proto_find_contours_relay.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p]

    P.proto_find_contours_tiled.restype = C.c_int
    P.proto_find_contours_tiled.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_int, C.c_void_p]

    def make(fn, *extra):
        def run(b, stats=None):
            b = np.ascontiguousarray(b, np.uint8)
            lens = np.zeros(200000, np.int32)
            pts = np.zeros((3000000, 2), np.int32)
            st = np.zeros(8, np.int64)
            n = fn(b.ctypes.data, b.shape[1], b.shape[0], *extra, lens.ctypes.data, len(lens), pts.ctypes.data,
                   len(pts), st.ctypes.data)
            assert n >= 0, n
            if stats is not None:
                stats[:] = st
            out, o = [], 0
            for l in lens[:n]:
                out.append(pts[o:o + l].copy())
                o += l
            return out
        return run
    # relay spacing K = 4, 8, 32: small K puts many grid markers on small test images (multi-segment borders),
    # K = 32 is what the kernel uses (most borders of these images then take the small-border path)
    # tiled: (grid spacing, tile width in cells): many tiles per test image at K = 4 / 8 (segments on shared grid lines, start
    # candidates on shared columns), one band x several column tiles and whole-width tiles at K = 16 / 32
    return {"trace": make(P.proto_find_contours), "relay4": make(P.proto_find_contours_relay, 2),
            "relay8": make(P.proto_find_contours_relay, 3), "relay32": make(P.proto_find_contours_relay, 5),
            "tiled4x1": make(P.proto_find_contours_tiled, 2, 1), "tiled4x3": make(P.proto_find_contours_tiled, 2, 3),
            "tiled8x2": make(P.proto_find_contours_tiled, 3, 2), "tiled16x1": make(P.proto_find_contours_tiled, 4, 1),
            "tiled32x1": make(P.proto_find_contours_tiled, 5, 1), "tiled32x20": make(P.proto_find_contours_tiled, 5, 20)}


@pytest.fixture(params=["trace", "relay4", "relay8", "relay32", "tiled4x1", "tiled4x3", "tiled8x2", "tiled16x1", "tiled32x1", "tiled32x20"])
def proto(request, protos):
    return protos[request.param]


def _same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("density", [0.05, 0.3, 0.5, 0.6, 0.75, 0.95])
def test_random_images(proto, oracle, density):
    rng = np.random.default_rng(int(density * 100))
    for k in range(5):
        b = (rng.random((37 + 11 * k, 53 + 7 * k)) < density).astype(np.uint8) * 255
        assert _same(oracle.find_contours(b), proto(b))


def test_structured_images(proto, oracle):
    b = np.zeros((30, 30), np.uint8)
    b[5:25, 5:25] = 255; b[10:20, 10:20] = 0; b[12:18, 12:18] = 255; b[14, 14] = 0
    assert _same(oracle.find_contours(b), proto(b))
    for fill in (0, 255):
        b = np.full((20, 20), fill, np.uint8)
        assert _same(oracle.find_contours(b), proto(b))
    b = np.zeros((12, 40), np.uint8); b[::2, ::2] = 255         # isolated pixels
    assert _same(oracle.find_contours(b), proto(b))
    b = np.zeros((40, 40), np.uint8); b[np.arange(40), np.arange(40)] = 255   # 8-connected diagonal
    assert _same(oracle.find_contours(b), proto(b))


def test_thresholded_scene(proto, oracle):
    img, _ = synth.scene(240, 320, 3, "ARUCO", 2, side_range=(40, 70))
    b = oracle.adaptive_threshold(img, 3, 7)
    assert _same(oracle.find_contours(b), proto(b))


def test_tiles_see_shared_grid_lines_once(protos, oracle):
    """Lines and blobs ON the grid lines between tiles (segments that lie entirely on a shared row / column, corners on grid
    crossings, sizes that are exact multiples of the spacing): every marker state has exactly one owner (the prototype returns
    an error otherwise) and the contours equal the sequential scan's."""
    st = np.zeros(8, np.int64)
    for K, name in ((4, "tiled4x1"), (4, "tiled4x3"), (8, "tiled8x2")):
        for H, W in ((4 * K, 6 * K), (4 * K + 1, 6 * K - 1), (3 * K - 1, 5 * K + 3)):
            b = np.zeros((H, W), np.uint8)
            # padded coordinate = image coordinate + 1: image row K - 1 is the relay row K
            b[K - 1, 1:W - 1] = 255                      # a line along a relay row, across several tiles
            b[2:H - 1, 2 * K - 1] = 255                  # a line along a relay column
            b[2 * K - 1:2 * K + 1, 3 * K - 1:3 * K + 1] = 255   # a 2 x 2 blob on a grid crossing
            b[K + 1:2 * K - 2, K + 1:2 * K - 2] = 255    # a blob strictly inside a cell (small border)
            b[2 * K:3 * K - 1, 4 * K - 1] = 255          # a piece that starts on a crossing and runs down a column
            assert _same(oracle.find_contours(b), protos[name](b, st)), (K, name, H, W)
            assert st[3] > 0        # segments on shared lines were met by two tiles and skipped by one
    rng = np.random.default_rng(77)
    for k in range(6):                                    # images whose sizes are multiples of the spacing, dense and sparse
        b = (rng.random((32, 48)) < (0.2 + 0.12 * k)).astype(np.uint8) * 255
        for name in ("tiled4x1", "tiled4x3", "tiled8x2", "tiled16x1"):
            assert _same(oracle.find_contours(b), protos[name](b)), (k, name)
