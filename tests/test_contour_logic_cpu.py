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

    P.proto_set_cut.argtypes = [C.c_int]
    P.proto_set_cut.restype = None

    def make(fn, *extra, cut=0):
        def run(b, stats=None):
            b = np.ascontiguousarray(b, np.uint8)
            lens = np.zeros(200000, np.int32)
            pts = np.zeros((3000000, 2), np.int32)
            st = np.zeros(8, np.int64)
            P.proto_set_cut(cut)
            n = fn(b.ctypes.data, b.shape[1], b.shape[0], *extra, lens.ctypes.data, len(lens), pts.ctypes.data,
                   len(pts), st.ctypes.data)
            P.proto_set_cut(0)
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
            "tiled32x1": make(P.proto_find_contours_tiled, 5, 1), "tiled32x20": make(P.proto_find_contours_tiled, 5, 20),
            # the same with recorded segments cut into pieces of at most `cut` states (> K + 1; the kernels: 40 at K = 32)
            "tiled4x1c6": make(P.proto_find_contours_tiled, 2, 1, cut=6), "tiled4x3c7": make(P.proto_find_contours_tiled, 2, 3, cut=7),
            "tiled8x2c10": make(P.proto_find_contours_tiled, 3, 2, cut=10), "tiled32x1c40": make(P.proto_find_contours_tiled, 5, 1, cut=40),
            "tiled32x20c34": make(P.proto_find_contours_tiled, 5, 20, cut=34)}


@pytest.fixture(params=["trace", "relay4", "relay8", "relay32", "tiled4x1", "tiled4x3", "tiled8x2", "tiled16x1", "tiled32x1", "tiled32x20",
                        "tiled4x1c6", "tiled4x3c7", "tiled8x2c10", "tiled32x1c40", "tiled32x20c34"])
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


def test_cut_segments_have_one_owner(protos, oracle):
    """A recorded segment is cut into pieces of at most `cut` states (the kernels keep a piece's directions in four registers: 40
    states at K = 32).  The piece after a cut starts at a state that is no marker, so it belongs to the tile that walked up to it,
    whatever grid line it runs on: combs and serpents inside one cell and across shared lines, random textures -- the prototype
    reports a piece owned twice, by nobody, or a walk that leaves its tile after a cut; the contours equal the sequential scan's
    and pieces were cut at all (stats[6])."""
    st = np.zeros(8, np.int64)
    rng = np.random.default_rng(7)
    total_cut = 0
    for name, K in (("tiled4x1c6", 4), ("tiled4x3c7", 4), ("tiled8x2c10", 8), ("tiled32x1c40", 32), ("tiled32x20c34", 32)):
        imgs = []
        b = np.zeros((4 * K + 3, 5 * K + 2), np.uint8)               # a comb inside one cell with a stem across a grid column and a grid row
        for x in range(K + 1, 2 * K - 1, 2):
            b[K + 1:2 * K - 2, x] = 255
        b[2 * K - 3, K - 3:2 * K - 1] = 255
        b[2 * K - 3:2 * K + 2, K + 1] = 255
        imgs.append(b)
        b = np.zeros((3 * K + 1, 4 * K + 1), np.uint8)               # a serpent that runs along both sides of a shared grid row
        for x in range(2, 4 * K - 2, 4):
            b[K - 3:K + 2, x] = 255
            b[K - 3 if (x // 4) % 2 else K + 1, x:x + 5] = 255
        imgs.append(b)
        for d in (0.35, 0.5, 0.65):
            imgs.append((rng.random((3 * K + 5, 4 * K + 3)) < d).astype(np.uint8) * 255)
        if K == 32:
            img, _ = synth.scene(240, 320, 5, "ARUCO", 2, side_range=(40, 70))
            imgs.append(oracle.adaptive_threshold(img, 3, 7))
        for b in imgs:
            assert _same(oracle.find_contours(b), protos[name](b, st)), (name, b.shape)
            total_cut += int(st[6])
    assert total_cut > 20
