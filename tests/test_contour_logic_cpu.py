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
    return {"trace": make(P.proto_find_contours), "relay4": make(P.proto_find_contours_relay, 2),
            "relay8": make(P.proto_find_contours_relay, 3), "relay32": make(P.proto_find_contours_relay, 5)}


@pytest.fixture(params=["trace", "relay4", "relay8", "relay32"])
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
