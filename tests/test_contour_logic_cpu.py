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
    P.proto_set_reductions.argtypes = [C.c_int, C.c_int]
    P.proto_set_reductions.restype = None
    P.proto_speck_clean.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    P.proto_speck_clean.restype = None

    def make(fn, *extra, cut=0, filt=0, clean=0):
        def run(b, stats=None):
            b = np.ascontiguousarray(b, np.uint8)
            lens = np.zeros(200000, np.int32)
            pts = np.zeros((3000000, 2), np.int32)
            st = np.zeros(8, np.int64)
            P.proto_set_cut(cut)
            P.proto_set_reductions(filt, clean)
            n = fn(b.ctypes.data, b.shape[1], b.shape[0], *extra, lens.ctypes.data, len(lens), pts.ctypes.data,
                   len(pts), st.ctypes.data)
            P.proto_set_cut(0)
            P.proto_set_reductions(0, 0)
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
            "tiled32x20c34": make(P.proto_find_contours_tiled, 5, 20, cut=34),
            # the same with the run tests on the start candidates (aruco_trace.hpp "FEWER WALKS" (1)): the contours do not change
            "trace_f": make(P.proto_find_contours, filt=1), "relay8_f": make(P.proto_find_contours_relay, 3, filt=1),
            "relay32_f": make(P.proto_find_contours_relay, 5, filt=1), "tiled4x3_f": make(P.proto_find_contours_tiled, 2, 3, filt=1),
            "tiled8x2_f": make(P.proto_find_contours_tiled, 3, 2, filt=1), "tiled32x1c40_f": make(P.proto_find_contours_tiled, 5, 1, cut=40, filt=1),
            # ... and on the bit image after the speck passes (2): what the kernels run on
            "trace_fc": make(P.proto_find_contours, filt=1, clean=1), "relay8_fc": make(P.proto_find_contours_relay, 3, filt=1, clean=1),
            "relay32_fc": make(P.proto_find_contours_relay, 5, filt=1, clean=1), "tiled8x2_fc": make(P.proto_find_contours_tiled, 3, 2, filt=1, clean=1),
            "tiled32x20c34_fc": make(P.proto_find_contours_tiled, 5, 20, cut=34, filt=1, clean=1),
            "speck_clean": lambda b: _speck_clean(P, b)}


def _speck_clean(P, b):
    b = np.ascontiguousarray(b, np.uint8)
    out = np.zeros_like(b)
    P.proto_speck_clean(b.ctypes.data, b.shape[1], b.shape[0], out.ctypes.data)
    return out * 255


@pytest.fixture(params=["trace", "relay4", "relay8", "relay32", "tiled4x1", "tiled4x3", "tiled8x2", "tiled16x1", "tiled32x1", "tiled32x20",
                        "tiled4x1c6", "tiled4x3c7", "tiled8x2c10", "tiled32x1c40", "tiled32x20c34",
                        "trace_f", "relay8_f", "relay32_f", "tiled4x3_f", "tiled8x2_f", "tiled32x1c40_f"])
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


# ---------------------------------------------------------------------------------------------------------------------------------
# The two reductions of aruco_trace.hpp "FEWER WALKS".  (1), the run tests on the start candidates, is in the `proto` fixture above
# (the *_f formulations must reproduce the sequential scan contour for contour).  (2), the speck passes, changes the bit image:

KEEP = 70   # the detector keeps borders of MORE than 70 points (markerdetector_impl.cpp:3046, :3217)
SHAPES = ((3, 5), (5, 3))   # ORBFE_SPECK_W1 x H1, W2 x H2


def _speck_pass_definition(fg, w, h):
    """One pass by its definition, pixel by pixel: anchor (ax, ay) = the rim's top-left corner, ax >= 0 in PADDED coordinates, the image
    zero beyond its frame; an empty rim clears the w x h pixels inside."""
    H, W = fg.shape
    pad = max(w, h) + 2
    P = np.zeros((H + 2 + 2 * pad, W + 2 + 2 * pad), bool)      # P[pad + py, pad + px] = padded pixel (px, py)
    P[pad + 1:pad + 1 + H, pad + 1:pad + 1 + W] = fg
    clr = np.zeros_like(P)
    for ay in range(-pad + 1, H + 2):
        for ax in range(0, W + 2):
            y0, x0 = pad + ay, pad + ax
            if P[y0, x0:x0 + w + 2].any() or P[y0 + h + 1, x0:x0 + w + 2].any():
                continue
            if P[y0 + 1:y0 + h + 1, x0].any() or P[y0 + 1:y0 + h + 1, x0 + w + 1].any():
                continue
            clr[y0 + 1:y0 + h + 1, x0 + 1:x0 + w + 1] = True
    out = P & ~clr
    return out[pad + 1:pad + 1 + H, pad + 1:pad + 1 + W]


def _speck_definition(b):
    fg = b != 0
    for w, h in SHAPES:
        fg = _speck_pass_definition(fg, w, h)
    return fg.astype(np.uint8) * 255


def _reduction_images(oracle):
    rng = np.random.default_rng(2024)
    imgs = []
    for d in (0.03, 0.1, 0.17, 0.3, 0.5, 0.8, 0.95):
        for shape in ((37, 53), (48, 64), (33, 31), (40, 97), (64, 32), (21, 130)):
            imgs.append((rng.random(shape) < d).astype(np.uint8) * 255)
    b = np.zeros((40, 70), np.uint8)                       # specks against the frame, in the corners, next to a large region
    b[0, 0] = b[0, 69] = b[39, 0] = b[39, 69] = 255
    b[0:3, 10:13] = 255; b[37:40, 20:25] = 255; b[10:15, 0:3] = 255; b[20:23, 65:70] = 255
    b[8:30, 30:50] = 255; b[12:15, 33:38] = 0; b[20:25, 40:43] = 0; b[13, 35] = 255   # holes in it, an island in a hole
    b[5:7, 52:56] = 255; b[16:21, 53:56] = 255
    imgs.append(b)
    img, _ = synth.scene(240, 320, 3, "ARUCO", 2, side_range=(40, 70))
    imgs.append(oracle.adaptive_threshold(img, 3, 7))
    imgs.append(oracle.adaptive_threshold(synth.stream(120, 200, 2, 1000, "ARUCO", n_markers=1)[1], 3, 7))
    return imgs


def test_speck_passes_match_their_definition(protos, oracle):
    """The word-parallel passes (funnel shifts, rim masks, dilation) against the definition evaluated pixel by pixel."""
    for b in _reduction_images(oracle)[:43]:
        assert np.array_equal(protos["speck_clean"](b), _speck_definition(b)), b.shape


def test_speck_passes_keep_every_long_border(protos, oracle):
    """What the passes clear has no border of more than 68 points, and no other border changes: the borders of more than 68 points of
    the cleaned image ARE those of the image (same points, same order); and something is cleared at all."""
    cleared = 0
    for b in _reduction_images(oracle):
        c = protos["speck_clean"](b)
        assert not np.any((c != 0) & (b == 0))
        cleared += int(np.count_nonzero(b) - np.count_nonzero(c))
        full = oracle.find_contours(b)
        assert _same([x for x in full if len(x) > 68], [x for x in oracle.find_contours(c) if len(x) > 68]), b.shape
    assert cleared > 1000


def test_a_border_has_at_most_four_points_per_pixel(oracle):
    """The bound behind the window sizes: every border, outer or hole, of a component of n pixels has at most 4 n points.  All
    patterns of a 3 x 5 window (2^15), and random blobs of up to 17 pixels grown in 7 x 7."""
    worst = 0
    canvas = np.zeros((7, 9), np.uint8)
    for code in range(1 << 15):
        bits = (code >> np.arange(15)) & 1
        canvas[1:6, 1:4] = bits.reshape(5, 3) * 255
        n = int(bits.sum())
        for c in oracle.find_contours(canvas):
            assert len(c) <= max(1, 4 * n) and len(c) <= 68
            worst = max(worst, len(c))
    rng = np.random.default_rng(5)
    from scipy import ndimage
    for _ in range(3000):
        g = (rng.random((7, 7)) < rng.uniform(0.2, 0.7))
        lab, k = ndimage.label(g, structure=np.ones((3, 3)))
        if not k:
            continue
        comp = lab == 1 + rng.integers(k)
        n = int(comp.sum())
        canvas2 = np.zeros((9, 9), np.uint8)
        canvas2[1:8, 1:8] = comp * 255
        for c in oracle.find_contours(canvas2):
            assert len(c) <= 4 * n, (n, len(c))
    assert worst >= 16


@pytest.mark.parametrize("name", ["trace_fc", "relay8_fc", "relay32_fc", "tiled8x2_fc", "tiled32x20c34_fc"])
def test_reduced_formulations(protos, oracle, name):
    """Run tests + speck passes, as the kernels run them: every formulation equals the sequential scan of the CLEANED image contour
    for contour, hence (previous test) the scan of the image itself on everything the detector keeps."""
    for b in _reduction_images(oracle)[::3]:
        c = protos["speck_clean"](b)
        got = protos[name](b)
        assert _same(oracle.find_contours(c), got), (name, b.shape)
        assert _same([x for x in oracle.find_contours(b) if len(x) > KEEP], [x for x in got if len(x) > KEEP])
