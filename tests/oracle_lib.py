"""ctypes binding of oracle/liborbfe_oracle.so (the CPU restatement).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liborbfe_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

MARKER_DTYPE = np.dtype([("id", "<i4"), ("corners", "<f4", (4, 2))])

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO) or any(
                os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(ORACLE_SO)
                for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h"))):
            build()
        L = C.CDLL(ORACLE_SO)
        L.oracle_orb_create.restype = C.c_void_p
        L.oracle_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.oracle_orb_destroy.argtypes = [C.c_void_p]
        L.oracle_orb_set_trig_libm.argtypes = [C.c_void_p, C.c_int]
        L.oracle_orb_set_gaussian_taps.argtypes = [C.c_void_p, C.c_int]
        L.oracle_gaussian7_taps_mode.argtypes = [C.c_void_p, C.c_int]
        L.oracle_orb_extract.restype = C.c_int
        L.oracle_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p,
                                         C.c_void_p, C.c_int]
        L.oracle_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.oracle_orb_level_size.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_orb_level_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_orb_level_keypoints.restype = C.c_int
        L.oracle_orb_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.oracle_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_gaussian7_taps.argtypes = [C.c_void_p]
        L.oracle_fast_score_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_fast_detect.restype = C.c_int
        L.oracle_fast_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.oracle_distribute.restype = C.c_int
        L.oracle_distribute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int]
        L.oracle_fast_atan2.restype = C.c_float
        L.oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_sincosf.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.oracle_cv_round.restype = C.c_int
        L.oracle_cv_round.argtypes = [C.c_double]
        L.oracle_descriptor_distance.restype = C.c_int
        L.oracle_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_knn2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
        L.oracle_features_in_area.restype = C.c_int
        L.oracle_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_knn2_csr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.oracle_search_for_initialization.restype = C.c_int
        L.oracle_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                       C.c_float, C.c_int, C.c_void_p]
        L.oracle_three_maxima.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "oracle_aruco_create"):
            _bind_aruco(L)
        _lib = L
    return _lib


def _bind_aruco(L):
    L.oracle_aruco_create.restype = C.c_void_p
    L.oracle_aruco_create.argtypes = [C.c_char_p]
    L.oracle_aruco_destroy.argtypes = [C.c_void_p]
    L.oracle_aruco_detect.restype = C.c_int
    L.oracle_aruco_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int]
    L.oracle_aruco_stage_image.restype = C.c_int
    L.oracle_aruco_stage_image.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_aruco_stage_count.restype = C.c_int
    L.oracle_aruco_stage_count.argtypes = [C.c_void_p, C.c_int]
    L.oracle_aruco_candidates.restype = C.c_int
    L.oracle_aruco_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.oracle_aruco_set_detection_mode.argtypes = [C.c_void_p, C.c_int, C.c_float]
    L.oracle_aruco_set_corner_method.argtypes = [C.c_void_p, C.c_int]
    L.oracle_aruco_set_enclosed.argtypes = [C.c_void_p, C.c_int]
    L.oracle_aruco_set_tracking.argtypes = [C.c_void_p, C.c_int]
    L.oracle_aruco_state.argtypes = [C.c_void_p, C.c_int]
    L.oracle_aruco_min_size.restype = C.c_float
    L.oracle_aruco_min_size.argtypes = [C.c_void_p]
    L.oracle_bgr_to_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int]
    L.oracle_resize_nearest.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.oracle_corner_subpix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double]
    L.oracle_otsu_of_histogram.argtypes = [C.c_void_p]
    L.oracle_adaptive_threshold.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.oracle_find_contours.restype = C.c_int
    L.oracle_find_contours.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.oracle_warp_perspective35.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.oracle_otsu_threshold.restype = C.c_int
    L.oracle_otsu_threshold.argtypes = [C.c_void_p, C.c_int]
    L.oracle_decode_marker.restype = C.c_int
    L.oracle_decode_marker.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    def __init__(self, nfeatures=1000, scale=1.2, nlevels=8, ini=20, mn=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.oracle_orb_create(nfeatures, scale, nlevels, ini, mn)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_orb_destroy(self.h)
            self.h = None

    def set_gaussian_taps(self, mode):
        """0: 18 34 49 55 49 34 18 (OpenCV 2.4 / 3.2 / early 3.4); 1: 18 34 48 56 48 34 18 (late 3.4.x / 4.x)."""
        self.L.oracle_orb_set_gaussian_taps(self.h, int(mode))

    def tables(self):
        n = self.nlevels
        f = [np.zeros(n, np.float32) for _ in range(4)]
        per = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.oracle_orb_tables(self.h, _p(f[0]), _p(f[1]), _p(f[2]), _p(f[3]), _p(per), _p(um))
        return dict(scale=f[0], inv_scale=f[1], sigma2=f[2], inv_sigma2=f[3], per_level=per, umax=um)

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        cap = self.nfeatures + 4 * self.nlevels + 16
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.oracle_orb_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), _p(desc),
                                      cap)
        assert n >= 0, "oracle capacity too small"
        return kps[:n].copy(), desc[:n].copy()

    def level_image(self, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        self.L.oracle_orb_level_size(self.h, level, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.oracle_orb_level_image(self.h, level, int(blurred), _p(out))
        return out

    def level_keypoints(self, level, stage):
        n = self.L.oracle_orb_level_keypoints(self.h, level, stage, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.oracle_orb_level_keypoints(self.h, level, stage, _p(out), n)
        return out[:n]


def resize_linear_u8(src, dw, dh):
    """cv::resize(INTER_LINEAR) on 8U as restated (App. B.2)."""
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear_u8(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def gaussian7_taps(mode=0):
    t = np.zeros(7, np.int32)
    lib().oracle_gaussian7_taps_mode(_p(t), int(mode))
    return t


def otsu_threshold(values):
    v = np.ascontiguousarray(values, np.uint8).reshape(-1)
    return lib().oracle_otsu_threshold(_p(v), len(v))


def knn2(Q, T, init=256):
    L = lib()
    Q = np.ascontiguousarray(Q, np.uint8)
    T = np.ascontiguousarray(T, np.uint8)
    bi = np.zeros(len(Q), np.int32)
    bd = np.zeros(len(Q), np.int32)
    sd = np.zeros(len(Q), np.int32)
    L.oracle_knn2(_p(Q), len(Q), _p(T), len(T), init, _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def _bounds(b):
    return None if b is None else np.ascontiguousarray(b, np.float32)


def features_in_area(kps2, cols, rows, qx, qy, r, min_level, max_level, bounds=None):
    L = lib()
    bounds = _bounds(bounds)
    bp = None if bounds is None else _p(bounds)
    kps2 = np.ascontiguousarray(kps2)
    qx = np.ascontiguousarray(qx, np.float32)
    qy = np.ascontiguousarray(qy, np.float32)
    off = np.zeros(len(qx) + 1, np.int32)
    total = L.oracle_features_in_area(_p(kps2), len(kps2), cols, rows, _p(qx), _p(qy), len(qx), r, min_level,
                                      max_level, _p(off), None, 0, bp)
    idx = np.zeros(max(total, 1), np.int32)
    L.oracle_features_in_area(_p(kps2), len(kps2), cols, rows, _p(qx), _p(qy), len(qx), r, min_level, max_level,
                              _p(off), _p(idx), total, bp)
    return off, idx[:total]


def knn2_csr(Q, T, off, idx, init=256):
    L = lib()
    Q = np.ascontiguousarray(Q, np.uint8)
    T = np.ascontiguousarray(T, np.uint8)
    bi = np.zeros(len(Q), np.int32)
    bd = np.zeros(len(Q), np.int32)
    sd = np.zeros(len(Q), np.int32)
    idx = np.ascontiguousarray(idx, np.int32)
    L.oracle_knn2_csr(_p(Q), len(Q), _p(T), _p(off), _p(idx), init, _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


WINDOW_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4")])


def search_by_projection(kps, desc, cols, rows, queries, qdesc, taken=None, mode=0, th_high=100, nnratio=0.8, bounds=None, q_observed=None):
    """ORBmatcher::SearchByProjection(Frame, MapPoints) matching loop on flat arrays; see oracle/match_oracle.cpp."""
    L = lib()
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, WINDOW_QUERY_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
    nq = len(queries)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8).copy()
    out = [np.zeros(nq, np.int32) for _ in range(6)]
    L.oracle_search_by_projection.restype = C.c_int
    L.oracle_search_by_projection.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 8
    qo = None if q_observed is None else np.ascontiguousarray(q_observed, np.uint8)
    nm = L.oracle_search_by_projection(_p(kps), _p(desc), len(kps), cols, rows, _p(queries), _p(qdesc), nq,
                                       None if tk is None else _p(tk), mode, th_high, nnratio, *[_p(o) for o in out],
                                       None if bounds is None else _p(_bounds(bounds)), None if qo is None else _p(qo))
    return dict(best_idx=out[0], best_dist=out[1], best_level=out[2], second_dist=out[3], second_level=out[4],
                match=out[5], nmatches=nm, taken=tk)


def undistort_points(pts, K4, dist):
    L = lib()
    src = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    K4 = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist, np.float32).reshape(-1)
    dst = np.empty_like(src)
    L.oracle_undistort_points.restype = None
    L.oracle_undistort_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_undistort_points(_p(src), len(src), _p(K4), _p(d) if len(d) else None, len(d), _p(dst))
    return dst


def compute_image_bounds(cols, rows, K4, dist):
    L = lib()
    K4 = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist, np.float32).reshape(-1)
    out = np.zeros(4, np.float32)
    L.oracle_compute_image_bounds.restype = None
    L.oracle_compute_image_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_compute_image_bounds(cols, rows, _p(K4), _p(d) if len(d) else None, len(d), _p(out))
    return out


def marker_pose(corners, marker_size, K4, dist):
    """-> (rvec1, tvec1, rvec2, tvec2) float64, err float32[2]"""
    L = lib()
    c = np.ascontiguousarray(corners, np.float32).reshape(4, 2)
    K4 = np.ascontiguousarray(K4, np.float32); d = np.ascontiguousarray(dist, np.float32).reshape(-1)
    out = np.zeros(12, np.float64); err = np.zeros(2, np.float32)
    L.oracle_marker_pose.restype = None
    L.oracle_marker_pose.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.oracle_marker_pose(_p(c), marker_size, _p(K4), _p(d) if len(d) else None, len(d), _p(out), _p(err))
    return out[0:3], out[3:6], out[6:9], out[9:12], err


def camera_resize(K4, cam_size, img_size):
    L = lib()
    K4 = np.ascontiguousarray(K4, np.float32); out = np.zeros(4, np.float32)
    L.oracle_camera_resize.restype = None
    L.oracle_camera_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.oracle_camera_resize(_p(K4), cam_size[0], cam_size[1], img_size[0], img_size[1], _p(out))
    return out


class VocabularyOracle:
    def __init__(self, handle):
        self.L = lib()
        self.h = handle

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.oracle_voc_destroy.argtypes = [C.c_void_p]
                self.L.oracle_voc_destroy(self.h)
                self.h = None
        except Exception:  # interpreter shutdown: module globals may be gone
            pass

    @classmethod
    def load_text(cls, filename):
        L = lib()
        L.oracle_voc_load_text.restype = C.c_void_p
        L.oracle_voc_load_text.argtypes = [C.c_char_p]
        h = L.oracle_voc_load_text(os.fsencode(filename))
        if not h:
            raise RuntimeError("oracle_voc_load_text failed")
        return cls(h)

    @classmethod
    def from_arrays(cls, k, Lv, scoring, weighting, parent, is_leaf, descriptors, weights):
        L = lib()
        L.oracle_voc_create.restype = C.c_void_p
        L.oracle_voc_create.argtypes = [C.c_int] * 4
        L.oracle_voc_add_node.restype = None
        L.oracle_voc_add_node.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double]
        h = L.oracle_voc_create(k, Lv, scoring, weighting)
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        for i in range(len(parent)):
            L.oracle_voc_add_node(h, int(parent[i]), int(is_leaf[i]), _p(d[i]), float(weights[i]))
        return cls(h)

    def info(self):
        out = np.zeros(6, np.int32)
        self.L.oracle_voc_info.restype = None
        self.L.oracle_voc_info.argtypes = [C.c_void_p, C.c_void_p]
        self.L.oracle_voc_info(self.h, _p(out))
        return dict(zip(("k", "L", "scoring", "weighting", "nodes", "words"), out.tolist()))

    def transform(self, descriptors, levelsup=4):
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
        self.L.oracle_voc_transform_features.restype = None
        self.L.oracle_voc_transform_features.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L.oracle_voc_transform_features(self.h, _p(d), n, levelsup, _p(word), _p(node), _p(weight))
        bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64); fn = np.zeros(max(n, 1), np.uint32)
        fo = np.zeros(n + 1, np.int32); ff = np.zeros(max(n, 1), np.uint32)
        nb, nf = C.c_int32(0), C.c_int32(0)
        self.L.oracle_voc_transform.restype = None
        self.L.oracle_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        self.L.oracle_voc_transform(self.h, _p(d), n, levelsup, _p(bw), _p(bv), C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf))
        nb, nf = nb.value, nf.value
        return dict(word=word, node=node, weight=weight, bow=(bw[:nb].copy(), bv[:nb].copy()),
                    fv=(fn[:nf].copy(), fo[:nf + 1].copy(), ff[:fo[nf]].copy()))


def search_by_bow(kps1, desc1, fv1, kps2, desc2, fv2, valid1=None, valid2=None, nnratio=0.7, check_orientation=True,
                  accept_max=50, factor=30 / 360.0):
    L = lib()
    k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    a = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    b = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    m12 = np.full(max(len(k1), 1), -1, np.int32); m21 = np.full(max(len(k2), 1), -1, np.int32)
    side = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.oracle_search_by_bow.restype = C.c_int
    L.oracle_search_by_bow.argtypes = side + side + [C.c_float, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    nm = L.oracle_search_by_bow(_p(k1), _p(d1), None if v1 is None else _p(v1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                _p(k2), _p(d2), None if v2 is None else _p(v2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                nnratio, int(check_orientation), accept_max, np.float32(factor), _p(m12), _p(m21))
    return nm, m12[:len(k1)], m21[:len(k2)]


def keyframe_features_pack(kps, desc, mp_index=None):
    L = lib()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    m = None if mp_index is None else np.ascontiguousarray(mp_index, np.uint64)
    out = np.zeros(len(k) * 68 + 8, np.uint8)
    L.oracle_keyframe_features_pack.restype = C.c_long
    L.oracle_keyframe_features_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    nb = L.oracle_keyframe_features_pack(_p(k), _p(d), None if m is None else _p(m), len(k), _p(out))
    return out[:nb].copy()


def keyframe_features_unpack(buf, n):
    L = lib()
    b = np.ascontiguousarray(buf, np.uint8)
    k = np.zeros(max(n, 1), KP_DTYPE); d = np.zeros((max(n, 1), 32), np.uint8); m = np.zeros(max(n, 1), np.uint64)
    L.oracle_keyframe_features_unpack.restype = C.c_long
    L.oracle_keyframe_features_unpack.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.oracle_keyframe_features_unpack(_p(b), n, _p(k), _p(d), _p(m))
    if rc < 0:
        raise ValueError("record %d: descriptor length is not 32" % (-rc - 1))
    return k[:n], d[:n], m[:n]


def search_by_projection_last_frame(kps_cur, desc_cur, cols, rows, kps_last, valid_last, x3Dw, mp_desc, Tcw, K4, scale_factors, th,
                                    taken_cur=None, mp_observed=None, th_high=100, check_orientation=True, bounds=None):
    L = lib()
    kc = np.ascontiguousarray(kps_cur, KP_DTYPE); dc = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
    kl = np.ascontiguousarray(kps_last, KP_DTYPE)
    opt = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
    vl, tc, ob, bnd = opt(valid_last, np.uint8), opt(taken_cur, np.uint8), opt(mp_observed, np.uint8), opt(bounds, np.float32)
    x = np.ascontiguousarray(x3Dw, np.float32).reshape(-1, 3); md = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy(); K = np.ascontiguousarray(K4, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    m = np.full(max(len(kc), 1), -1, np.int32)
    pp = lambda a: None if a is None else _p(a)
    vp = C.c_void_p
    L.oracle_search_by_projection_last_frame.restype = C.c_int
    L.oracle_search_by_projection_last_frame.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp,
                                                         C.c_float, C.c_int, C.c_int, vp]
    nm = L.oracle_search_by_projection_last_frame(_p(kc), _p(dc), len(kc), pp(tc), cols, rows, pp(bnd), _p(kl), len(kl), pp(vl), _p(x),
                                                  _p(md), pp(ob), _p(T), _p(K), _p(sf), th, th_high, int(check_orientation), _p(m))
    return nm, m[:len(kc)]


def search_by_projection_keyframe(kps_cur, desc_cur, cols, rows, kf_angle, valid, p3Dw, min_dist, max_dist, mp_desc, Tcw, Ow, K4, scale_factors,
                                  log_scale_factor, th, orb_dist, taken_cur=None, check_orientation=True, bounds=None):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1476-1603) -> (nmatches, match_cur)."""
    L = lib()
    kc = np.ascontiguousarray(kps_cur, KP_DTYPE); dc = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx, md, ang = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist), f(mp_desc, np.uint8).reshape(-1, 32), f(kf_angle)
    opt = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
    v, tc, bnd = opt(valid, np.uint8), opt(taken_cur, np.uint8), opt(bounds, np.float32)
    T, O, K, sf = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors)
    m = np.full(max(len(kc), 1), -1, np.int32)
    pp = lambda a: None if a is None else _p(a)
    vp = C.c_void_p
    L.oracle_search_by_projection_keyframe.restype = C.c_int
    L.oracle_search_by_projection_keyframe.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                       C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, vp]
    nm = L.oracle_search_by_projection_keyframe(_p(kc), _p(dc), len(kc), pp(tc), cols, rows, pp(bnd), len(x), _p(ang), pp(v), _p(x), _p(mn),
                                                _p(mx), _p(md), _p(T), _p(O), _p(K), _p(sf), len(sf), log_scale_factor, th, int(orb_dist),
                                                int(check_orientation), _p(m))
    return nm, m[:len(kc)]


def distinctive_descriptors(desc, offsets):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:270-333) over CSR lists -> best index per point."""
    L = lib()
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); o = np.ascontiguousarray(offsets, np.int32)
    n = len(o) - 1
    bi = np.zeros(max(n, 1), np.int32)
    L.oracle_distinctive_descriptors.restype = None
    L.oracle_distinctive_descriptors.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.oracle_distinctive_descriptors(_p(d) if len(d) else None, _p(o), n, _p(bi))
    return bi[:n]


def predict_scale(max_distance, dist, log_scale_factor, nlevels):
    """MapPoint::PredictScale (MapPoint.cc:414-446)."""
    L = lib()
    L.oracle_predict_scale.restype = C.c_int
    L.oracle_predict_scale.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
    return L.oracle_predict_scale(max_distance, dist, log_scale_factor, nlevels)


def search_for_triangulation(kps1, desc1, fv1, kps2, desc2, fv2, F12, epipole, scale_factors, level_sigma2, has_mp1=None, has_mp2=None,
                             check_orientation=True):
    L = lib()
    k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    a = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    b = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    h1 = None if has_mp1 is None else np.ascontiguousarray(has_mp1, np.uint8)
    h2 = None if has_mp2 is None else np.ascontiguousarray(has_mp2, np.uint8)
    F = np.ascontiguousarray(F12, np.float32).reshape(9); sf = np.ascontiguousarray(scale_factors, np.float32)
    sg = np.ascontiguousarray(level_sigma2, np.float32)
    m12 = np.full(max(len(k1), 1), -1, np.int32)
    side = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.oracle_search_for_triangulation.restype = C.c_int
    L.oracle_search_for_triangulation.argtypes = side + side + [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    nm = L.oracle_search_for_triangulation(_p(k1), _p(d1), None if h1 is None else _p(h1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                           _p(k2), _p(d2), None if h2 is None else _p(h2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                           _p(F), np.float32(epipole[0]), np.float32(epipole[1]), _p(sf), _p(sg), int(check_orientation), _p(m12))
    return nm, m12[:len(k1)]


def fuse_search(kps, desc, cols, rows, p3Dw, valid, min_dist, max_dist, normal, mp_desc, Tcw, Ow, K4, scale_factors, inv_level_sigma2,
                log_scale_factor, th, chi2=5.99, bounds=None):
    L = lib()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx, nr, md = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist), f(normal).reshape(-1, 3), f(mp_desc, np.uint8).reshape(-1, 32)
    v = None if valid is None else f(valid, np.uint8); bnd = None if bounds is None else f(bounds)
    T, O, K, sf, isg = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors), f(inv_level_sigma2)
    bi = np.full(max(len(x), 1), -1, np.int32); bd = np.full(max(len(x), 1), 256, np.int32)
    pp = lambda a: None if a is None else _p(a)
    vp = C.c_void_p
    L.oracle_fuse_search.restype = None
    L.oracle_fuse_search.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int,
                                     C.c_float, C.c_float, C.c_double, vp, vp]
    L.oracle_fuse_search(_p(k), _p(d), len(k), cols, rows, pp(bnd), _p(x), pp(v), _p(mn), _p(mx), _p(nr), _p(md), len(x), _p(T), _p(O), _p(K),
                         _p(sf), _p(isg), len(sf), log_scale_factor, th, float(chi2), _p(bi), _p(bd))
    return bi[:len(x)], bd[:len(x)]


def search_by_sim3(kf1, kf2, cols, rows, T1w, T2w, sT12, sT21, K4, scale_factors, log_scale_factor, th, bounds=None):
    L = lib()
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    def side(kf):
        v = None if kf.get("valid") is None else f(kf["valid"], np.uint8)
        return [np.ascontiguousarray(kf["kps"], KP_DTYPE), f(kf["desc"], np.uint8).reshape(-1, 32), f(kf["p3Dw"]).reshape(-1, 3), v,
                f(kf["min_dist"]), f(kf["max_dist"]), f(kf["mp_desc"], np.uint8).reshape(-1, 32)]
    a, b = side(kf1), side(kf2)
    pp = lambda x: None if x is None else _p(x)
    bnd = None if bounds is None else f(bounds)
    Ts = [f(T).reshape(-1)[:12].copy() for T in (T1w, T2w, sT12, sT21)]
    K, sf = f(K4), f(scale_factors)
    m12 = np.full(max(len(a[0]), 1), -1, np.int32)
    vp = C.c_void_p
    L.oracle_search_by_sim3.restype = C.c_int
    L.oracle_search_by_sim3.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, vp] + [vp] * 10 + [vp] * 6 + [C.c_int, C.c_float, C.c_float, vp]
    nf = L.oracle_search_by_sim3(_p(a[0]), _p(a[1]), len(a[0]), _p(b[0]), _p(b[1]), len(b[0]), cols, rows, pp(bnd),
                                 _p(a[2]), pp(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(b[2]), pp(b[3]), _p(b[4]), _p(b[5]), _p(b[6]),
                                 _p(Ts[0]), _p(Ts[1]), _p(Ts[2]), _p(Ts[3]), _p(K), _p(sf), len(sf), log_scale_factor, th, _p(m12))
    return nf, m12[:len(a[0])]


def search_by_projection_sim3(kps, desc, cols, rows, matched, p3Dw, valid, min_dist, max_dist, normal, mp_desc, Tcw, Ow, K4, scale_factors,
                              log_scale_factor, th, bounds=None):
    L = lib()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx, nr, md = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist), f(normal).reshape(-1, 3), f(mp_desc, np.uint8).reshape(-1, 32)
    v = None if valid is None else f(valid, np.uint8); bnd = None if bounds is None else f(bounds)
    mt = None if matched is None else f(matched, np.uint8)
    T, O, K, sf = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors)
    m = np.full(max(len(k), 1), -1, np.int32)
    pp = lambda a: None if a is None else _p(a)
    vp = C.c_void_p
    L.oracle_search_by_projection_sim3.restype = C.c_int
    L.oracle_search_by_projection_sim3.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp,
                                                   C.c_int, C.c_float, C.c_int, vp]
    nm = L.oracle_search_by_projection_sim3(_p(k), _p(d), len(k), cols, rows, pp(bnd), pp(mt), _p(x), pp(v), _p(mn), _p(mx), _p(nr), _p(md),
                                            len(x), _p(T), _p(O), _p(K), _p(sf), len(sf), log_scale_factor, int(th), _p(m))
    return nm, m[:len(k)]


def search_for_initialization(k1, d1, k2, d2, cols, rows, prev=None, window=100, nnratio=0.9, check_ori=True, bounds=None):
    L = lib()
    k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    if prev is None:
        prev = np.stack([k1["x"], k1["y"]], 1)
    prev = np.ascontiguousarray(prev, np.float32).copy()
    m = np.zeros(len(k1), np.int32)
    n = L.oracle_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), cols, rows, _p(prev),
                                           _p(m), window, nnratio, int(check_ori),
                                           None if bounds is None else _p(_bounds(bounds)))
    return n, m, prev


class ArucoOracle:
    def __init__(self, dictionary="ARUCO"):
        self.L = lib()
        self.h = self.L.oracle_aruco_create(dictionary.encode())
        if not self.h:
            raise ValueError("unknown dictionary " + dictionary)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_aruco_destroy(self.h)
            self.h = None

    def set_params(self, error_correction_rate=0.0, corner_lines=True):
        """MarkerDetector::Params::error_correction_rate and cornerRefinementM == CORNER_LINES (False: CORNER_NONE)."""
        self.L.oracle_aruco_set_params.argtypes = [C.c_void_p, C.c_float, C.c_int]
        self.L.oracle_aruco_set_params(self.h, error_correction_rate, int(corner_lines))

    def set_detection_mode(self, dm, min_marker_size=0.0):
        """Params::setDetectionMode (markerdetector.cpp:374-391): 0 DM_NORMAL, 1 DM_FAST, 2 DM_VIDEO_FAST."""
        self.L.oracle_aruco_set_detection_mode(self.h, int(dm), float(min_marker_size))

    def set_corner_method(self, m):
        """Params::setCornerRefinementMethod (:392-395): 0 CORNER_SUBPIX, 1 CORNER_LINES, 2 CORNER_NONE."""
        self.L.oracle_aruco_set_corner_method(self.h, int(m))

    def detect_enclosed_markers(self, on=True):
        """Params::detectEnclosedMarkers (markerdetector.h:126)."""
        self.L.oracle_aruco_set_enclosed(self.h, int(on))

    def set_tracking(self, min_detections):
        """Params::trackingMinDetections (markerdetector.h:187)."""
        self.L.oracle_aruco_set_tracking(self.h, int(min_detections))

    def tracked(self):
        return self.L.oracle_aruco_state(self.h, 4)

    def state(self):
        return {"threshold": self.L.oracle_aruco_state(self.h, 0), "min_size": self.L.oracle_aruco_min_size(self.h),
                "attempts": self.L.oracle_aruco_state(self.h, 1),
                "work_shape": (self.L.oracle_aruco_state(self.h, 3), self.L.oracle_aruco_state(self.h, 2))}

    def detect(self, img, capacity=256, bits15=0):
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim == 3:   # CV_8UC3: cvtColor(BGR2GRAY) first
            img = bgr_to_gray(img, bits15)
        out = np.zeros(capacity, MARKER_DTYPE)
        n = self.L.oracle_aruco_detect(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(out), capacity)
        return out[:min(n, capacity)].copy()

    def stage_image(self, stage):
        w, h = C.c_int(), C.c_int()
        if self.L.oracle_aruco_stage_image(self.h, stage, None, C.byref(w), C.byref(h)) != 0:
            return None
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.oracle_aruco_stage_image(self.h, stage, _p(out), C.byref(w), C.byref(h))
        return out

    def stage_count(self, which):
        return self.L.oracle_aruco_stage_count(self.h, which)

    def candidates(self, which):
        n = self.L.oracle_aruco_candidates(self.h, which, None, 0)
        out = np.zeros((max(n, 1), 9), np.float32)
        self.L.oracle_aruco_candidates(self.h, which, _p(out), n)
        return out[:n]

    def decode(self, patch):
        patch = np.ascontiguousarray(patch, np.uint8)
        rot = C.c_int(0)
        i = self.L.oracle_decode_marker(self.h, _p(patch), patch.shape[0], C.byref(rot))
        return i, rot.value


def bgr_to_gray(bgr, bits15=0):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.zeros(bgr.shape[:2], np.uint8)
    lib().oracle_bgr_to_gray(_p(bgr), bgr.shape[0], bgr.shape[1], bgr.strides[0], _p(out), int(bits15))
    return out


def resize_nearest(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_nearest(_p(img), img.shape[1], img.shape[0], _p(out), dw, dh)
    return out


def corner_subpix(img, pts, win, max_iters, eps):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.ascontiguousarray(pts, np.float32).reshape(-1, 2).copy()
    lib().oracle_corner_subpix(_p(img), img.shape[1], img.shape[0], _p(out), len(out), int(win), int(max_iters), float(eps))
    return out


def otsu_of_histogram(hist):
    hist = np.ascontiguousarray(hist, np.float32)
    return lib().oracle_otsu_of_histogram(_p(hist))


def adaptive_threshold(img, win, Cc=7):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros_like(img)
    lib().oracle_adaptive_threshold(_p(img), img.shape[1], img.shape[0], _p(out), win, Cc)
    return out


def find_contours(binimg, max_contours=200000, max_points=4000000):
    binimg = np.ascontiguousarray(binimg, np.uint8)
    lens = np.zeros(max_contours, np.int32)
    pts = np.zeros((max_points, 2), np.int32)
    n = lib().oracle_find_contours(_p(binimg), binimg.shape[1], binimg.shape[0], _p(lens), max_contours, _p(pts),
                                   max_points)
    lens = lens[:n]
    out, o = [], 0
    for l in lens:
        out.append(pts[o:o + l].copy())
        o += l
    return out


def warp35(img, quad, S=35):
    img = np.ascontiguousarray(img, np.uint8)
    quad = np.ascontiguousarray(quad, np.float32)
    out = np.zeros((S, S), np.uint8)
    lib().oracle_warp_perspective35(_p(img), img.shape[1], img.shape[0], _p(quad), _p(out), S)
    return out
