"""ctypes binding of liborbfe.so (the C ABI of include/orbfe.h).

This is plumbing for tests and bench.py; the product is the shared library.  There is no Python or CPU
implementation behind these classes: if the library is missing or no HIP device is usable, construction fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBFE_LIB", os.path.join(_HERE, "liborbfe.so"))

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
MARKER_DTYPE = np.dtype([("id", "<i4"), ("corners", "<f4", (4, 2))])
POSE_DTYPE = np.dtype([("rvec", "<f4", 3), ("tvec", "<f4", 3), ("rvec2", "<f4", 3), ("tvec2", "<f4", 3), ("err", "<f4", 2)])
assert KP_DTYPE.itemsize == 28 and MARKER_DTYPE.itemsize == 36

# every symbol include/orbfe.h declares (tests/test_abi.py checks the header against this list and the .so)
SYMBOLS = [
    "orbfe_last_error", "orbfe_version", "orbfe_device_count",
    "orbfe_extractor_create", "orbfe_extractor_destroy", "orbfe_extractor_get_levels",
    "orbfe_extractor_get_scale_factor", "orbfe_extractor_get_scale_factors",
    "orbfe_extractor_get_inverse_scale_factors", "orbfe_extractor_get_scale_sigma_squares",
    "orbfe_extractor_get_inverse_scale_sigma_squares", "orbfe_extractor_get_features_per_level",
    "orbfe_extractor_max_keypoints", "orbfe_extract", "orbfe_extract_batch", "orbfe_extract_batch_device",
    "orbfe_extractor_batch_status", "orbfe_search_for_initialization_batch_status", "orbfe_extractor_set_gaussian_taps",
    "orbfe_extractor_debug_level_size", "orbfe_extractor_debug_level_image",
    "orbfe_extractor_debug_level_keypoints", "orbfe_extractor_debug_kernel_times", "orbfe_extractor_set_aux_stream", "orbfe_extractor_set_early_stream", "orbfe_extractor_follow", "orbfe_extractor_stage_wait", "orbfe_extractor_pair_detector",
    "orbfe_debug_control", "orbfe_hamming", "orbfe_three_maxima", "orbfe_epipolar_distance_ok", "orbfe_knn2", "orbfe_knn2_csr", "orbfe_knn2_batch_device", "orbfe_search_for_initialization",
    "orbfe_search_for_initialization_batch_device", "orbfe_search_by_projection",
    "orbfe_undistort_points", "orbfe_undistort_keypoints_batch_device", "orbfe_compute_image_bounds",
    "orbfe_aruco_create", "orbfe_aruco_destroy", "orbfe_aruco_set_dictionary", "orbfe_aruco_max_markers",
    "orbfe_aruco_detect", "orbfe_aruco_detect_batch", "orbfe_aruco_detect_batch_device", "orbfe_aruco_debug_image",
    "orbfe_aruco_debug_kernel_times", "orbfe_aruco_set_aux_stream",
    "orbfe_aruco_batch_status", "orbfe_aruco_set_big_frames", "orbfe_aruco_set_error_correction_rate",
    "orbfe_aruco_set_detection_mode", "orbfe_aruco_set_corner_refinement", "orbfe_aruco_marker_contour", "orbfe_aruco_marker_contours",
    "orbfe_camera_resize", "orbfe_marker_poses", "orbfe_marker_poses_batch_device", "orbfe_aruco_detect_poses",
    "orbfe_aruco_detect_bgr", "orbfe_aruco_detect_poses_bgr", "orbfe_aruco_get_state", "orbfe_aruco_set_gray_conversion", "orbfe_aruco_set_enclosed_markers", "orbfe_aruco_set_tracking", "orbfe_aruco_last_tracked", "orbfe_corner_subpix",
    "orbfe_vocabulary_load_text", "orbfe_vocabulary_create", "orbfe_vocabulary_destroy", "orbfe_vocabulary_info",
    "orbfe_vocabulary_transform", "orbfe_vocabulary_transform_batch_device",
    "orbfe_search_by_bow", "orbfe_search_by_bow_batch_device",
    "orbfe_search_for_triangulation", "orbfe_search_for_triangulation_batch_device",
    "orbfe_distinctive_descriptors", "orbfe_distinctive_descriptors_device", "orbfe_search_by_projection_batch_device",
    "orbfe_search_by_projection_batch_status", "orbfe_release_stream_scratch", "orbfe_search_by_projection_last_frame", "orbfe_search_by_projection_best", "orbfe_search_by_projection_keyframe", "orbfe_fuse_search", "orbfe_fuse_search_batch_device", "orbfe_project_map_points", "orbfe_search_by_sim3", "orbfe_search_by_projection_sim3",
    "orbfe_keyframe_features_pack", "orbfe_keyframe_features_unpack", "orbfe_keyframe_features_pack_device",
    "orbfe_keyframe_features_unpack_device",
    # the batched-video mode (csrc/pipeline.hip; ctypes prototypes in pipeline.py)
    "orbfe_pipeline_config_default", "orbfe_pipeline_create", "orbfe_pipeline_destroy", "orbfe_pipeline_layout", "orbfe_pipeline_step",
    "orbfe_pipeline_flush", "orbfe_pipeline_synchronize", "orbfe_pipeline_input_done", "orbfe_pipeline_status", "orbfe_pipeline_set_big_frames",
    "orbfe_pipeline_records", "orbfe_pipeline_matches", "orbfe_pipeline_reset_stream", "orbfe_pipeline_extractor", "orbfe_pipeline_detector",
    "orbfe_pipeline_engine_sets", "orbfe_pipeline_enable_timing", "orbfe_pipeline_timing_us", "orbfe_pipeline_env_defaults",
    "orbfe_pipeline_comm_unique_id", "orbfe_pipeline_comm_init", "orbfe_pipeline_set_comm", "orbfe_pipeline_gathered",
    "orbfe_pipeline_host_copy_us", "orbfe_pipeline_gathered_set", "orbfe_pipeline_gathered_wait", "orbfe_pipeline_gathered_release", "orbfe_pipeline_gathered_batch", "orbfe_pipeline_copy_stream_priority", "orbfe_pipeline_gather_plan",
    "orbfe_pipeline_step_host", "orbfe_pipeline_host_records", "orbfe_host_alloc", "orbfe_host_free",
    "orbfe_device_alloc", "orbfe_device_free", "orbfe_device_upload_rows", "orbfe_device_download",
]

_lib = None


class OrbfeError(RuntimeError):
    pass


def load():
    """Load liborbfe.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbfeError("%s not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950)"
                         % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_version.restype = C.c_char_p
    L.orbfe_extractor_create.restype = vp
    L.orbfe_extractor_create.argtypes = [i32, f32, i32, i32, i32, i32]
    L.orbfe_extractor_destroy.argtypes = [vp]
    L.orbfe_extractor_get_levels.argtypes = [vp]
    L.orbfe_extractor_get_scale_factor.argtypes = [vp]
    L.orbfe_extractor_get_scale_factor.restype = f32
    for nm in ("scale_factors", "inverse_scale_factors", "scale_sigma_squares", "inverse_scale_sigma_squares",
               "features_per_level"):
        getattr(L, "orbfe_extractor_get_" + nm).argtypes = [vp, vp]
    L.orbfe_extractor_max_keypoints.argtypes = [vp]
    L.orbfe_extract.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, vp]
    L.orbfe_extract_batch.argtypes = [vp, vp, i32, sz, i32, i32, sz, vp, vp, i32, vp]
    L.orbfe_extract_batch_device.argtypes = [vp, vp, i32, sz, i32, i32, sz, vp, vp, i32, vp, vp]
    L.orbfe_extractor_batch_status.argtypes = [vp, vp]
    L.orbfe_extractor_set_gaussian_taps.argtypes = [vp, i32]
    L.orbfe_extractor_debug_level_size.argtypes = [vp, i32, vp, vp]
    L.orbfe_extractor_debug_level_image.argtypes = [vp, i32, i32, i32, vp]
    L.orbfe_extractor_debug_level_keypoints.argtypes = [vp, i32, i32, i32, vp, i32, vp]
    L.orbfe_extractor_debug_kernel_times.argtypes = [vp, vp, i32]
    L.orbfe_extractor_set_early_stream.argtypes = [vp, vp]
    L.orbfe_extractor_follow.argtypes = [vp, vp, i32]
    L.orbfe_extractor_stage_wait.argtypes = [vp, i32, vp]
    L.orbfe_extractor_pair_detector.argtypes = [vp, vp]
    L.orbfe_extractor_set_aux_stream.argtypes = [vp, vp]
    if hasattr(L, "orbfe_knn2"):
        L.orbfe_debug_control.argtypes = [C.c_char_p, i32]
        L.orbfe_search_by_projection.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, i32, i32, C.c_float] + [vp] * 7 + [i32]
        L.orbfe_hamming.argtypes = [vp, vp]
        L.orbfe_search_by_projection_last_frame.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32,
                                                            i32, i32, vp, vp, i32]
        L.orbfe_fuse_search.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, f32, f32, C.c_double,
                                        vp, vp, i32]
        L.orbfe_search_by_sim3.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, vp] + [vp] * 10 + [vp] * 6 + [i32, f32, f32, i32, vp, vp, i32]
        L.orbfe_search_by_projection_sim3.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, f32, i32,
                                                      vp, vp, i32]
        L.orbfe_project_map_points.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, i32, vp, i32, f32, f32, i32, i32, vp, i32]
        L.orbfe_search_by_projection_keyframe.argtypes = [vp, vp, i32, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, f32, f32,
                                                          i32, i32, vp, vp, i32]
        L.orbfe_search_by_projection_best.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, f32, vp, vp, i32]
        L.orbfe_search_by_projection_batch_device.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, f32, f32, i32] + [vp] * 9
        L.orbfe_search_by_projection_batch_status.argtypes = [vp, vp]
        L.orbfe_fuse_search_batch_device.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, f32, f32,
                                                     C.c_double, vp, vp, vp]
        L.orbfe_distinctive_descriptors.argtypes = [vp, vp, i32, vp, vp, i32]
        L.orbfe_distinctive_descriptors_device.argtypes = [vp, vp, i32, vp, vp, vp]
        L.orbfe_undistort_points.argtypes = [vp, i32, vp, vp, i32, vp, i32]
        L.orbfe_undistort_keypoints_batch_device.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp]
        L.orbfe_compute_image_bounds.argtypes = [i32, i32, vp, vp, i32, vp, i32]
        L.orbfe_knn2.argtypes = [vp, i32, vp, i32, i32, vp, vp, vp, i32]
        L.orbfe_knn2_csr.argtypes = [vp, i32, vp, i32, vp, vp, i32, vp, vp, vp, i32]
        L.orbfe_knn2_batch_device.argtypes = [vp, vp, sz, i32, vp, vp, sz, i32, i32, i32, vp, vp, vp, vp]
        L.orbfe_search_for_initialization.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, i32, f32, i32, vp,
                                                      i32]
        L.orbfe_search_for_initialization_batch_device.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, i32, f32, i32, vp,
                                                                   vp, vp]
        L.orbfe_search_for_initialization_batch_status.argtypes = [vp, vp]
    if hasattr(L, "orbfe_aruco_create"):
        L.orbfe_aruco_create.restype = vp
        L.orbfe_aruco_create.argtypes = [C.c_char_p, i32]
        L.orbfe_aruco_destroy.argtypes = [vp]
        L.orbfe_aruco_set_dictionary.argtypes = [vp, C.c_char_p]
        L.orbfe_aruco_max_markers.argtypes = [vp]
        L.orbfe_aruco_detect.argtypes = [vp, vp, i32, i32, sz, vp, i32, vp]
        L.orbfe_aruco_detect_poses.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, vp, f32, vp, vp, i32]
        L.orbfe_aruco_detect_batch.argtypes = [vp, vp, i32, sz, i32, i32, sz, vp, i32, vp]
        L.orbfe_aruco_detect_batch_device.argtypes = [vp, vp, i32, sz, i32, i32, sz, vp, i32, vp, vp]
        L.orbfe_aruco_debug_image.argtypes = [vp, i32, i32, vp]
        L.orbfe_aruco_debug_kernel_times.argtypes = [vp, vp, i32]
        L.orbfe_aruco_set_aux_stream.argtypes = [vp, vp]
        L.orbfe_aruco_batch_status.argtypes = [vp, vp, vp]
        L.orbfe_aruco_set_big_frames.argtypes = [vp, i32]
        L.orbfe_aruco_set_error_correction_rate.argtypes = [vp, f32]
        L.orbfe_aruco_set_detection_mode.argtypes = [vp, i32, f32]
        L.orbfe_aruco_set_corner_refinement.argtypes = [vp, i32]
        L.orbfe_aruco_detect_bgr.argtypes = [vp, vp, i32, i32, sz, vp, i32, vp]
        L.orbfe_aruco_detect_poses_bgr.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, vp, f32, vp, vp, i32]
        L.orbfe_aruco_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orbfe_aruco_set_gray_conversion.argtypes = [vp, i32]
        L.orbfe_aruco_set_enclosed_markers.argtypes = [vp, i32]
        L.orbfe_aruco_set_tracking.argtypes = [vp, i32]
        L.orbfe_aruco_last_tracked.argtypes = [vp]
        L.orbfe_corner_subpix.argtypes = [vp, i32, i32, sz, vp, i32, i32, i32, C.c_double, i32]
        L.orbfe_aruco_marker_contour.argtypes = [vp, i32, i32, vp, i32, vp]
        L.orbfe_aruco_marker_contours.argtypes = [vp, i32, i32, vp, i32, vp]
        L.orbfe_camera_resize.argtypes = [vp, i32, i32, i32, i32, vp]
    if hasattr(L, "orbfe_vocabulary_create"):
        L.orbfe_vocabulary_load_text.restype = vp
        L.orbfe_vocabulary_load_text.argtypes = [C.c_char_p, i32]
        L.orbfe_vocabulary_create.restype = vp
        L.orbfe_vocabulary_create.argtypes = [i32, i32, i32, i32, i32, vp, vp, vp, vp, i32]
        L.orbfe_vocabulary_destroy.restype = None
        L.orbfe_vocabulary_destroy.argtypes = [vp]
        L.orbfe_vocabulary_info.argtypes = [vp, vp]
        L.orbfe_vocabulary_transform.argtypes = [vp, vp, i32, i32] + [vp] * 10
        L.orbfe_vocabulary_transform_batch_device.argtypes = [vp, vp, vp, i32, i32, i32] + [vp] * 11
        L.orbfe_keyframe_features_pack.argtypes = [vp, vp, vp, i32, vp, i32]
        L.orbfe_keyframe_features_unpack.argtypes = [vp, i32, vp, vp, vp, i32]
        L.orbfe_keyframe_features_unpack_device.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp, vp]
        L.orbfe_keyframe_features_pack_device.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp]
        side = [vp, vp, vp, i32, vp, vp, vp, i32]
        L.orbfe_search_by_bow.argtypes = side + side + [f32, i32, i32, f32, vp, vp, vp, i32]
        L.orbfe_search_for_triangulation.argtypes = side + side + [vp, f32, f32, vp, vp, i32, i32, vp, vp, i32]
        L.orbfe_search_for_triangulation_batch_device.argtypes = [vp] * 8 + [i32, vp, vp, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
        L.orbfe_search_by_bow_batch_device.argtypes = [vp] * 8 + [i32, vp, vp, i32, i32, f32, i32, i32, f32, vp, vp, vp, vp]
        L.orbfe_marker_poses.argtypes = [vp, i32, f32, vp, vp, i32, vp, i32]
        L.orbfe_marker_poses_batch_device.argtypes = [vp, vp, i32, i32, f32, vp, vp, i32, vp, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _check(L, rc, what):
    if rc < 0:
        raise OrbfeError("%s failed (%d): %s" % (what, rc, L.orbfe_last_error().decode()))
    return rc


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:45-113) over the C ABI."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0):
        self.L = load()
        self.nlevels = nlevels
        self.h = self.L.orbfe_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device)
        if not self.h:
            raise OrbfeError("orbfe_extractor_create: " + self.L.orbfe_last_error().decode())
        self.capacity = self.L.orbfe_extractor_max_keypoints(self.h)

    @classmethod
    def wrap(cls, handle, nlevels):
        """A non-owning view of an extractor another object owns (the engines of an orbfe_pipeline)."""
        self = cls.__new__(cls)
        self.L = load()
        self.nlevels = nlevels
        self.h = handle
        self._borrowed = True
        self.capacity = self.L.orbfe_extractor_max_keypoints(self.h)
        return self

    def __del__(self):
        if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
            self.L.orbfe_extractor_destroy(self.h)
        self.h = None

    # reference getters (ORBextractor.h:63-83)
    def GetLevels(self):
        return self.L.orbfe_extractor_get_levels(self.h)

    def GetScaleFactor(self):
        return self.L.orbfe_extractor_get_scale_factor(self.h)

    def _vec(self, name, dtype=np.float32):
        out = np.zeros(self.nlevels, dtype)
        _check(self.L, getattr(self.L, "orbfe_extractor_get_" + name)(self.h, _p(out)), name)
        return out

    def GetScaleFactors(self):
        return self._vec("scale_factors")

    def GetInverseScaleFactors(self):
        return self._vec("inverse_scale_factors")

    def GetScaleSigmaSquares(self):
        return self._vec("scale_sigma_squares")

    def GetInverseScaleSigmaSquares(self):
        return self._vec("inverse_scale_sigma_squares")

    def features_per_level(self):
        return self._vec("features_per_level", np.int32)

    def __call__(self, image, mask=None):
        """operator()(image, mask, keypoints, descriptors): returns (keypoints[KP_DTYPE], descriptors[n,32])."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1050)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int32(0)
        _check(self.L, self.L.orbfe_extract(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0],
                                            _p(kps), _p(desc), self.capacity, C.byref(n)), "orbfe_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: (B, rows, cols) uint8 host array. Returns list of (kps, desc)."""
        images = np.ascontiguousarray(images, np.uint8)
        B, rows, cols = images.shape
        kps = np.zeros((B, self.capacity), KP_DTYPE)
        desc = np.zeros((B, self.capacity, 32), np.uint8)
        n = np.zeros(B, np.int32)
        _check(self.L, self.L.orbfe_extract_batch(self.h, _p(images), B, images.strides[0], rows, cols,
                                                  images.strides[1], _p(kps), _p(desc), self.capacity, _p(n)),
               "orbfe_extract_batch")
        return [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(B)]

    def extract_batch_device(self, d_imgs_ptr, B, frame_stride, rows, cols, step, d_kps_ptr, d_desc_ptr, capacity,
                             d_n_ptr, stream=0):
        _check(self.L, self.L.orbfe_extract_batch_device(self.h, d_imgs_ptr, B, frame_stride, rows, cols, step,
                                                         d_kps_ptr, d_desc_ptr, capacity, d_n_ptr, stream),
               "orbfe_extract_batch_device")

    def set_gaussian_taps(self, mode):
        """0: taps 18 34 49 55 49 34 18 (OpenCV 2.4 / 3.2 / early 3.4); 1: 18 34 48 56 48 34 18 (late 3.4.x / 4.x)."""
        _check(self.L, self.L.orbfe_extractor_set_gaussian_taps(self.h, int(mode)), "orbfe_extractor_set_gaussian_taps")

    def batch_status(self):
        """0, or the largest per-frame keypoint total of the last device batch that did not fit its capacity."""
        ovf = C.c_int32(0)
        _check(self.L, self.L.orbfe_extractor_batch_status(self.h, C.byref(ovf)), "orbfe_extractor_batch_status")
        return ovf.value

    # stage read-back for parity tests
    def level_image(self, frame, level, blurred=False):
        w, h = C.c_int(), C.c_int()
        _check(self.L, self.L.orbfe_extractor_debug_level_size(self.h, level, C.byref(w), C.byref(h)), "level_size")
        out = np.zeros((h.value, w.value), np.uint8)
        _check(self.L, self.L.orbfe_extractor_debug_level_image(self.h, frame, level, int(blurred), _p(out)),
               "level_image")
        return out

    def level_keypoints(self, frame, level, stage):
        n = C.c_int32(0)
        _check(self.L, self.L.orbfe_extractor_debug_level_keypoints(self.h, frame, level, stage, None, 0, C.byref(n)),
               "level_keypoints")
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        _check(self.L, self.L.orbfe_extractor_debug_level_keypoints(self.h, frame, level, stage, _p(out), n.value,
                                                                    C.byref(n)), "level_keypoints")
        return out[:n.value]

    def level_sizes(self):
        out = []
        for l in range(self.nlevels):
            w, h = C.c_int(), C.c_int()
            _check(self.L, self.L.orbfe_extractor_debug_level_size(self.h, l, C.byref(w), C.byref(h)), "level_size")
            out.append((w.value, h.value))
        return out

    def set_aux_stream(self, stream_ptr):
        """Run the extractor's forked launch (blur) on the caller's stream (None: the handle's own)."""
        _check(self.L, self.L.orbfe_extractor_set_aux_stream(self.h, stream_ptr), "set_aux_stream")

    def pair_detector(self, detector):
        """Start `detector` (a MarkerDetector or None) on every single frame this extractor is given: the detector's next detect()
        of the same image finds its work done (orbfe_extractor_pair_detector)."""
        _check(self.L, self.L.orbfe_extractor_pair_detector(self.h, detector.h if detector is not None else None), "pair_detector")
        self._paired = detector      # keeps the detector alive as long as it is paired

    def follow(self, other, stage):
        """Every batch of this handle starts behind stage 1 (FAST) / 2 (quadtree) / 3 (descriptors) of `other`'s latest batch; 0: free."""
        _check(self.L, self.L.orbfe_extractor_follow(self.h, other.h if other is not None else None, int(stage)), "follow")
        self._followed = other

    def set_early_stream(self, stream_ptr):
        """Run the extractor's launch that needs no pyramid (FAST of level 0) on the caller's stream (None: the handle's own)."""
        _check(self.L, self.L.orbfe_extractor_set_early_stream(self.h, stream_ptr), "set_early_stream")

    def enable_kernel_timing(self, on=True):
        self.L.orbfe_extractor_debug_kernel_times(self.h, None, int(on))

    def set_blur_on_matrix_cores(self, on=True):
        """k_blur7_mfma (default) / k_blur7: the tests run both."""
        self.L.orbfe_extractor_debug_kernel_times(self.h, None, 23 if on else 24)

    def force_general_quadtree(self, on=True):
        """Test hook: bypass the count-pyramid fast path of DistributeOctTree."""
        self.L.orbfe_extractor_debug_kernel_times(self.h, None, 2 if on else 3)

    def set_pyramid_depth(self, depth=0):
        """Test hook: depth of the quadtree count pyramid (0 = default); shallow values force the fallback."""
        self.L.orbfe_extractor_debug_kernel_times(self.h, None, 10 + depth)

    def quadtree_fell_back(self, frame, level):
        n = C.c_int32(0)
        _check(self.L, self.L.orbfe_extractor_debug_level_keypoints(self.h, frame, level, 3, None, 0, C.byref(n)),
               "level_keypoints")
        return bool(n.value)

    # order of kernel_times_us(); blur7 runs on the extractor's second stream, concurrently with fast_cells + distribute
    STAGES = ["resize", "blur7", "fast_cells", "distribute", "orient_describe"]
    # with FAST of level 0 on its own stream from the start of the batch (the default): its interval comes first, and
    # "fast_cells" is the launch over levels >= 1
    STAGES_L0 = ["fast_cells_l0"] + STAGES

    @classmethod
    def stage_names(cls, n):
        return cls.STAGES_L0 if n == len(cls.STAGES_L0) else cls.STAGES

    def kernel_times_us(self, median=False):
        """Stage times of the last batch, or (median=True) the per-stage median over the batches since timing was enabled."""
        out = np.zeros(32, np.float32)
        n = self.L.orbfe_extractor_debug_kernel_times(self.h, _p(out), -32 if median else 32)
        return out[:n]


# ------------------------------------------------------------------------------------------ matching --
def hamming(a, b):
    L = load()
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return L.orbfe_hamming(_p(a), _p(b))


def knn2_csr(Q, T, offsets, idx, init=256, device=0):
    """Best / second-best of every query over its own candidate list (CSR)."""
    L = load()
    Q = np.ascontiguousarray(Q, np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, np.int32); idx = np.ascontiguousarray(idx, np.int32)
    bi = np.full(len(Q), -1, np.int32); bd = np.full(len(Q), init, np.int32); sd = np.full(len(Q), init, np.int32)
    _check(L, L.orbfe_knn2_csr(_p(Q), len(Q), _p(T), len(T), _p(offsets), _p(idx), init, _p(bi), _p(bd), _p(sd), device),
           "orbfe_knn2_csr")
    return bi, bd, sd


def debug_control(key, value):
    L = load()
    _check(L, L.orbfe_debug_control(key.encode(), int(value)), "orbfe_debug_control")


def distinctive_descriptors(desc, offsets, device=0):
    """MapPoint::ComputeDistinctiveDescriptors over many map points (MapPoint.cc:270-333) -> (best index per point, chosen descriptors)."""
    L = load()
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    o = np.ascontiguousarray(offsets, np.int32)
    n = len(o) - 1
    bi = np.full(max(n, 1), -1, np.int32); bd = np.zeros((max(n, 1), 32), np.uint8)
    _check(L, L.orbfe_distinctive_descriptors(_p(d) if len(d) else None, _p(o), n, _p(bi), _p(bd), device), "orbfe_distinctive_descriptors")
    return bi[:n], bd[:n]


def knn2(Q, T, init=256, device=0):
    """All-pairs best / second-best (ORBmatcher inner loop). Returns (best_idx, best_dist, second_dist)."""
    L = load()
    Q = np.ascontiguousarray(Q, np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(T, np.uint8).reshape(-1, 32)
    bi = np.full(len(Q), -1, np.int32); bd = np.full(len(Q), init, np.int32); sd = np.full(len(Q), init, np.int32)
    _check(L, L.orbfe_knn2(_p(Q), len(Q), _p(T), len(T), init, _p(bi), _p(bd), _p(sd), device), "orbfe_knn2")
    return bi, bd, sd


WINDOW_QUERY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4")])


def search_by_projection(kps, desc, cols, rows, queries, qdesc, taken=None, mode=0, th_high=100, nnratio=0.8, device=0,
                         bounds=None, q_observed=None):
    """The matching loop of ORBmatcher::SearchByProjection(Frame&, vpMapPoints, th) (ORBmatcher.cc:45-129) on flat arrays:
    queries = WINDOW_QUERY_DTYPE records (projected position, radius, octave range), qdesc = their descriptors.
    mode 0: best / second-best + octaves per query; mode 1: the whole loop (accept rule, taken keypoints)."""
    L = load()
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    queries = np.ascontiguousarray(queries, WINDOW_QUERY_DTYPE); qdesc = np.ascontiguousarray(qdesc, np.uint8)
    nq = len(queries)
    tk = None if taken is None else np.ascontiguousarray(taken, np.uint8).copy()
    out = [np.zeros(max(nq, 1), np.int32) for _ in range(6)]
    nm = C.c_int32(0)
    bnd = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
    _check(L, L.orbfe_search_by_projection(_p(kps), _p(desc), len(kps), cols, rows, None if bnd is None else _p(bnd),
                                           _p(queries), _p(qdesc), nq,
                                           None if tk is None else _p(tk),
                                           None if q_observed is None else _p(np.ascontiguousarray(q_observed, np.uint8)),
                                           mode, th_high, nnratio, *[_p(o) for o in out],
                                           C.byref(nm), device), "orbfe_search_by_projection")
    return dict(best_idx=out[0][:nq], best_dist=out[1][:nq], best_level=out[2][:nq], second_dist=out[3][:nq],
                second_level=out[4][:nq], match=out[5][:nq], nmatches=nm.value, taken=tk)


def _camera(K, dist):
    """K: 3x3 matrix or (fx, fy, cx, cy); dist: mDistCoef (4 or 5 values, up to 12)."""
    K = np.asarray(K, np.float32)
    K4 = np.ascontiguousarray([K[0, 0], K[1, 1], K[0, 2], K[1, 2]] if K.ndim == 2 else K.reshape(4), np.float32)
    d = np.ascontiguousarray(np.asarray([] if dist is None else dist, np.float32).reshape(-1))
    return K4, d


def undistort_points(pts, K, dist, device=0):
    """cv::undistortPoints(pts, K, dist, R=I, P=K) on an (n, 2) float array (Frame.cc:357-416)."""
    L = load()
    K4, d = _camera(K, dist)
    src = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    dst = np.empty_like(src)
    _check(L, L.orbfe_undistort_points(_p(src), len(src), _p(K4), _p(d) if len(d) else None, len(d), _p(dst), device),
           "orbfe_undistort_points")
    return dst


def UndistortKeyPoints(kps, K, dist, device=0):
    """Frame::UndistortKeyPoints (Frame.cc:357-387): mvKeysUn = mvKeys with (x, y) undistorted; unchanged when dist[0] == 0."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    K4, d = _camera(K, dist)
    if len(d) == 0 or d[0] == 0.0 or len(kps) == 0:
        return kps.copy()
    un = kps.copy()
    xy = undistort_points(np.stack([kps["x"], kps["y"]], 1), K4, d, device)
    un["x"] = xy[:, 0]; un["y"] = xy[:, 1]
    return un


def ComputeImageBounds(cols, rows, K, dist, device=0):
    """Frame::ComputeImageBounds (Frame.cc:418-451) -> float32 [mnMinX, mnMinY, mnMaxX, mnMaxY] (the `bounds` of the searches)."""
    L = load()
    K4, d = _camera(K, dist)
    out = np.zeros(4, np.float32)
    _check(L, L.orbfe_compute_image_bounds(cols, rows, _p(K4), _p(d) if len(d) else None, len(d), _p(out), device),
           "orbfe_compute_image_bounds")
    return out


def search_by_projection_last_frame(kps_cur, desc_cur, cols, rows, kps_last, valid_last, x3Dw, mp_desc, Tcw, K4, scale_factors, th,
                                    taken_cur=None, mp_observed=None, th_high=100, check_orientation=True, bounds=None, device=0):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, mono) (ORBmatcher.cc:1332-1474) -> (nmatches, match_cur)."""
    L = load()
    kc = np.ascontiguousarray(kps_cur, KP_DTYPE); dc = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
    kl = np.ascontiguousarray(kps_last, KP_DTYPE)
    opt = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
    vl, tc, ob, bnd = opt(valid_last, np.uint8), opt(taken_cur, np.uint8), opt(mp_observed, np.uint8), opt(bounds, np.float32)
    x = np.ascontiguousarray(x3Dw, np.float32).reshape(-1, 3); md = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy(); K = np.ascontiguousarray(K4, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    m = np.full(len(kc), -1, np.int32); nm = C.c_int32(0)
    pp = lambda a: None if a is None else _p(a)
    _check(L, L.orbfe_search_by_projection_last_frame(_p(kc), _p(dc), len(kc), pp(tc), cols, rows, pp(bnd), _p(kl), len(kl), pp(vl), _p(x),
                                                      _p(md), pp(ob), _p(T), _p(K), _p(sf), len(sf), th, th_high, int(check_orientation),
                                                      _p(m), C.byref(nm), device), "orbfe_search_by_projection_last_frame")
    return nm.value, m


def search_by_projection_best(kps, desc, cols, rows, queries, q_angle, qdesc, th_high, factor, q_blocks=None, taken=None,
                              check_orientation=True, bounds=None, device=0):
    """Best-only guided search with rotation consistency (ORBmatcher.cc:1476-1603 and relatives) -> (nmatches, match_cur)."""
    L = load()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    q = np.ascontiguousarray(queries, WINDOW_QUERY_DTYPE); qa = np.ascontiguousarray(q_angle, np.float32)
    qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    opt = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
    qb, tk, bnd = opt(q_blocks, np.uint8), opt(taken, np.uint8), opt(bounds, np.float32)
    pp = lambda a: None if a is None else _p(a)
    m = np.full(len(k), -1, np.int32); nm = C.c_int32(0)
    _check(L, L.orbfe_search_by_projection_best(_p(k), _p(d), len(k), cols, rows, pp(bnd), _p(q), _p(qa), _p(qd), pp(qb), len(q), pp(tk),
                                                th_high, int(check_orientation), np.float32(factor), _p(m), C.byref(nm), device),
           "orbfe_search_by_projection_best")
    return nm.value, m


def fuse_search(kps, desc, cols, rows, p3Dw, valid, min_dist, max_dist, normal, mp_desc, Tcw, Ow, K4, scale_factors, inv_level_sigma2,
                log_scale_factor, th, chi2=5.99, bounds=None, device=0):
    """Matching part of ORBmatcher::Fuse (ORBmatcher.cc:829-970; chi2=0: the Scw variant :972-1104) -> (best_idx, best_dist)."""
    L = load()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx, nr, md = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist), f(normal).reshape(-1, 3), f(mp_desc, np.uint8).reshape(-1, 32)
    v = None if valid is None else f(valid, np.uint8); bnd = None if bounds is None else f(bounds)
    T, O, K, sf, isg = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors), f(inv_level_sigma2)
    bi = np.full(len(x), -1, np.int32); bd = np.full(len(x), 256, np.int32)
    pp = lambda a: None if a is None else _p(a)
    _check(L, L.orbfe_fuse_search(_p(k), _p(d), len(k), cols, rows, pp(bnd), _p(x), pp(v), _p(mn), _p(mx), _p(nr), _p(md), len(x), _p(T), _p(O),
                                  _p(K), _p(sf), _p(isg), len(sf), log_scale_factor, th, float(chi2), _p(bi), _p(bd), device), "orbfe_fuse_search")
    return bi, bd


def search_by_projection_keyframe(kps_cur, desc_cur, cols, rows, kf_angle, valid, p3Dw, min_dist, max_dist, mp_desc, Tcw, Ow, K4, scale_factors,
                                  log_scale_factor, th, orb_dist, taken_cur=None, check_orientation=True, bounds=None, device=0):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1476-1603) -> (nmatches, match_cur)."""
    L = load()
    kc = np.ascontiguousarray(kps_cur, KP_DTYPE); dc = np.ascontiguousarray(desc_cur, np.uint8).reshape(-1, 32)
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx, md, ang = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist), f(mp_desc, np.uint8).reshape(-1, 32), f(kf_angle)
    opt = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
    v, tc, bnd = opt(valid, np.uint8), opt(taken_cur, np.uint8), opt(bounds, np.float32)
    T, O, K, sf = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors)
    m = np.full(len(kc), -1, np.int32); nm = C.c_int32(0)
    pp = lambda a: None if a is None else _p(a)
    _check(L, L.orbfe_search_by_projection_keyframe(_p(kc), _p(dc), len(kc), pp(tc), cols, rows, pp(bnd), len(x), _p(ang), pp(v), _p(x), _p(mn),
                                                    _p(mx), _p(md), _p(T), _p(O), _p(K), _p(sf), len(sf), log_scale_factor, th, int(orb_dist),
                                                    int(check_orientation), _p(m), C.byref(nm), device), "orbfe_search_by_projection_keyframe")
    return nm.value, m


def project_map_points(p3Dw, valid, min_dist, max_dist, normal, Tcw, Ow, K4, cols, rows, scale_factors, log_scale_factor, th, level_below,
                       level_above, strict_max=True, bounds=None, device=0):
    """Projection + gates + PredictScale -> WINDOW_QUERY_DTYPE records (r < 0 = not searched).  strict_max=False selects the
    projection of SearchByProjection(CurrentFrame, pKF, ...) (ORBmatcher.cc:1503-1512), see orbfe.h."""
    L = load()
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist)
    nr = None if normal is None else f(normal).reshape(-1, 3)
    v = None if valid is None else f(valid, np.uint8); bnd = None if bounds is None else f(bounds)
    T, O, K, sf = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors)
    q = np.zeros(len(x), WINDOW_QUERY_DTYPE)
    pp = lambda a: None if a is None else _p(a)
    _check(L, L.orbfe_project_map_points(_p(x), pp(v), _p(mn), _p(mx), pp(nr), len(x), _p(T), _p(O), _p(K), cols, rows, pp(bnd), int(strict_max),
                                         _p(sf), len(sf), log_scale_factor, th, level_below, level_above, _p(q), device), "orbfe_project_map_points")
    return q


def search_by_sim3(kf1, kf2, cols, rows, T1w, T2w, sT12, sT21, K4, scale_factors, log_scale_factor, th, th_high=100, bounds=None, device=0):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1106-1330).  kf = dict(kps, desc, p3Dw, valid, min_dist, max_dist, mp_desc) -> (nFound, match12)."""
    L = load()
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
    m12 = np.full(len(a[0]), -1, np.int32); nf = C.c_int32(0)
    _check(L, L.orbfe_search_by_sim3(_p(a[0]), _p(a[1]), len(a[0]), _p(b[0]), _p(b[1]), len(b[0]), cols, rows, pp(bnd),
                                     _p(a[2]), pp(a[3]), _p(a[4]), _p(a[5]), _p(a[6]), _p(b[2]), pp(b[3]), _p(b[4]), _p(b[5]), _p(b[6]),
                                     _p(Ts[0]), _p(Ts[1]), _p(Ts[2]), _p(Ts[3]), _p(K), _p(sf), len(sf), log_scale_factor, th, th_high,
                                     _p(m12), C.byref(nf), device), "orbfe_search_by_sim3")
    return nf.value, m12


def search_by_projection_sim3(kps, desc, cols, rows, matched, p3Dw, valid, min_dist, max_dist, normal, mp_desc, Tcw, Ow, K4, scale_factors,
                              log_scale_factor, th, bounds=None, device=0):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:294-407) -> (nmatches, match_kf)."""
    L = load()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    f = lambda a, t=np.float32: np.ascontiguousarray(a, t)
    x, mn, mx, nr, md = f(p3Dw).reshape(-1, 3), f(min_dist), f(max_dist), f(normal).reshape(-1, 3), f(mp_desc, np.uint8).reshape(-1, 32)
    v = None if valid is None else f(valid, np.uint8); bnd = None if bounds is None else f(bounds)
    mt = None if matched is None else f(matched, np.uint8)
    T, O, K, sf = f(Tcw).reshape(-1)[:12].copy(), f(Ow), f(K4), f(scale_factors)
    m = np.full(len(k), -1, np.int32); nm = C.c_int32(0)
    pp = lambda a: None if a is None else _p(a)
    _check(L, L.orbfe_search_by_projection_sim3(_p(k), _p(d), len(k), cols, rows, pp(bnd), pp(mt), _p(x), pp(v), _p(mn), _p(mx), _p(nr), _p(md),
                                                len(x), _p(T), _p(O), _p(K), _p(sf), len(sf), log_scale_factor, int(th), _p(m), C.byref(nm),
                                                device), "orbfe_search_by_projection_sim3")
    return nm.value, m


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:41-83) for the rows built so far."""
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self.mfNNratio = nnratio
        self.mbCheckOrientation = checkOri
        self.device = device
        self.L = load()

    @staticmethod
    def DescriptorDistance(a, b):
        return hamming(a, b)

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, cols, rows, vbPrevMatched=None, windowSize=10, bounds=None):
        """Returns (nmatches, vnMatches12, vbPrevMatched'); frames are given as (keypoints, descriptors)."""
        k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        if vbPrevMatched is None:
            vbPrevMatched = np.stack([k1["x"], k1["y"]], 1)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.full(len(k1), -1, np.int32)
        n = C.c_int32(0)
        bnd = None if bounds is None else np.ascontiguousarray(bounds, np.float32)
        _check(self.L, self.L.orbfe_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), cols,
                                                              rows, None if bnd is None else _p(bnd), _p(prev), _p(m12), windowSize, self.mfNNratio,
                                                              int(self.mbCheckOrientation), C.byref(n), self.device),
               "orbfe_search_for_initialization")
        return n.value, m12, prev


# ------------------------------------------------------------------------------------------ ArUco ----
RECT_DTYPE = np.dtype([("corners", "<f4", (4, 2)), ("off", "<i4"), ("len", "<i4")])


def camera_resize(K, cam_size, img_size):
    """CameraParameters::resize (cameraparameters.cpp:158-173); sizes are (width, height)."""
    L = load()
    K4, _ = _camera(K, None)
    out = np.zeros(4, np.float32)
    _check(L, L.orbfe_camera_resize(_p(K4), int(cam_size[0]), int(cam_size[1]), int(img_size[0]), int(img_size[1]), _p(out)),
           "orbfe_camera_resize")
    return out


def marker_poses(markers, marker_size, K, dist, device=0):
    """Both IPPE poses + reprojection errors of every marker (marker.cpp:322-344, ippe.cpp:72-100, Frame.cc:170-174)."""
    L = load()
    K4, d = _camera(K, dist)
    mk = np.ascontiguousarray(markers, MARKER_DTYPE)
    out = np.zeros(len(mk), POSE_DTYPE)
    _check(L, L.orbfe_marker_poses(_p(mk), len(mk), marker_size, _p(K4), _p(d) if len(d) else None, len(d), _p(out), device),
           "orbfe_marker_poses")
    return out


def search_by_bow(kps1, desc1, fv1, kps2, desc2, fv2, valid1=None, valid2=None, nnratio=0.7, check_orientation=True,
                  accept_max=50, factor=30 / 360.0, device=0):
    """ORBmatcher::SearchByBoW on flat arrays (ORBmatcher.cc:159-292; :526-659 with valid2, accept_max=49, factor=1/30).
    fv = (node ids, offsets, feature indices) as ORBVocabulary.transform returns.  -> (nmatches, match12, match21)."""
    L = load()
    k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    a = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    b = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    m12 = np.full(len(k1), -1, np.int32); m21 = np.full(len(k2), -1, np.int32); nm = C.c_int32(0)
    _check(L, L.orbfe_search_by_bow(_p(k1), _p(d1), None if v1 is None else _p(v1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                    _p(k2), _p(d2), None if v2 is None else _p(v2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                    nnratio, int(check_orientation), accept_max, np.float32(factor), _p(m12), _p(m21), C.byref(nm),
                                    device), "orbfe_search_by_bow")
    return nm.value, m12, m21


KF_FEATURE_BYTES = 68


def keyframe_features_pack(kps, desc, mp_index=None, device=0):
    """Feature records of Map::SaveKeyFrame (Map.cc:297-321) -> bytes (n x 68)."""
    L = load()
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    m = None if mp_index is None else np.ascontiguousarray(mp_index, np.uint64)
    out = np.zeros(len(k) * KF_FEATURE_BYTES, np.uint8)
    _check(L, L.orbfe_keyframe_features_pack(_p(k), _p(d), None if m is None else _p(m), len(k), _p(out), device),
           "orbfe_keyframe_features_pack")
    return out


def keyframe_features_unpack(buf, n, device=0):
    """The read side (Map::LoadKeyFrame, Map.cc:478-511) -> (keypoints, descriptors, map point indices)."""
    L = load()
    b = np.ascontiguousarray(buf, np.uint8)
    assert len(b) >= n * KF_FEATURE_BYTES
    k = np.zeros(n, KP_DTYPE); d = np.zeros((n, 32), np.uint8); m = np.zeros(n, np.uint64)
    _check(L, L.orbfe_keyframe_features_unpack(_p(b), n, _p(k), _p(d), _p(m), device), "orbfe_keyframe_features_unpack")
    return k, d, m


def search_for_triangulation(kps1, desc1, fv1, kps2, desc2, fv2, F12, epipole, scale_factors, level_sigma2, has_mp1=None, has_mp2=None,
                             check_orientation=True, device=0):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:661-827, mono) -> (nmatches, match12)."""
    L = load()
    k1 = np.ascontiguousarray(kps1, KP_DTYPE); k2 = np.ascontiguousarray(kps2, KP_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    a = [np.ascontiguousarray(fv1[0], np.uint32), np.ascontiguousarray(fv1[1], np.int32), np.ascontiguousarray(fv1[2], np.uint32)]
    b = [np.ascontiguousarray(fv2[0], np.uint32), np.ascontiguousarray(fv2[1], np.int32), np.ascontiguousarray(fv2[2], np.uint32)]
    h1 = None if has_mp1 is None else np.ascontiguousarray(has_mp1, np.uint8)
    h2 = None if has_mp2 is None else np.ascontiguousarray(has_mp2, np.uint8)
    F = np.ascontiguousarray(F12, np.float32).reshape(9); sf = np.ascontiguousarray(scale_factors, np.float32)
    sg = np.ascontiguousarray(level_sigma2, np.float32)
    m12 = np.full(len(k1), -1, np.int32); nm = C.c_int32(0)
    _check(L, L.orbfe_search_for_triangulation(_p(k1), _p(d1), None if h1 is None else _p(h1), len(k1), _p(a[0]), _p(a[1]), _p(a[2]), len(a[0]),
                                               _p(k2), _p(d2), None if h2 is None else _p(h2), len(k2), _p(b[0]), _p(b[1]), _p(b[2]), len(b[0]),
                                               _p(F), np.float32(epipole[0]), np.float32(epipole[1]), _p(sf), _p(sg), len(sf),
                                               int(check_orientation), _p(m12), C.byref(nm), device), "orbfe_search_for_triangulation")
    return nm.value, m12


class ORBVocabulary:
    """Mirror of ORB_SLAM2::ORBVocabulary (include/ORBVocabulary.h: DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>) for
    the per-frame path: loadFromTextFile + transform(features, BowVector, FeatureVector, levelsup)."""

    def __init__(self, device=0):
        self.L = load()
        self.h = None
        self.device = device

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbfe_vocabulary_destroy(self.h)
            self.h = None

    def loadFromTextFile(self, filename):
        if self.h:
            self.L.orbfe_vocabulary_destroy(self.h)
        self.h = self.L.orbfe_vocabulary_load_text(os.fsencode(filename), self.device)
        if not self.h:
            raise OrbfeError("orbfe_vocabulary_load_text: " + self.L.orbfe_last_error().decode())
        return True

    @classmethod
    def from_arrays(cls, k, L, scoring, weighting, parent, is_leaf, descriptors, weights, device=0):
        self = cls(device)
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        descriptors = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        weights = np.ascontiguousarray(weights, np.float64)
        self.h = self.L.orbfe_vocabulary_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(descriptors),
                                                _p(weights), device)
        if not self.h:
            raise OrbfeError("orbfe_vocabulary_create: " + self.L.orbfe_last_error().decode())
        return self

    def info(self):
        out = np.zeros(6, np.int32)
        _check(self.L, self.L.orbfe_vocabulary_info(self.h, _p(out)), "orbfe_vocabulary_info")
        return dict(zip(("k", "L", "scoring", "weighting", "nodes", "words"), out.tolist()))

    def transform(self, descriptors, levelsup=4):
        """-> dict(word, node, weight per feature; bow = (word ids, values); fv = (node ids, offsets, feature indices))"""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
        bw = np.zeros(n, np.uint32); bv = np.zeros(n, np.float64); fn = np.zeros(n, np.uint32)
        fo = np.zeros(n + 1, np.int32); ff = np.zeros(n, np.uint32)
        nb, nf = C.c_int32(0), C.c_int32(0)
        _check(self.L, self.L.orbfe_vocabulary_transform(self.h, _p(d), n, levelsup, _p(word), _p(node), _p(weight), _p(bw), _p(bv),
                                                         C.byref(nb), _p(fn), _p(fo), _p(ff), C.byref(nf)),
               "orbfe_vocabulary_transform")
        nb, nf = nb.value, nf.value
        return dict(word=word, node=node, weight=weight, bow=(bw[:nb].copy(), bv[:nb].copy()),
                    fv=(fn[:nf].copy(), fo[:nf + 1].copy(), ff[:fo[nf]].copy()))


def corner_subpix(image, pts, win, max_iters, eps, device=0):
    """cv::cornerSubPix(image, pts, Size(win, win), Size(-1, -1), TermCriteria(MAX_ITER | EPS, max_iters, eps)) -> refined (n, 2) float32."""
    L = load()
    image = np.ascontiguousarray(image, np.uint8)
    out = np.ascontiguousarray(pts, np.float32).reshape(-1, 2).copy()
    _check(L, L.orbfe_corner_subpix(_p(image), image.shape[0], image.shape[1], image.strides[0], _p(out), len(out), int(win), int(max_iters),
                                    float(eps), device), "orbfe_corner_subpix")
    return out


class MarkerDetector:
    """Mirror of aruco::MarkerDetector as the reference configures it (src/Frame.cc:129-142):
    setDictionary(name) + DM_NORMAL + CORNER_LINES; detect(image) -> markers sorted by id."""
    # order of kernel_times_us(); the pyramid runs on the detector's second stream, concurrently with threshold + contours
    STAGES = ["pyramid", "threshold", "contours", "decode", "finalize"]

    def __init__(self, dictionary="ARUCO", device=0):
        self.L = load()
        self.h = self.L.orbfe_aruco_create(dictionary.encode(), device)
        if not self.h:
            raise OrbfeError("orbfe_aruco_create: " + self.L.orbfe_last_error().decode())
        self.capacity = self.L.orbfe_aruco_max_markers(self.h)
        self._shape = None

    @classmethod
    def wrap(cls, handle):
        """A non-owning view of a detector another object owns (the detector of an orbfe_pipeline)."""
        self = cls.__new__(cls)
        self.L = load()
        self.h = handle
        self._borrowed = True
        self.capacity = self.L.orbfe_aruco_max_markers(self.h)
        self._shape = None
        return self

    def __del__(self):
        if getattr(self, "h", None) and not getattr(self, "_borrowed", False):
            self.L.orbfe_aruco_destroy(self.h)
        self.h = None

    DM_NORMAL, DM_FAST, DM_VIDEO_FAST = 0, 1, 2              # aruco::DetectionMode (markerdetector.h:60)
    CORNER_SUBPIX, CORNER_LINES, CORNER_NONE = 0, 1, 2       # aruco::CornerRefinementMethod (:62)

    def setDictionary(self, name, error_correction_rate=0.0):
        _check(self.L, self.L.orbfe_aruco_set_dictionary(self.h, name.encode()), "orbfe_aruco_set_dictionary")
        _check(self.L, self.L.orbfe_aruco_set_error_correction_rate(self.h, error_correction_rate), "orbfe_aruco_set_error_correction_rate")

    def setDetectionMode(self, dm, minMarkerSize=0.0):
        _check(self.L, self.L.orbfe_aruco_set_detection_mode(self.h, int(dm), minMarkerSize), "orbfe_aruco_set_detection_mode")

    def setCornerRefinementMethod(self, method):
        _check(self.L, self.L.orbfe_aruco_set_corner_refinement(self.h, int(method)), "orbfe_aruco_set_corner_refinement")

    def detectEnclosedMarkers(self, on=True):
        """Params::detectEnclosedMarkers (markerdetector.h:126)."""
        _check(self.L, self.L.orbfe_aruco_set_enclosed_markers(self.h, int(on)), "orbfe_aruco_set_enclosed_markers")

    def setTracking(self, min_detections):
        """Params::trackingMinDetections (markerdetector.h:187); resets the history."""
        _check(self.L, self.L.orbfe_aruco_set_tracking(self.h, int(min_detections)), "orbfe_aruco_set_tracking")

    def tracked(self):
        return self.L.orbfe_aruco_last_tracked(self.h)

    def setGrayConversion(self, fractional_bits):
        """cvtColor(BGR2GRAY) of CV_8UC3 frames: 14 fractional bits (OpenCV <= 3.4.1, default) or 15 (3.4.2 and later)."""
        _check(self.L, self.L.orbfe_aruco_set_gray_conversion(self.h, int(fractional_bits)), "orbfe_aruco_set_gray_conversion")

    def state(self):
        """Params::ThresHold and Params::minSize as the last call left them, its threshold passes and working image size."""
        t, a, r, c, m = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_float(0)
        _check(self.L, self.L.orbfe_aruco_get_state(self.h, C.byref(t), C.byref(m), C.byref(a), C.byref(r), C.byref(c)), "orbfe_aruco_get_state")
        return {"threshold": t.value, "min_size": m.value, "attempts": a.value, "work_shape": (r.value, c.value)}

    def contour(self, marker, frame=0):
        """aruco::Marker::contourPoints of output marker `marker` of the last call -> (n, 2) int32."""
        n = C.c_int32(0)
        _check(self.L, self.L.orbfe_aruco_marker_contour(self.h, frame, marker, None, 0, C.byref(n)), "orbfe_aruco_marker_contour")
        xy = np.zeros((max(n.value, 1), 2), np.int32)
        _check(self.L, self.L.orbfe_aruco_marker_contour(self.h, frame, marker, _p(xy), n.value, C.byref(n)), "orbfe_aruco_marker_contour")
        return xy[:n.value]

    def contours(self, nmarkers, frame=0):
        """contourPoints of the first `nmarkers` output markers of `frame` in one round trip -> list of (n, 2) int32."""
        off = np.zeros(nmarkers + 1, np.int32)
        _check(self.L, self.L.orbfe_aruco_marker_contours(self.h, frame, nmarkers, None, 0, _p(off)), "orbfe_aruco_marker_contours")
        xy = np.zeros((max(int(off[-1]), 1), 2), np.int32)
        _check(self.L, self.L.orbfe_aruco_marker_contours(self.h, frame, nmarkers, _p(xy), int(off[-1]), _p(off)), "orbfe_aruco_marker_contours")
        return [xy[off[i]:off[i + 1]] for i in range(nmarkers)]

    def detect(self, image, camera=None, markerSizeMeters=-1.0):
        """detect(image) -> MARKER_DTYPE records.  With camera = (K, dist, (cam_width, cam_height)) and a marker size
        (markerdetector.h:276-312; Frame.cc:142 passes 0.187) -> (markers, POSE_DTYPE poses): the camera matrix is
        rescaled to the image size first (markerdetector_impl.cpp:1110-1172), then every marker gets its IPPE pose."""
        if camera is not None and markerSizeMeters > 0:
            K, dist, cam_size = camera
            K4, d = _camera(K, dist)
            image = np.asarray(image)
            if image.size == 0:
                return np.zeros(0, MARKER_DTYPE), np.zeros(0, POSE_DTYPE)
            bgr = image.ndim == 3 and image.shape[2] == 3   # CV_8UC3: cvtColor(BGR2GRAY) first (markerdetector_impl.cpp:5892)
            assert image.dtype == np.uint8 and (image.ndim == 2 or bgr)
            if image.strides[1] != (3 if bgr else 1) or (bgr and image.strides[2] != 1):
                image = np.ascontiguousarray(image)
            rows, cols = image.shape[:2]
            K4 = np.ascontiguousarray(camera_resize(K4, cam_size, (cols, rows)), np.float32)
            out = np.zeros(self.capacity, MARKER_DTYPE)
            poses = np.zeros(self.capacity, POSE_DTYPE)
            n = C.c_int32(0)
            fn = self.L.orbfe_aruco_detect_poses_bgr if bgr else self.L.orbfe_aruco_detect_poses
            _check(self.L, fn(self.h, _p(image), rows, cols, image.strides[0], _p(out), _p(poses), self.capacity,
                                                           C.byref(n), markerSizeMeters, _p(K4), _p(d) if d is not None and len(d) else None,
                                                           0 if d is None else len(d)), "orbfe_aruco_detect_poses")
            self._shape = image.shape[:2]
            return out[:n.value].copy(), poses[:n.value].copy()
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, MARKER_DTYPE)
        bgr = image.ndim == 3 and image.shape[2] == 3
        assert image.dtype == np.uint8 and (image.ndim == 2 or bgr)
        if image.strides[1] != (3 if bgr else 1) or (bgr and image.strides[2] != 1):
            image = np.ascontiguousarray(image)
        out = np.zeros(self.capacity, MARKER_DTYPE)
        n = C.c_int32(0)
        fn = self.L.orbfe_aruco_detect_bgr if bgr else self.L.orbfe_aruco_detect
        _check(self.L, fn(self.h, _p(image), image.shape[0], image.shape[1], image.strides[0],
                          _p(out), self.capacity, C.byref(n)), "orbfe_aruco_detect")
        self._shape = image.shape[:2]
        return out[:n.value].copy()

    def detect_batch(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        B, rows, cols = images.shape
        out = np.zeros((B, self.capacity), MARKER_DTYPE)
        n = np.zeros(B, np.int32)
        _check(self.L, self.L.orbfe_aruco_detect_batch(self.h, _p(images), B, images.strides[0], rows, cols,
                                                       images.strides[1], _p(out), self.capacity, _p(n)),
               "orbfe_aruco_detect_batch")
        self._shape = (rows, cols)
        return [out[f, :n[f]].copy() for f in range(B)]

    def detect_batch_device(self, d_imgs_ptr, B, frame_stride, rows, cols, step, d_out_ptr, capacity, d_n_ptr,
                            stream=0):
        _check(self.L, self.L.orbfe_aruco_detect_batch_device(self.h, d_imgs_ptr, B, frame_stride, rows, cols, step,
                                                              d_out_ptr, capacity, d_n_ptr, stream),
               "orbfe_aruco_detect_batch_device")

    def batch_status(self):
        """(frames of the last device batch with incomplete results, union of their capacity flags)"""
        n, fl = C.c_int32(0), C.c_int32(0)
        _check(self.L, self.L.orbfe_aruco_batch_status(self.h, C.byref(n), C.byref(fl)), "orbfe_aruco_batch_status")
        return n.value, fl.value

    def set_big_frames(self, on=True):
        _check(self.L, self.L.orbfe_aruco_set_big_frames(self.h, int(on)), "orbfe_aruco_set_big_frames")

    # stage read-back for parity tests
    def thresholded(self, frame=0):
        out = np.zeros(self._shape, np.uint8)
        _check(self.L, self.L.orbfe_aruco_debug_image(self.h, frame, 0, _p(out)), "debug_image")
        return out

    def pyramid_level(self, level, frame=0):
        """Debug: level >= 1 of the detector's /2 pyramid of the last batch (None past the last level)."""
        out = np.zeros(self._shape, np.uint8)
        if self.L.orbfe_aruco_debug_image(self.h, frame, level + 1, _p(out)) != 0:
            return None
        h, w = self._shape
        for _ in range(level):
            w, h = w // 2, h // 2        # exact halves only (the levels this accessor is for)
        return out.reshape(-1)[:w * h].reshape(h, w).copy()

    def counts(self, frame=0):
        out = np.zeros(4, np.int32)
        _check(self.L, self.L.orbfe_aruco_debug_image(self.h, frame, 100, _p(out)), "debug_image")
        return dict(nkept=int(out[0]), nrect=int(out[1]), flags=int(out[2]), ncand=int(out[3]) & 0x3fffffff,
                    fell_back=bool(int(out[3]) >> 30))   # the relay contour kernel handed the frame to the legacy one

    def force_legacy_contours(self, on=True):
        """Debug: run every frame through the single-walker contour kernel (the relay kernel's fallback)."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 2 if on else 3)

    def set_tiled_contours(self, mode):
        """Debug: the tiled contour path (aruco_tiles.hip) None = by frame / batch size (default), True = every batch, False = never."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 4 if mode is None else 5 if mode else 6)

    def set_speck_passes(self, on=True):
        """Debug: the speck passes between threshold and contours (k_speck_clean) on / off (default); the results do not change."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 8 if on else 9)

    def set_threshold_on_matrix_cores(self, on=True):
        """True: k_threshold_mfma wherever it applies (windows up to 15); False: the dot-product kernels; None: the default rule -- the
        matrix-core kernel for calls of 8 frames and more, k_threshold_pyr (one launch instead of five) for the drop-in call's few."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 16 if on is None else 14 if on else 15)

    def set_threshold_pyramid_kernel(self, on=True):
        """k_threshold_pyr (threshold + the /2 pyramid levels a tile holds, the default where it applies) / k_adaptive_threshold_t + k_half_area4."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 12 if on else 13)

    def set_half_pyramid_kernel(self, on=True):
        """k_half_pyr (the leading exact /2 levels in one launch, default) / one k_half_area4 launch a level."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 18 if on else 19)

    def set_speck_passes_in_kernel(self, on=True):
        """Debug: the speck passes inside the one-workgroup relay kernels (full batches of frames whose bit image fits LDS) on / off (default)."""
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, 10 if on else 11)

    def contour_image(self, frame=0):
        """Debug: the bit image the contour kernels of the last batch read from HBM (the thresholded image after the speck passes where
        they ran as a launch; with the in-kernel passes the cleaned image only exists in LDS and this is the thresholded image)."""
        out = np.zeros(self._shape, np.uint8)
        _check(self.L, self.L.orbfe_aruco_debug_image(self.h, frame, 104, _p(out)), "debug_image")
        return out

    def contour_retries(self):
        """Debug: how many batches of this detector were done again on the next contour path (tiled -> one workgroup -> single walker)
        because a frame exceeded a capacity of the one they ran on."""
        return int(self.L.orbfe_aruco_debug_kernel_times(self.h, None, 7))

    def rects(self, frame=0):
        out = np.zeros(self.capacity, RECT_DTYPE)
        _check(self.L, self.L.orbfe_aruco_debug_image(self.h, frame, 101, _p(out)), "debug_image")
        return out[:self.counts(frame)["nrect"]]

    def set_aux_stream(self, stream_ptr):
        """Run the detector's forked launches (pyramid) on the caller's stream (None: the handle's own)."""
        _check(self.L, self.L.orbfe_aruco_set_aux_stream(self.h, stream_ptr), "set_aux_stream")

    def enable_kernel_timing(self, on=True):
        self.L.orbfe_aruco_debug_kernel_times(self.h, None, int(on))

    def kernel_times_us(self, median=False):
        """Stage times of the last batch, or (median=True) the per-stage median over the batches since timing was enabled."""
        out = np.zeros(32, np.float32)
        n = self.L.orbfe_aruco_debug_kernel_times(self.h, _p(out), -32 if median else 32)
        return out[:n]

    @staticmethod
    def algorithmic_bytes(rows, cols):
        """Per-frame algorithmic bytes of each ArUco stage (terms of SURVEY 8d's B_aruco ~ 3.33 W H)."""
        wh = rows * cols
        return {"aruco_threshold": wh + wh // 8, "aruco_pyramid": wh + wh // 3, "aruco_contours": wh // 8,
                "aruco_decode": 0, "aruco_finalize": 0}
