"""The C++ shims of include/shims/ are real code: they compile (-Wall -Werror) against minimal mock OpenCV / SLAM headers
(tests/mock_cv/, test infrastructure), link against liborbfe.so, and without a GPU the driver stops at the library's loud
"no device" answer instead of falling back to anything."""
import subprocess

import pytest

import shim_build


def test_shims_compile_link_and_refuse_without_a_gpu(tmp_path):
    from orb_slam2_aruco_amd import binding
    exe = shim_build.build(str(tmp_path))
    if binding.load().orbfe_device_count() > 0:
        pytest.skip("a GPU is present: tests/test_shims_gpu.py runs the driver")
    r = subprocess.run([exe, "none", "1", "1", "1", str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


def test_shim_headers_declare_the_reference_interfaces():
    """Signatures the call sites of the reference need (ORBextractor.h:59-83, markerdetector.h:222-312, ORBmatcher.h:44-83)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex = open(os.path.join(root, "include", "shims", "ORBextractor.h")).read()
    for sig in ("ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)",
                "void operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)",
                "GetLevels()", "GetScaleFactor()", "GetScaleFactors()", "GetInverseScaleFactors()", "GetScaleSigmaSquares()",
                "GetInverseScaleSigmaSquares()", "std::vector<cv::Mat> mvImagePyramid"):
        assert sig in ex, sig
    md = open(os.path.join(root, "include", "shims", "MarkerDetector.h")).read()
    for sig in ("MarkerDetector(std::string dict_type, float error_correction_rate = 0)",
                "void setDictionary(std::string dict_type, float error_correction_rate = 0)",
                "void setDetectionMode(DetectionMode dm, float minMarkerSize = 0)", "Params& getParameters()",
                "std::vector<aruco::Marker> detect(const cv::Mat& input)",
                "std::vector<aruco::Marker> detect(const cv::Mat& input, const CameraParameters& camParams, float markerSizeMeters,",
                "void detect(const cv::Mat& input, std::vector<Marker>& detectedMarkers, CameraParameters camParams, float markerSizeMeters = -1,",
                "void detect(const cv::Mat& input, std::vector<Marker>& detectedMarkers, cv::Mat camMatrix = cv::Mat(), cv::Mat distCoeff = cv::Mat(),"):
        assert sig in md, sig
    om = open(os.path.join(root, "include", "shims", "ORBmatcher_orbfe.cc")).read()
    for name in ("DescriptorDistance", "SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints",
                 "SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame", "SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF",
                 "SearchByProjection(KeyFrame* pKF, cv::Mat Scw", "SearchByBoW(KeyFrame* pKF, Frame& F", "SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2",
                 "SearchForInitialization(Frame& F1, Frame& F2", "SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2", "SearchBySim3(KeyFrame* pKF1",
                 "Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints", "Fuse(KeyFrame* pKF, cv::Mat Scw"):
        assert "ORBmatcher::" + name in om, name
