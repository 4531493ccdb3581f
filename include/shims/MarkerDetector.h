/*
 * MarkerDetector.h (shim) -- aruco::MarkerDetector with the reference's public interface
 * (Thirdparty/aruco/aruco/markerdetector.h:46-312) for what ORB_SLAM2_aruco calls (src/Frame.cc:129-142), implemented on
 * liborbfe.so.  aruco::Marker and aruco::CameraParameters stay the library's own classes (marker.h, cameraparameters.h): they are
 * data holders, and Marker::calculateExtrinsics is replaced by the poses the detector returns.  Include this header instead of
 * "markerdetector.h" and leave markerdetector.cpp / markerdetector_impl.cpp / dictionary_based.cpp out of the build.
 */
#ifndef _ARUCO_MarkerDetector_H
#define _ARUCO_MarkerDetector_H

#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>

#include "cameraparameters.h"
#include "marker.h"
#include "orbfe.h"

namespace aruco
{

enum DetectionMode : int { DM_NORMAL = 0, DM_FAST = 1, DM_VIDEO_FAST = 2 };          // markerdetector.h:60
enum CornerRefinementMethod : int { CORNER_SUBPIX = 0, CORNER_LINES = 1, CORNER_NONE = 2 }; // :62

class MarkerDetector
{
public:
    enum ThresMethod : int { THRES_ADAPTIVE = 0, THRES_AUTO_FIXED = 1 };

    // the operating parameters ORB_SLAM2_aruco touches; every setter forwards to the library and fails as loudly as it does
    struct Params
    {
        void setDetectionMode(DetectionMode dm, float minMarkerSize)
        {
            // the library keeps the parameters and their frame-to-frame state (THRES_AUTO_FIXED threshold, automatic minSize); the
            // members below are the caller-visible copy (a following setCornerRefinementMethod(CORNER_LINES / CORNER_NONE) resets
            // minSize to 0 on both sides as in the reference, markerdetector.cpp:374-395)
            check(orbfe_aruco_set_detection_mode(owner->handle(), (int)dm, minMarkerSize));
            detectMode = dm;
            minSize = minMarkerSize;
        }
        void detectEnclosedMarkers(bool do_) // markerdetector.h:126
        {
            check(orbfe_aruco_set_enclosed_markers(owner->handle(), do_ ? 1 : 0));
            enclosedMarker = do_;
        }
        void setTrackingMinDetections(int n) // the reference assigns Params::trackingMinDetections directly (markerdetector.h:187): call this instead
        {
            check(orbfe_aruco_set_tracking(owner->handle(), n));
            trackingMinDetections = n;
        }
        void setCornerRefinementMethod(CornerRefinementMethod method)
        {
            check(orbfe_aruco_set_corner_refinement(owner->handle(), (int)method));
            cornerRefinementM = method;
            if (method != CORNER_SUBPIX) minSize = 0; // markerdetector.cpp:399-402
        }
        DetectionMode detectMode = DM_NORMAL;
        CornerRefinementMethod cornerRefinementM = CORNER_LINES;
        float minSize = 0;
        bool enclosedMarker = false;
        int trackingMinDetections = 0;
        float error_correction_rate = 0;
        std::string dictionary = "ARUCO";
        int maxThreads = 1;

    private:
        friend class MarkerDetector;
        MarkerDetector* owner = nullptr;
        static void check(int rc)
        {
            if (rc != ORBFE_OK) throw cv::Exception(9001, orbfe_last_error(), "MarkerDetector::Params", __FILE__, __LINE__);
        }
    };

    MarkerDetector() { _params.owner = this; }
    MarkerDetector(std::string dict_type, float error_correction_rate = 0)
    {
        _params.owner = this;
        setDictionary(dict_type, error_correction_rate);
    }
    ~MarkerDetector() { orbfe_aruco_destroy(h_); }
    MarkerDetector(const MarkerDetector&) = delete;
    MarkerDetector& operator=(const MarkerDetector&) = delete;

    // markerdetector.h:330-348: predefined dictionaries by name; an unknown name is treated as a file by the reference and throws
    // std::runtime_error (dictionary.cpp:44-62)
    void setDictionary(std::string dict_type, float error_correction_rate = 0)
    {
        if (!h_) h_ = orbfe_aruco_create(dict_type.c_str(), /*device*/ 0);
        else if (orbfe_aruco_set_dictionary(h_, dict_type.c_str()) != ORBFE_OK) throw std::runtime_error(orbfe_last_error());
        if (!h_) throw std::runtime_error(orbfe_last_error());
        if (orbfe_aruco_set_error_correction_rate(h_, error_correction_rate) != ORBFE_OK) throw std::runtime_error(orbfe_last_error());
        _params.dictionary = dict_type;
        _params.error_correction_rate = error_correction_rate;
    }

    void setDetectionMode(DetectionMode dm, float minMarkerSize = 0) { _params.setDetectionMode(dm, minMarkerSize); }
    DetectionMode getDetectionMode() { return _params.detectMode; }
    Params getParameters() const { return _params; }
    Params& getParameters() { return _params; }

    std::vector<aruco::Marker> detect(const cv::Mat& input)
    {
        std::vector<Marker> detectedMarkers;
        detect(input, detectedMarkers, cv::Mat(), cv::Mat(), -1, false);
        return detectedMarkers;
    }
    std::vector<aruco::Marker> detect(const cv::Mat& input, const CameraParameters& camParams, float markerSizeMeters,
                                      bool setYPerperdicular = false)
    {
        std::vector<Marker> detectedMarkers;
        detect(input, detectedMarkers, camParams, markerSizeMeters, setYPerperdicular);
        return detectedMarkers;
    }
    void detect(const cv::Mat& input, std::vector<Marker>& detectedMarkers, CameraParameters camParams, float markerSizeMeters = -1,
                bool setYPerperdicular = false)
    {
        if (camParams.CamSize != input.size() && camParams.isValid() && markerSizeMeters > 0)
        {
            // the camera matrix is rescaled to the image size (markerdetector_impl.cpp:1110-1172 -> CameraParameters::resize)
            CameraParameters cp_aux = camParams;
            cp_aux.resize(input.size());
            detect(input, detectedMarkers, cp_aux.CameraMatrix, cp_aux.Distorsion, markerSizeMeters, setYPerperdicular);
        }
        else
            detect(input, detectedMarkers, camParams.CameraMatrix, camParams.Distorsion, markerSizeMeters, setYPerperdicular);
    }
    // NOTE (as in the reference): the camera matrix must be the one of this image size
    void detect(const cv::Mat& input, std::vector<Marker>& detectedMarkers, cv::Mat camMatrix = cv::Mat(), cv::Mat distCoeff = cv::Mat(),
                float markerSizeMeters = -1, bool setYPerperdicular = false)
    {
        if (!h_) setDictionary(_params.dictionary, _params.error_correction_rate);
        // CV_8UC1 (what Frame.cc:142 passes) or CV_8UC3, converted with BGR2GRAY on the device (markerdetector_impl.cpp:5892)
        if (input.type() != CV_8UC1 && input.type() != CV_8UC3)
            throw cv::Exception(9001, "MarkerDetector::detect takes CV_8UC1 or CV_8UC3 images", "MarkerDetector::detect", __FILE__, __LINE__);
        const bool bgr = input.type() == CV_8UC3;
        const int cap = orbfe_aruco_max_markers(h_);
        std::vector<orbfe_marker> m(cap);
        std::vector<orbfe_marker_pose> poses;
        int32_t n = 0;
        // with a camera and a marker size (what Frame.cc:142 passes) detection and the IPPE poses are one library call
        const bool want_pose = camMatrix.rows != 0 && markerSizeMeters > 0;
        if (want_pose)
        {
            cv::Mat K32, D32;
            camMatrix.convertTo(K32, CV_32F);
            distCoeff.convertTo(D32, CV_32F);
            const float K4[4] = {K32.at<float>(0, 0), K32.at<float>(1, 1), K32.at<float>(0, 2), K32.at<float>(1, 2)};
            poses.resize(cap);
            if ((bgr ? orbfe_aruco_detect_poses_bgr : orbfe_aruco_detect_poses)(h_, input.data, input.rows, input.cols, input.step, m.data(), poses.data(), cap, &n, markerSizeMeters,
                                         K4, D32.empty() ? nullptr : D32.ptr<float>(), (int)D32.total()) != ORBFE_OK)
                throw cv::Exception(9001, orbfe_last_error(), "MarkerDetector::detect", __FILE__, __LINE__);
        }
        else if ((bgr ? orbfe_aruco_detect_bgr : orbfe_aruco_detect)(h_, input.data, input.rows, input.cols, input.step, m.data(), cap, &n) != ORBFE_OK)
            throw cv::Exception(9001, orbfe_last_error(), "MarkerDetector::detect", __FILE__, __LINE__);
        detectedMarkers.clear();
        detectedMarkers.resize(n);
        // Marker::contourPoints (marker.h:56) of all markers of the frame in one round trip
        std::vector<int32_t> off((size_t)n + 1, 0), xy;
        if (n > 0 && orbfe_aruco_marker_contours(h_, 0, n, nullptr, 0, off.data()) == ORBFE_OK && off[n] > 0)
        {
            xy.resize((size_t)off[n] * 2);
            if (orbfe_aruco_marker_contours(h_, 0, n, xy.data(), off[n], off.data()) != ORBFE_OK) xy.clear();
        }
        for (int i = 0; i < n; i++)
        {
            Marker& M = detectedMarkers[i];
            M.id = m[i].id;
            M.dict_info = _params.dictionary;
            for (int k = 0; k < 4; k++) M.push_back(cv::Point2f(m[i].corners[k][0], m[i].corners[k][1]));
            if (!xy.empty())
            {
                const int len = off[i + 1] - off[i];
                M.contourPoints.resize(len);
                for (int j = 0; j < len; j++) M.contourPoints[j] = cv::Point(xy[2 * (off[i] + j)], xy[2 * (off[i] + j) + 1]);
            }
        }
        // detect the position of detected markers if desired (markerdetector_impl.cpp:8720-8780 -> Marker::calculateExtrinsics)
        if (want_pose && n > 0)
        {
            for (int i = 0; i < n; i++)
            {
                Marker& M = detectedMarkers[i];
                if (setYPerperdicular)
                {
                    // Marker::rotateXAxis is private: the rarely used option goes through the library's own Marker class
                    // (marker.cpp stays in the build), which is the reference path itself
                    M.calculateExtrinsics(markerSizeMeters, camMatrix, distCoeff, true);
                    continue;
                }
                M.Rvec.create(3, 1, CV_32F);
                M.Tvec.create(3, 1, CV_32F);
                for (int k = 0; k < 3; k++) { M.Rvec.at<float>(k, 0) = poses[i].rvec[k]; M.Tvec.at<float>(k, 0) = poses[i].tvec[k]; }
                M.ssize = markerSizeMeters;
            }
        }
    }

    orbfe_aruco* handle()
    {
        if (!h_) setDictionary(_params.dictionary, _params.error_correction_rate);
        return h_;
    }

private:
    orbfe_aruco* h_ = nullptr;
    Params _params;
};

} // namespace aruco

#endif
