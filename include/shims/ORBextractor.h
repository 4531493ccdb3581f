/*
 * ORBextractor.h (shim) -- ORB_SLAM2::ORBextractor with the reference's public interface (include/ORBextractor.h:45-113),
 * implemented on liborbfe.so.  Drop it in place of the reference header and leave src/ORBextractor.cc out of the build:
 * Frame::ExtractORB (src/Frame.cc:200-206), the Frame constructor's getter calls (:82-88) and Tracking's
 * `new ORBextractor(...)` compile unchanged.  Needs OpenCV's core header, like the class it replaces.
 */
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <cassert>
#include <list>
#include <stdexcept>
#include <vector>

#include <opencv2/core/core.hpp>

#include "orbfe.h"

namespace ORB_SLAM2
{

class ORBextractor
{
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : h_(orbfe_extractor_create(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, /*device*/ 0))
    {
        if (!h_) throw std::runtime_error(orbfe_last_error());
        cap_ = orbfe_extractor_max_keypoints(h_);
        nlevels_ = orbfe_extractor_get_levels(h_);
    }
    ~ORBextractor() { orbfe_extractor_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored, as in the reference (ORBextractor.cc:1043-1105).
    void operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return; // :1046: outputs untouched
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1); // :1050
        static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint is 28 bytes: pt, size, angle, response, octave, class_id");
        _keypoints.resize(cap_);
        std::vector<unsigned char> desc((size_t)cap_ * 32);
        int32_t n = 0;
        const int rc = orbfe_extract(h_, image.data, image.rows, image.cols, image.step, reinterpret_cast<orbfe_keypoint*>(_keypoints.data()),
                                     desc.data(), cap_, &n);
        if (rc != ORBFE_OK) throw std::runtime_error(orbfe_last_error());
        _keypoints.resize(n);
        if (n == 0) { _descriptors.release(); return; } // :1064-1065
        _descriptors.create(n, 32, CV_8U);              // :1068
        cv::Mat descriptors = _descriptors.getMat();
        for (int i = 0; i < n; i++) std::copy(desc.begin() + (size_t)i * 32, desc.begin() + (size_t)(i + 1) * 32, descriptors.ptr(i));
        if (keep_pyramid_) { // ComputePyramid's result (:1107-1132) read back from the device, level by level (without the 19-pixel borders)
            mvImagePyramid.resize(nlevels_);
            for (int l = 0; l < nlevels_; l++) {
                int w = 0, hgt = 0;
                if (orbfe_extractor_debug_level_size(h_, l, &w, &hgt) != ORBFE_OK) throw std::runtime_error(orbfe_last_error());
                mvImagePyramid[l].create(hgt, w, CV_8UC1);
                if (orbfe_extractor_debug_level_image(h_, 0, l, 0, mvImagePyramid[l].data) != ORBFE_OK) throw std::runtime_error(orbfe_last_error());
            }
        }
    }

    int inline GetLevels() { return nlevels_; }
    float inline GetScaleFactor() { return orbfe_extractor_get_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return table(orbfe_extractor_get_scale_factors); }
    std::vector<float> inline GetInverseScaleFactors() { return table(orbfe_extractor_get_inverse_scale_factors); }
    std::vector<float> inline GetScaleSigmaSquares() { return table(orbfe_extractor_get_scale_sigma_squares); }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return table(orbfe_extractor_get_inverse_scale_sigma_squares); }

    // Only read by the stereo code path (Frame.cc:456, :546-563), which the monocular system never runs: left empty unless asked for
    // -- keepImagePyramid(true) makes operator() read the levels back from the device (eight copies per frame: not for the hot path).
    std::vector<cv::Mat> mvImagePyramid;
    void keepImagePyramid(bool on) { keep_pyramid_ = on; if (!on) mvImagePyramid.clear(); }

    orbfe_extractor* handle() { return h_; } // for the batched-video entry points (orbfe_extract_batch_device)

private:
    std::vector<float> table(int (*get)(const orbfe_extractor*, float*))
    {
        std::vector<float> v(nlevels_);
        get(h_, v.data());
        return v;
    }
    orbfe_extractor* h_;
    int cap_ = 0, nlevels_ = 0;
    bool keep_pyramid_ = false;
};

} // namespace ORB_SLAM2

#endif
