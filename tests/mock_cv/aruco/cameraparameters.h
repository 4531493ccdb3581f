// TEST INFRASTRUCTURE ONLY -- aruco::CameraParameters as far as the detector shim reads it (cameraparameters.h:42-75, :158-173)
#ifndef MOCK_ARUCO_CAMERAPARAMETERS_H
#define MOCK_ARUCO_CAMERAPARAMETERS_H
#include <opencv2/core/core.hpp>
namespace aruco {
class CameraParameters {
public:
    cv::Mat CameraMatrix, Distorsion;
    cv::Size CamSize = cv::Size(-1, -1);
    bool isValid() const { return CameraMatrix.rows != 0 && CameraMatrix.cols != 0 && Distorsion.rows != 0 && Distorsion.cols != 0 && CamSize.width != -1 && CamSize.height != -1; }
    void resize(cv::Size size)
    {
        if (size == CamSize) return;
        const float AxFactor = float(size.width) / float(CamSize.width), AyFactor = float(size.height) / float(CamSize.height);
        CameraMatrix = CameraMatrix.clone();
        CameraMatrix.at<float>(0, 0) *= AxFactor; CameraMatrix.at<float>(0, 2) *= AxFactor;
        CameraMatrix.at<float>(1, 1) *= AyFactor; CameraMatrix.at<float>(1, 2) *= AyFactor;
        CamSize = size;
    }
};
}
#endif
