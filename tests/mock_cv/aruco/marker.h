// TEST INFRASTRUCTURE ONLY -- the data members of aruco::Marker (Thirdparty/aruco/aruco/marker.h:47-56) the detector shim fills
#ifndef MOCK_ARUCO_MARKER_H
#define MOCK_ARUCO_MARKER_H
#include <string>
#include <vector>
#include <opencv2/core/core.hpp>
namespace aruco {
class Marker : public std::vector<cv::Point2f> {
public:
    int id = -1;
    float ssize = -1;
    cv::Mat Rvec, Tvec;
    std::string dict_info;
    std::vector<cv::Point> contourPoints;
    void calculateExtrinsics(float, cv::Mat, cv::Mat = cv::Mat(), bool = true) { throw std::runtime_error("mock Marker::calculateExtrinsics"); }
};
}
#endif
