// TEST INFRASTRUCTURE ONLY -- the members of ORB_SLAM2::Frame the ORBmatcher shim touches (include/Frame.h)
#ifndef MOCK_FRAME_H
#define MOCK_FRAME_H
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class Frame {
public:
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvuRight;
    cv::Mat mTcw;
    DBoW2::FeatureVector mFeatVec;
    int mnScaleLevels = 8;
    float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    static float fx, fy, cx, cy, invfx, invfy;
    static float mnMinX, mnMaxX, mnMinY, mnMaxY;
};
}
#endif
