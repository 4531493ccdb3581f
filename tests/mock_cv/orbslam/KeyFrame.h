// TEST INFRASTRUCTURE ONLY -- the members of ORB_SLAM2::KeyFrame the ORBmatcher shim touches (include/KeyFrame.h)
#ifndef MOCK_KEYFRAME_H
#define MOCK_KEYFRAME_H
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
class KeyFrame {
public:
    int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    DBoW2::FeatureVector mFeatVec;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    int mnScaleLevels = 8;
    float mfScaleFactor = 1.2f, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    cv::Mat Rcw, tcw, Ow;
    cv::Mat GetRotation() { return Rcw.clone(); }
    cv::Mat GetTranslation() { return tcw.clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
};
}
#endif
