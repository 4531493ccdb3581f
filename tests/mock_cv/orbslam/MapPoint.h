// TEST INFRASTRUCTURE ONLY -- the members of ORB_SLAM2::MapPoint the ORBmatcher shim touches, with the reference's names and
// types (include/MapPoint.h), so that include/shims/ORBmatcher_orbfe.cc compiles and runs without the SLAM system.
#ifndef MOCK_MAPPOINT_H
#define MOCK_MAPPOINT_H
#include <map>
#include <opencv2/core/core.hpp>
namespace ORB_SLAM2 {
class KeyFrame;
class MapPoint {
public:
    cv::Mat mWorldPos, mNormal, mDescriptor;
    float mfMinDistance = 0, mfMaxDistance = 0;
    int nObs = 1;
    bool mbBad = false;
    std::map<KeyFrame*, size_t> mObservations;
    MapPoint* mpReplaced = nullptr;
    // tracking fields (MapPoint.h:93-100)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 1;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormal.clone(); }
    cv::Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return nObs; }
    bool isBad() { return mbBad; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    float GetMinDistance() { return mfMinDistance; }   // the two getters the shim asks the maintainer to add
    float GetMaxDistance() { return mfMaxDistance; }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    void AddObservation(KeyFrame* pKF, size_t idx) { if (!mObservations.count(pKF)) { mObservations[pKF] = idx; nObs++; } }
    void Replace(MapPoint* pMP) { if (pMP != this) { mbBad = true; mpReplaced = pMP; } }
};
}
#endif
