// TEST INFRASTRUCTURE ONLY -- the declaration of ORB_SLAM2::ORBmatcher with the reference's member signatures
// (include/ORBmatcher.h:41-83): in an integration the reference's own header is used, here it lets the shim compile.
#ifndef MOCK_ORBMATCHER_H
#define MOCK_ORBMATCHER_H
#include <set>
#include <vector>
#include <opencv2/core/core.hpp>
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
namespace ORB_SLAM2 {
using std::pair;
using std::vector;
class ORBmatcher {
public:
    ORBmatcher(float nnratio = 0.6, bool checkOri = true);
    static int DescriptorDistance(const cv::Mat& a, const cv::Mat& b);
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo);
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th);
    int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint);
    static const int TH_LOW, TH_HIGH, HISTO_LENGTH;
protected:
    bool CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF);
    float RadiusByViewingCos(const float& viewCos);
    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);
    float mfNNratio;
    bool mbCheckOrientation;
};
}
#endif
