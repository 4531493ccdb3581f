/*
 * ORBmatcher_orbfe.cc (shim) -- the members of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-83) implemented on liborbfe.so.
 * Compile it INSTEAD of src/ORBmatcher.cc; include/ORBmatcher.h, Frame, KeyFrame and MapPoint stay the reference's own.
 *
 * What stays on the host is what only the host can do: walking the pointer graph (MapPoint*, KeyFrame*), and the map
 * bookkeeping after a match (mvpMapPoints[...] = pMP, Replace, AddObservation).  Everything between -- projection, gates,
 * PredictScale, the Frame grid, Hamming distances, best / second-best rules, rotation histograms -- is one library call per
 * member.  Each function names the reference lines it replaces.
 *
 * One addition to the reference is needed (include/MapPoint.h): PredictScale divides the protected member mfMaxDistance, which
 * the public getters only return scaled by 1.2f / 0.8f (MapPoint.cc:402-412), and (1.2f * m) / 1.2f is not m in float.  Add
 *     float GetMaxDistance() { unique_lock<mutex> lock(mMutexPos); return mfMaxDistance; }
 *     float GetMinDistance() { unique_lock<mutex> lock(mMutexPos); return mfMinDistance; }
 * next to GetMaxDistanceInvariance().
 */
#include "ORBmatcher.h"

#include <climits>
#include <stdexcept>

#include "orbfe.h"

using namespace std;

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

namespace
{

inline void check(int rc)
{
    if (rc != ORBFE_OK) throw std::runtime_error(orbfe_last_error());
}

template <class FrameLike> inline const orbfe_keypoint* keys(const FrameLike& F)
{
    static_assert(sizeof(cv::KeyPoint) == sizeof(orbfe_keypoint), "cv::KeyPoint layout");
    return reinterpret_cast<const orbfe_keypoint*>(F.mvKeysUn.data());
}

// {mnMinX, mnMinY, mnMaxX, mnMaxY} (Frame::ComputeImageBounds, Frame.cc:418-451); cols / rows only size internal tables
template <class FrameLike> struct Bounds
{
    float b[4];
    int cols, rows;
    explicit Bounds(const FrameLike& F)
    {
        b[0] = F.mnMinX; b[1] = F.mnMinY; b[2] = F.mnMaxX; b[3] = F.mnMaxY;
        cols = (int)(F.mnMaxX - F.mnMinX);
        rows = (int)(F.mnMaxY - F.mnMinY);
    }
};

// per-map-point arrays of the guided searches (Fuse, SearchBySim3, the keyframe variants of SearchByProjection)
struct MapPointArrays
{
    vector<float> p3Dw, min_dist, max_dist, normal;
    vector<uint8_t> desc, valid;
    void resize(size_t n)
    {
        p3Dw.assign(3 * n, 0.f); min_dist.assign(n, 0.f); max_dist.assign(n, 0.f); normal.assign(3 * n, 0.f);
        desc.assign(32 * n, 0); valid.assign(n, 0);
    }
    void set(size_t i, MapPoint* pMP)
    {
        const cv::Mat p = pMP->GetWorldPos(), nrm = pMP->GetNormal(), d = pMP->GetDescriptor();
        for (int k = 0; k < 3; k++) { p3Dw[3 * i + k] = p.at<float>(k); normal[3 * i + k] = nrm.at<float>(k); }
        min_dist[i] = pMP->GetMinDistance();
        max_dist[i] = pMP->GetMaxDistance();
        std::copy(d.data, d.data + 32, desc.begin() + 32 * i);
        valid[i] = 1;
    }
};

inline void pose34(const cv::Mat& R, const cv::Mat& t, float T[12])
{
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T[4 * r + c] = R.at<float>(r, c);
        T[4 * r + 3] = t.at<float>(r);
    }
}

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) as the flat arrays of orbfe_search_by_bow
struct FlatFeatVec
{
    vector<uint32_t> node, feature;
    vector<int32_t> offset;
    explicit FlatFeatVec(const DBoW2::FeatureVector& fv)
    {
        offset.push_back(0);
        for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            node.push_back(it->first);
            feature.insert(feature.end(), it->second.begin(), it->second.end());
            offset.push_back((int32_t)feature.size());
        }
    }
};

} // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

// src/ORBmatcher.cc:1651-1667
int ORBmatcher::DescriptorDistance(const cv::Mat& a, const cv::Mat& b) { return orbfe_hamming(a.data, b.data); }

// src/ORBmatcher.cc:131-137
float ORBmatcher::RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5 : 4.0; }

// src/ORBmatcher.cc:45-129 (Tracking::SearchLocalPoints)
int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)
{
    const bool bFactor = th != 1.0;
    vector<orbfe_window_query> q;
    vector<uint8_t> qd, observed;
    vector<MapPoint*> who;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPoint* pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView) continue;
        if (pMP->isBad()) continue;
        const int nPredictedLevel = pMP->mnTrackScaleLevel;
        float r = RadiusByViewingCos(pMP->mTrackViewCos);
        if (bFactor) r *= th;
        const orbfe_window_query w = {pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1,
                                      nPredictedLevel};
        q.push_back(w);
        const cv::Mat d = pMP->GetDescriptor();
        qd.insert(qd.end(), d.data, d.data + 32);
        observed.push_back(pMP->Observations() > 0);
        who.push_back(pMP);
    }
    vector<uint8_t> taken(F.N, 0);
    for (int i = 0; i < F.N; i++) taken[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0; // :91-93
    vector<int32_t> match(q.size(), -1);
    int32_t n = 0;
    const Bounds<Frame> B(F);
    check(orbfe_search_by_projection(keys(F), F.mDescriptors.data, F.N, B.cols, B.rows, B.b, q.data(), qd.data(), (int)q.size(), taken.data(),
                                     observed.data(), /*mode*/ 1, TH_HIGH, mfNNratio, NULL, NULL, NULL, NULL, NULL, match.data(), &n, 0));
    for (size_t k = 0; k < match.size(); k++)
        if (match[k] >= 0) F.mvpMapPoints[match[k]] = who[k]; // :125, in query order: a later assignment overwrites
    return n;
}

// src/ORBmatcher.cc:1332-1474 (Tracking::TrackWithMotionModel), monocular
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
{
    if (!bMono) throw std::runtime_error("orbfe: the stereo / RGB-D branches of SearchByProjection are not part of the monocular system");
    const int NL = LastFrame.N;
    vector<uint8_t> valid(NL, 0), observed(NL, 0), desc((size_t)NL * 32, 0);
    vector<float> x3Dw((size_t)NL * 3, 0.f);
    for (int i = 0; i < NL; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue; // :1358-1362
        valid[i] = 1;
        observed[i] = pMP->Observations() > 0;
        const cv::Mat p = pMP->GetWorldPos(), d = pMP->GetDescriptor();
        for (int k = 0; k < 3; k++) x3Dw[3 * i + k] = p.at<float>(k);
        std::copy(d.data, d.data + 32, desc.begin() + (size_t)32 * i);
    }
    vector<uint8_t> taken(CurrentFrame.N, 0);
    for (int i = 0; i < CurrentFrame.N; i++)
        taken[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0; // :1397-1399
    float T[12];
    pose34(CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3), CurrentFrame.mTcw.rowRange(0, 3).col(3), T);
    const float K4[4] = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy};
    vector<int32_t> match(CurrentFrame.N, -1);
    int32_t n = 0;
    const Bounds<Frame> B(CurrentFrame);
    check(orbfe_search_by_projection_last_frame(keys(CurrentFrame), CurrentFrame.mDescriptors.data, CurrentFrame.N, taken.data(), B.cols, B.rows,
                                                B.b, keys(LastFrame), NL, valid.data(), x3Dw.data(), desc.data(), observed.data(), T, K4,
                                                CurrentFrame.mvScaleFactors.data(), CurrentFrame.mnScaleLevels, th, TH_HIGH,
                                                mbCheckOrientation, match.data(), &n, 0));
    for (int i2 = 0; i2 < CurrentFrame.N; i2++)
        if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[match[i2]]; // :1423 (the rotation check already removed the rest)
    return n;
}

// src/ORBmatcher.cc:1476-1603 (Tracking::Relocalization)
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const int n_kf = (int)vpMPs.size();
    MapPointArrays A;
    A.resize(n_kf);
    vector<float> angle(n_kf, 0.f);
    for (int i = 0; i < n_kf; i++) {
        MapPoint* pMP = vpMPs[i];
        angle[i] = pKF->mvKeysUn[i].angle;
        if (pMP && !pMP->isBad() && !sAlreadyFound.count(pMP)) A.set(i, pMP); // :1494-1498
    }
    vector<uint8_t> taken(CurrentFrame.N, 0);
    for (int i = 0; i < CurrentFrame.N; i++) taken[i] = CurrentFrame.mvpMapPoints[i] != NULL; // :1547
    float T[12];
    pose34(Rcw, tcw, T);
    const float O[3] = {Ow.at<float>(0), Ow.at<float>(1), Ow.at<float>(2)};
    const float K4[4] = {CurrentFrame.fx, CurrentFrame.fy, CurrentFrame.cx, CurrentFrame.cy};
    vector<int32_t> match(CurrentFrame.N, -1);
    int32_t n = 0;
    const Bounds<Frame> B(CurrentFrame);
    check(orbfe_search_by_projection_keyframe(keys(CurrentFrame), CurrentFrame.mDescriptors.data, CurrentFrame.N, taken.data(), B.cols, B.rows,
                                              B.b, n_kf, angle.data(), A.valid.data(), A.p3Dw.data(), A.min_dist.data(), A.max_dist.data(),
                                              A.desc.data(), T, O, K4, CurrentFrame.mvScaleFactors.data(), CurrentFrame.mnScaleLevels,
                                              CurrentFrame.mfLogScaleFactor, th, ORBdist, mbCheckOrientation, match.data(), &n, 0));
    for (int i2 = 0; i2 < CurrentFrame.N; i2++)
        if (match[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = vpMPs[match[i2]]; // :1561
    return n;
}

// src/ORBmatcher.cc:294-407 (LoopClosing::ComputeSim3)
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, vector<MapPoint*>& vpMatched, int th)
{
    // decompose Scw (:303-308)
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    const int nmp = (int)vpPoints.size();
    MapPointArrays A;
    A.resize(nmp);
    for (int i = 0; i < nmp; i++)
        if (!vpPoints[i]->isBad() && !spAlreadyFound.count(vpPoints[i])) A.set(i, vpPoints[i]); // :321-322
    vector<uint8_t> matched(pKF->N, 0);
    for (int i = 0; i < pKF->N; i++) matched[i] = vpMatched[i] != NULL; // :383
    float T[12];
    pose34(Rcw, tcw, T);
    const float O[3] = {Ow.at<float>(0), Ow.at<float>(1), Ow.at<float>(2)};
    const float K4[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
    vector<int32_t> match(pKF->N, -1);
    int32_t n = 0;
    const Bounds<KeyFrame> B(*pKF);
    check(orbfe_search_by_projection_sim3(keys(*pKF), pKF->mDescriptors.data, pKF->N, B.cols, B.rows, B.b, matched.data(), A.p3Dw.data(),
                                          A.valid.data(), A.min_dist.data(), A.max_dist.data(), A.normal.data(), A.desc.data(), nmp, T, O, K4,
                                          pKF->mvScaleFactors.data(), pKF->mnScaleLevels, pKF->mfLogScaleFactor, th, match.data(), &n, 0));
    for (int i = 0; i < pKF->N; i++)
        if (match[i] >= 0) vpMatched[i] = vpPoints[match[i]]; // :399
    return n;
}

// src/ORBmatcher.cc:159-292 (Tracking::TrackReferenceKeyFrame, Relocalization)
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
{
    const vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    vector<uint8_t> valid1(pKF->N, 0);
    for (int i = 0; i < pKF->N; i++) valid1[i] = vpMapPointsKF[i] && !vpMapPointsKF[i]->isBad(); // :196-201
    const FlatFeatVec fk(pKF->mFeatVec), ff(F.mFeatVec);
    vector<int32_t> m12(pKF->N, -1), m21(F.N, -1);
    int32_t n = 0;
    // this fork's histogram factor HISTO_LENGTH / 360.0f (:174); accept bestDist1 <= TH_LOW (:233)
    check(orbfe_search_by_bow(keys(*pKF), pKF->mDescriptors.data, valid1.data(), pKF->N, fk.node.data(), fk.offset.data(), fk.feature.data(),
                              (int)fk.node.size(), keys(F), F.mDescriptors.data, NULL, F.N, ff.node.data(), ff.offset.data(), ff.feature.data(),
                              (int)ff.node.size(), mfNNratio, mbCheckOrientation, TH_LOW, HISTO_LENGTH / 360.0f, m12.data(), m21.data(), &n, 0));
    for (int i2 = 0; i2 < F.N; i2++)
        if (m21[i2] >= 0) vpMapPointMatches[i2] = vpMapPointsKF[m21[i2]]; // :239
    return n;
}

// src/ORBmatcher.cc:526-659 (LoopClosing::ComputeSim3)
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12)
{
    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
    vector<uint8_t> valid1(pKF1->N, 0), valid2(pKF2->N, 0);
    for (int i = 0; i < pKF1->N; i++) valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad(); // :566-570
    for (int i = 0; i < pKF2->N; i++) valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad(); // :583-590
    const FlatFeatVec f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    vector<int32_t> m12(pKF1->N, -1), m21(pKF2->N, -1);
    int32_t n = 0;
    // factor 1.0f / HISTO_LENGTH (:546); accept bestDist1 < TH_LOW (:602)
    check(orbfe_search_by_bow(keys(*pKF1), pKF1->mDescriptors.data, valid1.data(), pKF1->N, f1.node.data(), f1.offset.data(), f1.feature.data(),
                              (int)f1.node.size(), keys(*pKF2), pKF2->mDescriptors.data, valid2.data(), pKF2->N, f2.node.data(),
                              f2.offset.data(), f2.feature.data(), (int)f2.node.size(), mfNNratio, mbCheckOrientation, TH_LOW - 1,
                              1.0f / HISTO_LENGTH, m12.data(), m21.data(), &n, 0));
    for (int i1 = 0; i1 < pKF1->N; i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]]; // :606
    return n;
}

// src/ORBmatcher.cc:409-524 (Tracking::MonocularInitialization)
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize)
{
    vnMatches12 = vector<int>(F1.mvKeysUn.size(), -1);
    static_assert(sizeof(cv::Point2f) == 8, "cv::Point2f is two floats");
    int32_t n = 0;
    const Bounds<Frame> B(F2);
    check(orbfe_search_for_initialization(keys(F1), F1.mDescriptors.data, F1.N, keys(F2), F2.mDescriptors.data, F2.N, B.cols, B.rows, B.b,
                                          reinterpret_cast<float*>(vbPrevMatched.data()), vnMatches12.data(), windowSize, mfNNratio,
                                          mbCheckOrientation, &n, 0));
    return n;
}

// src/ORBmatcher.cc:661-827 (LocalMapping::CreateNewMapPoints), monocular
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vector<pair<size_t, size_t> >& vMatchedPairs,
                                       const bool bOnlyStereo)
{
    if (bOnlyStereo) { vMatchedPairs.clear(); return 0; } // a monocular keyframe has no stereo observation (:716-720)
    // epipole of camera 1 in image 2 (:669-675)
    cv::Mat Cw = pKF1->GetCameraCenter();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat C2 = R2w * Cw + t2w;
    const float invz = 1.0f / C2.at<float>(2);
    const float ex = pKF2->fx * C2.at<float>(0) * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.at<float>(1) * invz + pKF2->cy;
    vector<uint8_t> has1(pKF1->N, 0), has2(pKF2->N, 0);
    for (int i = 0; i < pKF1->N; i++) has1[i] = pKF1->GetMapPoint(i) != NULL; // :710-714
    for (int i = 0; i < pKF2->N; i++) has2[i] = pKF2->GetMapPoint(i) != NULL; // :731-735
    const FlatFeatVec f1(pKF1->mFeatVec), f2(pKF2->mFeatVec);
    float F[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
    vector<int32_t> m12(pKF1->N, -1);
    int32_t n = 0;
    check(orbfe_search_for_triangulation(keys(*pKF1), pKF1->mDescriptors.data, has1.data(), pKF1->N, f1.node.data(), f1.offset.data(),
                                         f1.feature.data(), (int)f1.node.size(), keys(*pKF2), pKF2->mDescriptors.data, has2.data(), pKF2->N,
                                         f2.node.data(), f2.offset.data(), f2.feature.data(), (int)f2.node.size(), F, ex, ey,
                                         pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(), pKF2->mnScaleLevels, mbCheckOrientation,
                                         m12.data(), &n, 0));
    vMatchedPairs.clear();
    vMatchedPairs.reserve(n);
    for (size_t i = 0; i < m12.size(); i++) // :816-824
        if (m12[i] >= 0) vMatchedPairs.push_back(make_pair(i, (size_t)m12[i]));
    return n;
}

// src/ORBmatcher.cc:1106-1330 (LoopClosing::ComputeSim3)
int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12,
                             const cv::Mat& t12, const float th)
{
    cv::Mat R1w = pKF1->GetRotation(), t1w = pKF1->GetTranslation(), R2w = pKF2->GetRotation(), t2w = pKF2->GetTranslation();
    // transformation between cameras (:1123-1126)
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) { // :1134-1145
        MapPoint* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    MapPointArrays A1, A2;
    A1.resize(N1);
    A2.resize(N2);
    for (int i = 0; i < N1; i++)
        if (vpMapPoints1[i] && !vbAlreadyMatched1[i] && !vpMapPoints1[i]->isBad()) A1.set(i, vpMapPoints1[i]); // :1148-1155
    for (int i = 0; i < N2; i++)
        if (vpMapPoints2[i] && !vbAlreadyMatched2[i] && !vpMapPoints2[i]->isBad()) A2.set(i, vpMapPoints2[i]); // :1230-1236
    float T1[12], T2[12], S12[12], S21[12];
    pose34(R1w, t1w, T1); pose34(R2w, t2w, T2); pose34(sR12, t12, S12); pose34(sR21, t21, S21);
    const float K4[4] = {pKF1->fx, pKF1->fy, pKF1->cx, pKF1->cy};
    vector<int32_t> m12(N1, -1);
    int32_t n = 0;
    const Bounds<KeyFrame> B(*pKF1);
    check(orbfe_search_by_sim3(keys(*pKF1), pKF1->mDescriptors.data, N1, keys(*pKF2), pKF2->mDescriptors.data, N2, B.cols, B.rows, B.b,
                               A1.p3Dw.data(), A1.valid.data(), A1.min_dist.data(), A1.max_dist.data(), A1.desc.data(), A2.p3Dw.data(),
                               A2.valid.data(), A2.min_dist.data(), A2.max_dist.data(), A2.desc.data(), T1, T2, S12, S21, K4,
                               pKF1->mvScaleFactors.data(), pKF1->mnScaleLevels, pKF1->mfLogScaleFactor, th, TH_HIGH, m12.data(), &n, 0));
    for (int i1 = 0; i1 < N1; i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]]; // :1316
    return n;
}

// src/ORBmatcher.cc:829-970 (LocalMapping::SearchInNeighbors)
int ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, const float th)
{
    cv::Mat Rcw = pKF->GetRotation(), tcw = pKF->GetTranslation(), Ow = pKF->GetCameraCenter();
    const int nMPs = (int)vpMapPoints.size();
    MapPointArrays A;
    A.resize(nMPs);
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF)) A.set(i, pMP); // :850-856
    }
    float T[12];
    pose34(Rcw, tcw, T);
    const float O[3] = {Ow.at<float>(0), Ow.at<float>(1), Ow.at<float>(2)};
    const float K4[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
    vector<int32_t> bestIdx(nMPs, -1), bestDist(nMPs, 256);
    const Bounds<KeyFrame> B(*pKF);
    check(orbfe_fuse_search(keys(*pKF), pKF->mDescriptors.data, pKF->N, B.cols, B.rows, B.b, A.p3Dw.data(), A.valid.data(), A.min_dist.data(),
                            A.max_dist.data(), A.normal.data(), A.desc.data(), nMPs, T, O, K4, pKF->mvScaleFactors.data(),
                            pKF->mvInvLevelSigma2.data(), pKF->mnScaleLevels, pKF->mfLogScaleFactor, th, 5.99, bestIdx.data(), bestDist.data(), 0));
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) { // :943-964
        if (!A.valid[i] || bestDist[i] > TH_LOW) continue;
        MapPoint* pMP = vpMapPoints[i];
        // the reference evaluates :850-856 inside this sequential loop, after earlier iterations' Replace / AddObservation: a point
        // that occurs twice in vpMapPoints, or was replaced a moment ago, is skipped there -- so the gates are asked again here
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx[i]);
            pKF->AddMapPoint(pMP, bestIdx[i]);
        }
        nFused++;
    }
    return nFused;
}

// src/ORBmatcher.cc:972-1104 (LoopClosing::SearchAndFuse)
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint)
{
    cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
    const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
    cv::Mat Rcw = sRcw / scw;
    cv::Mat tcw = Scw.rowRange(0, 3).col(3) / scw;
    cv::Mat Ow = -Rcw.t() * tcw;
    const set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    MapPointArrays A;
    A.resize(nPoints);
    for (int i = 0; i < nPoints; i++)
        if (!vpPoints[i]->isBad() && !spAlreadyFound.count(vpPoints[i])) A.set(i, vpPoints[i]); // :1003-1006
    float T[12];
    pose34(Rcw, tcw, T);
    const float O[3] = {Ow.at<float>(0), Ow.at<float>(1), Ow.at<float>(2)};
    const float K4[4] = {pKF->fx, pKF->fy, pKF->cx, pKF->cy};
    vector<int32_t> bestIdx(nPoints, -1), bestDist(nPoints, 256);
    const Bounds<KeyFrame> B(*pKF);
    check(orbfe_fuse_search(keys(*pKF), pKF->mDescriptors.data, pKF->N, B.cols, B.rows, B.b, A.p3Dw.data(), A.valid.data(), A.min_dist.data(),
                            A.max_dist.data(), A.normal.data(), A.desc.data(), nPoints, T, O, K4, pKF->mvScaleFactors.data(),
                            pKF->mvInvLevelSigma2.data(), pKF->mnScaleLevels, pKF->mfLogScaleFactor, th, 0.0, bestIdx.data(), bestDist.data(), 0));
    int nFused = 0;
    for (int i = 0; i < nPoints; i++) { // :1085-1099
        if (!A.valid[i] || bestDist[i] > TH_LOW) continue;
        MapPoint* pMP = vpPoints[i];
        if (pMP->isBad()) continue; // :1003 is evaluated per iteration in the reference (spAlreadyFound is a snapshot there too)
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx[i]);
            pKF->AddMapPoint(pMP, bestIdx[i]);
        }
        nFused++;
    }
    return nFused;
}

// the two protected helpers stay callable for code that derives from ORBmatcher
bool ORBmatcher::CheckDistEpipolarLine(const cv::KeyPoint& kp1, const cv::KeyPoint& kp2, const cv::Mat& F12, const KeyFrame* pKF2)
{
    float F[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
    return orbfe_epipolar_distance_ok(kp1.pt.x, kp1.pt.y, kp2.pt.x, kp2.pt.y, F, pKF2->mvLevelSigma2[kp2.octave]) != 0;
}

void ORBmatcher::ComputeThreeMaxima(vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3)
{
    vector<int32_t> counts(L);
    for (int i = 0; i < L; i++) counts[i] = (int32_t)histo[i].size();
    int32_t ind[3];
    orbfe_three_maxima(counts.data(), L, ind);
    // the reference leaves an index it never assigned untouched (callers initialise all three to -1)
    if (ind[0] >= 0) ind1 = ind[0];
    if (ind[1] >= 0 || ind[0] >= 0) ind2 = ind[1];
    if (ind[2] >= 0 || ind[0] >= 0) ind3 = ind[2];
}

} // namespace ORB_SLAM2
