// TEST INFRASTRUCTURE ONLY -- drives the C++ shims of include/shims/ (ORBextractor, aruco::MarkerDetector, ORBmatcher) the way the
// reference's Frame / Tracking code does, against the mock OpenCV / SLAM headers of tests/mock_cv/, and dumps inputs and results as
// raw arrays; tests/test_shims_gpu.py repeats every call through the ctypes binding and compares.
//   shim_driver <frames.raw> <rows> <cols> <nframes> <out prefix>          exit 3 = no GPU (the library has no CPU fallback)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "MarkerDetector.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM2;

float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::invfx, Frame::invfy;
float Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY;

static std::string g_prefix;
template <class T> static void dump(const char* name, const T* p, size_t n)
{
    FILE* f = fopen((g_prefix + "_" + name + ".bin").c_str(), "wb");
    if (n) fwrite(p, sizeof(T), n, f);
    fclose(f);
}
template <class T> static void dump(const char* name, const std::vector<T>& v) { dump(name, v.data(), v.size()); }

static cv::Mat mat32(int r, int c, std::initializer_list<float> v)
{
    cv::Mat m(r, c, CV_32F);
    int i = 0;
    for (float x : v) { m.at<float>(i / c, i % c) = x; i++; }
    return m;
}

static void make_frame(Frame& F, ORBextractor& ex, const cv::Mat& im)
{
    // Frame::Frame (Frame.cc:74-127) as far as the matcher needs it
    ex(im, cv::Mat(), F.mvKeys, F.mDescriptors);
    F.N = (int)F.mvKeys.size();
    F.mvKeysUn = F.mvKeys; // no distortion
    F.mvpMapPoints.assign(F.N, (MapPoint*)NULL);
    F.mvbOutlier.assign(F.N, false);
    F.mnScaleLevels = ex.GetLevels();
    F.mfScaleFactor = ex.GetScaleFactor();
    F.mfLogScaleFactor = log(F.mfScaleFactor);
    F.mvScaleFactors = ex.GetScaleFactors();
    F.mvInvScaleFactors = ex.GetInverseScaleFactors();
    F.mvLevelSigma2 = ex.GetScaleSigmaSquares();
    F.mvInvLevelSigma2 = ex.GetInverseScaleSigmaSquares();
}

int main(int argc, char** argv)
{
    if (argc < 6) return 2;
    const int rows = atoi(argv[2]), cols = atoi(argv[3]), nframes = atoi(argv[4]);
    g_prefix = argv[5];
    if (orbfe_device_count() == 0) { fprintf(stderr, "no HIP device: %s\n", orbfe_last_error()); return 3; }
    std::vector<unsigned char> raw((size_t)rows * cols * nframes);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) return 2;
    fclose(f);
    Frame::fx = 517.3f; Frame::fy = 516.5f; Frame::cx = 318.6f; Frame::cy = 255.3f;
    Frame::mnMinX = 0; Frame::mnMinY = 0; Frame::mnMaxX = (float)cols; Frame::mnMaxY = (float)rows;

    // ---- ORBextractor through operator() (Frame::ExtractORB, Frame.cc:200-206)
    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<Frame> F(nframes);
    std::vector<int> nk;
    for (int i = 0; i < nframes; i++) {
        cv::Mat im(rows, cols, CV_8UC1, raw.data() + (size_t)i * rows * cols, (size_t)cols);
        make_frame(F[i], ex, im);
        nk.push_back(F[i].N);
        dump(("kps" + std::to_string(i)).c_str(), F[i].mvKeys);
        dump(("desc" + std::to_string(i)).c_str(), F[i].mDescriptors.data, (size_t)F[i].N * 32);
    }
    dump("nk", nk);
    {   // empty image: outputs untouched (ORBextractor.cc:1046)
        std::vector<cv::KeyPoint> k(3);
        cv::Mat d;
        ex(cv::Mat(), cv::Mat(), k, d);
        if (k.size() != 3) return 10;
    }

    // ---- aruco::MarkerDetector as Frame.cc:129-142 configures and calls it
    aruco::MarkerDetector det;
    det.setDictionary("ARUCO");
    det.setDetectionMode(aruco::DM_NORMAL);
    det.getParameters().setCornerRefinementMethod(aruco::CORNER_LINES);
    aruco::CameraParameters cam;
    cam.CameraMatrix = mat32(3, 3, {517.306408f, 0, 318.643040f, 0, 516.469215f, 255.313989f, 0, 0, 1});
    cam.Distorsion = mat32(1, 5, {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f});
    cam.CamSize = cv::Size(1280, 720);
    {
        cv::Mat im(rows, cols, CV_8UC1, raw.data(), (size_t)cols);
        std::vector<aruco::Marker> mk = det.detect(im, cam, 0.187f);
        std::vector<float> rec; // id, 8 corner coordinates, rvec, tvec, contour length
        for (auto& m : mk) {
            rec.push_back((float)m.id);
            for (int k = 0; k < 4; k++) { rec.push_back(m[k].x); rec.push_back(m[k].y); }
            for (int k = 0; k < 3; k++) rec.push_back(m.Rvec.at<float>(k, 0));
            for (int k = 0; k < 3; k++) rec.push_back(m.Tvec.at<float>(k, 0));
            rec.push_back((float)m.contourPoints.size());
            if (m.ssize != 0.187f || m.dict_info != "ARUCO") return 11;
        }
        dump("markers", rec);
        // the other modes through the shim: a CV_8UC3 frame with three equal channels is its grey image (same markers, bit for bit);
        // DM_FAST + CORNER_SUBPIX (the reference's default corner method) finds the same ids; a value outside the enums throws
        {
            std::vector<unsigned char> raw3((size_t)rows * cols * 3);
            for (size_t i = 0; i < (size_t)rows * cols; i++) raw3[3 * i] = raw3[3 * i + 1] = raw3[3 * i + 2] = raw[i];
            cv::Mat im3(rows, cols, CV_8UC3, raw3.data(), (size_t)cols * 3);
            std::vector<aruco::Marker> mk3 = det.detect(im3, cam, 0.187f);
            if (mk3.size() != mk.size()) return 14;
            for (size_t i = 0; i < mk.size(); i++)
                for (int k = 0; k < 4; k++)
                    if (mk3[i].id != mk[i].id || mk3[i][k].x != mk[i][k].x || mk3[i][k].y != mk[i][k].y) return 14;
            det.getParameters().setCornerRefinementMethod(aruco::CORNER_SUBPIX);
            det.setDetectionMode(aruco::DM_FAST, 0.f);
            std::vector<aruco::Marker> mkf = det.detect(im);
            if (mkf.size() != mk.size()) return 15;
            for (size_t i = 0; i < mk.size(); i++)
                if (mkf[i].id != mk[i].id || std::fabs(mkf[i][0].x - mk[i][0].x) > 2.f) return 15;
            det.setDetectionMode(aruco::DM_NORMAL);
            det.getParameters().setCornerRefinementMethod(aruco::CORNER_LINES);
        }
        bool threw = false;
        try { det.setDetectionMode((aruco::DetectionMode)7); } catch (const cv::Exception&) { threw = true; }
        if (!threw) return 12;
        threw = false;
        try { det.setDictionary("NO_SUCH_DICTIONARY"); } catch (const std::runtime_error&) { threw = true; }
        if (!threw) return 13;
    }

    // ---- ORBmatcher members
    std::vector<int> res;
    ORBmatcher matcher(0.9f, true);
    {   // Tracking::MonocularInitialization (Tracking.cc:520-532)
        std::vector<cv::Point2f> prev(F[0].mvKeysUn.size());
        for (size_t i = 0; i < prev.size(); i++) prev[i] = F[0].mvKeysUn[i].pt;
        std::vector<int> m12;
        res.push_back(matcher.SearchForInitialization(F[0], F[1], prev, m12, 100));
        dump("sfi_m12", m12);
        dump("sfi_prev", (const float*)prev.data(), prev.size() * 2);
        res.push_back(ORBmatcher::DescriptorDistance(F[0].mDescriptors.row(0), F[1].mDescriptors.row(0)));
    }
    // map points of frame 0: back-projection at a depth that depends on the index; frame 1 sees them from a small motion
    Frame& L = F[0];
    Frame& C = F[1];
    std::vector<MapPoint> mps(L.N);
    std::vector<float> x3(3 * (size_t)L.N), dmin(L.N), dmax(L.N);
    for (int i = 0; i < L.N; i++) {
        const float z = 1.0f + 0.1f * (float)(i % 50);
        const float X = (L.mvKeysUn[i].pt.x - Frame::cx) / Frame::fx * z, Y = (L.mvKeysUn[i].pt.y - Frame::cy) / Frame::fy * z;
        mps[i].mWorldPos = mat32(3, 1, {X, Y, z});
        const float d = sqrt(X * X + Y * Y + z * z);
        mps[i].mNormal = mat32(3, 1, {X / d, Y / d, z / d});
        mps[i].mDescriptor = L.mDescriptors.row(i);
        mps[i].mfMaxDistance = d * L.mvScaleFactors[L.mvKeysUn[i].octave];
        mps[i].mfMinDistance = mps[i].mfMaxDistance / L.mvScaleFactors[L.mnScaleLevels - 1];
        mps[i].nObs = (i % 7 == 0) ? 0 : 2;
        x3[3 * i] = X; x3[3 * i + 1] = Y; x3[3 * i + 2] = z;
        dmin[i] = mps[i].mfMinDistance; dmax[i] = mps[i].mfMaxDistance;
        if (i % 9 != 0) L.mvpMapPoints[i] = &mps[i];
    }
    dump("x3", x3); dump("dmin", dmin); dump("dmax", dmax);
    C.mTcw = mat32(4, 4, {1, -0.002f, 0.001f, 0.004f, 0.002f, 1, -0.003f, -0.006f, -0.001f, 0.003f, 1, 0.01f, 0, 0, 0, 1});
    {   // Tracking::TrackWithMotionModel (Tracking.cc:1011)
        res.push_back(matcher.SearchByProjection(C, L, 15.0f, true));
        std::vector<int> got(C.N, -1);
        for (int i = 0; i < C.N; i++) got[i] = C.mvpMapPoints[i] ? (int)(C.mvpMapPoints[i] - mps.data()) : -1;
        dump("last_frame", got);
        C.mvpMapPoints.assign(C.N, (MapPoint*)NULL);
    }
    KeyFrame KF; // frame 0 as a keyframe
    KF.N = L.N; KF.mvKeys = L.mvKeys; KF.mvKeysUn = L.mvKeysUn; KF.mDescriptors = L.mDescriptors; KF.mvpMapPoints = L.mvpMapPoints;
    KF.fx = Frame::fx; KF.fy = Frame::fy; KF.cx = Frame::cx; KF.cy = Frame::cy;
    KF.mnMinX = 0; KF.mnMinY = 0; KF.mnMaxX = cols; KF.mnMaxY = rows;
    KF.mnScaleLevels = L.mnScaleLevels; KF.mfScaleFactor = L.mfScaleFactor; KF.mfLogScaleFactor = L.mfLogScaleFactor;
    KF.mvScaleFactors = L.mvScaleFactors; KF.mvLevelSigma2 = L.mvLevelSigma2; KF.mvInvLevelSigma2 = L.mvInvLevelSigma2;
    KF.Rcw = mat32(3, 3, {1, 0, 0, 0, 1, 0, 0, 0, 1}); KF.tcw = mat32(3, 1, {0, 0, 0}); KF.Ow = mat32(3, 1, {0, 0, 0});
    {   // Tracking::Relocalization (Tracking.cc:1858): SearchByProjection(CurrentFrame, pKF, sFound, 10, 100)
        std::set<MapPoint*> sFound;
        for (int i = 0; i < L.N; i += 5) sFound.insert(&mps[i]);
        for (int i = 0; i < C.N; i += 11) C.mvpMapPoints[i] = &mps[0]; // keypoints that already carry a point
        std::vector<int> before(C.N);
        for (int i = 0; i < C.N; i++) before[i] = C.mvpMapPoints[i] != NULL;
        res.push_back(matcher.SearchByProjection(C, &KF, sFound, 10.0f, 100));
        std::vector<int> got(C.N, -1);
        for (int i = 0; i < C.N; i++) got[i] = (C.mvpMapPoints[i] && !before[i]) ? (int)(C.mvpMapPoints[i] - mps.data()) : -1;
        dump("keyframe", got);
        C.mvpMapPoints.assign(C.N, (MapPoint*)NULL);
    }
    {   // Tracking::SearchLocalPoints (Tracking.cc:1515): SearchByProjection(F, vpMapPoints, th)
        std::vector<MapPoint*> local;
        std::vector<float> proj;
        for (int i = 0; i < L.N; i += 2) {
            MapPoint* p = &mps[i];
            p->mbTrackInView = true;
            p->mTrackProjX = L.mvKeysUn[i].pt.x + 1.5f; p->mTrackProjY = L.mvKeysUn[i].pt.y - 1.0f;
            p->mnTrackScaleLevel = L.mvKeysUn[i].octave;
            p->mTrackViewCos = (i % 3) ? 0.9f : 0.999f;
            local.push_back(p);
        }
        res.push_back(matcher.SearchByProjection(L, local, 3.0f)); // frame 0's own keypoints: every other one already holds its point
        std::vector<int> got(L.N, -1);
        for (int i = 0; i < L.N; i++) got[i] = L.mvpMapPoints[i] ? (int)(L.mvpMapPoints[i] - mps.data()) : -1;
        dump("local_points", got);
    }
    {   // LocalMapping::SearchInNeighbors: Fuse(pKF, vpMapPoints)
        std::vector<MapPoint> dup(mps.begin(), mps.end()); // a second set of points at the same places: they fuse with the keyframe's
        std::vector<MapPoint*> cand;
        for (auto& m : dup) { m.mObservations.clear(); m.nObs = 1; cand.push_back(&m); }
        res.push_back(matcher.Fuse(&KF, cand, 3.0f));
        int replaced = 0, added = 0;
        for (auto& m : dup) replaced += m.mbBad ? 1 : 0;
        for (auto& m : dup) added += m.mObservations.count(&KF) ? 1 : 0;
        res.push_back(replaced);
        res.push_back(added);
    }
    // ---- the members that need FeatureVectors or a second keyframe: both SearchByBoW variants, SearchForTriangulation, SearchBySim3,
    // SearchByProjection(pKF, Scw, ...), Fuse(pKF, Scw, ...).  argv[6] = a vocabulary in DBoW2's text format (tests/voc_cases.py).
    if (argc > 6) {
        orbfe_vocabulary* voc = orbfe_vocabulary_load_text(argv[6], 0);
        if (!voc) { fprintf(stderr, "vocabulary: %s\n", orbfe_last_error()); return 20; }
        auto featvec = [&](const cv::Mat& desc, int n, DBoW2::FeatureVector& fv) { // Frame::ComputeBoW (Frame.cc:348-355): transform(.., 4)
            std::vector<uint32_t> bw(n), fn(n), ff(n);
            std::vector<double> bv(n);
            std::vector<int32_t> fo(n + 1);
            int32_t nb = 0, nf = 0;
            if (orbfe_vocabulary_transform(voc, desc.data, n, 4, NULL, NULL, NULL, bw.data(), bv.data(), &nb, fn.data(), fo.data(), ff.data(), &nf)) return false;
            fv.clear();
            for (int i = 0; i < nf; i++) fv[fn[i]] = std::vector<unsigned int>(ff.begin() + fo[i], ff.begin() + fo[i + 1]);
            return true;
        };
        if (!featvec(L.mDescriptors, L.N, KF.mFeatVec) || !featvec(C.mDescriptors, C.N, C.mFeatVec)) return 21;
        // the second keyframe: frame 1 seen from a pure translation (every derived pose quantity is then exact in float), its own map
        // points back-projected from its keypoints
        KeyFrame KF2;
        KF2.N = C.N; KF2.mvKeys = C.mvKeys; KF2.mvKeysUn = C.mvKeysUn; KF2.mDescriptors = C.mDescriptors; KF2.mFeatVec = C.mFeatVec;
        KF2.fx = Frame::fx; KF2.fy = Frame::fy; KF2.cx = Frame::cx; KF2.cy = Frame::cy;
        KF2.mnMinX = 0; KF2.mnMinY = 0; KF2.mnMaxX = cols; KF2.mnMaxY = rows;
        KF2.mnScaleLevels = C.mnScaleLevels; KF2.mfScaleFactor = C.mfScaleFactor; KF2.mfLogScaleFactor = C.mfLogScaleFactor;
        KF2.mvScaleFactors = C.mvScaleFactors; KF2.mvLevelSigma2 = C.mvLevelSigma2; KF2.mvInvLevelSigma2 = C.mvInvLevelSigma2;
        const float t2[3] = {0.004f, -0.006f, 0.01f};
        KF2.Rcw = mat32(3, 3, {1, 0, 0, 0, 1, 0, 0, 0, 1}); KF2.tcw = mat32(3, 1, {t2[0], t2[1], t2[2]}); KF2.Ow = mat32(3, 1, {-t2[0], -t2[1], -t2[2]});
        std::vector<MapPoint> mps2(C.N);
        std::vector<float> x3b(3 * (size_t)C.N), dminb(C.N), dmaxb(C.N);
        KF2.mvpMapPoints.assign(C.N, (MapPoint*)NULL);
        for (int i = 0; i < C.N; i++) {
            const float z = 1.0f + 0.1f * (float)(i % 50);
            const float Xc = (C.mvKeysUn[i].pt.x - Frame::cx) / Frame::fx * z, Yc = (C.mvKeysUn[i].pt.y - Frame::cy) / Frame::fy * z;
            const float X = Xc - t2[0], Y = Yc - t2[1], Z = z - t2[2]; // world = camera - tcw (R = I)
            mps2[i].mWorldPos = mat32(3, 1, {X, Y, Z});
            const float d = sqrt(Xc * Xc + Yc * Yc + z * z);
            mps2[i].mNormal = mat32(3, 1, {Xc / d, Yc / d, z / d});
            mps2[i].mDescriptor = C.mDescriptors.row(i);
            mps2[i].mfMaxDistance = d * C.mvScaleFactors[C.mvKeysUn[i].octave];
            mps2[i].mfMinDistance = mps2[i].mfMaxDistance / C.mvScaleFactors[C.mnScaleLevels - 1];
            x3b[3 * i] = X; x3b[3 * i + 1] = Y; x3b[3 * i + 2] = Z;
            dminb[i] = mps2[i].mfMinDistance; dmaxb[i] = mps2[i].mfMaxDistance;
            if (i % 4 != 0) KF2.mvpMapPoints[i] = &mps2[i];
        }
        dump("x3b", x3b); dump("dminb", dminb); dump("dmaxb", dmaxb);
        KF.mvpMapPoints = L.mvpMapPoints; // (SearchByProjection(F, local) and Fuse above changed frame 0's / the keyframe's points)
        for (int i = 0; i < L.N; i++) KF.mvpMapPoints[i] = (i % 9 != 0) ? &mps[i] : NULL;
        for (auto& m : mps) { m.mbBad = false; m.mObservations.clear(); }
        std::vector<int> kfmp(L.N);
        for (int i = 0; i < L.N; i++) kfmp[i] = KF.mvpMapPoints[i] != NULL;
        dump("kf1_has_mp", kfmp);
        {   // Tracking::TrackReferenceKeyFrame (Tracking.cc:946): SearchByBoW(pKF, F, vpMapPointMatches)
            ORBmatcher m07(0.7f, true);
            std::vector<MapPoint*> got;
            res.push_back(m07.SearchByBoW(&KF, C, got));
            std::vector<int> idx(C.N, -1);
            for (int i = 0; i < C.N; i++) idx[i] = got[i] ? (int)(got[i] - mps.data()) : -1;
            dump("bow_kf_f", idx);
        }
        {   // LoopClosing::ComputeSim3 (LoopClosing.cc:286): SearchByBoW(pKF1, pKF2, vpMatches12)
            ORBmatcher m075(0.75f, true);
            std::vector<MapPoint*> got;
            res.push_back(m075.SearchByBoW(&KF, &KF2, got));
            std::vector<int> idx(L.N, -1);
            for (int i = 0; i < L.N; i++) idx[i] = got[i] ? (int)(got[i] - mps2.data()) : -1;
            dump("bow_kf_kf", idx);
        }
        {   // LocalMapping::CreateNewMapPoints (LocalMapping.cc:287): SearchForTriangulation(pKF1, pKF2, F12, pairs, false) on keyframes
            // most of whose keypoints have no map point yet
            // (the second view: the same keypoints seen after a pure translation -- a point at any depth then lies on the epipolar line
            // through its own pixel, so the epipolar gate has something to accept, which two unrelated synthetic frames do not give it)
            KeyFrame A = KF, B2 = KF;
            const float t3[3] = {0.05f, 0.0f, 0.001f};
            B2.tcw = mat32(3, 1, {t3[0], t3[1], t3[2]}); B2.Ow = mat32(3, 1, {-t3[0], -t3[1], -t3[2]});
            for (int i = 0; i < A.N; i++) if (i % 3) A.mvpMapPoints[i] = NULL;
            for (int i = 0; i < B2.N; i++) if (i % 3 != 1) B2.mvpMapPoints[i] = NULL;
            // F12 = K1^-T [t12]x R12 K2^-1 (LocalMapping::ComputeF12) for R12 = I, t12 = -t3, written out
            const float fx = Frame::fx, fy = Frame::fy, cx0 = Frame::cx, cy0 = Frame::cy, tx = -t3[0], ty = -t3[1], tz = -t3[2];
            const float E[9] = {0, -tz, ty, tz, 0, -tx, -ty, tx, 0};
            const float Ki[9] = {1 / fx, 0, -cx0 / fx, 0, 1 / fy, -cy0 / fy, 0, 0, 1};
            float EK[9], F12a[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { float v = 0; for (int k = 0; k < 3; k++) v += E[3 * r + k] * Ki[3 * k + c]; EK[3 * r + c] = v; }
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { float v = 0; for (int k = 0; k < 3; k++) v += Ki[3 * k + r] * EK[3 * k + c]; F12a[3 * r + c] = v; }
            dump("F12", F12a, 9);
            cv::Mat F12 = mat32(3, 3, {F12a[0], F12a[1], F12a[2], F12a[3], F12a[4], F12a[5], F12a[6], F12a[7], F12a[8]});
            std::vector<std::pair<size_t, size_t> > pairs;
            ORBmatcher m06(0.6f, false);
            res.push_back(m06.SearchForTriangulation(&A, &B2, F12, pairs, false));
            std::vector<int> flat;
            for (auto& pr : pairs) { flat.push_back((int)pr.first); flat.push_back((int)pr.second); }
            dump("triangulation", flat);
        }
        {   // LoopClosing::ComputeSim3 (LoopClosing.cc:322): SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, 7.5)
            std::vector<MapPoint*> m12(L.N, (MapPoint*)NULL);
            for (int i = 0; i < L.N && i < C.N; i += 13) if (KF2.mvpMapPoints[i]) { m12[i] = KF2.mvpMapPoints[i]; mps2[i].mObservations[&KF2] = i; } // already matched
            std::vector<int> pre(L.N);
            for (int i = 0; i < L.N; i++) pre[i] = m12[i] ? (int)(m12[i] - mps2.data()) : -1;
            dump("sim3_pre", pre);
            cv::Mat R12 = mat32(3, 3, {1, 0, 0, 0, 1, 0, 0, 0, 1}), t12 = mat32(3, 1, {-t2[0], -t2[1], -t2[2]});
            const float s12 = 1.0f;
            res.push_back(matcher.SearchBySim3(&KF, &KF2, m12, s12, R12, t12, 7.5f));
            std::vector<int> idx(L.N);
            for (int i = 0; i < L.N; i++) idx[i] = m12[i] ? (int)(m12[i] - mps2.data()) : -1;
            dump("sim3", idx);
            for (auto& m : mps2) m.mObservations.clear();
        }
        // the two Scw members look at a keyframe through a similarity that is ALMOST the keyframe's own pose (frame 0 seen from 0.5 mm
        // to the side, scale 1): loop-closure candidates then project next to the keypoints they came from, as they do after a good Sim3
        const float ts[3] = {0.0005f, -0.0003f, 0.0f};
        cv::Mat Scws = mat32(4, 4, {1, 0, 0, ts[0], 0, 1, 0, ts[1], 0, 0, 1, ts[2], 0, 0, 0, 1});
        std::vector<MapPoint> mps3(mps.begin(), mps.end()); // the keyframe's own points (other objects than the candidates)
        KeyFrame KF3 = KF;
        for (int i = 0; i < KF3.N; i++) KF3.mvpMapPoints[i] = (i % 4 != 0) ? &mps3[i] : NULL;
        {   // LoopClosing::ComputeSim3 (LoopClosing.cc:372): SearchByProjection(mpCurrentKF, mScw, vpLoopMapPoints, vpMatched, 10)
            std::vector<MapPoint*> pts;
            for (auto& m : mps) pts.push_back(&m);
            std::vector<MapPoint*> matched(KF3.N, (MapPoint*)NULL);
            for (int i = 0; i < KF3.N; i += 17) matched[i] = &mps[i];
            std::vector<int> pre(KF3.N);
            for (int i = 0; i < KF3.N; i++) pre[i] = matched[i] ? (int)(matched[i] - mps.data()) : -1;
            dump("proj_sim3_pre", pre);
            res.push_back(matcher.SearchByProjection(&KF3, Scws, pts, matched, 10));
            std::vector<int> idx(KF3.N);
            for (int i = 0; i < KF3.N; i++) idx[i] = matched[i] ? (int)(matched[i] - mps.data()) : -1;
            dump("proj_sim3", idx);
        }
        {   // LoopClosing::SearchAndFuse (LoopClosing.cc:700): Fuse(pKF, Scw, vpLoopMapPoints, 4, vpReplacePoints)
            std::vector<MapPoint*> pts;
            for (auto& m : mps) pts.push_back(&m);
            std::vector<MapPoint*> rep(pts.size(), (MapPoint*)NULL);
            res.push_back(matcher.Fuse(&KF3, Scws, pts, 4.0f, rep));
            std::vector<int> idx(pts.size());
            int added = 0;
            for (size_t i = 0; i < pts.size(); i++) {
                // a replacement names a point of the keyframe -- or, when an earlier candidate was added at that keypoint, that candidate (-2)
                idx[i] = !rep[i] ? -1 : (rep[i] >= mps3.data() && rep[i] < mps3.data() + mps3.size()) ? (int)(rep[i] - mps3.data()) : -2;
                added += mps[i].mObservations.count(&KF3) ? 1 : 0;
            }
            dump("fuse_scw_replace", idx);
            res.push_back(added);
        }
        orbfe_vocabulary_destroy(voc);
    }
    {   // ORBextractor::mvImagePyramid (ORBextractor.h:85) is filled on request: keepImagePyramid(true), then operator()
        ORBextractor ex2(1000, 1.2f, 8, 20, 7);
        ex2.keepImagePyramid(true);
        cv::Mat im(rows, cols, CV_8UC1, raw.data(), (size_t)cols);
        std::vector<cv::KeyPoint> k;
        cv::Mat d;
        ex2(im, cv::Mat(), k, d);
        if ((int)ex2.mvImagePyramid.size() != 8 || ex2.mvImagePyramid[0].rows != rows || ex2.mvImagePyramid[0].cols != cols) return 30;
        std::vector<int> dims;
        for (auto& m : ex2.mvImagePyramid) { dims.push_back(m.cols); dims.push_back(m.rows); }
        dump("pyr_dims", dims);
        dump("pyr_level3", ex2.mvImagePyramid[3].data, (size_t)ex2.mvImagePyramid[3].rows * ex2.mvImagePyramid[3].cols);
        if (memcmp(ex2.mvImagePyramid[0].data, raw.data(), (size_t)rows * cols) != 0) return 31;
    }
    dump("results", res);
    printf("ok %d frames, %d keypoints in frame 0\n", nframes, F[0].N);
    return 0;
}
