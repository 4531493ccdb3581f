/*
 * match_oracle.cpp -- CPU restatement of the reference's descriptor matching primitives.
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp header for the rules).
 *
 * Restates: ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1651-1667), the
 * best / second-best update rule shared by every SearchBy* (App. D), the Frame
 * keypoint grid (src/Frame.cc:183-198, 280-345), ORBmatcher::SearchForInitialization
 * (src/ORBmatcher.cc:409-524) and ComputeThreeMaxima (:1605-1646).
 * These are integer algorithms fully contained in the reference, so this part of
 * the oracle is pinned by known-answer tests (popcount identity, hand-built cases).
 */
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct KeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30; /* ORBmatcher.cc:37-39 */
const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;    /* Frame.h:40-41 */

/* MapPoint::GetMinDistanceInvariance / GetMaxDistanceInvariance (src/MapPoint.cc:402-412): the per-point arrays of the guided
 * searches carry the members mfMinDistance / mfMaxDistance themselves. */
inline float GetMinDistanceInvariance(float mfMinDistance) { return 0.8f * mfMinDistance; }
inline float GetMaxDistanceInvariance(float mfMaxDistance) { return 1.2f * mfMaxDistance; }

/* MapPoint::PredictScale (src/MapPoint.cc:414-446).  It divides mfMaxDistance -- not the 1.2f * mfMaxDistance of the range gate.
 * MapPoint.cc sees `using namespace std` (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36, included through KeyFrame.h), so
 * log(ratio) and ceil(...) on a float resolve to the float overloads: logf, a float division by mfLogScaleFactor, ceilf. */
inline int PredictScale(float mfMaxDistance, float currentDist, float mfLogScaleFactor, int mnScaleLevels)
{
    float ratio = mfMaxDistance / currentDist;
    int nScale = (int)std::ceil(std::log(ratio) / mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= mnScaleLevels) nScale = mnScaleLevels - 1;
    return nScale;
}

/* ORBmatcher.cc:1651-1667 (the bit-hack itself, not __builtin_popcount) */
int DescriptorDistance(const uint8_t* a, const uint8_t* b)
{
    const int32_t* pa = (const int32_t*)a;
    const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

/* ORBmatcher.cc:1605-1646 */
void ComputeThreeMaxima(const int* histo_sizes, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo_sizes[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

/* Frame grid for an undistorted camera: mnMinX=0, mnMaxX=cols, mnMinY=0, mnMaxY=rows (Frame.cc:440-446) */
struct FrameGrid {
    float mnMinX, mnMinY, mnMaxX, mnMaxY, invW, invH;
    const KeyPoint* kps;
    int N;
    std::vector<int> cell[FRAME_GRID_COLS][FRAME_GRID_ROWS];

    /* bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY} of the undistorted image (Frame::ComputeImageBounds, Frame.cc:418-451);
     * NULL = no distortion: 0, 0, cols, rows (Frame.cc:440-446) */
    FrameGrid(const KeyPoint* k, int n, int cols, int rows, const float* bounds = nullptr) : kps(k), N(n)
    {
        mnMinX = 0.f; mnMaxX = (float)cols; mnMinY = 0.f; mnMaxY = (float)rows;
        if (bounds) { mnMinX = bounds[0]; mnMinY = bounds[1]; mnMaxX = bounds[2]; mnMaxY = bounds[3]; }
        invW = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(mnMaxX - mnMinX); /* Frame.cc:112 */
        invH = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(mnMaxY - mnMinY);
        for (int i = 0; i < N; i++) { /* Frame.cc:183-198, :335-345 */
            int px = (int)std::round((kps[i].x - mnMinX) * invW);
            int py = (int)std::round((kps[i].y - mnMinY) * invH);
            if (px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS) continue;
            cell[px][py].push_back(i);
        }
    }
    /* Frame.cc:280-333 */
    std::vector<int> GetFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel) const
    {
        std::vector<int> v;
        const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * invW));
        if (nMinCellX >= FRAME_GRID_COLS) return v;
        const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - mnMinX + r) * invW));
        if (nMaxCellX < 0) return v;
        const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * invH));
        if (nMinCellY >= FRAME_GRID_ROWS) return v;
        const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - mnMinY + r) * invH));
        if (nMaxCellY < 0) return v;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<int>& vCell = cell[ix][iy];
                for (size_t j = 0; j < vCell.size(); j++) {
                    const KeyPoint& kp = kps[vCell[j]];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(vCell[j]);
                }
            }
        return v;
    }
};

} // namespace

extern "C" {

int oracle_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

/* All-pairs best / second-best with the reference update rule (strict '<', first candidate wins):
 *   if d<best {second=best; best=d; idx=i} else if d<second {second=d}
 * init = starting value of best and second (256 or INT_MAX in the reference, App. D). */
void oracle_knn2(const uint8_t* Q, int nq, const uint8_t* T, int nt, int init, int32_t* best_idx, int32_t* best_dist,
                 int32_t* second_dist)
{
    for (int q = 0; q < nq; q++) {
        int best = init, second = init, idx = -1;
        for (int t = 0; t < nt; t++) {
            int d = DescriptorDistance(Q + (size_t)q * 32, T + (size_t)t * 32);
            if (d < best) { second = best; best = d; idx = t; }
            else if (d < second) second = d;
        }
        best_idx[q] = idx; best_dist[q] = best; second_dist[q] = second;
    }
}

/* Candidate lists of Frame::GetFeaturesInArea for a batch of queries, CSR output.
 * offsets has nq+1 entries; returns total count (idx may be NULL to size it). */
int oracle_features_in_area(const void* kps2, int n2, int cols, int rows, const float* qx, const float* qy, int nq,
                            float r, int minLevel, int maxLevel, int32_t* offsets, int32_t* idx, int capacity,
                            const float* bounds)
{
    FrameGrid g((const KeyPoint*)kps2, n2, cols, rows, bounds);
    int total = 0;
    for (int q = 0; q < nq; q++) {
        offsets[q] = total;
        std::vector<int> v = g.GetFeaturesInArea(qx[q], qy[q], r, minLevel, maxLevel);
        for (int i : v) {
            if (idx && total < capacity) idx[total] = i;
            total++;
        }
    }
    offsets[nq] = total;
    return total;
}

/* Guided best/second-best over CSR candidate lists, plain rule (no cross-query state). */
void oracle_knn2_csr(const uint8_t* Q, int nq, const uint8_t* T, const int32_t* offsets, const int32_t* idx, int init,
                     int32_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    for (int q = 0; q < nq; q++) {
        int best = init, second = init, bi = -1;
        for (int k = offsets[q]; k < offsets[q + 1]; k++) {
            int d = DescriptorDistance(Q + (size_t)q * 32, T + (size_t)idx[k] * 32);
            if (d < best) { second = best; best = d; bi = idx[k]; }
            else if (d < second) second = d;
        }
        best_idx[q] = bi; best_dist[q] = best; second_dist[q] = second;
    }
}

/* The matching loop of ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th), ORBmatcher.cc:45-129, with
 * the MapPoint / Frame objects flattened to arrays (the projection, the viewing-cosine radius and the bookkeeping of map
 * points stay with the caller).  Per query q (a map point in view): window centre (x, y), radius r (already multiplied by
 * the scale factor of the predicted level, :71), level range [min_level, max_level] (:71: predicted-1 .. predicted),
 * descriptor.  Candidates = Frame::GetFeaturesInArea (Frame.cc:280-333).  taken[i] != 0 marks keypoints that already
 * carry a map point with observations (:91-93).
 *   mode 0: raw -- best / second-best and their octaves per query (:84-118), no cross-query state; this is also the inner
 *           loop of the (CurrentFrame, LastFrame) and (CurrentFrame, KeyFrame) variants (:1413-1433, :1551-1571).
 *   mode 1: the whole loop -- accept if best <= th_high and not (same octave and best > nnratio * second) (:121-128), and
 *           the accepted keypoint becomes taken for the following queries (F.mvpMapPoints[bestIdx] = pMP; map points in
 *           view have observations).  match[q] = keypoint index or -1.  taken[] is updated in place.  Returns nmatches. */
struct WindowQuery { float x, y, r; int32_t min_level, max_level; };

int oracle_search_by_projection(const void* kps_, const uint8_t* desc, int n, int cols, int rows, const void* queries_,
                                const uint8_t* qdesc, int nq, uint8_t* taken, int mode, int th_high, float nnratio,
                                int32_t* best_idx, int32_t* best_dist, int32_t* best_level, int32_t* second_dist,
                                int32_t* second_level, int32_t* match, const float* bounds, const uint8_t* q_observed)
{
    /* q_observed[q] = the query's map point has Observations() > 0: only then does the keypoint it is assigned to block the
     * queries after it (:91-93 test F.mvpMapPoints[idx]->Observations() > 0); NULL = every map point is observed */
    const KeyPoint* k = (const KeyPoint*)kps_;
    const WindowQuery* Q = (const WindowQuery*)queries_;
    FrameGrid F(k, n, cols, rows, bounds);
    int nmatches = 0;
    std::vector<uint8_t> blocked(n, 0); /* F.mvpMapPoints[idx] && F.mvpMapPoints[idx]->Observations() > 0; taken == NULL: none yet */
    for (int i = 0; i < n; i++) blocked[i] = taken ? taken[i] : 0;
    for (int q = 0; q < nq; q++) {
        const std::vector<int> vIndices = F.GetFeaturesInArea(Q[q].x, Q[q].y, Q[q].r, Q[q].min_level, Q[q].max_level);
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int idx : vIndices) {
            if (blocked[idx]) continue;
            const int dist = DescriptorDistance(qdesc + (size_t)q * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = k[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = k[idx].octave;
                bestDist2 = dist;
            }
        }
        if (best_idx) { best_idx[q] = bestIdx; best_dist[q] = bestDist; best_level[q] = bestLevel; second_dist[q] = bestDist2; second_level[q] = bestLevel2; }
        if (mode == 1) {
            match[q] = -1;
            if (bestIdx >= 0 && bestDist <= th_high) {
                if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
                match[q] = bestIdx;
                if (!q_observed || q_observed[q]) { blocked[bestIdx] = 1; if (taken) taken[bestIdx] = 1; }
                nmatches++;
            }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchForInitialization, ORBmatcher.cc:409-524.
 * prevMatched (n1 x 2 floats) is updated in place like vbPrevMatched; matches12 gets n1 ints. */
int oracle_search_for_initialization(const void* kps1_, const uint8_t* desc1, int n1, const void* kps2_,
                                     const uint8_t* desc2, int n2, int cols, int rows, float* prevMatched,
                                     int32_t* matches12, int windowSize, float nnratio, int checkOrientation,
                                     const float* bounds)
{
    const KeyPoint* k1 = (const KeyPoint*)kps1_;
    const KeyPoint* k2 = (const KeyPoint*)kps2_;
    FrameGrid F2(k2, n2, cols, rows, bounds);
    int nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX);
    std::vector<int> vnMatches21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        int level1 = k1[i1].octave;
        if (level1 > 0) continue;
        std::vector<int> vIndices2 =
            F2.GetFeaturesInArea(prevMatched[2 * i1], prevMatched[2 * i1 + 1], (float)windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int i2 : vIndices2) {
            int dist = DescriptorDistance(d1, desc2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    matches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOrientation) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOrientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i])
                if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            prevMatched[2 * i1] = k2[matches12[i1]].x;
            prevMatched[2 * i1 + 1] = k2[matches12[i1]].y;
        }
    return nmatches;
}

/* cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) as Frame::UndistortKeyPoints (src/Frame.cc:357-387),
 * Frame::UndistortArucoCorners (:389-416) and Frame::ComputeImageBounds (:418-451) call it.  OpenCV is not in this image;
 * this restates the published OpenCV 3.4 algorithm (imgproc/src/undistort.cpp, cvUndistortPointsInternal with the default
 * criteria MAX_ITER 5): the matrices are converted to double, each point is normalised by K, the distortion is inverted
 * by five fixed-point iterations, and the point is re-projected through the 3x3 product P*R = K. */
void oracle_undistort_points(const float* src, int n, const float* K4, const float* dist, int ndist, float* dst)
{
    double A[3][3] = {{K4[0], 0, K4[2]}, {0, K4[1], K4[3]}, {0, 0, 1}};
    double RR[3][3] = {{K4[0], 0, K4[2]}, {0, K4[1], K4[3]}, {0, 0, 1}};
    double k[14] = {0};
    for (int i = 0; i < ndist && i < 14; i++) k[i] = dist[i];
    const double fx = A[0][0], fy = A[1][1], ifx = 1. / fx, ify = 1. / fy, cx = A[0][2], cy = A[1][2];
    for (int i = 0; i < n; i++) {
        double x = src[2 * i], y = src[2 * i + 1];
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        if (ndist > 0) {
            double x0 = x, y0 = y;
            for (int j = 0; j < 5; j++) {
                double r2 = x * x + y * y;
                double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
                double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
                double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
                x = (x0 - deltaX) * icdist;
                y = (y0 - deltaY) * icdist;
            }
        }
        double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        dst[2 * i] = (float)(xx * ww);
        dst[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::ComputeImageBounds (src/Frame.cc:418-451): the four image corners undistorted when mDistCoef[0] != 0;
 * out = {mnMinX, mnMinY, mnMaxX, mnMaxY}. */
void oracle_compute_image_bounds(int cols, int rows, const float* K4, const float* dist, int ndist, float* out)
{
    if (ndist > 0 && dist[0] != 0.0f) {
        float c[8] = {0, 0, (float)cols, 0, 0, (float)rows, (float)cols, (float)rows}, u[8];
        oracle_undistort_points(c, 4, K4, dist, ndist, u);
        out[0] = std::min(u[0], u[4]);
        out[2] = std::max(u[2], u[6]);
        out[1] = std::min(u[1], u[3]);
        out[3] = std::max(u[5], u[7]);
    } else {
        out[0] = 0.0f; out[1] = 0.0f; out[2] = (float)cols; out[3] = (float)rows;
    }
}

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:159-292) and
 * SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (:526-659) on flat arrays.  Side 1 = the keyframe whose map points are
 * looked for, side 2 = the frame / second keyframe.  valid1[i] = "has a map point that is not bad" (:196-201, :564-568),
 * valid2 likewise for the KF-KF variant (:583-590; NULL = every feature, the KF-Frame variant).  The two variants differ
 * in the accept test (best <= TH_LOW :233 vs best < TH_LOW :608: pass accept_max = 50 or 49) and in the histogram factor
 * (HISTO_LENGTH/360.0f :174 in this fork vs 1.0f/HISTO_LENGTH :546).  Outputs: match12[i1] = i2 or -1, match21[i2] = i1 or
 * -1 (the reference's vpMapPointMatches[i2] = map point of match21[i2]; vpMatches12[i1] = map point of match12[i1]). */
int oracle_search_by_bow(const void* kps1_, const uint8_t* desc1, const uint8_t* valid1, int n1, const uint32_t* fv_node1,
                         const int32_t* fv_off1, const uint32_t* fv_feat1, int nfv1, const void* kps2_, const uint8_t* desc2,
                         const uint8_t* valid2, int n2, const uint32_t* fv_node2, const int32_t* fv_off2, const uint32_t* fv_feat2,
                         int nfv2, float nnratio, int check_orientation, int accept_max, float factor, int32_t* match12,
                         int32_t* match21)
{
    const KeyPoint* k1 = (const KeyPoint*)kps1_;
    const KeyPoint* k2 = (const KeyPoint*)kps2_;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    for (int i = 0; i < n2; i++) match21[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    int a = 0, b = 0;
    while (a != nfv1 && b != nfv2) {
        if (fv_node1[a] == fv_node2[b]) {
            for (int p1 = fv_off1[a]; p1 < fv_off1[a + 1]; p1++) {
                const int idx1 = (int)fv_feat1[p1];
                if (valid1 && !valid1[idx1]) continue;
                const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int p2 = fv_off2[b]; p2 < fv_off2[b + 1]; p2++) {
                    const int idx2 = (int)fv_feat2[p2];
                    if (match21[idx2] >= 0 || (valid2 && !valid2[idx2])) continue;
                    const int dist = DescriptorDistance(d1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = idx2;
                    } else if (dist < bestDist2) {
                        bestDist2 = dist;
                    }
                }
                if (bestDist1 <= accept_max) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        match12[idx1] = bestIdx2;
                        match21[bestIdx2] = idx1;
                        if (check_orientation) {
                            float rot = k1[idx1].angle - k2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            a++;
            b++;
        } else if (fv_node1[a] < fv_node2[b]) {
            a = (int)(std::lower_bound(fv_node1, fv_node1 + nfv1, fv_node2[b]) - fv_node1);
        } else {
            b = (int)(std::lower_bound(fv_node2, fv_node2 + nfv2, fv_node1[a]) - fv_node2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                match21[match12[idx1]] = -1;
                match12[idx1] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono = true) (src/ORBmatcher.cc:1332-1474)
 * on flat arrays.  Per feature i of the last frame: valid_last[i] = "has a map point and is not an outlier" (:1358-1362),
 * x3Dw = the map point's world position, mp_desc = its descriptor (:1391), mp_observed[i] = its Observations() > 0 -- what
 * makes CurrentFrame.mvpMapPoints[i2] block later queries (:1397-1399; NULL = all observed).  taken_cur = keypoints of the
 * current frame that already hold an observed map point.  Tcw = 3x4 row-major [Rcw | tcw], K4 = fx fy cx cy.  The stereo
 * branches (bForward / bBackward, mvuRight) do not exist in the monocular system.  OpenCV's `Rcw*x3Dw+tcw` on 3x3 * 3x1
 * floats: each row is a float sum left to right, then the float translation is added.
 * match_cur[i2] = i (last-frame feature whose map point the keypoint received) or -1; returns nmatches. */
int oracle_search_by_projection_last_frame(const void* kps_cur_, const uint8_t* desc_cur, int n_cur, const uint8_t* taken_cur, int cols,
                                           int rows, const float* bounds, const void* kps_last_, int n_last, const uint8_t* valid_last,
                                           const float* x3Dw, const uint8_t* mp_desc, const uint8_t* mp_observed, const float* Tcw,
                                           const float* K4, const float* scale_factors, float th, int th_high, int check_orientation,
                                           int32_t* match_cur)
{
    const KeyPoint* kc = (const KeyPoint*)kps_cur_;
    const KeyPoint* kl = (const KeyPoint*)kps_last_;
    FrameGrid grid(kc, n_cur, cols, rows, bounds);
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> blocked(n_cur, 0); /* CurrentFrame.mvpMapPoints[i2] && Observations() > 0 */
    for (int i = 0; i < n_cur; i++) { match_cur[i] = -1; blocked[i] = taken_cur ? taken_cur[i] : 0; }
    for (int i = 0; i < n_last; i++) {
        if (valid_last && !valid_last[i]) continue;
        const float X = x3Dw[3 * i], Y = x3Dw[3 * i + 1], Z = x3Dw[3 * i + 2];
        float t0 = Tcw[0] * X + Tcw[1] * Y + Tcw[2] * Z;
        float t1 = Tcw[4] * X + Tcw[5] * Y + Tcw[6] * Z;
        float t2 = Tcw[8] * X + Tcw[9] * Y + Tcw[10] * Z;
        const float xc = (float)(t0 * 1.0 + 1.0 * Tcw[3]);
        const float yc = (float)(t1 * 1.0 + 1.0 * Tcw[7]);
        const float zc = (float)(t2 * 1.0 + 1.0 * Tcw[11]);
        const float invzc = 1.0 / zc;
        if (invzc < 0) continue;
        float u = fx * xc * invzc + cx;
        float v = fy * yc * invzc + cy;
        if (u < grid.mnMinX || u > grid.mnMaxX) continue;
        if (v < grid.mnMinY || v > grid.mnMaxY) continue;
        if (u != u || v != v) continue; /* NaN: undefined in the reference (grid indices from NaN); dropped */
        int nLastOctave = kl[i].octave;
        float radius = th * scale_factors[nLastOctave];
        std::vector<int> vIndices2 = grid.GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const int i2 = vIndices2[k];
            if (blocked[i2]) continue;
            const int dist = DescriptorDistance(dMP, desc_cur + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= th_high) {
            match_cur[bestIdx2] = i;
            blocked[bestIdx2] = mp_observed ? mp_observed[i] : 1;
            nmatches++;
            if (check_orientation) {
                float rot = kl[i].angle - kc[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match_cur[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine (src/ORBmatcher.cc:139-157) */
static bool CheckDistEpipolarLine(const KeyPoint& kp1, const KeyPoint& kp2, const float* F12, const float* mvLevelSigma2)
{
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float num = a * kp2.x + b * kp2.y + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * mvLevelSigma2[kp2.octave];
}

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:661-827), monocular (bOnlyStereo = false, no right coordinates), on
 * flat arrays.  has_mp1 / has_mp2 = "the feature already has a map point" (:710-714, :731-735).  F12 3x3 row-major, (ex, ey)
 * the epipole in the second image (:669-675).  vbMatched2 is never set in the reference, so candidates are not consumed.
 * match12[i1] = i2 or -1 (vMatchedPairs = the pairs in ascending i1); returns nmatches. */
int oracle_search_for_triangulation(const void* kps1_, const uint8_t* desc1, const uint8_t* has_mp1, int n1, const uint32_t* fv_node1,
                                    const int32_t* fv_off1, const uint32_t* fv_feat1, int nfv1, const void* kps2_, const uint8_t* desc2,
                                    const uint8_t* has_mp2, int n2, const uint32_t* fv_node2, const int32_t* fv_off2,
                                    const uint32_t* fv_feat2, int nfv2, const float* F12, float ex, float ey, const float* mvScaleFactors,
                                    const float* mvLevelSigma2, int check_orientation, int32_t* vMatches12)
{
    const KeyPoint* k1 = (const KeyPoint*)kps1_;
    const KeyPoint* k2 = (const KeyPoint*)kps2_;
    int nmatches = 0;
    std::vector<bool> vbMatched2(n2, false);
    for (int i = 0; i < n1; i++) vMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int a = 0, b = 0;
    while (a != nfv1 && b != nfv2) {
        if (fv_node1[a] == fv_node2[b]) {
            for (int p1 = fv_off1[a]; p1 < fv_off1[a + 1]; p1++) {
                const int idx1 = (int)fv_feat1[p1];
                if (has_mp1 && has_mp1[idx1]) continue;
                const KeyPoint& kp1 = k1[idx1];
                const uint8_t* d1 = desc1 + 32 * (size_t)idx1;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int p2 = fv_off2[b]; p2 < fv_off2[b + 1]; p2++) {
                    const int idx2 = (int)fv_feat2[p2];
                    if (vbMatched2[idx2] || (has_mp2 && has_mp2[idx2])) continue;
                    const int dist = DescriptorDistance(d1, desc2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const KeyPoint& kp2 = k2[idx2];
                    {
                        const float distex = ex - kp2.x;
                        const float distey = ey - kp2.y;
                        if (distex * distex + distey * distey < 100 * mvScaleFactors[kp2.octave]) continue;
                    }
                    if (CheckDistEpipolarLine(kp1, kp2, F12, mvLevelSigma2)) {
                        bestIdx2 = idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    const KeyPoint& kp2 = k2[bestIdx2];
                    vMatches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kp1.angle - kp2.angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            a++;
            b++;
        } else if (fv_node1[a] < fv_node2[b]) {
            a = (int)(std::lower_bound(fv_node1, fv_node1 + nfv1, fv_node2[b]) - fv_node1);
        } else {
            b = (int)(std::lower_bound(fv_node2, fv_node2 + nfv2, fv_node1[a]) - fv_node2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                vMatches12[rotHist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    return nmatches;
}

/* The matching part of ORBmatcher::Fuse(KeyFrame*, vpMapPoints, th) (src/ORBmatcher.cc:829-970) and of Fuse(KeyFrame*, Scw,
 * vpPoints, th, vpReplacePoint) (:972-1104) on flat arrays: per map point the projection and gates (:848-893 / :1008-1049),
 * MapPoint::PredictScale (MapPoint.cc:414-429; `log` evaluated in double), KeyFrame::GetFeaturesInArea (KeyFrame.cc:672-711,
 * the Frame grid), levels [predicted - 1, predicted], the mono reprojection gate e2 * invSigma2 > chi2 (:920-931; chi2 = 0:
 * the Scw variant has none), best distance.  What is done with (bestIdx, bestDist <= TH_LOW) -- Replace / AddObservation --
 * is map bookkeeping and stays with the caller.  valid[i] = pMP && !isBad() && !IsInKeyFrame / !alreadyFound.
 * Tcw = 3x4 row-major [Rcw | tcw], Ow = camera centre.  best_idx[i] = keypoint or -1, best_dist[i] (256 if none). */
void oracle_fuse_search(const void* kps_, const uint8_t* desc, int n, int cols, int rows, const float* bounds, const float* p3Dw,
                        const uint8_t* valid, const float* min_dist, const float* max_dist, const float* normal, const uint8_t* mp_desc,
                        int nmp, const float* Tcw, const float* Ow, const float* K4, const float* mvScaleFactors,
                        const float* mvInvLevelSigma2, int nlevels, float mfLogScaleFactor, float th, double chi2, int32_t* best_idx,
                        int32_t* best_dist)
{
    const KeyPoint* kps = (const KeyPoint*)kps_;
    FrameGrid grid(kps, n, cols, rows, bounds);
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    for (int i = 0; i < nmp; i++) {
        best_idx[i] = -1; best_dist[i] = 256;
        if (valid && !valid[i]) continue;
        const float X = p3Dw[3 * i], Y = p3Dw[3 * i + 1], Z = p3Dw[3 * i + 2];
        float t0 = Tcw[0] * X + Tcw[1] * Y + Tcw[2] * Z, t1 = Tcw[4] * X + Tcw[5] * Y + Tcw[6] * Z, t2 = Tcw[8] * X + Tcw[9] * Y + Tcw[10] * Z;
        const float p3Dc[3] = {(float)(t0 * 1.0 + 1.0 * Tcw[3]), (float)(t1 * 1.0 + 1.0 * Tcw[7]), (float)(t2 * 1.0 + 1.0 * Tcw[11])};
        if (p3Dc[2] < 0.0f) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz;
        const float y = p3Dc[1] * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!(u >= grid.mnMinX && u < grid.mnMaxX && v >= grid.mnMinY && v < grid.mnMaxY)) continue; /* KeyFrame::IsInImage */
        const float maxDistance = GetMaxDistanceInvariance(max_dist[i]), minDistance = GetMinDistanceInvariance(min_dist[i]);
        const float PO[3] = {X - Ow[0], Y - Ow[1], Z - Ow[2]};
        const float dist3D = std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]); /* cv::norm */
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const double dot = (double)PO[0] * normal[3 * i] + (double)PO[1] * normal[3 * i + 1] + (double)PO[2] * normal[3 * i + 2];
        if (dot < 0.5 * dist3D) continue;
        const int nPredictedLevel = PredictScale(max_dist[i], dist3D, mfLogScaleFactor, nlevels);
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices = grid.GetFeaturesInArea(u, v, radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t* dMP = mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const int idx = vIndices[k];
            const KeyPoint& kp = kps[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (chi2 > 0) {
                const float ex = u - kp.x;
                const float ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * mvInvLevelSigma2[kpLevel] > chi2) continue;
            }
            const int dist = DescriptorDistance(dMP, desc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
}

/* ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1106-1330) on flat arrays.  Feature i of a keyframe carries its map point's world
 * position, scale-invariance range and descriptor; valid*[i] = "has a map point that is not bad and is not already matched"
 * (:1148-1155, :1230-1236).  T1w / T2w = 3x4 row-major [R | t] of the keyframes; sT12 = [sR12 | t12], sT21 = [sR21 | t21]
 * as computed at :1123-1126.  Both cameras share K.  vnMatch = best keypoint at levels [predicted - 1, predicted] with
 * distance <= TH_HIGH; a pair is kept when both directions agree (:1302-1318).  match12[i1] = i2 or -1; returns nFound. */
int oracle_search_by_sim3(const void* kps1_, const uint8_t* desc1, int n1, const void* kps2_, const uint8_t* desc2, int n2, int cols, int rows,
                          const float* bounds, const float* p3Dw1, const uint8_t* valid1, const float* min1, const float* max1,
                          const uint8_t* mpd1, const float* p3Dw2, const uint8_t* valid2, const float* min2, const float* max2,
                          const uint8_t* mpd2, const float* T1w, const float* T2w, const float* sT12, const float* sT21, const float* K4,
                          const float* mvScaleFactors, int nlevels, float mfLogScaleFactor, float th, int32_t* match12)
{
    const KeyPoint* k1 = (const KeyPoint*)kps1_;
    const KeyPoint* k2 = (const KeyPoint*)kps2_;
    FrameGrid grid1(k1, n1, cols, rows, bounds), grid2(k2, n2, cols, rows, bounds);
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    auto mul = [](const float* T, const float* p, float* o) { /* cv: 3x3 * 3x1 float rows, then + t */
        for (int r = 0; r < 3; r++) {
            float t = T[4 * r] * p[0] + T[4 * r + 1] * p[1] + T[4 * r + 2] * p[2];
            o[r] = (float)(t * 1.0 + 1.0 * T[4 * r + 3]);
        }
    };
    auto direction = [&](int nA, const float* pw, const uint8_t* valid, const float* mind, const float* maxd, const uint8_t* mpd,
                         const float* TAw, const float* sTBA, const FrameGrid& gridB, const KeyPoint* kB, const uint8_t* descB,
                         std::vector<int>& vnMatch) {
        for (int i = 0; i < nA; i++) {
            if (valid && !valid[i]) continue;
            float cA[3], cB[3];
            mul(TAw, pw + 3 * i, cA);
            mul(sTBA, cA, cB);
            if (cB[2] < 0.0) continue;
            const float invz = 1.0 / cB[2];
            const float x = cB[0] * invz;
            const float y = cB[1] * invz;
            const float u = fx * x + cx;
            const float v = fy * y + cy;
            if (!(u >= gridB.mnMinX && u < gridB.mnMaxX && v >= gridB.mnMinY && v < gridB.mnMaxY)) continue;
            const float dist3D = std::sqrt((double)cB[0] * cB[0] + (double)cB[1] * cB[1] + (double)cB[2] * cB[2]);
            if (dist3D < GetMinDistanceInvariance(mind[i]) || dist3D > GetMaxDistanceInvariance(maxd[i])) continue;
            const int nPredictedLevel = PredictScale(maxd[i], dist3D, mfLogScaleFactor, nlevels);
            const float radius = th * mvScaleFactors[nPredictedLevel];
            const std::vector<int> vIndices = gridB.GetFeaturesInArea(u, v, radius, -1, -1);
            if (vIndices.empty()) continue;
            int bestDist = INT_MAX, bestIdx = -1;
            for (size_t k = 0; k < vIndices.size(); k++) {
                const int idx = vIndices[k];
                if (kB[idx].octave < nPredictedLevel - 1 || kB[idx].octave > nPredictedLevel) continue;
                const int dist = DescriptorDistance(mpd + 32 * (size_t)i, descB + 32 * (size_t)idx);
                if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
            }
            if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
        }
    };
    std::vector<int> vnMatch1(n1, -1), vnMatch2(n2, -1);
    direction(n1, p3Dw1, valid1, min1, max1, mpd1, T1w, sT21, grid2, k2, desc2, vnMatch1);
    direction(n2, p3Dw2, valid2, min2, max2, mpd2, T2w, sT12, grid1, k1, desc1, vnMatch2);
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        match12[i1] = -1;
        int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { match12[i1] = idx2; nFound++; }
        }
    }
    return nFound;
}

/* ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:294-407) on flat arrays; see
 * oracle_fuse_search for the per-point inputs.  matched[i] = vpMatched[i] != NULL on entry.  match_kf[idx] = iMP or -1. */
int oracle_search_by_projection_sim3(const void* kps_, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                                     const uint8_t* matched, const float* p3Dw, const uint8_t* valid, const float* min_dist,
                                     const float* max_dist, const float* normal, const uint8_t* mp_desc, int nmp, const float* Tcw,
                                     const float* Ow, const float* K4, const float* mvScaleFactors, int nlevels, float mfLogScaleFactor, int th,
                                     int32_t* match_kf)
{
    const KeyPoint* kps = (const KeyPoint*)kps_;
    FrameGrid grid(kps, n, cols, rows, bounds);
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    std::vector<uint8_t> vpMatched(n, 0);
    for (int i = 0; i < n; i++) { match_kf[i] = -1; vpMatched[i] = matched ? matched[i] : 0; }
    int nmatches = 0;
    for (int iMP = 0; iMP < nmp; iMP++) {
        if (valid && !valid[iMP]) continue;
        const float X = p3Dw[3 * iMP], Y = p3Dw[3 * iMP + 1], Z = p3Dw[3 * iMP + 2];
        float t0 = Tcw[0] * X + Tcw[1] * Y + Tcw[2] * Z, t1 = Tcw[4] * X + Tcw[5] * Y + Tcw[6] * Z, t2 = Tcw[8] * X + Tcw[9] * Y + Tcw[10] * Z;
        const float p3Dc[3] = {(float)(t0 * 1.0 + 1.0 * Tcw[3]), (float)(t1 * 1.0 + 1.0 * Tcw[7]), (float)(t2 * 1.0 + 1.0 * Tcw[11])};
        if (p3Dc[2] < 0.0) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz;
        const float y = p3Dc[1] * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!(u >= grid.mnMinX && u < grid.mnMaxX && v >= grid.mnMinY && v < grid.mnMaxY)) continue;
        const float PO[3] = {X - Ow[0], Y - Ow[1], Z - Ow[2]};
        const float dist = std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]);
        if (dist < GetMinDistanceInvariance(min_dist[iMP]) || dist > GetMaxDistanceInvariance(max_dist[iMP])) continue;
        const double dot = (double)PO[0] * normal[3 * iMP] + (double)PO[1] * normal[3 * iMP + 1] + (double)PO[2] * normal[3 * iMP + 2];
        if (dot < 0.5 * dist) continue;
        const int nPredictedLevel = PredictScale(max_dist[iMP], dist, mfLogScaleFactor, nlevels);
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices = grid.GetFeaturesInArea(u, v, radius, -1, -1);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (size_t k = 0; k < vIndices.size(); k++) {
            const int idx = vIndices[k];
            if (vpMatched[idx]) continue;
            const int kpLevel = kps[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int d = DescriptorDistance(mp_desc + 32 * (size_t)iMP, desc + 32 * (size_t)idx);
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        }
        if (bestDist <= TH_LOW) {
            vpMatched[bestIdx] = 1;
            match_kf[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1476-1603; what Tracking::Relocalization runs with (th, ORBdist) = (10, 100) and (3, 64), Tracking.cc:1858,
 * :1875) on flat arrays.  Per feature i of the keyframe: valid[i] = "pMP && !pMP->isBad() && !sAlreadyFound.count(pMP)" (:1494-1498),
 * p3Dw = GetWorldPos(), min_dist / max_dist = mfMinDistance / mfMaxDistance, mp_desc = GetDescriptor(), kf_angle[i] =
 * pKF->mvKeysUn[i].angle.  taken_cur[i2] = CurrentFrame.mvpMapPoints[i2] != NULL on entry; a keypoint that receives a point is
 * taken for the points after it (:1547-1548, :1561).  The projection has NO positive-depth gate here (:1503-1512), invzc is
 * `1.0 / z` evaluated in double, u = fx * xc * invzc + cx left to right, Frame bounds (<=).  Ow = -Rcw' * tcw is the caller's
 * (:1482).  match_cur[i2] = i (the keypoint received the map point of keyframe feature i) or -1; returns nmatches. */
int oracle_search_by_projection_keyframe(const void* kps_cur_, const uint8_t* desc_cur, int n_cur, const uint8_t* taken_cur, int cols,
                                         int rows, const float* bounds, int n_kf, const float* kf_angle, const uint8_t* valid,
                                         const float* p3Dw, const float* min_dist, const float* max_dist, const uint8_t* mp_desc,
                                         const float* Tcw, const float* Ow, const float* K4, const float* mvScaleFactors, int nlevels,
                                         float mfLogScaleFactor, float th, int ORBdist, int check_orientation, int32_t* match_cur)
{
    const KeyPoint* kc = (const KeyPoint*)kps_cur_;
    FrameGrid grid(kc, n_cur, cols, rows, bounds);
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<uint8_t> mvpMapPoints(n_cur, 0);
    for (int i = 0; i < n_cur; i++) { match_cur[i] = -1; mvpMapPoints[i] = taken_cur ? taken_cur[i] : 0; }
    for (int i = 0; i < n_kf; i++) {
        if (valid && !valid[i]) continue;
        const float X = p3Dw[3 * i], Y = p3Dw[3 * i + 1], Z = p3Dw[3 * i + 2];
        float t0 = Tcw[0] * X + Tcw[1] * Y + Tcw[2] * Z, t1 = Tcw[4] * X + Tcw[5] * Y + Tcw[6] * Z, t2 = Tcw[8] * X + Tcw[9] * Y + Tcw[10] * Z;
        const float xc = (float)(t0 * 1.0 + 1.0 * Tcw[3]);
        const float yc = (float)(t1 * 1.0 + 1.0 * Tcw[7]);
        const float zc = (float)(t2 * 1.0 + 1.0 * Tcw[11]);
        const float invzc = 1.0 / zc;
        const float u = fx * xc * invzc + cx;
        const float v = fy * yc * invzc + cy;
        if (u < grid.mnMinX || u > grid.mnMaxX) continue;
        if (v < grid.mnMinY || v > grid.mnMaxY) continue;
        if (u != u || v != v) continue; /* NaN: undefined in the reference (grid indices from NaN); dropped */
        const float PO[3] = {X - Ow[0], Y - Ow[1], Z - Ow[2]};
        float dist3D = std::sqrt((double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2]); /* cv::norm */
        const float maxDistance = GetMaxDistanceInvariance(max_dist[i]);
        const float minDistance = GetMinDistanceInvariance(min_dist[i]);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        int nPredictedLevel = PredictScale(max_dist[i], dist3D, mfLogScaleFactor, nlevels);
        const float radius = th * mvScaleFactors[nPredictedLevel];
        const std::vector<int> vIndices2 = grid.GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t k = 0; k < vIndices2.size(); k++) {
            const int i2 = vIndices2[k];
            if (mvpMapPoints[i2]) continue;
            const int dist = DescriptorDistance(dMP, desc_cur + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= ORBdist) {
            mvpMapPoints[bestIdx2] = 1;
            match_cur[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = kf_angle[i] - kc[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        ComputeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) {
                    match_cur[rotHist[i][j]] = -1; /* CurrentFrame.mvpMapPoints[...] = NULL */
                    nmatches--;
                }
    }
    return nmatches;
}

/* MapPoint::PredictScale alone, for the known-answer tests */
int oracle_predict_scale(float mfMaxDistance, float currentDist, float mfLogScaleFactor, int mnScaleLevels)
{
    return PredictScale(mfMaxDistance, currentDist, mfLogScaleFactor, mnScaleLevels);
}

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:270-333) on flat arrays: point p owns descriptors offsets[p] ..
 * offsets[p + 1] (the observations of non-bad keyframes, in the order of the std::map mObservations).  best_idx[p] = index inside
 * the point's list of the descriptor with the least median distance to the others, -1 for a point without descriptors (the
 * reference returns before touching mDescriptor). */
void oracle_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx)
{
    for (int p = 0; p < npoints; p++) {
        const uint8_t* vDescriptors = desc + 32 * (size_t)offsets[p];
        const size_t N = (size_t)(offsets[p + 1] - offsets[p]);
        if (N == 0) { best_idx[p] = -1; continue; }
        std::vector<std::vector<float>> Distances(N, std::vector<float>(N));
        for (size_t i = 0; i < N; i++) {
            Distances[i][i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                int distij = DescriptorDistance(vDescriptors + 32 * i, vDescriptors + 32 * j);
                Distances[i][j] = distij;
                Distances[j][i] = distij;
            }
        }
        int BestMedian = INT_MAX;
        int BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
            std::sort(vDists.begin(), vDists.end());
            int median = vDists[0.5 * (N - 1)];
            if (median < BestMedian) { BestMedian = median; BestIdx = i; }
        }
        best_idx[p] = BestIdx;
    }
}

void oracle_three_maxima(const int* sizes, int L, int* out3)
{
    int a = -1, b = -1, c = -1;
    ComputeThreeMaxima(sizes, L, a, b, c);
    out3[0] = a; out3[1] = b; out3[2] = c;
}

int oracle_match_constants(int which) { return which == 0 ? TH_HIGH : which == 1 ? TH_LOW : HISTO_LENGTH; }

} /* extern "C" */
