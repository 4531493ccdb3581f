/*
 * orb_oracle.cpp -- CPU restatement of the reference ORB extractor.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP extractor.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (orb_slam2_aruco_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED: the reference (ORB_SLAM2_aruco) ships no tests or golden
 * vectors and its image primitives live in OpenCV 3.4 (not vendored, not
 * installable here), so this restatement follows src/ORBextractor.cc line by
 * line and the OpenCV 3.4 generic (non-IPP) code paths as documented in
 * SURVEY.md App. B.  Each function cites the reference lines it restates.
 *
 * Plain single-threaded C++ (no OpenCV, no SIMD intrinsics); compile with
 * -O3 -ffp-contract=off (see oracle/Makefile).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "../include/orbfe_math.h"
#include "../orb_slam2_aruco_amd/csrc/orbfe_tables.inc" /* bit_pattern_31_ (data) */

namespace {

struct KeyPoint { /* cv::KeyPoint field order, 28 bytes */
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
};

const int PATCH_SIZE = 31;      /* ORBextractor.cc:72 */
const int HALF_PATCH_SIZE = 15; /* :73 */
const int EDGE_THRESHOLD = 19;  /* :74 */

/* ---------------------------------------------------------------- resize -- */
/* cv::resize(..., INTER_LINEAR) on CV_8UC1, OpenCV 3.4 generic path
 * (resizeGeneric_ + HResizeLinear<uchar,int,short,2048> + VResizeLinear fixed
 * point); called at ORBextractor.cc:1120.  SURVEY App. B.2. */
void resize_linear_u8(const Image& src, Image& dst, int dw, int dh)
{
    dst = Image(dw, dh);
    const int sw = src.w, sh = src.h;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = orbfe_floor_d(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[dx * 2] = (short)orbfe_round_f(c0 * 2048.f);
        ialpha[dx * 2 + 1] = (short)orbfe_round_f(c1 * 2048.f);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = orbfe_floor_d(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[dy * 2] = (short)orbfe_round_f(c0 * 2048.f);
        ibeta[dy * 2 + 1] = (short)orbfe_round_f(c1 * 2048.f);
    }
    std::vector<int> r0(dw), r1(dw);
    auto hresize = [&](int sy, std::vector<int>& out) {
        sy = std::min(std::max(sy, 0), sh - 1); /* clip() in resizeGeneric_Invoker */
        const uint8_t* S = src.row(sy);
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            out[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) out[dx] = S[xofs[dx]] * 2048;
    };
    for (int dy = 0; dy < dh; dy++) {
        hresize(yofs[dy], r0);
        hresize(yofs[dy] + 1, r1);
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst.row(dy);
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

/* ------------------------------------------------------------------ blur -- */
inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on 8U (ORBextractor.cc:1086); out = sat_u8((sum + 2^15) >> 16).  App. B.3.
 * The 8-bit taps depend on the OpenCV release -- a stated, switchable choice (the reference binary's 3.4 patch level is unknown):
 *   mode 0 (default): every tap rounded on its own, round(256 * normalised exp(-x^2/8)) = 18 34 49 55 49 34 18 (sum 257, not
 *           renormalised): the 8U path of sepFilter2D in 2.4 / 3.2 (CMakeLists.txt:32-38 asks for 3.2.0) and the first fixed-point
 *           GaussianBlur of 3.4 (ufixedpoint16 taps, each rounded to nearest);
 *   mode 1: the later "bit-exact" kernel (late 3.4.x, 4.x): the taps go to 8.8 fixed point left to right with the rounding error
 *           carried into the next tap, mirrored, and the centre tap takes what is left of 256: 18 34 48 56 48 34 18 (sum 256). */
void gaussian7_taps(int taps[7], int mode)
{
    double k[7], sum = 0;
    for (int i = 0; i < 7; i++) {
        double x = i - 3;
        k[i] = std::exp(-0.5 * x * x / 4.0);
        sum += k[i];
    }
    if (mode == 0) {
        for (int i = 0; i < 7; i++) taps[i] = orbfe_round_d(k[i] / sum * 256.0);
        return;
    }
    double err = 0;
    int acc = 0;
    for (int i = 0; i < 3; i++) {
        const double adj = k[i] / sum * 256.0 + err;
        const int v = orbfe_round_d(adj);
        err = adj - v;
        taps[i] = taps[6 - i] = v;
        acc += v;
    }
    taps[3] = 256 - 2 * acc;
}

void gaussian_blur7(const Image& src, Image& dst, int mode = 0)
{
    int taps[7];
    gaussian7_taps(taps, mode);
    const int w = src.w, h = src.h;
    dst = Image(w, h);
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src.row(y);
        int* T = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            int s = 0;
            if (x >= 3 && x < w - 3) { /* interior: no border handling needed */
                for (int k = -3; k <= 3; k++) s += taps[k + 3] * S[x + k];
            } else {
                for (int k = -3; k <= 3; k++) s += taps[k + 3] * S[reflect101(x + k, w)];
            }
            T[x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst.row(y);
        const int* R[7];
        for (int k = -3; k <= 3; k++) R[k + 3] = tmp.data() + (size_t)reflect101(y + k, h) * w;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = 0; k < 7; k++) s += taps[k] * R[k][x];
            int v = (s + 32768) >> 16;
            D[x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

/* ------------------------------------------------------------------ FAST -- */
/* ring offsets of cv::FAST TYPE_9_16 (fast.cpp makeOffsets), App. B.1 */
const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* cornerScore<16>: largest threshold for which the pixel is still a 9-16
 * corner, 0-based "score = max(max_arc min(d), max_arc min(-d)) - 1".
 * Returns a value < t for non-corners at threshold t. */
int fast_score(const uint8_t* p, int stride)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 16; k++) d[k] = v - p[RING_DY[k] * stride + RING_DX[k]];
    for (int k = 16; k < 25; k++) d[k] = d[k - 16];
    int best = -255;
    for (int s = 0; s < 16; s++) {
        int mn = d[s], mx = d[s];
        for (int j = 1; j < 9; j++) {
            mn = std::min(mn, d[s + j]);
            mx = std::max(mx, d[s + j]);
        }
        best = std::max(best, std::max(mn, -mx));
    }
    return best - 1;
}

/* cv::FAST(roi, kps, threshold, nonmaxSuppression=true) on a w x h ROI.
 * Keypoints in raster order; response = score.  (ORBextractor.cc:809,814) */
void fast_detect(const uint8_t* img, int stride, int w, int h, int threshold,
                 std::vector<KeyPoint>& out)
{
    out.clear();
    if (w < 7 || h < 7) return;
    std::vector<int> score((size_t)w * h, 0);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            /* early out, like cv::FAST's table test: a 9-arc always covers two ADJACENT compass points of the ring, so
             * two adjacent ones must both be brighter / both darker than the centre by more than the threshold */
            const int v = p[0], lo = v - threshold, hi = v + threshold;
            const int r0 = p[3 * stride], r4 = p[3], r8 = p[-3 * stride], r12 = p[-3];
            const bool dk = (r0 < lo && (r4 < lo || r12 < lo)) || (r8 < lo && (r4 < lo || r12 < lo));
            const bool br = (r0 > hi && (r4 > hi || r12 > hi)) || (r8 > hi && (r4 > hi || r12 > hi));
            if (!dk && !br) continue;
            int s = fast_score(p, stride);
            score[(size_t)y * w + x] = s >= threshold ? s : 0;
        }
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = score[(size_t)y * w + x];
            if (s <= 0) continue; /* threshold >= 1 in every reference configuration */
            bool keep = true;
            for (int dy = -1; dy <= 1 && keep; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    if (score[(size_t)(y + dy) * w + (x + dx)] >= s) { keep = false; break; }
                }
            if (keep) out.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
        }
}

/* -------------------------------------------------------------- quadtree -- */
struct ExtractorNode { /* include/ORBextractor.h:38-53 */
    std::vector<KeyPoint> vKeys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    long seq = 0; /* creation order: stand-in for the heap address used as tie-break at :684 */

    /* ORBextractor.cc:481-537 */
    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) const
    {
        const int halfX = (int)std::ceil(static_cast<float>(URx - ULx) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(BRy - ULy) / 2);
        n1.ULx = ULx; n1.ULy = ULy;
        n1.URx = ULx + halfX; n1.URy = ULy;
        n1.BLx = ULx; n1.BLy = ULy + halfY;
        n1.BRx = ULx + halfX; n1.BRy = ULy + halfY;
        n2.ULx = n1.URx; n2.ULy = n1.URy;
        n2.URx = URx; n2.URy = URy;
        n2.BLx = n1.BRx; n2.BLy = n1.BRy;
        n2.BRx = URx; n2.BRy = ULy + halfY;
        n3.ULx = n1.BLx; n3.ULy = n1.BLy;
        n3.URx = n1.BRx; n3.URy = n1.BRy;
        n3.BLx = BLx; n3.BLy = BLy;
        n3.BRx = n1.BRx; n3.BRy = BLy;
        n4.ULx = n3.URx; n4.ULy = n3.URy;
        n4.URx = n2.BRx; n4.URy = n2.BRy;
        n4.BLx = n3.BRx; n4.BLy = n3.BRy;
        n4.BRx = BRx; n4.BRy = BRy;
        for (size_t i = 0; i < vKeys.size(); i++) {
            const KeyPoint& kp = vKeys[i];
            if (kp.x < n1.URx) {
                if (kp.y < n1.BRy) n1.vKeys.push_back(kp);
                else n3.vKeys.push_back(kp);
            } else if (kp.y < n1.BRy) n2.vKeys.push_back(kp);
            else n4.vKeys.push_back(kp);
        }
        if (n1.vKeys.size() == 1) n1.bNoMore = true;
        if (n2.vKeys.size() == 1) n2.bNoMore = true;
        if (n3.vKeys.size() == 1) n3.bNoMore = true;
        if (n4.vKeys.size() == 1) n4.bNoMore = true;
    }
};

/* ORBextractor.cc:539-763.  Tie-break note: the reference sorts
 * pair<int, ExtractorNode*> (:684), i.e. equal-population nodes are ordered by
 * heap address, which is allocator dependent.  The oracle DEFINES the order as
 * creation sequence (a later-created node compares greater), see DESIGN.md. */
std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX,
                                        int maxX, int minY, int maxY, int N)
{
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<ExtractorNode> lNodes;
    std::vector<ExtractorNode*> vpIniNodes(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; i++) {
        ExtractorNode ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KeyPoint& kp = vToDistributeKeys[i];
        vpIniNodes[(int)(kp.x / hX)]->vKeys.push_back(kp);
    }
    auto lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
        else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
        else lit++;
    }
    bool bFinish = false;
    typedef std::pair<int, ExtractorNode*> SP;
    auto cmp = [](const SP& a, const SP& b) {
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;
    };
    std::vector<SP> vSizeAndPointerToNode;
    auto add_children = [&](ExtractorNode* n[4], int* nToExpand) {
        for (int c = 0; c < 4; c++) {
            if (n[c]->vKeys.size() > 0) {
                n[c]->seq = seq++;
                lNodes.push_front(*n[c]);
                if (n[c]->vKeys.size() > 1) {
                    if (nToExpand) (*nToExpand)++;
                    vSizeAndPointerToNode.push_back(std::make_pair((int)n[c]->vKeys.size(), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        }
    };
    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) { lit++; continue; }
            ExtractorNode n1, n2, n3, n4;
            lit->DivideNode(n1, n2, n3, n4);
            ExtractorNode* ch[4] = {&n1, &n2, &n3, &n4};
            add_children(ch, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SP> vPrev = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrev.begin(), vPrev.end(), cmp);
                for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
                    ExtractorNode n1, n2, n3, n4;
                    vPrev[j].second->DivideNode(n1, n2, n3, n4);
                    ExtractorNode* ch[4] = {&n1, &n2, &n3, &n4};
                    add_children(ch, nullptr);
                    lNodes.erase(vPrev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    std::vector<KeyPoint> vResultKeys;
    for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
        const std::vector<KeyPoint>& vNodeKeys = it->vKeys;
        const KeyPoint* pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++)
            if (vNodeKeys[k].response > maxResponse) {
                pKP = &vNodeKeys[k];
                maxResponse = vNodeKeys[k].response;
            }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

/* ------------------------------------------------------------- extractor -- */
struct Extractor {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor; /* double member initialised from a float (ORBextractor.h:98) */
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<int> mnFeaturesPerLevel, umax;
    int gaussian_mode = 0; /* gaussian7_taps: 0 = taps rounded one by one (sum 257), 1 = bit-exact kernel (sum 256) */
    int trig_libm = 0; /* diagnostic: use host libm cosf/sinf like the reference binary would */
    /* kept for per-stage parity tests */
    std::vector<Image> pyramid, blurred;
    std::vector<std::vector<KeyPoint>> toDistribute, distributed;

    /* ORBextractor.cc:410-470 */
    Extractor(int nf, float sf, int nl, int ini, int mn)
        : nfeatures(nf), nlevels(nl), iniThFAST(ini), minThFAST(mn), scaleFactor(sf)
    {
        mvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f;
        mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels);
        mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesiredFeaturesPerScale =
            (float)(nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels)));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = orbfe_round_f(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = orbfe_floor_d(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = orbfe_ceil_d(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = orbfe_round_d(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    /* ORBextractor.cc:1107-1132 (the 19-px border is never read by this path, App. A.2) */
    void ComputePyramid(const Image& image)
    {
        pyramid.resize(nlevels);
        for (int level = 0; level < nlevels; ++level) {
            float scale = mvInvScaleFactor[level];
            int sw = orbfe_round_f((float)image.w * scale), sh = orbfe_round_f((float)image.h * scale);
            if (level != 0) resize_linear_u8(pyramid[level - 1], pyramid[level], sw, sh);
            else pyramid[0] = image;
        }
    }

    /* ORBextractor.cc:77-104 */
    float IC_Angle(const Image& image, float ptx, float pty) const
    {
        int m_01 = 0, m_10 = 0;
        const int step = image.w;
        const uint8_t* center = image.d.data() + (size_t)orbfe_round_f(pty) * step + orbfe_round_f(ptx);
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return orbfe_fast_atan2((float)m_01, (float)m_10);
    }

    /* ORBextractor.cc:765-853 */
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints)
    {
        allKeypoints.assign(nlevels, std::vector<KeyPoint>());
        toDistribute.assign(nlevels, std::vector<KeyPoint>());
        const float W = 30;
        for (int level = 0; level < nlevels; ++level) {
            const Image& im = pyramid[level];
            const int minBorderX = EDGE_THRESHOLD - 3;
            const int minBorderY = minBorderX;
            const int maxBorderX = im.w - EDGE_THRESHOLD + 3;
            const int maxBorderY = im.h - EDGE_THRESHOLD + 3;
            std::vector<KeyPoint>& vToDistributeKeys = toDistribute[level];
            const float width = (float)(maxBorderX - minBorderX);
            const float height = (float)(maxBorderY - minBorderY);
            const int nCols = (int)(width / W);
            const int nRows = (int)(height / W);
            if (nCols <= 0 || nRows <= 0) continue; /* reference would divide by zero; image too small */
            const int wCell = (int)std::ceil(width / nCols);
            const int hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    std::vector<KeyPoint> vKeysCell;
                    const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
                    fast_detect(im.row(y0) + x0, im.w, cw, ch, iniThFAST, vKeysCell);
                    if (vKeysCell.empty()) fast_detect(im.row(y0) + x0, im.w, cw, ch, minThFAST, vKeysCell);
                    for (auto& kp : vKeysCell) {
                        kp.x += j * wCell;
                        kp.y += i * hCell;
                        vToDistributeKeys.push_back(kp);
                    }
                }
            }
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            if (!vToDistributeKeys.empty()) /* no candidates -> every root node is erased (:582) -> empty result */
                keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                              mnFeaturesPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (auto& kp : keypoints) {
                kp.x += minBorderX;
                kp.y += minBorderY;
                kp.octave = level;
                kp.size = (float)scaledPatchSize;
            }
        }
        distributed = allKeypoints;
        for (int level = 0; level < nlevels; ++level)
            for (auto& kp : allKeypoints[level]) kp.angle = IC_Angle(pyramid[level], kp.x, kp.y);
    }

    /* ORBextractor.cc:108-147 */
    void computeOrbDescriptor(const KeyPoint& kpt, const Image& img, uint8_t* desc) const
    {
        const float factorPI = (float)(3.14159265358979323846 / 180.f);
        float angle = (float)kpt.angle * factorPI;
        float a, b;
        if (trig_libm) { a = (float)cosf(angle); b = (float)sinf(angle); }
        else orbfe_sincosf(angle, &b, &a);
        const int step = img.w;
        const uint8_t* center = img.d.data() + (size_t)orbfe_round_f(kpt.y) * step + orbfe_round_f(kpt.x);
        const signed char* pattern = ORBFE_BIT_PATTERN_31;
        auto get = [&](int idx) -> int {
            float px = (float)pattern[2 * idx], py = (float)pattern[2 * idx + 1];
            return center[orbfe_round_f(px * b + py * a) * step + orbfe_round_f(px * a - py * b)];
        };
        for (int i = 0; i < 32; ++i, pattern += 32) {
            int val = 0;
            for (int k = 0; k < 8; k++) {
                int t0 = get(2 * k), t1 = get(2 * k + 1);
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    /* ORBextractor.cc:1043-1105.  Returns number of keypoints. */
    int extract(const uint8_t* img, int rows, int cols, size_t step, std::vector<KeyPoint>& kps,
                std::vector<uint8_t>& desc)
    {
        kps.clear();
        desc.clear();
        if (!img || rows <= 0 || cols <= 0) return 0;
        Image image(cols, rows);
        for (int y = 0; y < rows; y++) memcpy(image.row(y), img + (size_t)y * step, cols);
        ComputePyramid(image);
        std::vector<std::vector<KeyPoint>> allKeypoints;
        ComputeKeyPointsOctTree(allKeypoints);
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
        desc.assign((size_t)nkeypoints * 32, 0);
        blurred.assign(nlevels, Image());
        int offset = 0;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            int n = (int)keypoints.size();
            if (n == 0) continue;
            gaussian_blur7(pyramid[level], blurred[level], gaussian_mode);
            for (int i = 0; i < n; i++)
                computeOrbDescriptor(keypoints[i], blurred[level], desc.data() + (size_t)(offset + i) * 32);
            offset += n;
            if (level != 0) {
                float scale = mvScaleFactor[level];
                for (auto& kp : keypoints) { kp.x *= scale; kp.y *= scale; }
            }
            kps.insert(kps.end(), keypoints.begin(), keypoints.end());
        }
        return nkeypoints;
    }
};

} // namespace

/* ------------------------------------------------------------- C exports -- */
extern "C" {

void* oracle_orb_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
{
    return new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST);
}
void oracle_orb_destroy(void* h) { delete (Extractor*)h; }
void oracle_orb_set_trig_libm(void* h, int on) { ((Extractor*)h)->trig_libm = on; }
void oracle_orb_set_gaussian_taps(void* h, int mode) { ((Extractor*)h)->gaussian_mode = mode; }

/* kps: capacity x 28 B (cv::KeyPoint layout), desc: capacity x 32 B. Returns n (or -n if capacity too small). */
int oracle_orb_extract(void* h, const uint8_t* img, int rows, int cols, size_t step, void* kps, uint8_t* desc,
                       int capacity)
{
    Extractor* e = (Extractor*)h;
    std::vector<KeyPoint> k;
    std::vector<uint8_t> d;
    int n = e->extract(img, rows, cols, step, k, d);
    if (n > capacity) return -n;
    if (n) {
        memcpy(kps, k.data(), (size_t)n * sizeof(KeyPoint));
        memcpy(desc, d.data(), (size_t)n * 32);
    }
    return n;
}

void oracle_orb_tables(void* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int* per_level,
                       int* umax16)
{
    Extractor* e = (Extractor*)h;
    for (int i = 0; i < e->nlevels; i++) {
        scale[i] = e->mvScaleFactor[i];
        inv_scale[i] = e->mvInvScaleFactor[i];
        sigma2[i] = e->mvLevelSigma2[i];
        inv_sigma2[i] = e->mvInvLevelSigma2[i];
        per_level[i] = e->mnFeaturesPerLevel[i];
    }
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

/* stage accessors (valid after oracle_orb_extract) */
void oracle_orb_level_size(void* h, int level, int* w, int* hh)
{
    Extractor* e = (Extractor*)h;
    *w = e->pyramid[level].w;
    *hh = e->pyramid[level].h;
}
void oracle_orb_level_image(void* h, int level, int blurred, uint8_t* out)
{
    Extractor* e = (Extractor*)h;
    const Image& im = blurred ? e->blurred[level] : e->pyramid[level];
    if (!im.d.empty()) memcpy(out, im.d.data(), im.d.size());
}
/* stage 0: candidates handed to DistributeOctTree (coords relative to minBorder); stage 1: after it (level coords) */
int oracle_orb_level_keypoints(void* h, int level, int stage, void* out, int capacity)
{
    Extractor* e = (Extractor*)h;
    const std::vector<KeyPoint>& v = stage == 0 ? e->toDistribute[level] : e->distributed[level];
    int n = (int)v.size();
    if (out && n <= capacity && n) memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return n;
}

/* primitives, exported for known-answer tests */
void oracle_resize_linear_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    Image s(sw, sh), d;
    memcpy(s.d.data(), src, (size_t)sw * sh);
    resize_linear_u8(s, d, dw, dh);
    memcpy(dst, d.d.data(), (size_t)dw * dh);
}
void oracle_gaussian_blur7(const uint8_t* src, int w, int h, uint8_t* dst)
{
    Image s(w, h), d;
    memcpy(s.d.data(), src, (size_t)w * h);
    gaussian_blur7(s, d);
    memcpy(dst, d.d.data(), (size_t)w * h);
}
void oracle_gaussian7_taps(int* taps) { gaussian7_taps(taps, 0); }
void oracle_gaussian7_taps_mode(int* taps, int mode) { gaussian7_taps(taps, mode); }
/* score map of an image (0 in the 3-px frame); used to test the FAST kernel in isolation */
void oracle_fast_score_map(const uint8_t* img, int w, int h, int* out)
{
    for (int i = 0; i < w * h; i++) out[i] = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) out[y * w + x] = fast_score(img + (size_t)y * w + x, w);
}
int oracle_fast_detect(const uint8_t* img, int w, int h, int threshold, void* out, int capacity)
{
    std::vector<KeyPoint> v;
    fast_detect(img, w, w, h, threshold, v);
    int n = (int)v.size();
    if (n <= capacity && n) memcpy(out, v.data(), (size_t)n * sizeof(KeyPoint));
    return n;
}
int oracle_distribute(const void* kps, int n, int minX, int maxX, int minY, int maxY, int N, void* out, int capacity)
{
    std::vector<KeyPoint> in((const KeyPoint*)kps, (const KeyPoint*)kps + n);
    std::vector<KeyPoint> r = DistributeOctTree(in, minX, maxX, minY, maxY, N);
    int m = (int)r.size();
    if (m <= capacity && m) memcpy(out, r.data(), (size_t)m * sizeof(KeyPoint));
    return m;
}
float oracle_fast_atan2(float y, float x) { return orbfe_fast_atan2(y, x); }
void oracle_sincosf(float x, float* s, float* c) { orbfe_sincosf(x, s, c); }
int oracle_cv_round(double v) { return orbfe_round_d(v); }

} /* extern "C" */
