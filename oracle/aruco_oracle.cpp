/*
 * aruco_oracle.cpp -- CPU restatement of the reference ArUco detector as configured by
 * src/Frame.cc:129-142 (dictionary by name, DM_NORMAL, CORNER_LINES), and of the rest of its parameter
 * surface (markerdetector.cpp:364-402: DM_FAST / DM_VIDEO_FAST = THRES_AUTO_FIXED with rand() retries and
 * frame-to-frame state, Params::minSize > 0 with cornerUpsample, CORNER_SUBPIX, CV_8UC3 input).
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED for the OpenCV primitives (adaptiveThreshold, findContours, approxPolyDP,
 * getPerspectiveTransform/warpPerspective, threshold(OTSU), solve(SVD), threshold, resize(INTER_NEAREST),
 * cornerSubPix / getRectSubPix, cvtColor(BGR2GRAY)): the reference ships no
 * tests and OpenCV 3.4 is not available here, so these follow the OpenCV 3.4 generic code paths
 * from knowledge of that source.  The detector logic itself follows the
 * de-obfuscated Thirdparty/aruco/aruco/markerdetector_impl.cpp and dictionary_based.cpp
 * (recipe in SURVEY.md App. C); cites use the line numbers of the shipped (obfuscated) files.
 *
 * Deliberate, documented deviations (also in DESIGN.md):
 *  - getPerspectiveTransform / the line fits use double-precision Gaussian elimination / normal
 *    equations where OpenCV 3.4 calls solve(DECOMP_SVD); results agree to ~1e-6 relative.
 *  - pose (Marker::calculateExtrinsics, step 12 of App. C) is out of scope.
 */
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/orbfe_math.h"
#include "../orb_slam2_aruco_amd/csrc/orbfe_tables.inc"

namespace {

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t* row(int y) { return d.data() + (size_t)y * w; }
    const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
};
struct Pt { int x, y; };
struct Ptf { float x, y; };

struct Candidate {
    Ptf c[4];
    std::vector<Pt> contour;
    int id = -1;
};

/* ------------------------------------------------------- adaptiveThreshold -- */
/* cv::adaptiveThreshold(src, dst, 255, ADAPTIVE_THRESH_MEAN_C, THRESH_BINARY_INV, win, C)
 * (markerdetector_impl.cpp:2983): normalized box mean with BORDER_REPLICATE, mean rounded to
 * nearest, dst = 255 where src - mean <= -floor(C). */
void adaptive_threshold_inv(const Image& src, Image& dst, int win, int C)
{
    const int w = src.w, h = src.h, r = win / 2;
    dst = Image(w, h);
    const double scale = 1.0 / (win * win);
    std::vector<int> colsum(w);
    std::vector<int> rowsum((size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src.row(y);
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -r; k <= r; k++) s += S[std::min(std::max(x + k, 0), w - 1)];
            rowsum[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src.row(y);
        uint8_t* D = dst.row(y);
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int k = -r; k <= r; k++) s += rowsum[(size_t)std::min(std::max(y + k, 0), h - 1) * w + x];
            int mean = orbfe_round_d(s * scale);
            if (mean > 255) mean = 255;
            D[x] = (S[x] - mean <= -C) ? 255 : 0;
        }
    }
}

/* ---------------------------------------------------------- findContours -- */
/* cv::findContours(img, contours, RETR_LIST, CHAIN_APPROX_NONE) (markerdetector_impl.cpp:3104),
 * OpenCV 3.4: the image is copied into a zero-padded buffer (offset -1,-1), binarised to 0/1,
 * scanned in raster order (cvFindNextContour) and every border is followed by icvFetchContour
 * with nbd = 2.  The output order is the REVERSE of discovery (cvInsertNodeIntoTree prepends). */
void find_contours_list(const Image& img, std::vector<std::vector<Pt>>& out)
{
    const int W = img.w + 2, H = img.h + 2;
    std::vector<signed char> buf((size_t)W * H, 0);
    for (int y = 0; y < img.h; y++)
        for (int x = 0; x < img.w; x++) buf[(size_t)(y + 1) * W + (x + 1)] = img.row(y)[x] ? 1 : 0;
    const int step = W;
    const int deltas[16] = {1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1,
                            1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1};
    static const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, cdy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    std::vector<std::vector<Pt>> found;
    for (int y = 1; y < H - 1; y++) {
        signed char* row = buf.data() + (size_t)y * W;
        int prev = 0;
        for (int x = 1; x < W - 1; x++) {
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            bool trace = false;
            if (prev == 0 && p == 1) trace = true;
            else if (p == 0 && prev >= 1) { is_hole = 1; trace = true; }
            if (trace) {
                std::vector<Pt> c;
                signed char* i0 = row + x - is_hole;
                Pt pt{x - is_hole - 1, y - 1}; /* offset (-1,-1): back to image coordinates */
                const signed char nbd = 2;
                int s_end, s;
                s_end = s = is_hole ? 0 : 4;
                signed char* i1;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + deltas[s];
                } while (*i1 == 0 && s != s_end);
                if (s == s_end) { /* single pixel domain */
                    *i0 = (signed char)(nbd | -128);
                    c.push_back(pt);
                } else {
                    signed char* i3 = i0;
                    signed char* i4 = nullptr;
                    for (;;) {
                        s_end = s;
                        while (s < 15) {
                            i4 = i3 + deltas[++s];
                            if (*i4 != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
                        else if (*i3 == 1) *i3 = nbd;
                        c.push_back(pt);
                        pt.x += cdx[s];
                        pt.y += cdy[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                found.push_back(std::move(c));
                p = row[x];
            }
            prev = p;
        }
    }
    out.assign(found.rbegin(), found.rend());
}

/* ---------------------------------------------------------- approxPolyDP -- */
/* cv::approxPolyDP(contour, out, eps, closed=true) for integer points (approx.cpp approxPolyDP_<int>,
 * OpenCV 3.4.2+ incl. the successive-inner-product guard in the clean-up pass). */
int approx_poly_dp_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dst)
{
    const int count = (int)src.size();
    dst.clear();
    if (count == 0) return 0;
    struct Range { int start, end; };
    std::vector<Range> stack;
    std::vector<Pt> out;
    Range slice{0, 0}, right_slice{0, 0};
    Pt start_pt{-1000000, -1000000}, end_pt{0, 0}, pt{0, 0};
    int pos = 0;
    bool le_eps = false;
    eps *= eps;
    auto read_pt = [&](Pt& p, int& ps) { p = src[ps]; if (++ps >= count) ps = 0; };
    right_slice.start = 0;
    for (int i = 0; i < 3; i++) {
        double max_dist = 0;
        pos = (pos + right_slice.start) % count;
        read_pt(start_pt, pos);
        for (int j = 1; j < count; j++) {
            read_pt(pt, pos);
            double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            double dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_slice.start = j; }
        }
        le_eps = max_dist <= eps;
    }
    if (!le_eps) {
        right_slice.end = slice.start = pos % count;
        slice.end = right_slice.start = (right_slice.start + slice.start) % count;
        stack.push_back(right_slice);
        stack.push_back(slice);
    } else out.push_back(start_pt);
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end_pt = src[slice.end];
        pos = slice.start;
        read_pt(start_pt, pos);
        if (pos != slice.end) {
            double max_dist = 0;
            double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            while (pos != slice.end) {
                read_pt(pt, pos);
                double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; right_slice.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = src[slice.start];
        }
        if (le_eps) out.push_back(start_pt);
        else {
            right_slice.end = slice.end;
            slice.end = right_slice.start;
            stack.push_back(right_slice);
            stack.push_back(slice);
        }
    }
    /* clean-up pass */
    int new_count = (int)out.size();
    const int cnt = new_count;
    auto read_dst = [&](Pt& p, int& ps) { p = out[ps]; if (++ps >= cnt) ps = 0; };
    pos = cnt - 1;
    read_dst(start_pt, pos);
    int wpos = pos;
    read_dst(pt, pos);
    for (int i = 0; i < cnt && new_count > 2; i++) {
        read_dst(end_pt, pos);
        double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        double sip = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            new_count--;
            out[wpos] = start_pt = end_pt;
            if (++wpos >= cnt) wpos = 0;
            read_dst(pt, pos);
            i++;
            continue;
        }
        out[wpos] = start_pt = pt;
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    dst.assign(out.begin(), out.begin() + new_count);
    return new_count;
}

/* cv::isContourConvex for integer points (convhull.cpp isContourConvex_<int>) */
bool is_contour_convex(const Pt* p, int n)
{
    Pt prev_pt = p[(n - 2 + n) % n], cur_pt = p[n - 1];
    int dx0 = cur_pt.x - prev_pt.x, dy0 = cur_pt.y - prev_pt.y, orientation = 0;
    for (int i = 0; i < n; i++) {
        prev_pt = cur_pt;
        cur_pt = p[i];
        int dx = cur_pt.x - prev_pt.x, dy = cur_pt.y - prev_pt.y;
        int dxdy0 = dx * dy0, dydx0 = dy * dx0;
        orientation |= (dydx0 > dxdy0) ? 1 : ((dydx0 < dxdy0) ? 2 : 3);
        if (orientation == 3) return false;
        dx0 = dx;
        dy0 = dy;
    }
    return true;
}

/* -------------------------------------------------------- warpPerspective -- */
/* getPerspectiveTransform(src quad -> (0,0),(S-1,0),(S-1,S-1),(0,S-1)) followed by
 * warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) to S x S (markerdetector_impl.cpp:10844-11111).
 * The 8x8 system is solved by Gaussian elimination with partial pivoting in double (OpenCV 3.4:
 * solve(DECOMP_SVD)); the map is inverted with the closed-form 3x3 inverse; coordinates are rounded
 * to 1/32 px and blended with the 15-bit bilinear table exactly like remapBilinear. */
bool solve_linear(double* A, double* b, int n) /* in place, row-major n x n; returns false if singular */
{
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
        if (A[piv * n + c] == 0.0) return false;
        if (piv != c) {
            for (int k = 0; k < n; k++) std::swap(A[c * n + k], A[piv * n + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] / A[c * n + c];
            if (f == 0.0) continue;
            for (int k = c; k < n; k++) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < n; k++) s -= A[r * n + k] * b[k];
        b[r] = s / A[r * n + r];
    }
    return true;
}

bool perspective_inverse_map(const Ptf q[4], int S, double Minv[9])
{
    const float dstx[4] = {0.f, (float)(S - 1), (float)(S - 1), 0.f};
    const float dsty[4] = {0.f, 0.f, (float)(S - 1), (float)(S - 1)};
    double a[64], b[8];
    memset(a, 0, sizeof(a));
    for (int i = 0; i < 4; i++) {
        a[i * 8 + 0] = a[(i + 4) * 8 + 3] = q[i].x;
        a[i * 8 + 1] = a[(i + 4) * 8 + 4] = q[i].y;
        a[i * 8 + 2] = a[(i + 4) * 8 + 5] = 1;
        a[i * 8 + 6] = -(double)q[i].x * dstx[i];
        a[i * 8 + 7] = -(double)q[i].y * dstx[i];
        a[(i + 4) * 8 + 6] = -(double)q[i].x * dsty[i];
        a[(i + 4) * 8 + 7] = -(double)q[i].y * dsty[i];
        b[i] = dstx[i];
        b[i + 4] = dsty[i];
    }
    if (!solve_linear(a, b, 8)) return false;
    const double M[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
    /* 3x3 inverse (cv::invert on 3x3 uses the closed form with d = 1/det) */
    double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
                 M[2] * (M[3] * M[7] - M[4] * M[6]);
    if (det == 0.0) return false;
    double d = 1.0 / det;
    Minv[0] = (M[4] * M[8] - M[5] * M[7]) * d;
    Minv[1] = (M[2] * M[7] - M[1] * M[8]) * d;
    Minv[2] = (M[1] * M[5] - M[2] * M[4]) * d;
    Minv[3] = (M[5] * M[6] - M[3] * M[8]) * d;
    Minv[4] = (M[0] * M[8] - M[2] * M[6]) * d;
    Minv[5] = (M[2] * M[3] - M[0] * M[5]) * d;
    Minv[6] = (M[3] * M[7] - M[4] * M[6]) * d;
    Minv[7] = (M[1] * M[6] - M[0] * M[7]) * d;
    Minv[8] = (M[0] * M[4] - M[1] * M[3]) * d;
    return true;
}

inline int sat_int(double v)
{
    if (v <= (double)INT_MIN) return INT_MIN;
    if (v >= (double)INT_MAX) return INT_MAX;
    return orbfe_round_d(v);
}
inline int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

void warp_perspective(const Image& src, const double M[9], int S, uint8_t* dst)
{
    for (int y = 0; y < S; y++) {
        const double X0 = M[0] * 0 + M[1] * y + M[2];
        const double Y0 = M[3] * 0 + M[4] * y + M[5];
        const double W0 = M[6] * 0 + M[7] * y + M[8];
        for (int x = 0; x < S; x++) {
            double Wd = W0 + M[6] * x;
            Wd = Wd ? 32.0 / Wd : 0;
            double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, (X0 + M[0] * x) * Wd));
            double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, (Y0 + M[3] * x) * Wd));
            int X = sat_int(fX), Y = sat_int(fY);
            int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
            int ax = X & 31, ay = Y & 31;
            int w00 = (32 - ax) * (32 - ay) * 32, w01 = ax * (32 - ay) * 32, w10 = (32 - ax) * ay * 32, w11 = ax * ay * 32;
            auto px = [&](int xx, int yy) -> int {
                if (xx < 0 || yy < 0 || xx >= src.w || yy >= src.h) return 0;
                return src.row(yy)[xx];
            };
            int v = px(sx, sy) * w00 + px(sx + 1, sy) * w01 + px(sx, sy + 1) * w10 + px(sx + 1, sy + 1) * w11;
            v = (v + (1 << 14)) >> 15;
            dst[y * S + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
}

/* --------------------------------------------------------------- Otsu ---- */
/* cv::threshold(..., THRESH_BINARY | THRESH_OTSU) threshold value (thresh.cpp getThreshVal_Otsu_8u) */
int otsu_threshold(const uint8_t* p, int n)
{
    int h[256] = {0};
    for (int i = 0; i < n; i++) h[p[i]]++;
    double mu = 0, scale = 1. / n;
    for (int i = 0; i < 256; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        double q2 = 1. - q1;
        if (std::min(q1, q2) < FLT_EPSILON || std::max(q1, q2) > 1. - FLT_EPSILON) continue;
        mu1 = (mu1 + i * p_i) / q1;
        double mu2 = (mu - q1 * mu1) / q2;
        double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    return (int)max_val;
}


/* ---------------------------------------------------------------- primitives of the modes outside Frame.cc:135-137 ----
 * (DM_FAST / DM_VIDEO_FAST = THRES_AUTO_FIXED, Params::minSize > 0, CORNER_SUBPIX, CV_8UC3 input).  PARITY UNPINNED like the
 * rest of the image path: cv::threshold, cv::resize(INTER_NEAREST), cv::cornerSubPix / getRectSubPix and cvtColor(BGR2GRAY)
 * follow the OpenCV 3.x generic code paths from knowledge of that source. */

/* cv::threshold(src, dst, int(thr), 255, THRESH_BINARY_INV) (markerdetector_impl.cpp:2833-2870) */
void threshold_fixed_inv(const Image& src, Image& dst, int thr)
{
    dst = Image(src.w, src.h);
    for (size_t i = 0; i < src.d.size(); i++) dst.d[i] = src.d[i] > thr ? 0 : 255;
}

/* Params::detectEnclosedMarkers with THRES_AUTO_FIXED (markerdetector_impl.cpp:2871-2950): erode with a MORPH_CROSS element of
 * size k (anchor at its centre; pixels outside the image do not constrain the minimum: morphologyDefaultBorderValue), then
 * bitwise_xor with the thresholded image -- the inner edge band of every thresholded region */
void erode_cross_xor(Image& thres, int k)
{
    const int r = k / 2, w = thres.w, h = thres.h;
    Image er(w, h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint8_t m = 255;
            for (int d = -r; d <= r; d++) {
                if (x + d >= 0 && x + d < w) m = std::min(m, thres.row(y)[x + d]);
                if (y + d >= 0 && y + d < h) m = std::min(m, thres.row(y + d)[x]);
            }
            er.row(y)[x] = m;
        }
    for (size_t i = 0; i < thres.d.size(); i++) thres.d[i] ^= er.d[i];
}

/* cv::resize(src, dst, dsize, 0, 0, INTER_NEAREST) (resize.cpp resizeNN): sx = min(floor(x * ifx), sw - 1), ifx = 1 / (dw / sw) */
void resize_nearest(const Image& src, Image& dst)
{
    const double inv_fx = (double)dst.w / src.w, inv_fy = (double)dst.h / src.h;
    const double ifx = 1. / inv_fx, ify = 1. / inv_fy;
    std::vector<int> xo(dst.w);
    for (int x = 0; x < dst.w; x++) xo[x] = std::min(orbfe_floor_d(x * ifx), src.w - 1);
    for (int y = 0; y < dst.h; y++) {
        const uint8_t* S = src.row(std::min(orbfe_floor_d(y * ify), src.h - 1));
        uint8_t* D = dst.row(y);
        for (int x = 0; x < dst.w; x++) D[x] = S[xo[x]];
    }
}

/* cvtColor(BGR2GRAY) on CV_8UC3 (markerdetector_impl.cpp:5892): OpenCV <= 3.4.1 tables with 14 fractional bits
 * (B 1868, G 9617, R 4899); 3.4.2+ / 4.x use 15 bits (B 3735, G 19235, R 9798): `bits15` selects. */
void bgr_to_gray(const uint8_t* bgr, size_t step, int w, int h, Image& dst, int bits15)
{
    dst = Image(w, h);
    for (int y = 0; y < h; y++) {
        const uint8_t* s = bgr + (size_t)y * step;
        for (int x = 0; x < w; x++, s += 3)
            dst.row(y)[x] = bits15 ? (uint8_t)((s[0] * 3735 + s[1] * 19235 + s[2] * 9798 + (1 << 14)) >> 15)
                                   : (uint8_t)((s[0] * 1868 + s[1] * 9617 + s[2] * 4899 + (1 << 13)) >> 14);
    }
}

/* getRectSubPix(src 8u, Size(ww, wh), center, dst 32f) (samplers.cpp getRectSubPix_Cn_<uchar, float, float>), dst row-major ww x wh */
void get_rect_subpix(const Image& src, int ww, int wh, float cx, float cy, float* dst)
{
    cx -= (ww - 1) * 0.5f;
    cy -= (wh - 1) * 0.5f;
    const int ipx = orbfe_floor_d(cx), ipy = orbfe_floor_d(cy);
    const float a = cx - ipx, b = cy - ipy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b, b1 = 1.f - b, b2 = b;
    if (0 <= ipx && ipx < src.w - ww && 0 <= ipy && ipy < src.h - wh) {
        for (int i = 0; i < wh; i++) {
            const uint8_t* s = src.row(ipy + i) + ipx;
            const uint8_t* s2 = s + src.w;
            for (int j = 0; j < ww; j++) dst[i * ww + j] = s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22;
        }
        return;
    }
    /* adjustRect(): the part of the window inside the image is [rx, rw) x [ry, rh); outside it the border is replicated */
    long so = 0; /* offset of `src` (in pixels) from the image origin */
    int rx, rw, ry, rh;
    if (ipx >= 0) { so += ipx; rx = 0; }
    else { rx = -ipx; if (rx > ww) rx = ww; }
    if (ipx < src.w - ww) rw = ww;
    else { rw = src.w - ipx - 1; if (rw < 0) { so += rw; rw = 0; } }
    if (ipy >= 0) { so += (long)ipy * src.w; ry = 0; }
    else ry = -ipy;
    if (ipy < src.h - wh) rh = wh;
    else { rh = src.h - ipy - 1; if (rh < 0) { so += (long)rh * src.w; rh = 0; } }
    const uint8_t* s = src.d.data() + so - rx;
    for (int i = 0; i < wh; i++) {
        const uint8_t* s2 = s + src.w;
        if (i < ry || i >= rh) s2 -= src.w;
        int j = 0;
        for (; j < rx; j++) dst[i * ww + j] = s[rx] * b1 + s2[rx] * b2;
        for (; j < rw; j++) dst[i * ww + j] = s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22;
        for (; j < ww; j++) dst[i * ww + j] = s[rw] * b1 + s2[rw] * b2;
        if (i < rh) s = s2;
    }
}

/* cv::cornerSubPix(src, corners, Size(win, win), Size(-1, -1), criteria) (cornersubpix.cpp): max_iters = criteria.maxCount clamped to
 * [1, 100] (100 without MAX_ITER), eps = criteria.epsilon (0 without EPS) */
void corner_subpix(const Image& src, std::vector<Ptf>& corners, int win, int max_iters, double eps)
{
    if (corners.empty()) return;
    const int ww = 2 * win + 1;
    eps *= eps;
    std::vector<float> mask((size_t)ww * ww), buf((size_t)(ww + 2) * (ww + 2));
    for (int i = 0; i < ww; i++) {
        const float y = (float)(i - win) / win;
        const float vy = std::exp(-y * y);
        for (int j = 0; j < ww; j++) {
            const float x = (float)(j - win) / win;
            mask[(size_t)i * ww + j] = (float)(vy * std::exp(-x * x));
        }
    }
    for (auto& cr : corners) {
        const Ptf cT = cr;
        Ptf cI = cT;
        int iter = 0;
        double err = 0;
        do {
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            get_rect_subpix(src, ww + 2, ww + 2, cI.x, cI.y, buf.data());
            const float* sp = buf.data() + (ww + 2) + 1;
            for (int i = 0, k = 0; i < ww; i++, sp += ww + 2) {
                const double py = i - win;
                for (int j = 0; j < ww; j++, k++) {
                    const double m = mask[k];
                    const double tgx = sp[j + 1] - sp[j - 1];
                    const double tgy = sp[j + ww + 2] - sp[j - ww - 2];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - win;
                    a += gxx; b += gxy; c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            const double det = a * c - b * b;
            if (std::fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            const double scale = 1.0 / det;
            Ptf cI2;
            cI2.x = (float)(cI.x + c * scale * bb1 - b * scale * bb2);
            cI2.y = (float)(cI.y - b * scale * bb1 + a * scale * bb2);
            err = (cI2.x - cI.x) * (cI2.x - cI.x) + (cI2.y - cI.y) * (cI2.y - cI.y);
            cI = cI2;
            if (cI.x < 0 || cI.x >= src.w || cI.y < 0 || cI.y >= src.h) break;
        } while (++iter < max_iters && err > eps);
        if (std::fabs(cI.x - cT.x) > win || std::fabs(cI.y - cT.y) > win) cI = cT;
        cr = cI;
    }
}

/* Otsu over the float histogram accumulated from the warped patches of the detected markers (markerdetector_impl.cpp:6121-6380):
 * normalises the histogram in place; -1 when no split has both classes above 1e-4 (an empty histogram gives NaNs and -1) */
int otsu_of_histogram(std::vector<float>& hist)
{
    float sum = 0, invsum;
    for (auto c : hist) sum += c;
    invsum = 1. / sum;
    for (auto& c : hist) c *= invsum;
    float maxVar = 0;
    int bestT = -1;
    for (int t = 1; t < 256; t++) {
        float w0 = 0, w1 = 0, mean0 = 0, mean1 = 0;
        for (int v = 0; v < t; v++) { w0 += hist[v]; mean0 += float(v) * hist[v]; }
        for (int v = t; v < 256; v++) { w1 += hist[v]; mean1 += hist[v] * float(v); }
        if (w0 > 1e-4 && w1 > 1e-4) {
            mean0 /= w0;
            mean1 /= w1;
            const float var = w0 * w1 * (mean0 - mean1) * (mean0 - mean1);
            if (var > maxVar) { maxVar = var; bestT = t; }
        }
    }
    return bestT;
}

/* ------------------------------------------------------------ dictionary -- */
struct Dict {
    std::string name;
    int nbits = 0, n = 0, tau = 0;
    const unsigned long long* codes = nullptr;
    float error_correction_rate = 0; /* MarkerDetector::Params::error_correction_rate (markerdetector.h:171); 0 in the reference's setup */
    /* Dictionary::fromVector (dictionary.cpp:99-103): map.insert keeps the FIRST id of a duplicated code */
    int lookup(unsigned long long c) const
    {
        for (int i = 0; i < n; i++)
            if (codes[i] == c) return i;
        return -1;
    }
};

bool load_dict(const char* name, Dict& d)
{
    for (int i = 0; i < ORBFE_NDICTS; i++)
        if (!strcmp(ORBFE_DICTS[i].name, name)) {
            d.name = name;
            d.nbits = ORBFE_DICTS[i].nbits;
            d.n = ORBFE_DICTS[i].ncodes;
            d.codes = ORBFE_DICTS[i].codes;
            d.tau = ORBFE_DICTS[i].tau; /* Dictionary::tau(), dictionary.cpp:108-248 */
            return true;
        }
    return false;
}

/* DictionaryBased::detect (dictionary_based.cpp:1059-1628) with getInnerCode (:1635-2322), touulong (:2372-2499),
 * rotate (:2501-2645).  in: S x S gray patch.  Returns id or -1; nRot = number of rotations. */
int decode_marker(const uint8_t* patch, int S, const Dict& dict, int* nRot)
{
    std::vector<uint8_t> bin((size_t)S * S);
    const int th = otsu_threshold(patch, S * S);
    for (int i = 0; i < S * S; i++) bin[i] = patch[i] > th ? 255 : 0;
    const int nb = (int)std::sqrt((double)dict.nbits);
    const int n = nb + 2;
    std::vector<int> ones(n * n, 0), tot(n * n, 0);
    for (int y = 0; y < S; y++) {
        const int my = (int)(float(n) * float(y) / float(S));
        for (int x = 0; x < S; x++) {
            const int mx = (int)(float(n) * float(x) / float(S));
            if (bin[y * S + x] > 125) ones[my * n + mx]++;
            tot[my * n + mx]++;
        }
    }
    std::vector<uint8_t> bits(n * n);
    for (int i = 0; i < n * n; i++) bits[i] = ones[i] > tot[i] / 2 ? 1 : 0;
    for (int y = 0; y < n; y++) {
        const int inc = (y == 0 || y == n - 1) ? 1 : n - 1;
        for (int x = 0; x < n; x += inc)
            if (bits[y * n + x] != 0) return -1;
    }
    std::vector<uint8_t> inner(nb * nb), tmp(nb * nb);
    for (int y = 0; y < nb; y++)
        for (int x = 0; x < nb; x++) inner[y * nb + x] = bits[(y + 1) * n + (x + 1)];
    unsigned long long ids[4];
    for (int r = 0; r < 4; r++) {
        unsigned long long v = 0;
        int b = 0;
        for (int y = nb - 1; y >= 0; y--)
            for (int x = nb - 1; x >= 0; x--) v |= (unsigned long long)inner[y * nb + x] << b++;
        ids[r] = v;
        for (int i = 0; i < nb; i++)
            for (int j = 0; j < nb; j++) tmp[i * nb + j] = inner[(nb - j - 1) * nb + i];
        inner = tmp;
    }
    if (ids[0] == 0) return -1; /* :1232 */
    for (int r = 0; r < 4; r++) {
        int id = dict.lookup(ids[r]);
        if (id >= 0) { *nRot = r; return id; }
    }
    /* error correction (dictionary_based.cpp:1423-1560): maxCorrection = int(float(tau) * rate); the dictionary's code map
     * (std::map<uint64_t, uint16_t>: ascending code, first id of a duplicated code) is walked entry by entry, the four rotations
     * inside; the first entry with hamm_distance < maxCorrection wins */
    const int maxCorrection = static_cast<int>(static_cast<float>(dict.tau) * dict.error_correction_rate);
    if (maxCorrection > 0) {
        std::map<unsigned long long, int> code_id;
        for (int i = 0; i < dict.n; i++) code_id.insert(std::make_pair(dict.codes[i], i));
        for (const auto& ci : code_id)
            for (int r = 0; r < 4; r++)
                if (__builtin_popcountll(ci.first ^ ids[r]) < maxCorrection) { *nRot = r; return ci.second; }
    }
    return -1;
}

/* ---------------------------------------------------------------- detector -- */
float pt_norm(float dx, float dy) { return (float)std::sqrt((double)dx * dx + (double)dy * dy); }

int perimeter(const Ptf c[4]) /* markerdetector_impl.cpp:11119-11296 */
{
    int sum = 0;
    for (int i = 0; i < 4; i++) {
        int i2 = (i + 1) % 4;
        sum += static_cast<int>(std::sqrt((c[i].x - c[i2].x) * (c[i].x - c[i2].x) + (c[i].y - c[i2].y) * (c[i].y - c[i2].y)));
    }
    return sum;
}

float get_area(const Ptf c[4]) /* marker.cpp:405-416 */
{
    float v01x = c[1].x - c[0].x, v01y = c[1].y - c[0].y, v03x = c[3].x - c[0].x, v03y = c[3].y - c[0].y;
    float area1 = std::fabs(v01x * v03y - v01y * v03x);
    float v21x = c[1].x - c[2].x, v21y = c[1].y - c[2].y, v23x = c[3].x - c[2].x, v23y = c[3].y - c[2].y;
    float area2 = std::fabs(v21x * v23y - v21y * v23x);
    return (area2 + area1) / 2.f;
}

/* least-squares line through points (interpolate2Dline, :11301-11890); out = (a, b, c) with a x + b y + c = 0 */
void interpolate2Dline(const std::vector<Ptf>& pts, float line[3])
{
    float minX, maxX, minY, maxY;
    minX = maxX = pts[0].x;
    minY = maxY = pts[0].y;
    for (size_t i = 1; i < pts.size(); i++) {
        minX = std::min(minX, pts[i].x); maxX = std::max(maxX, pts[i].x);
        minY = std::min(minY, pts[i].y); maxY = std::max(maxY, pts[i].y);
    }
    const bool xdom = (maxX - minX > maxY - minY);
    /* fit v = p*u + q */
    double su = 0, sv = 0, suu = 0, suv = 0;
    const double n = (double)pts.size();
    for (const Ptf& p : pts) {
        double u = xdom ? p.x : p.y, v = xdom ? p.y : p.x;
        su += u; sv += v; suu += u * u; suv += u * v;
    }
    double det = n * suu - su * su, pa, qa;
    if (std::fabs(det) > 1e-9 * std::max(1.0, n * suu)) {
        pa = (n * suv - su * sv) / det;
        qa = (sv * suu - su * suv) / det;
    } else { /* rank deficient: minimum-norm solution, as the SVD solve would return */
        double um = su / n, vm = sv / n;
        pa = vm * um / (um * um + 1.0);
        qa = vm / (um * um + 1.0);
    }
    if (xdom) { line[0] = (float)pa; line[1] = -1.f; line[2] = (float)qa; }
    else { line[0] = -1.f; line[1] = (float)pa; line[2] = (float)qa; }
}

Ptf cross_point(const float l1[3], const float l2[3]) /* getCrossPoint, :11899-12073 */
{
    double a = l1[0], b = l1[1], c = l2[0], d = l2[1], e = -(double)l1[2], f = -(double)l2[2];
    double det = a * d - b * c;
    Ptf r{0.f, 0.f};
    if (det != 0.0) { r.x = (float)((e * d - b * f) / det); r.y = (float)((a * f - e * c) / det); }
    return r;
}

struct Detector {
    Dict dict;
    bool corner_lines = true; /* Params::cornerRefinementM == CORNER_LINES (Frame.cc:137) */
    /* MarkerDetector::Params outside the configuration of Frame.cc:135-137 (markerdetector.h:158-196, markerdetector.cpp:364-402) */
    int corner_method = 1;     /* CORNER_SUBPIX 0, CORNER_LINES 1, CORNER_NONE 2 (kept in step with corner_lines) */
    int thres_method = 0;      /* THRES_ADAPTIVE 0, THRES_AUTO_FIXED 1 */
    int ThresHold = 7;         /* adaptive: the constant C; AUTO_FIXED: the global threshold carried from frame to frame */
    int NAttemptsAutoThresFix = 3;
    float minSize = 0.f;       /* what setDetectionMode(dm, minMarkerSize) leaves (the default -1 behaves like 0) */
    bool autoSize = false;
    float ts = 0.25f;
    bool enclosed = false;     /* Params::enclosedMarker (detectEnclosedMarkers) */
    /* Params::trackingMinDetections (markerdetector.h:187): a marker seen in that many calls and missing now is looked for among the
     * candidates the dictionary rejected, near its last position (markerdetector_impl.cpp:7107-7890) */
    int trackingMinDetections = 0;
    std::map<int, int> markerCounts;      /* id -> calls it was found in (decays by one per call without it) */
    std::vector<Candidate> prevMarkers;   /* the markers the previous call returned */
    int last_tracked = 0;                 /* markers the last call recovered that way (test read-back) */
    int last_attempts = 0;     /* threshold passes of the last call (test read-back) */
    int last_work_w = 0, last_work_h = 0;
    /* stage data of the last call (per-stage parity tests) */
    Image thres;
    std::vector<Image> pyramid;
    std::vector<Candidate> rects, prefiltered;
    int win = 0;

    /* buildPyramid (:1299-1488): halve while width > maxsize; cv::resize default INTER_LINEAR, which the exact
     * 2x case redirects to INTER_AREA (2x2 mean) */
    void build_pyramid(const Image& gray, int maxsize)
    {
        pyramid.clear();
        pyramid.push_back(gray);
        int npyr = 1, w = gray.w, h = gray.h;
        while (w > maxsize) { w /= 2; h /= 2; npyr++; }
        for (int i = 1; i < npyr; i++) {
            const Image& s = pyramid[i - 1];
            Image d(s.w / 2, s.h / 2);
            if (s.w == d.w * 2 && s.h == d.h * 2) {
                for (int y = 0; y < d.h; y++)
                    for (int x = 0; x < d.w; x++)
                        d.row(y)[x] = (uint8_t)((s.row(2 * y)[2 * x] + s.row(2 * y)[2 * x + 1] + s.row(2 * y + 1)[2 * x] +
                                                 s.row(2 * y + 1)[2 * x + 1] + 2) >> 2);
            } else {
                resize_generic(s, d);
            }
            pyramid.push_back(d);
        }
    }
    /* generic INTER_LINEAR (same arithmetic as the ORB pyramid's resize) */
    static void resize_generic(const Image& src, Image& dst)
    {
        const int sw = src.w, sh = src.h, dw = dst.w, dh = dst.h;
        const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
        std::vector<int> xofs(dw), yofs(dh), xa0(dw), xa1(dw), yb0(dh), yb1(dh);
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = orbfe_floor_d(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            xofs[dx] = sx;
            xa0[dx] = (short)orbfe_round_f((1.f - fx) * 2048.f);
            xa1[dx] = (short)orbfe_round_f(fx * 2048.f);
        }
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = orbfe_floor_d(fy);
            fy -= sy;
            yofs[dy] = sy;
            yb0[dy] = (short)orbfe_round_f((1.f - fy) * 2048.f);
            yb1[dy] = (short)orbfe_round_f(fy * 2048.f);
        }
        for (int dy = 0; dy < dh; dy++) {
            const uint8_t* S0 = src.row(std::min(std::max(yofs[dy], 0), sh - 1));
            const uint8_t* S1 = src.row(std::min(std::max(yofs[dy] + 1, 0), sh - 1));
            for (int dx = 0; dx < dw; dx++) {
                int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
                int h0 = S0[sx] * xa0[dx] + S0[sx1] * xa1[dx], h1 = S1[sx] * xa0[dx] + S1[sx1] * xa1[dx];
                dst.row(dy)[dx] = (uint8_t)((((yb0[dy] * (h0 >> 4)) >> 16) + ((yb1[dy] * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }

    /* thresholdAndDetectRectangles (:2705-3607 and :3765-3895) */
    void threshold_and_detect(const Image& gray)
    {
        int w = std::max(3, int(15 * float(gray.w) / 1920.));
        if (w % 2 == 0) w++;
        win = w; /* = _tooNearDistance */
        int enlarge_k = w; /* the adaptive window; the erosion size with THRES_AUTO_FIXED (:2871-2990) */
        if (thres_method == 1) { /* THRES_AUTO_FIXED (:2833-2870) */
            threshold_fixed_inv(gray, thres, ThresHold);
            if (enclosed) {
                enlarge_k = int(std::max(3.0, 3. / 1920. * float(thres.w)));
                if (enlarge_k % 2 == 0) enlarge_k++;
                erode_cross_xor(thres, enlarge_k);
            }
        } else adaptive_threshold_inv(gray, thres, w, ThresHold);
        std::vector<std::vector<Pt>> contours;
        find_contours_list(thres, contours);
        rects.clear();
        const int thres_len = int(3.5 * float(20)); /* lowResMarkerSize = 20 */
        std::vector<Pt> approx;
        for (auto& c : contours) {
            if (thres_len < (int)c.size()) {
                approx_poly_dp_closed(c, double(c.size()) * 0.05, approx);
                if (approx.size() == 4 && is_contour_convex(approx.data(), 4)) {
                    Candidate cd;
                    for (int j = 0; j < 4; j++) cd.c[j] = Ptf{(float)approx[j].x, (float)approx[j].y};
                    if (enclosed) enlarge_candidate(cd, float(enlarge_k) / 2.);
                    cd.contour = c;
                    rects.push_back(std::move(cd));
                }
            }
        }
    }

    /* enlargeMarkerCandidate (:10620-10690): both diagonals pushed outwards by `fact` pixels along the octant of their direction */
    static void enlarge_candidate(Candidate& cand, int fact)
    {
        for (int j = 0; j < 2; j++) {
            int startp = j, endp = (j + 2) % 4;
            if (cand.c[startp].x > cand.c[endp].x) std::swap(startp, endp);
            const float _180 = 3.14159f;
            const float _22 = 3.14159 / 8.f;
            const float _3_22 = 3. * 3.14159f / 8.f;
            const float _5_22 = 5.f * 3.14159f / 8.f;
            const float _7_22 = 7.f * 3.14159f / 8.f;
            int incx = 0, incy = 0;
            const float vx = cand.c[endp].x - cand.c[startp].x, vy = cand.c[endp].y - cand.c[startp].y;
            const float angle = std::atan2(vy, vx);
            if (_22 < angle && angle < 3 * _22) incx = incy = fact;
            else if (-_22 < angle && angle < _22) { incx = fact; incy = 0; }
            else if (-_3_22 < angle && angle < -_22) { incx = fact; incy = -fact; }
            else if (-_5_22 < angle && angle < -_3_22) { incx = 0; incy = -fact; }
            else if (-_7_22 < angle && angle < -_5_22) { incx = -fact; incy = -fact; }
            else if ((-_180 < angle && angle < -_7_22) || (_7_22 < angle && angle < _180)) { incx = -fact; incy = 0; }
            else if (_5_22 < angle && angle < _7_22) { incx = -fact; incy = fact; }
            else if (_3_22 < angle && angle < _5_22) { incx = fact; incy = fact; }
            cand.c[endp].x += incx; cand.c[endp].y += incy;
            cand.c[startp].x -= incx; cand.c[startp].y -= incy;
        }
    }

    /* prefilterCandidates (:4347-5347) */
    void prefilter(int W, int H)
    {
        std::vector<Candidate>& c = rects;
        for (auto& cd : c) {
            double dx1 = cd.c[1].x - cd.c[0].x, dy1 = cd.c[1].y - cd.c[0].y;
            double dx2 = cd.c[2].x - cd.c[0].x, dy2 = cd.c[2].y - cd.c[0].y;
            double o = (dx1 * dy2) - (dy1 * dx2);
            if (o < 0.0) std::swap(cd.c[1], cd.c[3]);
        }
        std::vector<std::pair<int, int>> tooNear;
        for (size_t i = 0; i < c.size(); i++)
            for (size_t j = i + 1; j < c.size(); j++) {
                float d[4];
                for (int k = 0; k < 4; k++) d[k] = pt_norm(c[i].c[k].x - c[j].c[k].x, c[i].c[k].y - c[j].c[k].y);
                if (d[0] < win && d[1] < win && d[2] < win && d[3] < win) tooNear.push_back({(int)i, (int)j});
            }
        std::vector<bool> rm(c.size(), false);
        for (auto& pr : tooNear) {
            if (perimeter(c[pr.first].c) > perimeter(c[pr.second].c)) rm[pr.second] = true;
            else rm[pr.first] = true;
        }
        const int bx = static_cast<int>(0.015f * float(W)), by = static_cast<int>(0.015f * float(H));
        for (size_t i = 0; i < c.size(); i++)
            for (int k = 0; k < 4; k++)
                if (c[i].c[k].x < bx || c[i].c[k].y < by || c[i].c[k].x > W - bx || c[i].c[k].y > H - by) rm[i] = true;
        prefiltered.clear();
        for (size_t i = 0; i < c.size(); i++)
            if (!rm[i]) prefiltered.push_back(c[i]);
    }

    /* refineCornerWithContourLines (:8978-10044) with empty camera matrices */
    static void refine_corners(Candidate& m)
    {
        const std::vector<Pt>& contour = m.contour;
        int ci[4] = {-1, -1, -1, -1};
        float dist[4] = {FLT_MAX, FLT_MAX, FLT_MAX, FLT_MAX};
        for (unsigned j = 0; j < contour.size(); j++)
            for (unsigned k = 0; k < 4; k++) {
                float d = (contour[j].x - m.c[k].x) * (contour[j].x - m.c[k].x) +
                          (contour[j].y - m.c[k].y) * (contour[j].y - m.c[k].y);
                if (d < dist[k]) { ci[k] = j; dist[k] = d; }
            }
        bool inverse;
        if ((ci[1] > ci[0]) && (ci[2] > ci[1] || ci[2] < ci[0])) inverse = false;
        else if (ci[2] > ci[1] && ci[2] < ci[0]) inverse = false;
        else inverse = true;
        const int inc = inverse ? -1 : 1;
        std::vector<Ptf> lines[4];
        const int sz = (int)contour.size();
        for (unsigned l = 0; l < 4; l++) {
            /* bounded: the reference loop can only run away on degenerate index sets; cap at 2*sz iterations */
            int guard = 0;
            for (int j = ci[l]; j != ci[(l + 1) % 4] && guard < 2 * sz + 4; j += inc, guard++) {
                if (j == sz && !inverse) j = 0;
                else if (j == 0 && inverse) j = sz - 1;
                lines[l].push_back(Ptf{(float)contour[j].x, (float)contour[j].y});
                if (j == ci[(l + 1) % 4]) break;
            }
        }
        float L[4][3];
        for (int l = 0; l < 4; l++) {
            if (lines[l].empty()) { L[l][0] = L[l][1] = L[l][2] = 0.f; continue; }
            interpolate2Dline(lines[l], L[l]);
        }
        for (unsigned i = 0; i < 4; i++) m.c[i] = cross_point(L[(i - 1) % 4], L[i]); /* unsigned (i-1)%4: 3 for i=0 */
    }

    /* marker_analyzer (markerdetector_impl.h:2460-2530): centre, area and the inside test of a quadrilateral */
    struct Quad {
        Ptf c[4], center;
        float area;
        void set(const Ptf q[4])
        {
            for (int k = 0; k < 4; k++) c[k] = q[k];
            const float ax = c[1].x - c[0].x, ay = c[1].y - c[0].y, bx = c[3].x - c[0].x, by = c[3].y - c[0].y;
            const float a1 = std::fabs(ax * by - ay * bx);
            const float cx = c[1].x - c[2].x, cy = c[1].y - c[2].y, dx = c[3].x - c[2].x, dy = c[3].y - c[2].y;
            const float a2 = std::fabs(cx * dy - cy * dx);
            area = (a2 + a1) / 2.f;
            center = Ptf{0, 0};
            for (int k = 0; k < 4; k++) { center.x += c[k].x; center.y += c[k].y; }
            center.x = (float)(center.x * (1. / 4.)); center.y = (float)(center.y * (1. / 4.));
        }
        static float signed_dist(Ptf p1, Ptf p2, Ptf p)
        {
            return ((p1.y - p2.y) * p.x + (p2.x - p1.x) * p.y + (p1.x * p2.y - p2.x * p1.y)) /
                   std::sqrt((p2.x - p1.x) * (p2.x - p1.x) + (p2.y - p1.y) * (p2.y - p1.y));
        }
        bool is_into(Ptf p) const
        {
            for (int k = 0; k < 4; k++)
                if (signed_dist(c[k], c[(k + 1) % 4], p) < 0) return false;
            return true;
        }
    };
    /* direction agreement of the first sides (the lambda at :7700-7760): unit vectors in float, scaled by a double reciprocal norm */
    static float side_agreement(const Ptf a[4], const Ptf b[4])
    {
        Ptf u{a[1].x - a[0].x, a[1].y - a[0].y}, v{b[1].x - b[0].x, b[1].y - b[0].y};
        const double nu = 1. / std::sqrt((double)u.x * u.x + (double)u.y * u.y), nv = 1. / std::sqrt((double)v.x * v.x + (double)v.y * v.y);
        u.x = (float)(u.x * nu); u.y = (float)(u.y * nu);
        v.x = (float)(v.x * nv); v.y = (float)(v.y * nv);
        return u.x * v.x + u.y * v.y;
    }
    /* the tracking block of detect() (:7107-7890); `rejected` = the candidates the dictionary did not accept, in candidate order */
    void track_missing(std::vector<Candidate>& detected, std::vector<Candidate>& rejected)
    {
        last_tracked = 0;
        if (trackingMinDetections <= 0) return;
        auto found = [&](int id) { for (auto& m : detected) if (m.id == id) return true; return false; };
        for (auto& mc : markerCounts)
            if (!found(mc.first)) mc.second = std::max(mc.second - 1, 0);
        struct Info { Quad q; int best = -1; double dist = std::numeric_limits<double>::max(); const Candidate* prev = nullptr; };
        std::map<int, Info> need;
        for (auto& m : prevMarkers)
            if (!found(m.id) && markerCounts.count(m.id) != 0 && markerCounts.at(m.id) >= trackingMinDetections && !need.count(m.id)) {
                Info in; in.q.set(m.c); in.prev = &m;
                need.insert({m.id, in});
            }
        if (!need.empty()) {
            for (size_t ci = 0; ci < rejected.size(); ci++) {
                Quad qc; qc.set(rejected[ci].c);
                for (auto& kv : need) {
                    Info& in = kv.second;
                    if (!in.q.is_into(qc.center)) continue;
                    const float dx = in.q.center.x - qc.center.x, dy = in.q.center.y - qc.center.y;
                    const double dist = std::sqrt((double)dx * dx + (double)dy * dy);
                    const float sizeDiff = std::fabs(in.q.area - qc.area) / in.q.area;
                    if (sizeDiff < 0.3f && dist < in.dist) { in.best = (int)ci; in.dist = dist; }
                }
            }
            std::vector<bool> used(rejected.size(), false);
            for (auto& kv : need) {
                Info& in = kv.second;
                if (in.best == -1 || used[in.best]) continue; /* (the reference hands a candidate claimed twice on as an empty marker: undefined; first claim wins here) */
                Candidate m = rejected[in.best];
                m.id = kv.first;
                int best_r = -1;
                double best_s = -1;
                for (int r = 0; r < 4; r++) {
                    Ptf rot[4];
                    for (int k = 0; k < 4; k++) rot[k] = m.c[(k + r) % 4];
                    const float sc = side_agreement(in.prev->c, rot);
                    if (sc > best_s) { best_r = r; best_s = sc; }
                }
                if (best_r > 0) std::rotate(m.c, m.c + best_r, m.c + 4);
                detected.push_back(std::move(m));
                used[in.best] = true;
                last_tracked++;
            }
        }
        for (auto& m : detected) {
            if (markerCounts.count(m.id) == 0) markerCounts[m.id] = 1;
            else markerCounts[m.id]++;
        }
    }

    /* setDetectionMode (markerdetector.cpp:374-391) */
    void set_detection_mode(int dm, float minMarkerSize)
    {
        minSize = minMarkerSize;
        if (dm == 0) { autoSize = false; ts = 0.25f; thres_method = 0; ThresHold = 7; }
        else if (dm == 1) { autoSize = false; ts = 0.25f; thres_method = 1; ThresHold = 100; }
        else { thres_method = 1; ThresHold = 100; autoSize = true; ts = 0.3f; }
    }
    /* setCornerRefinementMethod (:392-395) */
    void set_corner_method(int m)
    {
        corner_method = m;
        corner_lines = m == 1;
        if (m != 0) minSize = 0;
    }
    /* getMinMarkerSizePix (:10690-10760) with minSize_pix == -1 */
    int min_marker_size_pix(int w, int h) const
    {
        const int maxDim = std::max(w, h);
        return (int)(static_cast<float>(minSize) * static_cast<float>(maxDim));
    }
    /* cornerUpsample (:14028-14220): from the level of the pyramid just above the working size down to the input, scale the corners
     * and refine them with cornerSubPix(TermCriteria(MAX_ITER, 4, 0.5)) */
    void corner_upsample(std::vector<Candidate>& ms, int work_w)
    {
        if (ms.empty()) return;
        int start = 0;
        for (size_t i = 0; i < pyramid.size(); i++) {
            if (work_w < pyramid[i].w) start = (int)i;
            else break;
        }
        int prev_w = work_w;
        for (int l = start; l >= 0; l--) {
            const float factor = float(pyramid[l].w) / float(prev_w);
            std::vector<Ptf> pts;
            for (auto& m : ms)
                for (int k = 0; k < 4; k++) {
                    m.c[k].x *= factor; m.c[k].y *= factor;
                    pts.push_back(m.c[k]);
                }
            const int halfw = (int)(0.5 + 2.5 * factor);
            corner_subpix(pyramid[l], pts, halfw, 4, 0.0);
            size_t q = 0;
            for (auto& m : ms)
                for (int k = 0; k < 4; k++) m.c[k] = pts[q++];
            prev_w = pyramid[l].w;
        }
    }

    /* MarkerDetector_Impl::detect (:5870-8800) */
    int detect(const Image& input, std::vector<Candidate>& out)
    {
        out.clear();
        const int nb = (int)std::sqrt((double)dict.nbits);
        const int S = 5 * (nb + 2); /* getMarkerWarpSize (:1199-1292): markerWarpPixSize * nSubdivisions */
        /* the working image: reduced with INTER_NEAREST when markers below minSize need not be found (:5990-6090) */
        Image reduced;
        const Image* work = &input;
        const int minpix = min_marker_size_pix(input.w, input.h);
        if (20 < minpix) { /* lowResMarkerSize = 20 */
            const float scale = float(20) / float(minpix);
            if (scale < 0.9) {
                int rw = float(input.w) * scale + 0.5, rh = float(input.h) * scale + 0.5;
                if (rw % 2 != 0) rw++;
                if (rh % 2 != 0) rh++;
                reduced = Image(rw, rh);
                resize_nearest(input, reduced);
                work = &reduced;
            }
        }
        last_work_w = work->w; last_work_h = work->h;
        build_pyramid(input, 2 * S);
        const float desiredarea = std::pow(static_cast<float>(S), 2.f);
        std::vector<uint8_t> patch((size_t)S * S);
        std::vector<float> hist(256, 0.f);
        std::vector<Candidate> rejected;
        int nattempts = 0;
        bool again;
        last_attempts = 0;
        do {
            last_attempts++;
            threshold_and_detect(*work);
            prefilter(work->w, work->h);
            out.clear();
            rejected.clear();
            for (auto& v : hist) v = 0;
            for (auto& cand : prefiltered) {
                size_t lvl = 0;
                for (size_t p = 1; p < pyramid.size(); p++) {
                    if (get_area(cand.c) / std::pow(4, p) >= desiredarea) lvl = p;
                    else break;
                }
                const Image& im = pyramid[lvl];
                const float ratio = float(im.w) / float(work->w);
                Ptf q[4];
                for (int k = 0; k < 4; k++) q[k] = Ptf{cand.c[k].x * ratio, cand.c[k].y * ratio};
                double Minv[9];
                int nRot = 0, id = -1;
                if (perspective_inverse_map(q, S, Minv)) {
                    warp_perspective(im, Minv, S, patch.data());
                    id = decode_marker(patch.data(), S, dict, &nRot);
                }
                if (id < 0) rejected.push_back(cand);
                if (id >= 0) {
                    Candidate m = cand;
                    m.id = id;
                    std::rotate(m.c, m.c + 4 - nRot, m.c + 4);
                    out.push_back(std::move(m));
                    if (thres_method == 1) /* addToImageHist (:6121-6190): the warped patch, before the labeler thresholds its copy */
                        for (size_t i = 0; i < patch.size(); i++) hist[patch[i]]++;
                }
            }
            /* nothing found with the carried-over threshold: try a random one (:6903-6990) */
            if (out.size() == 0 && thres_method == 1 && ++nattempts < NAttemptsAutoThresFix) {
                ThresHold = 10 + rand() % 230;
                again = true;
            } else again = false;
        } while (again);
        if (thres_method == 1) { /* the threshold for the next call: Otsu over the detected markers' pixels (:7003-7040) */
            const int t = otsu_of_histogram(hist);
            if (t > 0) ThresHold = float(t);
        }
        if (input.w != work->w) corner_upsample(out, work->w); /* :7046-7071 */
        track_missing(out, rejected);                           /* :7107-7890 (rejected candidates keep the working image's coordinates) */
        /* sort by id; among equal ids keep the larger perimeter (:8153-8365) */
        std::stable_sort(out.begin(), out.end(), [](const Candidate& a, const Candidate& b) { return a.id < b.id; });
        std::vector<bool> rm(out.size(), false);
        for (int i = 0; i < (int)out.size() - 1; i++)
            for (int j = i + 1; j < (int)out.size() && !rm[i]; j++)
                if (out[i].id == out[j].id) {
                    if (perimeter(out[i].c) < perimeter(out[j].c)) rm[i] = true;
                    else rm[j] = true;
                }
        std::vector<Candidate> kept;
        for (size_t i = 0; i < out.size(); i++)
            if (!rm[i]) kept.push_back(std::move(out[i]));
        out.swap(kept);
        if (out.size() > 0 && input.w == work->w && input.h == work->h) { /* :8420-8701 */
            if (corner_method == 0) {
                const int halfw = 4 * float(input.w) / float(work->w) + 0.5;
                std::vector<Ptf> pts;
                for (auto& m : out)
                    for (int k = 0; k < 4; k++) pts.push_back(m.c[k]);
                corner_subpix(input, pts, halfw, 12, 0.005);
                size_t q = 0;
                for (auto& m : out)
                    for (int k = 0; k < 4; k++) m.c[k] = pts[q++];
            } else if (corner_method == 1) /* CORNER_LINES (:8634-8701); CORNER_NONE leaves the approxPolyDP corners */
                for (auto& m : out) refine_corners(m);
        }
        /* the smallest marker of this frame sets the next frame's minSize (:8790-8880) */
        float mlength = std::numeric_limits<float>::max();
        for (const auto& m : out) {
            float l = 0;
            for (int c = 0; c < 4; c++) {
                const float dx = m.c[c].x - m.c[(c + 1) % 4].x, dy = m.c[c].y - m.c[(c + 1) % 4].y;
                l += std::sqrt((double)dx * dx + (double)dy * dy); /* float += cv::norm (double) */
            }
            if (mlength > l) mlength = l;
        }
        float markerMinSize;
        if (mlength != std::numeric_limits<float>::max()) markerMinSize = mlength / (4 * std::max(input.w, input.h));
        else markerMinSize = 0;
        if (autoSize) minSize = markerMinSize * (1 - ts);
        prevMarkers = out;
        return (int)out.size();
    }
};

struct MarkerRec {
    int32_t id;
    float corners[4][2];
};

Image make_image(const uint8_t* img, int rows, int cols, size_t step)
{
    Image g(cols, rows);
    for (int y = 0; y < rows; y++) memcpy(g.row(y), img + (size_t)y * step, cols);
    return g;
}

} // namespace

extern "C" {

void* oracle_aruco_create(const char* dictionary)
{
    Detector* d = new Detector();
    if (!load_dict(dictionary, d->dict)) { delete d; return nullptr; }
    return d;
}
void oracle_aruco_destroy(void* h) { delete (Detector*)h; }
void oracle_aruco_set_params(void* h, float error_correction_rate, int corner_lines)
{
    ((Detector*)h)->dict.error_correction_rate = error_correction_rate;
    ((Detector*)h)->corner_lines = corner_lines != 0;
    ((Detector*)h)->corner_method = corner_lines ? 1 : 2;
}

/* setDetectionMode(dm, minMarkerSize) then setCornerRefinementMethod(corner_method), in the order a caller of the reference uses */
void oracle_aruco_set_detection_mode(void* h, int dm, float min_marker_size) { ((Detector*)h)->set_detection_mode(dm, min_marker_size); }
void oracle_aruco_set_corner_method(void* h, int m) { ((Detector*)h)->set_corner_method(m); }
void oracle_aruco_set_enclosed(void* h, int on) { ((Detector*)h)->enclosed = on != 0; }
void oracle_aruco_set_tracking(void* h, int min_detections) { ((Detector*)h)->trackingMinDetections = min_detections; }
/* state read-back: 0 ThresHold, 1 threshold passes of the last call, 2 / 3 working width / height of the last call */
int oracle_aruco_state(void* h, int which)
{
    Detector* d = (Detector*)h;
    return which == 0 ? d->ThresHold : which == 1 ? d->last_attempts : which == 2 ? d->last_work_w : which == 3 ? d->last_work_h : d->last_tracked;
}
float oracle_aruco_min_size(void* h) { return ((Detector*)h)->minSize; }
void oracle_bgr_to_gray(const uint8_t* bgr, int rows, int cols, size_t step, uint8_t* out, int bits15)
{
    Image g;
    bgr_to_gray(bgr, step, cols, rows, g, bits15);
    memcpy(out, g.d.data(), g.d.size());
}
void oracle_resize_nearest(const uint8_t* src, int w, int h, uint8_t* dst, int dw, int dh)
{
    Image s(w, h), d(dw, dh);
    memcpy(s.d.data(), src, (size_t)w * h);
    resize_nearest(s, d);
    memcpy(dst, d.d.data(), d.d.size());
}
/* pts: n (x, y) float pairs, refined in place */
void oracle_corner_subpix(const uint8_t* src, int w, int h, float* pts, int n, int win, int max_iters, double eps)
{
    Image s(w, h);
    memcpy(s.d.data(), src, (size_t)w * h);
    std::vector<Ptf> c(n);
    for (int i = 0; i < n; i++) c[i] = Ptf{pts[2 * i], pts[2 * i + 1]};
    corner_subpix(s, c, win, max_iters, eps);
    for (int i = 0; i < n; i++) { pts[2 * i] = c[i].x; pts[2 * i + 1] = c[i].y; }
}
int oracle_otsu_of_histogram(const float* hist256)
{
    std::vector<float> h(hist256, hist256 + 256);
    return otsu_of_histogram(h);
}

int oracle_aruco_detect(void* h, const uint8_t* img, int rows, int cols, size_t step, void* out, int capacity)
{
    Detector* d = (Detector*)h;
    std::vector<Candidate> m;
    d->detect(make_image(img, rows, cols, step), m);
    MarkerRec* o = (MarkerRec*)out;
    for (int i = 0; i < (int)m.size() && i < capacity; i++) {
        o[i].id = m[i].id;
        for (int k = 0; k < 4; k++) { o[i].corners[k][0] = m[i].c[k].x; o[i].corners[k][1] = m[i].c[k].y; }
    }
    return (int)m.size();
}

/* stage 0: thresholded image; 1..: detector pyramid level (stage-1). Returns 0 on success. */
int oracle_aruco_stage_image(void* h, int stage, uint8_t* out, int* w, int* hh)
{
    Detector* d = (Detector*)h;
    const Image* im = nullptr;
    if (stage == 0) im = &d->thres;
    else if (stage - 1 < (int)d->pyramid.size()) im = &d->pyramid[stage - 1];
    if (!im) return -1;
    *w = im->w; *hh = im->h;
    if (out) memcpy(out, im->d.data(), im->d.size());
    return 0;
}
/* 0: rectangles after approxPolyDP/convexity, 1: after prefilterCandidates, 2: pyramid levels */
int oracle_aruco_stage_count(void* h, int which)
{
    Detector* d = (Detector*)h;
    return which == 0 ? (int)d->rects.size() : which == 1 ? (int)d->prefiltered.size() : (int)d->pyramid.size();
}
/* corners (float[4][2]) + contour length of stage-`which` candidates */
int oracle_aruco_candidates(void* h, int which, float* out, int capacity)
{
    Detector* d = (Detector*)h;
    const std::vector<Candidate>& v = which == 0 ? d->rects : d->prefiltered;
    for (int i = 0; i < (int)v.size() && i < capacity; i++) {
        for (int k = 0; k < 4; k++) { out[i * 9 + 2 * k] = v[i].c[k].x; out[i * 9 + 2 * k + 1] = v[i].c[k].y; }
        out[i * 9 + 8] = (float)v[i].contour.size();
    }
    return (int)v.size();
}

void oracle_adaptive_threshold(const uint8_t* src, int w, int h, uint8_t* dst, int win, int C)
{
    Image s(w, h), d;
    memcpy(s.d.data(), src, (size_t)w * h);
    adaptive_threshold_inv(s, d, win, C);
    memcpy(dst, d.d.data(), (size_t)w * h);
}

/* contours of a binary image: lengths[] per contour (in output order) and the concatenated points (x,y int32 pairs).
 * Returns the number of contours; point capacity is in points. */
int oracle_find_contours(const uint8_t* img, int w, int h, int32_t* lengths, int max_contours, int32_t* points,
                         int max_points)
{
    Image s(w, h);
    memcpy(s.d.data(), img, (size_t)w * h);
    std::vector<std::vector<Pt>> c;
    find_contours_list(s, c);
    int np = 0;
    for (int i = 0; i < (int)c.size(); i++) {
        if (i < max_contours) lengths[i] = (int)c[i].size();
        for (auto& p : c[i]) {
            if (np < max_points) { points[2 * np] = p.x; points[2 * np + 1] = p.y; }
            np++;
        }
    }
    return (int)c.size();
}

/* quad: 8 floats; out: S*S bytes */
void oracle_warp_perspective35(const uint8_t* img, int w, int h, const float* quad, uint8_t* out, int S)
{
    Image s(w, h);
    memcpy(s.d.data(), img, (size_t)w * h);
    Ptf q[4];
    for (int k = 0; k < 4; k++) q[k] = Ptf{quad[2 * k], quad[2 * k + 1]};
    double Minv[9];
    memset(out, 0, (size_t)S * S);
    if (perspective_inverse_map(q, S, Minv)) warp_perspective(s, Minv, S, out);
}

int oracle_otsu_threshold(const uint8_t* p, int n) { return otsu_threshold(p, n); }

/* returns id or -1; rot gets the rotation count */
int oracle_decode_marker(void* h, const uint8_t* patch, int S, int* rot)
{
    Detector* d = (Detector*)h;
    int r = 0;
    int id = decode_marker(patch, S, d->dict, &r);
    if (rot) *rot = r;
    return id;
}

int oracle_approx_poly(const int32_t* pts, int n, double eps, int32_t* out, int capacity)
{
    std::vector<Pt> s(n), d;
    for (int i = 0; i < n; i++) s[i] = Pt{pts[2 * i], pts[2 * i + 1]};
    approx_poly_dp_closed(s, eps, d);
    for (int i = 0; i < (int)d.size() && i < capacity; i++) { out[2 * i] = d[i].x; out[2 * i + 1] = d[i].y; }
    return (int)d.size();
}

} /* extern "C" */
