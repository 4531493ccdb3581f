// aruco_modes.hip -- the kernels of MarkerDetector::Params outside the configuration of src/Frame.cc:135-137:
//   THRES_AUTO_FIXED (DM_FAST / DM_VIDEO_FAST, markerdetector.cpp:380-391): a global threshold, the grey-level histogram of the
//     detected markers' warped patches (the next frame's threshold is Otsu over it, markerdetector_impl.cpp:6121-6380, :7003-7040);
//   Params::minSize > 0 (:5990-6090): detection on an INTER_NEAREST reduction of the frame, corners brought back through the /2
//     pyramid with cv::cornerSubPix at every level (cornerUpsample, :14028-14220);
//   CORNER_SUBPIX (:8430-8620): cv::cornerSubPix on the full-resolution frame;
//   CV_8UC3 input (:5892): cvtColor(BGR2GRAY).
// All of them are byte / HBM work of a few hundred KB per frame, or -- cornerSubPix -- a serial chain per corner: nothing here
// is near a roofline, the kernels exist so that the reference's whole parameter surface runs on the device.
#include <hip/hip_runtime.h>

#include "../../include/orbfe.h"
#include "aruco_kernels.hpp"
#include "orbfe_common.hpp"

namespace orbfe {

// cv::threshold(src, dst, thr, 255, THRESH_BINARY_INV) as a bit image: bit = src <= thr.  One 32-pixel word per thread.
__global__ __launch_bounds__(256) void k_fixed_threshold(ImgView src, int W, int H, int thr, uint32_t* __restrict__ bits,
                                                         size_t bits_fstride, int wpr)
{
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= wpr * H) return;
    const int y = i / wpr, j = i - y * wpr, x0 = j * 32;
    const uint8_t* row = src.base + (size_t)f * src.fstride + (size_t)y * src.pitch;
    uint32_t word = 0;
    const bool dwords = (((uintptr_t)row) & 3) == 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int x = x0 + 4 * k;
        if (x + 3 < W && dwords) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(row + x);
#pragma unroll
            for (int b = 0; b < 4; b++) word |= (uint32_t)((int)((v >> (8 * b)) & 255u) <= thr) << (4 * k + b);
        } else {
            for (int b = 0; b < 4; b++)
                if (x + b < W) word |= (uint32_t)((int)row[x + b] <= thr) << (4 * k + b);
        }
    }
    bits[(size_t)f * bits_fstride + i] = word;
}

// Params::detectEnclosedMarkers with THRES_AUTO_FIXED (markerdetector_impl.cpp:2871-2950): cv::erode with a MORPH_CROSS element of
// size 2 r + 1 (pixels outside the image do not constrain the minimum) and bitwise_xor with the thresholded image, on the bit
// image: out = b & ~(AND over the row run & AND over the column run).  One 32-pixel word per thread; r <= 15.
__global__ __launch_bounds__(256) void k_erode_cross_xor(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, size_t bits_fstride,
                                                         int wpr, int W, int H, int r)
{
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i >= wpr * H) return;
    const int y = i / wpr, j = i - y * wpr;
    const uint32_t* b = in + (size_t)f * bits_fstride;
    // a word with the pixels outside the image set (they must not clear anything): bits >= W of the last word, words outside the row
    auto ext = [&](int yy, int jj) -> uint32_t {
        if (yy < 0 || yy >= H || jj < 0 || jj >= wpr) return 0xffffffffu;
        const int nvalid = W - 32 * jj;
        const uint32_t v = b[(size_t)yy * wpr + jj];
        return nvalid >= 32 ? v : v | ~((1u << nvalid) - 1u);
    };
    const uint32_t cur = ext(y, j), prv = ext(y, j - 1), nxt = ext(y, j + 1);
    uint32_t hand = cur, vand = cur;
    for (int d = 1; d <= r; d++) {
        hand &= (cur >> d) | (nxt << (32 - d));   // pixel x + d
        hand &= (cur << d) | (prv >> (32 - d));   // pixel x - d
        vand &= ext(y - d, j) & ext(y + d, j);
    }
    out[(size_t)f * bits_fstride + i] = b[(size_t)y * wpr + j] & ~(hand & vand);
}

// enlargeMarkerCandidate (markerdetector_impl.cpp:10620-10690, Params::detectEnclosedMarkers): both diagonals of every rectangle
// candidate pushed outwards by `fact` pixels along the octant of their direction (the reference's own octant limits, 3.14159 / 8 ...)
__global__ __launch_bounds__(256) void k_enlarge_candidates(ArRect* __restrict__ rects, int rect_cap, const int32_t* __restrict__ counts, int fact)
{
    const int f = blockIdx.x, i = threadIdx.x;
    if (i >= min(counts[f * 4 + 1], rect_cap)) return;
    ArRect* R = rects + (size_t)f * rect_cap + i;
    for (int j = 0; j < 2; j++) {
        int startp = j, endp = (j + 2) % 4;
        if (R->c[startp][0] > R->c[endp][0]) { const int t = startp; startp = endp; endp = t; }
        const float _180 = 3.14159f;
        const float _22 = 3.14159 / 8.f;
        const float _3_22 = 3. * 3.14159f / 8.f;
        const float _5_22 = 5.f * 3.14159f / 8.f;
        const float _7_22 = 7.f * 3.14159f / 8.f;
        int incx = 0, incy = 0;
        const float vx = R->c[endp][0] - R->c[startp][0], vy = R->c[endp][1] - R->c[startp][1];
        const float angle = atan2f(vy, vx);
        if (_22 < angle && angle < 3 * _22) incx = incy = fact;
        else if (-_22 < angle && angle < _22) { incx = fact; incy = 0; }
        else if (-_3_22 < angle && angle < -_22) { incx = fact; incy = -fact; }
        else if (-_5_22 < angle && angle < -_3_22) { incx = 0; incy = -fact; }
        else if (-_7_22 < angle && angle < -_5_22) { incx = -fact; incy = -fact; }
        else if ((-_180 < angle && angle < -_7_22) || (_7_22 < angle && angle < _180)) { incx = -fact; incy = 0; }
        else if (_5_22 < angle && angle < _7_22) { incx = -fact; incy = fact; }
        else if (_3_22 < angle && angle < _5_22) { incx = fact; incy = fact; }
        R->c[endp][0] += (float)incx; R->c[endp][1] += (float)incy;
        R->c[startp][0] -= (float)incx; R->c[startp][1] -= (float)incy;
    }
}

// cv::resize(INTER_NEAREST): sx = min(floor(x * ifx), sw - 1) with ifx = 1 / (dw / sw) in double (resize.cpp resizeNN)
__global__ __launch_bounds__(256) void k_resize_nearest(ImgView src, ImgView dst, int sw, int sh, int dw, int dh, double ifx, double ify)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= dw || y >= dh) return;
    const int sx = min((int)floor((double)x * ifx), sw - 1), sy = min((int)floor((double)y * ify), sh - 1);
    dst.base_w[(size_t)f * dst.fstride + (size_t)y * dst.pitch + x] = src.base[(size_t)f * src.fstride + (size_t)sy * src.pitch + sx];
}

// cvtColor(BGR2GRAY), 8-bit: 14 fractional bits (OpenCV <= 3.4.1) or 15 (3.4.2+ / 4.x)
__global__ __launch_bounds__(256) void k_bgr_to_gray(const uint8_t* __restrict__ bgr, size_t bgr_fstride, size_t step, ImgView dst,
                                                     int W, int H, int bits15)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), f = blockIdx.z;
    if (x >= W || y >= H) return;
    const uint8_t* s = bgr + (size_t)f * bgr_fstride + (size_t)y * step + 3 * (size_t)x;
    const int b = s[0], g = s[1], r = s[2];
    dst.base_w[(size_t)f * dst.fstride + (size_t)y * dst.pitch + x] =
        bits15 ? (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15) : (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
}

// Histogram of the warped patches of the candidates the dictionary accepted (addToImageHist is called before sort / dedupe, so
// every accepted candidate counts): a workgroup per frame, a thread per bin, over the batch's decode work list.
__global__ __launch_bounds__(256) void k_marker_hist(const uint32_t* __restrict__ work, const int32_t* __restrict__ wctr,
                                                     const int32_t* __restrict__ result, int rect_cap,
                                                     const uint16_t* __restrict__ hist, uint32_t* __restrict__ out)
{
    const int f = blockIdx.x, bin = threadIdx.x;
    const int nitems = wctr[0];
    uint32_t acc = 0;
    for (int it = 0; it < nitems; it++) {
        const uint32_t wi = work[it];
        if ((int)(wi >> 16) != f) continue;
        if (result[((size_t)f * rect_cap + (wi & 0xffffu)) * 2] < 0) continue;
        acc += hist[(size_t)it * 256 + bin];
    }
    out[(size_t)f * 256 + bin] = acc;
}

// ---- cv::cornerSubPix for one corner, run by one wave: the lanes fill the (2 win + 3)^2 window of getRectSubPix (each sample
// with the reference's own float expression), lane 0 alone runs the gradient sums in the reference's order -- they are
// double-precision accumulations whose rounding decides when the iteration stops, so their order is kept.
#define SP_MAXWIN 8
#define SP_BUF ((2 * SP_MAXWIN + 3) * (2 * SP_MAXWIN + 3))

__device__ __forceinline__ float sp_sample(const uint8_t* img, int pitch, int W, int H, int ww, int wh, int ipx, int ipy, float a11,
                                           float a12, float a21, float a22, float b1, float b2, int i, int j)
{
    if (0 <= ipx && ipx < W - ww && 0 <= ipy && ipy < H - wh) {
        const uint8_t* s = img + (size_t)(ipy + i) * pitch + ipx;
        const uint8_t* s2 = s + pitch;
        return (float)s[j] * a11 + (float)s[j + 1] * a12 + (float)s2[j] * a21 + (float)s2[j + 1] * a22;
    }
    // adjustRect (samplers.cpp): the window's part inside the image is columns [rx, rw) and rows [ry, rh); the border is replicated
    int col0 = 0, row0 = 0, rx, rw, ry, rh;
    if (ipx >= 0) { col0 += ipx; rx = 0; }
    else { rx = -ipx; if (rx > ww) rx = ww; }
    if (ipx < W - ww) rw = ww;
    else { rw = W - ipx - 1; if (rw < 0) { col0 += rw; rw = 0; } }
    if (ipy >= 0) { row0 += ipy; ry = 0; }
    else ry = -ipy;
    if (ipy < H - wh) rh = wh;
    else { rh = H - ipy - 1; if (rh < 0) { row0 += rh; rh = 0; } }
    col0 -= rx;
    const int srow = row0 + max(0, min(i, rh) - ry), s2row = srow + ((i >= ry && i < rh) ? 1 : 0);
    const uint8_t* s = img + (size_t)srow * pitch + col0;
    const uint8_t* s2 = img + (size_t)s2row * pitch + col0;
    if (j < rx) return (float)s[rx] * b1 + (float)s2[rx] * b2;
    if (j < rw) return (float)s[j] * a11 + (float)s[j + 1] * a12 + (float)s2[j] * a21 + (float)s2[j + 1] * a22;
    return (float)s[rw] * b1 + (float)s2[rw] * b2;
}

// returns the refined corner in every lane
__device__ float2 sp_refine(const uint8_t* img, int pitch, int W, int H, float2 c0, int win, int max_iters, double eps2,
                            const float* __restrict__ mask, float* buf, int lane)
{
    const int WWm = 2 * win + 1, ww = WWm + 2;
    float2 cI = c0;
    int iter = 0;
    for (;;) {
        float cx = cI.x - (float)(ww - 1) * 0.5f, cy = cI.y - (float)(ww - 1) * 0.5f;
        const int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
        const float a = cx - (float)ipx, b = cy - (float)ipy;
        const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b, b1 = 1.f - b, b2 = b;
        for (int q = lane; q < ww * ww; q += 64) {
            const int i = q / ww, j = q - i * ww;
            buf[q] = sp_sample(img, pitch, W, H, ww, ww, ipx, ipy, a11, a12, a21, a22, b1, b2, i, j);
        }
        __builtin_amdgcn_wave_barrier();
        int stop = 0;
        float2 cN = cI;
        if (lane == 0) {
            double A = 0, B = 0, C = 0, bb1 = 0, bb2 = 0;
            const float* sp = buf + ww + 1;
            for (int i = 0, k = 0; i < WWm; i++, sp += ww) {
                const double py = i - win;
                for (int j = 0; j < WWm; j++, k++) {
                    const double m = mask[k];
                    const double tgx = sp[j + 1] - sp[j - 1];
                    const double tgy = sp[j + ww] - sp[j - ww];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - win;
                    A += gxx; B += gxy; C += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            const double det = A * C - B * B;
            if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) stop = 1; // DBL_EPSILON^2: the corner stays
            else {
                const double scale = 1.0 / det;
                cN.x = (float)(cI.x + C * scale * bb1 - B * scale * bb2);
                cN.y = (float)(cI.y - B * scale * bb1 + A * scale * bb2);
                const double err = (double)((cN.x - cI.x) * (cN.x - cI.x) + (cN.y - cI.y) * (cN.y - cI.y));
                if (cN.x < 0 || cN.x >= (float)W || cN.y < 0 || cN.y >= (float)H) stop = 1;
                else if (!(++iter < max_iters && err > eps2)) stop = 1;
            }
        }
        cI.x = __shfl(cN.x, 0); cI.y = __shfl(cN.y, 0);
        iter = __shfl(iter, 0);
        stop = __shfl(stop, 0);
        __builtin_amdgcn_wave_barrier();
        if (stop) break;
    }
    // too far from the start: poor convergence, the start is kept
    if (fabsf(cI.x - c0.x) > (float)win || fabsf(cI.y - c0.y) > (float)win) cI = c0;
    return cI;
}

#define SP_WAVES 4

// CORNER_SUBPIX: every corner of every output marker, on the full-resolution frame (markerdetector_impl.cpp:8430-8620)
__global__ __launch_bounds__(SP_WAVES * 64) void k_corner_subpix_markers(ImgView src, int W, int H, orbfe_marker* __restrict__ markers,
                                                                         const int32_t* __restrict__ n_out, int capacity, int win,
                                                                         int max_iters, double eps2, const float* __restrict__ mask)
{
    __shared__ float s_buf[SP_WAVES][SP_BUF];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, f = blockIdx.y;
    const int n = min(n_out[f], capacity);
    const uint8_t* img = src.base + (size_t)f * src.fstride;
    for (int q = blockIdx.x * SP_WAVES + wid; q < 4 * n; q += gridDim.x * SP_WAVES) {
        float* cp = &markers[(size_t)f * capacity + (q >> 2)].corners[q & 3][0];
        const float2 c = sp_refine(img, src.pitch, W, H, make_float2(cp[0], cp[1]), win, max_iters, eps2, mask, s_buf[wid], lane);
        if (lane == 0) { cp[0] = c.x; cp[1] = c.y; }
    }
}

// cornerUpsample (:14028-14220) on the rectangles the dictionary accepted, before they are sorted and deduplicated (the perimeter
// that decides between two markers of one id is the upsampled one): from pyramid level `start` down to the input, scale by the
// width ratio and refine with cornerSubPix(TermCriteria(MAX_ITER, 4, 0.5)), window int(0.5 + 2.5 * ratio).
__global__ __launch_bounds__(SP_WAVES * 64) void k_upsample_corners(ImgView src0, ImgView pyr, const ArLevel* __restrict__ levels, int start,
                                                                    int work_w, ArRect* __restrict__ rects, int rect_cap,
                                                                    const int32_t* __restrict__ cand_idx, const uint32_t* __restrict__ work,
                                                                    const int32_t* __restrict__ wctr, const int32_t* __restrict__ result,
                                                                    const float* __restrict__ masks /* window w at (w - 1) * 17 * 17 */)
{
    __shared__ float s_buf[SP_WAVES][SP_BUF];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nitems = wctr[0];
    for (int q = blockIdx.x * SP_WAVES + wid; q < 4 * nitems; q += gridDim.x * SP_WAVES) {
        const uint32_t wi = work[q >> 2];
        const int f = (int)(wi >> 16), slot = (int)(wi & 0xffffu);
        if (result[((size_t)f * rect_cap + slot) * 2] < 0) continue;
        float* cp = &rects[(size_t)f * rect_cap + cand_idx[(size_t)f * rect_cap + slot]].c[q & 3][0];
        float2 c = make_float2(cp[0], cp[1]);
        int prev_w = work_w;
        for (int l = start; l >= 0; l--) {
            const ArLevel L = levels[l];
            const float factor = __fdiv_rn((float)L.w, (float)prev_w);
            c.x = __fmul_rn(c.x, factor); c.y = __fmul_rn(c.y, factor);
            const int halfw = (int)(0.5 + 2.5 * (double)factor);
            const uint8_t* img = (l == 0) ? src0.base + (size_t)f * src0.fstride : pyr.base + (size_t)f * pyr.fstride + L.off;
            c = sp_refine(img, l == 0 ? src0.pitch : L.pitch, L.w, L.h, c, halfw, 4, 0.0, masks + (size_t)(halfw - 1) * 17 * 17, s_buf[wid], lane);
            prev_w = L.w;
        }
        if (lane == 0) { cp[0] = c.x; cp[1] = c.y; }
    }
}

} // namespace orbfe
