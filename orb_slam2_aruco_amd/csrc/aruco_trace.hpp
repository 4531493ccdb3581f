// aruco_trace.hpp -- border following on a bit-packed binary image, written once for the gfx950 kernels
// (aruco_kernels.hip) and for the host-side unit test of the same logic (tests/proto_contours.cpp).
//
// Reference behaviour being reproduced: cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) as called at
// Thirdparty/aruco/aruco/markerdetector_impl.cpp:3104.  OpenCV's implementation is a SEQUENTIAL raster scan
// (Suzuki-Abe) that marks visited pixels so that every border is followed exactly once, from its raster-first
// pixel.  Here every border is followed independently and read-only:
//
//   * a pixel can only be the raster-first pixel of an outer border if it is foreground, its W, NW, N, NE
//     neighbours are background ("local top");
//   * a background pixel can only be the raster-first pixel of a hole if its W and N neighbours are foreground;
//   * such a candidate is the real start iff following its border never meets a raster-smaller foreground pixel
//     (outer) / never examines a raster-smaller background pixel (hole); otherwise the walk is abandoned.
//
// The walk itself is icvFetchContour (OpenCV 3.4 contours.cpp) minus the marking.
#pragma once
#include <stdint.h>

#include "../../include/orbfe_math.h"

namespace orbfe {

// Bit image with a one-pixel zero frame: pixel (x, y) of the W x H image is bit (x+1) of row (y+1).
struct BitImage {
    const uint32_t* bits;
    int wpr; // 32-bit words per row
    int W, H;
    ORBFE_HD int get(int px, int py) const // padded coordinates, 0 <= px < W+2, 0 <= py < H+2
    {
        return (bits[py * wpr + (px >> 5)] >> (px & 31)) & 1;
    }
};

// direction codes 0..7 = E, NE, N, NW, W, SW, S, SE (OpenCV icvCodeDeltas)
ORBFE_HD int dir_dx(int s) { return (s == 0 || s == 1 || s == 7) ? 1 : (s == 3 || s == 4 || s == 5) ? -1 : 0; }
ORBFE_HD int dir_dy(int s) { return (s == 1 || s == 2 || s == 3) ? -1 : (s == 5 || s == 6 || s == 7) ? 1 : 0; }

// Follows one border.  (sx, sy): start pixel in PADDED coordinates; is_hole selects the hole start rule.
// For holes (hx, hy) = (sx+1, sy) is the hole's candidate raster-first background pixel.
// out (may be null): receives the points as (x | y << 16) in IMAGE coordinates, at most out_cap.
// Returns the number of border points, or -1 if the walk met a raster-smaller pixel (= not the canonical start),
// or -2 if max_steps was exceeded.
ORBFE_HD int trace_border(const BitImage& im, int sx, int sy, int is_hole, uint32_t* out, int out_cap, int max_steps)
{
    const int start_key = is_hole ? (sy * 65536 + sx + 1) : (sy * 65536 + sx);
    int s_end, s;
    s_end = s = is_hole ? 0 : 4;
    int i1x, i1y;
    do {
        s = (s - 1) & 7;
        i1x = sx + dir_dx(s);
        i1y = sy + dir_dy(s);
    } while (im.get(i1x, i1y) == 0 && s != s_end);
    if (s == s_end) { // single pixel domain
        if (out && out_cap > 0) out[0] = (uint32_t)(sx - 1) | ((uint32_t)(sy - 1) << 16);
        return 1;
    }
    int i3x = sx, i3y = sy, i4x = 0, i4y = 0, n = 0;
    for (;;) {
        s_end = s;
        while (s < 15) {
            ++s;
            i4x = i3x + dir_dx(s & 7);
            i4y = i3y + dir_dy(s & 7);
            if (im.get(i4x, i4y)) break;
            // an examined background pixel: for holes it belongs to the hole being followed (4-neighbours only)
            if (is_hole && !(s & 1) && (i4y * 65536 + i4x) < start_key) return -1;
        }
        s &= 7;
        if (!is_hole && (i3y * 65536 + i3x) < start_key) return -1;
        if (out && n < out_cap) out[n] = (uint32_t)(i3x - 1) | ((uint32_t)(i3y - 1) << 16);
        n++;
        if (n > max_steps) return -2;
        if (i4x == sx && i4y == sy && i3x == i1x && i3y == i1y) break;
        i3x = i4x;
        i3y = i4y;
        s = (s + 4) & 7;
    }
    return n;
}

// Candidate tests on the padded bit image (px, py in 1..W, 1..H).
ORBFE_HD bool outer_start_candidate(const BitImage& im, int px, int py)
{
    return im.get(px, py) && !im.get(px - 1, py) && !im.get(px - 1, py - 1) && !im.get(px, py - 1) &&
           !im.get(px + 1, py - 1);
}
// (px, py) is the BACKGROUND pixel; the border start is (px-1, py)
ORBFE_HD bool hole_start_candidate(const BitImage& im, int px, int py)
{
    return !im.get(px, py) && im.get(px - 1, py) && im.get(px, py - 1);
}

} // namespace orbfe
