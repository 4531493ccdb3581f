// Host-side harness for the border-following logic shared with the gfx950 kernel (aruco_trace.hpp).
// TEST INFRASTRUCTURE: lets the CPU test-suite compare "independent read-only traces from candidate starts"
// against the oracle's sequential Suzuki-Abe scan without a GPU.  Not part of the product library.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../orb_slam2_aruco_amd/csrc/aruco_trace.hpp"

using namespace orbfe;

extern "C" int proto_find_contours(const uint8_t* img, int w, int h, int32_t* lengths, int max_contours,
                                   int32_t* points, int max_points, int64_t* work_steps)
{
    const int wpr = (w + 2 + 31) / 32;
    std::vector<uint32_t> bits((size_t)wpr * (h + 2) + 2, 0); // + spare word for ring8's funnel read
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (img[(size_t)y * w + x]) bits[(size_t)(y + 1) * wpr + ((x + 1) >> 5)] |= 1u << ((x + 1) & 31);
    BitImage im{bits.data(), wpr, w, h};
    struct C { int key; std::vector<uint32_t> pts; };
    std::vector<C> found;
    std::vector<uint32_t> tmp((size_t)w * h * 4 + 16);
    int64_t steps = 0;
    for (int py = 1; py <= h; py++)
        for (int px = 1; px <= w; px++) {
            int is_hole = -1;
            if (outer_start_candidate(im, px, py)) is_hole = 0;
            else if (px >= 2 && hole_start_candidate(im, px, py)) is_hole = 1;
            if (is_hole < 0) continue;
            int n = trace_border(im, px - is_hole, py, is_hole, tmp.data(), (int)tmp.size(), 1 << 30);
            if (n < 0) continue;
            steps += n;
            found.push_back(C{py * 65536 + px, std::vector<uint32_t>(tmp.begin(), tmp.begin() + n)});
        }
    // discovery order = raster order of the transition pixel; output = reverse discovery order
    std::sort(found.begin(), found.end(), [](const C& a, const C& b) { return a.key > b.key; });
    int np = 0;
    for (int i = 0; i < (int)found.size(); i++) {
        if (i < max_contours) lengths[i] = (int)found[i].pts.size();
        for (uint32_t v : found[i].pts) {
            if (np < max_points) { points[2 * np] = (int)(v & 0xffff); points[2 * np + 1] = (int)(v >> 16); }
            np++;
        }
    }
    if (work_steps) *work_steps = steps;
    return (int)found.size();
}
