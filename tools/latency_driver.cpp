// The drop-in call sequence of one frame from C++, without Python: what the reference's Tracking thread would run with the shims in
// place (Frame.cc:91 ORBextractor::operator(), Frame.cc:142 MarkerDetector::detect with camera + marker size, Tracking.cc:610
// ORBmatcher::SearchForInitialization), through include/orbfe.h.  bench.py --latency compiles and runs it (g++, no HIP) and reports
// its medians next to the ones measured through the ctypes binding: the difference is the binding's marshalling, not the library.
//
//     latency_driver <frames.u8> <nframes> <rows> <cols> <calls> <nfeatures> <nlevels> <dictionary> <fx> <fy> <cx> <cy>
//
// prints one JSON object: medians in ms of the three calls and their sum, plain and with the detector paired to the extractor.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/orbfe.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s: %d (%s)\n", #call, rc_, orbfe_last_error()); return 1; } } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0 : 0.5 * (v[(v.size() - 1) / 2] + v[v.size() / 2]);
}

int main(int argc, char** argv)
{
    if (argc < 13) { fprintf(stderr, "usage: see the header of tools/latency_driver.cpp\n"); return 2; }
    const int nframes = atoi(argv[2]), rows = atoi(argv[3]), cols = atoi(argv[4]), calls = atoi(argv[5]), nfeatures = atoi(argv[6]), nlevels = atoi(argv[7]);
    const char* dict = argv[8];
    const float K4[4] = {(float)atof(argv[9]), (float)atof(argv[10]), (float)atof(argv[11]), (float)atof(argv[12])};
    const float dist[5] = {0.262383f, -0.953104f, -0.005358f, 0.002628f, 1.163314f}; // Examples/Monocular/TUM1.yaml
    std::vector<uint8_t> frames((size_t)nframes * rows * cols);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(frames.data(), 1, frames.size(), f) != frames.size()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);

    orbfe_extractor* ex = orbfe_extractor_create(nfeatures, 1.2f, nlevels, 20, 7, 0);
    orbfe_aruco* det = orbfe_aruco_create(dict, 0);
    if (!ex || !det) { fprintf(stderr, "create: %s\n", orbfe_last_error()); return 1; }
    const int cap = orbfe_extractor_max_keypoints(ex), mcap = orbfe_aruco_max_markers(det);
    std::vector<orbfe_keypoint> kps[2] = {std::vector<orbfe_keypoint>(cap), std::vector<orbfe_keypoint>(cap)};
    std::vector<uint8_t> desc[2] = {std::vector<uint8_t>((size_t)cap * 32), std::vector<uint8_t>((size_t)cap * 32)};
    std::vector<orbfe_marker> markers(mcap);
    std::vector<orbfe_marker_pose> poses(mcap);
    std::vector<float> prev((size_t)cap * 2);
    std::vector<int32_t> m12(cap);
    double med[2][4];
    long checksum = 0;
    for (int paired = 0; paired < 2; paired++) {
        CHECK(orbfe_extractor_pair_detector(ex, paired ? det : nullptr));
        std::vector<double> t_ex, t_det, t_sfi, t_all;
        int32_t n[2] = {0, 0};
        for (int i = 0; i < calls + 20; i++) {
            const uint8_t* img = frames.data() + (size_t)(i % nframes) * rows * cols;
            const int cur = i & 1, last = cur ^ 1;
            const double t0 = now_ms();
            CHECK(orbfe_extract(ex, img, rows, cols, cols, kps[cur].data(), desc[cur].data(), cap, &n[cur]));
            const double t1 = now_ms();
            int32_t nm = 0;
            CHECK(orbfe_aruco_detect_poses(det, img, rows, cols, cols, markers.data(), poses.data(), mcap, &nm, 0.187f, K4, dist, 5));
            const double t2 = now_ms();
            int32_t nmatches = 0;
            if (i > 0) {
                for (int k = 0; k < n[last]; k++) { prev[2 * k] = kps[last][k].x; prev[2 * k + 1] = kps[last][k].y; } // Initializer's mvbPrevMatched
                CHECK(orbfe_search_for_initialization(kps[last].data(), desc[last].data(), n[last], kps[cur].data(), desc[cur].data(), n[cur], cols, rows,
                                                      nullptr, prev.data(), m12.data(), 100, 0.9f, 1, &nmatches, 0));
            }
            const double t3 = now_ms();
            checksum += n[cur] + 1000 * nm + 1000000L * nmatches;
            if (i >= 20) { t_ex.push_back(t1 - t0); t_det.push_back(t2 - t1); t_sfi.push_back(t3 - t2); t_all.push_back(t3 - t0); }
        }
        med[paired][0] = median(t_ex); med[paired][1] = median(t_det); med[paired][2] = median(t_sfi); med[paired][3] = median(t_all);
    }
    CHECK(orbfe_extractor_pair_detector(ex, nullptr));
    orbfe_extractor_destroy(ex);
    orbfe_aruco_destroy(det);
    printf("{\"value\": %.6f, \"median_ms\": {\"orbfe_extract\": %.6f, \"orbfe_aruco_detect_poses\": %.6f, \"orbfe_search_for_initialization\": %.6f}, "
           "\"paired\": {\"value\": %.6f, \"median_ms\": {\"orbfe_extract\": %.6f, \"orbfe_aruco_detect_poses\": %.6f, \"orbfe_search_for_initialization\": %.6f}}, "
           "\"calls\": %d, \"checksum\": %ld}\n",
           med[0][3], med[0][0], med[0][1], med[0][2], med[1][3], med[1][0], med[1][1], med[1][2], calls, checksum);
    return 0;
}
