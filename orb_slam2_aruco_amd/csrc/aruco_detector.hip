// aruco_detector.hip -- host side of the ArUco marker detector behind the C ABI (include/orbfe.h).
//
// Mirrors aruco::MarkerDetector as configured by the reference at src/Frame.cc:129-142:
//   setDictionary(name), setDetectionMode(DM_NORMAL), setCornerRefinementMethod(CORNER_LINES), detect(gray).
// Effective parameters: adaptive threshold window max(3, 15*W/1920) made odd, C = 7, one
// threshold image, contours longer than 70 points, approxPolyDP eps 5 %, markerWarpPixSize 5, pyrfactor 2,
// borderDistThres 0.015, error correction off.  Pose estimation (step 12) is not part of this path yet.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <functional>
#include <limits>
#include <map>

#include "aruco_kernels.hpp"
#include "aruco_pose.hpp"
#include "orbfe_common.hpp"
#include "orbfe_tables.inc"

using namespace orbfe;

// What a run of the device pipeline does differently from the configuration of Frame.cc:135-137 (aruco_modes.hip)
struct ModeRun {
    const uint8_t* d_full = nullptr; // minSize > 0: the full-resolution frames (pyramid, warps, cornerUpsample); d_imgs is the reduction
    size_t full_fstride = 0, full_step = 0;
    int full_rows = 0, full_cols = 0;
    int fixed_thr = -1;              // THRES_AUTO_FIXED: the global threshold (-1: adaptive)
    uint32_t* d_hist = nullptr;      // THRES_AUTO_FIXED: per frame, 256 bins over the accepted candidates' patches
    std::function<int(hipStream_t)> before_finalize; // trackingMinDetections: the host looks at the decode results and may adopt rejected candidates
};

struct orbfe_aruco {
    int device = 0;
    std::string dict_name;
    int nbits = 0, nb = 0, S = 0, ncodes = 0;
    hipStream_t own_stream = nullptr, aux_stream = nullptr;
    hipStream_t user_aux = nullptr; // orbfe_aruco_set_aux_stream: run the pyramid there instead of on aux_stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int rows = 0, cols = 0, batch_cap = 0;
    int win = 0, wpr = 0, npyr = 0;
    uint32_t th_magic = 0; // multiply-high constant of the box mean (0: use the generic threshold kernel)
    std::vector<ArLevel> levels;
    std::vector<int> lvl_exact;           // 1 if level p is an exact 2x reduction of level p-1
    std::vector<size_t> tab_off;          // resize tables for non-exact levels
    size_t pyr_fbytes = 0, bits_fu32 = 0, candq_fu32 = 0, pool_fu32 = 0, gpad_fu32 = 0;
    int lds_bits_words = 0;
    DevBuf d_twork, d_trect, d_tctr; // k_tail_prep -> k_tail_approx -> k_tail_finish: work list, 4-gon flags, list length
    DevBuf d_rstate, d_lut; // k_contours_relay -> k_contours_small: per-frame grid shift and pool fill; the walks' step table
    DevBuf d_segs, d_tailkeys, d_tailoff, d_small, d_hint; // d_hint: the relay kernel's grid spacing of the previous batch
    int relay_kshift = 5;      // initial grid spacing (log2) of k_contours_relay
    PinnedBuf pinned; // staging of the host-pointer entry points
    DevBuf d_poses;   // orbfe_aruco_detect_poses
    // speculation for a paired extractor (orbfe_extractor_pair_detector; orbfe_common.hpp)
    struct Spec {
        bool pending = false, has_pose = false; // work enqueued and not consumed yet; poses were computed with `cam` / `size`
        int rows = 0, cols = 0;
        const uint8_t* host_copy = nullptr; // the extractor's staged frame (orbfe_common.hpp)
        size_t host_pitch = 0;
        PoseCamera cam{};
        float size = 0.f;
    } spec;
    bool last_cam_valid = false; // camera and marker size of the last detect-with-poses call: what the speculation assumes
    PoseCamera last_cam{};
    float last_size = 0.f;
    bool relay_wide = !(getenv("ORBFE_ARUCO_RELAY_WIDE") && !atoi(getenv("ORBFE_ARUCO_RELAY_WIDE")));
    DevBuf d_dwork, d_dctr, d_ditems, d_dhist, d_dpatch; // k_prefilter -> k_decode_warp / _otsu / _vote: the batch's candidates
    bool decode_dirty = false; // the decode work-list counter may be non-zero
    bool tail_dirty = false;   // the work-list counters may be non-zero (set while the tail's three launches are being enqueued)
    int relay_kcap = RL_KCAP;  // kept borders per frame the relay kernels and their tail hold
    // experiment (ORBFE_ARUCO_SMALL_SEPARATE=1): k_contours_small also for frames whose bit image is in LDS
    // phase (c) of LDS-resident frames as its own launch (k_contours_small): -1 = by batch size (a few frames leave most of the chip
    // idle, so the many small workgroups of the separate kernel shorten the call: 0.62 -> 0.57 ms for one 640 x 480 frame; a full
    // batch issues more instructions that way and the pipeline is bound by those: 1.85 -> 1.98 ms per C2 step), 0 / 1 = forced
    int small_separate_mode = getenv("ORBFE_ARUCO_SMALL_SEPARATE") ? atoi(getenv("ORBFE_ARUCO_SMALL_SEPARATE")) : -1;
    // the tiled relay formulation (aruco_tiles.hip: k_ct_band or k_ct_walk / k_ct_lists / k_ct_points).  -1 = by frame and batch size:
    // frames for which the one-workgroup relay kernel needs its 8192-slot table and a CU to itself (1280 x 720: 300-frame step 4.09 -
    // 4.13 -> 3.96 - 4.07 ms) or whose bit image does not fit LDS at all (1920 x 1080: 100-frame step 4.88 -> 3.41 ms), and batches of
    // up to 32 frames (a frame's walks spread over ~40 CUs instead of one); full batches of 640 x 480 frames keep the one-workgroup
    // kernel, whose single launch costs the pipeline less than band + lists + points (1.32 - 1.34 against 1.45 - 1.62 ms per step).
    // ORBFE_ARUCO_TILED = 0 / 1 forces it off / on for every batch (tests, A/B); ORBFE_ARUCO_TILE_W = tile width in pixels,
    // ORBFE_ARUCO_TPW = tiles per wave of k_ct_walk: measurement switches.
    int tiled = getenv("ORBFE_ARUCO_TILED") ? (atoi(getenv("ORBFE_ARUCO_TILED")) ? 1 : 0) : -1;
    bool tiled_off = false;    // set while a batch is redone by the relay kernels
    bool tiled_ran = false;    // the last batch took the tiled path
    int n_escalations = 0;     // batches done again on the next contour path (debug query 7: the tests assert 0 for ordinary frames)
    // the walks of the tiled path by BANDS of cell rows, a workgroup of eight waves each (k_ct_band), instead of a wave per tile
    // (k_ct_walk): -1 = by frame / batch size, 0 / 1 forced; ORBFE_ARUCO_BAND_ROWS = cell rows per band (0: what fits ~36 KB of LDS, at most 8)
    int banded = getenv("ORBFE_ARUCO_BANDED") ? (atoi(getenv("ORBFE_ARUCO_BANDED")) ? 1 : 0) : -1;
    int band_rows_env = getenv("ORBFE_ARUCO_BAND_ROWS") ? atoi(getenv("ORBFE_ARUCO_BAND_ROWS")) : 0;
    DevBuf d_ctmlist;
    int tile_w_env = getenv("ORBFE_ARUCO_TILE_W") ? atoi(getenv("ORBFE_ARUCO_TILE_W")) : 0;
    int tpw_env = getenv("ORBFE_ARUCO_TPW") ? atoi(getenv("ORBFE_ARUCO_TPW")) : 0;
    int ct_segcap = 0, ct_hbits = 0, ct_lcap = 0, ct_items_per_frame = 0;
    bool ct_dirty = true;      // the per-frame counters of the walk kernel may be non-zero (first use; a batch abandoned before k_ct_lists)
    unsigned ct_gen = 0;       // generation tag of the hash table's entries (16 bits; the table is cleared when it wraps and before first use)
    bool ct_tab_dirty = true;
    DevBuf d_ctseg, d_cthtab, d_ctelem, d_ctstate, d_ctitemsA, d_ctitemsB, d_ctnitems, d_ctcodes;
    // The speck passes (aruco_trace.hpp "FEWER WALKS" (2)): result-neutral, a quarter of the contour stage's start candidates and walks
    // gone -- and OFF by default, because neither way of running them pays in the pipeline:
    //   = 1: as a launch of their own between threshold and contours (k_speck_clean), for every contour path: the contour stage of
    //     300 x 640 x 480 alone 462 -> 408 us, but one more launch on the detector's chain costs the pipeline more than that (C2 step
    //     1.40 - 1.46 against 1.34 - 1.38 ms, single-frame detect 0.357 against 0.358 ms; profiles/r05_contour_reductions_ab.txt);
    //   = 2: inside the one-workgroup relay kernels, on the bit image they hold in LDS anyway (speck_pass_frame; batches of more than
    //     32 frames whose image fits LDS): 462 -> 440 us alone and 140 -> 120 us of VALU issue, the C2 step unchanged (1.343 against
    //     1.333 ms, four interleaved runs) -- and the rim masks and anchors of a frame (120 KB) go through scratch in HBM, which
    //     doubles the stage's HBM traffic (148 -> 268 MB per step).
    // ORBFE_ARUCO_SPECKS = 0 (default) / 1 / 2.  Debug codes 8 / 9: the launch on / off, 10 / 11: inside.  Tested either way
    // (tests/test_aruco_gpu.py, tests/test_stress_gpu.py).
    // the speck passes as a launch between threshold and contours: -1 = where they pay (full batches on the one-workgroup relay kernels:
    // 1.246 against 1.263 ms per C2 step with them, round 6; on the tiled paths 3.99 against 3.74 ms at 1280 x 720, 3.53 against 3.21 at
    // 1920 x 1080), 0 / 1 = never / wherever their tile fits LDS (ORBFE_ARUCO_SPECKS, debug codes 8 / 9)
    int specks = !getenv("ORBFE_ARUCO_SPECKS") ? -1 : atoi(getenv("ORBFE_ARUCO_SPECKS")) == 1 ? 1 : 0;
    bool specks_inkernel = getenv("ORBFE_ARUCO_SPECKS") && atoi(getenv("ORBFE_ARUCO_SPECKS")) == 2;
    bool half_pyr = true;   // the leading exact pyramid levels in one launch (k_half_pyr; debug code 18 / 19 = on / off)
    bool thr_mfma = true;   // k_threshold_mfma where it applies (windows up to 15; debug code 14 / 15)
    bool thr_mfma_auto = true; // ... but k_threshold_pyr for calls of fewer than 8 frames (debug code 14 forces the matrix-core kernel, 16 = this rule again)
    DevBuf d_tstrips, d_ttabs, d_ttab2;
    int n_tstrips = 0, ttab_rows = 0, ttab_cols = 0, ttab_win = 0, ttab_rb = 0;
    bool thr_mfma_ok = false;
    // Tables of k_threshold_mfma: per 32-column strip the pass-1 matrices (box K blocks a / b, selection a / b) in the B-operand layout of
    // v_mfma_i32_32x32x32_i8, BORDER_REPLICATE folded in; the pass-2 matrices (box over the previous / this block, centre x -WIN^2).
    int build_threshold_tables()
    {
        if (ttab_rows == rows && ttab_cols == cols && ttab_win == win) return ORBFE_OK;
        ttab_rows = rows; ttab_cols = cols; ttab_win = win;
        const int R = win / 2, n2 = win * win, W = cols, rb = R <= 3 ? 4 : 8;
        thr_mfma_ok = false;
        if (n2 > 240 || R > 7 || W < 48) return ORBFE_OK;
        std::vector<ThrStrip> st;
        std::vector<uint8_t> tabs;
        bool ok = true;
        for (int X = 0; X < W; X += 32) {
            ThrStrip S{};
            S.x0 = X; S.tab = (int)(tabs.size() / 1024);
            auto cl = [&](int c) { return std::min(std::max(c, 0), W - 16); };
            S.c0 = cl(X - rb); S.c1 = cl(X - rb + 16); S.c2 = cl(X - rb + 32);
            const int cs[3] = {S.c0, S.c1, S.c2};
            std::vector<int> Wb((size_t)W * 32, 0), Wi((size_t)W * 32, 0);
            for (int n = 0; n < 32; n++) {
                if (X + n >= W) continue;
                for (int u = -R; u <= R; u++) Wb[(size_t)std::min(std::max(X + n + u, 0), W - 1) * 32 + n] += 1;
                Wi[(size_t)(X + n) * 32 + n] = 1;
            }
            std::vector<int> owner((size_t)W, -1);
            for (int x = 0; x < W; x++)
                for (int pz = 0; pz < 3 && owner[x] < 0; pz++)
                    if (x >= cs[pz] && x < cs[pz] + 16) owner[x] = pz;
            for (int x = 0; x < W && ok; x++)
                for (int n = 0; n < 32; n++)
                    if (Wb[(size_t)x * 32 + n] && owner[x] < 0) ok = false;
            const size_t base = tabs.size();
            tabs.resize(base + 4096, 0);
            for (int m = 0; m < 4; m++)   // box a, box b, selection a, selection b
                for (int lane = 0; lane < 64; lane++) {
                    const int n = lane & 31, half = lane >> 5, ab = m & 1;
                    const int piece = ab == 0 ? half : (half == 0 ? 2 : -1);
                    if (piece < 0) continue;
                    for (int i = 0; i < 16; i++) {
                        const int x = cs[piece] + i;
                        if (x < 0 || x >= W || owner[x] != piece) continue;
                        tabs[base + (size_t)m * 1024 + (size_t)lane * 16 + i] = (uint8_t)(int8_t)((m < 2 ? Wb : Wi)[(size_t)x * 32 + n]);
                    }
                }
            st.push_back(S);
        }
        if (!ok || st.empty()) return ORBFE_OK;
        std::vector<uint8_t> t2(6144, 0);
        const int cw1 = n2 <= 127 ? n2 : 113, cw2 = n2 - cw1;   // the centre's weight -n2 in one signed byte, or in two
        for (int m = 0; m < 6; m++)   // box over the previous block, over this block, centre in the previous block, in this block (x 2)
            for (int lane = 0; lane < 64; lane++) {
                const int n = lane & 31, half = lane >> 5;
                for (int i = 0; i < 16; i++) {
                    const int q = 4 * half + (i & 3) + 8 * (i >> 2);
                    const int d = ((m & 1) ? 32 : 0) + q - rb - n;   // the row's offset from the output row
                    int v = 0;
                    if (m < 2) v = (d >= -R && d <= R) ? 1 : 0;
                    else v = d == 0 ? -(m < 4 ? cw1 : cw2) : 0;
                    t2[(size_t)m * 1024 + (size_t)lane * 16 + i] = (uint8_t)(int8_t)v;
                }
            }
        n_tstrips = (int)st.size();
        ttab_rb = rb;
        int rc;
        if ((rc = d_tstrips.ensure(st.size() * sizeof(ThrStrip))) || (rc = d_ttabs.ensure(tabs.size())) || (rc = d_ttab2.ensure(t2.size()))) return rc;
        ORBFE_HIP(hipMemcpy(d_tstrips.p, st.data(), st.size() * sizeof(ThrStrip), hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(d_ttabs.p, tabs.data(), tabs.size(), hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(d_ttab2.p, t2.data(), t2.size(), hipMemcpyHostToDevice));
        thr_mfma_ok = true;
        return ORBFE_OK;
    }
    bool thr_v2 = true;   // k_threshold_pyr where it applies (debug code 12 / 13: the tests run both threshold kernels)
    bool specks_ran = false;   // the last batch's contour kernels read d_bitsc
    DevBuf d_bitsc;
    bool relay_global = false; // k_contours_relay8g: the bit image stays in HBM (it does not fit LDS)
    int relay_tbits = 0;       // hash-table size of k_contours_relay (0: the kernel cannot run at this image size)
    bool force_legacy = false; // debug: always use k_contours_t
    bool big_mode = false;     // frames with more kept borders than the LDS-resident kernels hold: bit image in HBM, AR_MAX_KEPT_BIG
    DevBuf d_codes, d_levels, d_tabs, d_bits, d_pyr, d_candq, d_pool, d_kept, d_rects, d_counts, d_candidx, d_ncand,
        d_result, d_gpad;
    DevBuf d_in, d_out, d_nout;
    // MarkerDetector::Params the ABI exposes (markerdetector.h:96-214): error_correction_rate, cornerRefinementM
    float error_rate = 0.0f;
    int max_corr = 0, tau = 0, nsorted = 0; // int(tau * error_rate); the dictionary's code map in std::map order
    int corner_method = 1;                  // aruco::CornerRefinementMethod: 0 CORNER_SUBPIX, 1 CORNER_LINES, 2 CORNER_NONE
    DevBuf d_scodes, d_sids, d_msrc;        // d_msrc: rectangle (= contour) of every output marker of the last batch
    // MarkerDetector::Params outside the configuration of Frame.cc:135-137 (markerdetector.h:158-196; kernels in aruco_modes.hip)
    int detect_mode = 0;         // DM_NORMAL 0, DM_FAST 1, DM_VIDEO_FAST 2
    int thres_method = 0;        // THRES_ADAPTIVE 0, THRES_AUTO_FIXED 1
    int thres_value = 7;         // Params::ThresHold: the adaptive constant C, or the global threshold carried from frame to frame
    int n_attempts_auto_fix = 3; // Params::NAttemptsAutoThresFix
    float min_size = 0.f;        // Params::minSize as setDetectionMode / the automatic size estimation leave it
    bool auto_size = false;      // Params::autoSize (DM_VIDEO_FAST)
    float ts = 0.25f;
    bool enclosed = false;       // Params::enclosedMarker (detectEnclosedMarkers, markerdetector.h:126)
    // Params::trackingMinDetections (markerdetector.h:187; markerdetector_impl.cpp:7107-7890): a marker found in that many calls and
    // missing now is looked for among the candidates the dictionary rejected, inside its last outline
    int tracking_min = 0;
    std::map<int, int> marker_counts;         // id -> calls it was found in (one less per call without it)
    std::vector<orbfe_marker> prev_markers;   // what the previous call returned
    int last_tracked = 0;
    int gray_bits15 = 0;         // BGR2GRAY with 15 fractional bits (OpenCV 3.4.2+) instead of 14
    int pyr_rows = 0, pyr_cols = 0;   // the frame the /2 pyramid starts from (the working image is smaller when minSize > 0)
    int last_attempts = 0, last_work_rows = 0, last_work_cols = 0;
    size_t rl_static = 0;
    DevBuf d_red, d_mhist, d_masks, d_bgr, d_bits2;
    // A batch with a frame that exceeded a capacity of the contour path it ran on is done again on the next one: tiled -> (its
    // segment lists full: noise) the one-workgroup relay kernels, which coarsen their grid -> (kept borders / pool) the single-walker
    // kernel in big-frame mode.  The callers restore tiled_off / big_mode afterwards.
    bool escalate(int flags_or)
    {
        if (big_mode || force_legacy) return false;
        const bool was_tiled = tiled_ran;
        // from the tiled path any exceeded capacity (segment lists, kept borders, pool) goes to the one-workgroup relay kernels first:
        // they coarsen their grid and follow what is left whole, and get through frames of dense noise that neither the tiles nor the
        // single-walker kernel's per-lane arenas hold (480 x 640 with +-40 grey levels of noise: 203 kept borders, no flag)
        if (was_tiled && (flags_or & (2 | 4 | RL_FALLBACK_FLAGS)) && relay_tbits) { tiled_off = true; n_escalations++; return true; }
        if (flags_or & (2 | 4 | (was_tiled ? RL_FALLBACK_FLAGS : 0))) { big_mode = true; n_escalations++; return true; }
        return false;
    }
    bool stateful() const { return thres_method == 1 || auto_size || tracking_min > 0; } // a frame's result depends on the frames before it
    KernelTimer timer;
    int last_nframes = 0;

    ~orbfe_aruco()
    {
        for (DevBuf* b : {&d_codes, &d_levels, &d_tabs, &d_bits, &d_pyr, &d_candq, &d_pool, &d_kept, &d_rects,
                          &d_counts, &d_candidx, &d_ncand, &d_result, &d_gpad, &d_in, &d_out, &d_nout, &d_segs, &d_tailkeys, &d_tailoff, &d_small, &d_hint, &d_rstate, &d_lut, &d_twork, &d_trect, &d_tctr, &d_dwork, &d_dctr, &d_ditems, &d_dhist, &d_dpatch, &d_poses, &d_scodes, &d_sids,
                          &d_msrc, &d_red, &d_mhist, &d_masks, &d_bgr, &d_bits2, &d_ctseg, &d_cthtab, &d_ctelem, &d_ctstate, &d_ctitemsA, &d_ctitemsB, &d_ctnitems, &d_ctcodes, &d_ctmlist, &d_bitsc, &d_tstrips, &d_ttabs, &d_ttab2})
            b->release();
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (aux_stream) (void)hipStreamDestroy(aux_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
    }

    int set_dictionary(const char* name)
    {
        for (int i = 0; i < ORBFE_NDICTS; i++)
            if (!strcmp(ORBFE_DICTS[i].name, name)) {
                const orbfe_dict_entry& d = ORBFE_DICTS[i];
                if (d.nbits > 36) return fail(ORBFE_ERR_DICT, "dictionary %s: %d-bit codes are not supported", name, d.nbits);
                dict_name = name;
                nbits = d.nbits;
                nb = (int)std::sqrt((double)nbits);
                S = 5 * (nb + 2); // getMarkerWarpSize(): markerWarpPixSize * nSubdivisions (markerdetector_impl.cpp:1199-1292)
                ncodes = d.ncodes;
                int rc = d_codes.ensure((size_t)std::max(ncodes, 1) * 8);
                if (rc) return rc;
                if (ncodes) ORBFE_HIP(hipMemcpy(d_codes.p, d.codes, (size_t)ncodes * 8, hipMemcpyHostToDevice));
                // Dictionary::getMapCode() for the error-correction pass: std::map<uint64_t, uint16_t> = codes ascending, the FIRST id of
                // a duplicated code (map.insert does not overwrite, dictionary.cpp:99-103)
                std::vector<std::pair<unsigned long long, int>> m;
                for (int k = 0; k < ncodes; k++) m.push_back({d.codes[k], k});
                std::stable_sort(m.begin(), m.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
                std::vector<unsigned long long> sc;
                std::vector<int32_t> si;
                for (size_t k = 0; k < m.size(); k++)
                    if (k == 0 || m[k].first != m[k - 1].first) { sc.push_back(m[k].first); si.push_back(m[k].second); }
                nsorted = (int)sc.size();
                if ((rc = d_scodes.ensure((size_t)std::max(nsorted, 1) * 8)) || (rc = d_sids.ensure((size_t)std::max(nsorted, 1) * 4))) return rc;
                if (nsorted) {
                    ORBFE_HIP(hipMemcpy(d_scodes.p, sc.data(), (size_t)nsorted * 8, hipMemcpyHostToDevice));
                    ORBFE_HIP(hipMemcpy(d_sids.p, si.data(), (size_t)nsorted * 4, hipMemcpyHostToDevice));
                }
                tau = d.tau;
                max_corr = (int)((float)tau * error_rate); // dictionary_based.cpp: static_cast<int>(static_cast<float>(tau()) * rate)
                rows = cols = 0; // pyramid depth depends on S
                return ORBFE_OK;
            }
        // the reference treats an unknown name as a file path and throws (dictionary.cpp:44-62)
        return fail(ORBFE_ERR_DICT, "unknown dictionary '%s'", name);
    }

    // rows_ x cols_: the image that is thresholded and traced; prows x pcols: the frame the /2 pyramid is built from
    int build_geometry(int rows_, int cols_, int prows, int pcols)
    {
        if (rows_ == rows && cols_ == cols && prows == pyr_rows && pcols == pyr_cols && !levels.empty()) return ORBFE_OK;
        if (cols_ > 8000 || rows_ > 8000) return fail(ORBFE_ERR_INVALID, "image larger than 8000 px");
        int w = std::max(3, int(15 * float(cols_) / 1920.)); // :3765-3809
        if (w % 2 == 0) w++;
        if (w > 2 * TH_MAXR_HOST + 1) return fail(ORBFE_ERR_INVALID, "threshold window %d too large", w);
        win = w;
        // k_adaptive_threshold_t computes the mean as (s + n/2) * magic >> 32, n = win^2: use it only if that equals the
        // reference's rint(s * (1.0 / n)) for every possible box sum
        th_magic = 0;
        {
            const int n = w * w;
            const uint32_t mg = (uint32_t)((0x100000000ull + n - 1) / n);
            bool same = (n & 1) != 0;
            for (int sum = 0; same && sum <= 255 * n; sum++)
                same = (int)(((unsigned long long)(sum + n / 2) * mg) >> 32) == orbfe_round_d((double)sum * (1.0 / n));
            if (same) th_magic = mg;
        }
        wpr = (cols_ + 31) / 32;
        bits_fu32 = (size_t)wpr * rows_;
        // buildPyramid (:1299-1488): halve while width > 2 * S
        levels.clear();
        lvl_exact.clear();
        std::vector<int> tabs;
        tab_off.clear();
        int lw = pcols, lh = prows;
        size_t off = 0;
        levels.push_back(ArLevel{lw, lh, 0, 0});
        lvl_exact.push_back(1);
        tab_off.resize(4, 0);
        int n = 1, tw = pcols;
        while (tw > 2 * S) { tw /= 2; n++; }
        for (int p = 1; p < n; p++) {
            const int sw = lw, sh = lh;
            lw /= 2; lh /= 2;
            if (lw < 1 || lh < 1) break;
            ArLevel L{lw, lh, (lw + 63) / 64 * 64, (long long)off};
            off += (size_t)L.pitch * lh;
            levels.push_back(L);
            const bool exact = (sw == 2 * lw && sh == 2 * lh);
            lvl_exact.push_back(exact);
            tab_off.resize((size_t)(p + 1) * 4, 0);
            if (!exact) { // generic INTER_LINEAR tables, same format as the ORB pyramid's
                const double scale_x = 1. / ((double)lw / sw), scale_y = 1. / ((double)lh / sh);
                const int dwp = (lw + 3) / 4 * 4;
                std::vector<int> xofs(dwp), xal(dwp), yofs(lh), ybe(lh);
                for (int dx = 0; dx < lw; dx++) {
                    float fx = (float)((dx + 0.5) * scale_x - 0.5);
                    int sx = orbfe_floor_d(fx);
                    fx -= sx;
                    if (sx < 0) { fx = 0; sx = 0; }
                    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
                    const int a0 = (short)orbfe_round_f((1.f - fx) * 2048.f), a1 = (short)orbfe_round_f(fx * 2048.f);
                    xofs[dx] = sx;
                    xal[dx] = (a0 & 0xffff) | (a1 << 16);
                }
                for (int dx = lw; dx < dwp; dx++) { xofs[dx] = xofs[lw - 1]; xal[dx] = xal[lw - 1]; }
                for (int dy = 0; dy < lh; dy++) {
                    float fy = (float)((dy + 0.5) * scale_y - 0.5);
                    int sy = orbfe_floor_d(fy);
                    fy -= sy;
                    const int b0 = (short)orbfe_round_f((1.f - fy) * 2048.f), b1 = (short)orbfe_round_f(fy * 2048.f);
                    yofs[dy] = sy;
                    ybe[dy] = (b0 & 0xffff) | (b1 << 16);
                }
                tab_off[p * 4 + 0] = tabs.size(); tabs.insert(tabs.end(), xofs.begin(), xofs.end());
                tab_off[p * 4 + 1] = tabs.size(); tabs.insert(tabs.end(), xal.begin(), xal.end());
                tab_off[p * 4 + 2] = tabs.size(); tabs.insert(tabs.end(), yofs.begin(), yofs.end());
                tab_off[p * 4 + 3] = tabs.size(); tabs.insert(tabs.end(), ybe.begin(), ybe.end());
            }
        }
        npyr = (int)levels.size();
        pyr_fbytes = off + 64;
        // HBM overflow of the single-walker kernel's long-walk queue / the relay kernels' start-candidate queue, + their speck scratch
        // (the speck scratch -- four times the queue at 640 x 480 -- only where the in-kernel passes are switched on)
        candq_fu32 = ((size_t)relay_queue_words(cols_, rows_) + (specks_inkernel ? speck_frame_scratch_words(cols_, rows_) : 0) + 63) / 64 * 64;
        pool_fu32 = (size_t)CT_THREADS * std::max(4096, rows_ * cols_ / 48); // one private arena per lane of k_contours
        const int pw = (cols_ + 2 + 31) / 32;
        const size_t padded_words = (size_t)pw * (rows_ + 2) + 2; // + spare words for ring8()
        // the padded bit image goes to LDS when it fits next to the other arrays (160 KiB per workgroup)
        lds_bits_words = (contours_lds_bytes((int)padded_words, AR_MAX_KEPT) + 256 <= 160 * 1024) ? (int)padded_words : 0;
        gpad_fu32 = padded_words; // always there: big_mode uses the HBM variant at any size
        // k_contours_relay needs the bit image AND its marker table in LDS; otherwise k_contours_t does all frames.
        // 4096 marker slots on a 32-pixel grid for ordinary frames (a 2048-slot table on a 64-pixel grid would let two workgroups
        // share a CU, but its longer segments cost more than the sharing wins: 857 vs 726 us).  Large frames have more grid
        // crossings than 4096 slots hold and would be coarsened to a 128-pixel grid, which doubles the kernel's time (640 x 480:
        // 463 / 592 / 949 us at 32 / 64 / 128 pixels): they get 8192 slots (k_contours_relay8) when the LDS allows.
        relay_tbits = 0;
        // static LDS of the relay kernels (step table, counters), asked from the runtime: a constant here once fell behind the kernels
        // and frames whose tables only just fitted (1582 x 619: 156,976 B dynamic) failed at launch
        if (!rl_static) {
            const void* fns[4] = {reinterpret_cast<const void*>(k_contours_relay), reinterpret_cast<const void*>(k_contours_relay8),
                                  reinterpret_cast<const void*>(k_contours_relay8g), reinterpret_cast<const void*>(k_contours_relay_wide)};
            for (const void* fn : fns) {
                hipFuncAttributes fa{};
                ORBFE_HIP(hipFuncGetAttributes(&fa, fn));
                rl_static = std::max(rl_static, (size_t)fa.sharedSizeBytes);
            }
            rl_static += 256;
        }
        const bool large = (size_t)rows_ * cols_ > (size_t)640 * 480 * 3 / 2;
        if (lds_bits_words && large && relay_lds_bytes(lds_bits_words, RL_KCAP, 13) + rl_static <= 160 * 1024) { relay_tbits = 13; relay_kshift = 5; }
        else if (lds_bits_words && relay_lds_bytes(lds_bits_words, RL_KCAP, 12) + rl_static <= 160 * 1024) { relay_tbits = 12; relay_kshift = 5; }
        // frames whose bit image does not fit LDS: the relay formulation with the bit image in HBM (k_contours_relay8g)
        // (and room for as many kept borders as the single-walker kernel's big-frame mode: busy 1920 x 1080 frames have > 1024)
        // (also frames whose bit image fits LDS for the single-walker kernel but not next to a marker table)
        // (round 2: the HBM-image formulation also for frames that fit LDS -- 45 KB workgroups instead of 151 KB -- was slower, 5.50 against 4.62 ms at C3)
        relay_global = !relay_tbits && relay_lds_bytes(0, AR_MAX_KEPT_BIG, 13) + rl_static <= 160 * 1024;
        relay_kcap = RL_KCAP;
        if (relay_global) { relay_tbits = 13; relay_kshift = 5; relay_kcap = AR_MAX_KEPT_BIG; }
        // tiled path: segments per frame the lists hold (a 640 x 480 frame of the synthetic streams has ~2000, salt noise ~15 k; ids
        // are 16 bits), hash slots (twice that), segments whose list arrays k_ct_lists keeps in LDS (more: the same arrays in HBM)
        {
            int sc = 4096;
            while (sc < rows_ * cols_ / 32 && sc < 65536) sc <<= 1;
            ct_segcap = std::min(sc, 65535);
            ct_hbits = 1;
            while ((1 << ct_hbits) < 2 * sc) ct_hbits++;
            ct_lcap = std::min(ct_segcap, large ? 16384 : 4096);   // list elements k_ct_lists keeps in LDS (8 B each)
            if (getenv("ORBFE_ARUCO_LCAP") && atoi(getenv("ORBFE_ARUCO_LCAP")) > 0) ct_lcap = std::min(ct_segcap, atoi(getenv("ORBFE_ARUCO_LCAP"))); // (measurement switch)
            ct_items_per_frame = std::max(4096, ct_segcap / 4);
        }
        rows = rows_; cols = cols_;
        pyr_rows = prows; pyr_cols = pcols;
        batch_cap = 0;
        if (tabs.empty()) tabs.push_back(0);
        int rc;
        if ((rc = d_levels.ensure(levels.size() * sizeof(ArLevel))) || (rc = d_tabs.ensure(tabs.size() * 4))) return rc;
        ORBFE_HIP(hipMemcpy(d_levels.p, levels.data(), levels.size() * sizeof(ArLevel), hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(d_tabs.p, tabs.data(), tabs.size() * 4, hipMemcpyHostToDevice));
        return ORBFE_OK;
    }
    static const int TH_MAXR_HOST = 15; // windows up to 31 (k_adaptive_threshold<15>): frames up to 4095 pixels wide

    int ensure_workspace(int B)
    {
        if (B <= batch_cap) return ORBFE_OK;
        int rc;
        if ((rc = d_bits.ensure(bits_fu32 * 4 * B)) || (rc = d_pyr.ensure(pyr_fbytes * B)) ||   // (d_bitsc: when the speck launch runs)
            (rc = d_candq.ensure(candq_fu32 * 4 * B)) || (rc = d_pool.ensure(pool_fu32 * 4 * B)) ||
            (rc = d_kept.ensure((size_t)AR_MAX_KEPT_BIG * sizeof(ArKept) * B)) ||
            (rc = d_rects.ensure((size_t)AR_MAX_RECTS * sizeof(ArRect) * B)) || (rc = d_counts.ensure((size_t)16 * B)) ||
            (rc = d_candidx.ensure((size_t)AR_MAX_RECTS * 4 * B)) || (rc = d_ncand.ensure((size_t)4 * B)) ||
            (rc = d_result.ensure((size_t)AR_MAX_RECTS * 8 * B)) || (rc = d_msrc.ensure((size_t)AR_MAX_RECTS * 4 * B)) ||
            (rc = d_gpad.ensure(std::max<size_t>(gpad_fu32 * 4 * B, 16))) ||
            (rc = d_segs.ensure(std::max<size_t>(((size_t)sizeof(RelaySeg) << relay_tbits) * B, 16))) ||
            (rc = d_tailkeys.ensure((size_t)relay_kcap * 8 * B)) || (rc = d_tailoff.ensure((size_t)relay_kcap * 4 * B)) ||
            (rc = d_small.ensure((size_t)relay_kcap * 16 * B)) || (rc = d_rstate.ensure((size_t)8 * B)) ||
            (rc = d_twork.ensure((size_t)relay_kcap * 32 * B)) || (rc = d_trect.ensure((size_t)relay_kcap * B)))
            return rc;
        if (tiled != 0) {
            const size_t elem_words = (size_t)ct_segcap + ((size_t)ct_segcap + 3) / 4; // u64 elements + u16 next ids, per frame
            if ((rc = d_ctseg.ensure((size_t)5 * ct_segcap * 4 * B)) || (rc = d_cthtab.ensure(((size_t)8 << ct_hbits) * B)) ||
                (rc = d_ctelem.ensure(elem_words * 8 * B)) || (rc = d_ctstate.ensure((size_t)CT_STATE_INTS * 4 * B)) ||
                (rc = d_ctitemsA.ensure((size_t)ct_items_per_frame * 16 * B)) || (rc = d_ctitemsB.ensure((size_t)ct_items_per_frame * 8 * B)) ||
                (rc = d_ctnitems.ensure((size_t)4 * B)) || (rc = d_ctcodes.ensure((size_t)ct_segcap * CT_CODE_WORDS * 4 * B)) ||
                (rc = d_ctmlist.ensure((size_t)CTB_MCAP * 4 * ((rows + 31) / 32) * B)))   // (one list per band; at most one band per cell row)
                return rc;
            ct_dirty = true; ct_tab_dirty = true;
        }
        if (!d_hint.p) {
            if ((rc = d_hint.ensure(16))) return rc;
            ORBFE_HIP(hipMemset(d_hint.p, 0, 16));
        }
        {
            const size_t items = (size_t)B * AR_MAX_RECTS;
            if ((rc = d_dwork.ensure(items * 4)) || (rc = d_ditems.ensure(items * sizeof(DcItem))) || (rc = d_dhist.ensure(items * 512)) ||
                (rc = d_dpatch.ensure(items * DC_PATCH_BYTES)))
                return rc;
        }
        if (!d_dctr.p) {   // k_finalize leaves the counter at zero for the next batch
            if ((rc = d_dctr.ensure(16))) return rc;
            ORBFE_HIP(hipMemset(d_dctr.p, 0, 16));
        }
        if (!d_tctr.p) {   // k_tail_finish leaves the two counters at zero for the next batch
            if ((rc = d_tctr.ensure(16))) return rc;
            ORBFE_HIP(hipMemset(d_tctr.p, 0, 16));
        }
        if (!d_lut.p) {
            if ((rc = d_lut.ensure(2048 * 2))) return rc;
            hipLaunchKernelGGL(k_relay_lut, dim3(8), dim3(256), 0, 0, d_lut.as<uint16_t>());
            ORBFE_HIP(hipDeviceSynchronize());
        }
        batch_cap = B;
        return ORBFE_OK;
    }

    // window tables of cv::cornerSubPix for half sizes 1 .. 8: exp(-y^2) exp(-x^2) in float, by the host's expf like the reference
    int ensure_subpix_masks()
    {
        if (d_masks.p) return ORBFE_OK;
        std::vector<float> m((size_t)8 * 17 * 17, 0.f);
        for (int w = 1; w <= 8; w++) {
            const int ww = 2 * w + 1;
            float* mk = m.data() + (size_t)(w - 1) * 17 * 17;
            for (int i = 0; i < ww; i++) {
                const float y = (float)(i - w) / w;
                const float vy = std::exp(-y * y);
                for (int j = 0; j < ww; j++) {
                    const float x = (float)(j - w) / w;
                    mk[i * ww + j] = (float)(vy * std::exp(-x * x));
                }
            }
        }
        int rc = d_masks.ensure(m.size() * 4);
        if (rc) return rc;
        ORBFE_HIP(hipMemcpy(d_masks.p, m.data(), m.size() * 4, hipMemcpyHostToDevice));
        return ORBFE_OK;
    }

    int run_device(const uint8_t* d_imgs, int B, size_t frame_stride, int rows_, int cols_, size_t step,
                   orbfe_marker* d_out_m, int capacity, int32_t* d_n, hipStream_t s, const ModeRun* mr = nullptr)
    {
        int rc;
        const bool reduced = mr && mr->d_full;
        if ((rc = build_geometry(rows_, cols_, reduced ? mr->full_rows : rows_, reduced ? mr->full_cols : cols_))) return rc;
        if ((rc = ensure_workspace(B))) return rc;
        if ((reduced || corner_method == 0) && (rc = ensure_subpix_masks())) return rc;
        last_nframes = B;
        const ImgView srcW{d_imgs, nullptr, frame_stride, (int)step}; // what is thresholded and traced
        // the frame the pyramid starts from and the patches are warped from (level 0)
        const ImgView src0 = reduced ? ImgView{mr->d_full, nullptr, mr->full_fstride, (int)mr->full_step} : srcW;
        ImgView pyr{d_pyr.as<uint8_t>(), d_pyr.as<uint8_t>(), pyr_fbytes, 0};
        timer.begin();
        timer.mark(s, "start");
        // The threshold kernel of the batched configuration writes the pyramid levels its 64 x 64 tiles hold whole (k_threshold_pyr): the
        // exact halvings, at most four, when the pyramid starts from the thresholded frame itself and n v + K stays within 16 bits
        int nfuse = 0;
        uint32_t thr_kk = 0;
        bool use_thr_mfma = false;
        {
            const long n2 = (long)win * win, K = n2 * thres_value - n2 / 2;
            const bool adaptive = !(mr && mr->fixed_thr >= 0);
            const bool fused_ok = adaptive && thr_v2 && th_magic && (win == 5 || win == 7 || win == 11 || win == 15) && K >= 0 && n2 * 255 + K <= 65535;
            // A call of a few frames (the drop-in call: one) is a chain of launches that each wait for the one before: there the kernel that
            // also writes the pyramid (one launch instead of five) is the shorter chain -- detect 0.333 -> 0.303 ms per 640 x 480 frame;
            // a batch has the pyramid next to the contour kernels on a stream of its own and takes the matrix-core kernel
            if (adaptive && thr_mfma && th_magic && K > -(1 << 20) && K < (1 << 20) && !(fused_ok && !reduced && B < 8 && thr_mfma_auto)) {
                if ((rc = build_threshold_tables())) return rc;
                use_thr_mfma = thr_mfma_ok;
            }
            if (!use_thr_mfma && fused_ok) {
                thr_kk = (uint32_t)K | ((uint32_t)K << 16);
                nfuse = -1;   // the kernel applies, with no level so far
                if (!reduced)
                    for (int p = 1; p < npyr && p <= 4; p++) {
                        if (!lvl_exact[p] || levels[p].pitch % 4 != 0 || levels[p].pitch < 4 * ((levels[p].w + 3) / 4)) break;
                        nfuse = p;
                    }
            }
        }
        const bool thr_pyr = nfuse != 0;
        if (nfuse < 0) nfuse = 0;
        // the /2 pyramid is only needed by k_decode: it runs on a second stream next to threshold + contours (what the threshold kernel
        // leaves of it: behind that kernel, in line)
        hipStream_t aux_stream = user_aux ? user_aux : this->aux_stream;
        if (nfuse) aux_stream = s;
        else {
            ORBFE_HIP(hipEventRecord(ev_fork, s));
            ORBFE_HIP(hipStreamWaitEvent(aux_stream, ev_fork, 0));
        }
        bool finish_in_prefilter = false;   // k_tail_finish's work inside k_prefilter (set where the tail kernels are launched)
        timer.mark(aux_stream, "pyramid starts", true);
        auto rest_of_pyramid = [&](int first) -> int {
        if (first == 1 && half_pyr) {
            // the leading exact halvings in one launch (k_half_pyr): four from 16 x 16 source blocks, or three from 8 x 8
            for (int nf = 4; nf >= 3 && first == 1; nf--) {
                const int bs = 1 << nf;
                bool ok = npyr > nf && levels[0].w % bs == 0 && levels[0].h % bs == 0 && src0.pitch % (bs == 16 ? 16 : 8) == 0 &&
                          src0.fstride % (bs == 16 ? 16 : 8) == 0 && ((uintptr_t)src0.base & (bs == 16 ? 15 : 7)) == 0 && pyr_fbytes % 8 == 0;
                for (int p = 1; ok && p <= nf; p++) {
                    const int al = bs >> p;   // bytes a thread stores per row of level p
                    ok = lvl_exact[p] && levels[p].w == levels[0].w >> p && levels[p].h == levels[0].h >> p && levels[p].pitch % al == 0 &&
                         levels[p].off % al == 0 && levels[p].pitch >= levels[p].w;
                }
                if (!ok) continue;
                HalfPyrDst P{};
                P.base = pyr.base_w; P.fstride = pyr_fbytes;
                for (int p = 1; p <= nf; p++) { P.off[p - 1] = (uint32_t)levels[p].off; P.pitch[p - 1] = levels[p].pitch; }
                const int bw = levels[0].w / bs, nblocks = bw * (levels[0].h / bs);
                if (nf == 4) hipLaunchKernelGGL(k_half_pyr<4>, dim3((nblocks + 255) / 256, B), dim3(256), 0, aux_stream, src0, P, bw, nblocks);
                else hipLaunchKernelGGL(k_half_pyr<3>, dim3((nblocks + 255) / 256, B), dim3(256), 0, aux_stream, src0, P, bw, nblocks);
                first = nf + 1;
            }
        }
        for (int p = first; p < npyr; p++) {
            const ArLevel& L = levels[p];
            const ArLevel& Lp = levels[p - 1];
            ImgView sv = (p == 1) ? src0 : ImgView{pyr.base + Lp.off, nullptr, pyr_fbytes, Lp.pitch};
            ImgView dv{pyr.base + L.off, pyr.base_w + L.off, pyr_fbytes, L.pitch};
            if (lvl_exact[p] && sv.pitch % 8 == 0 && sv.fstride % 8 == 0 && ((uintptr_t)sv.base & 7) == 0 && dv.pitch % 4 == 0 && dv.pitch >= 4 * ((L.w + 3) / 4)) {
                const int dw4 = (L.w + 3) / 4, nthreads = dw4 * ((L.h + 1) / 2);   // reads up to 2 * L.w + 6 < the source pitch (64-byte rows)
                hipLaunchKernelGGL(k_half_area4, dim3((nthreads + 255) / 256, B), dim3(256), 0, aux_stream, sv, dv, dw4, L.h);
            } else if (lvl_exact[p]) {
                hipLaunchKernelGGL(k_half_area, dim3((L.w + 63) / 64, (L.h + 3) / 4, B), dim3(256), 0, aux_stream, sv, dv, L.w, L.h);
            } else {
                const int dw4 = (L.w + 3) / 4;
                const double scale_x = 1. / ((double)L.w / Lp.w), scale_y = 1. / ((double)L.h / Lp.h);
                hipLaunchKernelGGL(k_resize_level, dim3((dw4 + 63) / 64, (L.h + 7) / 8, B), dim3(256), 0, aux_stream, sv, dv,
                                   Lp.w, Lp.h, dw4, L.h, scale_x, scale_y, L.w);
            }
        }
        return ORBFE_OK;
        };
        if (!nfuse && (rc = rest_of_pyramid(1))) return rc;
        timer.mark(aux_stream, "pyramid");
        if (!nfuse) ORBFE_HIP(hipEventRecord(ev_join, aux_stream));
        int enlarge_k = win;   // detectEnclosedMarkers: the candidates grow by half the adaptive window, or half the erosion size
        for (int r_ = 0; r_ < ORBFE_REPS_ARUCO(8); r_++) {
            const dim3 tg((cols + 63) / 64, (rows + 63) / 64, B);
            const int ntx = (cols + 63) / 64, ntl = ntx * ((rows + 63) / 64);
            const dim3 tg1(xcd_grid(ntl * B));
            uint32_t* bp = d_bits.as<uint32_t>();
            if (mr && mr->fixed_thr >= 0) {   // THRES_AUTO_FIXED: cv::threshold(THRESH_BINARY_INV) at the carried-over threshold
                if (enclosed) {               // detectEnclosedMarkers: the inner edge band of the thresholded regions (erode + xor)
                    if ((rc = d_bits2.ensure(bits_fu32 * 4 * B))) return rc;
                    int k = int(std::max(3.0, 3. / 1920. * float(cols)));
                    if (k % 2 == 0) k++;
                    enlarge_k = k;
                    if (k / 2 > 15) return fail(ORBFE_ERR_INVALID, "detectEnclosedMarkers: frame too wide (erosion size %d)", k);
                    hipLaunchKernelGGL(k_fixed_threshold, dim3((wpr * rows + 255) / 256, B), dim3(256), 0, s, srcW, cols, rows, mr->fixed_thr, d_bits2.as<uint32_t>(), bits_fu32, wpr);
                    hipLaunchKernelGGL(k_erode_cross_xor, dim3((wpr * rows + 255) / 256, B), dim3(256), 0, s, d_bits2.as<uint32_t>(), bp, bits_fu32, wpr, cols, rows, k / 2);
                } else
                    hipLaunchKernelGGL(k_fixed_threshold, dim3((wpr * rows + 255) / 256, B), dim3(256), 0, s, srcW, cols, rows, mr->fixed_thr, bp, bits_fu32, wpr);
            } else if (use_thr_mfma) {
                const int n2 = win * win, nxs = (n_tstrips + 3) / 4;
                hipLaunchKernelGGL(k_threshold_mfma, dim3(xcd_grid(nxs * B)), dim3(256), 0, s, srcW, cols, rows, ttab_rb, -(n2 * thres_value - n2 / 2),
                                   d_tstrips.as<ThrStrip>(), d_ttabs.as<uint4>(), d_ttab2.as<uint4>(), bp, bits_fu32, wpr, n_tstrips, nxs, nxs * B, n2 > 127 ? 1 : 0);
            } else if (thr_pyr) {
                ThrPyr P{};
                P.n = nfuse;
                for (int p = 1; p <= nfuse; p++) { P.w[p - 1] = levels[p].w; P.h[p - 1] = levels[p].h; P.pitch[p - 1] = levels[p].pitch; P.off[p - 1] = levels[p].off; }
                if (win == 5) hipLaunchKernelGGL(k_threshold_pyr<5>, tg1, dim3(256), 0, s, srcW, cols, rows, thr_kk, bp, bits_fu32, wpr, ntx, ntl, ntl * B, pyr, P);
                else if (win == 7) hipLaunchKernelGGL(k_threshold_pyr<7>, tg1, dim3(256), 0, s, srcW, cols, rows, thr_kk, bp, bits_fu32, wpr, ntx, ntl, ntl * B, pyr, P);
                else if (win == 11) hipLaunchKernelGGL(k_threshold_pyr<11>, tg1, dim3(256), 0, s, srcW, cols, rows, thr_kk, bp, bits_fu32, wpr, ntx, ntl, ntl * B, pyr, P);
                else hipLaunchKernelGGL(k_threshold_pyr<15>, tg1, dim3(256), 0, s, srcW, cols, rows, thr_kk, bp, bits_fu32, wpr, ntx, ntl, ntl * B, pyr, P);
                if (nfuse && r_ == 0 && (rc = rest_of_pyramid(nfuse + 1))) return rc;
            } else if (th_magic && win == 5) hipLaunchKernelGGL(k_adaptive_threshold_t<5>, tg1, dim3(256), 0, s, srcW, cols, rows, thres_value, th_magic, bp, bits_fu32, wpr, ntx, ntl, ntl * B);
            else if (th_magic && win == 7) hipLaunchKernelGGL(k_adaptive_threshold_t<7>, tg1, dim3(256), 0, s, srcW, cols, rows, thres_value, th_magic, bp, bits_fu32, wpr, ntx, ntl, ntl * B);
            else if (th_magic && win == 11) hipLaunchKernelGGL(k_adaptive_threshold_t<11>, tg1, dim3(256), 0, s, srcW, cols, rows, thres_value, th_magic, bp, bits_fu32, wpr, ntx, ntl, ntl * B);
            else if (th_magic && win == 15) hipLaunchKernelGGL(k_adaptive_threshold_t<15>, tg1, dim3(256), 0, s, srcW, cols, rows, thres_value, th_magic, bp, bits_fu32, wpr, ntx, ntl, ntl * B);
            else if (win <= 15) hipLaunchKernelGGL(k_adaptive_threshold<7>, tg, dim3(256), 0, s, srcW, cols, rows, win, thres_value, 1.0 / (win * win), bp, bits_fu32, wpr);
            else hipLaunchKernelGGL(k_adaptive_threshold<15>, tg, dim3(256), 0, s, srcW, cols, rows, win, thres_value, 1.0 / (win * win), bp, bits_fu32, wpr);
        }
        timer.mark(s, "threshold");
        const bool big = big_mode || !lds_bits_words;
        const int legacy_kcap = big ? AR_MAX_KEPT_BIG : AR_MAX_KEPT, legacy_ldsw = big ? 0 : lds_bits_words;
        const size_t lds = contours_lds_bytes(legacy_ldsw, legacy_kcap);
        ORBFE_HIP(hipGetLastError());
        auto kfn = big ? k_contours_t<false> : k_contours_t<true>;
        { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(kfn), (size_t)(lds)); if (rc_lds_) return rc_lds_; }
        const bool use_tiled = (tiled > 0 || (tiled < 0 && (relay_global || !relay_tbits || relay_tbits > 12 || B <= 32))) && !tiled_off && !force_legacy && !big_mode;
        tiled_ran = use_tiled;
        const bool relay = (relay_tbits || use_tiled) && !force_legacy && !big_mode;
        // the bit image the contour kernels read: after the speck passes, unless switched off or the frame is too wide for their LDS tile
        const size_t spk_lds = speck_lds_bytes(cols);
        specks_ran = (specks > 0 || (specks < 0 && relay && !use_tiled && !relay_global && B > 32)) && spk_lds <= 150 * 1024;
        if (specks_ran) {
            if ((rc = d_bitsc.ensure(bits_fu32 * 4 * batch_cap))) return rc;
            { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_speck_clean), spk_lds); if (rc_lds_) return rc_lds_; }
            hipLaunchKernelGGL(k_speck_clean, dim3((rows + SPK_ROWS - 1) / SPK_ROWS, B), dim3(SPK_THREADS), spk_lds, s, d_bits.as<uint32_t>(), bits_fu32, wpr,
                               cols, rows, d_bitsc.as<uint32_t>());
        }
        const uint32_t* cbits = specks_ran ? d_bitsc.as<uint32_t>() : d_bits.as<uint32_t>();

        for (int r_ = 0; relay && r_ < ORBFE_REPS_ARUCO(1); r_++) {
            if (use_tiled) {
                // Tile width and waves.  k_ct_walk's waves are persistent and overlap their tiles, so a wave wants several tiles (its
                // lanes always find work) and a SIMD wants several waves (a step is a chain of dependent LDS reads): narrow tiles for a
                // batch -- ORBFE_ARUCO_TILE_W / ORBFE_ARUCO_TPW (tiles per wave) are measurement switches --, and for a few frames as many
                // waves as there are tiles.
                const int target = tile_w_env > 0 ? tile_w_env : (B <= 32 ? 192 : 480);
                const int ncols0 = std::max(1, (cols + target - 1) / target);
                const int cw = std::min(CTW_MAX_CW, std::max(32, ((cols + ncols0 - 1) / ncols0 + 31) / 32 * 32));
                const int ncols = (cols + cw - 1) / cw, nbands = (rows + 31) / 32;
                const int wave_bytes = ctw_wave_lds_bytes(cw), wlds = wave_bytes * (CTW_THREADS / 64);
                const int total_tiles = ncols * nbands * B;
                const int tpw = tpw_env > 0 ? tpw_env : (B <= 32 ? 1 : 2);
                const int walk_wgs = std::max(1, std::min((total_tiles / tpw + CTW_THREADS / 64 - 1) / (CTW_THREADS / 64), 256 * 8));
                const size_t llds = (size_t)ct_lcap * 8 + 16;
                { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_ct_walk), (size_t)wlds); if (rc_lds_) return rc_lds_; }
                { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_ct_lists), llds); if (rc_lds_) return rc_lds_; }
                if (ct_dirty) ORBFE_HIP(hipMemsetAsync(d_ctstate.p, 0, (size_t)CT_STATE_INTS * 4 * B, s)); // first use, or a batch abandoned before k_ct_lists (which leaves them at zero)
                ct_gen = (ct_gen + 1) & 0xffffu;
                if (ct_gen == 0) { ct_gen = 1; ct_tab_dirty = true; }
                if (ct_tab_dirty) { ORBFE_HIP(hipMemsetAsync(d_cthtab.p, 0, d_cthtab.bytes, s)); ct_tab_dirty = false; }
                ct_dirty = true;
                // bands for full batches (eight waves level each other's load through the band's ticket counters) and, one cell row each, for
                // up to four frames (single-frame call 0.385 -> 0.355 ms: the waves of a band share its start candidates, where a wave of
                // k_ct_walk has its tile's to itself); a wave per tile in between
                const bool use_band = banded > 0 || (banded < 0 && (B > 32 || B <= 4));
                if (use_band) {
                    const int pw = (cols + 2 + 31) / 32, crows = (rows + 31) / 32;
                    int rb = band_rows_env > 0 ? band_rows_env : B <= 4 ? 1 : std::max(1, std::min(8, (int)((36 * 1024 / (pw * 4) - 3) / 32)));
                    rb = std::max(1, std::min(rb, crows));
                    const int nb = (crows + rb - 1) / rb;
                    const size_t blds = ctb_lds_bytes(cols, rb);
                    { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_ct_band), blds); if (rc_lds_) return rc_lds_; }
                    hipLaunchKernelGGL(k_ct_band, dim3(nb, B), dim3(CTB_THREADS), blds, s, cbits, bits_fu32, wpr, cols, rows, 70,
                                       d_lut.as<uint16_t>(), rb, d_ctmlist.as<uint32_t>(), CTB_MCAP, d_cthtab.as<unsigned long long>(), ct_hbits, ct_gen,
                                       d_ctseg.as<uint32_t>(), (size_t)5 * ct_segcap, ct_segcap, d_ctstate.as<int32_t>(), d_pool.as<uint32_t>(), pool_fu32,
                                       (int)pool_fu32, relay_kcap, d_tailkeys.as<unsigned long long>(), d_tailoff.as<int32_t>(), d_ctcodes.as<uint4>());
                } else
                hipLaunchKernelGGL(k_ct_walk, dim3(walk_wgs), dim3(CTW_THREADS), wlds, s, cbits, bits_fu32, wpr, cols, rows, 70,
                                   d_lut.as<uint16_t>(), cw, ncols, nbands, total_tiles, d_cthtab.as<unsigned long long>(), ct_hbits, ct_gen,
                                   d_ctseg.as<uint32_t>(), (size_t)5 * ct_segcap, ct_segcap, d_ctstate.as<int32_t>(), d_pool.as<uint32_t>(), pool_fu32,
                                   (int)pool_fu32, relay_kcap, d_tailkeys.as<unsigned long long>(), d_tailoff.as<int32_t>(), wave_bytes,
                                   d_ctcodes.as<uint4>());
                hipLaunchKernelGGL(k_ct_lists, dim3(B), dim3(ct_lcap > 4096 ? 1024 : 512), llds, s, d_ctseg.as<uint32_t>(), (size_t)5 * ct_segcap, ct_segcap,
                                   d_ctstate.as<int32_t>(), d_cthtab.as<unsigned long long>(), ct_hbits, ct_gen, d_ctelem.as<unsigned long long>(), ct_lcap, 70,
                                   (int)pool_fu32, relay_kcap, d_tailkeys.as<unsigned long long>(), d_tailoff.as<int32_t>(), d_counts.as<int32_t>(),
                                   d_rstate.as<int32_t>(), d_ctitemsA.as<uint4>(), d_ctitemsB.as<uint2>(), ct_items_per_frame, d_ctnitems.as<int32_t>());
                if (hipPeekAtLastError() == hipSuccess) ct_dirty = false;
                hipLaunchKernelGGL(k_ct_points, dim3(ct_items_per_frame >= 8192 ? 32 : 8, B), dim3(256), 0, s, d_ctitemsA.as<uint4>(), d_ctitemsB.as<uint2>(),
                                   ct_items_per_frame, d_ctnitems.as<int32_t>(), d_ctcodes.as<uint32_t>(), ct_segcap, d_pool.as<uint32_t>(), pool_fu32);
            } else {
            const size_t rlds = relay_lds_bytes(relay_global ? 0 : lds_bits_words, relay_kcap, relay_tbits);
            // (round 2 launched the large-frame kernels, whose workgroups take a CU's whole LDS, in chunks of N frames: no gain at 128 / 192,
            // worse below -- profiles/r02_relay_chunks.txt; one launch since round 5)
            const bool small_separate = small_separate_mode < 0 ? B <= 32 : small_separate_mode != 0;
            const int chunk = B;
            for (int f0 = 0; f0 < B; f0 += chunk) {
            const int nb_ = std::min(chunk, B - f0);
            if (relay_global) {
                { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_contours_relay8g), (size_t)(rlds)); if (rc_lds_) return rc_lds_; }
                hipLaunchKernelGGL(k_contours_relay8g, dim3(nb_), dim3(RL_THREADS_BIG), rlds, s, cbits, bits_fu32, wpr,
                                   cols, rows, 0, 70, relay_kshift, relay_tbits, d_segs.as<RelaySeg>(),
                                   d_pool.as<uint32_t>(), pool_fu32, (int)pool_fu32, d_kept.as<ArKept>(), relay_kcap, relay_kcap,
                                   d_tailkeys.as<unsigned long long>(), d_tailoff.as<int32_t>(), d_counts.as<int32_t>(), d_hint.as<int32_t>(),
                                   d_small.as<uint4>(), d_rstate.as<int32_t>(), d_gpad.as<uint32_t>(), gpad_fu32, d_lut.as<uint16_t>(), f0);
            } else {
            const bool wide = relay_tbits <= 12 && B <= 32 && relay_wide;   // few frames: 16 waves per frame (see k_contours_relay_wide)
            auto rfn = relay_tbits > 12 ? k_contours_relay8 : wide ? k_contours_relay_wide : k_contours_relay;
            { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(rfn), (size_t)(rlds)); if (rc_lds_) return rc_lds_; }
            hipLaunchKernelGGL(rfn, dim3(nb_), dim3(relay_tbits > 12 || wide ? RL_THREADS_BIG : RL_THREADS), rlds, s, cbits, bits_fu32, wpr,
                               cols, rows, lds_bits_words, 70, relay_kshift, relay_tbits, d_segs.as<RelaySeg>(),
                               d_pool.as<uint32_t>(), pool_fu32, (int)pool_fu32, d_kept.as<ArKept>(), relay_kcap, relay_kcap,
                               d_tailkeys.as<unsigned long long>(), d_tailoff.as<int32_t>(), d_counts.as<int32_t>(), d_hint.as<int32_t>(),
                               d_small.as<uint4>(), d_rstate.as<int32_t>(), (small_separate ? 1 : 0) | (specks_inkernel && !specks_ran && !small_separate ? 2 : 0), d_lut.as<uint16_t>(), f0,
                               d_candq.as<uint32_t>(), candq_fu32);
            }
            }
            // the borders that touch no grid line, for frames done with a grid by a relay kernel that leaves them out (the
            // HBM-resident one: its bands fit LDS here; for LDS-resident frames the separate launch halves the relay kernel's time
            // but issues twice the instructions of the in-kernel phase -- measured 1.85 -> 1.98 ms per C2 step -- so those keep
            // phase (c) inside): bands of K rows, K >= 2^relay_kshift
            if (relay_global || small_separate) {
                const int nwaves = ((cols >> RS_BLOCK_SHIFT) + 1) * ((rows >> relay_kshift) + 1); // blocks of the finest grid
                hipLaunchKernelGGL(k_contours_small, dim3((nwaves + RS_THREADS / 64 - 1) / (RS_THREADS / 64), B), dim3(RS_THREADS), 0, s,
                                   cbits, bits_fu32, wpr, cols, rows, 70, d_lut.as<uint16_t>(), d_rstate.as<int32_t>(),
                                   d_pool.as<uint32_t>(), pool_fu32, (int)pool_fu32, relay_kcap, d_tailkeys.as<unsigned long long>(),
                                   d_tailoff.as<int32_t>(), d_counts.as<int32_t>());
            }
            } // (the relay kernels)
            // (g): sort + rank per frame, approxPolyDP by persistent waves over the whole batch's borders, rectangles per frame
            {
                const int pts = RT_PTS;   // LDS point buffer per wave; longer borders are read from the pool
                const size_t alds = tail_approx_lds_bytes(pts);
                const int tail_wgs = std::min(RT_WGS, B * 128);
                { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_tail_prep), tail_prep_lds_bytes(relay_kcap)); if (rc_lds_) return rc_lds_; }
                { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_tail_approx), alds); if (rc_lds_) return rc_lds_; }
                if (tail_dirty) ORBFE_HIP(hipMemsetAsync(d_tctr.p, 0, 16, s));   // a previous batch was abandoned between prep and finish
                tail_dirty = true;
                hipLaunchKernelGGL(k_tail_prep, dim3(B), dim3(relay_global ? 1024 : 256), tail_prep_lds_bytes(relay_kcap), s,
                                   d_tailkeys.as<unsigned long long>(), d_tailoff.as<int32_t>(), relay_kcap, d_counts.as<int32_t>(),
                                   d_twork.as<uint4>(), (size_t)relay_kcap * B, d_tctr.as<int32_t>());
                hipLaunchKernelGGL(k_tail_approx, dim3(tail_wgs), dim3(256), alds, s, relay_kcap, d_twork.as<uint4>(), (size_t)relay_kcap * B, d_tctr.as<int32_t>(),
                                   d_pool.as<uint32_t>(), pool_fu32, d_kept.as<ArKept>(), relay_kcap, d_trect.as<uint8_t>(), pts);
                // (the rectangle lists: a launch of their own only when the enclosed-marker pass sits between them and k_prefilter)
                finish_in_prefilter = !enclosed;
                if (!finish_in_prefilter) {
                    hipLaunchKernelGGL(k_tail_finish, dim3(B), dim3(64), 0, s, relay_kcap, d_trect.as<uint8_t>(), d_kept.as<ArKept>(), relay_kcap,
                                       d_rects.as<ArRect>(), AR_MAX_RECTS, d_counts.as<int32_t>(), d_tctr.as<int32_t>());
                    if (hipPeekAtLastError() == hipSuccess) tail_dirty = false;   // k_tail_finish leaves the list lengths at zero
                }
            }
        }
        // the single-walker kernel: images whose bit image does not fit LDS next to the relay kernel's tables, or forced
        if (!relay && !ORBFE_SKIP_ARUCO(1)) hipLaunchKernelGGL(kfn, dim3(B), dim3(CT_PROBE_THREADS), lds, s, cbits, bits_fu32, wpr, cols, rows,
                           legacy_ldsw, 70, d_candq.as<uint32_t>(), candq_fu32, (int)candq_fu32,
                           d_pool.as<uint32_t>(), pool_fu32, (int)pool_fu32, d_kept.as<ArKept>(), legacy_kcap,
                           d_rects.as<ArRect>(), AR_MAX_RECTS, d_counts.as<int32_t>(), d_gpad.as<uint32_t>(),
                           gpad_fu32, 0);
        timer.mark(s, "contours");
        ORBFE_HIP(hipGetLastError());
        if (enclosed)   // enlargeMarkerCandidate on every rectangle, before prefilterCandidates sees them (:3560-3590)
            hipLaunchKernelGGL(k_enlarge_candidates, dim3(B), dim3(AR_MAX_RECTS), 0, s, d_rects.as<ArRect>(), AR_MAX_RECTS, d_counts.as<int32_t>(),
                               (int)(float(enlarge_k) / 2.));
        if (decode_dirty) ORBFE_HIP(hipMemsetAsync(d_dctr.p, 0, 16, s));   // a previous batch was abandoned between prefilter and finalize
        decode_dirty = true;
        {
            TailFinish tf{};
            if (finish_in_prefilter) tf = TailFinish{relay_kcap, d_trect.as<uint8_t>(), d_kept.as<ArKept>(), relay_kcap, d_tctr.as<int32_t>()};
            hipLaunchKernelGGL(k_prefilter, dim3(B), dim3(256), 0, s, d_rects.as<ArRect>(), AR_MAX_RECTS,
                               d_counts.as<int32_t>(), cols, rows, win, d_candidx.as<int32_t>(), d_ncand.as<int32_t>(),
                               d_dwork.as<uint32_t>(), d_dctr.as<int32_t>(), tf);
            if (finish_in_prefilter && hipPeekAtLastError() == hipSuccess) tail_dirty = false;   // (it leaves the tail's list lengths at zero)
        }
        if (!nfuse) ORBFE_HIP(hipStreamWaitEvent(s, ev_join, 0));
        {
            // the batch's candidates as one work list: a wave per candidate (persistent: 32 candidates per frame is more than the
            // streams here produce, a busier batch loops), 64 candidates per Otsu wave
            const int max_items = B * AR_MAX_RECTS;
            const int wgs = std::max(1, std::min((B * 32 + DC_WAVES - 1) / DC_WAVES, 4096));
            for (int r_ = 0; r_ < ORBFE_REPS_ARUCO(2); r_++) {
                hipLaunchKernelGGL(k_decode_warp, dim3(wgs), dim3(DC_WAVES * 64), 0, s, src0, pyr, d_levels.as<ArLevel>(), npyr,
                                   d_rects.as<ArRect>(), AR_MAX_RECTS, d_candidx.as<int32_t>(), S, cols, d_dwork.as<uint32_t>(),
                                   d_dctr.as<int32_t>(), d_ditems.as<DcItem>(), d_dhist.as<uint16_t>(), d_dpatch.as<uint8_t>());
                hipLaunchKernelGGL(k_decode_otsu, dim3((max_items + 63) / 64), dim3(64), 0, s, d_dctr.as<int32_t>(), d_ditems.as<DcItem>(),
                                   d_dhist.as<uint16_t>(), S);
                hipLaunchKernelGGL(k_decode_vote, dim3(wgs), dim3(DC_WAVES * 64), 0, s, src0, pyr, d_levels.as<ArLevel>(), AR_MAX_RECTS, S, nb,
                                   d_codes.as<unsigned long long>(), ncodes, d_scodes.as<unsigned long long>(), d_sids.as<int32_t>(),
                                   nsorted, max_corr, d_dwork.as<uint32_t>(), d_dctr.as<int32_t>(), d_ditems.as<DcItem>(),
                                   d_dpatch.as<uint8_t>(), d_result.as<int32_t>());
            }
        }
        timer.mark(s, "decode");
        if (mr && mr->d_hist)   // the pixels of the accepted candidates: the next frame's threshold is Otsu over them
            hipLaunchKernelGGL(k_marker_hist, dim3(B), dim3(256), 0, s, d_dwork.as<uint32_t>(), d_dctr.as<int32_t>(), d_result.as<int32_t>(),
                               AR_MAX_RECTS, d_dhist.as<uint16_t>(), mr->d_hist);
        if (reduced) {   // cornerUpsample: before sort / dedupe, whose perimeters are those of the upsampled corners
            int start = 0;
            for (int i = 0; i < npyr; i++) {
                if (cols < levels[i].w) start = i;
                else break;
            }
            const int wgs = std::max(1, std::min((B * 32 * 4 + 3) / 4, 2048));
            hipLaunchKernelGGL(k_upsample_corners, dim3(wgs), dim3(256), 0, s, src0, pyr, d_levels.as<ArLevel>(), start, cols,
                               d_rects.as<ArRect>(), AR_MAX_RECTS, d_candidx.as<int32_t>(), d_dwork.as<uint32_t>(), d_dctr.as<int32_t>(),
                               d_result.as<int32_t>(), d_masks.as<float>());
        }
        if (mr && mr->before_finalize && (rc = mr->before_finalize(s))) return rc;
        // corner refinement applies only when the input was not reduced (:8420): CORNER_LINES inside k_finalize, CORNER_SUBPIX after it
        for (int r_ = 0; r_ < ORBFE_REPS_ARUCO(4); r_++) hipLaunchKernelGGL(k_finalize, dim3(B), dim3(256), 0, s, d_rects.as<ArRect>(), AR_MAX_RECTS,
                           d_candidx.as<int32_t>(), d_ncand.as<int32_t>(), d_result.as<int32_t>(),
                           d_pool.as<uint32_t>(), pool_fu32, d_out_m, capacity, d_n, (corner_method == 1 && !reduced) ? 1 : 0, d_msrc.as<int32_t>(),
                           d_dctr.as<int32_t>());
        if (corner_method == 0 && !reduced)   // cornerSubPix(grey, Size(4, 4), TermCriteria(MAX_ITER | EPS, 12, 0.005)) (:8511)
            hipLaunchKernelGGL(k_corner_subpix_markers, dim3(16, B), dim3(256), 0, s, src0, cols, rows, d_out_m, d_n, capacity, 4, 12,
                               0.005 * 0.005, d_masks.as<float>() + (size_t)3 * 17 * 17);
        timer.mark(s, "finalize");
        ORBFE_HIP(hipGetLastError());
        decode_dirty = false;   // k_finalize leaves the work-list length at zero
        return ORBFE_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Marker pose (reference detect :8720-8780 -> Marker::calculateExtrinsics, and Frame.cc:170): TWO lanes per marker record, one per IPPE
// solution -- both compute the homography and the two rotations (the same instructions on both lanes), then each its own translation,
// reprojection error (Rodrigues with a double sine and cosine, four distorted projections) and rotation vector; the pair exchanges
// seven floats.  The same operations on the same values as one lane doing both; the kernel is one serial chain of double arithmetic
// at the end of the detector's launches, and this takes 40 % off it.  32 markers per 64-thread workgroup.
__global__ __launch_bounds__(64) void k_marker_poses(const orbfe_marker* __restrict__ markers, const int32_t* __restrict__ d_n,
                                                    int capacity, float marker_size, PoseCamera cam,
                                                    orbfe_marker_pose* __restrict__ poses)
{
    const int f = blockIdx.y, lane = threadIdx.x, i = blockIdx.x * 32 + (lane >> 1), which = lane & 1;
    const int n = d_n ? min(d_n[f], capacity) : capacity;
    if (i >= n) return;   // (both lanes of a pair: the exchange below is between lanes of one pair)
    const orbfe_marker m = markers[(size_t)f * capacity + i];
    float e, r[3], t[3];
    pose::solve_marker_half(m.corners, marker_size, cam, which, &e, r, t);
    // the partner's solution
    const float eo = __shfl_xor(e, 1);
    float ro[3], to[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { ro[k] = __shfl_xor(r[k], 1); to[k] = __shfl_xor(t[k], 1); }
    if (which == 0) {
        const bool a_first = e < eo; // ippe.cpp:786 (this lane holds solution a)
        orbfe_marker_pose out;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            out.rvec[k] = a_first ? r[k] : ro[k]; out.tvec[k] = a_first ? t[k] : to[k];
            out.rvec2[k] = a_first ? ro[k] : r[k]; out.tvec2[k] = a_first ? to[k] : t[k];
        }
        out.err[0] = a_first ? e : eo;
        out.err[1] = a_first ? eo : e;
        poses[(size_t)f * capacity + i] = out;
    }
}

static int pose_camera(const float* K4, const float* dist, int ndist, float marker_size, PoseCamera& c, const char* who)
{
    if (!(marker_size > 0.0f)) return fail(ORBFE_ERR_INVALID, "%s: markerSize<=0: invalid markerSize", who); // marker.cpp:328-329
    if (!K4 || ndist < 0 || ndist > 12 || (ndist && !dist) || !(K4[0] != 0.0f) || !(K4[1] != 0.0f))
        return fail(ORBFE_ERR_INVALID, "%s: invalid camera (K = {fx, fy, cx, cy} with fx, fy != 0; at most 12 coefficients)", who);
    c.fx = K4[0]; c.fy = K4[1]; c.cx = K4[2]; c.cy = K4[3];
    for (int i = 0; i < 12; i++) c.k[i] = i < ndist ? (double)dist[i] : 0.0;
    c.has_dist = ndist > 0;
    return ORBFE_OK;
}

struct PoseWorkspace {
    hipStream_t stream = nullptr; // not the null stream: that one synchronises with every blocking stream of the process
    PinnedBuf pinned;
    ~PoseWorkspace() { if (stream) (void)hipStreamDestroy(stream); }
    DevBuf markers, poses;
};
static thread_local ThreadWorkspaces<PoseWorkspace> tl_pose_ws; // per (thread, device)

// ---- trackingMinDetections: host logic, as in the reference (it walks two short lists; the device supplies the candidates) ----
namespace {
struct TrackQuad { // marker_analyzer (markerdetector_impl.h:2460-2530): centre, area, inside test
    float c[4][2], cx, cy, area;
    void set(const float q[4][2])
    {
        memcpy(c, q, sizeof c);
        const float a1 = std::fabs((c[1][0] - c[0][0]) * (c[3][1] - c[0][1]) - (c[1][1] - c[0][1]) * (c[3][0] - c[0][0]));
        const float a2 = std::fabs((c[1][0] - c[2][0]) * (c[3][1] - c[2][1]) - (c[1][1] - c[2][1]) * (c[3][0] - c[2][0]));
        area = (a2 + a1) / 2.f;
        float sx = 0, sy = 0;
        for (int k = 0; k < 4; k++) { sx += c[k][0]; sy += c[k][1]; }
        cx = (float)(sx * (1. / 4.)); cy = (float)(sy * (1. / 4.));
    }
    bool is_into(float px, float py) const
    {
        for (int k = 0; k < 4; k++) {
            const float* p1 = c[k];
            const float* p2 = c[(k + 1) % 4];
            const float d = ((p1[1] - p2[1]) * px + (p2[0] - p1[0]) * py + (p1[0] * p2[1] - p2[0] * p1[1])) /
                            std::sqrt((p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1]));
            if (d < 0) return false;
        }
        return true;
    }
};
// agreement of the first sides' directions (:7700-7760)
float track_side_agreement(const float a[4][2], const float b[4][2])
{
    float ux = a[1][0] - a[0][0], uy = a[1][1] - a[0][1], vx = b[1][0] - b[0][0], vy = b[1][1] - b[0][1];
    const double nu = 1. / std::sqrt((double)ux * ux + (double)uy * uy), nv = 1. / std::sqrt((double)vx * vx + (double)vy * vy);
    ux = (float)(ux * nu); uy = (float)(uy * nu);
    vx = (float)(vx * nv); vy = (float)(vy * nv);
    return ux * vx + uy * vy;
}
} // namespace

// The tracking block of detect() on the decode results of one frame: `ids[slot]` (-1 = rejected) and the candidates' corners.  Adopted
// candidates get the missing marker's id and the corner rotation that best continues its previous orientation (written back as
// the (id, nRot) pair k_finalize understands).  Returns the number of adopted candidates.
static int track_missing_markers(orbfe_aruco* h, int ncand, int32_t* result /* ncand x (id, nRot) */, const float (*corners)[4][2])
{
    auto found = [&](int id) { for (int s = 0; s < ncand; s++) if (result[2 * s] == id) return true; return false; };
    for (auto& mc : h->marker_counts)
        if (!found(mc.first)) mc.second = std::max(mc.second - 1, 0);
    struct Info { TrackQuad q; int best = -1; double dist = std::numeric_limits<double>::max(); const orbfe_marker* prev = nullptr; };
    std::map<int, Info> need;
    for (const orbfe_marker& m : h->prev_markers)
        if (!found(m.id) && h->marker_counts.count(m.id) != 0 && h->marker_counts.at(m.id) >= h->tracking_min && !need.count(m.id)) {
            Info in; in.q.set(m.corners); in.prev = &m;
            need.insert({m.id, in});
        }
    struct Adopt { int slot, id, nrot; };
    std::vector<Adopt> adopt;
    if (!need.empty()) {
        for (int s = 0; s < ncand; s++) {
            if (result[2 * s] >= 0) continue; // only what the dictionary rejected
            TrackQuad qc; qc.set(corners[s]);
            for (auto& kv : need) {
                Info& in = kv.second;
                if (!in.q.is_into(qc.cx, qc.cy)) continue;
                const float dx = in.q.cx - qc.cx, dy = in.q.cy - qc.cy;
                const double dist = std::sqrt((double)dx * dx + (double)dy * dy);
                const float size_diff = std::fabs(in.q.area - qc.area) / in.q.area;
                if (size_diff < 0.3f && dist < in.dist) { in.best = s; in.dist = dist; }
            }
        }
        std::vector<char> used((size_t)std::max(ncand, 1), 0);
        for (auto& kv : need) {
            Info& in = kv.second;
            if (in.best == -1 || used[in.best]) continue; // (a candidate claimed twice: undefined in the reference; the first claim wins)
            int best_r = 0;
            double best_s = -1;
            for (int r = 0; r < 4; r++) {
                float rot[4][2];
                for (int k = 0; k < 4; k++) { rot[k][0] = corners[in.best][(k + r) % 4][0]; rot[k][1] = corners[in.best][(k + r) % 4][1]; }
                const float sc = track_side_agreement(in.prev->corners, rot);
                if (sc > best_s) { best_r = r; best_s = sc; }
            }
            used[in.best] = 1;
            // k_finalize rotates the corners by 4 - nRot: a left rotation by r is nRot = (4 - r) % 4
            adopt.push_back(Adopt{in.best, kv.first, (4 - best_r) % 4});
        }
        for (const Adopt& a : adopt) { result[2 * a.slot] = a.id; result[2 * a.slot + 1] = a.nrot; } // (after the loop: found() above saw the dictionary's results only)
    }
    const int adopted = (int)adopt.size();
    for (int s = 0; s < ncand; s++)
        if (result[2 * s] >= 0) {
            const int id = result[2 * s];
            if (h->marker_counts.count(id) == 0) h->marker_counts[id] = 1;
            else h->marker_counts[id]++;
        }
    return adopted;
}

// The image detect() works on (markerdetector_impl.cpp:5990-6090): with Params::minSize > 0 markers smaller than minSize * max(cols,
// rows) need not be found, so the frame is reduced until such a marker would be lowResMarkerSize = 20 pixels.
static int work_size(const orbfe_aruco* h, int rows, int cols, int* wr, int* wc)
{
    *wr = rows; *wc = cols;
    const int maxdim = std::max(cols, rows);
    const int minpix = (int)(static_cast<float>(h->min_size) * static_cast<float>(maxdim)); // getMinMarkerSizePix, minSize_pix = -1
    if (20 < minpix) {
        const float scale = float(20) / float(minpix);
        if (scale < 0.9) {
            int w = float(cols) * scale + 0.5, hh = float(rows) * scale + 0.5;
            if (w % 2 != 0) w++;
            if (hh % 2 != 0) hh++;
            // cornerUpsample's cornerSubPix window is int(0.5 + 2.5 * width ratio to the pyramid level above): tables up to 8
            if (w < 64 || hh < 48)
                return fail(ORBFE_ERR_INVALID, "minMarkerSize %g reduces a %d x %d frame to %d x %d: working images below 64 x 48 are not supported",
                            (double)h->min_size, cols, rows, w, hh);
            *wr = hh; *wc = w;
        }
    }
    return ORBFE_OK;
}

// a batch on a reduced working image: INTER_NEAREST into the handle's buffer, then the pipeline with the full frames for the
// pyramid, the warps and cornerUpsample
static int reduced_batch(orbfe_aruco* h, const uint8_t* d_imgs, int B, size_t frame_stride, int rows, int cols, size_t step, int wr, int wc,
                         orbfe_marker* d_out, int capacity, int32_t* d_n, hipStream_t s, int fixed_thr, uint32_t* d_hist,
                         std::function<int(hipStream_t)> before_finalize = nullptr)
{
    const size_t rpitch = (size_t)(wc + 63) / 64 * 64, rframe = rpitch * wr;
    int rc = h->d_red.ensure(rframe * B + 64);
    if (rc) return rc;
    const double ifx = 1. / ((double)wc / cols), ify = 1. / ((double)wr / rows);
    hipLaunchKernelGGL(k_resize_nearest, dim3((wc + 63) / 64, (wr + 3) / 4, B), dim3(256), 0, s, ImgView{d_imgs, nullptr, frame_stride, (int)step},
                       ImgView{h->d_red.as<uint8_t>(), h->d_red.as<uint8_t>(), rframe, (int)rpitch}, cols, rows, wc, wr, ifx, ify);
    ModeRun mr;
    mr.d_full = d_imgs; mr.full_fstride = frame_stride; mr.full_step = step; mr.full_rows = rows; mr.full_cols = cols;
    mr.fixed_thr = fixed_thr; mr.d_hist = d_hist; mr.before_finalize = before_finalize;
    return h->run_device(h->d_red.as<uint8_t>(), B, rframe, wr, wc, rpitch, d_out, capacity, d_n, s, &mr);
}

// The threshold of the next frame: Otsu's criterion over the normalised float histogram of the detected markers' pixels, every
// split evaluated from scratch in single precision as markerdetector_impl.cpp:6121-6380 does (host logic there as here: 2 x 256 x 255
// additions).  -1: no split has both classes above 1e-4 (an empty histogram turns into NaNs and ends here too).
static int otsu_of_marker_histogram(const uint32_t* counts)
{
    float hist[256], total = 0;
    for (int v = 0; v < 256; v++) { hist[v] = (float)counts[v]; total += hist[v]; } // counts < 2^24: exact, like the reference's ++
    const float inv = 1. / total;
    for (int v = 0; v < 256; v++) hist[v] *= inv;
    float best = 0;
    int best_t = -1;
    for (int t = 1; t < 256; t++) {
        float wlo = 0, whi = 0, mlo = 0, mhi = 0;
        for (int v = 0; v < t; v++) { wlo += hist[v]; mlo += float(v) * hist[v]; }
        for (int v = t; v < 256; v++) { whi += hist[v]; mhi += hist[v] * float(v); }
        if (!(wlo > 1e-4 && whi > 1e-4)) continue;
        mlo /= wlo;
        mhi /= whi;
        const float between = wlo * whi * (mlo - mhi) * (mlo - mhi);
        if (between > best) { best = between; best_t = t; }
    }
    return best_t;
}

extern "C" {

orbfe_aruco* orbfe_aruco_create(const char* dictionary, int device)
{
    if (!dictionary) { fail(ORBFE_ERR_INVALID, "null dictionary name"); return nullptr; }
    if (use_device(device) != ORBFE_OK) return nullptr;
    orbfe_aruco* h = new orbfe_aruco();
    h->device = device;
    if (hipStreamCreate(&h->own_stream) != hipSuccess ||
        hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        fail(ORBFE_ERR_HIP, "hipStreamCreate failed");
        delete h;
        return nullptr;
    }
    if (h->set_dictionary(dictionary) != ORBFE_OK) { delete h; return nullptr; }
    return h;
}

void orbfe_aruco_destroy(orbfe_aruco* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->spec.pending) (void)hipStreamSynchronize(h->own_stream); // work started for a paired extractor (unpair before destroying)
    delete h;
}

// A setter that changes what detect() returns ends a speculation started with the old parameters (orbfe_extractor_pair_detector:
// the extractor's call may have run this detector on its frame already): the detector's next call then runs by itself.
static void end_speculation(orbfe_aruco* h)
{
    if (h->spec.pending) { (void)hipStreamSynchronize(h->own_stream); h->spec.pending = false; }
}

int orbfe_aruco_set_dictionary(orbfe_aruco* h, const char* dictionary)
{
    if (!h || !dictionary) return fail(ORBFE_ERR_INVALID, "null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    end_speculation(h);
    return h->set_dictionary(dictionary);
}

int orbfe_aruco_max_markers(const orbfe_aruco* h) { return h ? AR_MAX_RECTS : ORBFE_ERR_INVALID; }

int orbfe_aruco_set_error_correction_rate(orbfe_aruco* h, float rate)
{
    if (!h || !(rate >= 0.0f && rate <= 1.0f)) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_set_error_correction_rate: rate must be in [0, 1]");
    end_speculation(h);
    h->error_rate = rate;
    h->max_corr = (int)((float)h->tau * rate);
    return ORBFE_OK;
}

int orbfe_aruco_set_detection_mode(orbfe_aruco* h, int mode, float min_marker_size)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    // Params::setDetectionMode (markerdetector.cpp:374-391).  DM_NORMAL = 0: adaptive threshold, C = 7 (Frame.cc:136); DM_FAST = 1:
    // THRES_AUTO_FIXED, the global threshold starts at 100; DM_VIDEO_FAST = 2: the same plus the automatic size estimation with
    // ts = 0.3.  minMarkerSize (Params::minSize, a fraction of the larger image side) > 0 makes detect() work on a reduced frame.
    if (mode < 0 || mode > 2) return fail(ORBFE_ERR_INVALID, "detection mode %d: DM_NORMAL 0, DM_FAST 1, DM_VIDEO_FAST 2", mode);
    if (!(min_marker_size >= 0.0f && min_marker_size <= 1.0f))
        return fail(ORBFE_ERR_INVALID, "minMarkerSize %g: a fraction of the image size in [0, 1]", (double)min_marker_size);
    end_speculation(h);
    h->detect_mode = mode;
    h->min_size = min_marker_size;
    if (mode == 0) { h->auto_size = false; h->ts = 0.25f; h->thres_method = 0; h->thres_value = 7; }
    else if (mode == 1) { h->auto_size = false; h->ts = 0.25f; h->thres_method = 1; h->thres_value = 100; }
    else { h->thres_method = 1; h->thres_value = 100; h->auto_size = true; h->ts = 0.3f; }
    return ORBFE_OK;
}

int orbfe_aruco_set_corner_refinement(orbfe_aruco* h, int method)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    // aruco::CornerRefinementMethod (markerdetector.h:62): CORNER_SUBPIX = 0 (cv::cornerSubPix), CORNER_LINES = 1, CORNER_NONE = 2.
    // Params::setCornerRefinementMethod (markerdetector.cpp:392-395): anything but CORNER_SUBPIX resets minSize to 0.
    if (method < 0 || method > 2) return fail(ORBFE_ERR_INVALID, "corner refinement method %d: CORNER_SUBPIX 0, CORNER_LINES 1, CORNER_NONE 2", method);
    end_speculation(h);
    h->corner_method = method;
    if (method != 0) h->min_size = 0.f;
    return ORBFE_OK;
}

int orbfe_aruco_set_enclosed_markers(orbfe_aruco* h, int on)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    end_speculation(h);
    h->enclosed = on != 0;
    return ORBFE_OK;
}

int orbfe_aruco_set_tracking(orbfe_aruco* h, int min_detections)
{
    if (!h || min_detections < 0) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_set_tracking: trackingMinDetections >= 0");
    end_speculation(h);
    h->tracking_min = min_detections;
    h->marker_counts.clear();
    h->prev_markers.clear();
    h->last_tracked = 0;
    return ORBFE_OK;
}

int orbfe_aruco_last_tracked(const orbfe_aruco* h) { return h ? h->last_tracked : ORBFE_ERR_INVALID; }

int orbfe_aruco_set_gray_conversion(orbfe_aruco* h, int fractional_bits)
{
    if (!h || (fractional_bits != 14 && fractional_bits != 15))
        return fail(ORBFE_ERR_INVALID, "orbfe_aruco_set_gray_conversion: 14 (OpenCV <= 3.4.1) or 15 (3.4.2 and later) fractional bits");
    end_speculation(h);
    h->gray_bits15 = fractional_bits == 15;
    return ORBFE_OK;
}

int orbfe_aruco_get_state(const orbfe_aruco* h, int32_t* threshold, float* min_size, int32_t* attempts, int32_t* work_rows, int32_t* work_cols)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    if (threshold) *threshold = h->thres_value;
    if (min_size) *min_size = h->min_size;
    if (attempts) *attempts = h->last_attempts;
    if (work_rows) *work_rows = h->last_work_rows;
    if (work_cols) *work_cols = h->last_work_cols;
    return ORBFE_OK;
}

int orbfe_aruco_marker_contour(orbfe_aruco* h, int frame, int marker, int32_t* xy, int capacity, int32_t* n)
{
    if (!h || !n || frame < 0 || frame >= h->last_nframes || marker < 0 || marker >= AR_MAX_RECTS || (capacity > 0 && !xy))
        return fail(ORBFE_ERR_INVALID, "orbfe_aruco_marker_contour: invalid argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    ORBFE_HIP(hipDeviceSynchronize());
    int32_t src = -1;
    ORBFE_HIP(hipMemcpy(&src, h->d_msrc.as<int32_t>() + (size_t)frame * AR_MAX_RECTS + marker, 4, hipMemcpyDeviceToHost));
    if (src < 0 || src >= AR_MAX_RECTS) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_marker_contour: no such marker in the last batch");
    ArRect r;
    ORBFE_HIP(hipMemcpy(&r, h->d_rects.as<ArRect>() + (size_t)frame * AR_MAX_RECTS + src, sizeof r, hipMemcpyDeviceToHost));
    *n = r.len;
    const int m = std::min(r.len, capacity);
    if (m > 0) {
        std::vector<uint32_t> p(m);
        ORBFE_HIP(hipMemcpy(p.data(), h->d_pool.as<uint32_t>() + (size_t)frame * h->pool_fu32 + r.off, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) { xy[2 * i] = (int32_t)(p[i] & 0xffff); xy[2 * i + 1] = (int32_t)(p[i] >> 16); }
    }
    return ORBFE_OK;
}

int orbfe_aruco_marker_contours(orbfe_aruco* h, int frame, int nmarkers, int32_t* xy, int capacity, int32_t* offsets)
{
    if (!h || !offsets || frame < 0 || frame >= h->last_nframes || nmarkers < 0 || nmarkers > AR_MAX_RECTS || (capacity > 0 && !xy))
        return fail(ORBFE_ERR_INVALID, "orbfe_aruco_marker_contours: invalid argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    offsets[0] = 0;
    if (nmarkers == 0) return ORBFE_OK;
    ORBFE_HIP(hipDeviceSynchronize());
    std::vector<int32_t> src((size_t)nmarkers);
    std::vector<ArRect> rects(AR_MAX_RECTS);
    ORBFE_HIP(hipMemcpy(src.data(), h->d_msrc.as<int32_t>() + (size_t)frame * AR_MAX_RECTS, (size_t)nmarkers * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(rects.data(), h->d_rects.as<ArRect>() + (size_t)frame * AR_MAX_RECTS, sizeof(ArRect) * AR_MAX_RECTS, hipMemcpyDeviceToHost));
    // the borders of a frame's markers lie in one pool: fetch the span that covers them once, then cut it up
    uint32_t lo = ~0u, hi = 0;
    for (int i = 0; i < nmarkers; i++) {
        if (src[i] < 0 || src[i] >= AR_MAX_RECTS) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_marker_contours: no marker %d in the last batch", i);
        const ArRect& r = rects[(size_t)src[i]];
        offsets[i + 1] = offsets[i] + r.len;
        if (r.len > 0) { lo = std::min(lo, (uint32_t)r.off); hi = std::max(hi, (uint32_t)(r.off + r.len)); }
    }
    if (offsets[nmarkers] > capacity || hi <= lo) return ORBFE_OK; // the caller reads the total from offsets and comes back with room
    std::vector<uint32_t> p((size_t)(hi - lo));
    ORBFE_HIP(hipMemcpy(p.data(), h->d_pool.as<uint32_t>() + (size_t)frame * h->pool_fu32 + lo, p.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < nmarkers; i++) {
        const ArRect& r = rects[(size_t)src[i]];
        int32_t* o = xy + 2 * (size_t)offsets[i];
        for (int k = 0; k < r.len; k++) { const uint32_t v = p[(size_t)(r.off - lo) + k]; o[2 * k] = (int32_t)(v & 0xffff); o[2 * k + 1] = (int32_t)(v >> 16); }
    }
    return ORBFE_OK;
}

int orbfe_aruco_detect_batch_device(orbfe_aruco* h, const uint8_t* d_imgs, int nframes, size_t frame_stride, int rows,
                                    int cols, size_t step, orbfe_marker* d_out, int capacity, int32_t* d_n_out,
                                    void* stream)
{
    if (!h || !d_imgs || !d_out || !d_n_out || nframes <= 0 || rows <= 0 || cols <= 0 || step < (size_t)cols ||
        capacity <= 0)
        return fail(ORBFE_ERR_INVALID, "orbfe_aruco_detect_batch_device: invalid argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    if (h->spec.pending) { ORBFE_HIP(hipStreamSynchronize(h->own_stream)); h->spec.pending = false; } // the handle's buffers are in use
    // THRES_AUTO_FIXED and the automatic size estimation carry state from one frame to the next and decide on the host (retry with
    // a random threshold): frames must go one at a time through the host-pointer entry points
    if (h->stateful())
        return fail(ORBFE_ERR_INVALID, "orbfe_aruco_detect_batch_device: DM_FAST / DM_VIDEO_FAST are frame-sequential (use orbfe_aruco_detect)");
    int wr, wc;
    if ((rc = work_size(h, rows, cols, &wr, &wc))) return rc;
    if (wc != cols) return reduced_batch(h, d_imgs, nframes, frame_stride, rows, cols, step, wr, wc, d_out, capacity, d_n_out, (hipStream_t)stream, -1, nullptr);
    return h->run_device(d_imgs, nframes, frame_stride, rows, cols, step, d_out, capacity, d_n_out, (hipStream_t)stream);
}

} // extern "C"

// staging layout of a one-frame host call (page-locked): [frame in] [n] [counts x 4] [marker records] [poses]
struct DetOffsets { size_t n, cnt, mk, ps, end; };
static DetOffsets det_offsets(size_t dframe, int nframes, bool with_pose)
{
    DetOffsets o;
    o.n = (dframe * nframes + 255) / 256 * 256;
    o.cnt = o.n + ((size_t)nframes * 4 + 63) / 64 * 64;
    o.mk = o.cnt + ((size_t)nframes * 16 + 63) / 64 * 64;
    o.ps = o.mk + (size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker);
    o.end = o.ps + (with_pose ? (size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker_pose) : 0);
    return o;
}

static bool same_camera(const PoseCamera& a, const PoseCamera& b) { return memcmp(&a, &b, sizeof(PoseCamera)) == 0; }

namespace orbfe {

// is `img` byte for byte the frame the paired extractor staged?
static bool same_frame(const uint8_t* img, size_t step, const uint8_t* copy, size_t pitch, int rows, int cols)
{
    if (!copy) return false;
    for (int y = 0; y < rows; y++)
        if (memcmp(img + (size_t)y * step, copy + (size_t)y * pitch, (size_t)cols) != 0) return false;
    return true;
}

int aruco_speculate(orbfe_aruco* h, const uint8_t* d_img, size_t dframe, int rows, int cols, size_t dpitch, hipEvent_t uploaded,
                    const uint8_t* host_copy, size_t host_pitch)
{
    h->spec.pending = false;
    if (h->big_mode) return ORBFE_OK; // the rare big-frame mode is left to the detector's own call
    if (h->stateful() || h->min_size > 0.f) return ORBFE_OK; // frame-sequential modes (aruco_modes.hip): the detector's own call too
    int rc;
    if ((rc = h->d_out.ensure((size_t)AR_MAX_RECTS * sizeof(orbfe_marker))) || (rc = h->d_nout.ensure(4))) return rc;
    const bool pose = h->last_cam_valid;
    const DetOffsets o = det_offsets(dframe, 1, true);
    if ((rc = h->pinned.ensure(o.end)) || (rc = h->d_poses.ensure((size_t)AR_MAX_RECTS * sizeof(orbfe_marker_pose)))) return rc;
    uint8_t* hp = h->pinned.as<uint8_t>();
    hipStream_t s = h->own_stream;
    ORBFE_HIP(hipStreamWaitEvent(s, uploaded, 0));
    if ((rc = h->run_device(d_img, 1, dframe, rows, cols, dpitch, h->d_out.as<orbfe_marker>(), AR_MAX_RECTS, h->d_nout.as<int32_t>(), s))) return rc;
    OutPack op;   // (one launch for the call's results: orbfe_common.hpp)
    op.add(hp + o.n, h->d_nout.p, 4);
    op.add(hp + o.cnt, h->d_counts.p, 16);
    op.add(hp + o.mk, h->d_out.p, (size_t)AR_MAX_RECTS * sizeof(orbfe_marker));
    if (pose) {
        hipLaunchKernelGGL(k_marker_poses, dim3((AR_MAX_RECTS + 31) / 32, 1), dim3(64), 0, s, h->d_out.as<orbfe_marker>(), h->d_nout.as<int32_t>(),
                           AR_MAX_RECTS, h->last_size, h->last_cam, h->d_poses.as<orbfe_marker_pose>());
        op.add(hp + o.ps, h->d_poses.p, (size_t)AR_MAX_RECTS * sizeof(orbfe_marker_pose));
    }
    if ((rc = op.flush<1>(s))) return rc;
    h->spec.pending = true; h->spec.rows = rows; h->spec.cols = cols; h->spec.host_copy = host_copy; h->spec.host_pitch = host_pitch;
    h->spec.has_pose = pose; h->spec.cam = h->last_cam; h->spec.size = h->last_size;
    return ORBFE_OK;
}

void aruco_speculation_wait(orbfe_aruco* h)
{
    if (h && h->spec.pending) { (void)hipStreamSynchronize(h->own_stream); h->spec.pending = false; }
}

void aruco_unpair_notice(orbfe_aruco* h)
{
    if (h) h->spec.pending = false;
}

int aruco_device_of(const orbfe_aruco* h) { return h ? h->device : -1; }


} // namespace orbfe

// The modes whose frames go one at a time (THRES_AUTO_FIXED and the automatic size estimation carry state from frame to frame and
// decide on the host; a reduced working image or a BGR frame just take this path too): upload, BGR -> grey, the pipeline -- again
// with a random threshold when nothing was found (markerdetector_impl.cpp:6903-6990; rand() is the process's own sequence, as in
// the reference) --, the next frame's threshold and minimum size (:7003-7040, :8790-8880), the poses.
static int detect_frames_modes(orbfe_aruco* h, const uint8_t* imgs, int nframes, size_t frame_stride, int rows, int cols, size_t step,
                               int channels, orbfe_marker* out, int capacity, int32_t* n_out, const PoseCamera* cam, float marker_size,
                               orbfe_marker_pose* poses_out)
{
    int rc;
    if (h->spec.pending) { ORBFE_HIP(hipStreamSynchronize(h->own_stream)); h->spec.pending = false; }
    const size_t dpitch = (size_t)(cols + 63) / 64 * 64, dframe = dpitch * rows;
    const size_t in_bytes = channels == 3 ? (size_t)cols * 3 * rows : dframe;
    // page-locked: [frame in] [n | counts | histogram | markers | poses] out
    const size_t o_n = (in_bytes + 255) / 256 * 256, o_cnt = o_n + 64, o_h = o_cnt + 64, o_mk = o_h + 1024,
                 o_ps = o_mk + (size_t)AR_MAX_RECTS * sizeof(orbfe_marker), o_tk = o_ps + (size_t)AR_MAX_RECTS * sizeof(orbfe_marker_pose),
                 o_tci = o_tk + 64, o_trc = o_tci + (size_t)AR_MAX_RECTS * 4, o_trs = o_trc + (size_t)AR_MAX_RECTS * sizeof(ArRect),
                 o_end = o_trs + (size_t)AR_MAX_RECTS * 8; // trackingMinDetections: candidate count, indices, rectangles, decode results
    if ((rc = h->pinned.ensure(o_end)) || (rc = h->d_in.ensure(dframe + 64)) || (rc = h->d_out.ensure((size_t)AR_MAX_RECTS * sizeof(orbfe_marker))) ||
        (rc = h->d_nout.ensure(4)) || (rc = h->d_mhist.ensure(1024)) || (cam && (rc = h->d_poses.ensure((size_t)AR_MAX_RECTS * sizeof(orbfe_marker_pose)))) ||
        (channels == 3 && (rc = h->d_bgr.ensure(in_bytes + 64))))
        return rc;
    uint8_t* hp = h->pinned.as<uint8_t>();
    hipStream_t s = h->own_stream;
    const int32_t* counts = reinterpret_cast<const int32_t*>(hp + o_cnt);
    const int32_t* np = reinterpret_cast<const int32_t*>(hp + o_n);
    const orbfe_marker* mk = reinterpret_cast<const orbfe_marker*>(hp + o_mk);
    const bool user_big_mode = h->big_mode;
    for (int f = 0; f < nframes; f++) {
        const uint8_t* img = imgs + (size_t)f * frame_stride;
        if (channels == 3) {
            for (int y = 0; y < rows; y++) memcpy(hp + (size_t)y * cols * 3, img + (size_t)y * step, (size_t)cols * 3);
            ORBFE_HIP(hipMemcpyAsync(h->d_bgr.p, hp, in_bytes, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_bgr_to_gray, dim3((cols + 63) / 64, (rows + 3) / 4, 1), dim3(256), 0, s, h->d_bgr.as<uint8_t>(), (size_t)0, (size_t)cols * 3,
                               ImgView{h->d_in.as<uint8_t>(), h->d_in.as<uint8_t>(), dframe, (int)dpitch}, cols, rows, h->gray_bits15);
        } else {
            for (int y = 0; y < rows; y++) memcpy(hp + (size_t)y * dpitch, img + (size_t)y * step, (size_t)cols);
            ORBFE_HIP(hipMemcpyAsync(h->d_in.p, hp, dframe, hipMemcpyHostToDevice, s));
        }
        int wr, wc;
        if ((rc = work_size(h, rows, cols, &wr, &wc))) return rc;
        h->last_work_rows = wr; h->last_work_cols = wc;
        int attempts = 0;
        h->last_attempts = 0;
        h->last_tracked = 0;
        // trackingMinDetections (:7107-7890): between the dictionary's verdicts and sort / dedupe the host looks at the frame's
        // candidates (2 KB of results, 10 KB of rectangles) and may hand a rejected one the id of a marker that has gone missing
        int pre_detected = -1;
        const std::map<int, int> counts_at_start = h->marker_counts;
        std::function<int(hipStream_t)> track_hook;
        if (h->tracking_min > 0)
            track_hook = [&](hipStream_t st) -> int {
                ORBFE_HIP(hipMemcpyAsync(hp + o_tk, h->d_ncand.p, 4, hipMemcpyDeviceToHost, st));
                ORBFE_HIP(hipMemcpyAsync(hp + o_tci, h->d_candidx.p, (size_t)AR_MAX_RECTS * 4, hipMemcpyDeviceToHost, st));
                ORBFE_HIP(hipMemcpyAsync(hp + o_trc, h->d_rects.p, (size_t)AR_MAX_RECTS * sizeof(ArRect), hipMemcpyDeviceToHost, st));
                ORBFE_HIP(hipMemcpyAsync(hp + o_trs, h->d_result.p, (size_t)AR_MAX_RECTS * 8, hipMemcpyDeviceToHost, st));
                ORBFE_HIP(hipStreamSynchronize(st));
                const int ncand = std::min(*reinterpret_cast<const int32_t*>(hp + o_tk), (int32_t)AR_MAX_RECTS);
                const int32_t* cidx = reinterpret_cast<const int32_t*>(hp + o_tci);
                const ArRect* rc_ = reinterpret_cast<const ArRect*>(hp + o_trc);
                int32_t* res = reinterpret_cast<int32_t*>(hp + o_trs);
                pre_detected = 0;
                for (int q = 0; q < ncand; q++) pre_detected += res[2 * q] >= 0;
                // a pass that will be repeated with another threshold: the tracking block runs once, after the last pass
                if (pre_detected == 0 && h->thres_method == 1 && attempts + 1 < h->n_attempts_auto_fix) return ORBFE_OK;
                h->marker_counts = counts_at_start;
                std::vector<std::array<std::array<float, 2>, 4>> cs((size_t)std::max(ncand, 1));
                for (int q = 0; q < ncand; q++) memcpy(&cs[q], rc_[cidx[q]].c, sizeof(float) * 8);
                h->last_tracked = track_missing_markers(h, ncand, res, reinterpret_cast<const float (*)[4][2]>(cs.data()));
                if (h->last_tracked) ORBFE_HIP(hipMemcpyAsync(h->d_result.p, res, (size_t)AR_MAX_RECTS * 8, hipMemcpyHostToDevice, st));
                return ORBFE_OK;
            };
        for (;;) {
            h->last_attempts++;
            const int thr = h->thres_method == 1 ? h->thres_value : -1;
            uint32_t* d_hist = h->thres_method == 1 ? h->d_mhist.as<uint32_t>() : nullptr;
            for (int pass = 0; pass < 3; pass++) { // a frame that exceeds the capacities of the contour path it ran on is done again on the next one (escalate())
                if (wc != cols) rc = reduced_batch(h, h->d_in.as<uint8_t>(), 1, dframe, rows, cols, dpitch, wr, wc, h->d_out.as<orbfe_marker>(), AR_MAX_RECTS,
                                                   h->d_nout.as<int32_t>(), s, thr, d_hist, track_hook);
                else {
                    ModeRun mr;
                    mr.fixed_thr = thr; mr.d_hist = d_hist; mr.before_finalize = track_hook;
                    rc = h->run_device(h->d_in.as<uint8_t>(), 1, dframe, rows, cols, dpitch, h->d_out.as<orbfe_marker>(), AR_MAX_RECTS,
                                       h->d_nout.as<int32_t>(), s, &mr);
                }
                if (rc) { h->big_mode = user_big_mode; h->tiled_off = false; return rc; }
                ORBFE_HIP(hipMemcpyAsync(hp + o_n, h->d_nout.p, 4, hipMemcpyDeviceToHost, s));
                ORBFE_HIP(hipMemcpyAsync(hp + o_cnt, h->d_counts.p, 16, hipMemcpyDeviceToHost, s));
                if (d_hist) ORBFE_HIP(hipMemcpyAsync(hp + o_h, d_hist, 1024, hipMemcpyDeviceToHost, s));
                ORBFE_HIP(hipMemcpyAsync(hp + o_mk, h->d_out.p, (size_t)AR_MAX_RECTS * sizeof(orbfe_marker), hipMemcpyDeviceToHost, s));
                ORBFE_HIP(hipStreamSynchronize(s));
                if (!h->escalate(counts[2])) break;
            }
            h->big_mode = user_big_mode; h->tiled_off = false;
            if (counts[2]) return fail(ORBFE_ERR_CAPACITY, "frame %d: internal detector capacity exceeded (flags 0x%x)", f, counts[2]);
            // (the retry is decided on what the dictionary found, before the tracking block adds anything: :6903)
            if ((h->tracking_min > 0 ? pre_detected : np[0]) == 0 && h->thres_method == 1 && ++attempts < h->n_attempts_auto_fix) {
                h->thres_value = 10 + rand() % 230;
                continue;
            }
            break;
        }
        const int n = np[0];
        if (h->thres_method == 1) {
            const int t = otsu_of_marker_histogram(reinterpret_cast<const uint32_t*>(hp + o_h));
            if (t > 0) h->thres_value = t;
        }
        // the smallest marker of this frame sets the minimum size the next frame looks for (:8790-8880)
        float shortest = std::numeric_limits<float>::max();
        for (int i = 0; i < n && i < AR_MAX_RECTS; i++) {
            float per = 0;
            for (int c = 0; c < 4; c++) {
                const float dx = mk[i].corners[c][0] - mk[i].corners[(c + 1) % 4][0], dy = mk[i].corners[c][1] - mk[i].corners[(c + 1) % 4][1];
                per += std::sqrt((double)dx * dx + (double)dy * dy);
            }
            if (shortest > per) shortest = per;
        }
        const float marker_min = shortest != std::numeric_limits<float>::max() ? shortest / (4 * std::max(cols, rows)) : 0.f;
        if (h->auto_size) h->min_size = marker_min * (1 - h->ts);
        if (h->tracking_min > 0) h->prev_markers.assign(mk, mk + std::min(n, (int)AR_MAX_RECTS));
        n_out[f] = n;
        if (n > capacity) return fail(ORBFE_ERR_CAPACITY, "frame %d has %d markers, capacity is %d", f, n, capacity);
        if (n) memcpy(out + (size_t)f * capacity, mk, (size_t)n * sizeof(orbfe_marker));
        if (n && cam) {
            hipLaunchKernelGGL(k_marker_poses, dim3((AR_MAX_RECTS + 31) / 32, 1), dim3(64), 0, s, h->d_out.as<orbfe_marker>(), h->d_nout.as<int32_t>(),
                               AR_MAX_RECTS, marker_size, *cam, h->d_poses.as<orbfe_marker_pose>());
            ORBFE_HIP(hipMemcpyAsync(hp + o_ps, h->d_poses.p, (size_t)n * sizeof(orbfe_marker_pose), hipMemcpyDeviceToHost, s));
            ORBFE_HIP(hipStreamSynchronize(s));
            memcpy(poses_out + (size_t)f * capacity, hp + o_ps, (size_t)n * sizeof(orbfe_marker_pose));
        }
    }
    return ORBFE_OK;
}

// detect (+ the IPPE pose of every marker when a camera is given: one call, one wait, instead of a detect call and a pose call)
static int detect_batch_impl(orbfe_aruco* h, const uint8_t* imgs, int nframes, size_t frame_stride, int rows, int cols,
                             size_t step, orbfe_marker* out, int capacity, int32_t* n_out, const PoseCamera* cam, float marker_size,
                             orbfe_marker_pose* poses_out, int channels = 1)
{
    if (!h || !n_out) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_detect_batch: null argument");
    if (!imgs || rows <= 0 || cols <= 0 || nframes <= 0) {
        for (int f = 0; f < nframes; f++) n_out[f] = 0;
        return ORBFE_OK;
    }
    if (!out || step < (size_t)cols * channels || capacity <= 0) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_detect_batch: invalid argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    {
        int wr, wc;
        if ((rc = work_size(h, rows, cols, &wr, &wc))) return rc;
        if (h->stateful() || wc != cols || channels == 3)
            return detect_frames_modes(h, imgs, nframes, frame_stride, rows, cols, step, channels, out, capacity, n_out, cam, marker_size, poses_out);
        h->last_attempts = 1; h->last_work_rows = rows; h->last_work_cols = cols;
    }
    const size_t dpitch = (size_t)(cols + 63) / 64 * 64, dframe = dpitch * rows;
    if (cam) { h->last_cam = *cam; h->last_size = marker_size; h->last_cam_valid = true; } // what a paired extractor's speculation assumes
    // A paired extractor has started this detector on the image it was given (aruco_speculate): if this call is handed the same
    // image, the work is done or under way on the device -- wait for it and take the results (and the poses, when the camera is the
    // one of the last call); anything else -- another image, a capacity flag -- and the call runs as if nothing had happened.
    if (h->spec.pending && nframes != 1) { ORBFE_HIP(hipStreamSynchronize(h->own_stream)); h->spec.pending = false; }
    if (h->spec.pending && nframes == 1) {
        h->spec.pending = false;
        const bool match = h->spec.rows == rows && h->spec.cols == cols && same_frame(imgs, step, h->spec.host_copy, h->spec.host_pitch, rows, cols);
        hipStream_t s = h->own_stream;
        if (!match) ORBFE_HIP(hipStreamSynchronize(s)); // its buffers are about to be reused
        else {
            const DetOffsets o = det_offsets(dframe, 1, true);
            uint8_t* hp = h->pinned.as<uint8_t>();
            const bool pose_ready = cam && h->spec.has_pose && h->spec.size == marker_size && same_camera(h->spec.cam, *cam);
            if (cam && !pose_ready) { // same markers, another camera: only the poses are still to do
                hipLaunchKernelGGL(k_marker_poses, dim3((AR_MAX_RECTS + 31) / 32, 1), dim3(64), 0, s, h->d_out.as<orbfe_marker>(),
                                   h->d_nout.as<int32_t>(), AR_MAX_RECTS, marker_size, *cam, h->d_poses.as<orbfe_marker_pose>());
                ORBFE_HIP(hipMemcpyAsync(hp + o.ps, h->d_poses.p, (size_t)AR_MAX_RECTS * sizeof(orbfe_marker_pose), hipMemcpyDeviceToHost, s));
            }
            ORBFE_HIP(hipStreamSynchronize(s));
            const int32_t* counts = reinterpret_cast<const int32_t*>(hp + o.cnt);
            const int32_t n = *reinterpret_cast<const int32_t*>(hp + o.n);
            if (!counts[2] && n <= capacity) { // no capacity flag: the speculated run is the result
                n_out[0] = n;
                if (n) memcpy(out, hp + o.mk, (size_t)n * sizeof(orbfe_marker));
                if (n && cam) memcpy(poses_out, hp + o.ps, (size_t)n * sizeof(orbfe_marker_pose));
                return ORBFE_OK;
            }
        }
    }
    if ((rc = h->d_in.ensure(dframe * nframes + 64)) || (rc = h->d_out.ensure((size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker))) ||
        (rc = h->d_nout.ensure((size_t)nframes * 4)))
        return rc;
    // page-locked staging: [frames in] then [n per frame | counts (4 per frame) | marker records] out -- three copies queued behind
    // the kernels and one wait instead of a blocking copy per array
    const DetOffsets o_ = det_offsets(dframe, nframes, cam != nullptr);
    const size_t o_n = o_.n, o_cnt = o_.cnt, o_mk = o_.mk, o_ps = o_.ps, o_end = o_.end;
    if ((rc = h->pinned.ensure(o_end))) return rc;
    if (cam && (rc = h->d_poses.ensure((size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker_pose)))) return rc;
    uint8_t* hp = h->pinned.as<uint8_t>();
    hipStream_t s = h->own_stream;
    for (int f = 0; f < nframes; f++)
        for (int y = 0; y < rows; y++) memcpy(hp + f * dframe + (size_t)y * dpitch, imgs + f * frame_stride + (size_t)y * step, (size_t)cols);
    ORBFE_HIP(hipMemcpyAsync(h->d_in.p, hp, dframe * nframes, hipMemcpyHostToDevice, s));
    const int32_t* counts = reinterpret_cast<const int32_t*>(hp + o_cnt);
    const bool user_big_mode = h->big_mode; // orbfe_aruco_set_big_frames applies to all following batches: keep it
    for (int attempt = 0; attempt < 3; attempt++) {
        rc = h->run_device(h->d_in.as<uint8_t>(), nframes, dframe, rows, cols, dpitch, h->d_out.as<orbfe_marker>(),
                           AR_MAX_RECTS, h->d_nout.as<int32_t>(), s);
        if (rc) { h->big_mode = user_big_mode; h->tiled_off = false; return rc; }
        if (cam)
            hipLaunchKernelGGL(k_marker_poses, dim3((AR_MAX_RECTS + 31) / 32, nframes), dim3(64), 0, s, h->d_out.as<orbfe_marker>(),
                               h->d_nout.as<int32_t>(), AR_MAX_RECTS, marker_size, *cam, h->d_poses.as<orbfe_marker_pose>());
        if (nframes <= 16) {   // a frame or a few: the results in one launch that writes the staging buffer (OutPack, orbfe_common.hpp)
            OutPack op;
            op.add(hp + o_n, h->d_nout.p, (size_t)nframes * 4);
            op.add(hp + o_cnt, h->d_counts.p, (size_t)nframes * 16);
            op.add(hp + o_mk, h->d_out.p, (size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker));
            if (cam) op.add(hp + o_ps, h->d_poses.p, (size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker_pose));
            if ((rc = op.flush<2>(s))) { h->big_mode = user_big_mode; h->tiled_off = false; return rc; }
        } else {
            ORBFE_HIP(hipMemcpyAsync(hp + o_n, h->d_nout.p, (size_t)nframes * 4, hipMemcpyDeviceToHost, s));
            ORBFE_HIP(hipMemcpyAsync(hp + o_cnt, h->d_counts.p, (size_t)nframes * 16, hipMemcpyDeviceToHost, s));
            ORBFE_HIP(hipMemcpyAsync(hp + o_mk, h->d_out.p, (size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker), hipMemcpyDeviceToHost, s));
            if (cam) ORBFE_HIP(hipMemcpyAsync(hp + o_ps, h->d_poses.p, (size_t)AR_MAX_RECTS * nframes * sizeof(orbfe_marker_pose), hipMemcpyDeviceToHost, s));
        }
        ORBFE_HIP(hipStreamSynchronize(s));
        int flags_or = 0;
        for (int f = 0; f < nframes; f++) flags_or |= counts[f * 4 + 2];
        // a frame with more segments than the tiled path's lists hold: the batch is done again by the relay kernels; one with more kept
        // borders (or border points) than those hold: by the single-walker kernel with its tables sized for AR_MAX_KEPT_BIG
        if (!h->escalate(flags_or)) break;
    }
    h->big_mode = user_big_mode; h->tiled_off = false;
    memcpy(n_out, hp + o_n, (size_t)nframes * 4);
    for (int f = 0; f < nframes; f++) {
        if (counts[f * 4 + 2])
            return fail(ORBFE_ERR_CAPACITY, "frame %d: internal detector capacity exceeded (flags 0x%x)", f, counts[f * 4 + 2]);
        if (n_out[f] > capacity) return fail(ORBFE_ERR_CAPACITY, "frame %d has %d markers, capacity is %d", f, n_out[f], capacity);
        if (n_out[f])
            memcpy(out + (size_t)f * capacity, hp + o_mk + (size_t)f * AR_MAX_RECTS * sizeof(orbfe_marker), (size_t)n_out[f] * sizeof(orbfe_marker));
        if (n_out[f] && cam)
            memcpy(poses_out + (size_t)f * capacity, hp + o_ps + (size_t)f * AR_MAX_RECTS * sizeof(orbfe_marker_pose),
                   (size_t)n_out[f] * sizeof(orbfe_marker_pose));
    }
    return ORBFE_OK;
}

extern "C" {

int orbfe_aruco_detect_batch(orbfe_aruco* h, const uint8_t* imgs, int nframes, size_t frame_stride, int rows, int cols,
                             size_t step, orbfe_marker* out, int capacity, int32_t* n_out)
{
    return detect_batch_impl(h, imgs, nframes, frame_stride, rows, cols, step, out, capacity, n_out, nullptr, 0.f, nullptr);
}

int orbfe_aruco_detect_poses(orbfe_aruco* h, const uint8_t* img, int rows, int cols, size_t step, orbfe_marker* out,
                             orbfe_marker_pose* poses, int capacity, int32_t* n_out, float marker_size, const float* K4,
                             const float* dist, int ndist)
{
    if (!poses) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_detect_poses: null argument");
    PoseCamera c;
    int rc = pose_camera(K4, dist, ndist, marker_size, c, "orbfe_aruco_detect_poses");
    if (rc) return rc;
    return detect_batch_impl(h, img, 1, 0, rows, cols, step, out, capacity, n_out, &c, marker_size, poses);
}

int orbfe_aruco_detect(orbfe_aruco* h, const uint8_t* img, int rows, int cols, size_t step, orbfe_marker* out,
                       int capacity, int32_t* n_out)
{
    return orbfe_aruco_detect_batch(h, img, 1, 0, rows, cols, step, out, capacity, n_out);
}

int orbfe_aruco_detect_bgr(orbfe_aruco* h, const uint8_t* bgr, int rows, int cols, size_t step, orbfe_marker* out, int capacity, int32_t* n_out)
{
    return detect_batch_impl(h, bgr, 1, 0, rows, cols, step, out, capacity, n_out, nullptr, 0.f, nullptr, 3);
}

int orbfe_aruco_detect_poses_bgr(orbfe_aruco* h, const uint8_t* bgr, int rows, int cols, size_t step, orbfe_marker* out,
                                 orbfe_marker_pose* poses, int capacity, int32_t* n_out, float marker_size, const float* K4,
                                 const float* dist, int ndist)
{
    if (!poses) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_detect_poses_bgr: null argument");
    PoseCamera c;
    int rc = pose_camera(K4, dist, ndist, marker_size, c, "orbfe_aruco_detect_poses_bgr");
    if (rc) return rc;
    return detect_batch_impl(h, bgr, 1, 0, rows, cols, step, out, capacity, n_out, &c, marker_size, poses, 3);
}

int orbfe_corner_subpix(const uint8_t* img, int rows, int cols, size_t step, float* pts, int n, int win, int max_iters, double eps, int device)
{
    if (!img || rows <= 0 || cols <= 0 || step < (size_t)cols || n < 0 || (n && !pts) || win < 1 || win > 8 || max_iters < 1 || !(eps >= 0.0))
        return fail(ORBFE_ERR_INVALID, "orbfe_corner_subpix: invalid argument (window half size 1 .. 8, max_iters >= 1, eps >= 0)");
    if (n == 0) return ORBFE_OK;
    int rc = use_device(device);
    if (rc) return rc;
    // the markers kernel on a scratch detector-free path: corners as n / 4 "markers" (padded), one frame
    const int nm = (n + 3) / 4;
    DevBuf d_img, d_mk, d_n, d_mask;
    const size_t pitch = (size_t)(cols + 63) / 64 * 64;
    if ((rc = d_img.ensure(pitch * rows + 64)) || (rc = d_mk.ensure((size_t)nm * sizeof(orbfe_marker))) || (rc = d_n.ensure(4)) || (rc = d_mask.ensure(17 * 17 * 4))) return rc;
    std::vector<orbfe_marker> mk((size_t)nm);
    memset(mk.data(), 0, mk.size() * sizeof(orbfe_marker));
    for (int i = 0; i < n; i++) { mk[i >> 2].corners[i & 3][0] = pts[2 * i]; mk[i >> 2].corners[i & 3][1] = pts[2 * i + 1]; }
    for (int i = n; i < 4 * nm; i++) { mk[i >> 2].corners[i & 3][0] = pts[0]; mk[i >> 2].corners[i & 3][1] = pts[1]; }
    std::vector<float> mask((size_t)17 * 17, 0.f);
    {
        const int ww = 2 * win + 1;
        for (int i = 0; i < ww; i++) {
            const float y = (float)(i - win) / win;
            const float vy = std::exp(-y * y);
            for (int j = 0; j < ww; j++) {
                const float x = (float)(j - win) / win;
                mask[(size_t)i * ww + j] = (float)(vy * std::exp(-x * x));
            }
        }
    }
    ORBFE_HIP(hipMemcpy2D(d_img.p, pitch, img, step, (size_t)cols, (size_t)rows, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(d_mk.p, mk.data(), mk.size() * sizeof(orbfe_marker), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(d_n.p, &nm, 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(d_mask.p, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_corner_subpix_markers, dim3(std::max(1, std::min(nm, 256)), 1), dim3(256), 0, 0, ImgView{d_img.as<uint8_t>(), nullptr, 0, (int)pitch}, cols, rows,
                       d_mk.as<orbfe_marker>(), d_n.as<int32_t>(), nm, win, std::min(max_iters, 100), eps * eps, d_mask.as<float>());
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpy(mk.data(), d_mk.p, mk.size() * sizeof(orbfe_marker), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) { pts[2 * i] = mk[i >> 2].corners[i & 3][0]; pts[2 * i + 1] = mk[i >> 2].corners[i & 3][1]; }
    return ORBFE_OK;
}

int orbfe_aruco_debug_image(orbfe_aruco* h, int frame, int stage, uint8_t* out)
{
    if (!h || !out || frame < 0 || frame >= h->last_nframes) return fail(ORBFE_ERR_INVALID, "debug_image: invalid argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    ORBFE_HIP(hipDeviceSynchronize());
    if (stage == 0) {
        std::vector<uint32_t> bits(h->bits_fu32);
        ORBFE_HIP(hipMemcpy(bits.data(), h->d_bits.as<uint32_t>() + (size_t)frame * h->bits_fu32, bits.size() * 4,
                            hipMemcpyDeviceToHost));
        for (int y = 0; y < h->rows; y++)
            for (int x = 0; x < h->cols; x++)
                out[(size_t)y * h->cols + x] = ((bits[(size_t)y * h->wpr + (x >> 5)] >> (x & 31)) & 1) ? 255 : 0;
        return ORBFE_OK;
    }
    if (stage == 104) { // the bit image the contour kernels of the last batch read (after the speck passes, if they ran)
        std::vector<uint32_t> bits(h->bits_fu32);
        ORBFE_HIP(hipMemcpy(bits.data(), (h->specks_ran ? h->d_bitsc : h->d_bits).as<uint32_t>() + (size_t)frame * h->bits_fu32, bits.size() * 4,
                            hipMemcpyDeviceToHost));
        for (int y = 0; y < h->rows; y++)
            for (int x = 0; x < h->cols; x++)
                out[(size_t)y * h->cols + x] = ((bits[(size_t)y * h->wpr + (x >> 5)] >> (x & 31)) & 1) ? 255 : 0;
        return ORBFE_OK;
    }
    if (stage >= 1 && stage < h->npyr + 1 && stage - 1 >= 1) { // pyramid level stage-1 (>= 1)
        const ArLevel& L = h->levels[stage - 1];
        ORBFE_HIP(hipMemcpy2D(out, L.w, h->d_pyr.as<uint8_t>() + (size_t)frame * h->pyr_fbytes + L.off, L.pitch, L.w, L.h,
                              hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    if (stage == 100) { // counts: nkept, nrect, flags, ncand as 4 int32
        ORBFE_HIP(hipMemcpy(out, h->d_counts.as<int32_t>() + frame * 4, 16, hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    if (stage == 101) { // rectangle candidates: AR_MAX_RECTS x ArRect (40 B)
        ORBFE_HIP(hipMemcpy(out, h->d_rects.as<ArRect>() + (size_t)frame * AR_MAX_RECTS, sizeof(ArRect) * AR_MAX_RECTS,
                            hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    if (stage == 102) { // tail of the kept array (phase timing words of instrumented builds), 96 bytes
        ORBFE_HIP(hipMemcpy(out, h->d_kept.as<ArKept>() + (size_t)frame * AR_MAX_KEPT + AR_MAX_KEPT - 4, 96,
                            hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    if (stage == 103) { // decode results of the frame, raw: AR_MAX_RECTS x (id, rotations)
        ORBFE_HIP(hipMemcpy(out, h->d_result.as<int32_t>() + (size_t)frame * AR_MAX_RECTS * 2, (size_t)AR_MAX_RECTS * 8,
                            hipMemcpyDeviceToHost));
        return ORBFE_OK;
    }
    return fail(ORBFE_ERR_INVALID, "debug_image: unknown stage %d", stage);
}

int orbfe_aruco_batch_status(orbfe_aruco* h, int32_t* nflagged, int32_t* flags_or)
{
    if (!h || !nflagged) return fail(ORBFE_ERR_INVALID, "orbfe_aruco_batch_status: null argument");
    int rc = use_device(h->device);
    if (rc) return rc;
    *nflagged = 0;
    if (flags_or) *flags_or = 0;
    if (h->last_nframes <= 0) return ORBFE_OK;
    ORBFE_HIP(hipDeviceSynchronize());
    std::vector<int32_t> counts((size_t)h->last_nframes * 4);
    ORBFE_HIP(hipMemcpy(counts.data(), h->d_counts.p, counts.size() * 4, hipMemcpyDeviceToHost));
    for (int f = 0; f < h->last_nframes; f++)
        if (counts[f * 4 + 2]) { (*nflagged)++; if (flags_or) *flags_or |= counts[f * 4 + 2]; }
    return ORBFE_OK;
}

int orbfe_aruco_set_big_frames(orbfe_aruco* h, int on)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    end_speculation(h);
    h->big_mode = on != 0;
    return ORBFE_OK;
}

int orbfe_aruco_set_aux_stream(orbfe_aruco* h, void* stream)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    h->user_aux = (hipStream_t)stream;
    return ORBFE_OK;
}

int orbfe_aruco_debug_kernel_times(orbfe_aruco* h, float* out_us, int capacity)
{
    if (!h) return fail(ORBFE_ERR_INVALID, "null handle");
    if (!out_us) { // control codes: 0/1 kernel timing off/on, 2/3 force the legacy contour kernel on/off, 4/5/6 tiled contour path by size / always / never,
                   // 7 returns the number of batches that were done again on the next contour path, 8 / 9 the speck passes on / off
        if (capacity == 7) return h->n_escalations;
        if (capacity == 8 || capacity == 9) { h->specks = capacity == 8 ? 1 : 0; return 0; }   // the speck passes on / off (default)
        if (capacity == 10 || capacity == 11) { h->specks_inkernel = capacity == 10; h->rows = h->cols = 0; return 0; }   // (the queue's size depends on it: geometry rebuilt)
        if (capacity == 14 || capacity == 15) { h->thr_mfma = capacity == 14; h->thr_mfma_auto = false; return 0; }
        if (capacity == 16) { h->thr_mfma = true; h->thr_mfma_auto = true; return 0; }
        if (capacity == 18 || capacity == 19) { h->half_pyr = capacity == 18; return 0; }   // k_half_pyr on (default) / off   // the threshold on the matrix cores on (default) / off
        if (capacity == 12 || capacity == 13) { h->thr_v2 = capacity == 12; return 0; }   // the threshold kernel with the fused pyramid on (default) / off   // ... inside the relay kernels on / off (default)
        if (capacity == 2 || capacity == 3) h->force_legacy = capacity == 2;
        else if (capacity >= 4 && capacity <= 6) {
            const int t = capacity == 4 ? -1 : capacity == 5 ? 1 : 0;
            if (h->tiled == 0 && t != 0) h->batch_cap = 0;   // the tiled path's workspace is only allocated while it can run: allocate on the next batch
            h->tiled = t;
        }
        else { h->timer.enabled = capacity != 0; h->timer.reset_history(); }
        return 0;
    }
    if (capacity < 0) return h->timer.collect_median(out_us, -capacity, nullptr);
    return h->timer.collect(out_us, capacity);
}

int orbfe_camera_resize(const float* K4, int cam_width, int cam_height, int img_width, int img_height, float* K4_out)
{
    if (!K4 || !K4_out || cam_width <= 0 || cam_height <= 0 || img_width <= 0 || img_height <= 0)
        return fail(ORBFE_ERR_INVALID, "orbfe_camera_resize: invalid argument");
    float k[4] = {K4[0], K4[1], K4[2], K4[3]};
    if (!(img_width == cam_width && img_height == cam_height)) { // cameraparameters.cpp:162-172
        const float ax = float(img_width) / float(cam_width), ay = float(img_height) / float(cam_height);
        k[0] *= ax; k[2] *= ax; k[1] *= ay; k[3] *= ay;
    }
    for (int i = 0; i < 4; i++) K4_out[i] = k[i];
    return ORBFE_OK;
}

int orbfe_marker_poses_batch_device(const orbfe_marker* d_markers, const int32_t* d_n, int capacity, int nframes,
                                    float marker_size, const float* K4, const float* dist, int ndist,
                                    orbfe_marker_pose* d_poses, void* stream)
{
    if (nframes < 0 || capacity < 0 || (nframes && capacity && (!d_markers || !d_poses)))
        return fail(ORBFE_ERR_INVALID, "orbfe_marker_poses_batch_device: invalid argument");
    PoseCamera c;
    int rc = pose_camera(K4, dist, ndist, marker_size, c, "orbfe_marker_poses_batch_device");
    if (rc) return rc;
    if (nframes == 0 || capacity == 0) return ORBFE_OK;
    hipLaunchKernelGGL(k_marker_poses, dim3((capacity + 31) / 32, nframes), dim3(64), 0, (hipStream_t)stream, d_markers, d_n,
                       capacity, marker_size, c, d_poses);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_marker_poses(const orbfe_marker* markers, int n, float marker_size, const float* K4, const float* dist, int ndist,
                       orbfe_marker_pose* poses, int device)
{
    if (n < 0 || (n && (!markers || !poses))) return fail(ORBFE_ERR_INVALID, "orbfe_marker_poses: invalid argument");
    PoseCamera c;
    int rc = pose_camera(K4, dist, ndist, marker_size, c, "orbfe_marker_poses");
    if (rc || (rc = use_device(device))) return rc;
    if (n == 0) return ORBFE_OK;
    PoseWorkspace& w = tl_pose_ws.get();
    const size_t mb = (size_t)n * sizeof(orbfe_marker), pb = (size_t)n * sizeof(orbfe_marker_pose);
    if ((rc = w.markers.ensure(mb)) || (rc = w.poses.ensure(pb)) || (rc = w.pinned.ensure(mb + pb + 64))) return rc;
    if (!w.stream) ORBFE_HIP(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    uint8_t* hp = w.pinned.as<uint8_t>();
    const size_t o_p = (mb + 63) / 64 * 64;
    memcpy(hp, markers, mb);
    ORBFE_HIP(hipMemcpyAsync(w.markers.p, hp, mb, hipMemcpyHostToDevice, w.stream));
    hipLaunchKernelGGL(k_marker_poses, dim3((n + 31) / 32, 1), dim3(64), 0, w.stream, w.markers.as<orbfe_marker>(), (const int32_t*)nullptr,
                       n, marker_size, c, w.poses.as<orbfe_marker_pose>());
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpyAsync(hp + o_p, w.poses.p, pb, hipMemcpyDeviceToHost, w.stream));
    ORBFE_HIP(hipStreamSynchronize(w.stream));
    memcpy(poses, hp + o_p, pb);
    return ORBFE_OK;
}

} // extern "C"
