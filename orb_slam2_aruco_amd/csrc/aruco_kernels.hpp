// aruco_kernels.hpp -- device-side structs and kernel declarations of the ArUco detector.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aruco_trace.hpp"
#include "orb_kernels.hpp" // ImgView, k_resize_level

namespace orbfe {

#define AR_MAX_RECTS 256
#define AR_MAX_KEPT 1024
#define AR_MAX_KEPT_BIG 4096 // the single-walker kernel with the bit image in HBM: LDS has room for this many kept borders

// a border that passed the length gate and the 4-gon/convexity test
struct ArKept {
    int off, len;     // its points inside the frame's point pool
    short vx[4], vy[4];
};
// rectangle candidate (MarkerCandidate): corners + contour
struct ArRect {
    float c[4][2];
    int off, len;
};
// one level of the detector's /2 pyramid
struct ArLevel {
    int w, h, pitch;
    long long off; // byte offset inside a frame's pyramid block (level 0 is the input image)
};

template <int TH_MAXR>
__global__ void k_adaptive_threshold(ImgView src, int W, int H, int win, int C, double scale, uint32_t* bits,
                                     size_t bits_fstride, int wpr);
template <int WIN>
__global__ void k_adaptive_threshold_t(ImgView src, int W, int H, int C, uint32_t magic, uint32_t* bits,
                                       size_t bits_fstride, int wpr, int ntx, int ntiles, int total);
// k_threshold_mfma: a 32-column strip per wave; c0 / c1 / c2 = byte columns of the three 16-byte pieces of a row it loads, tab = index
// (units of 64 uint4) of the strip's four pass-1 matrices (aruco_detector.hip: build_threshold_tables)
struct ThrStrip { int x0, c0, c1, c2, tab; };
__global__ void k_threshold_mfma(ImgView src, int W, int H, int rb, int kinit, const ThrStrip* strips, const uint4* tabs, const uint4* tab2,
                                 uint32_t* bits, size_t bits_fstride, int wpr, int nstrips, int nx, int total, int two_centre);
// threshold + the /2 pyramid levels a 64 x 64 tile holds whole (levels 1 .. n <= 4 of the detector's pyramid), one launch
struct ThrPyr { int n; int w[4], h[4], pitch[4]; long long off[4]; };
template <int WIN>
__global__ void k_threshold_pyr(ImgView src, int W, int H, uint32_t kk, uint32_t* bits, size_t bits_fstride, int wpr, int ntx, int ntiles,
                                int total, ImgView pyr, ThrPyr P);
// the run tests on the start candidates of the contour kernels (aruco_trace.hpp "FEWER WALKS" (1)); a build parameter so that a
// library without them can be measured next to the shipped one (tools/build_variant.sh); result-neutral
#ifndef ORBFE_CAND_FILTER
#define ORBFE_CAND_FILTER 1
#endif
// the speck passes ("FEWER WALKS" (2)): bit image -> the bit image the contour kernels are handed
#define SPK_THREADS 256
#define SPK_ROWS 48   // output rows per workgroup (+ ORBFE_SPECK_REACH above and below)
static inline size_t speck_lds_bytes(int cols) { return (size_t)4 * (SPK_ROWS + 2 * ORBFE_SPECK_REACH) * (((cols + 2 + 31) >> 5) + 1) * 4; }
__global__ void k_speck_clean(const uint32_t* bits, size_t bits_fstride, int wpr_g, int W, int H, uint32_t* out);
// Per-frame scratch of the relay kernels behind the long-walk queue words of d_candq: [start-candidate queue | rim masks and anchors
// of the two speck passes when they run inside the kernel (speck_pass_frame): 3 arrays a pass, each (padded rows + 2 * (H + 1)) rows]
__host__ __device__ inline int relay_queue_words(int W, int H) { return (W * H) / 16 + 64; }
__host__ __device__ inline int speck_frame_rows(int H, int HH) { return H + 2 + 2 * (HH + 1); }
__host__ __device__ inline size_t speck_frame_scratch_words(int W, int H)
{
    const size_t pw = (size_t)((W + 2 + 31) >> 5);
    return 3 * pw * (size_t)speck_frame_rows(H, ORBFE_SPECK_H1) + 3 * pw * (size_t)speck_frame_rows(H, ORBFE_SPECK_H2);
}
__global__ void k_half_area(ImgView src, ImgView dst, int dw, int dh);
__global__ void k_half_area4(ImgView src, ImgView dst, int dw4, int dh);
// levels 1 .. NF of the /2 pyramid from the source in one launch (NF = 4: 16 x 16 source blocks, 16-byte aligned rows; NF = 3: 8 x 8, 8-byte)
struct HalfPyrDst { uint8_t* base; size_t fstride; uint32_t off[4]; int pitch[4]; };
template <int NF> __global__ void k_half_pyr(ImgView src, HalfPyrDst P, int bw, int nblocks);
template <bool LDS_BITS>
__global__ void k_contours_t(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H, int lds_bits_words,
                           int min_len, uint32_t* candq, size_t candq_fstride, int candq_cap, uint32_t* pool,
                           size_t pool_fstride, int pool_cap, ArKept* kept_out, int kept_cap, ArRect* rects_out,
                           int rect_cap, int32_t* counts, uint32_t* gpadded, size_t gpadded_fstride, int only_flagged);

// HBM scratch of k_contours_relay, one per hash-table slot
struct RelaySeg {
    uint32_t nxt;    // slot of the next segment of the border
    uint32_t len;    // points of this segment
    uint32_t minoff; // offset of the segment's smallest start state; bit 31: that state is a hole-border start
    uint32_t stg;    // where the segment's points are staged inside the frame's pool
    uint32_t mn;     // the smallest start state the segment passes (0xffffffff: none)
};
__global__ void k_contours_relay(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H,
                                 int lds_bits_words, int min_len, int kshift, int tbits, RelaySeg* segs, uint32_t* pool,
                                 size_t pool_fstride, int pool_cap, ArKept* kept_out, int kept_cap, int kcap,
                                 unsigned long long* tail_keys, int32_t* tail_off, int32_t* counts, int32_t* hint, uint4* small_g, int32_t* rstate, int small_elsewhere, const uint16_t* lut_g, int f0, uint32_t* candq_g, size_t candq_fstride);
__global__ void k_contours_relay8(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H,
                                 int lds_bits_words, int min_len, int kshift, int tbits, RelaySeg* segs, uint32_t* pool,
                                 size_t pool_fstride, int pool_cap, ArKept* kept_out, int kept_cap, int kcap,
                                 unsigned long long* tail_keys, int32_t* tail_off, int32_t* counts, int32_t* hint, uint4* small_g, int32_t* rstate, int small_elsewhere, const uint16_t* lut_g, int f0, uint32_t* candq_g, size_t candq_fstride);
__global__ void k_contours_relay_wide(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H,
                                 int lds_bits_words, int min_len, int kshift, int tbits, RelaySeg* segs, uint32_t* pool,
                                 size_t pool_fstride, int pool_cap, ArKept* kept_out, int kept_cap, int kcap,
                                 unsigned long long* tail_keys, int32_t* tail_off, int32_t* counts, int32_t* hint, uint4* small_g, int32_t* rstate, int small_elsewhere, const uint16_t* lut_g, int f0, uint32_t* candq_g, size_t candq_fstride);
__global__ void k_contours_relay8g(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H,
                                 int lds_bits_words, int min_len, int kshift, int tbits, RelaySeg* segs, uint32_t* pool,
                                 size_t pool_fstride, int pool_cap, ArKept* kept_out, int kept_cap, int kcap,
                                 unsigned long long* tail_keys, int32_t* tail_off, int32_t* counts, int32_t* hint, uint4* small_g, int32_t* rstate,
                                 uint32_t* gpad, size_t gpad_fstride, const uint16_t* lut_g, int f0);
__global__ void k_relay_lut(uint16_t* lut);
__global__ void k_contours_small(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H, int min_len, const uint16_t* lut_g,
                                 int32_t* rstate, uint32_t* pool, size_t pool_fstride, int pool_cap, int kcap, unsigned long long* tail_keys,
                                 int32_t* tail_off, int32_t* counts);
__global__ void k_tail_prep(const unsigned long long* tail_keys, const int32_t* tail_off, int kcap, const int32_t* counts, uint4* work, size_t work_half,
                            int32_t* ctr);
__global__ void k_tail_approx(int kcap, const uint4* work, size_t work_half, const int32_t* ctr, const uint32_t* pool, size_t pool_fstride, ArKept* kept_out,
                              int kept_cap, uint8_t* rectflag, int pts);
__global__ void k_tail_finish(int kcap, const uint8_t* rectflag, const ArKept* kept_out, int kept_cap, ArRect* rects_out, int rect_cap,
                              int32_t* counts, int32_t* ctr);
// what k_tail_finish does, when k_prefilter does it in front of its own work (rectflag == nullptr: k_tail_finish ran as a launch)
struct TailFinish { int kcap; const uint8_t* rectflag; const ArKept* kept; int kept_cap; int32_t* ctr; };
__global__ void k_prefilter(ArRect* rects, int rect_cap, int32_t* counts, int W, int H, int too_near,
                            int32_t* cand_idx, int32_t* ncand_out, uint32_t* work, int32_t* wctr, TailFinish tf);
struct DcItem { // one rectangle candidate between the decode kernels
    double Mi[9]; // inverse homography of the warp
    int lvl, ok;  // pyramid level; 0 = singular system
    int isum, th; // first moment of the patch histogram; Otsu threshold
    int bin_lo, bin_hi; // first / last non-empty histogram bin
};
__global__ void k_decode_warp(ImgView src0, ImgView pyr, const ArLevel* levels, int nlevels, const ArRect* rects, int rect_cap,
                              const int32_t* cand_idx, int S, int W0, const uint32_t* work, const int32_t* wctr, DcItem* items,
                              uint16_t* hist, uint8_t* patch);
__global__ void k_decode_otsu(const int32_t* wctr, DcItem* items, const uint16_t* hist, int S);
__global__ void k_decode_vote(ImgView src0, ImgView pyr, const ArLevel* levels, int rect_cap, int S, int nb,
                              const unsigned long long* codes, int ncodes, const unsigned long long* scodes, const int32_t* sids,
                              int nsorted, int max_corr, const uint32_t* work, const int32_t* wctr, const DcItem* items,
                              const uint8_t* patch, int32_t* result);
__global__ void k_finalize(const ArRect* rects, int rect_cap, const int32_t* cand_idx, const int32_t* ncand,
                           const int32_t* result, const uint32_t* pool, size_t pool_fstride, orbfe_marker* out,
                           int out_cap, int32_t* n_out, int refine_lines, int32_t* out_src, int32_t* wctr);
// aruco_modes.hip: THRES_AUTO_FIXED, Params::minSize > 0, CORNER_SUBPIX, CV_8UC3 input
__global__ void k_fixed_threshold(ImgView src, int W, int H, int thr, uint32_t* bits, size_t bits_fstride, int wpr);
__global__ void k_erode_cross_xor(const uint32_t* in, uint32_t* out, size_t bits_fstride, int wpr, int W, int H, int r);
__global__ void k_enlarge_candidates(ArRect* rects, int rect_cap, const int32_t* counts, int fact);
__global__ void k_resize_nearest(ImgView src, ImgView dst, int sw, int sh, int dw, int dh, double ifx, double ify);
__global__ void k_bgr_to_gray(const uint8_t* bgr, size_t bgr_fstride, size_t step, ImgView dst, int W, int H, int bits15);
__global__ void k_marker_hist(const uint32_t* work, const int32_t* wctr, const int32_t* result, int rect_cap, const uint16_t* hist,
                              uint32_t* out);
__global__ void k_corner_subpix_markers(ImgView src, int W, int H, orbfe_marker* markers, const int32_t* n_out, int capacity, int win,
                                        int max_iters, double eps2, const float* mask);
__global__ void k_upsample_corners(ImgView src0, ImgView pyr, const ArLevel* levels, int start, int work_w, ArRect* rects, int rect_cap,
                                   const int32_t* cand_idx, const uint32_t* work, const int32_t* wctr, const int32_t* result,
                                   const float* masks);

#define DC_PATCH_BYTES 1232   // = DC_PXCAP of aruco_kernels.hip: bytes per kept patch

#define CT_THREADS 256          // threads that run the whole kernel
#define CT_WAVES (CT_THREADS / 64)
#define CT_PROBE_THREADS 1024   // threads at launch: the extra ones help with the (throughput-bound) probe phase, then exit
#define AP_STACK 64
#define AP_OUT 64
#define CT_PROBE 24 // steps a border start is followed before it is queued as a long walk (< the 70-point gate)

#ifndef DC_WAVES
#define DC_WAVES 4   // k_decode_warp / _vote: waves per workgroup (a candidate per wave)
#endif
#ifndef RL_THREADS
#define RL_THREADS 512   // 8 waves per frame: measured against 1024 (contours alone 713 -> 640 us per 300 frames, step 2.00 -> 1.95 ms) and 256 (893 us)
#endif
#ifndef RT_QUAD_MAX
#define RT_QUAD_MAX 256           // k_tail_approx: borders of fewer points are done by 16 lanes, four to a wave (a multiple of 8)
#endif
#ifndef RT_PTS
#define RT_PTS 1024                // k_tail_approx: points of a long border kept in LDS per wave (>= 4 * RT_QUAD_MAX: the four short ones share it)
#endif
#define RT_BUCKETS 256           // k_tail_prep: length classes of the work list's counting sort
#ifndef RT_WGS
#define RT_WGS 2048              // k_tail_approx: persistent workgroups of 4 waves (16 waves per CU at its ~128 VGPRs)
#endif
#ifndef RL_THREADS_BIG
#define RL_THREADS_BIG 1024     // k_contours_relay8 (large frames): its workgroup owns the CU (LDS), so it brings 16 waves
#endif
#define RL_SLOTS_PER_THREAD (4096 / RL_THREADS)   // table slots <= RL_THREADS * RL_SLOTS_PER_THREAD (tbits <= 12)
#define RL_NIL 0xffff
#define RL_COPY_CAP RL_THREADS  // (documentation) kept segments per frame the flat copy lists (== RL_THREADS; 14 bytes each <= the key table)
#define RL_FLAG_TABLE 32        // (kernel-internal) markers did not fit: coarsen the grid
#define RL_FLAG_BUG 64          // an invariant of the relay formulation failed: redone by k_contours_t as well
#define RL_FALLBACK_FLAGS (RL_FLAG_TABLE | RL_FLAG_BUG)
#define RL_KCAP AR_MAX_KEPT      // kept borders per frame (k_contours_relay + k_contours_tail)
#ifndef RL_STEPS_PER_ITER
#define RL_STEPS_PER_ITER 2      // walk steps between two looks at the work queue
#endif
// k_contours_small (phase (c) of frames with a grid, one wave per block of K rows x 128 columns): threads per workgroup, columns
// per block (log2), tile words per row (the block's 8 + one to the left + one to the right + ring8()'s funnel word), tile rows
// (K <= 128: K + 4), queue entries per wave (>= the start candidates one row of a block can have), rows per round, steps per
// look at the queue
#define RS_THREADS 256
#ifndef RS_BLOCK_SHIFT
#define RS_BLOCK_SHIFT 7   // 128 columns: measured against 256 and 64 on the 1920 x 1080 batch (1.32 / 1.47 / 1.78 ms)
#endif
#define RS_TW ((1 << (RS_BLOCK_SHIFT - 5)) + 3)
#define RS_TILE_ROWS (128 + 4)
#ifndef RS_QCAP
#define RS_QCAP 512
#endif
#define RS_ROUND_ROWS 32
#ifndef RS_STEPS
#define RS_STEPS 2
#endif

// ---- the tiled relay formulation (aruco_tiles.hip)
#define CTW_THREADS 256          // k_ct_walk: four independent waves per workgroup, a tile each
#define CTW_ROWS 35              // tile rows in LDS: the closed band of 33 rows + one above + one below
#ifndef CTW_QCAP
#define CTW_QCAP 512             // queue entries per wave (>= 64: what one enumeration item can yield)
#endif
#define CTW_FCAP 64              // finished segments a wave collects before it appends them to the frame's list
#ifndef CTW_STEPS
#define CTW_STEPS 4              // walk steps between two looks at the queue (1920 x 1080 step 3.89 / 3.75 / 3.68 / 3.65 ms with 1 / 2 / 3 / 4: the take /
                                 // expand / finish blocks of a loop trip run for a few lanes each and cost as much issue time as two steps)
#endif
#ifndef CTW_TAKE
#define CTW_TAKE 256             // enumeration items per refill of the queue (halved while they do not fit)
#endif
#define CTW_MAX_CW 480           // tile width limit: the marker pixels of all relay columns of a tile (31 x (cw / 32 + 1)) fit the queue
#define CT_STATE_INTS 8          // per frame: segments, kept small borders, pool words in use, flags, start candidates
#define CT_CODE_WORDS 4          // chain code of a segment: 10 steps of 3 bits per word, kept in registers while the segment is walked and written with
                                 // its record (16 bytes); a walk that fills it ends the segment there and goes on as the next one
#define CT_CODE_STEPS (10 * CT_CODE_WORDS)
#define CTB_FCAP 32              // finished segments a wave of k_ct_band collects before it appends them to the frame's list
#define CTL_THREADS 1024
#define CTB_THREADS 512          // k_ct_band: eight waves per band
#define CTB_MCAP 8192            // marker pixels per (frame, band) its list holds
#ifndef CTB_STEPS
#define CTB_STEPS 2
#endif
inline size_t ctb_lds_bytes(int W, int rb) { return ((size_t)((W + 2 + 31) >> 5) * (32 * rb + 3) + 2) * 4 + 16; }   // the band's rows of the padded bit image
__global__ void k_ct_band(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H, int min_len, const uint16_t* lut_g, int rb,
                          uint32_t* mlist, int mcap, unsigned long long* htab, int hbits, unsigned gen, uint32_t* seg, size_t seg_fstride, int segcap,
                          int32_t* ctstate, uint32_t* pool, size_t pool_fstride, int pool_cap, int kcap, unsigned long long* tail_keys, int32_t* tail_off,
                          uint4* codes);
inline int ctw_wave_lds_bytes(int cw) { return (2 * (CTW_ROWS * ((cw >> 5) + 2) + 2) * 4 + CTW_QCAP * 2 + CTW_FCAP * 40 + 15) & ~15; }   // two tile slots, queue, finished segments
__global__ void k_ct_walk(const uint32_t* gbits, size_t bits_fstride, int wpr_g, int W, int H, int min_len, const uint16_t* lut_g, int cw, int ncols,
                          int nbands, int total_tiles, unsigned long long* htab, int hbits, unsigned gen, uint32_t* seg, size_t seg_fstride,
                          int segcap, int32_t* ctstate, uint32_t* pool, size_t pool_fstride, int pool_cap, int kcap, unsigned long long* tail_keys,
                          int32_t* tail_off, int wave_bytes, uint4* codes);
__global__ void k_ct_lists(const uint32_t* seg, size_t seg_fstride, int segcap, int32_t* ctstate, const unsigned long long* htab, int hbits, unsigned gen,
                           unsigned long long* gelem, int lcap, int min_len, int pool_cap, int kcap, unsigned long long* tail_keys, int32_t* tail_off,
                           int32_t* counts, int32_t* rstate, uint4* itemsA, uint2* itemsB, int ipf, int32_t* nitems);
__global__ void k_ct_points(const uint4* itemsA, const uint2* itemsB, int ipf, const int32_t* nitems, const uint32_t* codes,
                            int segcap, uint32_t* pool, size_t pool_fstride);

// LDS of k_contours_relay: region R (bit image | list arrays) followed by the marker keys
__host__ __device__ inline size_t relay_region_bytes(int lds_bits_words, int kcap, int tbits)
{
    const size_t bits = ((size_t)lds_bits_words * 4 + 15) & ~(size_t)15;
    const size_t lists = (size_t)kcap * 12 + ((size_t)8 << tbits);
    size_t r = bits > lists ? bits : lists;
    return (r + 15) & ~(size_t)15;
}
inline size_t relay_lds_bytes(int lds_bits_words, int kcap, int tbits)
{
    return relay_region_bytes(lds_bits_words, kcap, tbits) + ((size_t)4 << tbits);
}

// LDS of k_tail_prep (keys, pool offsets, lengths per kept border) and of k_tail_approx (per wave: approx output + stack + `pts` points)
inline size_t tail_prep_lds_bytes(int kcap) { return (size_t)kcap * (8 + 4) + 16; }
inline size_t tail_approx_lds_bytes(int pts) { return (size_t)4 * ((AP_OUT + AP_STACK) * 8 + (size_t)pts * 4) + 16; }

inline size_t contours_lds_bytes(int lds_bits_words, int kept_cap)
{
    size_t b = ((size_t)lds_bits_words * 4 + 15) & ~(size_t)15;
    b += (size_t)kept_cap * 8;      // keys
    b += (size_t)kept_cap * 4 * 4;  // arena offsets, len, off, rect flag (the last three double as the long-walk queue)
    b += (size_t)CT_WAVES * AP_OUT * 8;
    b += (size_t)CT_WAVES * AP_STACK * 8;
    b += (size_t)kept_cap * 2; // length ranking
    return b + 16;
}

} // namespace orbfe
