/*
 * orbfe.h -- C ABI of the MI355X-native per-frame front-end for ORB_SLAM2_aruco.
 *
 * One shared library (liborbfe.so, built by hipcc for gfx950) exports exactly the
 * entry points below.  Plain pointers and sizes only: no C++ types, no torch
 * types, nothing thrown across the boundary (status codes instead; the C++ shim
 * classes in include/shims/ re-raise to keep reference behaviour).
 *
 * Each group names the reference interface it replaces (file:line in the
 * reference tree, CarminLiu/ORB_SLAM2_aruco):
 *
 *   orbfe_extractor_*   ORB_SLAM2::ORBextractor            include/ORBextractor.h:45-113
 *                       ctor                               src/ORBextractor.cc:410-470
 *                       operator()(image, mask, kps, desc) src/ORBextractor.cc:1043-1105
 *                       Get{Levels,ScaleFactor(s),...}     include/ORBextractor.h:63-83
 *                       called from Frame::ExtractORB      src/Frame.cc:200-206
 *   orbfe_aruco_*       aruco::MarkerDetector              Thirdparty/aruco/aruco/markerdetector.h:96-312
 *                       setDictionary / setDetectionMode / setCornerRefinementMethod / detect
 *                       configured + called at             src/Frame.cc:129-142
 *   orbfe_hamming*,     ORBmatcher::DescriptorDistance     src/ORBmatcher.cc:1651-1667
 *   orbfe_knn2*,        best/second-best inner loop of every SearchBy*  
 *   orbfe_search_for_initialization   ORBmatcher::SearchForInitialization  src/ORBmatcher.cc:409-524
 *
 * Memory convention: functions without a suffix take HOST pointers (drop-in for
 * the reference's call sites, which hand over cv::Mat / std::vector storage) and
 * do their own H2D/D2H staging.  Functions ending in _device take DEVICE pointers
 * (hipMalloc'd, e.g. torch CUDA tensors) and a hipStream_t passed as void*; they
 * launch asynchronously on that stream and are the batched-video path.
 *
 * Threading: an extractor / detector handle is a stateful, non-re-entrant object
 * like the classes it replaces (one handle per stream of frames).  The
 * matching functions are thread-safe (device scratch per calling thread, device and stream, released when the thread
 * exits); so is orbfe_vocabulary_transform on one shared
 * vocabulary handle (ComputeBoW runs on the Tracking, LocalMapping and LoopClosing threads).
 */
#ifndef ORBFE_H
#define ORBFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* liborbfe.so is built with -fvisibility=hidden: what is declared between here and the matching pop is its whole dynamic symbol table */
/* the four opaque handles (their members are not part of the ABI: declared before the push, so that the library's own definitions
 * of them keep hidden visibility) */
typedef struct orbfe_extractor orbfe_extractor;   /* ORB_SLAM2::ORBextractor */
typedef struct orbfe_aruco orbfe_aruco;           /* aruco::MarkerDetector */
typedef struct orbfe_vocabulary orbfe_vocabulary; /* DBoW2 ORBVocabulary */
typedef struct orbfe_pipeline orbfe_pipeline;     /* the batched-video mode */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* status codes (0 = ok, <0 = error; orbfe_last_error() holds the message of the calling thread) */
#define ORBFE_OK 0
#define ORBFE_ERR_INVALID (-1)   /* bad argument (null pointer, non-positive size, ...) */
#define ORBFE_ERR_NO_DEVICE (-2) /* no gfx950 device / HIP runtime unusable: the product path has NO CPU fallback */
#define ORBFE_ERR_HIP (-3)       /* a HIP call failed */
#define ORBFE_ERR_CAPACITY (-4)  /* caller-provided output capacity too small */
#define ORBFE_ERR_DICT (-5)      /* unknown dictionary name */

/* cv::KeyPoint layout (pt.x, pt.y, size, angle, response, octave, class_id) = 28 bytes;
 * fields are set as at src/ORBextractor.cc:822-846,:477 */
typedef struct orbfe_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbfe_keypoint;

/* aruco::Marker essentials (marker.h:42-56): id + 4 refined corners (x,y) in detection order */
typedef struct orbfe_marker {
    int32_t id;
    float corners[4][2];
} orbfe_marker;

const char* orbfe_last_error(void);
const char* orbfe_version(void);
int orbfe_device_count(void); /* number of usable HIP devices, 0 if none */

/* ------------------------------------------------------------------ ORB extractor -- */

/* ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST); device = HIP device ordinal. NULL on error. */
orbfe_extractor* orbfe_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                                        int device);
void orbfe_extractor_destroy(orbfe_extractor* h);

int orbfe_extractor_get_levels(const orbfe_extractor* h);                                 /* GetLevels */
float orbfe_extractor_get_scale_factor(const orbfe_extractor* h);                         /* GetScaleFactor */
int orbfe_extractor_get_scale_factors(const orbfe_extractor* h, float* out);              /* GetScaleFactors */
int orbfe_extractor_get_inverse_scale_factors(const orbfe_extractor* h, float* out);      /* GetInverseScaleFactors */
int orbfe_extractor_get_scale_sigma_squares(const orbfe_extractor* h, float* out);        /* GetScaleSigmaSquares */
int orbfe_extractor_get_inverse_scale_sigma_squares(const orbfe_extractor* h, float* out);/* GetInverseScaleSigmaSquares */
int orbfe_extractor_get_features_per_level(const orbfe_extractor* h, int32_t* out);       /* mnFeaturesPerLevel */
/* upper bound on keypoints per frame (the quadtree may overshoot a level quota by a few nodes,
 * src/ORBextractor.cc:730): size kps/desc buffers with this */
int orbfe_extractor_max_keypoints(const orbfe_extractor* h);

/* operator()(image, mask(ignored), keypoints, descriptors) for one CV_8UC1 frame in host memory.
 * img: rows x cols bytes with row stride `step`.  kps: capacity records; desc: capacity x 32 bytes, row i belongs to
 * kps[i].  *n_out = number of keypoints.  Empty image (rows==0 || cols==0 || img==NULL) -> ORBFE_OK with *n_out
 * untouched semantics of :1046 mapped to *n_out = 0.
 * Frame geometry this implementation accepts (ORBFE_ERR_INVALID otherwise): every pyramid level at least 62 pixels in both
 * directions (below that the reference's cell grid has no column, ORBextractor.cc:780-784); levels up to 4127 pixels (a keypoint travels as 12 + 12 bits inside the kernels);
 * 1 to 16 quadtree roots per level, nIni = round(width / height) of the border-less level (ORBextractor.cc:544): portrait frames give
 * nIni = 0, on which the reference divides by zero and indexes an empty vector; frames wider than 16.5 : 1 are a capacity limit of
 * the quadtree kernels; a level's keypoint quota (mnFeaturesPerLevel) up to about 2700 -- its node lists live in LDS -- i.e.
 * nfeatures up to ~12000 at 8 levels and scale 1.2 (ORBFE_ERR_CAPACITY beyond). */
int orbfe_extract(orbfe_extractor* h, const uint8_t* img, int rows, int cols, size_t step, orbfe_keypoint* kps,
                  uint8_t* desc, int capacity, int32_t* n_out);

/* Batched video mode, host buffers: nframes images `frame_stride` bytes apart; outputs are nframes blocks of
 * `capacity` records; n_out[nframes]. */
int orbfe_extract_batch(orbfe_extractor* h, const uint8_t* imgs, int nframes, size_t frame_stride, int rows, int cols,
                        size_t step, orbfe_keypoint* kps, uint8_t* desc, int capacity, int32_t* n_out);

/* Batched video mode, device buffers, asynchronous on `stream` (hipStream_t). d_n_out: int32[nframes] on device. */
int orbfe_extract_batch_device(orbfe_extractor* h, const uint8_t* d_imgs, int nframes, size_t frame_stride, int rows,
                               int cols, size_t step, orbfe_keypoint* d_kps, uint8_t* d_desc, int capacity,
                               int32_t* d_n_out, void* stream);
/* The 8-bit taps of the extractor's GaussianBlur(7x7, sigma 2) (ORBextractor.cc:1086) depend on the OpenCV release the
 * reference is built against.  mode 0 (default): every tap rounded on its own, 18 34 49 55 49 34 18 (sum 257) -- OpenCV 2.4 / 3.2
 * (what CMakeLists.txt:32-38 asks for) and the first fixed-point GaussianBlur of 3.4.  mode 1: the later bit-exact kernel
 * (rounding error carried from tap to tap, the centre tap takes the remainder), 18 34 48 56 48 34 18 (sum 256) -- late 3.4.x, 4.x.
 * Output is sat_u8((sum + 2^15) >> 16) either way. */
int orbfe_extractor_set_gaussian_taps(orbfe_extractor* h, int mode);

/* The device-pointer entry point is asynchronous and cannot return a capacity error: a frame whose keypoint total exceeds
 * `capacity` is clamped to it.  After the batch (synchronises the device): *overflow = 0, or the largest per-frame total
 * that did not fit -- the batch's records are then incomplete and the call must be repeated with capacity >= *overflow
 * (orbfe_extractor_max_keypoints() always suffices).  The flag covers every batch since it was last read (reading clears it). */
int orbfe_extractor_batch_status(orbfe_extractor* h, int32_t* overflow);

/* Stage read-back for parity tests (valid after an extract call; `frame` indexes the last batch).
 * stage 0: pyramid level image, 1: blurred level image -> out must hold w*h bytes (tightly packed). */
int orbfe_extractor_debug_level_size(orbfe_extractor* h, int level, int* w, int* hgt);
int orbfe_extractor_debug_level_image(orbfe_extractor* h, int frame, int level, int stage, uint8_t* out);
/* stage 0: FAST candidates handed to the quadtree (x,y relative to the 16-px border, response=score),
 * stage 1: keypoints kept by the quadtree (level coordinates).  Returns count in *n (<= capacity copied). */
int orbfe_extractor_debug_level_keypoints(orbfe_extractor* h, int frame, int level, int stage, orbfe_keypoint* out,
                                          int capacity, int32_t* n);
/* per-kernel timing of the last batch call on this handle, microseconds, in launch order; returns number written.
 * out_us == NULL: `capacity` is a control code (0 / 1 = event timing off / on, which also clears the history).  capacity < 0:
 * the per-interval MEDIAN over the batches recorded since timing was switched on (the newest 64 at most), -capacity slots --
 * what bench.py reports, because one batch's events are not the steady state of a pipelined run. */
/* The extractor forks one launch (the blur, which only needs the pyramid) onto a second stream so that it overlaps with
 * FAST and the quadtree.  By default that is a stream the handle owns; an application that already has a lightly
 * loaded stream (bench.py: the matching stream) can lend it instead -- ROCm maps streams onto a few hardware queues, and
 * two busy streams on one queue serialise.  NULL restores the internal stream.  Ordering is by events either way. */
int orbfe_extractor_set_aux_stream(orbfe_extractor* h, void* stream);
/* A second fork (experimental; it did not shorten the batched step; naming a stream switches it on): level 0 of the pyramid is the
 * caller's image, so its FAST cells (a third of all pixels) can be searched from the start of the batch, next to the resize chain
 * instead of behind it, on the stream named here.  An application whose
 * streams already fill the hardware queues (ROCm has four; a fifth stream shares one and queues behind whatever runs there) names
 * a stream with little work at the start of a batch.  NULL switches the early launch off again (the default).  Ordering is by events either way. */
int orbfe_extractor_set_early_stream(orbfe_extractor* h, void* stream);
/* Phase lock between the extractor handles of a pipeline (device-pointer batches): every batch of `h` starts behind a stage of the
 * latest batch enqueued on `other` -- stage 1 = its FAST, 2 = its quadtree, 3 = its descriptors, 4 = its resize chain (FAST starts); 0 or
 * other == NULL: free running.
 * Two engine sets that follow each other run a fixed half-period apart instead of in whatever phase contention leaves them
 * (bench.py: measured).  stage + 10 * g adds a second gate: the handle's FAST waits for stage g of `other` (with stage
 * % 10 == 0 the resize chain starts freely; measured slower in every combination, docs/history/DESIGN_rounds_1-4.md §6b).  `other` must outlive the
 * relation.  Results do not depend on it. */
int orbfe_extractor_follow(orbfe_extractor* h, orbfe_extractor* other, int stage);
/* The same relation for any other work of the pipeline: whatever is enqueued on `stream` after this call starts behind stage 1 .. 4
 * of the latest batch enqueued on `h` (nothing to wait for before its first batch). */
int orbfe_extractor_stage_wait(orbfe_extractor* h, int stage, void* stream);

/* Pairing for the drop-in path.  The reference builds a Frame by calling ORBextractor::operator() and then
 * MarkerDetector::detect on the SAME grey image (Frame.cc:91 / :200-206, then :142), one after the other on the Tracking thread; the
 * two are independent.  With a detector paired to an extractor, a one-frame orbfe_extract uploads the image once and starts the
 * detector on that device copy on the detector's own stream, next to its own launches; the following orbfe_aruco_detect /
 * orbfe_aruco_detect_poses call, if it is handed the same image (rows, cols and every pixel, compared with the copy the extractor
 * staged), waits for that
 * work and returns its results -- poses included when camera and marker size are those of the detector's previous call.  Any
 * other image, a batch call, or a capacity flag: the call runs as if nothing had been started.  Results are identical either way.
 * detector == NULL unpairs; unpair (or destroy the extractor) before destroying the detector.  Both handles on one device. */
int orbfe_extractor_pair_detector(orbfe_extractor* h, struct orbfe_aruco* detector);
int orbfe_extractor_debug_kernel_times(orbfe_extractor* h, float* out_us, int capacity);

/* ------------------------------------------------------------------ descriptor matching -- */
/* Debug/test switch.  "knn2_path": 0 = pick by problem size (default), 1 = VALU tile kernel, 2 = matrix-core kernel.
 * Both kernels give identical results.  No key of the shipped library skips work ("orb_skip" / "aruco_skip" exist only in
 * the -DORBFE_ABLATION diagnosis build, whose orbfe_version() says "+ablation"; here they are ORBFE_ERR_INVALID). */
int orbfe_debug_control(const char* key, int value);

/* popcount(a XOR b) over 256 bits, host pointers (ORBmatcher::DescriptorDistance) */
int orbfe_hamming(const uint8_t* a, const uint8_t* b);
/* Two more host-side scalar helpers for the reference's protected members (the device searches carry their own copies):
 * ComputeThreeMaxima (ORBmatcher.cc:1605-1646) on L bin populations -> ind3[0..2], -1 where the 10 % rule drops a runner-up;
 * CheckDistEpipolarLine (ORBmatcher.cc:139-157): 1 if keypoint 2 lies within 3.84 sigma^2 of the epipolar line of keypoint 1
 * (F12 row-major 3 x 3, level_sigma2 = mvLevelSigma2[kp2.octave]). */
void orbfe_three_maxima(const int32_t* counts, int L, int32_t* ind3);
int orbfe_epipolar_distance_ok(float x1, float y1, float x2, float y2, const float* F12, float level_sigma2);

/* All-pairs best / second-best of nq query descriptors against nt train descriptors (32 B rows), rule of App. D:
 *   if d<best {second=best; best=d; idx=t} else if d<second {second=d}   (first candidate wins ties)
 * best and second start at `init` (256 or INT_MAX in the reference). idx = -1 when nt == 0. Host pointers. */
int orbfe_knn2(const uint8_t* Q, int nq, const uint8_t* T, int nt, int init, int32_t* best_idx, int32_t* best_dist,
               int32_t* second_dist, int device);

/* The same rule over per-query candidate lists in CSR form (offsets[nq + 1] starting at 0, idx[offsets[nq]] rows of T):
 * the inner loop of the guided searches whose candidates the caller builds (SearchByBoW node lists, ORBmatcher.cc:159-292;
 * Fuse; SearchForTriangulation).  Candidates are visited in list order.  Host pointers. */
int orbfe_knn2_csr(const uint8_t* Q, int nq, const uint8_t* T, int nt, const int32_t* offsets, const int32_t* idx, int init,
                   int32_t* best_idx, int32_t* best_dist, int32_t* second_dist, int device);

/* Batched device variant: npairs independent (Q,T) problems. Q/T: device pointers to descriptor blocks,
 * block p at Q + p*q_stride bytes with d_nq[p] valid rows (<= max_nq), same for T. Outputs blocks of max_nq. */
int orbfe_knn2_batch_device(const uint8_t* d_Q, const int32_t* d_nq, size_t q_stride, int max_nq, const uint8_t* d_T,
                            const int32_t* d_nt, size_t t_stride, int max_nt, int npairs, int init,
                            int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist, void* stream);

/* ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize).  The keypoints are the
 * UNDISTORTED ones (mvKeysUn).  bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY} of the undistorted image
 * (Frame::ComputeImageBounds, Frame.cc:418-451): the 64 x 48 Frame grid spans them (Frame.cc:112-113); NULL = a camera
 * without distortion, 0, 0, cols, rows (Frame.cc:440-446).  prev_matched: n1 x 2 floats, updated in place like
 * vbPrevMatched; matches12: n1 ints (-1 = none).  Returns ORBFE_OK and the match count in *nmatches. */
int orbfe_search_for_initialization(const orbfe_keypoint* kps1, const uint8_t* desc1, int n1,
                                    const orbfe_keypoint* kps2, const uint8_t* desc2, int n2, int cols, int rows,
                                    const float* bounds, float* prev_matched, int32_t* matches12, int window_size, float nnratio,
                                    int check_orientation, int32_t* nmatches, int device);

/* The matching loop of ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.cc:45-129) on
 * flat arrays; the projection of the map points, the viewing-cosine radius and the MapPoint bookkeeping stay with the
 * caller.  Query q = a map point in view: window centre (x, y), radius r (already times the scale factor of the predicted
 * level, :71), octave range [min_level, max_level] (:71 passes predicted-1 .. predicted; max_level < 0 = no upper bound,
 * both <= 0 / < 0 = no octave test, as Frame::GetFeaturesInArea, Frame.cc:280-333), 32-byte descriptor.  Candidates are
 * GetFeaturesInArea's, in its order.  taken[i] != 0: keypoint i already carries a map point with observations (:91-93).
 *   mode 0: best / second-best distance and octave per query (:84-118), no state between queries -- also the inner loop
 *           of the (CurrentFrame, LastFrame) and (CurrentFrame, KeyFrame) variants (:1413-1433, :1551-1571);
 *           best_idx = -1, distances 256, levels -1 when there is no candidate.  match / nmatches may be NULL.
 *   mode 1: the whole loop: accept when best <= th_high (TH_HIGH = 100) and not (same octave and best > nnratio*second)
 *           (:121-128); an accepted keypoint is taken for the queries after it when the map point it received is observed
 *           (q_observed[q] = pMP->Observations() > 0, the test of :91-93; NULL = all observed).  match[q] = keypoint index or
 *           -1 (a keypoint assigned twice keeps the later query, as F.mvpMapPoints[bestIdx] = pMP does), *nmatches =
 *           accepted count, taken[] updated in place.  The raw outputs may be NULL.
 * Host pointers.  bounds as for orbfe_search_for_initialization; mono (no right-image test). */
typedef struct orbfe_window_query {
    float x, y, r;
    int32_t min_level, max_level;
} orbfe_window_query;
int orbfe_search_by_projection(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                               const orbfe_window_query* queries, const uint8_t* qdesc, int nq, uint8_t* taken, const uint8_t* q_observed,
                               int mode, int th_high, float nnratio, int32_t* best_idx, int32_t* best_dist, int32_t* best_level,
                               int32_t* second_dist, int32_t* second_level, int32_t* match, int32_t* nmatches, int device);

/* The same loop over the frames of a batch resident on the device (no host round trip per call): frame f owns block f of
 * `capacity` keypoint records (d_n[f] valid, the layout of orbfe_extract_batch_device) and block f of `qcapacity` queries
 * (d_nq[f] valid) with their descriptors, flags and outputs.  mode 0 / 1 as above; mode 2 = the best-only loop of
 * orbfe_search_by_projection_best (d_q_angle, factor, check_orientation, d_match_cur[f][keypoint] = query or -1; d_q_observed
 * plays q_blocks).  d_taken (may be NULL) is updated in place.  Asynchronous on `stream`; a candidate row that overflowed the
 * per-stream scratch truncates silently, so ask orbfe_search_by_projection_batch_status afterwards: *overflow = 0, or the
 * longest candidate list -- the scratch has then been grown and repeating the call succeeds.  The flag is sticky like the extractor's
 * and SearchForInitialization's: it covers every _batch_device search / fuse call on this stream since it was last read (reading
 * clears it), so two batches may be enqueued before one status call. */
int orbfe_search_by_projection_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int capacity, int nframes,
                                            int cols, int rows, const float* bounds, const orbfe_window_query* d_queries,
                                            const uint8_t* d_qdesc, const int32_t* d_nq, int qcapacity, uint8_t* d_taken,
                                            const uint8_t* d_q_observed, const float* d_q_angle, int mode, int th_high, float nnratio,
                                            float factor, int check_orientation, int32_t* d_best_idx, int32_t* d_best_dist,
                                            int32_t* d_best_level, int32_t* d_second_dist, int32_t* d_second_level, int32_t* d_match,
                                            int32_t* d_match_cur, int32_t* d_nmatches, void* stream);
int orbfe_search_by_projection_batch_status(void* stream, int32_t* overflow);

/* Scratch of the matching entry points is kept per (calling thread, device, stream) so that Tracking / LocalMapping / LoopClosing
 * calls do not serialise on each other.  Reuse a small fixed set of streams: at most 16 (device, stream) slots are kept per thread,
 * least recently used first out.  Before destroying a stream that was handed to a `_device` matching call, release its scratch
 * (grown candidate strides, unread overflow flags) so that a new stream at the same address starts clean.  Synchronises `stream`. */
int orbfe_release_stream_scratch(void* stream);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (src/ORBmatcher.cc:1332-1474; what
 * TrackWithMotionModel runs every frame), monocular, whole on the device: the projection of the last frame's map points
 * with the current pose (:1362-1389), the window search on the Frame grid (levels octave +- 1, :1396), best match <= th_high
 * (TH_HIGH = 100, :1421), the blocking of keypoints that received an observed map point (:1397-1399), the rotation
 * histogram (factor 1/HISTO_LENGTH, :1341) and ComputeThreeMaxima (:1440-1471).
 * Per feature i of the last frame: valid_last[i] = has a map point and is not an outlier (NULL = all); x3Dw = its world
 * position (n_last x 3); mp_desc = MapPoint::GetDescriptor() (n_last x 32); mp_observed[i] = Observations() > 0 (NULL = all).
 * taken_cur[i2] = the keypoint already holds an observed map point (NULL = none).  Tcw = 3x4 row-major [Rcw | tcw]
 * (CurrentFrame.mTcw), K4 = fx fy cx cy, scale_factors = mvScaleFactors.  match_cur[i2] = i (the keypoint receives the
 * map point of last-frame feature i) or -1; *nmatches = the return value.  Host pointers. */
int orbfe_search_by_projection_last_frame(const orbfe_keypoint* kps_cur, const uint8_t* desc_cur, int n_cur, const uint8_t* taken_cur,
                                          int cols, int rows, const float* bounds, const orbfe_keypoint* kps_last, int n_last,
                                          const uint8_t* valid_last, const float* x3Dw, const uint8_t* mp_desc, const uint8_t* mp_observed,
                                          const float* Tcw, const float* K4, const float* scale_factors, int nlevels, float th, int th_high,
                                          int check_orientation, int32_t* match_cur, int32_t* nmatches, int device);
/* The same loop for queries projected by the caller -- SearchByProjection(CurrentFrame, KeyFrame, sAlreadyFound, th, ORBdist)
 * (:1476-1603: th_high = ORBdist, levels nPredictedLevel +- 1 from MapPoint::PredictScale, q_blocks = NULL) and the other
 * best-only variants: queries (r < 0 = skip) with their descriptors, q_angle = the source keypoints' angles, factor = the
 * histogram factor of the variant. */
int orbfe_search_by_projection_best(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                                    const orbfe_window_query* queries, const float* q_angle, const uint8_t* qdesc, const uint8_t* q_blocks,
                                    int nq, const uint8_t* taken, int th_high, int check_orientation, float factor, int32_t* match_cur,
                                    int32_t* nmatches, int device);

/* The matching part of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:829-970; chi2 = 5.99) and of Fuse(pKF, Scw,
 * vpPoints, th, vpReplacePoint) (:972-1104; chi2 = 0, Tcw / Ow from the decomposed Scw) on the device: per map point the
 * projection and gates (positive depth, KeyFrame::IsInImage, scale-invariance range, viewing angle < 60 deg), PredictScale,
 * the window search on the keyframe grid at levels [predicted - 1, predicted], the reprojection gate and the best
 * descriptor distance.  valid[i] = "pMP && !isBad() && !IsInKeyFrame(pKF)" resp. "!isBad() && !alreadyFound" (NULL = all);
 * min_dist / max_dist = the map point's mfMinDistance / mfMaxDistance: the range gate applies the 0.8f / 1.2f of
 * Get{Min,Max}DistanceInvariance() (MapPoint.cc:402-412) and MapPoint::PredictScale divides mfMaxDistance itself (:419) -- the
 * two are not recoverable from the getters' values bit-exactly, so the members are what every guided search here takes;
 * normal = GetNormal() (nmp x 3), mp_desc = GetDescriptor().
 * Tcw = 3x4 row-major [Rcw | tcw], Ow = camera centre (3).  best_idx[i] = keypoint or -1, best_dist[i] = its distance (256
 * if none); the caller applies bestDist <= TH_LOW and does the map bookkeeping (Replace / AddObservation, :943-964). */
int orbfe_fuse_search(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds, const float* p3Dw,
                      const uint8_t* valid, const float* min_dist, const float* max_dist, const float* normal, const uint8_t* mp_desc, int nmp,
                      const float* Tcw, const float* Ow, const float* K4, const float* scale_factors, const float* inv_level_sigma2, int nlevels,
                      float log_scale_factor, float th, double chi2, int32_t* best_idx, int32_t* best_dist, int device);

/* The same over several keyframes resident on the device: LocalMapping::SearchInNeighbors fuses the current keyframe's map points
 * into every neighbour (`matcher.Fuse(pKFi, vpMapPointMatches)` in a loop, src/LocalMapping.cc:850-858) -- one call instead of a
 * host round trip per keyframe.  Keyframe k owns block k of `capacity` keypoint records / descriptors (d_n[k] valid; the layout of
 * orbfe_extract_batch_device); the nmp map points (device arrays) are shared, d_valid (may be NULL) is [nkf][nmp] because
 * IsInKeyFrame(pKF) depends on the keyframe; Tcw / Ow are HOST arrays of nkf poses (12 / 3 floats each).  Outputs [nkf][nmp].
 * Asynchronous on `stream`; a candidate row that overflowed the per-stream scratch truncates silently:
 * orbfe_search_by_projection_batch_status(stream) reports it (and grows the scratch for a repeat). */
int orbfe_fuse_search_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int capacity, int nkf, int cols, int rows,
                                   const float* bounds, const float* d_p3Dw, const uint8_t* d_valid, const float* d_min_dist,
                                   const float* d_max_dist, const float* d_normal, const uint8_t* d_mp_desc, int nmp, const float* Tcw,
                                   const float* Ow, const float* K4, const float* scale_factors, const float* inv_level_sigma2, int nlevels,
                                   float log_scale_factor, float th, double chi2, int32_t* d_best_idx, int32_t* d_best_dist, void* stream);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/ORBmatcher.cc:1106-1330): the map points of
 * each keyframe are carried into the other camera (world -> own camera -> sR | t of the similarity, each stage rounded to
 * float), gated (positive depth, IsInImage, distance in the scale-invariance range of the FINAL camera coordinates),
 * PredictScale, window search at levels [predicted - 1, predicted], best distance <= th_high (TH_HIGH = 100); pairs that
 * agree in both directions are returned.  Feature i of keyframe k carries p3Dw_k[i], min/max_dist_k[i], mp_desc_k[i];
 * valid_k[i] = "map point exists, is not bad, is not already matched" (:1148-1155, :1230-1236; NULL = all).  T1w / T2w =
 * 3x4 row-major [R | t]; sT12 = [s12 R12 | t12], sT21 = [(1/s12) R12' | t21] as the caller computes them (:1123-1126).
 * match12[i1] = i2 or -1 (vpMatches12[i1] = vpMapPoints2[i2]); *nfound = the return value. */
int orbfe_search_by_sim3(const orbfe_keypoint* kps1, const uint8_t* desc1, int n1, const orbfe_keypoint* kps2, const uint8_t* desc2, int n2,
                         int cols, int rows, const float* bounds, const float* p3Dw1, const uint8_t* valid1, const float* min_dist1,
                         const float* max_dist1, const uint8_t* mp_desc1, const float* p3Dw2, const uint8_t* valid2, const float* min_dist2,
                         const float* max_dist2, const uint8_t* mp_desc2, const float* T1w, const float* T2w, const float* sT12,
                         const float* sT21, const float* K4, const float* scale_factors, int nlevels, float log_scale_factor, float th,
                         int th_high, int32_t* match12, int32_t* nfound, int device);

/* ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:294-407, loop closing): map points
 * projected with the similarity's rigid part (Tcw = [Rcw | tcw], Ow as computed at :303-308), gated as in Fuse incl. the viewing
 * angle, searched at levels [predicted - 1, predicted] among keypoints not yet matched (matched[i] = vpMatched[i] != NULL; a
 * keypoint that receives a point is matched for the points after it), best distance <= TH_LOW.  valid[i] = "!isBad() && not
 * in vpMatched".  match_kf[i] = index of the map point the keypoint received or -1; *nmatches = the return value. */
int orbfe_search_by_projection_sim3(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                                    const uint8_t* matched, const float* p3Dw, const uint8_t* valid, const float* min_dist,
                                    const float* max_dist, const float* normal, const uint8_t* mp_desc, int nmp, const float* Tcw,
                                    const float* Ow, const float* K4, const float* scale_factors, int nlevels, float log_scale_factor, int th,
                                    int32_t* match_kf, int32_t* nmatches, int device);

/* Only the projection + gates + PredictScale, as window queries for orbfe_search_by_projection / _best: r < 0 = not searched,
 * min_level = predicted - level_below, max_level = predicted + level_above.  normal == NULL: no viewing-angle gate.
 * keyframe_variant != 0: the projection of Fuse / SearchBySim3 / SearchByProjection(pKF, Scw, ...) (:321-358, :848-893): positive
 * depth, invz = 1 / z in float, u = fx * (xc * invz) + cx, KeyFrame::IsInImage (u < mnMaxX).  keyframe_variant == 0: the
 * projection of SearchByProjection(CurrentFrame, pKF, ...) (:1503-1512): NO depth gate, invzc = 1.0 / z in double,
 * u = (fx * xc) * invzc + cx, the Frame bounds test (u <= mnMaxX). */
int orbfe_project_map_points(const float* p3Dw, const uint8_t* valid, const float* min_dist, const float* max_dist, const float* normal, int n,
                             const float* Tcw, const float* Ow, const float* K4, int cols, int rows, const float* bounds, int keyframe_variant,
                             const float* scale_factors, int nlevels, float log_scale_factor, float th, int level_below, int level_above,
                             orbfe_window_query* queries, int device);

/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * (src/ORBmatcher.cc:1476-1603; Tracking::Relocalization calls it with (10, 100) and (3, 64), Tracking.cc:1858, :1875), whole on
 * the device: the projection above (keyframe_variant = 0), the distance gate, PredictScale, the window search at levels
 * predicted - 1 .. predicted + 1 among keypoints without a map point (taken_cur[i2] = CurrentFrame.mvpMapPoints[i2] != NULL; a
 * keypoint that receives a point is taken for the points after it), best distance <= orb_dist, rotation histogram (factor
 * 1 / HISTO_LENGTH) and ComputeThreeMaxima.  Per feature i of the keyframe: valid[i] = "has a map point that is not bad and
 * not in sAlreadyFound" (NULL = all), p3Dw, min_dist / max_dist, mp_desc as below, kf_angle[i] = pKF->mvKeysUn[i].angle.
 * Ow = -Rcw' tcw (:1482).  match_cur[i2] = i or -1; *nmatches = the return value. */
int orbfe_search_by_projection_keyframe(const orbfe_keypoint* kps_cur, const uint8_t* desc_cur, int n_cur, const uint8_t* taken_cur, int cols,
                                        int rows, const float* bounds, int n_kf, const float* kf_angle, const uint8_t* valid, const float* p3Dw,
                                        const float* min_dist, const float* max_dist, const uint8_t* mp_desc, const float* Tcw, const float* Ow,
                                        const float* K4, const float* scale_factors, int nlevels, float log_scale_factor, float th, int orb_dist,
                                        int check_orientation, int32_t* match_cur, int32_t* nmatches, int device);

/* Batched device variant over npairs frame pairs (frame t vs t-1 of a stream); all arrays are blocks of `capacity`
 * records per frame; pair p matches frame p (as F1) against frame p+1 (as F2). prev_matched == NULL means
 * "start from F1's own keypoint positions" (what Tracking does on the first call, src/Tracking.cc:520-523). */
int orbfe_search_for_initialization_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                                                 const int32_t* d_n, int capacity, int npairs, int cols, int rows, const float* bounds,
                                                 int window_size, float nnratio, int check_orientation,
                                                 int32_t* d_matches12, int32_t* d_nmatches, void* stream);
/* The batch entry point cannot return a capacity error either (the host-pointer one retries by itself): the candidate rows of a
 * frame pair live in one pool, and a pair that needs more entries than the pool holds loses rows.  After a batch issued by THIS
 * thread on `stream` (synchronises the stream): *overflow = 0, or the number of pool entries the fullest pair needed -- the batch's
 * matches are then incomplete; the per-stream pool has been grown, so repeating the batch call succeeds.  More than 1024 level-0
 * keypoints in a frame: ORBFE_ERR_CAPACITY (*overflow = that count).  The flag covers every batch since it was last read (reading
 * clears it). */
int orbfe_search_for_initialization_batch_status(void* stream, int32_t* overflow);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:270-333; called after every new observation by Tracking,
 * LocalMapping and LoopClosing): for every map point, among the descriptors it was observed with (CSR: point p owns rows
 * offsets[p] .. offsets[p + 1] of desc, in the order of its observation map), the one with the least median Hamming distance
 * to the others -- median = sorted row[(int)(0.5 * (N - 1))], first row wins ties.  best_idx[p] = its index inside the point's
 * list (-1 for a point without descriptors, which the reference leaves untouched); best_desc (may be NULL) receives the chosen
 * descriptor (npoints x 32).  At most 256 observations per point.  Host pointers / device pointers + stream. */
int orbfe_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx, uint8_t* best_desc, int device);
int orbfe_distinctive_descriptors_device(const uint8_t* d_desc, const int32_t* d_offsets, int npoints, int32_t* d_best_idx, uint8_t* d_best_desc,
                                         void* stream);

/* ------------------------------------------------------------------ DBoW2 vocabulary transform -- */
/* ORBVocabulary (= DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, include/ORBVocabulary.h) as Frame::ComputeBoW and
 * KeyFrame::ComputeBoW use it (src/Frame.cc:348-355, src/KeyFrame.cc): transform(descriptors, BowVector, FeatureVector, 4).
 * The handle owns the tree in HBM.  scoring / weighting are DBoW2's enums (BowVector.h:37-55): weighting 0 TF_IDF, 1 TF,
 * 2 IDF, 3 BINARY; scoring 0 L1_NORM .. 5 DOT_PRODUCT (decides the normalisation, ScoringObject.h:74-91). */

/* TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425; System.cc loads
 * ORBvoc.txt with it): "k L scoring weighting" then one line per node "parent isLeaf d0..d31 weight".  An empty line --
 * the one the reference's `while(!f.eof())` reads after the file's last newline -- becomes a node exactly as there (child of
 * the root, no word, weight 0; its descriptor, left unset by the reference, is zero here).  NULL + orbfe_last_error() on failure. */
orbfe_vocabulary* orbfe_vocabulary_load_text(const char* filename, int device);
/* The same tree from arrays: nodes 1..nnodes in file order (the root, node 0, is implicit; parent[i] refers to node ids, so
 * parent[i] <= i); descriptors nnodes x 32 bytes; weights as parsed. */
orbfe_vocabulary* orbfe_vocabulary_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t* parent,
                                          const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights, int device);
void orbfe_vocabulary_destroy(orbfe_vocabulary* v);
int orbfe_vocabulary_info(const orbfe_vocabulary* v, int32_t* out6); /* k, L, scoring, weighting, nodes (incl. root), words */

/* transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1127-1194) for one frame, host pointers.  Per feature
 * (each may be NULL): word_id, node_id (the ancestor at level L - levelsup, :1218-1259), weight.  The two vectors (all
 * seven pointers or none): BowVector as *nbow (word id ascending, value) pairs; FeatureVector as *nfv node ids ascending
 * with fv_offset[0..*nfv] into fv_feature (feature indices ascending within a node).  Capacity of every array: n
 * (fv_offset: n + 1).  Bit-exact doubles: weights are added and normalised in the reference's map order.
 * n > 4096 with vectors is ORBFE_ERR_CAPACITY. */
int orbfe_vocabulary_transform(orbfe_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id,
                               double* weight, uint32_t* bow_word, double* bow_value, int32_t* nbow, uint32_t* fv_node,
                               int32_t* fv_offset, uint32_t* fv_feature, int32_t* nfv);
/* Batch on the device over the extractor's output (blocks of `capacity` records per frame, d_n[f] valid; fv_offset
 * blocks are capacity + 1).  d_word / d_node / d_weight are required (scratch for the vector kernel). */
int orbfe_vocabulary_transform_batch_device(orbfe_vocabulary* v, const uint8_t* d_desc, const int32_t* d_n, int capacity, int nframes,
                                            int levelsup, int32_t* d_word, int32_t* d_node, double* d_weight, uint32_t* d_bow_word,
                                            double* d_bow_value, int32_t* d_nbow, uint32_t* d_fv_node, int32_t* d_fv_offset,
                                            uint32_t* d_fv_feature, int32_t* d_nfv, void* stream);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) (src/ORBmatcher.cc:159-292) and SearchByBoW(KeyFrame*,
 * KeyFrame*, vpMatches12) (:526-659) on flat arrays: side 1 = the keyframe whose map points are looked for, side 2 = the
 * frame / second keyframe, each with the FeatureVector orbfe_vocabulary_transform returned.  valid1[i] != 0 = "feature i has
 * a map point that is not bad" (:196-201; NULL = all); valid2 the same for side 2 (the KF-KF variant, :583-590; NULL = the
 * KF-Frame variant).  The variants also differ in accept_max (best <= TH_LOW = 50, :233, vs best < TH_LOW, :608 -> 49) and
 * factor (this fork's HISTO_LENGTH / 360.0f, :174, vs 1.0f / HISTO_LENGTH, :546).  match12[i1] = i2 or -1 and match21[i2] =
 * i1 or -1 (vpMapPointMatches[i2] = map point of match21[i2]; vpMatches12[i1] = map point of match12[i1]); *nmatches is the
 * return value.  At most 4096 features per side.  Host pointers. */
int orbfe_search_by_bow(const orbfe_keypoint* kps1, const uint8_t* desc1, const uint8_t* valid1, int n1, const uint32_t* fv_node1,
                        const int32_t* fv_offset1, const uint32_t* fv_feature1, int nfv1, const orbfe_keypoint* kps2,
                        const uint8_t* desc2, const uint8_t* valid2, int n2, const uint32_t* fv_node2, const int32_t* fv_offset2,
                        const uint32_t* fv_feature2, int nfv2, float nnratio, int check_orientation, int accept_max, float factor,
                        int32_t* match12, int32_t* match21, int32_t* nmatches, int device);
/* Batch over npairs pairs of frames of one set of per-frame blocks (the layouts of orbfe_extract_batch_device and
 * orbfe_vocabulary_transform_batch_device; d_valid may be NULL): pair p matches frame d_pair1[p] (side 1) with frame
 * d_pair2[p] (side 2); NULL index arrays mean p and p + 1.  use_valid2 != 0 applies d_valid to side 2 as well.
 * d_match12 / d_match21 are blocks of `capacity` per pair. */
int orbfe_search_by_bow_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_valid, const int32_t* d_n,
                                     const uint32_t* d_fv_node, const int32_t* d_fv_offset, const uint32_t* d_fv_feature,
                                     const int32_t* d_nfv, int capacity, const int32_t* d_pair1, const int32_t* d_pair2, int npairs,
                                     int use_valid2, float nnratio, int check_orientation, int accept_max, float factor,
                                     int32_t* d_match12, int32_t* d_match21, int32_t* d_nmatches, void* stream);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo = false) (src/ORBmatcher.cc:661-827),
 * monocular: features of the two keyframes without a map point (has_mp* != 0 = skip, :710-714, :731-735; NULL = none has
 * one), paired per vocabulary node, distance <= TH_LOW, not within 10 px * sqrt(scale) of the epipole (ex, ey) (:752-757,
 * the caller computes it, :669-675), within the 3.84 sigma^2 band of the epipolar line x1' F12 (CheckDistEpipolarLine,
 * :139-157; F12 3x3 row-major), rotation histogram with factor 1/HISTO_LENGTH.  scale_factors2 / level_sigma2_2 =
 * pKF2->mvScaleFactors / mvLevelSigma2.  match12[i1] = i2 or -1: vMatchedPairs = the pairs in ascending i1. */
int orbfe_search_for_triangulation(const orbfe_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_mp1, int n1, const uint32_t* fv_node1,
                                   const int32_t* fv_offset1, const uint32_t* fv_feature1, int nfv1, const orbfe_keypoint* kps2,
                                   const uint8_t* desc2, const uint8_t* has_mp2, int n2, const uint32_t* fv_node2, const int32_t* fv_offset2,
                                   const uint32_t* fv_feature2, int nfv2, const float* F12, float ex, float ey, const float* scale_factors2,
                                   const float* level_sigma2_2, int nlevels, int check_orientation, int32_t* match12, int32_t* nmatches,
                                   int device);
/* Batch over pairs of frames as orbfe_search_by_bow_batch_device; d_free[i] != 0 = "feature i has no map point" (NULL = all),
 * d_F12 npairs x 9 and d_epipole npairs x 2 on the device, the level tables on the host; d_scratch21 = capacity ints per pair. */
int orbfe_search_for_triangulation_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_free, const int32_t* d_n,
                                                const uint32_t* d_fv_node, const int32_t* d_fv_offset, const uint32_t* d_fv_feature,
                                                const int32_t* d_nfv, int capacity, const int32_t* d_pair1, const int32_t* d_pair2, int npairs,
                                                const float* d_F12, const float* d_epipole, const float* scale_factors,
                                                const float* level_sigma2, int nlevels, int check_orientation, int32_t* d_match12,
                                                int32_t* d_scratch21, int32_t* d_nmatches, void* stream);

/* ------------------------------------------------------------------ Frame glue: undistortion -- */
/* cv::undistortPoints(src, dst, K, distCoeffs, noArray(), K) (OpenCV 3.4: 5 fixed-point iterations in double) on n
 * (x, y) float pairs -- what Frame::UndistortKeyPoints (src/Frame.cc:357-387) and Frame::UndistortArucoCorners
 * (:389-416) run over mvKeys / the 4*NA marker corners.  K4 = {fx, fy, cx, cy}; dist = k1 k2 p1 p2 [k3 [k4 k5 k6
 * [s1 s2 s3 s4]]] (ORB_SLAM2's mDistCoef has 4 or 5).  Always undistorts (the callers' "mDistCoef[0] == 0 -> copy"
 * shortcut is the caller's, see the batch variant).  Host pointers; src may equal dst. */
int orbfe_undistort_points(const float* src, int n, const float* K4, const float* dist, int ndist, float* dst, int device);

/* Frame::UndistortKeyPoints over a batch of extractor outputs on the device: frame f's first min(d_n[f], capacity)
 * records get undistorted (x, y), every other field is kept (Frame.cc:379-386); with ndist == 0 or dist[0] == 0 the
 * records are copied unchanged (:359-363).  d_kps_un may equal d_kps. */
int orbfe_undistort_keypoints_batch_device(const orbfe_keypoint* d_kps, const int32_t* d_n, int capacity, int nframes,
                                           const float* K4, const float* dist, int ndist, orbfe_keypoint* d_kps_un, void* stream);

/* Frame::ComputeImageBounds (src/Frame.cc:418-451): bounds = {mnMinX, mnMinY, mnMaxX, mnMaxY}, the `bounds` argument
 * of the search entry points above.  {0, 0, cols, rows} when dist[0] == 0. */
int orbfe_compute_image_bounds(int cols, int rows, const float* K4, const float* dist, int ndist, float* bounds, int device);

/* ------------------------------------------------------------------ on-disk keyframe features -- */
/* The feature records of the reference's binary map files (src/Map.cc:297-321 SaveKeyFrame, :478-511 LoadKeyFrame): per
 * feature pt.x pt.y size angle response (f32) octave (i32) | mDescriptors.cols (i32 = 32) | 32 descriptor bytes | map
 * point index (u64, ULONG_MAX = none) = 68 bytes, packed back to back after each keyframe header (id u64, timestamp f64,
 * quaternion 4 x f32, translation 3 x f32, N i32).  class_id is not stored; unpack sets -1 (cv::KeyPoint's default). */
#define ORBFE_KF_FEATURE_BYTES 68
/* n records <-> arrays, host pointers.  mp_index may be NULL (pack: ULONG_MAX everywhere; unpack: not returned).
 * unpack fails with ORBFE_ERR_INVALID when a record's descriptor length is not 32 (LoadKeyFrame would read a
 * different number of bytes there, Map.cc:492-495). */
int orbfe_keyframe_features_pack(const orbfe_keypoint* kps, const uint8_t* desc, const uint64_t* mp_index, int n, uint8_t* out, int device);
int orbfe_keyframe_features_unpack(const uint8_t* in, int n, orbfe_keypoint* kps, uint8_t* desc, uint64_t* mp_index, int device);
/* A whole map file image resident on the device: segment s = the features of one keyframe, starting at byte
 * d_seg_offset[s] of d_file (4-byte aligned, as every offset of the format is) and occupying output indices
 * d_seg_first[s] .. d_seg_first[s + 1].  NULL offsets = one segment at offset 0 of max_records_per_segment records.
 * *d_bad counts records whose descriptor length is not 32 (zero it first). */
int orbfe_keyframe_features_unpack_device(const uint8_t* d_file, const uint64_t* d_seg_offset, const int32_t* d_seg_first, int nsegments,
                                          int max_records_per_segment, orbfe_keypoint* d_kps, uint8_t* d_desc, uint64_t* d_mp_index,
                                          int32_t* d_bad, void* stream);
int orbfe_keyframe_features_pack_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const uint64_t* d_mp_index,
                                        const uint64_t* d_seg_offset, const int32_t* d_seg_first, int nsegments,
                                        int max_records_per_segment, uint8_t* d_file, void* stream);

/* ------------------------------------------------------------------ ArUco marker detector -- */

/* MarkerDetector + setDictionary(dict) + setDetectionMode(DM_NORMAL) +
 * setCornerRefinementMethod(CORNER_LINES): the configuration of src/Frame.cc:135-137. NULL on error. */
orbfe_aruco* orbfe_aruco_create(const char* dictionary, int device);
void orbfe_aruco_destroy(orbfe_aruco* h);
int orbfe_aruco_set_dictionary(orbfe_aruco* h, const char* dictionary);
int orbfe_aruco_max_markers(const orbfe_aruco* h);
/* MarkerDetector::Params the shim forwards (markerdetector.h:96-214).
 *   setDictionary(name, error_correction_rate) / MarkerDetector(dict, rate): rate in [0, 1]; > 0 enables the error-correction pass
 *     of DictionaryBased::detect (a code closer than int(tau * rate) bits to a dictionary entry is accepted; default 0 = exact).
 *   setDetectionMode(dm, minMarkerSize) (markerdetector.cpp:374-391): DM_NORMAL (0, Frame.cc:136), DM_FAST (1) and DM_VIDEO_FAST (2).
 *     The fast modes threshold with THRES_AUTO_FIXED: one global threshold, carried from frame to frame (Otsu over the pixels of the
 *     markers just found, markerdetector_impl.cpp:7003-7040) and replaced by 10 + rand() % 230 -- the process's own rand() sequence,
 *     as in the reference -- when a frame yields nothing (:6903-6990); DM_VIDEO_FAST also derives the next frame's minMarkerSize from
 *     the smallest marker of this one (:8790-8880).  Such a handle is frame-sequential: the host-pointer entry points take its frames
 *     one at a time in order, orbfe_aruco_detect_batch_device refuses it.  minMarkerSize in [0, 1] (Params::minSize, a fraction of
 *     the larger image side) > 0 detects on an INTER_NEAREST reduction of the frame and brings the corners back through the /2
 *     pyramid with cv::cornerSubPix (cornerUpsample, :14028-14220); reductions below 64 x 48 pixels are refused.
 *   setCornerRefinementMethod(m): CORNER_SUBPIX (0: cv::cornerSubPix, window 4, 12 iterations, eps 0.005, :8511), CORNER_LINES
 *     (1, default here, Frame.cc:137) and CORNER_NONE (2); anything but CORNER_SUBPIX resets minMarkerSize to 0 (markerdetector.cpp:392-395).
 *   orbfe_aruco_get_state: Params::ThresHold and Params::minSize as the last call left them, the threshold passes of the last call
 *     and the size of the image it worked on (any pointer may be NULL). */
int orbfe_aruco_set_error_correction_rate(orbfe_aruco* h, float rate);
int orbfe_aruco_set_detection_mode(orbfe_aruco* h, int mode, float min_marker_size);
int orbfe_aruco_set_corner_refinement(orbfe_aruco* h, int method);
/* Params::detectEnclosedMarkers (markerdetector.h:126): markers whose corners touch other dark squares (chessboard-like boards).
 * With THRES_AUTO_FIXED the thresholded image is replaced by its inner edge band (erode with a cross, xor); in every mode each
 * rectangle candidate is enlarged by half the threshold window along its diagonals (markerdetector_impl.cpp:2871-2950, :10620-10690). */
int orbfe_aruco_set_enclosed_markers(orbfe_aruco* h, int on);
/* Params::trackingMinDetections (markerdetector.h:187; 0 = off, the default): a marker that was found in that many calls and is
 * missing from this one is looked for among the rectangle candidates the dictionary rejected -- centre inside its last outline,
 * area within 30 %, nearest centre -- and returned with its id and the corner order that continues its previous orientation
 * (markerdetector_impl.cpp:7107-7890).  Host logic over two short lists, as in the reference; the handle becomes frame-sequential.
 * Setting it resets the history.  orbfe_aruco_last_tracked: markers the last call recovered that way. */
int orbfe_aruco_set_tracking(orbfe_aruco* h, int min_detections);
int orbfe_aruco_last_tracked(const orbfe_aruco* h);
int orbfe_aruco_get_state(const orbfe_aruco* h, int32_t* threshold, float* min_size, int32_t* attempts, int32_t* work_rows,
                          int32_t* work_cols);
/* cvtColor(BGR2GRAY) of the CV_8UC3 entry points: 14 fractional bits (OpenCV <= 3.4.1: B 1868, G 9617, R 4899; default) or 15
 * (3.4.2 and later: 3735, 19235, 9798). */
int orbfe_aruco_set_gray_conversion(orbfe_aruco* h, int fractional_bits);
/* aruco::Marker::contourPoints (marker.h:56) of marker `marker` (index into the output of the last detect / batch call) of frame
 * `frame`: the full border the rectangle came from, (x, y) int32 pairs.  *n = its length; min(*n, capacity) points are written.
 * Host pointers; synchronises the device. */
int orbfe_aruco_marker_contour(orbfe_aruco* h, int frame, int marker, int32_t* xy, int capacity, int32_t* n);
/* The same for the first `nmarkers` markers of the frame in ONE round trip: offsets[0 .. nmarkers] are filled in any case (marker i owns
 * points offsets[i] .. offsets[i+1]); the (x, y) pairs are written only when offsets[nmarkers] <= capacity (points), so a caller
 * asks once with capacity 0 for the total or, simply, passes a generous buffer. */
int orbfe_aruco_marker_contours(orbfe_aruco* h, int frame, int nmarkers, int32_t* xy, int capacity, int32_t* offsets);

/* detect(image) -> markers sorted by id, corners refined by contour lines. Host pointers, one CV_8UC1 frame.
 * Frames up to 4095 pixels wide (adaptive-threshold windows up to 31, markerdetector_impl.cpp:3765-3809). */
int orbfe_aruco_detect(orbfe_aruco* h, const uint8_t* img, int rows, int cols, size_t step, orbfe_marker* out,
                       int capacity, int32_t* n_out);
/* detect() on a CV_8UC3 (BGR) frame (markerdetector_impl.cpp:5892: cvtColor(BGR2GRAY) first); `step` in bytes, >= 3 * cols. */
int orbfe_aruco_detect_bgr(orbfe_aruco* h, const uint8_t* bgr, int rows, int cols, size_t step, orbfe_marker* out, int capacity,
                           int32_t* n_out);
int orbfe_aruco_detect_batch(orbfe_aruco* h, const uint8_t* imgs, int nframes, size_t frame_stride, int rows, int cols,
                             size_t step, orbfe_marker* out, int capacity, int32_t* n_out);
int orbfe_aruco_detect_batch_device(orbfe_aruco* h, const uint8_t* d_imgs, int nframes, size_t frame_stride, int rows,
                                    int cols, size_t step, orbfe_marker* d_out, int capacity, int32_t* d_n_out,
                                    void* stream);
/* The batch entry point on device pointers cannot report a frame that exceeded the detector's internal capacities (more
 * than 1024 contours longer than 70 points in a frame that the LDS-resident contour kernels handle; the host-pointer
 * entry points retry such a batch with the big-frame kernel themselves).  After the batch: *nflagged = frames of the last
 * batch whose results are incomplete, *flags_or = the union of their capacity flags (synchronises the device).
 * orbfe_aruco_set_big_frames(h, 1) selects the big-frame contour kernel (bit image in HBM, 4096 kept contours) for all
 * following batches. */
int orbfe_aruco_batch_status(orbfe_aruco* h, int32_t* nflagged, int32_t* flags_or);
int orbfe_aruco_set_big_frames(orbfe_aruco* h, int on);
/* stage read-back for parity tests: 0 = thresholded image (rows x cols bytes, 0/255) of `frame`; 104 = the bit image the contour
 * kernels read from HBM: the thresholded image minus the specks that cannot have a border of more than 70 points where the speck
 * passes ran as a launch (the default in front of the one-workgroup kernels of a full batch), the thresholded image itself otherwise
 * -- also when the passes run INSIDE the relay kernels (ORBFE_ARUCO_SPECKS=2): their cleaned image only ever exists in LDS */
int orbfe_aruco_debug_image(orbfe_aruco* h, int frame, int stage, uint8_t* out);
/* As orbfe_extractor_set_aux_stream, for the detector's forked launches (the /2 pyramid). */
int orbfe_aruco_set_aux_stream(orbfe_aruco* h, void* stream);
int orbfe_aruco_debug_kernel_times(orbfe_aruco* h, float* out_us, int capacity);

/* ------------------------------------------------------------------ marker pose (IPPE) -- */
/* Marker pose: what MarkerDetector::detect adds to every marker when it is given camera parameters and a marker size
 * (markerdetector_impl.cpp:8720-8780 -> Marker::calculateExtrinsics, marker.cpp:322-344 -> aruco::solvePnP, ippe.cpp:91-100)
 * and what Frame.cc:170-174 asks for again to judge the ambiguity (aruco::solvePnP returning both IPPE solutions,
 * ippe.cpp:72-89): rvec/tvec = the solution with the lower reprojection error (Marker::Rvec / Tvec as CV_32F), rvec2/tvec2
 * the other one, err = their reprojection errors in pixels (Frame.cc:172: mvbArucoGood = err[0] / err[1] < 0.7).
 * Object frame: marker centre, corners (-s/2, s/2, 0) (s/2, s/2, 0) (s/2, -s/2, 0) (-s/2, -s/2, 0) (marker.cpp:358-369). */
typedef struct orbfe_marker_pose {
    float rvec[3], tvec[3];
    float rvec2[3], tvec2[3];
    float err[2];
} orbfe_marker_pose;

/* CameraParameters::resize (cameraparameters.cpp:158-173): detect() rescales the camera matrix when the image size is not
 * CamSize (markerdetector_impl.cpp:1110-1172; Frame.cc:132 fixes CamSize = 1280x720).  K4 = {fx, fy, cx, cy}. Host only. */
int orbfe_camera_resize(const float* K4, int cam_width, int cam_height, int img_width, int img_height, float* K4_out);

/* Poses of n markers (host pointers).  marker_size in metres (Frame.cc:131: 0.187); marker_size <= 0 or an empty camera
 * matrix is ORBFE_ERR_INVALID (the reference throws cv::Exception 9004, marker.cpp:325-331).  dist as orbfe_undistort_points. */
int orbfe_marker_poses(const orbfe_marker* markers, int n, float marker_size, const float* K4, const float* dist, int ndist,
                       orbfe_marker_pose* poses, int device);

/* The same over the detector's batch output on the device: frame f's first min(d_n[f], capacity) records of d_markers get
 * a pose in the same slot of d_poses.  d_n == NULL: all `capacity` records of every frame. */
int orbfe_marker_poses_batch_device(const orbfe_marker* d_markers, const int32_t* d_n, int capacity, int nframes,
                                    float marker_size, const float* K4, const float* dist, int ndist,
                                    orbfe_marker_pose* d_poses, void* stream);
/* detect(image, CameraParameters, markerSize) in ONE call (markerdetector.h:276-283: what Frame.cc:142 calls): the markers of
 * orbfe_aruco_detect plus poses[i] of markers[i] as orbfe_marker_poses would give them, with one upload, one wait and one set of
 * result copies instead of two of each.  K4 = {fx, fy, cx, cy} for THIS image size (orbfe_camera_resize when CamSize differs). */
int orbfe_aruco_detect_poses(orbfe_aruco* h, const uint8_t* img, int rows, int cols, size_t step, orbfe_marker* out,
                             orbfe_marker_pose* poses, int capacity, int32_t* n_out, float marker_size, const float* K4,
                             const float* dist, int ndist);
/* The same on a CV_8UC3 (BGR) frame. */
int orbfe_aruco_detect_poses_bgr(orbfe_aruco* h, const uint8_t* bgr, int rows, int cols, size_t step, orbfe_marker* out,
                                 orbfe_marker_pose* poses, int capacity, int32_t* n_out, float marker_size, const float* K4,
                                 const float* dist, int ndist);
/* cv::cornerSubPix(image, corners, Size(win, win), Size(-1, -1), TermCriteria(MAX_ITER | EPS, max_iters, eps)) on a CV_8UC1 image
 * (what the detector's CORNER_SUBPIX mode and its cornerUpsample run, markerdetector_impl.cpp:8511, :14182): pts = n (x, y) pairs,
 * refined in place; window half size 1 .. 8.  Host pointers. */
int orbfe_corner_subpix(const uint8_t* img, int rows, int cols, size_t step, float* pts, int n, int win, int max_iters, double eps,
                        int device);

/* ------------------------------------------------------------------ the batched-video mode -- */
/* What the reference's Tracking thread does per frame -- ORBextractor::operator() (Frame.cc:200-206), MarkerDetector::detect(image,
 * camera, 0.187) (Frame.cc:142), frame t matched against t-1 (Tracking.cc:531-532: all-pairs knn2 + SearchForInitialization) -- for
 * a STREAM of same-sized frames handed over in batches resident on the device (BASELINE north_star).  The pipeline
 * owns the engines (two extractor sets that alternate batches, phase-locked; one detector), its HIP streams and events, `record_sets`
 * result sets in rotation, and -- on N > 1 ranks -- the batch's one collective: the gather of the record set to rank `dst` by RCCL
 * send / recv on the pipeline's matching stream.  orbfe_pipeline_step only ENQUEUES (a few hundred microseconds of host time per
 * 300-frame batch) and returns the record set the batch is written to; a batch's matching (and gather) is enqueued one step late
 * where that shortens the step (frames up to 640 x 480), so call orbfe_pipeline_flush before waiting for the newest batch.
 *
 * A record set is ONE contiguous device buffer (orbfe_record_layout), array by array over the frames; the keypoint arrays carry one
 * slot in front of the batch (`halo` = 1): the last frame of the previous batch, so that the first frame of a batch is matched
 * against its predecessor in the stream too.  Frame f of the batch is slot f + halo.  Matching outputs of the newest batch: pair p
 * = slot p (queries, F1) against slot p + 1 (train, F2), p = 0 .. frames - 1; pair 0 is the pair across the batch boundary (no
 * keypoints on the F1 side for the first batch of a stream). */
typedef struct orbfe_pipeline_config {
    int32_t frames, rows, cols;        /* frames per batch; frame size */
    int32_t nfeatures, nlevels;        /* ORBextractor(nfeatures, scale_factor, nlevels, ini_th_fast, min_th_fast) */
    float scale_factor;
    int32_t ini_th_fast, min_th_fast;
    char dictionary[32];               /* MarkerDetector::setDictionary */
    int32_t device;
    int32_t marker_capacity;           /* marker (+ pose) records per frame in a record set (<= orbfe_aruco_max_markers) */
    int32_t use_orb, use_aruco;        /* 0 leaves an engine out (diagnostics) */
    float marker_size;                 /* metres (Frame.cc:131: 0.187) */
    float K[4], dist[12];              /* camera of the marker poses FOR THIS FRAME SIZE (orbfe_camera_resize), ndist coefficients */
    int32_t ndist;
    int32_t window_size;               /* SearchForInitialization: 100 */
    float nnratio;                     /* 0.9 */
    int32_t check_orientation;         /* 1 */
    /* scheduling; -1 = the measured default for the frame size: extractor engine sets, result sets in rotation,
     * phase lock of the engine sets (orbfe_extractor_follow stage), stage of the extractor's previous batch the detector's batch starts
     * behind, matching enqueued one step late, the detector's /2 pyramid in line on its stream */
    int32_t engine_sets, record_sets, phase_pin, det_pin, defer_post, det_nofork;
} orbfe_pipeline_config;
typedef struct orbfe_record_layout {
    int32_t frames, capacity, marker_capacity, halo;
    uint64_t off_kps, off_desc, off_n;                 /* (frames + halo) x capacity x 28 B | x 32 B | (frames + halo) x int32 */
    uint64_t off_markers, off_nmarkers, off_poses;     /* frames x marker_capacity x 36 B | frames x int32 | frames x marker_capacity x 56 B */
    uint64_t nbytes;
} orbfe_record_layout;

/* the defaults of BASELINE configs[1]: nfeatures 1000, 8 levels, 1.2, FAST 20 / 7, "ARUCO", 64 marker records, both engines, marker
 * size 0.187, TUM1 camera (Examples/Monocular/TUM1.yaml) rescaled from 1280 x 720 (Frame.cc:132), window 100, ratio 0.9; scheduling -1 */
int orbfe_pipeline_config_default(orbfe_pipeline_config* cfg, int frames, int rows, int cols);
orbfe_pipeline* orbfe_pipeline_create(const orbfe_pipeline_config* cfg);
void orbfe_pipeline_destroy(orbfe_pipeline* p);
int orbfe_pipeline_layout(const orbfe_pipeline* p, orbfe_record_layout* out);
/* One batch: d_imgs = frames x rows x pitch bytes on the pipeline's device (pitch >= cols; frames only have to stay valid until the
 * step's engines are done -- orbfe_pipeline_input_done).  *record_set = index of the set the batch is written to. */
int orbfe_pipeline_step(orbfe_pipeline* p, const uint8_t* d_imgs, size_t pitch, int32_t* record_set);
/* The same with the frames in HOST memory (rows of `step` bytes, frames contiguous; page-locked -- orbfe_host_alloc -- for the copy to
 * overlap): the pipeline uploads the batch into a ring of three device buffers on a copy stream of its own, ahead of the engines,
 * and copies every batch's record set back to page-locked host memory on a second copy stream behind the batch's post-work; both
 * overlap with the engines of the neighbouring batches.  h_imgs must stay valid until the upload is done (orbfe_pipeline_input_done).
 * orbfe_pipeline_host_records: the host copy of record set `set` (waits for its copy; valid until the set is written again, R steps on). */
int orbfe_pipeline_step_host(orbfe_pipeline* p, const uint8_t* h_imgs, size_t step, int32_t* record_set);
int orbfe_pipeline_host_records(orbfe_pipeline* p, int set, const uint8_t** h_records);
/* after steps from host memory: out[0] = microseconds the newest uploads took on the copy stream (mean over the input ring), out[1] =
 * the newest read-back of a record set.  Synchronises. */
int orbfe_pipeline_host_copy_us(orbfe_pipeline* p, float out[2]);
/* The HIP stream priority of the upload / read-back streams of orbfe_pipeline_step_host and the range the device offers (numerically
 * larger = lower).  They are created with the LOWEST one so that they come out of another pool of hardware queues than the engines'
 * streams; lowest == highest means the device has no second pool and uploads queue behind the engines (a warning is printed once). */
int orbfe_pipeline_copy_stream_priority(orbfe_pipeline* p, int* priority, int* lowest, int* highest);
void* orbfe_host_alloc(size_t bytes);                        /* page-locked host memory (hipHostMalloc) / NULL */
void orbfe_host_free(void* p);
/* Device memory for a caller without a HIP toolchain of its own (the Python wrapper, a host language over FFI): zeroed memory on
 * `device` / NULL, a blocking upload of `nrows` rows of `width` bytes (source rows `spitch` apart, destination rows `dpitch`), a
 * blocking download of `bytes`.  The pointers are ordinary HIP device pointers of the library's runtime. */
void* orbfe_device_alloc(int device, size_t bytes);
void orbfe_device_free(void* d);
int orbfe_device_upload_rows(void* d_dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t nrows);
int orbfe_device_download(void* dst, const void* d_src, size_t bytes);
int orbfe_pipeline_flush(orbfe_pipeline* p);                 /* enqueue the held-back post-work (matching, gather) of the newest batch */
int orbfe_pipeline_synchronize(orbfe_pipeline* p);           /* flush + wait for everything enqueued */
/* wait until the engines of the batch written to `record_set` have read their frames (the input buffer may be reused).  The engines
 * of a step are enqueued by the step itself, so nothing has to be flushed first. */
int orbfe_pipeline_input_done(orbfe_pipeline* p, int record_set);
/* capacity flags since the last call (synchronises): out[0] extractor overflow, [1] SearchForInitialization pool overflow (the pool
 * has been grown: repeat), [2] frames the detector flagged, [3] the union of their flags.  All zero = results complete. */
int orbfe_pipeline_status(orbfe_pipeline* p, int32_t out[4]);
int orbfe_pipeline_set_big_frames(orbfe_pipeline* p, int on); /* orbfe_aruco_set_big_frames on the pipeline's detector */
/* device pointers: record set `set`; the matching outputs of the newest batch ([frames][capacity] each, nmatches [frames]) */
int orbfe_pipeline_records(orbfe_pipeline* p, int set, uint8_t** d_records);
int orbfe_pipeline_matches(orbfe_pipeline* p, int32_t** d_best_idx, int32_t** d_best_dist, int32_t** d_second_dist, int32_t** d_matches12,
                           int32_t** d_nmatches);
/* a new stream: the next batch has no predecessor (the halo slots are emptied) */
int orbfe_pipeline_reset_stream(orbfe_pipeline* p);
/* the engines (for their debug entry points / kernel timers): extractor of engine set `set` (NULL past the last), the detector */
orbfe_extractor* orbfe_pipeline_extractor(orbfe_pipeline* p, int set);
orbfe_aruco* orbfe_pipeline_detector(orbfe_pipeline* p);
int orbfe_pipeline_engine_sets(const orbfe_pipeline* p, int32_t* engine_sets, int32_t* record_sets, int32_t* phase_pin, int32_t* det_pin,
                               int32_t* defer_post, int32_t* det_nofork);
/* launch timing of the matching / gather (HIP events on the matching stream; off by default).  timing_us: medians over the steps
 * since timing was switched on (the newest 64): out[0] knn2, [1] SearchForInitialization, [2] gather (0 without one); last != 0: the
 * newest step's instead.  Synchronises. */
int orbfe_pipeline_enable_timing(orbfe_pipeline* p, int on);
int orbfe_pipeline_timing_us(orbfe_pipeline* p, int last, float out[3]);
/* The ORBFE_* environment variables the pipeline and the engines read, with their defaults, as "NAME=default;NAME=default;..."
 * ("size" = decided by the frame size): what bench.py checks the environment against. */
const char* orbfe_pipeline_env_defaults(void);

/* Multi-GPU (one process per GPU): the batch's gather over RCCL (librccl is opened at run time; ORBFE_ERR_HIP when it is missing).
 * Either hand over a communicator the application owns (set_comm; ncclComm_t as void*), or let the pipeline create one:
 * comm_unique_id on one rank (ncclGetUniqueId, 128 bytes), the bytes sent to all ranks by the application's own means, then
 * comm_init on every rank (ncclCommInitRank).  From then on every batch's record set is gathered to rank `dst` behind the batch's
 * matching.  `dst` receives block r of rank r (its own too) in buffers allocated once: ONE SET OF BLOCKS PER RECORD SET -- the batch
 * written to record set s (orbfe_pipeline_step's *record_set) lands in block set s, which is written again record_sets batches later --
 * so a consumer on `dst` (one Tracking thread per stream, src/Tracking.cc:163-190) reads batch i while the following batches arrive:
 *     orbfe_pipeline_gathered_wait(p, s)          block until the gather of the batch last written to set s is complete (its deferred
 *                                                 post-work is flushed first if it was still held back); ORBFE_ERR_INVALID while no batch has
 *                                                 been gathered into s yet
 *     orbfe_pipeline_gathered_batch(p, s, &b)     which batch (step number of this pipeline, from 0) the newest gather into set s carries:
 *                                                 a consumer that expects batch i and reads b = i + record_sets has been overtaken
 *     orbfe_pipeline_gathered_set(p, s, r, &ptr)  rank r's block of set s (device pointer, layout = orbfe_pipeline_layout)
 *     orbfe_pipeline_gathered_release(p, s, st)   optional: the reads of set s enqueued on the consumer's HIP stream `st` so far (NULL: the
 *                                                 null stream) must finish before the set is received into again; without it the set is
 *                                                 simply overwritten record_sets batches later.  Call it BEFORE the step that writes set s
 *                                                 again is enqueued: a release that comes later delays the gather after that one instead
 * orbfe_pipeline_gathered(p, r, &ptr) = block r of the newest gather that was enqueued.  world == 1 runs the same branch with a send /
 * recv to itself.  ORBFE_RCCL_LIB names a library to bind in place of librccl (tests/fake_rccl.cpp). */
int orbfe_pipeline_comm_unique_id(uint8_t id[128]);
int orbfe_pipeline_comm_init(orbfe_pipeline* p, const uint8_t id[128], int rank, int world, int dst);
int orbfe_pipeline_set_comm(orbfe_pipeline* p, void* nccl_comm, int rank, int world, int dst);
int orbfe_pipeline_gathered(orbfe_pipeline* p, int rank, uint8_t** d_block);
int orbfe_pipeline_gathered_set(orbfe_pipeline* p, int record_set, int rank, uint8_t** d_block);
int orbfe_pipeline_gathered_wait(orbfe_pipeline* p, int record_set);
int orbfe_pipeline_gathered_release(orbfe_pipeline* p, int record_set, void* stream);
int orbfe_pipeline_gathered_batch(orbfe_pipeline* p, int record_set, long long* batch);
/* The transport operations one batch's gather consists of on rank `rank`, in issue order (csrc/gather_plan.hpp: the function the
 * pipeline itself executes with ncclRecv / ncclSend / a device copy).  Host logic only -- no device is touched -- so the N-rank control
 * flow (who receives what into which of the rotating blocks) can be driven over any transport: tests/test_multigpu_cpu.py runs it
 * between two gloo ranks on the CPU.  Returns the number of operations written to ops[0 .. capacity), or ORBFE_ERR_INVALID. */
#define ORBFE_GATHER_RECV 0      /* receive rank `peer`'s record set into the block buffer at byte `offset` */
#define ORBFE_GATHER_SEND 1      /* send this rank's record set to rank `peer` */
#define ORBFE_GATHER_COPY_OWN 2  /* this rank's own record set into the block buffer at byte `offset` */
typedef struct orbfe_gather_op { int32_t kind, peer; unsigned long long offset; } orbfe_gather_op;
int orbfe_pipeline_gather_plan(int rank, int world, int dst, int record_set, int record_sets, size_t nbytes, orbfe_gather_op* ops, int capacity);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ORBFE_H */
