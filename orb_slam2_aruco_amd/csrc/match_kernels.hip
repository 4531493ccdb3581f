// match_kernels.hip -- descriptor matching on gfx950 behind the C ABI (include/orbfe.h, "descriptor matching").
//
//   k_knn2_tiles / k_knn2_merge    all-pairs best / second-best Hamming with the reference update rule
//                                  (inner loop of every ORBmatcher::SearchBy*)
//   k_sfi_grid / _rows / _accept   ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:409-524) incl. the Frame
//                                  grid (src/Frame.cc:183-198, 280-345): a sort per frame, a wave per query for the candidate
//                                  rows, one wave per frame pair for the coupled accept loop
//
// Bound: integer VALU issue (XOR + v_bcnt_u32_b32), not HBM: 1000 x 1000 descriptors are 64 KB of traffic for
// 16 M lane-ops.  Train descriptors are staged through LDS and broadcast to all lanes of a wave.
#include <climits>

#include "orbfe_common.hpp"
#include "wave_dpp.hpp"

namespace orbfe {

__device__ __forceinline__ int hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1)
{
    int d = __popc(a0.x ^ b0.x);
    d += __popc(a0.y ^ b0.y);
    d += __popc(a0.z ^ b0.z);
    d += __popc(a0.w ^ b0.w);
    d += __popc(a1.x ^ b1.x);
    d += __popc(a1.y ^ b1.y);
    d += __popc(a1.z ^ b1.z);
    d += __popc(a1.w ^ b1.w);
    return d;
}

#define KNN_TILE 256

// grid: (ceil(max_nq/256), npairs, nsplit).  Split s handles train rows [s*chunk, (s+1)*chunk).
// Partial results go to part_* [pair][split][max_nq]; with nsplit == 1 they are the final arrays.
__global__ __launch_bounds__(256) void k_knn2_tiles(const uint8_t* __restrict__ Q, const int32_t* __restrict__ nq_arr,
                                                    size_t q_stride, int max_nq, const uint8_t* __restrict__ T,
                                                    const int32_t* __restrict__ nt_arr, size_t t_stride, int chunk,
                                                    int init, int32_t* __restrict__ best_idx,
                                                    int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist)
{
    __shared__ uint4 st[KNN_TILE * 2];
    const int pair = blockIdx.y, split = blockIdx.z, nsplit = gridDim.z;
    const int nq = nq_arr ? nq_arr[pair] : max_nq;
    const int nt = nt_arr[pair];
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= nq) return;
    const uint4* Qp = reinterpret_cast<const uint4*>(Q + (size_t)pair * q_stride);
    const uint4* Tp = reinterpret_cast<const uint4*>(T + (size_t)pair * t_stride);
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (q < nq) { a0 = Qp[2 * q]; a1 = Qp[2 * q + 1]; }
    int best = init, second = init, idx = -1;
    const int t_begin = split * chunk, t_end = min(nt, t_begin + chunk);
    for (int t0 = t_begin; t0 < t_end; t0 += KNN_TILE) {
        const int m = min(KNN_TILE, t_end - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * m; i += 256) st[i] = Tp[2 * t0 + i];
        __syncthreads();
        for (int t = 0; t < m; t++) {
            const int d = hamming256(a0, a1, st[2 * t], st[2 * t + 1]);
            if (d < best) { second = best; best = d; idx = t0 + t; }
            else if (d < second) second = d;
        }
    }
    if (q < nq) {
        const size_t o = ((size_t)pair * nsplit + split) * max_nq + q;
        best_idx[o] = idx; best_dist[o] = best; second_dist[o] = second;
    }
}

// ---- all-pairs Hamming distances on the matrix cores.
// For bit vectors a, b:  |a ^ b| = |a| + |b| - 2 a.b, and a.b over 256 bits is an int8 dot product of the bits spread to
// bytes -- an (nq x 256) x (256 x nt) GEMM, which is what MFMA is for.  v_mfma_i32_32x32x32_i8 does a 32 x 32 tile of
// dot products over 32 bits per instruction; eight of them finish a tile.  Per pair the VALU only has to fold the
// result into the running (best, second) keys: one mad, one med3, one min.
//   workgroup = KM_WAVES waves x 32 queries; the train descriptors are spread to bytes once per workgroup and chunk
//   (KM_CHUNK descriptors) in LDS and shared by all waves; the queries' A fragments stay in registers.
// Operand layout: lane l supplies row/column l & 31 and the 16 k-values of half l >> 5; A and B are loaded with the
// same rule, so the order of the k-values inside the instruction does not matter (a.b is a sum over k).
// C layout (cdna4 ISA, 32 x 32): column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
#define KM_WAVES 8
#define KM_CHUNK 128
#define KM_ROWB 272 // bytes per spread descriptor in LDS: 256 + 16 keeps the 16-byte reads of adjacent rows on distinct banks
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// middle value of three (the compiler emits v_med3_i32)
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
// 4 bits -> 4 bytes (bit i -> byte i)
__device__ __forceinline__ int spread4(uint32_t x) { return (int)(__umul24(x & 15u, 0x00204081u) & 0x01010101u); }
// x * 255 for words of 0 / 1 bytes as (x << 8) - x, the shift written as the instruction (the compiler folds the pair back into a
// v_mul_lo_u32 by 255: 5.3 cycles per wave against 2 x 3.4 - 4.5 for the pair, tools/valu_rate.hip -- a wash, kept for the shorter chain).
__device__ __forceinline__ int x255_1(int x)
{
    int t;
    asm("v_lshlrev_b32 %0, 8, %1" : "=v"(t) : "v"(x));
    return t - x;
}
__device__ __forceinline__ v4i x255(v4i p)
{
    v4i r;
    r.x = x255_1(p.x); r.y = x255_1(p.y); r.z = x255_1(p.z); r.w = x255_1(p.w);
    return r;
}
__device__ __forceinline__ v4i spread16(uint32_t bits)
{
    v4i r;
    r.x = spread4(bits); r.y = spread4(bits >> 4); r.z = spread4(bits >> 8); r.w = spread4(bits >> 12);
    return r;
}

__global__ __launch_bounds__(KM_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_knn2_mfma(const uint8_t* __restrict__ Q, const int32_t* __restrict__ nq_arr,
                                                             size_t q_stride, int max_nq, const uint8_t* __restrict__ T,
                                                             const int32_t* __restrict__ nt_arr, size_t t_stride, int chunk,
                                                             int init, int32_t* __restrict__ best_idx,
                                                             int32_t* __restrict__ best_dist,
                                                             int32_t* __restrict__ second_dist)
{
    // grid: (query tiles, pairs, nsplit).  Split z handles train rows [z * chunk, (z + 1) * chunk) and writes partials
    // [pair][split][max_nq] for k_knn2_merge, exactly as k_knn2_tiles does; with one split they are the final arrays.
    // the spread train descriptors of the current chunk and, after the loop, the cross-lane merge of the keys share one region:
    // 35 KB per workgroup instead of 69 (the footprint is a cost of its own: it decides what else fits on the CU)
    constexpr int KM_LDS = KM_CHUNK * KM_ROWB > KM_WAVES * 32 * 33 * 4 ? KM_CHUNK * KM_ROWB : KM_WAVES * 32 * 33 * 4;
    __shared__ __align__(16) unsigned char s_u[KM_LDS];
    unsigned char* s_t = s_u;
    int (*s_red)[32][33] = reinterpret_cast<int (*)[32][33]>(s_u);
    __shared__ int s_pt[KM_CHUNK];                                  // their popcounts
    const int pair = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nq = nq_arr ? nq_arr[pair] : max_nq;
    const int split = blockIdx.z, nsplit = gridDim.z;
    const int t_begin = split * chunk, nt = min(nt_arr[pair], t_begin + chunk);
    const int q0 = blockIdx.x * (KM_WAVES * 32);
    if (q0 >= nq) return;
    const uint32_t* Qp = reinterpret_cast<const uint32_t*>(Q + (size_t)pair * q_stride);
    const uint32_t* Tp = reinterpret_cast<const uint32_t*>(T + (size_t)pair * t_stride);
    const int col = lane & 31, half = lane >> 5;
    const int ie = min(init, 257); // distances are <= 256: any larger start value behaves like 257

    // A fragments: query row (q0 + 32 wid + col), k-step s = dword s of the descriptor, this lane's 16-bit half.  A set bit is
    // the byte -1, so the accumulator holds -(q.t) and a pair's key is ONE v_lshl_add_u32 (with +1 bytes the compiler needs a
    // shift and a subtraction)
    const int qa = q0 + 32 * wid + col;
    v4i a[8];
    {
        const uint4* qrow = reinterpret_cast<const uint4*>(Qp) + (size_t)min(qa, nq - 1) * 2; // clamped, not predicated
        const uint4 d0 = qrow[0], d1 = qrow[1];
        const uint32_t w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) {
            const v4i p = spread16((qa < nq ? w[s2] : 0u) >> (16 * half));
            a[s2] = x255(p); // 0x01 -> 0xff per byte, no carries
        }
    }
    // the 16 rows this lane accumulates: row(reg) = (reg & 3) + 8 * (reg >> 2) + 4 * half.  Keys are ((|t| - 2 q.t) << 16) | j: the
    // order of a row's candidates by (distance, index) without the row's own |q|; "accepted iff d < init" is applied when the
    // row is written, so the running keys simply start above every real key
    int bestk[16], secondk[16];
#pragma unroll
    for (int r = 0; r < 16; r++) bestk[r] = secondk[r] = 0x7fffffff;

    // A thread's part of a chunk: dwords 2 p, 2 p + 1 of train row (tid >> 2), p = tid & 3.  The NEXT chunk's part is loaded
    // before the current chunk's tiles are computed and waited for after them: global latency (and with two workgroups per CU
    // in the same rhythm there is nothing else to hide it) is off the chunk's critical path.
    const int ctr = tid >> 2, cpart = tid & 3;
    const uint2* Tp2 = reinterpret_cast<const uint2*>(Tp);
    uint2 nxt = make_uint2(0u, 0u);
    if (t_begin < nt) nxt = Tp2[(size_t)min(t_begin + ctr, nt - 1) * 4 + cpart];
    for (int t0 = t_begin; t0 < nt; t0 += KM_CHUNK) {
        const int m = min(KM_CHUNK, nt - t0);
        uint2 w = nxt;
        if (ctr >= m) w = make_uint2(0u, 0u); // rows past the end: zero bytes, popcount 0
        __syncthreads();                      // every wave is done with the previous chunk
        {
            v4i* dst = reinterpret_cast<v4i*>(s_t + ctr * KM_ROWB + cpart * 64);
            dst[0] = spread16(w.x);
            dst[1] = spread16(w.x >> 16);
            dst[2] = spread16(w.y);
            dst[3] = spread16(w.y >> 16);
            int pc = __popc(w.x) + __popc(w.y);                // |t|: sum over the row's four threads (one quad)
            pc += ORBFE_DPP(0, pc, 0xb1, 0xf);                 // quad_perm [1, 0, 3, 2]
            pc += ORBFE_DPP(0, pc, 0x4e, 0xf);                 // quad_perm [2, 3, 0, 1]
            if (cpart == 0) s_pt[ctr] = pc;
        }
        if (t0 + KM_CHUNK < nt) nxt = Tp2[(size_t)min(t0 + KM_CHUNK + ctr, nt - 1) * 4 + cpart];
        __syncthreads();
        // all four tiles of the chunk, straight-line (rows past the end are zero bytes and get the never-accepted key): one basic
        // block, so that a tile's matrix instructions can be scheduled under the previous tile's key updates
#pragma unroll
        for (int tt = 0; tt < KM_CHUNK; tt += 32) {
            v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const unsigned char* brow = s_t + (tt + col) * KM_ROWB + 16 * half;
            v4i b[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; s2++) b[s2] = *reinterpret_cast<const v4i*>(brow + 32 * s2); // all eight reads in flight
            const int pt = s_pt[tt + col];
#pragma unroll
            for (int s2 = 0; s2 < 8; s2++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s2], b[s2], acc, 0, 0, 0);
            // this lane's column is train t0 + tt + col; a column past the end has zero bytes (acc = 0) and a key above every real one
            const uint32_t cj = (tt + col < m) ? (((uint32_t)pt << 16) | (uint32_t)(t0 + tt + col)) : 0x7fffffffu;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int key = (int)(((uint32_t)acc[r] << 17) + cj); // ((|t| - 2 q.t) << 16) | j
                secondk[r] = med3i(bestk[r], secondk[r], key);       // middle of the three
                bestk[r] = min(bestk[r], key);
            }
        }
    }
    // merge the 32 column classes of every row through LDS: s_red[wave][row][column class]
    __syncthreads(); // every wave has read the last chunk: the region changes hands
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; r++) s_red[wid][(r & 3) + 8 * (r >> 2) + 4 * half][col] = pass == 0 ? bestk[r] : secondk[r];
        __builtin_amdgcn_wave_barrier();
        // lane = (row = col, half h): scans 16 column classes
        int b1 = 0x7fffffff, b2 = 0x7fffffff; // two smallest keys of the scanned entries
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int key = s_red[wid][col][16 * half + c];
            b2 = med3i(b1, b2, key);
            b1 = min(b1, key);
        }
        // combine the two halves of the row (lanes l and l ^ 32)
        const int o1 = __shfl_xor(b1, 32), o2 = __shfl_xor(b2, 32);
        const int n1 = min(b1, o1);
        const int n2 = min(max(b1, o1), min(b2, o2));
        if (pass == 0) { bestk[0] = n1; bestk[1] = n2; }       // two smallest "best" keys of the row
        else { secondk[0] = n1; }                              // smallest "second" key of the row
    }
    // row's best = smallest best key; row's second = min(second smallest best key, smallest second key)
    if (half == 0) {
        const int qr = q0 + 32 * wid + col;
        if (qr < nq) {
            uint4 d0 = reinterpret_cast<const uint4*>(Qp)[(size_t)qr * 2], d1 = reinterpret_cast<const uint4*>(Qp)[(size_t)qr * 2 + 1];
            const int pq = __popc(d0.x) + __popc(d0.y) + __popc(d0.z) + __popc(d0.w) + __popc(d1.x) + __popc(d1.y) + __popc(d1.z) + __popc(d1.w);
            const int kb = bestk[0], ks = min(bestk[1], secondk[0]);
            int bd = (kb >> 16) + pq, sd = (ks >> 16) + pq, bi = kb & 0xffff;
            if (bd >= ie) { bd = init; bi = -1; }
            if (sd >= ie) sd = init;
            const size_t o = ((size_t)pair * nsplit + split) * max_nq + qr;
            best_idx[o] = bi; best_dist[o] = bd; second_dist[o] = sd;
        }
    }
}

// Merge the per-split partials in ascending split order (earlier candidates win ties, as in the serial loop).
__global__ void k_knn2_merge(const int32_t* __restrict__ pidx, const int32_t* __restrict__ pbest,
                             const int32_t* __restrict__ psecond, int nsplit, int max_nq,
                             const int32_t* __restrict__ nq_arr, int32_t* __restrict__ best_idx,
                             int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist)
{
    const int pair = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int nq = nq_arr ? nq_arr[pair] : max_nq;
    if (q >= nq) return;
    int best = INT_MAX, second = INT_MAX, idx = -1;
    for (int s = 0; s < nsplit; s++) {
        const size_t o = ((size_t)pair * nsplit + s) * max_nq + q;
        const int b = pbest[o], sc = psecond[o], i = pidx[o];
        if (s == 0) { best = b; second = sc; idx = i; continue; }
        if (b < best) { second = min(best, sc); best = b; idx = i; }
        else second = min(second, b);
    }
    const size_t o = (size_t)pair * max_nq + q;
    best_idx[o] = idx; best_dist[o] = best; second_dist[o] = second;
}

// ---------------------------------------------------------------------------- SearchForInitialization ----------
#define GRID_COLS 64
#define GRID_ROWS 48


// One workgroup (256 threads) per frame pair p: F1 = frame p, F2 = frame p+1 of a stream.
// Phase A: F2's level-0 keypoints sorted by (grid column, grid row, index) in LDS.  That is exactly the order in which
//   Frame::GetFeaturesInArea walks the 64x48 grid (ix outer, iy inner, push_back order inside a cell), and
//   SearchForInitialization only ever asks for level 0 (minLevel = maxLevel = 0), so a query's candidate list is the
//   sub-sequence of this list whose cell lies in the query's cell rectangle and which passes the |dx|,|dy| < r test.
// Phase B (parallel, one wave per query): candidate indices + Hamming distances into a fixed-stride scratch row.
// Phase C (wave 0, serial over i1 as in the reference because vMatchedDistance couples the queries): best /
//   second-best with the "already matched better" skip, ratio + TH_LOW tests, mutual-uniqueness bookkeeping,
//   rotation histogram, ComputeThreeMaxima, final vbPrevMatched update.
#define SFI_MAXL0 1024
// Three launches (round 3; before: one 1024-thread workgroup + 55 KB of LDS per frame pair, fifteen of its sixteen waves parked
// while one wave ran the serial loop -- 146 us alone, 640 us next to the other engines' kernels):
//   k_sfi_grid    workgroup per FRAME: what a frame contributes in either role.  As F2: its level-0 keypoints inside the Frame
//                 grid sorted by (grid column, grid row, index) -- the order in which GetFeaturesInArea walks the 64 x 48 grid
//                 (Frame.cc:183-198, :280-345) -- with position, angle and descriptor laid out by rank; as F1: its level-0
//                 keypoints in index order (the queries).  A frame of a batch is F2 of one pair and F1 of the next.
//   k_sfi_rows    wave per query, all pairs of the batch in one grid (a throughput kernel): window test against F2's sorted
//                 list, Hamming distance of every candidate, one contiguous row (distance << 16 | rank, candidate order) per
//                 query in the pair's pool; rows are placed by an atomic cursor, so a pool is dense and any order.
//   k_sfi_accept  ONE WAVE per pair (64-thread workgroups): the reference's serial loop -- the "already matched better" skip and
//                 the mutual-uniqueness bookkeeping couple the queries (ORBmatcher.cc:448-449, :467-475) -- out of LDS, then the
//                 rotation histogram, ComputeThreeMaxima and the vbPrevMatched update.
struct SfiGrid { // [frame][SFI_MAXL0] arrays in HBM, written by k_sfi_grid
    int32_t* nl0;     // [frames] level-0 keypoints inside the grid (role F2), clamped to SFI_MAXL0
    int32_t* nq;      // [frames] level-0 keypoints (role F1), clamped to SFI_MAXL0
    uint32_t* sorted; // (cell << 16) | index, ascending
    float2* xy;       // by rank
    float* ang;       // by rank
    uint4* desc;      // by rank, two per keypoint
    uint16_t* query;  // by query position: keypoint index
    float2* qxy;      // by query position: the keypoint's own position (the window centre unless vbPrevMatched is given)
    int32_t* cursor;  // [pairs * SFI_CURSOR_PAD] next free pool entry (zeroed here, advanced by k_sfi_rows); one per 256 bytes, so that
                      //   the 65 k atomics of a batch do not queue on a handful of cache lines of one L2 channel
};
#define SFI_GRID_THREADS 256
#define SFI_CURSOR_PAD 64
#define SFI_ROWS_BX 64         // workgroups per pair in k_sfi_rows (4 waves each, one query per wave and trip)
#define SFI_POOL_LDS 6144      // pool entries k_sfi_accept stages in LDS (24 KB); later ones are read from the pool in HBM
#define SFI_ROW_BITS 11        // row record = offset << 11 | count (count <= SFI_MAXL0 = 1024)

__global__ __launch_bounds__(SFI_GRID_THREADS) void k_sfi_grid(const orbfe_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                               const int32_t* __restrict__ nkp, int capacity, int nframes, float4 bnd,
                                                               SfiGrid G, int32_t* __restrict__ overflow)
{
    __shared__ __align__(16) uint32_t s_key[SFI_MAXL0 + 4];
    __shared__ int s_nl0;
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const orbfe_keypoint* k = kps + (size_t)f * capacity;
    const uint8_t* d = desc + (size_t)f * capacity * 32;
    const int n = nkp[f];
    // bounds of the undistorted image (Frame::ComputeImageBounds; 0, 0, cols, rows without distortion) and Frame.cc:112-113
    const float mnMinX = bnd.x, mnMinY = bnd.y;
    const float invW = __fdiv_rn((float)GRID_COLS, bnd.z - bnd.x);
    const float invH = __fdiv_rn((float)GRID_ROWS, bnd.w - bnd.y);
    if (tid == 0) { s_nl0 = 0; if (f < nframes - 1) G.cursor[(size_t)f * SFI_CURSOR_PAD] = 0; }
    __syncthreads();
    // role F2: level-0 keypoints that fall inside the grid (Frame.cc:183-198, :335-345)
    for (int i = tid; i < n; i += SFI_GRID_THREADS) {
        const orbfe_keypoint kp = k[i];
        if (kp.octave != 0) continue;
        const int px = (int)roundf(__fmul_rn(kp.x - mnMinX, invW));
        const int py = (int)roundf(__fmul_rn(kp.y - mnMinY, invH));
        if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) {
            const int e = atomicAdd(&s_nl0, 1);
            if (e < SFI_MAXL0) s_key[e] = ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i;
        }
    }
    // role F1: level-0 keypoints in index order (ORBmatcher.cc:425-428), wave 0 by ballots
    if (wid == 0) {
        int nq = 0;
        uint16_t* qo = G.query + (size_t)f * SFI_MAXL0;
        float2* qp = G.qxy + (size_t)f * SFI_MAXL0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            orbfe_keypoint kp;
            kp.octave = 1;
            if (i < n) kp = k[i];
            const bool isq = i < n && kp.octave <= 0;
            const unsigned long long m = __ballot(isq);
            const int pos = nq + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            if (isq && pos < SFI_MAXL0) { qo[pos] = (uint16_t)i; qp[pos] = make_float2(kp.x, kp.y); }
            nq += __popcll(m);
        }
        if (lane == 0) {
            if (nq > SFI_MAXL0) { atomicMax(overflow, nq); nq = SFI_MAXL0; }
            G.nq[f] = nq;
        }
    }
    __syncthreads();
    int nl0 = s_nl0;
    if (nl0 > SFI_MAXL0) { if (tid == 0) atomicMax(overflow, nl0); nl0 = SFI_MAXL0; }
    if (tid == 0) G.nl0[f] = nl0;
    if (tid < 4) s_key[nl0 + tid] = 0xffffffffu; // the counting loop reads whole uint4s
    __syncthreads();
    // sort by counting: the keys are distinct (the index is part of them), so an entry's rank is the number of smaller keys --
    // broadcast LDS reads, no barrier; the rank-ordered record goes straight to HBM
    const uint4* k4 = reinterpret_cast<const uint4*>(s_key);
    for (int e = tid; e < nl0; e += SFI_GRID_THREADS) {
        const uint32_t key = s_key[e];
        int r = 0;
        for (int j = 0; j < (nl0 + 3) / 4; j++) {
            const uint4 v = k4[j];
            r += (v.x < key) + (v.y < key) + (v.z < key) + (v.w < key);
        }
        const int i2 = key & 0xffff;
        const orbfe_keypoint kp = k[i2];
        const size_t o = (size_t)f * SFI_MAXL0 + r;
        G.sorted[o] = key;
        G.xy[o] = make_float2(kp.x, kp.y);
        G.ang[o] = kp.angle;
        G.desc[2 * o] = reinterpret_cast<const uint4*>(d)[2 * i2];
        G.desc[2 * o + 1] = reinterpret_cast<const uint4*>(d)[2 * i2 + 1];
    }
}

// Candidates of one query among four 64-entry chunks (c0 .. c0 + 3) of F2's sorted list: window test, then the Hamming distances
// with all loads of the four chunks in flight together.  keys[k] = distance << 20 | position in the row << 10 | rank (0x7fffffff:
// no candidate); returns the number of candidates found (wave-uniform).
struct SfiWindow { float x, y, r; int minx, maxx, miny, maxy; };
__device__ __forceinline__ int sfi_chunks4(const SfiGrid& G, size_t g2, int nl0, int c0, int lane, const SfiWindow& W, const uint4& a0,
                                           const uint4& a1, int base, int* keys)
{
    uint32_t cell[4];
    float2 xy[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = (c0 + k) * 64 + lane;
        cell[k] = 0xffffffffu; xy[k] = make_float2(0.f, 0.f);
        if (j < nl0) { cell[k] = G.sorted[g2 + j] >> 16; xy[k] = G.xy[g2 + j]; }
    }
    bool ok[4];
    uint4 b0[4], b1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // Frame::GetFeaturesInArea (Frame.cc:280-333): the cell range of the window, then |dx| < r and |dy| < r
        const int cx = (int)(cell[k] / GRID_ROWS), cy = (int)(cell[k] - (uint32_t)cx * GRID_ROWS);
        ok[k] = cell[k] != 0xffffffffu && cx >= W.minx && cx <= W.maxx && cy >= W.miny && cy <= W.maxy && fabsf(xy[k].x - W.x) < W.r &&
                fabsf(xy[k].y - W.y) < W.r;
        if (ok[k]) { const size_t o = 2 * (g2 + (size_t)((c0 + k) * 64 + lane)); b0[k] = G.desc[o]; b1[k] = G.desc[o + 1]; }
    }
    int n = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned long long m = __ballot(ok[k]);
        keys[k] = 0x7fffffff;
        if (ok[k]) {
            const int pos = base + n + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
            const int dd = hamming256(a0, a1, b0[k], b1[k]); // 256 only for exact complements; 255 is far above TH_LOW as well
            keys[k] = (min(dd, 255) << 20) | (pos << 10) | ((c0 + k) * 64 + lane);
        }
        n += (int)__popcll(m);
    }
    return n;
}

#define SFI_ROWS_WAVES 4
__global__ __launch_bounds__(SFI_ROWS_WAVES * 64) void k_sfi_rows(const orbfe_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                                  int capacity, float4 bnd, float window, const float* __restrict__ prev_in,
                                                                  SfiGrid G, uint32_t* __restrict__ pool, int pool_cap,
                                                                  uint32_t* __restrict__ rowrec, int32_t* __restrict__ overflow)
{
    __shared__ int s_keys[SFI_ROWS_WAVES][SFI_MAXL0]; // the wave's row, by candidate position, for the rank-by-counting sort
    const int p = blockIdx.y, lane = threadIdx.x & 63, wid = wave_id();
    const int nq = G.nq[p], nl0 = G.nl0[p + 1];
    const size_t g1 = (size_t)p * SFI_MAXL0, g2 = (size_t)(p + 1) * SFI_MAXL0;
    const uint8_t* d1 = desc + (size_t)p * capacity * 32;
    const float* prev = prev_in ? prev_in + (size_t)p * capacity * 2 : nullptr;
    uint32_t* pl = pool + (size_t)p * pool_cap;
    const float mnMinX = bnd.x, mnMinY = bnd.y;
    const float invW = __fdiv_rn((float)GRID_COLS, bnd.z - bnd.x);
    const float invH = __fdiv_rn((float)GRID_ROWS, bnd.w - bnd.y);
    int* row = s_keys[wid];
    (void)kps;
    for (int q = blockIdx.x * SFI_ROWS_WAVES + wid; q < nq; q += gridDim.x * SFI_ROWS_WAVES) {
        const int i1 = __builtin_amdgcn_readfirstlane((int)G.query[g1 + q]);
        float2 c = G.qxy[g1 + q];
        if (prev) c = make_float2(prev[2 * i1], prev[2 * i1 + 1]);
        SfiWindow W;
        W.x = c.x; W.y = c.y; W.r = window;
        W.minx = max(0, (int)floorf(__fmul_rn(c.x - mnMinX - W.r, invW)));
        W.maxx = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(c.x - mnMinX + W.r, invW)));
        W.miny = max(0, (int)floorf(__fmul_rn(c.y - mnMinY - W.r, invH)));
        W.maxy = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(c.y - mnMinY + W.r, invH)));
        const bool any = !(W.minx >= GRID_COLS || W.maxx < 0 || W.miny >= GRID_ROWS || W.maxy < 0);
        const uint4 a0 = reinterpret_cast<const uint4*>(d1)[2 * i1];
        const uint4 a1 = reinterpret_cast<const uint4*>(d1)[2 * i1 + 1];
        int count = 0;
        if (any)
            for (int c0 = 0; c0 * 64 < nl0; c0 += 4) { // the row in candidate order, wave-private LDS
                int keys[4];
                const int n = sfi_chunks4(G, g2, nl0, c0, lane, W, a0, a1, count, keys);
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (keys[k] != 0x7fffffff) row[(keys[k] >> 10) & 0x3ff] = keys[k];
                count += n;
            }
        int off = 0;
        if (lane == 0 && count > 0) off = atomicAdd(&G.cursor[(size_t)p * SFI_CURSOR_PAD], count);
        // sorted by (distance, position): the accept loop then takes the first two candidates that are not skipped instead of
        // reducing the row.  Rank by counting (the keys are distinct), broadcast LDS reads; the first 64 entries are ranked while
        // the cursor's atomic is in flight.
        auto rank_of = [&](int e, int& key) {
            key = e < count ? row[e] : 0x7fffffff;
            int r = 0;
            for (int t = 0; t < count; t++) r += row[t] < key;
            return r;
        };
        int key0 = 0x7fffffff;
        const int r0 = count > 0 ? rank_of(lane, key0) : 0;
        off = __builtin_amdgcn_readfirstlane(off);
        bool fits = true;
        if (count > 0 && off + count > pool_cap) { // the pool is too small: flagged, the host grows it and the batch is repeated
            if (lane == 0) atomicMax(overflow + 1, off + count);
            fits = false;
        }
        if (lane == 0) rowrec[g1 + q] = fits ? ((uint32_t)off << SFI_ROW_BITS) | (uint32_t)count : 0u;
        if (count == 0 || !fits) continue;
        if (lane < count) pl[off + r0] = ((uint32_t)(key0 >> 20) << 16) | (uint32_t)(key0 & 0x3ff);
        for (int e0 = 64; e0 < count; e0 += 64) {
            int key;
            const int r = rank_of(e0 + lane, key);
            if (e0 + lane < count) pl[off + r] = ((uint32_t)(key >> 20) << 16) | (uint32_t)(key & 0x3ff);
        }
    }
}

// The serial loop of k_sfi_accept (see there).  IN_LDS: the pair's whole pool is staged in LDS -- no global path, hence no
// branch and no vmcnt wait inside the loop.
template <bool IN_LDS>
__device__ __forceinline__ void sfi_accept_loop(int nq, int lane, float nnratio, const uint32_t* s_pool, uint32_t* s_state,
                                                const uint32_t* s_row, uint16_t* s_acc, uint16_t* s_held, const uint32_t* __restrict__ pl)
{
    constexpr int NIL = 0xffff;
    auto rfl = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    constexpr uint32_t CNT = (1u << SFI_ROW_BITS) - 1;
    auto entry = [&](int o) { return IN_LDS || o < SFI_POOL_LDS ? s_pool[min(o, SFI_POOL_LDS - 1)] : pl[o]; };
    // rows longer than a wave whose first 64 entries did not settle best and second: the later chunks, in order
    auto long_row = [&](int e, int off, int& bestDist, int& bestRank, uint32_t& bestState, int& bestDist2) {
        for (int j0 = 64; j0 < e; j0 += 64) {
            const uint32_t en = j0 + lane < e ? entry(off + j0 + lane) : 0u;
            const int rk = (int)(en & 0x3ffu), dd = (int)(en >> 16);
            const uint32_t st = s_state[rk];
            unsigned long long open = __ballot(j0 + lane < e && (int)(st >> 16) > dd);
            if (open && bestDist < 0) {
                const int b = __builtin_ctzll(open);
                bestDist = __builtin_amdgcn_readlane(dd, b);
                bestRank = __builtin_amdgcn_readlane(rk, b);
                bestState = (uint32_t)__builtin_amdgcn_readlane((int)st, b);
                open &= open - 1;
            }
            if (open) { bestDist2 = __builtin_amdgcn_readlane(dd, __builtin_ctzll(open)); return; }
        }
    };
    uint32_t rr = nq > 0 ? rfl(s_row[0]) : 0u, rr1 = nq > 1 ? rfl(s_row[1]) : 0u;
    uint32_t ent = entry((int)(rr >> SFI_ROW_BITS) + lane);
    for (int q = 0; q < nq; q++) {
        const int e = (int)(rr & CNT), off = (int)(rr >> SFI_ROW_BITS);
        const int rk = (int)(ent & 0x3ffu), dd = (int)(ent >> 16);
        const uint32_t st = s_state[rk];                                        // the loop-carried read
        const uint32_t ent1 = entry((int)(rr1 >> SFI_ROW_BITS) + lane);         // row q + 1
        const uint32_t rr2 = s_row[min(q + 2, SFI_MAXL0 - 1)];                  // record of row q + 2
        unsigned long long open = __ballot(lane < e && (int)(st >> 16) > dd);
        int bestDist = -1, bestRank = 0, bestDist2 = INT_MAX;
        uint32_t bestState = 0;
        if (open) {
            const int b = __builtin_ctzll(open);
            bestDist = __builtin_amdgcn_readlane(dd, b);
            bestRank = __builtin_amdgcn_readlane(rk, b);
            bestState = (uint32_t)__builtin_amdgcn_readlane((int)st, b);
            open &= open - 1;
            if (open) bestDist2 = __builtin_amdgcn_readlane(dd, __builtin_ctzll(open));
        }
        if (e > 64 && !open) long_row(e, off, bestDist, bestRank, bestState, bestDist2);
        if (bestDist >= 0 && bestDist <= 50 && (float)bestDist < __fmul_rn((float)bestDist2, nnratio)) { // TH_LOW, ratio test
            if (lane == 0) {
                const int old = (int)(bestState & 0xffffu) - 1;
                if (old >= 0) s_held[old] = NIL;      // the earlier query loses the keypoint (:467-471)
                s_state[bestRank] = ((uint32_t)bestDist << 16) | (uint32_t)(q + 1);
                s_held[q] = (uint16_t)bestRank;
                s_acc[q] = (uint16_t)bestRank;
            }
            __builtin_amdgcn_wave_barrier();
        }
        rr = rr1; rr1 = q + 2 < nq ? rfl(rr2) : 0u; ent = ent1;
    }
}

__global__ __launch_bounds__(64) void k_sfi_accept(const orbfe_keypoint* __restrict__ kps, const int32_t* __restrict__ nkp, int capacity,
                                                   float nnratio, int check_ori, const float* __restrict__ prev_in,
                                                   float* __restrict__ prev_out, int32_t* __restrict__ matches12,
                                                   int32_t* __restrict__ nmatches_out, SfiGrid G, const uint32_t* __restrict__ pool,
                                                   int pool_cap, const uint32_t* __restrict__ rowrec)
{
    __builtin_amdgcn_s_setprio(2); // latency-bound: one wave per pair goes first when a VALU-bound kernel shares the CU
    __shared__ __align__(16) uint32_t s_pool[SFI_POOL_LDS];
    __shared__ uint32_t s_state[SFI_MAXL0]; // vMatchedDistance << 16 | (position of the query in vnMatches21 + 1), indexed by the RANK
                                            //   of an F2 keypoint (only level-0 keypoints of F2 inside the grid can ever be matched)
    __shared__ uint32_t s_row[SFI_MAXL0];   // per query: pool offset << 11 | candidates
    __shared__ uint16_t s_acc[SFI_MAXL0];   // per query: rank accepted with, or NIL
    __shared__ uint16_t s_held[SFI_MAXL0];  // per query: rank still held, or NIL
    __shared__ signed char s_rotbin[SFI_MAXL0]; // per query: histogram bin or -1
    __shared__ int s_hist[30];
    const int p = blockIdx.x, lane = threadIdx.x;
    const orbfe_keypoint* k1 = kps + (size_t)p * capacity;
    const orbfe_keypoint* k2 = kps + (size_t)(p + 1) * capacity;
    const int n1 = nkp[p];
    const int nq = G.nq[p], nl0 = G.nl0[p + 1];
    const size_t g1 = (size_t)p * SFI_MAXL0, g2 = (size_t)(p + 1) * SFI_MAXL0;
    const uint16_t* qi = G.query + g1;
    int32_t* m12 = matches12 + (size_t)p * capacity;
    const float* prev = prev_in ? prev_in + (size_t)p * capacity * 2 : nullptr;
    float* prevo = prev_out ? prev_out + (size_t)p * capacity * 2 : nullptr;
    const uint32_t* pl = pool + (size_t)p * pool_cap;
    constexpr int NIL = 0xffff;

    // the pair's candidate rows -> LDS (dense pool: a flat copy), the per-query and per-keypoint state.  All of it without
    // predicated loads -- a load under `if (i < n)` becomes a branch with a full wait behind it, one L2 round trip per trip -- the
    // buffers are sized so that whole batches may be read: the pool holds >= 16 K entries per pair (SFI_POOL_LDS are copied), the
    // row records SFI_MAXL0 per pair.
#ifdef ORBFE_SFI_TIMING // diagnosis build (tools/build_timing.sh): phase clocks of one pair, s_memrealtime ticks of 10 ns
    const unsigned long long t_0 = wall_clock64();
#endif
    const int total = min(G.cursor[(size_t)p * SFI_CURSOR_PAD], pool_cap);
    {
        const uint4* src = reinterpret_cast<const uint4*>(pl); // pool_cap is a multiple of 4
        uint4* dst = reinterpret_cast<uint4*>(s_pool);
        const int n4 = (min(total, SFI_POOL_LDS) + 3) / 4;
        static_assert(SFI_POOL_LDS % (4 * 64 * 8) == 0, "whole batches");
        for (int i0 = 0; i0 < n4; i0 += 64 * 8) { // eight loads in flight per lane
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = src[i0 + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 8; k++) dst[i0 + k * 64 + lane] = v[k];
        }
        for (int i0 = 0; i0 < nq; i0 += 64 * 4) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = rowrec[g1 + i0 + k * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + k * 64 + lane;
                s_row[i] = i < nq ? v[k] : 0u; s_acc[i] = NIL; s_held[i] = NIL; s_rotbin[i] = -1;
            }
        }
    }
    for (int i = lane; i < SFI_MAXL0; i += 64) s_state[i] = 0xffff0000u;
    if (lane < 30) s_hist[lane] = 0;
    for (int i = lane; i < n1; i += 64) m12[i] = -1;
    __builtin_amdgcn_wave_barrier();

    // ---- the serial matching loop (ORBmatcher.cc:423-490).  It carries only what couples the queries -- vMatchedDistance /
    // vnMatches21 per F2 keypoint and which query currently holds it; everything else (vnMatches12 in HBM, the rotation bins) is
    // written by all lanes after the loop from two per-query records: the rank a query was accepted with (rotHist keeps it even
    // if the match is stolen later, :470-488) and the rank it still holds.
    // A query's row is sorted by (distance, candidate position), so "best = first minimum among the candidates whose
    // vMatchedDistance exceeds their distance (:448-449), second = the next smallest" is: the first two candidates of the row that
    // are not skipped -- one gather of the keypoints' state words, one ballot, two scalar bit scans; no wave reduction on the
    // loop-carried chain.  One LDS round trip on that chain per query: the gather is issued first, the row of query q + 1 and the
    // row record of query q + 2 behind it (LDS answers in order, so the wait for the gather does not wait for them); every LDS
    // read of the loop is unconditional (clamped addresses), only the ballot looks at the row length.
    const float factor = 1.0f / 30; // HISTO_LENGTH; the upstream "1/30" quirk is kept (App. D)
#ifdef ORBFE_SFI_TIMING
    const unsigned long long t_1 = wall_clock64();
#endif
    if (total <= SFI_POOL_LDS) sfi_accept_loop<true>(nq, lane, nnratio, s_pool, s_state, s_row, s_acc, s_held, pl);
    else sfi_accept_loop<false>(nq, lane, nnratio, s_pool, s_state, s_row, s_acc, s_held, pl); // rows beyond the staged part of the pool
    __builtin_amdgcn_wave_barrier();
#ifdef ORBFE_SFI_TIMING
    const unsigned long long t_2 = wall_clock64();
#endif
    int nmatches = 0;
    for (int q0 = 0; q0 < nq; q0 += 4 * 64) {   // four queries a lane and trip: their index, match and angle loads in flight together
        int rr[4], ra[4], qidx[4];
        bool in[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = q0 + 64 * k + lane, qc = min(q, nq - 1);
            in[k] = q < nq;
            rr[k] = s_held[qc]; ra[k] = s_acc[qc];
            qidx[k] = qi[qc];
        }
        uint32_t srt[4];
        float a1[4], a2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            srt[k] = G.sorted[g2 + (rr[k] != NIL ? rr[k] : 0)];
            a1[k] = k1[qidx[k]].angle;
            a2[k] = G.ang[g2 + (ra[k] != NIL ? ra[k] : 0)];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int q = q0 + 64 * k + lane;
            const bool held = in[k] && rr[k] != NIL;
            if (held) m12[qidx[k]] = (int)(srt[k] & 0xffff);
            if (in[k] && check_ori && ra[k] != NIL) {
                float rot = a1[k] - a2[k];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(__fmul_rn(rot, factor));
                if (bin == 30) bin = 0;
                s_rotbin[q] = (signed char)bin; // rotHist[bin].push_back(i1): stays even if un-matched later
            }
            nmatches += (int)__popcll(__ballot(held));
        }
    }
    __threadfence_block();
    if (check_ori) {
        // histogram sizes count every push_back, including i1 whose match was later stolen (as in the reference)
        for (int i = lane; i < nq; i += 64)
            if (s_rotbin[i] >= 0) atomicAdd(&s_hist[s_rotbin[i]], 1);
        __builtin_amdgcn_wave_barrier();
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; i++) { // ComputeThreeMaxima, ORBmatcher.cc:1605-1646
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        int removed = 0;
        for (int q0 = 0; q0 < nq; q0 += 4 * 64) {
            int qidx[4];
#pragma unroll
            for (int k = 0; k < 4; k++) qidx[k] = qi[min(q0 + 64 * k + lane, nq - 1)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int q = q0 + 64 * k + lane;
                bool rm = false;
                if (q < nq) {
                    const int bin = s_rotbin[q];
                    rm = bin >= 0 && bin != ind1 && bin != ind2 && bin != ind3 && s_held[q] != NIL;
                    if (rm) m12[qidx[k]] = -1;
                }
                removed += (int)__popcll(__ballot(rm));
            }
        }
        nmatches -= removed;
    }
    __threadfence_block();
    if (prevo)
        for (int i0 = 0; i0 < n1; i0 += 4 * 64) {   // four entries a lane, their loads in flight together (they were a round trip each:
            float x[4], y[4];                        //   the match, then the keypoint it names, sixteen times over for one pair)
            int m[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = min(i0 + 64 * k + lane, n1 - 1);
                x[k] = prev ? prev[2 * i] : k1[i].x; y[k] = prev ? prev[2 * i + 1] : k1[i].y;
                m[k] = m12[i];
            }
            float mx[4], my[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int mm = max(m[k], 0); mx[k] = k2[mm].x; my[k] = k2[mm].y; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = i0 + 64 * k + lane;
                if (i < n1) {
                    prevo[2 * i] = m[k] >= 0 ? mx[k] : x[k];
                    prevo[2 * i + 1] = m[k] >= 0 ? my[k] : y[k];
                }
            }
        }
    if (lane == 0) nmatches_out[p] = nmatches;
#ifdef ORBFE_SFI_TIMING
    if (lane == 0 && (p == 7 || p == 150))
        printf("k_sfi_accept pair %d: nq %d nl0 %d pool %d | stage %llu loop %llu epilogue %llu (x 10 ns)\n", p, nq, nl0, total, t_1 - t_0,
               t_2 - t_1, wall_clock64() - t_2);
#endif
}

// ---------------------------------------------------------------------------- guided knn2 (CSR) ----------------
// Best / second-best of every query over ITS candidate list (CSR: offsets[nq + 1], idx[]) -- the inner loop of the guided
// searches whose candidates the caller builds (SearchByBoW node lists ORBmatcher.cc:159-292, Fuse, SearchForTriangulation).
// One wave per query; strict '<', the first candidate in list order wins ties.
__global__ __launch_bounds__(256) void k_knn2_csr(const uint8_t* __restrict__ Q, int nq, const uint8_t* __restrict__ T,
                                                  const int32_t* __restrict__ offsets, const int32_t* __restrict__ idx,
                                                  int init, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist,
                                                  int32_t* __restrict__ second_dist)
{
    const int lane = threadIdx.x & 63, q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    const uint4 a0 = reinterpret_cast<const uint4*>(Q)[2 * q], a1 = reinterpret_cast<const uint4*>(Q)[2 * q + 1];
    const int b = offsets[q], e = offsets[q + 1];
    unsigned long long bestk = ~0ull, secondk = ~0ull; // (distance << 32) | position in the list
    for (int j0 = b; j0 < e; j0 += 64) {
        const int j = j0 + lane;
        unsigned long long key = ~0ull;
        if (j < e) {
            const int t = idx[j];
            const int d = hamming256(a0, a1, reinterpret_cast<const uint4*>(T)[2 * t], reinterpret_cast<const uint4*>(T)[2 * t + 1]);
            if (d < init) key = ((unsigned long long)d << 32) | (unsigned)(j - b);
        }
        const unsigned long long m1 = wave_min_u64(key);
        const unsigned long long k2 = wave_min_u64(key == m1 ? ~0ull : key);
        if (m1 < bestk) { secondk = min(bestk, k2); bestk = m1; }
        else secondk = min(secondk, m1);
    }
    if (lane == 0) {
        best_idx[q] = bestk != ~0ull ? idx[b + (int)(bestk & 0xffffffffu)] : -1;
        best_dist[q] = bestk != ~0ull ? (int)(bestk >> 32) : init;
        second_dist[q] = secondk != ~0ull ? (int)(secondk >> 32) : init;
    }
}

// ---------------------------------------------------------------------------- SearchByProjection ---------------
// The matching loop of ORBmatcher::SearchByProjection (ORBmatcher.cc:45-129) on flat arrays; see include/orbfe.h.
// One workgroup (16 waves) per call.
//   A  the Frame grid (Frame.cc:183-198, :335-345): keypoints sorted by (cell, index); cell start offsets; per-rank
//      position / octave / taken flag in LDS.  Sorted order inside a run of cells of one grid column IS the order in
//      which GetFeaturesInArea (Frame.cc:280-333) walks them, so a query's candidates are one contiguous range per column.
//   B  one wave per query: candidates that pass the octave and window tests, with their Hamming distances, as a row
//      (rank u16, distance u8) in scratch, in candidate order.
//   C  mode 0: one wave per query resolves best / second-best over the row (strict '<', first candidate wins);
//      mode 1: ONE wave walks the queries in order, because an accepted keypoint is taken for the queries after it.
//      mode 2: the loop of SearchByProjection(CurrentFrame, LastFrame, th, mono) (ORBmatcher.cc:1355-1471): best only,
//              accept best <= th_high, the accepted keypoint is taken when the query's map point has observations, rotation
//              histogram + ComputeThreeMaxima; the result is per keypoint of the frame (match_cur[i2] = query or -1).
#define SBP_THREADS 1024
#define SBP_CELLS (GRID_COLS * GRID_ROWS)
struct SbpQuery { float x, y, r; int32_t min_level, max_level; }; // r < 0: no search (the point did not project into the frame)
struct SbpBest { // mode 2 only (q_blocks: modes 1 and 2)
    const float* q_angle;     // LastFrame.mvKeysUn[i].angle per query
    const uint8_t* q_blocks;  // "the query's map point has Observations() > 0" (:91-93, :1397-1399); NULL = all
    float factor;             // rotation histogram factor (:1341)
    int check_ori;
    int32_t* match_cur;       // n entries
    int32_t* qbin;            // nq entries of scratch
    // any mode: the reprojection gate of Fuse (ORBmatcher.cc:925-931): skip a candidate when e2 * inv_sigma2[level] > chi2
    double chi2;              // 0 = no gate
    float inv_sigma2[16];
};

__device__ __forceinline__ void sbp_frame(
    const orbfe_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc, int n, int ncap /*pow2 >= n*/, float4 bnd,
    const SbpQuery* __restrict__ queries, const uint8_t* __restrict__ qdesc, int nq, uint8_t* __restrict__ taken, int mode,
    int th_high, float nnratio, uint16_t* __restrict__ row_rank, uint8_t* __restrict__ row_dist, int32_t* __restrict__ row_cnt,
    int row_stride, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist, int32_t* __restrict__ best_level,
    int32_t* __restrict__ second_dist, int32_t* __restrict__ second_level, int32_t* __restrict__ match,
    int32_t* __restrict__ nmatches_out, int32_t* __restrict__ overflow, SbpBest bo)
{
    extern __shared__ __align__(16) unsigned char sbp_smem[];
    __shared__ int s_nin;
    __shared__ int s_hist[30];
    uint32_t* s_sorted = (uint32_t*)sbp_smem;                  // (cell << 16) | index, ascending; ncap entries
    float2* s_xy = (float2*)(s_sorted + ncap);                 // by rank
    uint16_t* s_cell0 = (uint16_t*)(s_xy + ncap);              // first rank of every cell, SBP_CELLS + 1 entries
    uint8_t* s_lvl = (uint8_t*)(s_cell0 + SBP_CELLS + 2);      // by rank
    uint8_t* s_taken = s_lvl + ncap;                           // by rank
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // bounds of the undistorted image (Frame::ComputeImageBounds; 0, 0, cols, rows without distortion) and Frame.cc:112-113
    const float mnMinX = bnd.x, mnMinY = bnd.y;
    const float invW = __fdiv_rn((float)GRID_COLS, bnd.z - bnd.x);
    const float invH = __fdiv_rn((float)GRID_ROWS, bnd.w - bnd.y);

    // ---- A
    if (tid == 0) s_nin = 0;
    if (tid < 30) s_hist[tid] = 0;
    if (mode == 2)
        for (int i = tid; i < n; i += SBP_THREADS) bo.match_cur[i] = -1;
    __syncthreads();
    for (int i = tid; i < n; i += SBP_THREADS) {
        const orbfe_keypoint kp = kps[i];
        const int px = (int)roundf(__fmul_rn(kp.x - mnMinX, invW));
        const int py = (int)roundf(__fmul_rn(kp.y - mnMinY, invH));
        if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) {
            const int k = atomicAdd(&s_nin, 1);
            s_sorted[k] = ((uint32_t)(px * GRID_ROWS + py) << 16) | (uint32_t)i;
        }
    }
    __syncthreads();
    const int nin = s_nin;
    {
        int P = 1;
        while (P < nin) P <<= 1;
        for (int i = nin + tid; i < P; i += SBP_THREADS) s_sorted[i] = 0xffffffffu;
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += SBP_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const uint32_t a = s_sorted[i], b = s_sorted[l];
                    const bool up = ((i & k) == 0);
                    if ((a > b) == up) { s_sorted[i] = b; s_sorted[l] = a; }
                }
                __syncthreads();
            }
    }
    for (int c = tid; c <= SBP_CELLS; c += SBP_THREADS) { // lower bound of (c << 16)
        const uint32_t key = (uint32_t)c << 16;
        int lo = 0, hi = nin;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_sorted[mid] < key) lo = mid + 1; else hi = mid;
        }
        s_cell0[c] = (uint16_t)lo;
    }
    for (int i = tid; i < nin; i += SBP_THREADS) {
        const int idx = s_sorted[i] & 0xffff;
        const orbfe_keypoint kp = kps[idx];
        s_xy[i] = make_float2(kp.x, kp.y);
        s_lvl[i] = (uint8_t)kp.octave;
        s_taken[i] = taken ? taken[idx] : 0;
    }
    __syncthreads();

    // ---- B
    for (int q = wid; q < nq; q += SBP_THREADS / 64) {
        const SbpQuery Q = queries[q];
        const float x = Q.x, y = Q.y, r = Q.r;
        int count = 0;
        const int nMinCellX = max(0, (int)floorf(__fmul_rn(x - mnMinX - r, invW)));
        const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(x - mnMinX + r, invW)));
        const int nMinCellY = max(0, (int)floorf(__fmul_rn(y - mnMinY - r, invH)));
        const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(y - mnMinY + r, invH)));
        if (!(r < 0.0f) && !(nMinCellX >= GRID_COLS || nMaxCellX < 0 || nMinCellY >= GRID_ROWS || nMaxCellY < 0)) {
            const bool check_levels = (Q.min_level > 0) || (Q.max_level >= 0);
            const uint4 a0 = reinterpret_cast<const uint4*>(qdesc)[2 * q];
            const uint4 a1 = reinterpret_cast<const uint4*>(qdesc)[2 * q + 1];
            uint16_t* rr = row_rank + (size_t)q * row_stride;
            uint8_t* rd = row_dist + (size_t)q * row_stride;
            for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
                const int j_begin = s_cell0[ix * GRID_ROWS + nMinCellY], j_end = s_cell0[ix * GRID_ROWS + nMaxCellY + 1];
                for (int j0 = j_begin; j0 < j_end; j0 += 64) {
                    const int j = j0 + lane;
                    bool ok = false;
                    if (j < j_end) {
                        const int lv = s_lvl[j];
                        ok = !check_levels || (lv >= Q.min_level && (Q.max_level < 0 || lv <= Q.max_level));
                        if (ok) {
                            const float2 p = s_xy[j];
                            ok = fabsf(p.x - x) < r && fabsf(p.y - y) < r;
                            if (ok && bo.chi2 > 0.0) {
                                const float ex = x - p.x, ey = y - p.y;
                                const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                                ok = !((double)__fmul_rn(e2, bo.inv_sigma2[min(lv, 15)]) > bo.chi2);
                            }
                        }
                    }
                    const unsigned long long m = __ballot(ok);
                    if (ok) {
                        const int pos = count + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                        if (pos < row_stride) {
                            const int i2 = s_sorted[j] & 0xffff;
                            const uint4 b0 = reinterpret_cast<const uint4*>(desc)[2 * i2];
                            const uint4 b1 = reinterpret_cast<const uint4*>(desc)[2 * i2 + 1];
                            const int d = hamming256(a0, a1, b0, b1);
                            rr[pos] = (uint16_t)j;
                            rd[pos] = (uint8_t)(d > 255 ? 255 : d); // 256 only for exact complements: never accepted
                        }
                    }
                    count += __popcll(m);
                }
            }
        }
        if (lane == 0) {
            if (count > row_stride) atomicMax(overflow, count);
            row_cnt[q] = min(count, row_stride);
        }
    }
    __threadfence_block();
    __syncthreads();

    // ---- C: best / second-best over a row; returns keys (d << 32 | position << 16 | rank), ~0 = none
    auto resolve = [&](int q, unsigned long long& bestk, unsigned long long& secondk) {
        const int e = row_cnt[q];
        const uint16_t* rr = row_rank + (size_t)q * row_stride;
        const uint8_t* rd = row_dist + (size_t)q * row_stride;
        bestk = ~0ull; secondk = ~0ull;
        for (int j0 = 0; j0 < e; j0 += 64) {
            const int j = j0 + lane;
            unsigned long long key = ~0ull;
            if (j < e) {
                const int rk = rr[j];
                if (!s_taken[rk]) key = ((unsigned long long)rd[j] << 32) | ((unsigned long long)j << 16) | (unsigned)rk;
            }
            const unsigned long long m1 = wave_min_u64(key);
            const unsigned long long k2 = wave_min_u64(key == m1 ? ~0ull : key);
            if (m1 < bestk) { secondk = min(bestk, k2); bestk = m1; }
            else secondk = min(secondk, m1);
        }
    };
    auto write_raw = [&](int q, unsigned long long bestk, unsigned long long secondk) {
        if (lane == 0 && best_idx) {
            const bool hb = bestk != ~0ull, hs = secondk != ~0ull;
            best_idx[q] = hb ? (int)(s_sorted[bestk & 0xffff] & 0xffff) : -1;
            best_dist[q] = hb ? (int)(bestk >> 32) : 256;
            best_level[q] = hb ? (int)s_lvl[bestk & 0xffff] : -1;
            second_dist[q] = hs ? (int)(secondk >> 32) : 256;
            second_level[q] = hs ? (int)s_lvl[secondk & 0xffff] : -1;
        }
    };
    if (mode == 0) {
        for (int q = wid; q < nq; q += SBP_THREADS / 64) {
            unsigned long long bk, sk;
            resolve(q, bk, sk);
            write_raw(q, bk, sk);
        }
        return;
    }
    if (wid != 0) return;
    int nmatches = 0;
    if (mode == 2) {
        for (int q = 0; q < nq; q++) {
            unsigned long long bk, sk;
            resolve(q, bk, sk);
            write_raw(q, bk, sk);
            int m = -1, bin = -1;
            if (bk != ~0ull && (int)(bk >> 32) <= th_high) { // :1421
                const int bestRank = (int)(bk & 0xffff);
                m = (int)(s_sorted[bestRank] & 0xffff);
                if (bo.check_ori) {
                    float rot = bo.q_angle[q] - kps[m].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    bin = (int)roundf(__fmul_rn(rot, bo.factor));
                    if (bin == 30) bin = 0;
                    bin = min(max(bin, 0), 29);
                }
                if (lane == 0) {
                    bo.match_cur[m] = q; // CurrentFrame.mvpMapPoints[bestIdx2] = pMP (:1423)
                    if (!bo.q_blocks || bo.q_blocks[q]) { s_taken[bestRank] = 1; if (taken) taken[m] = 1; }
                }
                nmatches++;
            }
            if (lane == 0) { match[q] = m; bo.qbin[q] = bin; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (bo.check_ori) { // :1440-1471: every push_back counts, also for a keypoint that was assigned twice
            __threadfence_block();
            for (int q = lane; q < nq; q += 64)
                if (match[q] >= 0) atomicAdd(&s_hist[bo.qbin[q]], 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < 30; i++) { // ComputeThreeMaxima, ORBmatcher.cc:1605-1646
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
                else if (sz > max3) { max3 = sz; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
            int removed = 0;
            for (int q0 = 0; q0 < nq; q0 += 64) {
                const int q = q0 + lane;
                bool rm = false;
                if (q < nq && match[q] >= 0) {
                    const int bin = bo.qbin[q];
                    rm = bin != ind1 && bin != ind2 && bin != ind3;
                    if (rm) bo.match_cur[match[q]] = -1;
                }
                removed += __popcll(__ballot(rm));
            }
            nmatches -= removed;
        }
        if (lane == 0) *nmatches_out = nmatches;
        return;
    }
    for (int q = 0; q < nq; q++) {
        unsigned long long bk, sk;
        resolve(q, bk, sk);
        write_raw(q, bk, sk);
        int m = -1;
        if (bk != ~0ull) {
            const int bestDist = (int)(bk >> 32), bestRank = (int)(bk & 0xffff);
            // a second-best of 256 (none) has level -1, which never equals the best's level
            const int bestDist2 = sk != ~0ull ? (int)(sk >> 32) : 256;
            const int bestLevel = s_lvl[bestRank], bestLevel2 = sk != ~0ull ? (int)s_lvl[sk & 0xffff] : -1;
            if (bestDist <= th_high && !(bestLevel == bestLevel2 && (float)bestDist > __fmul_rn(nnratio, (float)bestDist2))) {
                m = (int)(s_sorted[bestRank] & 0xffff);
                // the keypoint blocks later queries only if the map point it received is observed (:91-93)
                if (lane == 0 && (!bo.q_blocks || bo.q_blocks[q])) { s_taken[bestRank] = 1; if (taken) taken[m] = 1; }
                nmatches++;
            }
        }
        if (lane == 0) match[q] = m;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) *nmatches_out = nmatches;
}

__global__ __launch_bounds__(SBP_THREADS) void k_search_by_projection(
    const orbfe_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc, int n, int ncap /*pow2 >= n*/, float4 bnd,
    const SbpQuery* __restrict__ queries, const uint8_t* __restrict__ qdesc, int nq, uint8_t* __restrict__ taken, int mode,
    int th_high, float nnratio, uint16_t* __restrict__ row_rank, uint8_t* __restrict__ row_dist, int32_t* __restrict__ row_cnt,
    int row_stride, int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist, int32_t* __restrict__ best_level,
    int32_t* __restrict__ second_dist, int32_t* __restrict__ second_level, int32_t* __restrict__ match,
    int32_t* __restrict__ nmatches_out, int32_t* __restrict__ overflow, SbpBest bo)
{
    sbp_frame(kps, desc, n, ncap, bnd, queries, qdesc, nq, taken, mode, th_high, nnratio, row_rank, row_dist, row_cnt, row_stride, best_idx,
              best_dist, best_level, second_dist, second_level, match, nmatches_out, overflow, bo);
}

// The same search over the frames of a batch resident on the device: workgroup f = frame f of blocks of `capacity` keypoint
// records and `qcapacity` query records (the layouts of orbfe_extract_batch_device).
struct SbpBatch {
    const orbfe_keypoint* kps; const uint8_t* desc; const int32_t* n; int capacity;
    const SbpQuery* queries; const uint8_t* qdesc; const int32_t* nq; int qcapacity;
    uint8_t* taken; const uint8_t* q_observed; const float* q_angle;
    uint16_t* row_rank; uint8_t* row_dist; int32_t* row_cnt; int row_stride;
    int32_t *best_idx, *best_dist, *best_level, *second_dist, *second_level, *match, *nmatches, *overflow;
    int32_t *match_cur, *qbin;
    int qdesc_shared;     // every frame's queries use the same descriptor block (the same map points into several keyframes: Fuse)
    double chi2;          // > 0: reprojection gate of Fuse (SbpBest::chi2)
    float inv_sigma2[16];
};
__global__ __launch_bounds__(SBP_THREADS) void k_search_by_projection_batch(SbpBatch B, int ncap, float4 bnd, int mode, int th_high,
                                                                            float nnratio, float factor, int check_ori)
{
    const size_t f = blockIdx.x, ko = f * B.capacity, qo = f * B.qcapacity;
    const int n = min(B.n[f], B.capacity), nq = min(B.nq[f], B.qcapacity);
    SbpBest bo{};
    bo.q_angle = B.q_angle ? B.q_angle + qo : nullptr;
    bo.q_blocks = B.q_observed ? B.q_observed + qo : nullptr;
    bo.factor = factor; bo.check_ori = check_ori;
    bo.match_cur = B.match_cur ? B.match_cur + ko : nullptr;
    bo.qbin = B.qbin ? B.qbin + qo : nullptr;
    bo.chi2 = B.chi2;
#pragma unroll
    for (int l = 0; l < 16; l++) bo.inv_sigma2[l] = B.inv_sigma2[l];
    sbp_frame(B.kps + ko, B.desc + ko * 32, n, ncap, bnd, B.queries + qo, B.qdesc + (B.qdesc_shared ? 0 : qo * 32), nq, B.taken ? B.taken + ko : nullptr, mode, th_high,
              nnratio, B.row_rank + qo * B.row_stride, B.row_dist + qo * B.row_stride, B.row_cnt + qo, B.row_stride,
              B.best_idx ? B.best_idx + qo : nullptr, B.best_dist ? B.best_dist + qo : nullptr, B.best_level ? B.best_level + qo : nullptr,
              B.second_dist ? B.second_dist + qo : nullptr, B.second_level ? B.second_level + qo : nullptr, B.match + qo, B.nmatches + f,
              B.overflow, bo);
}

// ---------------------------------------------------------------------------------------------------------------------
// Frame glue: cv::undistortPoints with P = K (Frame.cc:357-451).  One point per lane, double arithmetic
// (orbfe_undistort_point in include/orbfe_math.h is the numerics contract); the 12 coefficients come by value.
struct UndistortParams {
    double fx, fy, cx, cy;
    double k[12];
};

// points: n (x, y) float pairs, src -> dst (may alias)
__global__ __launch_bounds__(256) void k_undistort_points(const float2* __restrict__ src, int n, UndistortParams P,
                                                          float2* __restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 p = src[i];
    float2 o;
    orbfe_undistort_point(p.x, p.y, P.fx, P.fy, P.cx, P.cy, P.k, &o.x, &o.y);
    dst[i] = o;
}

// keypoint records of a batch: frame = blockIdx.y, in place on (x, y) (mvKeysUn keeps the other fields, Frame.cc:379-386)
__global__ __launch_bounds__(256) void k_undistort_keypoints(const orbfe_keypoint* __restrict__ kin, const int32_t* __restrict__ d_n,
                                                             int capacity, UndistortParams P, orbfe_keypoint* __restrict__ kout)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(d_n[f], capacity)) return;
    orbfe_keypoint kp = kin[(size_t)f * capacity + i];
    orbfe_undistort_point(kp.x, kp.y, P.fx, P.fy, P.cx, P.cy, P.k, &kp.x, &kp.y);
    kout[(size_t)f * capacity + i] = kp;
}

static int undistort_params(const float* K4, const float* dist, int ndist, UndistortParams& P, const char* who)
{
    if (!K4 || ndist < 0 || ndist > 12 || (ndist && !dist) || !(K4[0] != 0.0f) || !(K4[1] != 0.0f))
        return fail(ORBFE_ERR_INVALID, "%s: invalid camera (K = {fx, fy, cx, cy} with fx, fy != 0; at most 12 coefficients)", who);
    P.fx = K4[0]; P.fy = K4[1]; P.cx = K4[2]; P.cy = K4[3];
    for (int i = 0; i < 12; i++) P.k[i] = i < ndist ? (double)dist[i] : 0.0;
    return ORBFE_OK;
}

// The projection of SearchByProjection(CurrentFrame, LastFrame) (ORBmatcher.cc:1362-1389): one lane per feature of the last
// frame.  x3Dc = Rcw * x3Dw + tcw is OpenCV's 3x3 float product (a row is summed left to right in float) plus the float
// translation; invzc is a double division rounded to float (:1371).  A query that is skipped gets r = -1.
__global__ __launch_bounds__(256) void k_project_last_frame(const float* __restrict__ x3Dw, const uint8_t* __restrict__ valid,
                                                            const orbfe_keypoint* __restrict__ kps_last, int n, const float* __restrict__ Tcw,
                                                            float4 K, float4 bnd, const float* __restrict__ scale, int nlevels, float th,
                                                            SbpQuery* __restrict__ queries, float* __restrict__ q_angle)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    SbpQuery Q{0.0f, 0.0f, -1.0f, 0, 0};
    const orbfe_keypoint kp = kps_last[i];
    q_angle[i] = kp.angle;
    if (!valid || valid[i]) {
        const float X = x3Dw[3 * i], Y = x3Dw[3 * i + 1], Z = x3Dw[3 * i + 2];
        const float xc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Tcw[0], X), __fmul_rn(Tcw[1], Y)), __fmul_rn(Tcw[2], Z)), Tcw[3]);
        const float yc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Tcw[4], X), __fmul_rn(Tcw[5], Y)), __fmul_rn(Tcw[6], Z)), Tcw[7]);
        const float zc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(Tcw[8], X), __fmul_rn(Tcw[9], Y)), __fmul_rn(Tcw[10], Z)), Tcw[11]);
        const float invzc = (float)(1.0 / (double)zc);
        if (!(invzc < 0)) {
            const float u = __fadd_rn(__fmul_rn(__fmul_rn(K.x, xc), invzc), K.z);
            const float v = __fadd_rn(__fmul_rn(__fmul_rn(K.y, yc), invzc), K.w);
            // NaN coordinates (zc == 0 with xc == 0) are dropped here; the reference would index the grid with them
            if (u >= bnd.x && u <= bnd.z && v >= bnd.y && v <= bnd.w) {
                const int oct = min(max(kp.octave, 0), nlevels - 1);
                Q = SbpQuery{u, v, __fmul_rn(th, scale[oct]), kp.octave - 1, kp.octave + 1}; // mono: :1396
            }
        }
    }
    queries[i] = Q;
}

// The projection and visibility gates shared by Fuse (ORBmatcher.cc:848-893 and :1008-1049) and the keyframe variants of
// SearchByProjection (:321-358, :1497-1530): camera coordinates (OpenCV's 3x3 float product + float translation), positive
// depth, inside the image, distance inside the scale-invariance range [0.8f * mfMinDistance, 1.2f * mfMaxDistance]
// (MapPoint::Get{Min,Max}DistanceInvariance, MapPoint.cc:402-412), optional viewing-angle gate (P - Ow) . n >= 0.5 d
// (cv::Mat::dot and cv::norm accumulate in double), MapPoint::PredictScale (MapPoint.cc:414-446: mfMaxDistance / dist, and --
// TemplatedVocabulary.h:36 puts `using namespace std` in front of MapPoint.cc -- std::log(float), float division, std::ceil(float);
// the float log here is the correctly rounded one), search radius th * scale[level].  One lane per map point; a point that is
// not searched gets r = -1.
// frame_variant: the relocalisation search SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) projects with other
// arithmetic (:1503-1512): NO depth gate, invzc = 1.0 / zc evaluated in double, u = (fx * xc) * invzc + cx, Frame bounds (<=).
struct ProjectParams {
    float T[12], Ow[3], K[4], bnd[4], scale[16];
    int nlevels, strict_max, level_below, level_above;
    float log_scale, th;
    // SearchBySim3 (ORBmatcher.cc:1158-1159, :1238-1239): a second transform applied to the camera coordinates of the first, each
    // stage rounded to float as the two cv::Mat expressions are; the distance is then the norm of the final camera coordinates
    int use_T2;
    float T2[12];
    int frame_variant;
};

__global__ __launch_bounds__(256) void k_project_map_points(const float* __restrict__ p3Dw, const uint8_t* __restrict__ valid,
                                                            const float* __restrict__ min_dist, const float* __restrict__ max_dist,
                                                            const float* __restrict__ normal, int n, ProjectParams P,
                                                            SbpQuery* __restrict__ queries)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    SbpQuery Q{0.0f, 0.0f, -1.0f, 0, 0};
    if (!valid || valid[i]) {
        const float X = p3Dw[3 * i], Y = p3Dw[3 * i + 1], Z = p3Dw[3 * i + 2];
        float xc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.T[0], X), __fmul_rn(P.T[1], Y)), __fmul_rn(P.T[2], Z)), P.T[3]);
        float yc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.T[4], X), __fmul_rn(P.T[5], Y)), __fmul_rn(P.T[6], Z)), P.T[7]);
        float zc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.T[8], X), __fmul_rn(P.T[9], Y)), __fmul_rn(P.T[10], Z)), P.T[11]);
        if (P.use_T2) {
            const float a = xc, b = yc, c = zc;
            xc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.T2[0], a), __fmul_rn(P.T2[1], b)), __fmul_rn(P.T2[2], c)), P.T2[3]);
            yc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.T2[4], a), __fmul_rn(P.T2[5], b)), __fmul_rn(P.T2[6], c)), P.T2[7]);
            zc = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(P.T2[8], a), __fmul_rn(P.T2[9], b)), __fmul_rn(P.T2[10], c)), P.T2[11]);
        }
        float u = 0.0f, v = 0.0f;
        bool inside = false;
        if (P.frame_variant) {
            const float invzc = (float)(1.0 / (double)zc);
            u = __fadd_rn(__fmul_rn(__fmul_rn(P.K[0], xc), invzc), P.K[2]);
            v = __fadd_rn(__fmul_rn(__fmul_rn(P.K[1], yc), invzc), P.K[3]);
            // NaN coordinates (zc == 0 with xc == 0) fail every comparison and are dropped; the reference would index the grid with them
            inside = u >= P.bnd[0] && u <= P.bnd[2] && v >= P.bnd[1] && v <= P.bnd[3];
        } else if (!(zc < 0.0f)) {
            const float invz = __fdiv_rn(1.0f, zc);
            u = __fadd_rn(__fmul_rn(P.K[0], __fmul_rn(xc, invz)), P.K[2]);
            v = __fadd_rn(__fmul_rn(P.K[1], __fmul_rn(yc, invz)), P.K[3]);
            inside = P.strict_max ? (u >= P.bnd[0] && u < P.bnd[2] && v >= P.bnd[1] && v < P.bnd[3])  // KeyFrame::IsInImage
                                  : (u >= P.bnd[0] && u <= P.bnd[2] && v >= P.bnd[1] && v <= P.bnd[3]);
        }
        if (inside) {
            const float px = P.use_T2 ? xc : X - P.Ow[0], py = P.use_T2 ? yc : Y - P.Ow[1], pz = P.use_T2 ? zc : Z - P.Ow[2];
            const float dist3D = (float)sqrt((double)px * px + (double)py * py + (double)pz * pz);
            const float mfMax = max_dist[i];
            bool ok = !(dist3D < __fmul_rn(0.8f, min_dist[i]) || dist3D > __fmul_rn(1.2f, mfMax));
            if (ok && normal) {
                const double dot = (double)px * normal[3 * i] + (double)py * normal[3 * i + 1] + (double)pz * normal[3 * i + 2];
                ok = !(dot < 0.5 * (double)dist3D);
            }
            if (ok) {
                const float ratio = __fdiv_rn(mfMax, dist3D);
                int lvl = (int)ceilf(__fdiv_rn((float)log((double)ratio), P.log_scale));
                if (lvl < 0) lvl = 0;
                else if (lvl >= P.nlevels) lvl = P.nlevels - 1;
                Q = SbpQuery{u, v, __fmul_rn(P.th, P.scale[lvl]), lvl - P.level_below, lvl + P.level_above};
            }
        }
    }
    queries[i] = Q;
}

// ---------------------------------------------------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:270-333): among the N descriptors a map point was observed with,
// the one with the least MEDIAN Hamming distance to the others -- median = sorted row[(int)(0.5 * (N - 1))], the first row wins
// ties (`median < BestMedian`).  One wave per map point, N <= 256: lane l keeps descriptors l, l + 64, l + 128, l + 192 in
// registers; row i's descriptor comes by v_readlane from the lane that owns it; the k-th smallest of a row's distances (all in
// 0 .. 256) is found by bisection on the VALUE with one ballot + popcount per chunk and step -- no sort, no LDS.
#define DD_MAXN 256
__global__ __launch_bounds__(256) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ offsets, int npoints,
                                                     int32_t* __restrict__ best_idx, uint8_t* __restrict__ best_desc)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= npoints) return;
    const int o0 = offsets[p], N = min(offsets[p + 1] - o0, DD_MAXN);
    if (N <= 0) {
        if (lane == 0) best_idx[p] = -1;
        return;
    }
    const uint4* D = reinterpret_cast<const uint4*>(desc) + (size_t)o0 * 2;
    uint32_t w[4][8];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int j = lane + 64 * c;
        uint4 a = make_uint4(0, 0, 0, 0), b = a;
        if (j < N) { a = D[2 * j]; b = D[2 * j + 1]; }
        w[c][0] = a.x; w[c][1] = a.y; w[c][2] = a.z; w[c][3] = a.w; w[c][4] = b.x; w[c][5] = b.y; w[c][6] = b.z; w[c][7] = b.w;
    }
    const int k = (int)(0.5 * (double)(N - 1)); // index of the median in the sorted row
    int best_median = 0x7fffffff, best = 0;
    for (int i = 0; i < N; i++) {
        uint32_t r[8];
        const int src_lane = i & 63, src_c = i >> 6;
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t v = src_c == 0 ? w[0][q] : src_c == 1 ? w[1][q] : src_c == 2 ? w[2][q] : w[3][q];
            r[q] = (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
        }
        int d[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int s_ = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) s_ += __popc(w[c][q] ^ r[q]);
            d[c] = s_;
        }
        int lo = 0, hi = 256; // smallest m with #(d <= m) >= k + 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) cnt += __popcll(__ballot(lane + 64 * c < N && d[c] <= mid));
            if (cnt >= k + 1) hi = mid; else lo = mid + 1;
        }
        if (lo < best_median) { best_median = lo; best = i; }
    }
    if (lane == 0) best_idx[p] = best;
    if (best_desc && lane < 8) reinterpret_cast<uint32_t*>(best_desc)[(size_t)p * 8 + lane] = reinterpret_cast<const uint32_t*>(D)[(size_t)best * 8 + lane];
}

struct MatchWorkspace {
    DevBuf pidx, pbest, psecond, csr_cnt, csr_idx, csr_dist, scratch, overflow, prev;
    DevBuf sbp_batch_overflow; // flag of the _batch_device searches: sticky until orbfe_search_by_projection_batch_status reads it
    DevBuf sfi_overflow;  // SearchForInitialization's two flag words: zeroed when allocated and whenever they are read (no memset launch per batch)
    DevBuf q, t, nq, nt, oidx, obest, osecond, kps, desc, nk, m12, nm;
    int csr_per_pair = 0; // candidate pool entries per frame pair of SearchForInitialization (grown on overflow)
    int sbp_stride = 0;   // candidate row stride of k_search_by_projection (grown on overflow)
    // host-pointer SearchForInitialization: a stream of its own (the null stream synchronises with every blocking stream of the
    // process) and page-locked staging, so that a call is a few queued copies and one wait
    hipStream_t host_stream = nullptr;
    PinnedBuf pinned;
    ~MatchWorkspace() { if (host_stream) (void)hipStreamDestroy(host_stream); }
};
// one workspace per (thread, device, stream): see ThreadWorkspaces.  The host-pointer entry points run on the null stream.
static thread_local ThreadWorkspaces<MatchWorkspace> tl_ws;
static MatchWorkspace& ws(hipStream_t s = nullptr) { return tl_ws.get(s); }
// a capacity flag that is zeroed once, when it is allocated, and again only by the status call that reads it
static int ensure_sticky_flag(DevBuf& b)
{
    if (b.p) return ORBFE_OK;
    int rc = b.ensure(16);
    if (rc) return rc;
    ORBFE_HIP(hipMemset(b.p, 0, 16));
    return ORBFE_OK;
}

// debug: 0 = choose by problem size, 1 = VALU tiles (+ split/merge), 2 = matrix cores (orbfe_debug_control "knn2_path")
static int g_knn2_path = 0;
#ifdef ORBFE_ABLATION
int g_orb_skip = 0, g_aruco_skip = 0;
#endif

static int knn2_launch(const uint8_t* d_Q, const int32_t* d_nq, size_t q_stride, int max_nq, const uint8_t* d_T,
                       const int32_t* d_nt, size_t t_stride, int max_nt, int npairs, int init, int32_t* d_best_idx,
                       int32_t* d_best_dist, int32_t* d_second_dist, hipStream_t s)
{
    const bool mfma_ok = max_nt <= 65535 && init > 0;
    const bool use_mfma = mfma_ok && g_knn2_path != 1;   // distances on the matrix cores unless the VALU kernel is forced
    // split the train set when there are too few (query-tile, pair) workgroups to fill 256 CUs
    const int qtiles = (max_nq + 255) / 256;
    int nsplit = 1, chunk;
    const long long wgs = (long long)qtiles * npairs;
    if (use_mfma) {
        // k_knn2_mfma: 8 waves of ~110 registers -> two workgroups per CU, 512 resident; the split aims at ONE round of them (a second,
        // half-empty round costs as much as a full one), in whole 128-descriptor chunks
        const int max_split = (max_nt + KM_CHUNK - 1) / KM_CHUNK;
        if (wgs < 512) nsplit = (int)std::max<long long>(1, std::min<long long>(512 / wgs, max_split));
        chunk = (max_nt + nsplit - 1) / nsplit;
        chunk = std::max(KM_CHUNK, (chunk + KM_CHUNK - 1) / KM_CHUNK * KM_CHUNK);
    } else {
        if (wgs < 1024) {
            nsplit = (int)std::min<long long>((1024 + wgs - 1) / wgs, (max_nt + KNN_TILE - 1) / KNN_TILE);
            if (nsplit < 1) nsplit = 1;
        }
        chunk = (max_nt + nsplit - 1) / nsplit;
        chunk = std::max(KNN_TILE, (chunk + KNN_TILE - 1) / KNN_TILE * KNN_TILE);
    }
    nsplit = std::max(1, (max_nt + chunk - 1) / chunk);
    const dim3 mgrid((max_nq + KM_WAVES * 32 - 1) / (KM_WAVES * 32), npairs, nsplit);
    if (nsplit == 1) {
        if (use_mfma)
            hipLaunchKernelGGL(k_knn2_mfma, mgrid, dim3(KM_WAVES * 64), 0, s, d_Q, d_nq, q_stride, max_nq, d_T, d_nt, t_stride, chunk, init,
                               d_best_idx, d_best_dist, d_second_dist);
        else
            hipLaunchKernelGGL(k_knn2_tiles, dim3(qtiles, npairs, 1), dim3(256), 0, s, d_Q, d_nq, q_stride, max_nq, d_T,
                               d_nt, t_stride, chunk, init, d_best_idx, d_best_dist, d_second_dist);
    } else {
        // too few (query tile, pair) workgroups to fill 256 CUs: the train set is split, partials merged in split order
        MatchWorkspace& w = ws(s);
        const size_t nb = (size_t)npairs * nsplit * max_nq * 4;
        int rc;
        if ((rc = w.pidx.ensure(nb)) || (rc = w.pbest.ensure(nb)) || (rc = w.psecond.ensure(nb))) return rc;
        if (use_mfma)
            hipLaunchKernelGGL(k_knn2_mfma, mgrid, dim3(KM_WAVES * 64), 0, s, d_Q, d_nq, q_stride, max_nq, d_T, d_nt, t_stride, chunk, init,
                               w.pidx.as<int32_t>(), w.pbest.as<int32_t>(), w.psecond.as<int32_t>());
        else
            hipLaunchKernelGGL(k_knn2_tiles, dim3(qtiles, npairs, nsplit), dim3(256), 0, s, d_Q, d_nq, q_stride, max_nq,
                               d_T, d_nt, t_stride, chunk, init, w.pidx.as<int32_t>(), w.pbest.as<int32_t>(),
                               w.psecond.as<int32_t>());
        hipLaunchKernelGGL(k_knn2_merge, dim3(qtiles, npairs), dim3(256), 0, s, w.pidx.as<int32_t>(),
                           w.pbest.as<int32_t>(), w.psecond.as<int32_t>(), nsplit, max_nq, d_nq, d_best_idx,
                           d_best_dist, d_second_dist);
    }
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

static float4 frame_bounds(int cols, int rows, const float* bounds)
{
    return bounds ? make_float4(bounds[0], bounds[1], bounds[2], bounds[3]) : make_float4(0.f, 0.f, (float)cols, (float)rows);
}

static int sfi_launch(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int capacity, int npairs,
                      float4 bnd, int window, float nnratio, int check_ori, const float* d_prev_in,
                      float* d_prev_out, int32_t* d_m12, int32_t* d_nm, hipStream_t s, MatchWorkspace& w)
{
    if (capacity > 65535) return fail(ORBFE_ERR_INVALID, "capacity above 65535 keypoints per frame is unsupported");
    // a pair's candidate rows share one dense pool; a pool that turns out too small is flagged, grown by the status call (or the
    // host wrapper) and the batch repeated.  16 K entries hold a 640 x 480 / 1000-feature pair at window 100 more than twice.
    const int pool_cap = std::max((w.csr_per_pair + 3) / 4 * 4, 16384);
    w.csr_per_pair = pool_cap;
    const int nframes = npairs + 1;
    const size_t F = (size_t)nframes * SFI_MAXL0;
    int rc;
    // grid records: nl0 | nq | cursor (ints), then sorted, xy, ang, desc, query per frame
    if ((rc = w.csr_cnt.ensure((size_t)nframes * (2 + SFI_CURSOR_PAD) * 4 + 64)) || (rc = w.csr_idx.ensure(F * (4 + 8 + 4 + 32 + 8 + 2) + 256)) ||
        (rc = w.csr_dist.ensure((size_t)npairs * pool_cap * 4 + 512 /* k_sfi_accept prefetches a row's next 64 entries unconditionally: up to 63 past the last pool */)) || (rc = w.scratch.ensure((size_t)npairs * SFI_MAXL0 * 4)))
        return rc;
    if (!w.sfi_overflow.p) {
        if ((rc = w.sfi_overflow.ensure(16))) return rc;
        ORBFE_HIP(hipMemset(w.sfi_overflow.p, 0, 16));
    }
    SfiGrid G;
    G.nl0 = w.csr_cnt.as<int32_t>();
    G.nq = G.nl0 + nframes;
    G.cursor = G.nq + nframes;
    uint8_t* b = w.csr_idx.as<uint8_t>();
    G.desc = reinterpret_cast<uint4*>(b); b += F * 32;
    G.xy = reinterpret_cast<float2*>(b); b += F * 8;
    G.sorted = reinterpret_cast<uint32_t*>(b); b += F * 4;
    G.qxy = reinterpret_cast<float2*>(b); b += F * 8;
    G.ang = reinterpret_cast<float*>(b); b += F * 4;
    G.query = reinterpret_cast<uint16_t*>(b);
    int32_t* ovf = w.sfi_overflow.as<int32_t>(); // [0]: level-0 keypoints beyond SFI_MAXL0, [1]: pool entries a pair needed
    hipLaunchKernelGGL(k_sfi_grid, dim3(nframes), dim3(SFI_GRID_THREADS), 0, s, d_kps, d_desc, d_n, capacity, nframes, bnd, G, ovf);
    // a wave per query for frames of up to 256 level-0 keypoints (ORB-SLAM's 1000 features at 640 x 480 have ~217), more per wave above
    hipLaunchKernelGGL(k_sfi_rows, dim3(SFI_ROWS_BX, npairs), dim3(SFI_ROWS_WAVES * 64), 0, s, d_kps, d_desc, capacity, bnd, (float)window, d_prev_in, G,
                       w.csr_dist.as<uint32_t>(), pool_cap, w.scratch.as<uint32_t>(), ovf);
    hipLaunchKernelGGL(k_sfi_accept, dim3(npairs), dim3(64), 0, s, d_kps, d_n, capacity, nnratio, check_ori, d_prev_in, d_prev_out, d_m12,
                       d_nm, G, w.csr_dist.as<uint32_t>(), pool_cap, w.scratch.as<uint32_t>());
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

// reads and clears the two flag words after the stream has drained; *need = 0, the level-0 count that does not fit (an error), or
// the pool size a pair needed (the pool is grown: repeat the call)
static int sfi_read_flags(MatchWorkspace& w, int32_t* need)
{
    int32_t f[2] = {0, 0};
    ORBFE_HIP(hipMemcpy(f, w.sfi_overflow.p, 8, hipMemcpyDeviceToHost));
    if (f[0] || f[1]) ORBFE_HIP(hipMemset(w.sfi_overflow.p, 0, 8)); // sticky until read
    *need = std::max(f[0], f[1]);
    if (f[0] > SFI_MAXL0) return fail(ORBFE_ERR_CAPACITY, "%d level-0 keypoints in a frame exceed the supported %d", f[0], SFI_MAXL0);
    if (f[1] > w.csr_per_pair) w.csr_per_pair = (f[1] + 1023) / 1024 * 1024; // the next batch on this stream has the room
    return ORBFE_OK;
}

} // namespace orbfe

using namespace orbfe;

extern "C" {

int orbfe_debug_control(const char* key, int value)
{
    if (key && !strcmp(key, "knn2_path") && value >= 0 && value <= 2) { g_knn2_path = value; return ORBFE_OK; }
#ifdef ORBFE_ABLATION // diagnosis build only; the shipped library cannot skip work
    if (key && !strcmp(key, "orb_skip")) { orbfe::g_orb_skip = value; return ORBFE_OK; }
    if (key && !strcmp(key, "aruco_skip")) { orbfe::g_aruco_skip = value; return ORBFE_OK; }
#endif
    return fail(ORBFE_ERR_INVALID, "orbfe_debug_control: unknown key or value");
}

int orbfe_hamming(const uint8_t* a, const uint8_t* b)
{
    // host-side scalar helper: the 256-bit Hamming distance of ORBmatcher::DescriptorDistance (ORBmatcher.cc:1651-1667)
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    int dist = 0;
    for (int i = 0; i < 4; i++) dist += __builtin_popcountll(x[i] ^ y[i]);
    return dist;
}

void orbfe_three_maxima(const int32_t* counts, int L, int32_t* ind3)
{
    // host-side scalar helper: ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1605-1646) on the bin populations; the device
    // searches carry their own copy (rot_hist_maxima).  A running top three with strict '>' (the earliest bin wins ties), then the
    // 10 % rule on the runner-ups.
    int top[3] = {0, 0, 0};
    ind3[0] = ind3[1] = ind3[2] = -1;
    for (int i = 0; i < L; i++) {
        const int c = counts[i];
        int k = 0;
        while (k < 3 && c <= top[k]) k++;
        if (k == 3) continue;
        for (int j = 2; j > k; j--) { top[j] = top[j - 1]; ind3[j] = ind3[j - 1]; }
        top[k] = c;
        ind3[k] = i;
    }
    const float floor_ = 0.1f * (float)top[0];
    if ((float)top[1] < floor_) ind3[1] = ind3[2] = -1;
    else if ((float)top[2] < floor_) ind3[2] = -1;
}

int orbfe_epipolar_distance_ok(float x1, float y1, float x2, float y2, const float* F12, float level_sigma2)
{
    // host-side scalar helper: ORBmatcher::CheckDistEpipolarLine (ORBmatcher.cc:139-157); l = x1' F12, row-major 3 x 3
    float l[3];
    for (int j = 0; j < 3; j++) l[j] = x1 * F12[j] + y1 * F12[3 + j] + F12[6 + j];
    const float num = l[0] * x2 + l[1] * y2 + l[2];
    const float den = l[0] * l[0] + l[1] * l[1];
    if (den == 0) return 0;
    return num * num / den < 3.84 * level_sigma2; // the comparison is in double, as in the reference
}

int orbfe_knn2_batch_device(const uint8_t* d_Q, const int32_t* d_nq, size_t q_stride, int max_nq, const uint8_t* d_T,
                            const int32_t* d_nt, size_t t_stride, int max_nt, int npairs, int init,
                            int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist, void* stream)
{
    if (!d_Q || !d_T || !d_nt || !d_best_idx || !d_best_dist || !d_second_dist || max_nq <= 0 || max_nt < 0 ||
        npairs <= 0)
        return fail(ORBFE_ERR_INVALID, "orbfe_knn2_batch_device: invalid argument");
    return knn2_launch(d_Q, d_nq, q_stride, max_nq, d_T, d_nt, t_stride, max_nt, npairs, init, d_best_idx,
                       d_best_dist, d_second_dist, (hipStream_t)stream);
}

int orbfe_knn2(const uint8_t* Q, int nq, const uint8_t* T, int nt, int init, int32_t* best_idx, int32_t* best_dist,
               int32_t* second_dist, int device)
{
    if (nq < 0 || nt < 0 || (nq && (!Q || !best_idx || !best_dist || !second_dist)) || (nt && !T))
        return fail(ORBFE_ERR_INVALID, "orbfe_knn2: invalid argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (nq == 0) return ORBFE_OK;
    MatchWorkspace& w = ws();
    if ((rc = w.q.ensure((size_t)nq * 32)) || (rc = w.t.ensure((size_t)std::max(nt, 1) * 32)) ||
        (rc = w.nt.ensure(16)) || (rc = w.oidx.ensure((size_t)nq * 4)) || (rc = w.obest.ensure((size_t)nq * 4)) ||
        (rc = w.osecond.ensure((size_t)nq * 4)))
        return rc;
    ORBFE_HIP(hipMemcpy(w.q.p, Q, (size_t)nq * 32, hipMemcpyHostToDevice));
    if (nt) ORBFE_HIP(hipMemcpy(w.t.p, T, (size_t)nt * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.nt.p, &nt, 4, hipMemcpyHostToDevice));
    rc = knn2_launch(w.q.as<uint8_t>(), nullptr, 0, nq, w.t.as<uint8_t>(), w.nt.as<int32_t>(), 0, nt, 1, init,
                     w.oidx.as<int32_t>(), w.obest.as<int32_t>(), w.osecond.as<int32_t>(), nullptr);
    if (rc) return rc;
    ORBFE_HIP(hipDeviceSynchronize());
    ORBFE_HIP(hipMemcpy(best_idx, w.oidx.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(best_dist, w.obest.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(second_dist, w.osecond.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_search_for_initialization_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc,
                                                 const int32_t* d_n, int capacity, int npairs, int cols, int rows,
                                                 const float* bounds, int window_size, float nnratio, int check_orientation,
                                                 int32_t* d_matches12, int32_t* d_nmatches, void* stream)
{
    if (!d_kps || !d_desc || !d_n || !d_matches12 || !d_nmatches || capacity <= 0 || npairs <= 0 || cols <= 0 ||
        rows <= 0)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_for_initialization_batch_device: invalid argument");
    return sfi_launch(d_kps, d_desc, d_n, capacity, npairs, frame_bounds(cols, rows, bounds), window_size, nnratio, check_orientation,
                      nullptr, nullptr, d_matches12, d_nmatches, (hipStream_t)stream, ws((hipStream_t)stream));
}

int orbfe_search_for_initialization_batch_status(void* stream, int32_t* overflow)
{
    if (!overflow) return fail(ORBFE_ERR_INVALID, "orbfe_search_for_initialization_batch_status: null argument");
    *overflow = 0;
    hipStream_t s = (hipStream_t)stream;
    MatchWorkspace& w = ws(s);
    if (!w.sfi_overflow.p) return ORBFE_OK; // no batch on this (thread, device, stream) yet
    ORBFE_HIP(hipStreamSynchronize(s));
    return sfi_read_flags(w, overflow);
}

int orbfe_search_for_initialization(const orbfe_keypoint* kps1, const uint8_t* desc1, int n1,
                                    const orbfe_keypoint* kps2, const uint8_t* desc2, int n2, int cols, int rows,
                                    const float* bounds, float* prev_matched, int32_t* matches12, int window_size, float nnratio,
                                    int check_orientation, int32_t* nmatches, int device)
{
    if (n1 < 0 || n2 < 0 || !nmatches || (n1 && (!kps1 || !desc1 || !matches12 || !prev_matched)) ||
        (n2 && (!kps2 || !desc2)) || cols <= 0 || rows <= 0)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_for_initialization: invalid argument");
    int rc = use_device(device);
    if (rc) return rc;
    *nmatches = 0;
    if (n1 == 0) return ORBFE_OK;
    const int cap = std::max(std::max(n1, n2), 1);
    MatchWorkspace& w = ws();
    if ((rc = w.kps.ensure((size_t)2 * cap * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure((size_t)2 * cap * 32)) ||
        (rc = w.nk.ensure(16)) || (rc = w.m12.ensure((size_t)cap * 4)) || (rc = w.nm.ensure(16)) ||
        (rc = w.prev.ensure((size_t)cap * 8)))
        return rc;
    if (!w.host_stream) ORBFE_HIP(hipStreamCreateWithFlags(&w.host_stream, hipStreamNonBlocking));
    hipStream_t s = w.host_stream;
    // page-locked staging: in [kps1 | kps2 | desc1 | desc2 | n1, n2 | prev], out [m12 | prev | nmatches | 2 flag words]
    const size_t kb = (size_t)cap * sizeof(orbfe_keypoint), db = (size_t)cap * 32;
    const size_t i_kps = 0, i_desc = 2 * kb, i_nk = i_desc + 2 * db, i_prev = i_nk + 64, o_m12 = i_prev + (size_t)cap * 8;
    const size_t o_prev = o_m12 + (size_t)cap * 4, o_nm = o_prev + (size_t)cap * 8, o_flags = o_nm + 16, o_end = o_flags + 16;
    if ((rc = w.pinned.ensure(o_end))) return rc;
    uint8_t* hp = w.pinned.as<uint8_t>();
    memcpy(hp + i_kps, kps1, (size_t)n1 * sizeof(orbfe_keypoint));
    memcpy(hp + i_desc, desc1, (size_t)n1 * 32);
    if (n2) {
        memcpy(hp + i_kps + kb, kps2, (size_t)n2 * sizeof(orbfe_keypoint));
        memcpy(hp + i_desc + db, desc2, (size_t)n2 * 32);
    }
    const int32_t nn[2] = {n1, n2};
    memcpy(hp + i_nk, nn, 8);
    memcpy(hp + i_prev, prev_matched, (size_t)n1 * 8);
    // the call's four inputs in one launch that reads the page-locked staging buffer, its four results in another (OutPack,
    // orbfe_common.hpp: every small copy would be a blit kernel of its own)
    {
        OutPack ip;
        ip.add(w.kps.p, hp + i_kps, 2 * kb);
        ip.add(w.desc.p, hp + i_desc, 2 * db);
        ip.add(w.nk.p, hp + i_nk, 8);
        ip.add(w.prev.p, hp + i_prev, (size_t)n1 * 8);   // (with the other inputs: one launch less on the call's chain)
        if ((rc = ip.flush<3>(s))) return rc;
    }
    for (int attempt = 0;; attempt++) {
        // the device copy of prev_matched is only overwritten by a run that did not overflow: a second attempt uploads it again
        if (attempt) { OutPack ip; ip.add(w.prev.p, hp + i_prev, (size_t)n1 * 8); if ((rc = ip.flush<3>(s))) return rc; }
        rc = sfi_launch(w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(), w.nk.as<int32_t>(), cap, 1, frame_bounds(cols, rows, bounds),
                        window_size, nnratio, check_orientation, w.prev.as<float>(), w.prev.as<float>(),
                        w.m12.as<int32_t>(), w.nm.as<int32_t>(), s, w);
        if (rc) return rc;
        // results and flags: copies queued behind the kernels, one wait
        {
            OutPack op;
            op.add(hp + o_m12, w.m12.p, (size_t)n1 * 4);
            op.add(hp + o_prev, w.prev.p, (size_t)n1 * 8);
            op.add(hp + o_nm, w.nm.p, 4);
            op.add(hp + o_flags, w.sfi_overflow.p, 8);
            if ((rc = op.flush<3>(s))) return rc;
        }
        ORBFE_HIP(hipStreamSynchronize(s));
        const int32_t* fl = reinterpret_cast<const int32_t*>(hp + o_flags);
        if (!fl[0] && !fl[1]) break;
        int32_t ovf = 0;
        if ((rc = sfi_read_flags(w, &ovf))) return rc; // clears the flags, grows the pool
        if (attempt) return fail(ORBFE_ERR_CAPACITY, "candidate pool overflow (%d)", ovf);
    }
    memcpy(matches12, hp + o_m12, (size_t)n1 * 4);
    memcpy(prev_matched, hp + o_prev, (size_t)n1 * 8);
    memcpy(nmatches, hp + o_nm, 4);
    return ORBFE_OK;
}

int orbfe_search_by_projection(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                               const orbfe_window_query* queries, const uint8_t* qdesc, int nq, uint8_t* taken, const uint8_t* q_observed,
                               int mode, int th_high, float nnratio, int32_t* best_idx, int32_t* best_dist, int32_t* best_level,
                               int32_t* second_dist, int32_t* second_level, int32_t* match, int32_t* nmatches, int device)
{
    if (n < 0 || nq < 0 || cols <= 0 || rows <= 0 || (mode != 0 && mode != 1) || (n && (!kps || !desc)) ||
        (nq && (!queries || !qdesc)) || (mode == 1 && nq && (!match || !nmatches)) ||
        (best_idx && (!best_dist || !best_level || !second_dist || !second_level)))
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection: invalid argument");
    if (n > 65535) return fail(ORBFE_ERR_INVALID, "more than 65535 keypoints per frame are unsupported");
    int rc = use_device(device);
    if (rc) return rc;
    if (nmatches) *nmatches = 0;
    if (nq == 0) return ORBFE_OK;
    int ncap = 64;
    while (ncap < n) ncap <<= 1;
    const size_t lds = (size_t)ncap * (4 + 8 + 1 + 1) + (SBP_CELLS + 2) * 2 + 64;
    if (lds > 150 * 1024) return fail(ORBFE_ERR_CAPACITY, "%d keypoints do not fit the grid kernel's LDS", n);
    MatchWorkspace& w = ws();
    const size_t qo = (size_t)nq * 4;
    for (int attempt = 0;; attempt++) {
        const int stride = std::max(w.sbp_stride, 128);
        if ((rc = w.kps.ensure((size_t)std::max(n, 1) * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure((size_t)std::max(n, 1) * 32)) ||
            (rc = w.q.ensure((size_t)nq * sizeof(orbfe_window_query))) || (rc = w.t.ensure((size_t)nq * 32)) ||
            (rc = w.prev.ensure((size_t)std::max(n, 1))) || (rc = w.csr_idx.ensure((size_t)nq * stride * 2)) ||
            (rc = w.csr_dist.ensure((size_t)nq * stride)) || (rc = w.csr_cnt.ensure(qo)) || (rc = w.obest.ensure(qo * 6)) ||
            (rc = w.nm.ensure(16)) || (rc = w.overflow.ensure(16)) || (rc = w.pidx.ensure((size_t)nq + 16)))
            return rc;
        if (q_observed) ORBFE_HIP(hipMemcpy(w.pidx.p, q_observed, (size_t)nq, hipMemcpyHostToDevice));
        if (n) {
            ORBFE_HIP(hipMemcpy(w.kps.p, kps, (size_t)n * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
            ORBFE_HIP(hipMemcpy(w.desc.p, desc, (size_t)n * 32, hipMemcpyHostToDevice));
            if (taken) ORBFE_HIP(hipMemcpy(w.prev.p, taken, (size_t)n, hipMemcpyHostToDevice));
        }
        ORBFE_HIP(hipMemcpy(w.q.p, queries, (size_t)nq * sizeof(orbfe_window_query), hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(w.t.p, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemset(w.overflow.p, 0, 4));
        ORBFE_HIP(hipMemset(w.nm.p, 0, 4));
        int32_t* o = w.obest.as<int32_t>();
        SbpBest sb{};
        sb.q_blocks = q_observed ? w.pidx.as<uint8_t>() : nullptr;
        { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(&k_search_by_projection), (size_t)(lds)); if (rc_lds_) return rc_lds_; }
        hipLaunchKernelGGL(k_search_by_projection, dim3(1), dim3(SBP_THREADS), lds, 0, w.kps.as<orbfe_keypoint>(),
                           w.desc.as<uint8_t>(), n, ncap, frame_bounds(cols, rows, bounds), w.q.as<SbpQuery>(), w.t.as<uint8_t>(), nq,
                           taken ? w.prev.as<uint8_t>() : nullptr, mode, th_high, nnratio, w.csr_idx.as<uint16_t>(),
                           w.csr_dist.as<uint8_t>(), w.csr_cnt.as<int32_t>(), stride, o, o + nq, o + 2 * nq, o + 3 * nq,
                           o + 4 * nq, o + 5 * nq, w.nm.as<int32_t>(), w.overflow.as<int32_t>(), sb);
        ORBFE_HIP(hipGetLastError());
        ORBFE_HIP(hipDeviceSynchronize());
        int32_t ovf = 0;
        ORBFE_HIP(hipMemcpy(&ovf, w.overflow.p, 4, hipMemcpyDeviceToHost));
        if (!ovf) break;
        if (attempt) return fail(ORBFE_ERR_CAPACITY, "candidate row overflow (%d)", ovf);
        w.sbp_stride = (ovf + 63) / 64 * 64; // the longest candidate list decides the row stride; run again
    }
    const int32_t* o = w.obest.as<int32_t>();
    if (best_idx) {
        ORBFE_HIP(hipMemcpy(best_idx, o, qo, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(best_dist, o + nq, qo, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(best_level, o + 2 * nq, qo, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(second_dist, o + 3 * nq, qo, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(second_level, o + 4 * nq, qo, hipMemcpyDeviceToHost));
    }
    if (mode == 1) {
        ORBFE_HIP(hipMemcpy(match, o + 5 * nq, qo, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(nmatches, w.nm.p, 4, hipMemcpyDeviceToHost));
        if (taken && n) ORBFE_HIP(hipMemcpy(taken, w.prev.p, (size_t)n, hipMemcpyDeviceToHost));
    }
    return ORBFE_OK;
}

static int project_run(MatchWorkspace& w, const float* p3Dw, const uint8_t* valid, const float* min_dist, const float* max_dist,
                       const float* normal, int n, const ProjectParams& P);

// shared plumbing of the "best only" entry points: queries either come from the host or are projected on the device
static int sbp_best_run(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                        const orbfe_window_query* queries, const float* q_angle, const uint8_t* qdesc, const uint8_t* q_blocks, int nq,
                        const uint8_t* taken, int th_high, int check_ori, float factor, int32_t* match_cur, int32_t* nmatches,
                        // projection request (x3Dw != NULL): the queries are built on the device
                        const float* x3Dw, const uint8_t* valid, const orbfe_keypoint* kps_last, const float* Tcw, const float* K4,
                        const float* scale, int nlevels, float th,
                        // map-point projection request (PP != NULL): k_project_map_points builds the queries from x3Dw / valid / the distances
                        const ProjectParams* PP = nullptr, const float* min_dist = nullptr, const float* max_dist = nullptr)
{
    if (n > 65535) return fail(ORBFE_ERR_INVALID, "more than 65535 keypoints per frame are unsupported");
    *nmatches = 0;
    for (int i = 0; i < n; i++) match_cur[i] = -1;
    if (nq == 0 || n == 0) return ORBFE_OK;
    int ncap = 64;
    while (ncap < n) ncap <<= 1;
    const size_t lds = (size_t)ncap * (4 + 8 + 1 + 1) + (SBP_CELLS + 2) * 2 + 64;
    if (lds > 150 * 1024) return fail(ORBFE_ERR_CAPACITY, "%d keypoints do not fit the grid kernel's LDS", n);
    MatchWorkspace& w = ws();
    const size_t qo = (size_t)nq * 4;
    int rc;
    for (int attempt = 0;; attempt++) {
        const int stride = std::max(w.sbp_stride, 128);
        if ((rc = w.kps.ensure((size_t)n * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure((size_t)n * 32)) ||
            (rc = w.q.ensure((size_t)nq * sizeof(orbfe_window_query))) || (rc = w.t.ensure((size_t)nq * 32)) ||
            (rc = w.prev.ensure((size_t)n)) || (rc = w.csr_idx.ensure((size_t)nq * stride * 2)) ||
            (rc = w.csr_dist.ensure((size_t)nq * stride)) || (rc = w.csr_cnt.ensure(qo)) || (rc = w.obest.ensure(qo * 3)) ||
            (rc = w.nm.ensure(16)) || (rc = w.overflow.ensure(16)) || (rc = w.m12.ensure((size_t)n * 4)) ||
            (rc = w.scratch.ensure((size_t)nq * 32 + 256)) || (rc = w.pidx.ensure((size_t)nq * sizeof(orbfe_keypoint) + nq + 256)))
            return rc;
        ORBFE_HIP(hipMemcpy(w.kps.p, kps, (size_t)n * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(w.desc.p, desc, (size_t)n * 32, hipMemcpyHostToDevice));
        if (taken) ORBFE_HIP(hipMemcpy(w.prev.p, taken, (size_t)n, hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(w.t.p, qdesc, (size_t)nq * 32, hipMemcpyHostToDevice));
        // scratch: [q_angle f32 x nq | x3Dw f32 x 3nq]; pidx: [kps_last | valid / blocks bytes]
        float* d_angle = w.scratch.as<float>();
        uint8_t* d_flags = w.pidx.as<uint8_t>() + (size_t)nq * sizeof(orbfe_keypoint);
        if (PP) {
            // the staging of project_run shares w.scratch / w.pidx (already large enough: no reallocation) with the arrays below
            if ((rc = project_run(w, x3Dw, valid, min_dist, max_dist, nullptr, nq, *PP))) return rc;
            ORBFE_HIP(hipDeviceSynchronize());
            ORBFE_HIP(hipMemcpy(d_angle, q_angle, (size_t)nq * 4, hipMemcpyHostToDevice));
        } else if (x3Dw) {
            float* d_x = d_angle + nq;
            ORBFE_HIP(hipMemcpy(d_x, x3Dw, (size_t)nq * 12, hipMemcpyHostToDevice));
            ORBFE_HIP(hipMemcpy(w.pidx.p, kps_last, (size_t)nq * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
            if (valid) ORBFE_HIP(hipMemcpy(d_flags, valid, (size_t)nq, hipMemcpyHostToDevice));
            float h[12 + 16];
            memcpy(h, Tcw, 48);
            for (int l = 0; l < 16; l++) h[12 + l] = l < nlevels ? scale[l] : 0.0f;
            if ((rc = w.pbest.ensure(sizeof h))) return rc;
            ORBFE_HIP(hipMemcpy(w.pbest.p, h, sizeof h, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_project_last_frame, dim3((nq + 255) / 256), dim3(256), 0, 0, d_x, valid ? d_flags : nullptr,
                               w.pidx.as<orbfe_keypoint>(), nq, w.pbest.as<float>(), make_float4(K4[0], K4[1], K4[2], K4[3]),
                               frame_bounds(cols, rows, bounds), w.pbest.as<float>() + 12, nlevels, th, w.q.as<SbpQuery>(), d_angle);
            ORBFE_HIP(hipGetLastError());
            ORBFE_HIP(hipDeviceSynchronize()); // d_flags is reused for the blocks flags below
        } else {
            ORBFE_HIP(hipMemcpy(w.q.p, queries, (size_t)nq * sizeof(orbfe_window_query), hipMemcpyHostToDevice));
            ORBFE_HIP(hipMemcpy(d_angle, q_angle, (size_t)nq * 4, hipMemcpyHostToDevice));
        }
        if (q_blocks) ORBFE_HIP(hipMemcpy(d_flags, q_blocks, (size_t)nq, hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemset(w.overflow.p, 0, 4));
        ORBFE_HIP(hipMemset(w.nm.p, 0, 4));
        int32_t* o = w.obest.as<int32_t>();
        SbpBest bo{d_angle, q_blocks ? d_flags : nullptr, factor, check_ori, w.m12.as<int32_t>(), o + nq};
        { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(&k_search_by_projection), (size_t)(lds)); if (rc_lds_) return rc_lds_; }
        hipLaunchKernelGGL(k_search_by_projection, dim3(1), dim3(SBP_THREADS), lds, 0, w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(), n,
                           ncap, frame_bounds(cols, rows, bounds), w.q.as<SbpQuery>(), w.t.as<uint8_t>(), nq,
                           taken ? w.prev.as<uint8_t>() : nullptr, 2, th_high, 0.0f, w.csr_idx.as<uint16_t>(), w.csr_dist.as<uint8_t>(),
                           w.csr_cnt.as<int32_t>(), stride, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                           (int32_t*)nullptr, o, w.nm.as<int32_t>(), w.overflow.as<int32_t>(), bo);
        ORBFE_HIP(hipGetLastError());
        ORBFE_HIP(hipDeviceSynchronize());
        int32_t ovf = 0;
        ORBFE_HIP(hipMemcpy(&ovf, w.overflow.p, 4, hipMemcpyDeviceToHost));
        if (!ovf) break;
        if (attempt) return fail(ORBFE_ERR_CAPACITY, "candidate row overflow (%d)", ovf);
        w.sbp_stride = (ovf + 63) / 64 * 64;
    }
    ORBFE_HIP(hipMemcpy(match_cur, w.m12.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(nmatches, w.nm.p, 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_search_by_projection_best(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                                    const orbfe_window_query* queries, const float* q_angle, const uint8_t* qdesc, const uint8_t* q_blocks,
                                    int nq, const uint8_t* taken, int th_high, int check_orientation, float factor, int32_t* match_cur,
                                    int32_t* nmatches, int device)
{
    if (n < 0 || nq < 0 || cols <= 0 || rows <= 0 || !nmatches || (n && (!kps || !desc || !match_cur)) ||
        (nq && (!queries || !qdesc || (check_orientation && !q_angle))))
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection_best: invalid argument");
    int rc = use_device(device);
    if (rc) return rc;
    std::vector<float> zero;
    if (!q_angle) { zero.assign(std::max(nq, 1), 0.0f); q_angle = zero.data(); }
    return sbp_best_run(kps, desc, n, cols, rows, bounds, queries, q_angle, qdesc, q_blocks, nq, taken, th_high, check_orientation, factor,
                        match_cur, nmatches, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.0f);
}

int orbfe_search_by_projection_last_frame(const orbfe_keypoint* kps_cur, const uint8_t* desc_cur, int n_cur, const uint8_t* taken_cur,
                                          int cols, int rows, const float* bounds, const orbfe_keypoint* kps_last, int n_last,
                                          const uint8_t* valid_last, const float* x3Dw, const uint8_t* mp_desc, const uint8_t* mp_observed,
                                          const float* Tcw, const float* K4, const float* scale_factors, int nlevels, float th, int th_high,
                                          int check_orientation, int32_t* match_cur, int32_t* nmatches, int device)
{
    if (n_cur < 0 || n_last < 0 || cols <= 0 || rows <= 0 || !nmatches || (n_cur && (!kps_cur || !desc_cur || !match_cur)) ||
        (n_last && (!kps_last || !x3Dw || !mp_desc)) || !Tcw || !K4 || !scale_factors || nlevels < 1 || nlevels > 16)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection_last_frame: invalid argument");
    int rc = use_device(device);
    if (rc) return rc;
    return sbp_best_run(kps_cur, desc_cur, n_cur, cols, rows, bounds, nullptr, nullptr, mp_desc, mp_observed, n_last, taken_cur, th_high,
                        check_orientation, 1.0f / 30 /* :1341 */, match_cur, nmatches, x3Dw, valid_last, kps_last, Tcw, K4, scale_factors,
                        nlevels, th);
}

static int project_params(ProjectParams& P, const float* Tcw, const float* Ow, const float* K4, int cols, int rows, const float* bounds,
                          int strict_max, const float* scale, int nlevels, float log_scale, float th, int below, int above, const char* who)
{
    if (!Tcw || !Ow || !K4 || !scale || nlevels < 1 || nlevels > 16 || cols <= 0 || rows <= 0 || !(log_scale > 0.0f))
        return fail(ORBFE_ERR_INVALID, "%s: invalid camera / pyramid argument", who);
    memcpy(P.T, Tcw, 48); memcpy(P.Ow, Ow, 12); memcpy(P.K, K4, 16);
    const float4 b = frame_bounds(cols, rows, bounds);
    P.bnd[0] = b.x; P.bnd[1] = b.y; P.bnd[2] = b.z; P.bnd[3] = b.w;
    for (int l = 0; l < 16; l++) P.scale[l] = scale[std::min(l, nlevels - 1)];
    P.nlevels = nlevels; P.strict_max = strict_max; P.level_below = below; P.level_above = above; P.log_scale = log_scale; P.th = th;
    P.use_T2 = 0;
    for (int i = 0; i < 12; i++) P.T2[i] = 0.0f;
    P.frame_variant = 0;
    return ORBFE_OK;
}

// uploads the map points and leaves their queries in w.q
static int project_run(MatchWorkspace& w, const float* p3Dw, const uint8_t* valid, const float* min_dist, const float* max_dist,
                       const float* normal, int n, const ProjectParams& P)
{
    int rc;
    const size_t N = (size_t)n;
    if ((rc = w.q.ensure(N * sizeof(orbfe_window_query))) || (rc = w.scratch.ensure(N * (12 + 4 + 4 + 12) + 256)) || (rc = w.pidx.ensure(N + 256)))
        return rc;
    float* d_p = w.scratch.as<float>();
    float *d_min = d_p + 3 * N, *d_max = d_min + N, *d_nrm = d_max + N;
    ORBFE_HIP(hipMemcpy(d_p, p3Dw, N * 12, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(d_min, min_dist, N * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(d_max, max_dist, N * 4, hipMemcpyHostToDevice));
    if (normal) ORBFE_HIP(hipMemcpy(d_nrm, normal, N * 12, hipMemcpyHostToDevice));
    if (valid) ORBFE_HIP(hipMemcpy(w.pidx.p, valid, N, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_project_map_points, dim3((n + 255) / 256), dim3(256), 0, 0, d_p, valid ? w.pidx.as<uint8_t>() : nullptr, d_min, d_max,
                       normal ? d_nrm : nullptr, n, P, w.q.as<SbpQuery>());
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_project_map_points(const float* p3Dw, const uint8_t* valid, const float* min_dist, const float* max_dist, const float* normal, int n,
                             const float* Tcw, const float* Ow, const float* K4, int cols, int rows, const float* bounds, int keyframe_variant,
                             const float* scale_factors, int nlevels, float log_scale_factor, float th, int level_below, int level_above,
                             orbfe_window_query* queries, int device)
{
    if (n < 0 || (n && (!p3Dw || !min_dist || !max_dist || !queries)))
        return fail(ORBFE_ERR_INVALID, "orbfe_project_map_points: invalid argument");
    ProjectParams P;
    int rc = project_params(P, Tcw, Ow, K4, cols, rows, bounds, keyframe_variant, scale_factors, nlevels, log_scale_factor, th, level_below,
                            level_above, "orbfe_project_map_points");
    P.frame_variant = !keyframe_variant;
    if (rc || (rc = use_device(device)) || n == 0) return rc;
    MatchWorkspace& w = ws();
    if ((rc = project_run(w, p3Dw, valid, min_dist, max_dist, normal, n, P))) return rc;
    ORBFE_HIP(hipMemcpy(queries, w.q.p, (size_t)n * sizeof(orbfe_window_query), hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

// projection + grid search + best distance for nmp map points against one keyframe (Fuse, SearchBySim3)
static int guided_best(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds, const float* p3Dw,
                       const uint8_t* valid, const float* min_dist, const float* max_dist, const float* normal, const uint8_t* mp_desc, int nmp,
                       const ProjectParams& P, double chi2, const float* inv_level_sigma2, int nlevels, int32_t* best_idx, int32_t* best_dist)
{
    if (n > 65535) return fail(ORBFE_ERR_INVALID, "more than 65535 keypoints per frame are unsupported");
    for (int i = 0; i < nmp; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    if (nmp == 0 || n == 0) return ORBFE_OK;
    int ncap = 64;
    while (ncap < n) ncap <<= 1;
    const size_t lds = (size_t)ncap * (4 + 8 + 1 + 1) + (SBP_CELLS + 2) * 2 + 64;
    if (lds > 150 * 1024) return fail(ORBFE_ERR_CAPACITY, "%d keypoints do not fit the grid kernel's LDS", n);
    MatchWorkspace& w = ws();
    const size_t qo = (size_t)nmp * 4;
    SbpBest bo{};
    bo.chi2 = chi2 > 0.0 ? chi2 : 0.0;
    if (chi2 > 0.0)
        for (int l = 0; l < 16; l++) bo.inv_sigma2[l] = inv_level_sigma2[std::min(l, nlevels - 1)];
    int rc;
    for (int attempt = 0;; attempt++) {
        const int stride = std::max(w.sbp_stride, 128);
        if ((rc = w.kps.ensure((size_t)n * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure((size_t)n * 32)) || (rc = w.t.ensure((size_t)nmp * 32)) ||
            (rc = w.csr_idx.ensure((size_t)nmp * stride * 2)) || (rc = w.csr_dist.ensure((size_t)nmp * stride)) || (rc = w.csr_cnt.ensure(qo)) ||
            (rc = w.obest.ensure(qo * 6)) || (rc = w.nm.ensure(16)) || (rc = w.overflow.ensure(16)))
            return rc;
        ORBFE_HIP(hipMemcpy(w.kps.p, kps, (size_t)n * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(w.desc.p, desc, (size_t)n * 32, hipMemcpyHostToDevice));
        ORBFE_HIP(hipMemcpy(w.t.p, mp_desc, (size_t)nmp * 32, hipMemcpyHostToDevice));
        if ((rc = project_run(w, p3Dw, valid, min_dist, max_dist, normal, nmp, P))) return rc;
        ORBFE_HIP(hipMemset(w.overflow.p, 0, 4));
        int32_t* o = w.obest.as<int32_t>();
        { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(&k_search_by_projection), (size_t)(lds)); if (rc_lds_) return rc_lds_; }
        hipLaunchKernelGGL(k_search_by_projection, dim3(1), dim3(SBP_THREADS), lds, 0, w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(), n,
                           ncap, frame_bounds(cols, rows, bounds), w.q.as<SbpQuery>(), w.t.as<uint8_t>(), nmp, (uint8_t*)nullptr, 0, 256, 0.0f,
                           w.csr_idx.as<uint16_t>(), w.csr_dist.as<uint8_t>(), w.csr_cnt.as<int32_t>(), stride, o, o + nmp, o + 2 * nmp,
                           o + 3 * nmp, o + 4 * nmp, o + 5 * nmp, w.nm.as<int32_t>(), w.overflow.as<int32_t>(), bo);
        ORBFE_HIP(hipGetLastError());
        ORBFE_HIP(hipDeviceSynchronize());
        int32_t ovf = 0;
        ORBFE_HIP(hipMemcpy(&ovf, w.overflow.p, 4, hipMemcpyDeviceToHost));
        if (!ovf) break;
        if (attempt) return fail(ORBFE_ERR_CAPACITY, "candidate row overflow (%d)", ovf);
        w.sbp_stride = (ovf + 63) / 64 * 64;
    }
    ORBFE_HIP(hipMemcpy(best_idx, w.obest.p, qo, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(best_dist, w.obest.as<int32_t>() + nmp, qo, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_fuse_search(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds, const float* p3Dw,
                      const uint8_t* valid, const float* min_dist, const float* max_dist, const float* normal, const uint8_t* mp_desc, int nmp,
                      const float* Tcw, const float* Ow, const float* K4, const float* scale_factors, const float* inv_level_sigma2, int nlevels,
                      float log_scale_factor, float th, double chi2, int32_t* best_idx, int32_t* best_dist, int device)
{
    if (n < 0 || nmp < 0 || (n && (!kps || !desc)) || (nmp && (!p3Dw || !min_dist || !max_dist || !normal || !mp_desc || !best_idx || !best_dist)) ||
        (chi2 > 0.0 && !inv_level_sigma2))
        return fail(ORBFE_ERR_INVALID, "orbfe_fuse_search: invalid argument");
    ProjectParams P;
    int rc = project_params(P, Tcw, Ow, K4, cols, rows, bounds, 1, scale_factors, nlevels, log_scale_factor, th, 1, 0, "orbfe_fuse_search");
    if (rc || (rc = use_device(device))) return rc;
    return guided_best(kps, desc, n, cols, rows, bounds, p3Dw, valid, min_dist, max_dist, normal, mp_desc, nmp, P, chi2, inv_level_sigma2, nlevels,
                       best_idx, best_dist);
}

int orbfe_search_by_sim3(const orbfe_keypoint* kps1, const uint8_t* desc1, int n1, const orbfe_keypoint* kps2, const uint8_t* desc2, int n2,
                         int cols, int rows, const float* bounds, const float* p3Dw1, const uint8_t* valid1, const float* min_dist1,
                         const float* max_dist1, const uint8_t* mp_desc1, const float* p3Dw2, const uint8_t* valid2, const float* min_dist2,
                         const float* max_dist2, const uint8_t* mp_desc2, const float* T1w, const float* T2w, const float* sT12,
                         const float* sT21, const float* K4, const float* scale_factors, int nlevels, float log_scale_factor, float th,
                         int th_high, int32_t* match12, int32_t* nfound, int device)
{
    if (n1 < 0 || n2 < 0 || !nfound || (n1 && (!kps1 || !desc1 || !p3Dw1 || !min_dist1 || !max_dist1 || !mp_desc1 || !match12)) ||
        (n2 && (!kps2 || !desc2 || !p3Dw2 || !min_dist2 || !max_dist2 || !mp_desc2)) || !T1w || !T2w || !sT12 || !sT21)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_sim3: invalid argument");
    const float zero3[3] = {0.0f, 0.0f, 0.0f};
    ProjectParams P12, P21;
    // KF1's points: camera 1, then sR21 | t21 into camera 2, searched in KF2 (:1148-1219); KF2's points the other way (:1228-1299)
    int rc = project_params(P12, T1w, zero3, K4, cols, rows, bounds, 1, scale_factors, nlevels, log_scale_factor, th, 1, 0, "orbfe_search_by_sim3");
    if (rc || (rc = project_params(P21, T2w, zero3, K4, cols, rows, bounds, 1, scale_factors, nlevels, log_scale_factor, th, 1, 0,
                                   "orbfe_search_by_sim3")) || (rc = use_device(device)))
        return rc;
    P12.use_T2 = 1; memcpy(P12.T2, sT21, 48);
    P21.use_T2 = 1; memcpy(P21.T2, sT12, 48);
    *nfound = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0) return ORBFE_OK;
    std::vector<int32_t> m1(n1), d1(n1), m2(n2), d2(n2);
    if ((rc = guided_best(kps2, desc2, n2, cols, rows, bounds, p3Dw1, valid1, min_dist1, max_dist1, nullptr, mp_desc1, n1, P12, 0.0, nullptr,
                          nlevels, m1.data(), d1.data())) ||
        (rc = guided_best(kps1, desc1, n1, cols, rows, bounds, p3Dw2, valid2, min_dist2, max_dist2, nullptr, mp_desc2, n2, P21, 0.0, nullptr,
                          nlevels, m2.data(), d2.data())))
        return rc;
    int found = 0;
    for (int i1 = 0; i1 < n1; i1++) { // check agreement (:1302-1318)
        const int idx2 = d1[i1] <= th_high ? m1[i1] : -1;
        if (idx2 >= 0) {
            const int idx1 = d2[idx2] <= th_high ? m2[idx2] : -1;
            if (idx1 == i1) { match12[i1] = idx2; found++; }
        }
    }
    *nfound = found;
    return ORBFE_OK;
}

int orbfe_search_by_projection_sim3(const orbfe_keypoint* kps, const uint8_t* desc, int n, int cols, int rows, const float* bounds,
                                    const uint8_t* matched, const float* p3Dw, const uint8_t* valid, const float* min_dist,
                                    const float* max_dist, const float* normal, const uint8_t* mp_desc, int nmp, const float* Tcw,
                                    const float* Ow, const float* K4, const float* scale_factors, int nlevels, float log_scale_factor, int th,
                                    int32_t* match_kf, int32_t* nmatches, int device)
{
    if (nmp < 0 || (nmp && (!p3Dw || !min_dist || !max_dist || !normal || !mp_desc)) || !nmatches)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection_sim3: invalid argument");
    std::vector<orbfe_window_query> q(std::max(nmp, 1));
    // :321-358 projection and gates, then the loop of :360-401: best only, <= TH_LOW, a matched keypoint is taken, no rotation check
    int rc = orbfe_project_map_points(p3Dw, valid, min_dist, max_dist, normal, nmp, Tcw, Ow, K4, cols, rows, bounds, 1, scale_factors, nlevels,
                                      log_scale_factor, (float)th, 1, 0, q.data(), device);
    if (rc) return rc;
    return orbfe_search_by_projection_best(kps, desc, n, cols, rows, bounds, q.data(), nullptr, mp_desc, nullptr, nmp, matched, 50, 0, 0.0f,
                                           match_kf, nmatches, device);
}

int orbfe_search_by_projection_keyframe(const orbfe_keypoint* kps_cur, const uint8_t* desc_cur, int n_cur, const uint8_t* taken_cur, int cols,
                                        int rows, const float* bounds, int n_kf, const float* kf_angle, const uint8_t* valid, const float* p3Dw,
                                        const float* min_dist, const float* max_dist, const uint8_t* mp_desc, const float* Tcw, const float* Ow,
                                        const float* K4, const float* scale_factors, int nlevels, float log_scale_factor, float th, int orb_dist,
                                        int check_orientation, int32_t* match_cur, int32_t* nmatches, int device)
{
    if (n_cur < 0 || n_kf < 0 || !nmatches || (n_cur && (!kps_cur || !desc_cur || !match_cur)) ||
        (n_kf && (!p3Dw || !min_dist || !max_dist || !mp_desc || (check_orientation && !kf_angle))))
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection_keyframe: invalid argument");
    ProjectParams P;
    // :1497-1530: Frame arithmetic and bounds, no viewing-angle gate, levels predicted - 1 .. predicted + 1 (:1532)
    int rc = project_params(P, Tcw, Ow, K4, cols, rows, bounds, 0, scale_factors, nlevels, log_scale_factor, th, 1, 1,
                            "orbfe_search_by_projection_keyframe");
    P.frame_variant = 1;
    if (rc || (rc = use_device(device))) return rc;
    std::vector<float> zero;
    if (!kf_angle) { zero.assign(std::max(n_kf, 1), 0.0f); kf_angle = zero.data(); }
    return sbp_best_run(kps_cur, desc_cur, n_cur, cols, rows, bounds, nullptr, kf_angle, mp_desc, nullptr, n_kf, taken_cur, orb_dist,
                        check_orientation, 1.0f / 30 /* :1488 */, match_cur, nmatches, p3Dw, valid, nullptr, nullptr, nullptr, nullptr, 0, 0.0f,
                        &P, min_dist, max_dist);
}

int orbfe_search_by_projection_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int capacity, int nframes,
                                            int cols, int rows, const float* bounds, const orbfe_window_query* d_queries,
                                            const uint8_t* d_qdesc, const int32_t* d_nq, int qcapacity, uint8_t* d_taken,
                                            const uint8_t* d_q_observed, const float* d_q_angle, int mode, int th_high, float nnratio,
                                            float factor, int check_orientation, int32_t* d_best_idx, int32_t* d_best_dist,
                                            int32_t* d_best_level, int32_t* d_second_dist, int32_t* d_second_level, int32_t* d_match,
                                            int32_t* d_match_cur, int32_t* d_nmatches, void* stream)
{
    if (!d_kps || !d_desc || !d_n || !d_queries || !d_qdesc || !d_nq || !d_match || !d_nmatches || capacity <= 0 || qcapacity <= 0 ||
        nframes <= 0 || cols <= 0 || rows <= 0 || mode < 0 || mode > 2 || (mode == 2 && (!d_match_cur || (check_orientation && !d_q_angle))) ||
        (d_best_idx && (!d_best_dist || !d_best_level || !d_second_dist || !d_second_level)))
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection_batch_device: invalid argument");
    if (capacity > 65535) return fail(ORBFE_ERR_INVALID, "more than 65535 keypoints per frame are unsupported");
    int ncap = 64;
    while (ncap < capacity) ncap <<= 1;
    const size_t lds = (size_t)ncap * (4 + 8 + 1 + 1) + (SBP_CELLS + 2) * 2 + 64;
    if (lds > 150 * 1024) return fail(ORBFE_ERR_CAPACITY, "%d keypoints do not fit the grid kernel's LDS", capacity);
    hipStream_t s = (hipStream_t)stream;
    MatchWorkspace& w = ws(s);
    const int stride = std::max(w.sbp_stride, 128);
    w.sbp_stride = stride;
    const size_t NQ = (size_t)nframes * qcapacity;
    int rc;
    if ((rc = w.csr_idx.ensure(NQ * stride * 2)) || (rc = w.csr_dist.ensure(NQ * stride)) || (rc = w.csr_cnt.ensure(NQ * 4)) ||
        (rc = w.scratch.ensure(NQ * 4 + 256)) || (rc = ensure_sticky_flag(w.sbp_batch_overflow)))
        return rc;
    SbpBatch B{};
    B.kps = d_kps; B.desc = d_desc; B.n = d_n; B.capacity = capacity;
    B.queries = reinterpret_cast<const SbpQuery*>(d_queries); B.qdesc = d_qdesc; B.nq = d_nq; B.qcapacity = qcapacity;
    B.taken = d_taken; B.q_observed = d_q_observed; B.q_angle = d_q_angle;
    B.row_rank = w.csr_idx.as<uint16_t>(); B.row_dist = w.csr_dist.as<uint8_t>(); B.row_cnt = w.csr_cnt.as<int32_t>(); B.row_stride = stride;
    B.best_idx = d_best_idx; B.best_dist = d_best_dist; B.best_level = d_best_level; B.second_dist = d_second_dist; B.second_level = d_second_level;
    B.match = d_match; B.nmatches = d_nmatches; B.overflow = w.sbp_batch_overflow.as<int32_t>();
    B.match_cur = d_match_cur; B.qbin = w.scratch.as<int32_t>();
    if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(&k_search_by_projection_batch), lds))) return rc;
    hipLaunchKernelGGL(k_search_by_projection_batch, dim3(nframes), dim3(SBP_THREADS), lds, s, B, ncap, frame_bounds(cols, rows, bounds), mode,
                       th_high, nnratio, factor, check_orientation);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_fuse_search_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_n, int capacity, int nkf, int cols, int rows,
                                   const float* bounds, const float* d_p3Dw, const uint8_t* d_valid, const float* d_min_dist,
                                   const float* d_max_dist, const float* d_normal, const uint8_t* d_mp_desc, int nmp, const float* Tcw,
                                   const float* Ow, const float* K4, const float* scale_factors, const float* inv_level_sigma2, int nlevels,
                                   float log_scale_factor, float th, double chi2, int32_t* d_best_idx, int32_t* d_best_dist, void* stream)
{
    if (!d_kps || !d_desc || !d_n || capacity <= 0 || nkf <= 0 || nmp <= 0 || !d_p3Dw || !d_min_dist || !d_max_dist || !d_normal || !d_mp_desc ||
        !Tcw || !Ow || !d_best_idx || !d_best_dist || (chi2 > 0.0 && !inv_level_sigma2))
        return fail(ORBFE_ERR_INVALID, "orbfe_fuse_search_batch_device: invalid argument");
    if (capacity > 65535) return fail(ORBFE_ERR_INVALID, "more than 65535 keypoints per frame are unsupported");
    int ncap = 64;
    while (ncap < capacity) ncap <<= 1;
    const size_t lds = (size_t)ncap * (4 + 8 + 1 + 1) + (SBP_CELLS + 2) * 2 + 64;
    if (lds > 150 * 1024) return fail(ORBFE_ERR_CAPACITY, "%d keypoints do not fit the grid kernel's LDS", capacity);
    hipStream_t s = (hipStream_t)stream;
    MatchWorkspace& w = ws(s);
    const int stride = std::max(w.sbp_stride, 128);
    w.sbp_stride = stride;
    const size_t NQ = (size_t)nkf * nmp;
    int rc;
    // obest: [best_level | second_dist | second_level | match] x NQ, then nq[nkf], nmatches[nkf]
    if ((rc = w.q.ensure(NQ * sizeof(orbfe_window_query))) || (rc = w.csr_idx.ensure(NQ * stride * 2)) || (rc = w.csr_dist.ensure(NQ * stride)) ||
        (rc = w.csr_cnt.ensure(NQ * 4)) || (rc = w.obest.ensure((NQ * 4 + 2 * (size_t)nkf) * 4 + 256)) || (rc = ensure_sticky_flag(w.sbp_batch_overflow)))
        return rc;
    for (int k = 0; k < nkf; k++) { // the projection and the gates of :848-915, one small launch per keyframe pose
        ProjectParams P;
        if ((rc = project_params(P, Tcw + 12 * k, Ow + 3 * k, K4, cols, rows, bounds, 1, scale_factors, nlevels, log_scale_factor, th, 1, 0,
                                 "orbfe_fuse_search_batch_device")))
            return rc;
        hipLaunchKernelGGL(k_project_map_points, dim3((nmp + 255) / 256), dim3(256), 0, s, d_p3Dw, d_valid ? d_valid + (size_t)k * nmp : nullptr,
                           d_min_dist, d_max_dist, d_normal, nmp, P, w.q.as<SbpQuery>() + (size_t)k * nmp);
    }
    int32_t* o = w.obest.as<int32_t>();
    int32_t* d_nq = o + 4 * NQ;
    ORBFE_HIP(hipMemsetD32Async((hipDeviceptr_t)d_nq, nmp, nkf, s));
    SbpBatch B{};
    B.kps = d_kps; B.desc = d_desc; B.n = d_n; B.capacity = capacity;
    B.queries = w.q.as<SbpQuery>(); B.qdesc = d_mp_desc; B.qdesc_shared = 1; B.nq = d_nq; B.qcapacity = nmp;
    B.row_rank = w.csr_idx.as<uint16_t>(); B.row_dist = w.csr_dist.as<uint8_t>(); B.row_cnt = w.csr_cnt.as<int32_t>(); B.row_stride = stride;
    B.best_idx = d_best_idx; B.best_dist = d_best_dist; B.best_level = o; B.second_dist = o + NQ; B.second_level = o + 2 * NQ;
    B.match = o + 3 * NQ; B.nmatches = d_nq + nkf; B.overflow = w.sbp_batch_overflow.as<int32_t>();
    B.chi2 = chi2 > 0.0 ? chi2 : 0.0;
    if (chi2 > 0.0)
        for (int l = 0; l < 16; l++) B.inv_sigma2[l] = inv_level_sigma2[std::min(l, nlevels - 1)];
    if ((rc = ensure_dyn_lds(reinterpret_cast<const void*>(&k_search_by_projection_batch), lds))) return rc;
    hipLaunchKernelGGL(k_search_by_projection_batch, dim3(nkf), dim3(SBP_THREADS), lds, s, B, ncap, frame_bounds(cols, rows, bounds), 0, 256, 0.0f,
                       0.0f, 0);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_search_by_projection_batch_status(void* stream, int32_t* overflow)
{
    if (!overflow) return fail(ORBFE_ERR_INVALID, "orbfe_search_by_projection_batch_status: null argument");
    *overflow = 0;
    hipStream_t s = (hipStream_t)stream;
    MatchWorkspace& w = ws(s);
    if (!w.sbp_batch_overflow.p) return ORBFE_OK; // no batch on this (thread, device, stream) yet
    ORBFE_HIP(hipStreamSynchronize(s));
    ORBFE_HIP(hipMemcpy(overflow, w.sbp_batch_overflow.p, 4, hipMemcpyDeviceToHost));
    if (*overflow) ORBFE_HIP(hipMemset(w.sbp_batch_overflow.p, 0, 4)); // covers every batch since it was last read
    if (*overflow > w.sbp_stride) w.sbp_stride = (*overflow + 63) / 64 * 64; // the next batch on this stream has the room
    return ORBFE_OK;
}

int orbfe_release_stream_scratch(void* stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(ORBFE_ERR_NO_DEVICE, "orbfe_release_stream_scratch: no HIP device (this library has no CPU fallback)");
    hipStream_t s = (hipStream_t)stream;
    ORBFE_HIP(hipStreamSynchronize(s));
    tl_ws.release(s);
    return ORBFE_OK;
}

int orbfe_knn2_csr(const uint8_t* Q, int nq, const uint8_t* T, int nt, const int32_t* offsets, const int32_t* idx, int init,
                   int32_t* best_idx, int32_t* best_dist, int32_t* second_dist, int device)
{
    if (nq < 0 || nt < 0 || (nq && (!Q || !offsets || !best_idx || !best_dist || !second_dist)))
        return fail(ORBFE_ERR_INVALID, "orbfe_knn2_csr: invalid argument");
    int rc = use_device(device);
    if (rc) return rc;
    if (nq == 0) return ORBFE_OK;
    const int total = offsets[nq];
    if (offsets[0] != 0 || total < 0 || (total && (!idx || !T))) return fail(ORBFE_ERR_INVALID, "orbfe_knn2_csr: bad candidate lists");
    for (int q = 0; q < nq; q++)
        if (offsets[q + 1] < offsets[q]) return fail(ORBFE_ERR_INVALID, "orbfe_knn2_csr: offsets must be non-decreasing");
    for (int k = 0; k < total; k++)
        if (idx[k] < 0 || idx[k] >= nt) return fail(ORBFE_ERR_INVALID, "orbfe_knn2_csr: candidate %d out of range", idx[k]);
    MatchWorkspace& w = ws();
    if ((rc = w.q.ensure((size_t)nq * 32)) || (rc = w.t.ensure((size_t)std::max(nt, 1) * 32)) ||
        (rc = w.csr_cnt.ensure((size_t)(nq + 1) * 4)) || (rc = w.scratch.ensure((size_t)std::max(total, 1) * 4)) ||
        (rc = w.oidx.ensure((size_t)nq * 4)) || (rc = w.obest.ensure((size_t)nq * 4)) || (rc = w.osecond.ensure((size_t)nq * 4)))
        return rc;
    ORBFE_HIP(hipMemcpy(w.q.p, Q, (size_t)nq * 32, hipMemcpyHostToDevice));
    if (nt) ORBFE_HIP(hipMemcpy(w.t.p, T, (size_t)nt * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.csr_cnt.p, offsets, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice));
    if (total) ORBFE_HIP(hipMemcpy(w.scratch.p, idx, (size_t)total * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_knn2_csr, dim3((nq + 3) / 4), dim3(256), 0, 0, w.q.as<uint8_t>(), nq, w.t.as<uint8_t>(),
                       w.csr_cnt.as<int32_t>(), w.scratch.as<int32_t>(), init, w.oidx.as<int32_t>(), w.obest.as<int32_t>(),
                       w.osecond.as<int32_t>());
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpy(best_idx, w.oidx.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(best_dist, w.obest.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(second_dist, w.osecond.p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_distinctive_descriptors_device(const uint8_t* d_desc, const int32_t* d_offsets, int npoints, int32_t* d_best_idx, uint8_t* d_best_desc,
                                         void* stream)
{
    if (npoints < 0 || (npoints && (!d_desc || !d_offsets || !d_best_idx)))
        return fail(ORBFE_ERR_INVALID, "orbfe_distinctive_descriptors_device: invalid argument");
    if (npoints == 0) return ORBFE_OK;
    hipLaunchKernelGGL(k_distinctive, dim3((npoints + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_desc, d_offsets, npoints, d_best_idx, d_best_desc);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_distinctive_descriptors(const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx, uint8_t* best_desc, int device)
{
    if (npoints < 0 || (npoints && (!offsets || !best_idx))) return fail(ORBFE_ERR_INVALID, "orbfe_distinctive_descriptors: invalid argument");
    if (npoints == 0) return ORBFE_OK;
    if (offsets[0] != 0) return fail(ORBFE_ERR_INVALID, "orbfe_distinctive_descriptors: offsets must start at 0");
    for (int p = 0; p < npoints; p++) {
        if (offsets[p + 1] < offsets[p]) return fail(ORBFE_ERR_INVALID, "orbfe_distinctive_descriptors: offsets must be non-decreasing");
        if (offsets[p + 1] - offsets[p] > DD_MAXN)
            return fail(ORBFE_ERR_CAPACITY, "map point %d has %d observations, at most %d are supported", p, offsets[p + 1] - offsets[p], DD_MAXN);
    }
    const int total = offsets[npoints];
    if (total && !desc) return fail(ORBFE_ERR_INVALID, "orbfe_distinctive_descriptors: null descriptors");
    int rc = use_device(device);
    if (rc) return rc;
    MatchWorkspace& w = ws();
    if ((rc = w.t.ensure((size_t)std::max(total, 1) * 32)) || (rc = w.csr_cnt.ensure((size_t)(npoints + 1) * 4)) ||
        (rc = w.oidx.ensure((size_t)npoints * 4)) || (rc = w.q.ensure((size_t)npoints * 32)))
        return rc;
    if (total) ORBFE_HIP(hipMemcpy(w.t.p, desc, (size_t)total * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.csr_cnt.p, offsets, (size_t)(npoints + 1) * 4, hipMemcpyHostToDevice));
    if ((rc = orbfe_distinctive_descriptors_device(w.t.as<uint8_t>(), w.csr_cnt.as<int32_t>(), npoints, w.oidx.as<int32_t>(),
                                                   best_desc ? w.q.as<uint8_t>() : nullptr, nullptr)))
        return rc;
    ORBFE_HIP(hipMemcpy(best_idx, w.oidx.p, (size_t)npoints * 4, hipMemcpyDeviceToHost));
    if (best_desc) ORBFE_HIP(hipMemcpy(best_desc, w.q.p, (size_t)npoints * 32, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_undistort_points(const float* src, int n, const float* K4, const float* dist, int ndist, float* dst, int device)
{
    if (n < 0 || (n && (!src || !dst))) return fail(ORBFE_ERR_INVALID, "orbfe_undistort_points: invalid argument");
    UndistortParams P;
    int rc = undistort_params(K4, dist, ndist, P, "orbfe_undistort_points");
    if (rc || (rc = use_device(device))) return rc;
    if (n == 0) return ORBFE_OK;
    MatchWorkspace& w = ws();
    if ((rc = w.prev.ensure((size_t)n * 8))) return rc;
    ORBFE_HIP(hipMemcpy(w.prev.p, src, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_undistort_points, dim3((n + 255) / 256), dim3(256), 0, 0, w.prev.as<float2>(), n, P, w.prev.as<float2>());
    ORBFE_HIP(hipGetLastError());
    ORBFE_HIP(hipMemcpy(dst, w.prev.p, (size_t)n * 8, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_undistort_keypoints_batch_device(const orbfe_keypoint* d_kps, const int32_t* d_n, int capacity, int nframes,
                                           const float* K4, const float* dist, int ndist, orbfe_keypoint* d_kps_un, void* stream)
{
    if (nframes < 0 || capacity < 0 || (nframes && capacity && (!d_kps || !d_n || !d_kps_un)))
        return fail(ORBFE_ERR_INVALID, "orbfe_undistort_keypoints_batch_device: invalid argument");
    UndistortParams P;
    int rc = undistort_params(K4, dist, ndist, P, "orbfe_undistort_keypoints_batch_device");
    if (rc) return rc;
    if (nframes == 0 || capacity == 0) return ORBFE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (ndist == 0 || dist[0] == 0.0f) { // Frame.cc:359-363: mvKeysUn = mvKeys
        if (d_kps_un != d_kps)
            ORBFE_HIP(hipMemcpyAsync(d_kps_un, d_kps, (size_t)nframes * capacity * sizeof(orbfe_keypoint), hipMemcpyDeviceToDevice, s));
        return ORBFE_OK;
    }
    hipLaunchKernelGGL(k_undistort_keypoints, dim3((capacity + 255) / 256, nframes), dim3(256), 0, s, d_kps, d_n, capacity, P, d_kps_un);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_compute_image_bounds(int cols, int rows, const float* K4, const float* dist, int ndist, float* bounds, int device)
{
    if (cols <= 0 || rows <= 0 || !bounds) return fail(ORBFE_ERR_INVALID, "orbfe_compute_image_bounds: invalid argument");
    if (ndist == 0 || (dist && dist[0] == 0.0f)) { // Frame.cc:444-450
        bounds[0] = 0.0f; bounds[1] = 0.0f; bounds[2] = (float)cols; bounds[3] = (float)rows;
        return ORBFE_OK;
    }
    const float c[8] = {0.0f, 0.0f, (float)cols, 0.0f, 0.0f, (float)rows, (float)cols, (float)rows};
    float u[8];
    int rc = orbfe_undistort_points(c, 4, K4, dist, ndist, u, device);
    if (rc) return rc;
    bounds[0] = std::min(u[0], u[4]); // mnMinX = min(top-left.x, bottom-left.x)   (Frame.cc:437)
    bounds[2] = std::max(u[2], u[6]); // mnMaxX = max(top-right.x, bottom-right.x) (:438)
    bounds[1] = std::min(u[1], u[3]); // mnMinY = min(top-left.y, top-right.y)     (:439)
    bounds[3] = std::max(u[5], u[7]); // mnMaxY = max(bottom-left.y, bottom-right.y) (:440)
    return ORBFE_OK;
}

} // extern "C"
